// Exact Chamfer nearest-neighbour search with spatial pruning (sorted sweep) for gfx950.
//
// The brute-force kernel (chamfer.hip) already keeps the vector ALUs busy ~94 % of the time, so the
// only way to go faster is to evaluate fewer pairs.  This path returns bit-identical dist / idx
// (same fmaf chain per evaluated pair, same "lowest index wins ties" rule) while skipping the pairs
// that provably cannot win:
//
//  1. sort: each cloud is sorted by x inside ONE workgroup (bitonic network on 64-bit
//     (sortable x | index) keys in LDS, up to 16384 points = 128 KB) and written out as float4
//     {x, y, z, original index}, padded to a multiple of 16 with +inf sentinels;
//  2. sweep: a wave owns 64*Q consecutive SORTED queries (so it covers a narrow x interval
//     [xlo, xhi]), finds its entry point in the sorted candidates by scalar binary search and walks
//     16-candidate chunks outwards in both directions.  A direction stops at the first chunk whose
//     x-gap g to the interval satisfies fl(g*g) > max over the wave of the current best distances:
//     every candidate from there on has computed distance >= fl(dx*dx) >= fl(g*g) (rounding is
//     monotone, the y/z terms only add), so it can neither win nor tie.  The chunk body is the
//     brute-force inner loop (SGPR-fed candidates, Q queries per lane, v_min3, packed fp32);
//  3. ties: candidates are no longer visited in index order, so an equal minimum met in a later chunk
//     is resolved on the spot (wave-uniform slow path, rare): the arg-min of the recorded chunk and of
//     the tying chunk are both materialised and the lower ORIGINAL index is kept.
//
// Degenerate inputs (all points on one x, pathological clustering) simply prune nothing and cost
// about what the brute force costs plus the sort.
#include <math.h>
#include <stdlib.h>

#include "ptops_common.h"

namespace ptops_sweep {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kCH = 16;
constexpr int kMaxSort = 16384;

struct __attribute__((aligned(16))) Cand4f {
  float v[16];  // 4 x {x, y, z, idx}
};

__device__ __forceinline__ float wave_max_f32(float v) {
  float t;
#define PTOPS_DPP_STEP(ctrl, rmask)                                                            \
  t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v),        \
                                                            __builtin_bit_cast(int, v), ctrl,  \
                                                            rmask, 0xf, false));               \
  v = fmaxf(v, t);
  PTOPS_DPP_STEP(0x111, 0xf)
  PTOPS_DPP_STEP(0x112, 0xf)
  PTOPS_DPP_STEP(0x114, 0xf)
  PTOPS_DPP_STEP(0x118, 0xf)
  PTOPS_DPP_STEP(0x142, 0xa)
  PTOPS_DPP_STEP(0x143, 0xc)
#undef PTOPS_DPP_STEP
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float sqdist(float qx, float qy, float qz, float cx, float cy, float cz) {
  const float dx = cx - qx, dy = cy - qy, dz = cz - qz;
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

__device__ __forceinline__ f32x2 sqdist2(f32x2 qx, f32x2 qy, f32x2 qz, float cx, float cy, float cz) {
  const f32x2 dx = f32x2{cx, cx} - qx, dy = f32x2{cy, cy} - qy, dz = f32x2{cz, cz} - qz;
  return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
}

// ---------------------------------------------------------------------------------------- sort
// grid = (clouds of set 1) + (clouds of set 2); 1024 threads; dynamic LDS = 8 * npad bytes.
__global__ __launch_bounds__(1024) void chamfer_sort_kernel(const float* __restrict__ xyz1, int n1, int n1p,
                                                            float4* __restrict__ out1,
                                                            const float* __restrict__ xyz2, int n2, int n2p,
                                                            float4* __restrict__ out2, int B) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  const bool second = (int)blockIdx.x >= B;
  const int b = second ? blockIdx.x - B : blockIdx.x;
  const int n = second ? n2 : n1, np16 = second ? n2p : n1p;
  const float* __restrict__ src = (second ? xyz2 : xyz1) + (size_t)b * n * 3;
  float4* __restrict__ dst = (second ? out2 : out1) + (size_t)b * np16;
  int N = 1;
  while (N < n) N <<= 1;
  if (N < 2) N = 2;
  const int tid = threadIdx.x;
  for (int i = tid; i < N; i += 1024) {
    unsigned long long key = ~0ull;  // padding sorts last
    if (i < n) {
      unsigned u = __float_as_uint(src[(size_t)i * 3]);
      u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;  // order-preserving map float -> uint
      key = ((unsigned long long)u << 32) | (unsigned)i;
    }
    keys[i] = key;
  }
  __syncthreads();
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (N >> 1); t += 1024) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // j is a power of two: no division
        const int l = i + j;
        const unsigned long long a = keys[i], c = keys[l];
        const bool up = (i & k) == 0;
        if ((a > c) == up) {
          keys[i] = c;
          keys[l] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < np16; i += 1024) {
    float4 o;
    if (i < n) {
      const unsigned id = (unsigned)keys[i];
      o.x = src[(size_t)id * 3 + 0];
      o.y = src[(size_t)id * 3 + 1];
      o.z = src[(size_t)id * 3 + 2];
      o.w = __int_as_float((int)id);
    } else {  // sentinel: infinitely far, never selected
      o.x = INFINITY;
      o.y = 0.0f;
      o.z = 0.0f;
      o.w = __int_as_float(-1);
    }
    dst[i] = o;
  }
}

// --------------------------------------------------------------------------------------- sweep
// arg-min inside one 16-candidate chunk: the lowest ORIGINAL index among candidates whose distance
// equals `best` (INT_MAX if none).  Per-lane chunk position: vector loads.
__device__ __forceinline__ int resolve_chunk(const float4* __restrict__ cb, int pos, float qx, float qy,
                                             float qz, float best) {
  int found = 0x7fffffff;
#pragma unroll 4
  for (int t = 0; t < kCH; ++t) {
    const float4 c = cb[pos + t];
    const float d = sqdist(qx, qy, qz, c.x, c.y, c.z);
    const int id = __float_as_int(c.w);
    found = (d == best && id >= 0 && id < found) ? id : found;
  }
  return found;
}

template <int Q>
__global__ __launch_bounds__(256) void chamfer_sweep_kernel(const float4* __restrict__ qs, int nq, int nqp,
                                                            const float4* __restrict__ cs, int ncp,
                                                            float* __restrict__ dist, int32_t* __restrict__ idx,
                                                            unsigned long long* __restrict__ stats) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_base = (blockIdx.x * 4 + wave) * (64 * Q);
  if (wave_base >= nq) return;  // wave-uniform
  const float4* __restrict__ qb = qs + (size_t)b * nqp;
  const float4* __restrict__ cb = cs + (size_t)b * ncp;
  const int nch = ncp / kCH;

  float qx[Q], qy[Q], qz[Q], best[Q];
  int qorig[Q], bpos[Q], bidx[Q];
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    const int sp = min(wave_base + i * 64 + lane, nq - 1);
    const float4 t = qb[sp];
    qx[i] = t.x;
    qy[i] = t.y;
    qz[i] = t.z;
    qorig[i] = (wave_base + i * 64 + lane < nq) ? __float_as_int(t.w) : -1;
    best[i] = INFINITY;
    bpos[i] = 0;
    bidx[i] = -1;
  }
  // x interval of this wave's (sorted) queries
  const float xlo = qb[wave_base].x;
  const float xhi = qb[min(wave_base + 64 * Q, nq) - 1].x;
  // entry point: first candidate with x >= xlo (scalar binary search; sentinels are +inf)
  int lo = 0, hi = ncp;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cb[mid].x < xlo) lo = mid + 1; else hi = mid;
  }
  int cR = min(lo / kCH, nch - 1), cL = cR - 1;
  bool moreR = true, moreL = cL >= 0;
  const int entry = cR;

  // One chunk (16 candidates at sorted position c*16): brute-force body + tie bookkeeping.
  auto do_chunk = [&](int c) {
    const Cand4f* __restrict__ cp = reinterpret_cast<const Cand4f*>(cb + (size_t)c * kCH);
    float m[Q];  // minimum over THIS chunk only (so that m == best really is a tie, not "no change")
#pragma unroll
    for (int i = 0; i < Q; ++i) m[i] = INFINITY;
#pragma unroll
    for (int gq = 0; gq < kCH / 4; ++gq) {
      const Cand4f cc = cp[gq];  // uniform address: s_load_dwordx16
      if constexpr (Q % 2 == 0) {
#pragma unroll
        for (int i = 0; i < Q; i += 2) {
          const f32x2 x2{qx[i], qx[i + 1]}, y2{qy[i], qy[i + 1]}, z2{qz[i], qz[i + 1]};
          const f32x2 d0 = sqdist2(x2, y2, z2, cc.v[0], cc.v[1], cc.v[2]);
          const f32x2 d1 = sqdist2(x2, y2, z2, cc.v[4], cc.v[5], cc.v[6]);
          const f32x2 d2 = sqdist2(x2, y2, z2, cc.v[8], cc.v[9], cc.v[10]);
          const f32x2 d3 = sqdist2(x2, y2, z2, cc.v[12], cc.v[13], cc.v[14]);
          m[i] = __builtin_fminf(__builtin_fminf(m[i], d0.x), d1.x);
          m[i] = __builtin_fminf(__builtin_fminf(m[i], d2.x), d3.x);
          m[i + 1] = __builtin_fminf(__builtin_fminf(m[i + 1], d0.y), d1.y);
          m[i + 1] = __builtin_fminf(__builtin_fminf(m[i + 1], d2.y), d3.y);
        }
      } else {
#pragma unroll
        for (int i = 0; i < Q; ++i) {
          const float d0 = sqdist(qx[i], qy[i], qz[i], cc.v[0], cc.v[1], cc.v[2]);
          const float d1 = sqdist(qx[i], qy[i], qz[i], cc.v[4], cc.v[5], cc.v[6]);
          const float d2 = sqdist(qx[i], qy[i], qz[i], cc.v[8], cc.v[9], cc.v[10]);
          const float d3 = sqdist(qx[i], qy[i], qz[i], cc.v[12], cc.v[13], cc.v[14]);
          m[i] = __builtin_fminf(__builtin_fminf(m[i], d0), d1);
          m[i] = __builtin_fminf(__builtin_fminf(m[i], d2), d3);
        }
      }
    }
    bool any_tie = false;
    bool tie[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      const bool lower = m[i] < best[i];
      tie[i] = m[i] == best[i] && best[i] < INFINITY;
      any_tie |= tie[i];
      best[i] = lower ? m[i] : best[i];
      bpos[i] = lower ? c * kCH : bpos[i];
      bidx[i] = lower ? -1 : bidx[i];
    }
    if (__any(any_tie)) {  // an equal minimum in another chunk: settle the index now
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        if (tie[i]) {
          if (bidx[i] < 0) bidx[i] = resolve_chunk(cb, bpos[i], qx[i], qy[i], qz[i], best[i]);
          const int other = resolve_chunk(cb, c * kCH, qx[i], qy[i], qz[i], best[i]);
          bidx[i] = other < bidx[i] ? other : bidx[i];
        }
      }
    }
  };

  // The prune test is made once per group of kGroup chunks and side, so that its cost (a wave
  // reduction and two dependent scalar loads) stays a few per cent even when nothing can be pruned.
  constexpr int kGroup = 4;
  while (moreR || moreL) {
    float wm = best[0];
#pragma unroll
    for (int i = 1; i < Q; ++i) wm = fmaxf(wm, best[i]);
    const float wmax = wave_max_f32(wm);  // +inf until every lane has a finite candidate
    if (moreR) {
      const float g = cb[cR * kCH].x - xhi;  // every candidate from here on has x >= this edge
      if (g > 0.0f && g * g > wmax) {
        moreR = false;
      } else {
        const int stop = min(cR + kGroup, nch);
        for (; cR < stop; ++cR) do_chunk(cR);
        moreR = cR < nch;
      }
    }
    if (moreL) {
      const float g = xlo - cb[cL * kCH + kCH - 1].x;  // every candidate from here down has x <= this edge
      if (g > 0.0f && g * g > wmax) {
        moreL = false;
      } else {
        const int stop = max(cL - kGroup, -1);
        for (; cL > stop; --cL) do_chunk(cL);
        moreL = cL >= 0;
      }
    }
  }

  // pruning statistics for the caller's path selection: chunks this wave evaluated / chunks in total
  if (stats != nullptr && lane == 0) {
    atomicAdd(&stats[0], (unsigned long long)((cR - entry) + (entry - 1 - cL)));
    atomicAdd(&stats[1], (unsigned long long)nch);
  }
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    if (qorig[i] < 0) continue;
    int id = bidx[i];
    if (id < 0) id = resolve_chunk(cb, bpos[i], qx[i], qy[i], qz[i], best[i]);
    if (id == 0x7fffffff) id = 0;  // only with non-finite inputs
    const size_t o = (size_t)b * nq + qorig[i];
    dist[o] = best[i];
    idx[o] = id;
  }
}

static int pad16(int n) { return (n + kCH - 1) / kCH * kCH; }

bool eligible(int B, int n, int m) {
  // both clouds sortable in one workgroup and large enough for the sort to pay off (measured on
  // MI355X, B=32: 2048x2048 0.113 ms swept vs 0.047 ms brute force; 16384x16384 0.69 vs 1.85 ms)
  return n >= 4096 && m >= 4096 && n <= kMaxSort && m <= kMaxSort && B >= 1;
}

size_t workspace_bytes(int B, int n, int m) {
  return sizeof(float4) * (size_t)B * ((size_t)pad16(n) + pad16(m));
}

static int ptops_sweep_q_override() {
#ifdef PTOPS_TUNING
  const char* v = getenv("PTOPS_TUNE_SWEEP_Q");
  return v ? atoi(v) : 0;
#else
  return 0;
#endif
}

template <int Q>
void launch_sweep(const float4* qs, int nq, int nqp, const float4* cs, int ncp, int B, float* dist,
                  int32_t* idx, unsigned long long* stats, hipStream_t stream) {
  dim3 grid(ptops::ceil_div(nq, 256 * Q), B);
  hipLaunchKernelGGL(chamfer_sweep_kernel<Q>, grid, dim3(256), 0, stream, qs, nq, nqp, cs, ncp, dist, idx, stats);
}

// dirmask bit 0: sweep xyz1 -> xyz2 (dist1/idx1), bit 1: xyz2 -> xyz1.  Both clouds are sorted either way.
int forward(const float* xyz1, const float* xyz2, int B, int n, int m, float* dist1, float* dist2,
            int32_t* idx1, int32_t* idx2, void* workspace, unsigned long long* stats, int dirmask,
            hipStream_t stream) {
  const int np = pad16(n), mp = pad16(m);
  float4* s1 = reinterpret_cast<float4*>(workspace);
  float4* s2 = s1 + (size_t)B * np;
  int N = 1;
  while (N < (n > m ? n : m)) N <<= 1;
  const size_t lds = sizeof(unsigned long long) * (size_t)N;
  // 128 KB of dynamic LDS for 16384 keys: above the 64 KB default cap (idempotent per-function setting)
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chamfer_sort_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 8 * kMaxSort);
  hipLaunchKernelGGL(chamfer_sort_kernel, dim3(2 * B), dim3(1024), lds, stream, xyz1, n, np, s1, xyz2, m, mp, s2, B);
  auto sweep = [&](const float4* qs, int nq, int nqp, const float4* cs, int ncp, float* d, int32_t* id,
                   unsigned long long* stats) {
    const int q = ptops_sweep_q_override() > 0 ? ptops_sweep_q_override() : (nq >= 8192 ? 4 : (nq >= 2048 ? 2 : 1));
    if (q == 4) launch_sweep<4>(qs, nq, nqp, cs, ncp, B, d, id, stats, stream);
    else if (q == 2) launch_sweep<2>(qs, nq, nqp, cs, ncp, B, d, id, stats, stream);
    else launch_sweep<1>(qs, nq, nqp, cs, ncp, B, d, id, stats, stream);
  };
  if (dirmask & 1) sweep(s1, n, np, s2, mp, dist1, idx1, stats);
  if (dirmask & 2) sweep(s2, m, mp, s1, np, dist2, idx2, stats ? stats + 2 : nullptr);
  return ptops::check_launch("chamfer_sweep");
}

}  // namespace ptops_sweep
