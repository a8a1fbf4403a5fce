// Chamfer nearest-neighbour forward / backward for gfx950 (MI355X).
//
// Replaces extensions/chamfer_dist/chamfer.cu:15-229 of the reference.  Not a port:
//  * the candidate cloud is streamed through SGPRs (wave-uniform scalar loads served by
//    the scalar cache / L2) instead of an LDS tile, so the inner loop is pure VALU with
//    one SGPR operand per instruction and no LDS traffic at all;
//  * every lane register-tiles Q queries, amortising each scalar candidate fetch over
//    Q*64 pair evaluations per wave;
//  * the arg-min is not carried through the inner loop: per 16-candidate chunk only the
//    running minimum (v_min3_f32, two candidates per instruction) is updated and one
//    compare/select records the chunk that last lowered it; the winning chunk is rescanned
//    once at the end for the first candidate whose distance equals the minimum.  That is
//    6.6 VALU ops per pair instead of 9 and keeps the reference's "lowest index wins ties"
//    rule exactly (chamfer.cu:47,57,137 use strict <);
//  * both directions run in ONE launch (the grid is the concatenation of both job lists),
//    and when a direction has too few queries to fill 256 CUs its candidate range is split
//    across workgroups that combine through atomicMin on a packed (dist_bits<<32 | idx)
//    u64 — order-independent, so the result stays deterministic and tie-exact.
//
// Arithmetic (bit-exact with oracle/ptops_ref.c sqdist_direct): d = fma(dz,dz,fma(dy,dy,dx*dx)),
// d* = candidate - query.  Compiled with -ffp-contract=off; all fmas are explicit.
#include <math.h>
#include <stdlib.h>

#include "ptops_common.h"

namespace {

constexpr int kTPB = 256;  // threads per workgroup (4 waves)
constexpr int kCH = 16;    // candidates per bookkeeping chunk

struct __attribute__((aligned(4))) Cand4 {
  float v[12];  // x0 y0 z0 x1 y1 z1 x2 y2 z2 x3 y3 z3
};

struct Dir {
  const float* q;      // [B,nq,3] queries
  const float* c;      // [B,nc,3] candidates
  float* dist;         // [B,nq]
  int32_t* idx;        // [B,nq]
  unsigned long long* ws;  // [B,nq] packed minima (splits > 1 only)
  int nq, nc;
  int qtiles;          // ceil(nq / (kTPB*Q))
  int splits;          // candidate ranges per query tile
  int cand_per_split;  // multiple of kCH
  int first_block;     // first blockIdx.x of this direction
};

struct Args {
  Dir d[2];
};

__device__ __forceinline__ float sqdist(float qx, float qy, float qz, float cx, float cy,
                                        float cz) {
  float dx = cx - qx, dy = cy - qy, dz = cz - qz;
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Two queries per instruction: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 (same IEEE operations,
// so the result is bit-identical to sqdist()).
__device__ __forceinline__ f32x2 sqdist2(f32x2 qx, f32x2 qy, f32x2 qz, float cx, float cy, float cz) {
  const f32x2 dx = f32x2{cx, cx} - qx, dy = f32x2{cy, cy} - qy, dz = f32x2{cz, cz} - qz;
  return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
}

template <int Q, bool PK>
__global__ __launch_bounds__(kTPB) void chamfer_nn_kernel(Args a) {
  const int bid = blockIdx.x;
  const bool second = bid >= a.d[1].first_block;
  // Field-wise selects keep the descriptor in SGPRs (no dynamic kernarg indexing).
  const float* __restrict__ qptr = second ? a.d[1].q : a.d[0].q;
  const float* __restrict__ cptr = second ? a.d[1].c : a.d[0].c;
  float* __restrict__ dist = second ? a.d[1].dist : a.d[0].dist;
  int32_t* __restrict__ idx = second ? a.d[1].idx : a.d[0].idx;
  unsigned long long* __restrict__ ws = second ? a.d[1].ws : a.d[0].ws;
  const int nq = second ? a.d[1].nq : a.d[0].nq;
  const int nc = second ? a.d[1].nc : a.d[0].nc;
  const int qtiles = second ? a.d[1].qtiles : a.d[0].qtiles;
  const int splits = second ? a.d[1].splits : a.d[0].splits;
  const int cps = second ? a.d[1].cand_per_split : a.d[0].cand_per_split;
  int local = bid - (second ? a.d[1].first_block : a.d[0].first_block);

  const int s = local % splits;
  local /= splits;
  const int qt = local % qtiles;
  const int b = local / qtiles;

  const float* __restrict__ qb = qptr + (size_t)b * nq * 3;
  const float* __restrict__ cb = cptr + (size_t)b * nc * 3;
  const int c_begin = s * cps;
  const int c_end = min(nc, c_begin + cps);

  float qx[Q], qy[Q], qz[Q], best[Q];
  int bch[Q];
  const int j0 = qt * (kTPB * Q) + threadIdx.x;
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    int j = min(j0 + i * kTPB, nq - 1);
    qx[i] = qb[j * 3 + 0];
    qy[i] = qb[j * 3 + 1];
    qz[i] = qb[j * 3 + 2];
    best[i] = INFINITY;
    bch[i] = c_begin;
  }

  const int full_end = c_begin + ((c_end - c_begin) / kCH) * kCH;
  // Running scalar pointer (s_add_u32/s_addc_u32 per chunk, no VALU address math).
  const Cand4* __restrict__ cp = reinterpret_cast<const Cand4*>(cb + (size_t)c_begin * 3);
  for (int k = c_begin; k < full_end; k += kCH, cp += kCH / 4) {
    float m[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) m[i] = best[i];
#pragma unroll
    for (int g = 0; g < kCH / 4; ++g) {
      // Uniform address -> s_load_dwordx4/x8: the candidates live in SGPRs.
      const Cand4 cc = cp[g];
      if constexpr (PK && (Q % 2 == 0)) {
#pragma unroll
        for (int i = 0; i < Q; i += 2) {
          const f32x2 x2{qx[i], qx[i + 1]}, y2{qy[i], qy[i + 1]}, z2{qz[i], qz[i + 1]};
          const f32x2 d0 = sqdist2(x2, y2, z2, cc.v[0], cc.v[1], cc.v[2]);
          const f32x2 d1 = sqdist2(x2, y2, z2, cc.v[3], cc.v[4], cc.v[5]);
          const f32x2 d2 = sqdist2(x2, y2, z2, cc.v[6], cc.v[7], cc.v[8]);
          const f32x2 d3 = sqdist2(x2, y2, z2, cc.v[9], cc.v[10], cc.v[11]);
          m[i] = __builtin_fminf(__builtin_fminf(m[i], d0.x), d1.x);
          m[i] = __builtin_fminf(__builtin_fminf(m[i], d2.x), d3.x);
          m[i + 1] = __builtin_fminf(__builtin_fminf(m[i + 1], d0.y), d1.y);
          m[i + 1] = __builtin_fminf(__builtin_fminf(m[i + 1], d2.y), d3.y);
        }
      } else {
#pragma unroll
        for (int i = 0; i < Q; ++i) {
          float d0 = sqdist(qx[i], qy[i], qz[i], cc.v[0], cc.v[1], cc.v[2]);
          float d1 = sqdist(qx[i], qy[i], qz[i], cc.v[3], cc.v[4], cc.v[5]);
          float d2 = sqdist(qx[i], qy[i], qz[i], cc.v[6], cc.v[7], cc.v[8]);
          float d3 = sqdist(qx[i], qy[i], qz[i], cc.v[9], cc.v[10], cc.v[11]);
          m[i] = __builtin_fminf(__builtin_fminf(m[i], d0), d1);
          m[i] = __builtin_fminf(__builtin_fminf(m[i], d2), d3);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      bool lower = m[i] < best[i];
      best[i] = lower ? m[i] : best[i];
      bch[i] = lower ? k : bch[i];
    }
  }
  if (full_end < c_end) {  // ragged tail chunk (< kCH candidates)
    float m[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) m[i] = best[i];
    for (int k = full_end; k < c_end; ++k) {
      const float cx = cb[(size_t)k * 3 + 0], cy = cb[(size_t)k * 3 + 1],
                  cz = cb[(size_t)k * 3 + 2];
#pragma unroll
      for (int i = 0; i < Q; ++i)
        m[i] = __builtin_fminf(m[i], sqdist(qx[i], qy[i], qz[i], cx, cy, cz));
    }
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      bool lower = m[i] < best[i];
      best[i] = lower ? m[i] : best[i];
      bch[i] = lower ? full_end : bch[i];
    }
  }

  // Resolve the arg-min inside the winning chunk: lowest k with d == best.
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    const int k0 = bch[i];
    int found = k0;
#pragma unroll
    for (int t = kCH - 1; t >= 0; --t) {
      const int k = min(k0 + t, c_end - 1);
      const float d = sqdist(qx[i], qy[i], qz[i], cb[(size_t)k * 3 + 0], cb[(size_t)k * 3 + 1],
                             cb[(size_t)k * 3 + 2]);
      found = (d == best[i]) ? k : found;
    }
    const int j = j0 + i * kTPB;
    if (j < nq) {
      const size_t o = (size_t)b * nq + j;
      if (splits == 1) {
        dist[o] = best[i];
        idx[o] = found;
      } else {
        // best >= +0, so its bit pattern orders like the float; low word breaks ties by index.
        unsigned long long key =
            ((unsigned long long)__float_as_uint(best[i]) << 32) | (unsigned int)found;
        atomicMin(&ws[o], key);
      }
    }
  }
}

__global__ __launch_bounds__(256) void chamfer_unpack_kernel(const unsigned long long* __restrict__ ws,
                                                             float* __restrict__ dist,
                                                             int32_t* __restrict__ idx, long long total) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
    unsigned long long key = ws[i];
    dist[i] = __uint_as_float((unsigned int)(key >> 32));
    idx[i] = (int32_t)(unsigned int)key;
  }
}

// One lane per point of either cloud; chamfer.cu:173-201 adds, for every point p with nearest neighbour o and
// incoming gradient g,  +2g(p - o) to p's gradient and -2g(p - o) to o's (atomicAdd into zeroed arrays).  Every
// point receives exactly one "own" term, so that half is a plain store in a first pass (which also replaces the
// memset) and only the neighbour half uses atomics in a second pass.
template <bool NEIGHBOUR>
__global__ __launch_bounds__(256) void chamfer_grad_kernel(
    const float* __restrict__ xyz1, const float* __restrict__ xyz2, const int32_t* __restrict__ idx1,
    const int32_t* __restrict__ idx2, const float* __restrict__ gd1, const float* __restrict__ gd2,
    int B, int n, int m, float* __restrict__ g1, float* __restrict__ g2) {
  const long long t1 = (long long)B * n, total = t1 + (long long)B * m;
  for (long long t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const bool second = t >= t1;
    const long long e = second ? t - t1 : t;
    const int np = second ? m : n, no = second ? n : m;
    const int b = (int)(e / np), j = (int)(e % np);
    const float* p = (second ? xyz2 : xyz1) + ((size_t)b * np + j) * 3;
    const int j2 = (second ? idx2 : idx1)[e];
    const float* o = (second ? xyz1 : xyz2) + ((size_t)b * no + j2) * 3;
    float* gp = (second ? g2 : g1) + ((size_t)b * np + j) * 3;
    float* go = (second ? g1 : g2) + ((size_t)b * no + j2) * 3;
    const float g = (second ? gd2 : gd1)[e] * 2.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float v = g * (p[d] - o[d]);
      if (NEIGHBOUR) atomicAdd(go + d, -v);
      else gp[d] = v;
    }
  }
}

// ---- deterministic backward (round 3): the neighbour half without float atomics ------------------------------------
// The reference (chamfer.cu:193-198) and chamfer_grad_kernel<true> add the neighbour terms with atomicAdd, i.e. in
// arrival order: gradients differ in the last bits from run to run.  Here the nearest-neighbour map is transposed once
// per call (for every point the list of points of the other cloud that chose it) and every point sums its own term and
// its list in a way that does not depend on timing:
//   * a list of up to kCsrShort (8) entries is put in ascending order and added one after the other, in exactly the order
//     of the scalar loops of the CPU oracle (bit-equal to it);
//   * a longer list (a prediction collapsed onto few target points sends thousands of points to one neighbour - the
//     randomly initialised model of the benchmark does) stays unordered and is summed by a whole wave in 64-bit FIXED
//     POINT, whose integer additions commute: same bits on every run, accurate to 2^-40 of the largest term.
//
// chamfer_csr_kernel: one workgroup per (cloud, direction): counting sort in LDS (integer atomics decide slots, never
// sums), short rows ranked into ascending order.
constexpr int kCsrT = 1024;
constexpr int kCsrShort = 8;    // lists up to this length are ranked (c^2 loads) and added in the oracle's order
constexpr int kCsrMid = 32;     // lists up to this length are summed by their own lane (fixed point), longer ones by the wave
constexpr int kCsrAgg = 6;      // leaders tried per wave-instruction before the remaining lanes add on their own

__global__ __launch_bounds__(kCsrT) void chamfer_csr_kernel(const int32_t* __restrict__ idx1, const int32_t* __restrict__ idx2,
                                                            int n, int m, int32_t* __restrict__ rowptr1, int32_t* __restrict__ tmp1,
                                                            int32_t* __restrict__ col1, int32_t* __restrict__ rowptr2,
                                                            int32_t* __restrict__ tmp2, int32_t* __restrict__ col2) {
  extern __shared__ __attribute__((aligned(16))) int csm[];   // cnt[nc] | cur[nc] | wtot[16]
  const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int nq = dir ? m : n, nc = dir ? n : m;               // nq entries (points that chose), nc rows (points chosen)
  const int32_t* __restrict__ ib = (dir ? idx2 : idx1) + (size_t)b * nq;
  int32_t* __restrict__ rp = (dir ? rowptr2 : rowptr1) + (size_t)b * (nc + 1);
  int32_t* __restrict__ tp = (dir ? tmp2 : tmp1) + (size_t)b * nq;
  int32_t* __restrict__ cl = (dir ? col2 : col1) + (size_t)b * nq;
  int* cnt = csm;
  int* cur = cnt + nc;
  int* wtot = cur + nc;
  for (int r = tid; r < nc; r += kCsrT) cnt[r] = 0;
  __syncthreads();
  // Counting with wave-aggregated atomics: lanes of a wave that hold the same row elect a leader, which adds their
  // number once.  A collapsed prediction sends thousands of entries to a few rows, and 16384 LDS atomics on a handful of
  // addresses serialise (0.8 ms on the benchmark's 16384^2 call without this).  At most kCsrAgg leaders are tried per
  // wave-instruction - enough when a few rows dominate, a small fixed cost when all rows differ - and the lanes that
  // are left add on their own.
  for (int e0 = 0; e0 < nq; e0 += kCsrT) {
    const int e = e0 + tid;
    const bool valid = e < nq;
    const int r = valid ? ib[e] : -1;
    unsigned long long todo = __ballot(valid);
    bool mine = valid;
    for (int it = 0; it < kCsrAgg && todo; ++it) {
      const int lead = __ffsll((long long)todo) - 1;
      const int rl = __shfl(r, lead);
      const unsigned long long same = __ballot(mine && r == rl);
      if (lane == lead) atomicAdd(&cnt[rl], (int)__popcll(same));
      mine = mine && r != rl;
      todo &= ~same;
    }
    if (mine) atomicAdd(&cnt[r], 1);
  }
  __syncthreads();
  // exclusive scan of cnt: thread t owns rows [t*per, (t+1)*per); wave scan of the per-thread totals, then the 16 wave totals
  const int per = (nc + kCsrT - 1) / kCsrT;
  int local = 0;
  for (int i = 0; i < per; ++i) {
    const int r = tid * per + i;
    if (r < nc) local += cnt[r];
  }
  int incl = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int run = incl - local;
  for (int w = 0; w < wave; ++w) run += wtot[w];
  for (int i = 0; i < per; ++i) {
    const int r = tid * per + i;
    if (r < nc) {
      cur[r] = run;
      rp[r] = run;
      run += cnt[r];
    }
  }
  if (tid == kCsrT - 1) rp[nc] = run;
  __syncthreads();
  for (int e0 = 0; e0 < nq; e0 += kCsrT) {                    // unordered fill (same aggregation): long rows straight
    const int e = e0 + tid;                                   // into col, short rows into tmp (ranked below)
    const bool valid = e < nq;
    const int r = valid ? ib[e] : -1;
    int pos = 0;
    unsigned long long todo = __ballot(valid);
    bool mine = valid;
    for (int it = 0; it < kCsrAgg && todo; ++it) {
      const int lead = __ffsll((long long)todo) - 1;
      const int rl = __shfl(r, lead);
      const unsigned long long same = __ballot(mine && r == rl);
      int base = 0;
      if (lane == lead) base = atomicAdd(&cur[rl], (int)__popcll(same));
      base = __shfl(base, lead);
      if (mine && r == rl) pos = base + (int)__popcll(same & ((1ull << lane) - 1ull));
      mine = mine && r != rl;
      todo &= ~same;
    }
    if (mine) pos = atomicAdd(&cur[r], 1);
    if (valid) (cnt[r] <= kCsrShort ? tp : cl)[pos] = e;
  }
  __syncthreads();                                            // (workgroup-scope fence: tmp is re-read by other threads below)
  for (int r = tid; r < nc; r += kCsrT) {                     // short rows: every entry to the slot of its rank
    const int c = cnt[r];
    if (c == 0 || c > kCsrShort) continue;
    const int beg = cur[r] - c;
    if (c == 1) {
      cl[beg] = tp[beg];
      continue;
    }
    for (int i = 0; i < c; ++i) {
      const int v = tp[beg + i];
      int rank = 0;
      for (int j = 0; j < c; ++j) rank += tp[beg + j] < v;
      cl[beg + rank] = v;
    }
  }
}

// One lane per point of either cloud.  Short rows: the additions happen in exactly the order of the scalar loops of the
// CPU oracle (oracle/ptops_ref.c: pass over cloud 1, then pass over cloud 2, each adding its own term to the point and the
// negated term to the neighbour) - a point of cloud 1 = own term, then the terms of the cloud-2 points that chose it by
// ascending index; a point of cloud 2 = those terms first, then its own.  Long rows: the whole wave, fixed point.
__global__ __launch_bounds__(256) void chamfer_grad_det_kernel(
    const float* __restrict__ xyz1, const float* __restrict__ xyz2, const int32_t* __restrict__ idx1,
    const int32_t* __restrict__ idx2, const float* __restrict__ gd1, const float* __restrict__ gd2, int B, int n, int m,
    const int32_t* __restrict__ rowptr1, const int32_t* __restrict__ col1, const int32_t* __restrict__ rowptr2,
    const int32_t* __restrict__ col2, float* __restrict__ g1, float* __restrict__ g2) {
  const long long t1 = (long long)B * n, total = t1 + (long long)B * m;
  const long long t = blockIdx.x * 256LL + threadIdx.x;
  const bool live = t < total;
  const bool second = live && t >= t1;
  const long long e = live ? (second ? t - t1 : t) : 0;
  const int np = second ? m : n, no = second ? n : m;
  const int b = (int)(e / np), j = (int)(e % np);
  const float* pb = (second ? xyz2 : xyz1) + (size_t)b * np * 3;     // my cloud
  const float* ob = (second ? xyz1 : xyz2) + (size_t)b * no * 3;     // the other cloud
  const float* god = (second ? gd1 : gd2) + (size_t)b * no;          // incoming gradients of the other cloud's distances
  // rows indexed by MY cloud's points hold the points of the other cloud that chose them: built from the other
  // direction's index array (rowptr1 / col1: rows = cloud 2, entries = cloud 1; rowptr2 / col2: rows = cloud 1)
  const int32_t* rp = (second ? rowptr1 : rowptr2) + (size_t)b * (np + 1);
  const int32_t* cl = (second ? col1 : col2) + (size_t)b * no;
  float px = 0.f, py = 0.f, pz = 0.f, ox = 0.f, oy = 0.f, oz = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
  int beg = 0, cnt = 0;
  if (live) {
    px = pb[(size_t)j * 3], py = pb[(size_t)j * 3 + 1], pz = pb[(size_t)j * 3 + 2];
    const int j2 = (second ? idx2 : idx1)[e];
    const float g = (second ? gd2 : gd1)[e] * 2.0f;
    ox = g * (px - ob[(size_t)j2 * 3]), oy = g * (py - ob[(size_t)j2 * 3 + 1]), oz = g * (pz - ob[(size_t)j2 * 3 + 2]);
    beg = rp[j];
    cnt = rp[j + 1] - beg;
    if (!second) ax += ox, ay += oy, az += oz;                     // cloud 1: the own term is added first
    if (cnt <= kCsrShort) {
      for (int i = 0; i < cnt; ++i) {
        const int k = cl[beg + i];
        const float gk = god[k] * 2.0f;
        ax += -(gk * (ob[(size_t)k * 3] - px));
        ay += -(gk * (ob[(size_t)k * 3 + 1] - py));
        az += -(gk * (ob[(size_t)k * 3 + 2] - pz));
      }
    } else if (cnt <= kCsrMid) {
      // medium lists (a 512-point prediction against 16384 targets: ~32 entries per point): the lane walks its own list,
      // unordered, in fixed point - every lane of the wave busy with a list of its own
      float mx = 0.0f;
      bool bad = false;
      for (int i = 0; i < cnt; ++i) {
        const int k = cl[beg + i];
        const float gk = god[k] * 2.0f;
        const float a0 = fabsf(gk * (ob[(size_t)k * 3] - px)), a1 = fabsf(gk * (ob[(size_t)k * 3 + 1] - py)),
                    a2 = fabsf(gk * (ob[(size_t)k * 3 + 2] - pz));
        const bool fin = (a0 < INFINITY) && (a1 < INFINITY) && (a2 < INFINITY);
        mx = fin ? fmaxf(mx, fmaxf(a0, fmaxf(a1, a2))) : mx;
        bad |= !fin;
      }
      int e2 = 0;
      (void)frexpf(mx > 0.0f ? mx : 1.0f, &e2);
      const double scale = ldexp(1.0, 46 - e2), inv = ldexp(1.0, e2 - 46);
      long long sx = 0, sy = 0, sz = 0;
      float fx = 0.f, fy = 0.f, fz = 0.f;
      for (int i = 0; i < cnt; ++i) {
        const int k = cl[beg + i];
        const float gk = god[k] * 2.0f;
        const float tx = -(gk * (ob[(size_t)k * 3] - px)), ty = -(gk * (ob[(size_t)k * 3 + 1] - py)),
                    tz = -(gk * (ob[(size_t)k * 3 + 2] - pz));
        if (fabsf(tx) < INFINITY) sx += __double2ll_rn((double)tx * scale); else fx += tx;
        if (fabsf(ty) < INFINITY) sy += __double2ll_rn((double)ty * scale); else fy += ty;
        if (fabsf(tz) < INFINITY) sz += __double2ll_rn((double)tz * scale); else fz += tz;
      }
      ax += (float)((double)sx * inv) + (bad ? fx : 0.0f);
      ay += (float)((double)sy * inv) + (bad ? fy : 0.0f);
      az += (float)((double)sz * inv) + (bad ? fz : 0.0f);
    }
  }
  unsigned long long todo = __ballot(live && cnt > kCsrMid);
  const int lane = threadIdx.x & 63;
  while (todo) {                                                   // wave-uniform
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int rb = __shfl(beg, src), rc = __shfl(cnt, src);
    const float qx = __shfl(px, src), qy = __shfl(py, src), qz = __shfl(pz, src);
    const int sb = __shfl(b, src), ssecond = __shfl((int)second, src);
    const int sno = ssecond ? n : m;
    const float* sob = (ssecond ? xyz1 : xyz2) + (size_t)sb * sno * 3;
    const float* sgod = (ssecond ? gd1 : gd2) + (size_t)sb * sno;
    const int32_t* scl = (ssecond ? col1 : col2) + (size_t)sb * sno;
    // pass 1: the largest |term| of the row fixes the scale (a pure function of the row's contents, not of its order)
    float mx = 0.0f;
    int bad = 0;
    // four entries per lane and iteration: their (dependent) index -> gradient / coordinate loads overlap
    for (int i0 = lane; i0 < rc; i0 += 256) {
      int kk[4];
      float gg[4], cx[4], cy[4], cz[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) kk[u] = scl[rb + min(i0 + 64 * u, rc - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        gg[u] = sgod[kk[u]] * 2.0f;
        cx[u] = sob[(size_t)kk[u] * 3], cy[u] = sob[(size_t)kk[u] * 3 + 1], cz[u] = sob[(size_t)kk[u] * 3 + 2];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (i0 + 64 * u >= rc) continue;
        const float a0 = fabsf(gg[u] * (cx[u] - qx)), a1 = fabsf(gg[u] * (cy[u] - qy)), a2 = fabsf(gg[u] * (cz[u] - qz));
        const float a = fmaxf(a0, fmaxf(a1, a2));
        const bool fin = (a0 < INFINITY) && (a1 < INFINITY) && (a2 < INFINITY);
        mx = fin ? fmaxf(mx, a) : mx;
        bad |= !fin;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    bad = __any(bad);
    int e2 = 0;
    (void)frexpf(mx > 0.0f ? mx : 1.0f, &e2);
    const double scale = ldexp(1.0, 46 - e2), inv = ldexp(1.0, e2 - 46);   // < 2^15 entries of < 2^46 each: 62 bits
    long long sx = 0, sy = 0, sz = 0;
    float fx = 0.f, fy = 0.f, fz = 0.f;                            // non-finite terms (NaN / infinity poison the sum)
    for (int i0 = lane; i0 < rc; i0 += 256) {
      int kk[4];
      float gg[4], cx[4], cy[4], cz[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) kk[u] = scl[rb + min(i0 + 64 * u, rc - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        gg[u] = sgod[kk[u]] * 2.0f;
        cx[u] = sob[(size_t)kk[u] * 3], cy[u] = sob[(size_t)kk[u] * 3 + 1], cz[u] = sob[(size_t)kk[u] * 3 + 2];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (i0 + 64 * u >= rc) continue;
        const float tx = -(gg[u] * (cx[u] - qx)), ty = -(gg[u] * (cy[u] - qy)), tz = -(gg[u] * (cz[u] - qz));
        if (fabsf(tx) < INFINITY) sx += __double2ll_rn((double)tx * scale); else fx += tx;
        if (fabsf(ty) < INFINITY) sy += __double2ll_rn((double)ty * scale); else fy += ty;
        if (fabsf(tz) < INFINITY) sz += __double2ll_rn((double)tz * scale); else fz += tz;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      sx += __shfl_xor(sx, off);
      sy += __shfl_xor(sy, off);
      sz += __shfl_xor(sz, off);
    }
    if (bad) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        fx += __shfl_xor(fx, off);
        fy += __shfl_xor(fy, off);
        fz += __shfl_xor(fz, off);
      }
    }
    if (lane == src) {
      ax += (float)((double)sx * inv) + fx;
      ay += (float)((double)sy * inv) + fy;
      az += (float)((double)sz * inv) + fz;
    }
  }
  if (live) {
    if (second) ax += ox, ay += oy, az += oz;                      // cloud 2: the own term is added last
    float* gp = (second ? g2 : g1) + ((size_t)b * np + j) * 3;
    gp[0] = ax, gp[1] = ay, gp[2] = az;
  }
}

struct Plan {
  int Q;
  bool pk;
  int qtiles[2], splits[2], cps[2];
  bool need_ws;
};

#ifdef PTOPS_TUNING  // benchmarks/chamfer_tune.py: `make clean && make TUNING=1`
int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
#else
inline int env_int(const char*, int dflt) { return dflt; }
#endif

// Pure function of (B, n, m): decides the query tiling and candidate splits.
// Measured on MI355X (profiles/round1/chamfer_tune.log): 4 queries per lane win when both clouds
// are large, 2 otherwise; the packed-math body (v_pk_*_f32, same flops per cycle, fewer issued
// instructions) is 2-3 % faster than the scalar body at every Q.
Plan make_plan(int B, int n, int m) {
  Plan p;
  const int small = n < m ? n : m;
  p.Q = small >= 4096 ? 4 : ((long long)B * (n > m ? n : m) >= 16 * 1024 ? 2 : 1);
  p.Q = env_int("PTOPS_TUNE_Q", p.Q);
  p.pk = env_int("PTOPS_TUNE_PK", 1) != 0;
  const int force_split = env_int("PTOPS_TUNE_SPLITS", 0);
  const int target_mult = env_int("PTOPS_TUNE_TARGET", 8);
  const int nq[2] = {n, m}, nc[2] = {m, n};
  const int target_blocks = target_mult * ptops::kNumCU;  // ~8 workgroups (32 waves) per CU
  p.need_ws = false;
  for (int d = 0; d < 2; ++d) {
    p.qtiles[d] = ptops::ceil_div(nq[d], kTPB * p.Q);
    const long long tiles = (long long)B * p.qtiles[d];
    const int chunks = ptops::ceil_div(nc[d], kCH);
    long long s = tiles >= target_blocks / 2 ? 1 : target_blocks / (tiles > 0 ? tiles : 1);
    const int max_s = chunks / 8 > 0 ? chunks / 8 : 1;  // >= 128 candidates per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    if (force_split > 0) s = force_split > max_s ? max_s : force_split;
    int cps = ptops::ceil_div(chunks, (int)s) * kCH;
    s = ptops::ceil_div(nc[d], cps);  // drop empty splits
    p.splits[d] = (int)s;
    p.cps[d] = cps;
    if (s > 1) p.need_ws = true;
  }
  return p;
}

// Path choice, a pure function of the sizes: the pruned sweep when both clouds can be sorted in one
// workgroup, else brute force.  (Tuning builds can force brute force with PTOPS_TUNE_SWEEP=0.)
bool use_sweep(int B, int n, int m) {
  return ptops_sweep::eligible(B, n, m) && env_int("PTOPS_TUNE_SWEEP", 1) != 0;
}

}  // namespace

static size_t brute_workspace_bytes(int B, int n, int m) {
  Plan p = make_plan(B, n, m);
  return p.need_ws ? sizeof(unsigned long long) * (size_t)B * ((size_t)n + m) : 0;
}

static size_t align256(size_t v) { return (v + 255) / 256 * 256; }

PTOPS_API size_t ptops_chamfer_forward_workspace_bytes(int B, int n, int m) {
  if (B <= 0 || n <= 0 || m <= 0) return 0;
  // room for all algorithms at once: the two directions of one call may use different ones
  // layout: [sweep: sorted clouds | mfma: operands, bounds, packed minima | brute: packed minima]
  return align256(use_sweep(B, n, m) ? ptops_sweep::workspace_bytes(B, n, m) : 0) +
         align256(ptops_mfma::workspace_bytes(B, n, m)) + brute_workspace_bytes(B, n, m);
}

PTOPS_API int ptops_chamfer_sweep_available(int B, int n, int m) {
  return (B > 0 && n > 0 && m > 0 && use_sweep(B, n, m)) ? 1 : 0;
}

PTOPS_API int ptops_chamfer_forward_ex(const float* xyz1, const float* xyz2, int B, int n, int m, float* dist1,
                                       float* dist2, int32_t* idx1, int32_t* idx2, void* workspace,
                                       int algorithm1, int algorithm2, unsigned long long* stats, void* stream_);

PTOPS_API int ptops_chamfer_forward(const float* xyz1, const float* xyz2, int B, int n, int m,
                                    float* dist1, float* dist2, int32_t* idx1, int32_t* idx2,
                                    void* workspace, void* stream_) {
  return ptops_chamfer_forward_ex(xyz1, xyz2, B, n, m, dist1, dist2, idx1, idx2, workspace, PTOPS_CHAMFER_AUTO,
                                  PTOPS_CHAMFER_AUTO, nullptr, stream_);
}

// Brute force for the directions in dirmask (bit 0: xyz1 -> xyz2, bit 1: xyz2 -> xyz1).
static int brute_forward(const float* xyz1, const float* xyz2, int B, int n, int m, float* dist1, float* dist2,
                         int32_t* idx1, int32_t* idx2, unsigned long long* ws, int dirmask, hipStream_t stream) {
  const Plan p = make_plan(B, n, m);
  const bool on[2] = {(dirmask & 1) != 0, (dirmask & 2) != 0};
  const bool split[2] = {on[0] && p.splits[0] > 1, on[1] && p.splits[1] > 1};
  if (split[0] || split[1]) {
    PTOPS_REQUIRE(ws != nullptr, "chamfer_forward: workspace required (%zu bytes)", brute_workspace_bytes(B, n, m));
    if (((uintptr_t)ws & 7) != 0) {
      ptops::set_error("chamfer_forward: workspace must be 8-byte aligned");
      return PTOPS_EALIGN;
    }
    if (split[0]) (void)ptops::fill_async(ws, 0xFF, sizeof(unsigned long long) * (size_t)B * n, stream);
    if (split[1]) (void)ptops::fill_async(ws + (size_t)B * n, 0xFF, sizeof(unsigned long long) * (size_t)B * m, stream);
  }
  Args a;
  const long long blocks0 = on[0] ? (long long)B * p.qtiles[0] * p.splits[0] : 0;
  const long long blocks1 = on[1] ? (long long)B * p.qtiles[1] * p.splits[1] : 0;
  a.d[0] = Dir{xyz1, xyz2, dist1, idx1, ws, n, m, p.qtiles[0], p.splits[0], p.cps[0], 0};
  a.d[1] = Dir{xyz2, xyz1, dist2, idx2, ws ? ws + (size_t)B * n : nullptr, m, n, p.qtiles[1],
               p.splits[1], p.cps[1], (int)blocks0};
  const long long blocks = blocks0 + blocks1;
  PTOPS_REQUIRE(blocks < (1LL << 31), "chamfer_forward: grid too large");
  if (blocks == 0) return PTOPS_OK;
  dim3 grid((unsigned)blocks), block(kTPB);
  switch (p.Q * 2 + (p.pk ? 1 : 0)) {
    case 17: hipLaunchKernelGGL((chamfer_nn_kernel<8, true>), grid, block, 0, stream, a); break;
    case 16: hipLaunchKernelGGL((chamfer_nn_kernel<8, false>), grid, block, 0, stream, a); break;
    case 9: hipLaunchKernelGGL((chamfer_nn_kernel<4, true>), grid, block, 0, stream, a); break;
    case 8: hipLaunchKernelGGL((chamfer_nn_kernel<4, false>), grid, block, 0, stream, a); break;
    case 5: hipLaunchKernelGGL((chamfer_nn_kernel<2, true>), grid, block, 0, stream, a); break;
    case 4: hipLaunchKernelGGL((chamfer_nn_kernel<2, false>), grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL((chamfer_nn_kernel<1, false>), grid, block, 0, stream, a); break;
  }
  int rc = ptops::check_launch("chamfer_nn_kernel");
  if (rc != PTOPS_OK) return rc;
  for (int d = 0; d < 2; ++d) {  // unpack only the directions that were split
    if (!split[d]) continue;
    const long long total = (long long)B * (d ? m : n);
    const int g = (int)(ptops::ceil_div_ll(total, 256) < 2048 ? ptops::ceil_div_ll(total, 256) : 2048);
    hipLaunchKernelGGL(chamfer_unpack_kernel, dim3(g), dim3(256), 0, stream, d ? ws + (size_t)B * n : ws,
                       d ? dist2 : dist1, d ? idx2 : idx1, total);
    rc = ptops::check_launch("chamfer_unpack_kernel");
  }
  return rc;
}

PTOPS_API int ptops_chamfer_forward_ex(const float* xyz1, const float* xyz2, int B, int n, int m, float* dist1,
                                       float* dist2, int32_t* idx1, int32_t* idx2, void* workspace,
                                       int algorithm1, int algorithm2, unsigned long long* stats, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int algo[2] = {algorithm1, algorithm2};
  for (int d = 0; d < 2; ++d)
    PTOPS_REQUIRE(algo[d] == PTOPS_CHAMFER_AUTO || algo[d] == PTOPS_CHAMFER_BRUTE || algo[d] == PTOPS_CHAMFER_SWEEP ||
                      algo[d] == PTOPS_CHAMFER_MFMA,
                  "chamfer_forward: unknown algorithm %d", algo[d]);
  PTOPS_REQUIRE(B >= 0 && n >= 0 && m >= 0, "chamfer_forward: negative size");
  if (B == 0) return PTOPS_OK;
  PTOPS_REQUIRE(dist1 && dist2 && idx1 && idx2, "chamfer_forward: null output");
  if (n == 0 || m == 0) {  // the reference returns its zero-initialised outputs (chamfer.cu:152-157)
    if (n) (void)ptops::fill_async(dist1, 0, sizeof(float) * (size_t)B * n, stream), (void)ptops::fill_async(idx1, 0, sizeof(int32_t) * (size_t)B * n, stream);
    if (m) (void)ptops::fill_async(dist2, 0, sizeof(float) * (size_t)B * m, stream), (void)ptops::fill_async(idx2, 0, sizeof(int32_t) * (size_t)B * m, stream);
    return ptops::check_launch("chamfer_forward(memset)");
  }
  PTOPS_REQUIRE(xyz1 && xyz2, "chamfer_forward: null input");
  PTOPS_REQUIRE((long long)B * n * 3 < (1LL << 31) && (long long)B * m * 3 < (1LL << 31),
                "chamfer_forward: B*n*3 must fit int32");
  const bool can_sweep = use_sweep(B, n, m);
  const long long pairs = (long long)B * n * m;   // pair evaluations of either direction
  int sweep_mask = 0, mfma_mask = 0;
  for (int d = 0; d < 2; ++d) {
    PTOPS_REQUIRE(algo[d] != PTOPS_CHAMFER_SWEEP || can_sweep,
                  "chamfer_forward: the sweep needs both clouds within [4096, 16384] points (n=%d m=%d)", n, m);
    if (algo[d] == PTOPS_CHAMFER_SWEEP || (algo[d] == PTOPS_CHAMFER_AUTO && can_sweep)) sweep_mask |= 1 << d;
    else if (algo[d] == PTOPS_CHAMFER_MFMA ||
             (algo[d] == PTOPS_CHAMFER_AUTO && pairs >= PTOPS_CHAMFER_MFMA_MIN_PAIRS)) mfma_mask |= 1 << d;
  }
  // workspace layout: see ptops_chamfer_forward_workspace_bytes
  char* wsp = (char*)workspace;
  const size_t sweep_bytes = align256(can_sweep ? ptops_sweep::workspace_bytes(B, n, m) : 0);
  const size_t mfma_bytes = align256(ptops_mfma::workspace_bytes(B, n, m));
  if (sweep_mask || mfma_mask) {
    PTOPS_REQUIRE(wsp != nullptr, "chamfer_forward: workspace required (%zu bytes)",
                  ptops_chamfer_forward_workspace_bytes(B, n, m));
    if (((uintptr_t)wsp & 15) != 0) {
      ptops::set_error("chamfer_forward: workspace must be 16-byte aligned");
      return PTOPS_EALIGN;
    }
  }
  if (sweep_mask) {
    int rc = ptops_sweep::forward(xyz1, xyz2, B, n, m, dist1, dist2, idx1, idx2, wsp, stats, sweep_mask, stream);
    if (rc != PTOPS_OK) return rc;
  }
  if (mfma_mask) {
    int rc = ptops_mfma::forward(xyz1, xyz2, B, n, m, dist1, dist2, idx1, idx2, wsp + sweep_bytes, stats, mfma_mask,
                                 stream);
    if (rc != PTOPS_OK) return rc;
  }
  if ((sweep_mask | mfma_mask) == 3) return PTOPS_OK;
  return brute_forward(xyz1, xyz2, B, n, m, dist1, dist2, idx1, idx2,
                       wsp ? (unsigned long long*)(wsp + sweep_bytes + mfma_bytes) : nullptr,
                       3 & ~(sweep_mask | mfma_mask), stream);
}

PTOPS_API int ptops_chamfer_backward(const float* xyz1, const float* xyz2, const int32_t* idx1,
                                     const int32_t* idx2, const float* grad_dist1,
                                     const float* grad_dist2, int B, int n, int m,
                                     float* grad_xyz1, float* grad_xyz2, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && n >= 0 && m >= 0, "chamfer_backward: negative size");
  if (B == 0) return PTOPS_OK;
  if (n == 0 || m == 0) {
    if (n) (void)ptops::fill_async(grad_xyz1, 0, sizeof(float) * (size_t)B * n * 3, stream);
    if (m) (void)ptops::fill_async(grad_xyz2, 0, sizeof(float) * (size_t)B * m * 3, stream);
    return ptops::check_launch("chamfer_backward(memset)");
  }
  PTOPS_REQUIRE(xyz1 && xyz2 && idx1 && idx2 && grad_dist1 && grad_dist2 && grad_xyz1 && grad_xyz2,
                "chamfer_backward: null pointer");
  const long long total = (long long)B * ((long long)n + m);
  const int g = (int)(ptops::ceil_div_ll(total, 256) < 4096 ? ptops::ceil_div_ll(total, 256) : 4096);
  hipLaunchKernelGGL(chamfer_grad_kernel<false>, dim3(g), dim3(256), 0, stream, xyz1, xyz2, idx1, idx2,
                     grad_dist1, grad_dist2, B, n, m, grad_xyz1, grad_xyz2);
  hipLaunchKernelGGL(chamfer_grad_kernel<true>, dim3(g), dim3(256), 0, stream, xyz1, xyz2, idx1, idx2,
                     grad_dist1, grad_dist2, B, n, m, grad_xyz1, grad_xyz2);
  return ptops::check_launch("chamfer_grad_kernel");
}

// Deterministic form of ptops_chamfer_backward: same values up to the order of the float additions, but that order is
// fixed - the order of the scalar loops of the CPU oracle - so repeated runs are bit-identical (and bit-equal to it).  Needs
// max(n, m) <= 16384 (the transposition's counters live in LDS) - otherwise PTOPS_EINVAL, use ptops_chamfer_backward.
PTOPS_API size_t ptops_chamfer_backward_det_workspace_bytes(int B, int n, int m) {
  if (B <= 0 || n <= 0 || m <= 0) return 16;
  // per direction: rowptr [B][rows + 1], unordered + ordered columns [B][entries] each
  return sizeof(int32_t) * ((size_t)B * ((size_t)n + m + 2) + 4 * (size_t)B * ((size_t)n + m)) + 64;
}

PTOPS_API int ptops_chamfer_backward_det(const float* xyz1, const float* xyz2, const int32_t* idx1, const int32_t* idx2,
                                         const float* grad_dist1, const float* grad_dist2, int B, int n, int m,
                                         float* grad_xyz1, float* grad_xyz2, void* workspace, size_t workspace_bytes,
                                         void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && n >= 0 && m >= 0, "chamfer_backward_det: negative size");
  if (B == 0) return PTOPS_OK;
  if (n == 0 || m == 0)
    return ptops_chamfer_backward(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2, B, n, m, grad_xyz1, grad_xyz2, stream_);
  PTOPS_REQUIRE(xyz1 && xyz2 && idx1 && idx2 && grad_dist1 && grad_dist2 && grad_xyz1 && grad_xyz2 && workspace,
                "chamfer_backward_det: null pointer");
  PTOPS_REQUIRE(n <= 16384 && m <= 16384, "chamfer_backward_det: clouds of at most 16384 points (n=%d m=%d)", n, m);
  PTOPS_REQUIRE(B <= 65535, "chamfer_backward_det: B exceeds the grid limit");
  PTOPS_REQUIRE(workspace_bytes >= ptops_chamfer_backward_det_workspace_bytes(B, n, m) && ((uintptr_t)workspace & 15) == 0,
                "chamfer_backward_det: workspace too small or misaligned");
  int32_t* w = (int32_t*)workspace;
  int32_t* rowptr1 = w;                 // direction 0: rows = points of cloud 2, entries = points of cloud 1 (idx1)
  w += (size_t)B * (m + 1);
  int32_t* rowptr2 = w;                 // direction 1: rows = points of cloud 1, entries = points of cloud 2 (idx2)
  w += (size_t)B * (n + 1);
  int32_t* tmp1 = w;
  w += (size_t)B * n;
  int32_t* col1 = w;
  w += (size_t)B * n;
  int32_t* tmp2 = w;
  w += (size_t)B * m;
  int32_t* col2 = w;
  const int nc_max = n > m ? n : m;
  const size_t lds = sizeof(int) * (2 * (size_t)nc_max + 16);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chamfer_csr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(chamfer_csr_kernel, dim3(B, 2), dim3(kCsrT), lds, stream, idx1, idx2, n, m, rowptr1, tmp1, col1, rowptr2,
                     tmp2, col2);
  int rc = ptops::check_launch("chamfer_csr_kernel");
  if (rc != PTOPS_OK) return rc;
  const long long total = (long long)B * ((long long)n + m);
  hipLaunchKernelGGL(chamfer_grad_det_kernel, dim3((unsigned)ptops::ceil_div_ll(total, 256)), dim3(256), 0, stream, xyz1, xyz2,
                     idx1, idx2, grad_dist1, grad_dist2, B, n, m, rowptr1, col1, rowptr2, col2, grad_xyz1, grad_xyz2);
  return ptops::check_launch("chamfer_grad_det_kernel");
}


// ---- reductions of the two distance vectors: ChamferDistanceL2 / L2_split / L1 / L1_PM ------------------------------
// (extensions/chamfer_dist/__init__.py:28-103 of the reference: torch.mean / torch.sqrt / + / 2 on dist1 and dist2, six
// element-wise / reduction launches forward and seven backward per loss term.)  One launch each way here:
//   forward : out[0] = mean(f(dist1)), out[1] = mean(f(dist2)), out[2] = out[0] + out[1], out[3] = out[2] / 2,
//             f = sqrt (take_sqrt) or identity; per-block partial sums, the last block to finish adds them in a fixed
//             order (deterministic) and re-arms the counter;
//   backward: the gradients of dist1 / dist2 for a gradient of one of those outputs, with autograd's own formulas
//             (mean: g x (1 / N), sqrt: g / (2 * result), "/ 2": g / 2) in the same order, so the values are aten's.
namespace {

constexpr int kRedThreads = 256, kRedChunk = 4096;

__device__ __forceinline__ float red_block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();                                       // sh may still be read from the previous use
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(kRedThreads) void chamfer_reduce_kernel(const float* __restrict__ d1, long long n1,
                                                                     const float* __restrict__ d2, long long n2,
                                                                     int take_sqrt, float* partial,
                                                                     unsigned int* counter, float* __restrict__ out) {
  __shared__ float sh[4];
  __shared__ int is_last;
  const int t = threadIdx.x;
  const long long nb1 = ptops::ceil_div_ll(n1, kRedChunk), nb2 = ptops::ceil_div_ll(n2, kRedChunk);
  const bool second = (long long)blockIdx.x >= nb1;
  const float* d = second ? d2 : d1;
  const long long n = second ? n2 : n1;
  const long long base = ((long long)blockIdx.x - (second ? nb1 : 0)) * kRedChunk;
  float s = 0.f;
#pragma unroll 4
  for (int i = 0; i < kRedChunk / kRedThreads; ++i) {
    const long long j = base + (long long)i * kRedThreads + t;
    if (j < n) {
      const float v = d[j];
      s += take_sqrt ? sqrtf(v) : v;
    }
  }
  s = red_block_sum(s, sh);
  if (t == 0) {
    partial[blockIdx.x] = s;
    __threadfence();
    is_last = atomicAdd(counter, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  float s1 = 0.f, s2 = 0.f;
  for (long long i = t; i < nb1; i += kRedThreads) s1 += __builtin_nontemporal_load(partial + i);
  for (long long i = t; i < nb2; i += kRedThreads) s2 += __builtin_nontemporal_load(partial + nb1 + i);
  s1 = red_block_sum(s1, sh);
  s2 = red_block_sum(s2, sh);
  if (t == 0) {
    // sum x (1 / N), the form of aten's mean kernel; 0 x inf = NaN for an empty cloud, as torch.mean
    const float m1 = s1 * (1.0f / (float)n1), m2 = s2 * (1.0f / (float)n2);
    out[0] = m1, out[1] = m2, out[2] = m1 + m2, out[3] = (m1 + m2) / 2.f;
    *counter = 0u;
  }
}

// which: 0 -> gradient of out[0] and out[1] separately (go1 / go2, either may be null = zero),
//        2 -> of out[2] (go1), 3 -> of out[3] (go1)
__global__ __launch_bounds__(256) void chamfer_reduce_grad_kernel(const float* __restrict__ d1, long long n1,
                                                                  const float* __restrict__ d2, long long n2,
                                                                  int take_sqrt, int which, const float* __restrict__ go1,
                                                                  const float* __restrict__ go2, float* __restrict__ g1,
                                                                  float* __restrict__ g2) {
  float a = go1 ? *go1 : 0.f, b = which == 0 ? (go2 ? *go2 : 0.f) : a;
  if (which == 3) a = a / 2.f, b = a;
  // mean backward: g / N, which aten evaluates as g x (1 / N) for a scalar divisor (so does "/ 2", exactly)
  const float ga = a * (1.0f / (float)n1), gb = b * (1.0f / (float)n2);
  const long long total = n1 + n2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const bool second = i >= n1;
    const long long j = second ? i - n1 : i;
    const float g = second ? gb : ga;
    const bool live = second ? (which != 0 || go2) : (go1 != nullptr);
    float r = g;
    if (take_sqrt) r = live ? g / (2.f * sqrtf((second ? d2 : d1)[j])) : 0.f;   // sqrt backward: g / (2 * result)
    (second ? g2 : g1)[j] = r;
  }
}

}  // namespace

PTOPS_API size_t ptops_chamfer_reduce_workspace_bytes(long long n1, long long n2) {
  return sizeof(float) * (size_t)(ptops::ceil_div_ll(n1 > 0 ? n1 : 0, kRedChunk) + ptops::ceil_div_ll(n2 > 0 ? n2 : 0, kRedChunk)) + 64;
}

// workspace: ptops_chamfer_reduce_workspace_bytes(n1, n2) bytes whose first 4 bytes are ZERO before the first use and
// are left zero by every launch (keep one workspace per stream); out: 4 floats.
PTOPS_API int ptops_chamfer_reduce(const float* dist1, long long n1, const float* dist2, long long n2, int take_sqrt,
                                   float* out, void* workspace, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(n1 >= 0 && n2 >= 0, "chamfer_reduce: negative size");
  PTOPS_REQUIRE(out && workspace && (dist1 || n1 == 0) && (dist2 || n2 == 0), "chamfer_reduce: null pointer");
  const long long blocks = ptops::ceil_div_ll(n1, kRedChunk) + ptops::ceil_div_ll(n2, kRedChunk);
  PTOPS_REQUIRE(blocks < (1LL << 31), "chamfer_reduce: too many elements");
  unsigned int* counter = (unsigned int*)workspace;
  float* partial = (float*)((char*)workspace + 64);
  // an empty problem still launches one block: it finds no work, is the last one and writes the NaN means
  hipLaunchKernelGGL(chamfer_reduce_kernel, dim3((unsigned)(blocks > 0 ? blocks : 1)), dim3(kRedThreads), 0, stream, dist1, n1,
                     dist2, n2, take_sqrt, partial, counter, out);
  return ptops::check_launch("chamfer_reduce_kernel");
}

PTOPS_API int ptops_chamfer_reduce_backward(const float* dist1, long long n1, const float* dist2, long long n2, int take_sqrt,
                                            int which, const float* grad_out1, const float* grad_out2, float* grad_dist1,
                                            float* grad_dist2, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(n1 >= 0 && n2 >= 0, "chamfer_reduce_backward: negative size");
  PTOPS_REQUIRE(which == 0 || which == 2 || which == 3, "chamfer_reduce_backward: which must be 0, 2 or 3");
  if (n1 + n2 == 0) return PTOPS_OK;
  PTOPS_REQUIRE((grad_dist1 || n1 == 0) && (grad_dist2 || n2 == 0) && (which == 0 || grad_out1),
                "chamfer_reduce_backward: null pointer");
  PTOPS_REQUIRE(!take_sqrt || ((dist1 || n1 == 0) && (dist2 || n2 == 0)), "chamfer_reduce_backward: null distances");
  const long long blocks = ptops::ceil_div_ll(n1 + n2, 256 * 8);
  hipLaunchKernelGGL(chamfer_reduce_grad_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, stream, dist1,
                     n1, dist2, n2, take_sqrt, which, grad_out1, grad_out2, grad_dist1, grad_dist2);
  return ptops::check_launch("chamfer_reduce_grad_kernel");
}
