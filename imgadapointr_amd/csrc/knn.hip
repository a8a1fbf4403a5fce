// Fused k-nearest-neighbour query for gfx950 (MI355X).
//
// Replaces knn_point + square_distance (models/Transformer_utils.py:19-51 of the reference),
// which materialise a [B,S,N] fp32 matrix (537 MB at B=32, S=N=2048) and run a generic top-k
// over it.  Here nothing but the [B,S,k] int32 result touches HBM:
//  * a workgroup (4 waves) stages a tile of up to 2048 candidates in LDS as float4
//    {x, y, z, |c|^2} (one conflict-free ds_read_b128 per lane and candidate) and serves
//    kQPB queries from it; larger clouds are walked tile by tile;
//  * one wave owns one query at a time: every lane evaluates P = tile/64 candidates into
//    registers (expansion-form distance, bit-exact with oracle/ptops_ref.c::sqdist_expanded,
//    query in SGPRs), then the k smallest are extracted in (distance, index) order by rounds
//    of {lane-local v_min3 -> DPP wave-min -> ballot/mbcnt compaction of every candidate equal
//    to that minimum, in index order}.  Ties (duplicated / zero-padded points) are therefore
//    emitted exactly as the oracle orders them, several per round;
//  * the running top-k of earlier tiles is carried as one extra (distance, index) register
//    pair (lane j = j-th best) and merged by the same rounds; a tile that cannot improve the
//    current k-th best is skipped after the distance pass.
#include <math.h>

#include "ptops_common.h"

namespace {

constexpr int kTPB = 256;
constexpr int kWaves = kTPB / 64;
// Queries per workgroup: 16 amortise the LDS staging of a tile when there are many queries; with few
// queries (the 64-seed denoise search over 16384 points) 4 — one per wave — fill the chip instead.
constexpr int kMaxK = 64;

// min over the wave (all lanes receive it).  Written as ISA: the compiler lowers fminf(v, update_dpp(v)) to v_mov_b32 +
// v_mov_b32_dpp + a canonicalising v_max + v_min (five instructions and a nop per step); `v_min_f32_dpp` does a step in
// one instruction.  Every extraction round of a query starts with this chain (up to 2 k of them per query), so it is
// most of a query's serial latency.  A lane whose source lane does not exist keeps its value; a DPP read of a VGPR
// written by the previous VALU instruction needs two wait states (s_nop 1), which the assembler does not add inside an
// asm block.  v_min_f32 returns the other operand for a quiet NaN, as fminf does (NaN coordinates are never selected).
__device__ __forceinline__ float wave_min_f32(float v) {
#define PTOPS_DPP_STEP(ctrl) "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 " ctrl " bank_mask:0xf\n\t"
  asm volatile(PTOPS_DPP_STEP("row_shr:1 row_mask:0xf") PTOPS_DPP_STEP("row_shr:2 row_mask:0xf")
               PTOPS_DPP_STEP("row_shr:4 row_mask:0xf") PTOPS_DPP_STEP("row_shr:8 row_mask:0xf")
               PTOPS_DPP_STEP("row_bcast:15 row_mask:0xa") PTOPS_DPP_STEP("row_bcast:31 row_mask:0xc") "s_nop 1"
               : "+v"(v));
#undef PTOPS_DPP_STEP
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float sumsq3(float x, float y, float z) {
  const float a = x * x, b = y * y, c = z * z;
  return (a + b) + c;
}

// P = candidates per lane per tile (tile = 64*P candidates); kQPB = queries per workgroup.
template <int P, int kQPB>
__global__ __launch_bounds__(kTPB) void knn_kernel(const float* __restrict__ xyz,
                                                   const float* __restrict__ new_xyz, int N, int S,
                                                   int k, int32_t* __restrict__ idx_out) {
  constexpr int TILE = 64 * P;
  __shared__ float4 cand[TILE];
  __shared__ float topd[kQPB][kMaxK];
  __shared__ int topi[kQPB][kMaxK];
  __shared__ float survd[kWaves][64];  // per wave: candidates that passed the prefilter, in index order
  __shared__ int survi[kWaves][64];

  const int b = blockIdx.y;
  const int q0 = blockIdx.x * kQPB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* __restrict__ cb = xyz + (size_t)b * N * 3;
  const float* __restrict__ qb = new_xyz + (size_t)b * S * 3;
  const float kInf = INFINITY;

  for (int c0 = 0; c0 < N; c0 += TILE) {
    if (c0 > 0) __syncthreads();  // previous tile fully consumed
    for (int t = tid; t < TILE; t += kTPB) {
      const int c = c0 + t;
      float4 v;
      if (c < N) {
        v.x = cb[(size_t)c * 3 + 0];
        v.y = cb[(size_t)c * 3 + 1];
        v.z = cb[(size_t)c * 3 + 2];
        v.w = sumsq3(v.x, v.y, v.z);
      } else {
        v.x = v.y = v.z = 0.0f;
        v.w = kInf;  // -> distance +inf, never selected
      }
      cand[t] = v;
    }
    __syncthreads();

    for (int qi = wave; qi < kQPB; qi += kWaves) {
      const int s = q0 + qi;
      if (s >= S) break;  // wave-uniform
      const float qx = qb[(size_t)s * 3 + 0], qy = qb[(size_t)s * 3 + 1], qz = qb[(size_t)s * 3 + 2];
      const float qq = sumsq3(qx, qy, qz);

      float d[P];
      float lmin = kInf;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const float4 c = cand[lane + 64 * p];
        const float dot = __builtin_fmaf(qz, c.z, __builtin_fmaf(qy, c.y, qx * c.x));
        const float t = __builtin_fmaf(-2.0f, dot, qq);
        d[p] = t + c.w;
        lmin = fminf(lmin, d[p]);
      }
      // carried top-k of earlier tiles: lane j holds the j-th best (all at lower indices)
      float dprev = kInf;
      int iprev = 0;
      if (c0 > 0) {
        if (lane < k) {
          dprev = topd[qi][lane];
          iprev = topi[qi][lane];
        }
        // the k-th best so far; a tile with nothing strictly below it changes nothing
        const float kth = __builtin_bit_cast(
            float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dprev), (k - 1) & 63));
        if (!(wave_min_f32(lmin) < kth)) continue;
      }

      int pos = 0;
      if constexpr (P >= 16) {
        // Prefilter (large tiles): the k-th smallest of the 64 per-lane minima is an upper bound tau on the k-th
        // smallest candidate overall (they are 64 distinct candidates), so only candidates <= tau can be among the k
        // best.  They are compacted in index order into a list of at most 64 (carried entries first: lower
        // indices) and the extraction rounds then run on ONE value per lane instead of P + 1.
        float t = fminf(lmin, dprev);
        float tau = kInf;
        for (int cnt = 0; cnt < k;) {
          const float m = wave_min_f32(t);
          if (!(m < kInf)) break;
          const unsigned long long mask = __ballot(t == m);
          cnt += __builtin_popcountll(mask);
          t = (t == m) ? kInf : t;
          tau = m;
        }
        int ns = 0;  // survivors so far (wave-uniform)
        float sd = kInf;
        int si = 0;
        bool fits = true;
        {
          const bool keep = dprev <= tau;
          const unsigned long long mask = __ballot(keep);
          const int slot = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
          if (keep) {
            survd[wave][slot] = dprev;
            survi[wave][slot] = iprev;
          }
          ns = __builtin_popcountll(mask);
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const bool keep = d[p] <= tau;
          const unsigned long long mask = __ballot(keep);
          if (mask) {
            const int slot = ns + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            if (keep && slot < 64) {
              survd[wave][slot] = d[p];
              survi[wave][slot] = c0 + lane + 64 * p;
            }
            ns += __builtin_popcountll(mask);
          }
        }
        fits = ns <= 64;
        if (fits) {
          __builtin_amdgcn_wave_barrier();
          if (lane < ns) {
            sd = survd[wave][lane];
            si = survi[wave][lane];
          }
          while (pos < k) {
            const float m = wave_min_f32(sd);
            if (!(m < kInf)) break;
            const bool eq = sd == m;
            const unsigned long long mask = __ballot(eq);
            const int slot = pos + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            if (eq && slot < k) {
              topd[qi][slot] = m;
              topi[qi][slot] = si;
            }
            pos += __builtin_popcountll(mask);
            sd = eq ? kInf : sd;
          }
          if (lane >= pos && lane < k) {
            topd[qi][lane] = kInf;
            topi[qi][lane] = 0;
          }
          continue;  // next query of this wave
        }
        // more than 64 candidates tie at or below tau (duplicated points): the general rounds below handle it
      }
      while (pos < k) {
        float lm = dprev;
#pragma unroll
        for (int p = 0; p < P; ++p) lm = fminf(lm, d[p]);
        const float m = wave_min_f32(lm);
        if (!(m < kInf)) break;  // fewer than k finite candidates (NaN/inf inputs only)
        {
          const bool eq = dprev == m;
          const unsigned long long mask = __ballot(eq);
          if (mask) {
            const int slot = pos + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                  __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            if (eq && slot < k) {
              topd[qi][slot] = m;
              topi[qi][slot] = iprev;
            }
            pos += __builtin_popcountll(mask);
            dprev = eq ? kInf : dprev;
          }
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const bool eq = d[p] == m;
          const unsigned long long mask = __ballot(eq);
          if (mask) {
            const int slot = pos + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                  __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            if (eq && slot < k) {
              topd[qi][slot] = m;
              topi[qi][slot] = c0 + lane + 64 * p;
            }
            pos += __builtin_popcountll(mask);
            d[p] = eq ? kInf : d[p];
          }
        }
      }
      // slots >= pos (only if fewer than k finite candidates): keep the list well-formed
      if (lane >= pos && lane < k) {
        topd[qi][lane] = kInf;
        topi[qi][lane] = 0;
      }
    }
  }
  __syncthreads();
  // write-out: [kQPB][k] ints, coalesced over (query, j)
  for (int t = tid; t < kQPB * k; t += kTPB) {
    const int qi = t / k, j = t - qi * k;
    const int s = q0 + qi;
    if (s < S) idx_out[((size_t)b * S + s) * k + j] = topi[qi][j];
  }
}

}  // namespace

PTOPS_API int ptops_knn(const float* xyz, const float* new_xyz, int B, int N, int S, int k,
                        int32_t* idx, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && N >= 1 && S >= 0, "knn: bad size B=%d N=%d S=%d", B, N, S);
  PTOPS_REQUIRE(k >= 1 && k <= kMaxK && k <= N, "knn: need 1 <= k <= min(N, %d), got k=%d N=%d", kMaxK, k, N);
  if (B == 0 || S == 0) return PTOPS_OK;
  PTOPS_REQUIRE(xyz && new_xyz && idx, "knn: null pointer");
  PTOPS_REQUIRE(B <= 65535, "knn: B=%d exceeds grid.y limit", B);
  // queries per workgroup: every workgroup of a cloud stages the same candidate tiles (through L2: 81 MB counted against
  // 19 MB algorithmic on the 2048 x 2048 call, profiles/round3/knn_pmc.json), so launches with enough queries to fill the
  // chip anyway (>= 4 workgroups per CU at 32 per workgroup) take 32 per workgroup and stage half as often
  const bool few = (long long)B * S < 16384;
  const bool many = (long long)B * S >= 32768;
  const int qpb = few ? 4 : (many ? 32 : 16);
  dim3 grid(ptops::ceil_div(S, qpb), B), block(kTPB);
#define PTOPS_KNN_LAUNCH(P)                                                                              \
  do {                                                                                                   \
    if (few) hipLaunchKernelGGL((knn_kernel<P, 4>), grid, block, 0, stream, xyz, new_xyz, N, S, k, idx);  \
    else if (many) hipLaunchKernelGGL((knn_kernel<P, 32>), grid, block, 0, stream, xyz, new_xyz, N, S, k, idx); \
    else hipLaunchKernelGGL((knn_kernel<P, 16>), grid, block, 0, stream, xyz, new_xyz, N, S, k, idx);     \
  } while (0)
  if (N <= 256) PTOPS_KNN_LAUNCH(4);
  else if (N <= 512) PTOPS_KNN_LAUNCH(8);
  else if (N <= 1024) PTOPS_KNN_LAUNCH(16);
  else PTOPS_KNN_LAUNCH(32);
#undef PTOPS_KNN_LAUNCH
  return ptops::check_launch("knn_kernel");
}
