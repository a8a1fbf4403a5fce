// Probe: "ping-pong" schedule for the split-bf16 fp32 GEMM (csrc/gemm_split3.hip), gfx950.
//
// gemm_split3_kernel runs all 8 waves of a tile through the same sequence per k-step (fragment reads, 12 MFMAs,
// split + ds_write of the next step, barrier); the two waves that share a SIMD are in lockstep, their MFMA phases
// serialise on the matrix pipe and their LDS / vector phases leave it idle (pipe busy 47 %, notes of round 2).
// Here the two halves of the workgroup alternate roles, one barrier per phase:
//     phase 2s    : waves 0-3 multiply step s            | waves 4-7 split + store operand B of step s+1, load s+2
//     phase 2s + 1: waves 0-3 split + store A of s+1 ... | waves 4-7 multiply step s
// so that every SIMD always has one wave on the matrix pipe and its partner on the vector ALU / LDS.  Same tile, same
// LDS image, same arithmetic and summation order as gemm_split3_kernel: results must be BIT-IDENTICAL.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include -I../../imgadapointr_amd/csrc \
//        gemm_pingpong_probe.hip -L../../imgadapointr_amd -lptops -Wl,-rpath,'$ORIGIN/../../imgadapointr_amd' -o gemm_pingpong_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

extern "C" int ptops_gemm_f32_split3(const float* A, long long lda, int a_k_contiguous, const float* B, long long ldb,
                                     int b_k_contiguous, float* C, long long ldc, int M, int N, int K, const float* bias,
                                     void* stream);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kBM = 128, kBN = 128, kBK = 16;
constexpr int kPlane = 132;

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  long long lda, ldb, ldc;
  int M, N, K;
  int tiles_n;
};

struct Pieces {
  uint4 h, m, l;
};

__device__ __forceinline__ Pieces split8(const float (&x)[8]) {
  bf16x8 h, m, l;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 a = (__bf16)x[i];
    const float r1 = x[i] - (float)a;
    const __bf16 b = (__bf16)r1;
    const __bf16 c = (__bf16)(r1 - (float)b);
    h[i] = a;
    m[i] = b;
    l[i] = c;
  }
  Pieces p;
  p.h = __builtin_bit_cast(uint4, h);
  p.m = __builtin_bit_cast(uint4, m);
  p.l = __builtin_bit_cast(uint4, l);
  return p;
}

template <bool KC>
__device__ __forceinline__ void load_entry(const float* __restrict__ P, long long ld, int rows, int r0, int k0, int t,
                                           float (&x)[8]) {
  if (KC) {
    const int row = min(r0 + (t >> 1), rows - 1);
    const float4* p = reinterpret_cast<const float4*>(P + (long long)row * ld + k0 + 8 * (t & 1));
    const float4 u = p[0], v = p[1];
    x[0] = u.x, x[1] = u.y, x[2] = u.z, x[3] = u.w, x[4] = v.x, x[5] = v.y, x[6] = v.z, x[7] = v.w;
  } else {
    const int row = min(r0 + (t & 127), rows - 1);
    const float* p = P + (long long)(k0 + 8 * (t >> 7)) * ld + row;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = p[(long long)i * ld];
  }
}

template <bool KC>
__device__ __forceinline__ int entry_index(int t) {
  return KC ? ((t & 1) * kPlane + (t >> 1)) : ((t >> 7) * kPlane + (t & 127));
}

// VARIANT 0: plain ping-pong.  VARIANT 1: + s_setprio(1) around the MFMAs.
template <bool AKC, bool BKC, int VARIANT>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(GemmArgs g) {
  __shared__ uint4 lds[2][2][3][2 * kPlane];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int col = lane & 31, half = lane >> 5;
  int bid;
  {
    const int orig = blockIdx.x, nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  }
  const int tile_m = bid / g.tiles_n, tile_n = bid - tile_m * g.tiles_n;
  const int m0 = tile_m * kBM, n0 = tile_n * kBN;
  const bool second = wave >= 4;                 // waves 4-7: stage B, multiply in the odd phases
  const int tt = t & 255;
  const int e = second ? entry_index<BKC>(tt) : entry_index<AKC>(tt);

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  float x[8];
  auto load = [&](int k0) {
    if (second) load_entry<BKC>(g.B, g.ldb, g.N, n0, k0, tt, x);
    else load_entry<AKC>(g.A, g.lda, g.M, m0, k0, tt, x);
  };
  auto store = [&](int buf) {
    const Pieces p = split8(x);
    uint4(*dst)[2 * kPlane] = lds[buf][second ? 1 : 0];
    dst[0][e] = p.h, dst[1][e] = p.m, dst[2][e] = p.l;
  };
  auto multiply = [&](int cur) {
    bf16x8 a[3], b[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      a[p] = __builtin_bit_cast(bf16x8, lds[cur][0][p][half * kPlane + wr * 32 + col]);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[p][j] = __builtin_bit_cast(bf16x8, lds[cur][1][p][half * kPlane + wc * 64 + j * 32 + col]);
    }
    if (VARIANT == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int term = 0; term < 6; ++term) {
      const int pa = term == 0 ? 2 : ((term == 1 || term == 3) ? 1 : 0);
      const int pb = term == 2 ? 2 : ((term == 1 || term == 4) ? 1 : 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b[pb][j], acc[j], 0, 0, 0);
    }
    if (VARIANT == 1) __builtin_amdgcn_s_setprio(0);
  };

  const int ksteps = g.K / kBK;
  load(0);
  store(0);
  if (ksteps > 1) load(kBK);
  __syncthreads();

  for (int s = 0; s < ksteps; ++s) {
    const int cur = s & 1;
    const bool more = s + 1 < ksteps, more2 = s + 2 < ksteps;
    // even phase
    if (!second) {
      multiply(cur);
    } else if (more) {
      store(cur ^ 1);
      if (more2) load((s + 2) * kBK);
    }
    __syncthreads();
    // odd phase
    if (second) {
      multiply(cur);
    } else if (more) {
      store(cur ^ 1);
      if (more2) load((s + 2) * kBK);
    }
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wc * 64 + j * 32 + col;
    if (n >= g.N) continue;
    const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (m < g.M) g.C[(long long)m * g.ldc + n] = acc[j][r] + bv;
    }
  }
}

// ---- the committed loop of gemm_split3_kernel (all waves: fragments, 12 MFMAs, split + store, one barrier per k-step) with
// the occupancy bound as a parameter: OCC workgroups per CU (3 needs <= 80 VGPRs).
template <bool AKC, bool BKC, int OCC>
__global__ __launch_bounds__(512, OCC) void gemm_base_kernel(GemmArgs g) {
  __shared__ uint4 lds[2][2][3][2 * kPlane];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int col = lane & 31, half = lane >> 5;
  int bid;
  {
    const int orig = blockIdx.x, nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  }
  const int tile_m = bid / g.tiles_n, tile_n = bid - tile_m * g.tiles_n;
  const int m0 = tile_m * kBM, n0 = tile_n * kBN;
  const bool stage_b = t >= 256;
  const int tt = t & 255;
  const int e = stage_b ? entry_index<BKC>(tt) : entry_index<AKC>(tt);
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float x[8];
  auto load = [&](int k0) {
    if (stage_b) load_entry<BKC>(g.B, g.ldb, g.N, n0, k0, tt, x);
    else load_entry<AKC>(g.A, g.lda, g.M, m0, k0, tt, x);
  };
  auto store = [&](int buf) {
    const Pieces p = split8(x);
    uint4(*dst)[2 * kPlane] = lds[buf][stage_b ? 1 : 0];
    dst[0][e] = p.h, dst[1][e] = p.m, dst[2][e] = p.l;
  };
  load(0);
  store(0);
  __syncthreads();
  const int ksteps = g.K / kBK;
  for (int ks = 0; ks < ksteps; ++ks) {
    const int cur = ks & 1;
    const bool more = ks + 1 < ksteps;
    if (more) load((ks + 1) * kBK);
    bf16x8 a[3], b[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      a[p] = __builtin_bit_cast(bf16x8, lds[cur][0][p][half * kPlane + wr * 32 + col]);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[p][j] = __builtin_bit_cast(bf16x8, lds[cur][1][p][half * kPlane + wc * 64 + j * 32 + col]);
    }
#pragma unroll
    for (int term = 0; term < 6; ++term) {
      const int pa = term == 0 ? 2 : ((term == 1 || term == 3) ? 1 : 0);
      const int pb = term == 2 ? 2 : ((term == 1 || term == 4) ? 1 : 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b[pb][j], acc[j], 0, 0, 0);
    }
    if (more) store(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wc * 64 + j * 32 + col;
    if (n >= g.N) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (m < g.M) g.C[(long long)m * g.ldc + n] = acc[j][r];
    }
  }
}

template <int OCC>
void launch_base(const float* A, long long lda, int akc, const float* B, long long ldb, int bkc, float* C, long long ldc, int M,
                 int N, int K, hipStream_t stream) {
  GemmArgs g;
  g.A = A, g.B = B, g.C = C, g.bias = nullptr;
  g.lda = lda, g.ldb = ldb, g.ldc = ldc;
  g.M = M, g.N = N, g.K = K;
  g.tiles_n = (N + kBN - 1) / kBN;
  const unsigned blocks = (unsigned)(((M + kBM - 1) / kBM) * g.tiles_n);
  if (akc && bkc) hipLaunchKernelGGL((gemm_base_kernel<true, true, OCC>), dim3(blocks), dim3(512), 0, stream, g);
  else if (akc) hipLaunchKernelGGL((gemm_base_kernel<true, false, OCC>), dim3(blocks), dim3(512), 0, stream, g);
  else if (bkc) hipLaunchKernelGGL((gemm_base_kernel<false, true, OCC>), dim3(blocks), dim3(512), 0, stream, g);
  else hipLaunchKernelGGL((gemm_base_kernel<false, false, OCC>), dim3(blocks), dim3(512), 0, stream, g);
}

// ---- ping-pong with 32-deep stages: 24 MFMAs per wave and phase (768 matrix cycles), two operand entries per thread and
// stage, 2 x 2 x 3 x 4 planes of LDS (99 KB: one workgroup per CU).  PREFETCH: the fragments of the first sub-step that are
// already complete are read at the end of the wave's staging phase, under the partner's MFMAs.
constexpr int kStage32 = 2 * 3 * 4 * kPlane;          // uint4 entries of one buffer: [operand][piece][k-quarter * kPlane + row]

template <bool KC>
__device__ __forceinline__ void load_entry2(const float* __restrict__ P, long long ld, int rows, int r0, int k0, int t,
                                            float (&x)[16]) {
  if (KC) {   // row t >> 1, 16 consecutive k starting at 16 * (t & 1): k-quarters 2 (t & 1) and 2 (t & 1) + 1
    const int row = min(r0 + (t >> 1), rows - 1);
    const float4* p = reinterpret_cast<const float4*>(P + (long long)row * ld + k0 + 16 * (t & 1));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 u = p[i];
      x[4 * i] = u.x, x[4 * i + 1] = u.y, x[4 * i + 2] = u.z, x[4 * i + 3] = u.w;
    }
  } else {    // row t & 127, k-quarters (t >> 7) and (t >> 7) + 2
    const int row = min(r0 + (t & 127), rows - 1);
    const float* p = P + (long long)(k0 + 8 * (t >> 7)) * ld + row;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = p[(long long)i * ld];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[8 + i] = p[(long long)(16 + i) * ld];
  }
}

template <bool AKC, bool BKC, bool PREFETCH>
__global__ __launch_bounds__(512, 1) void gemm_pp32_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds32[];      // [2][kStage32]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int col = lane & 31, half = lane >> 5;
  int bid;
  {
    const int orig = blockIdx.x, nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  }
  const int tile_m = bid / g.tiles_n, tile_n = bid - tile_m * g.tiles_n;
  const int m0 = tile_m * kBM, n0 = tile_n * kBN;
  const bool second = wave >= 4;
  const int tt = t & 255;
  const bool kc = second ? BKC : AKC;
  // entry indices of the two k-quarters this thread stages
  const int e0 = kc ? (2 * (tt & 1)) * kPlane + (tt >> 1) : (tt >> 7) * kPlane + (tt & 127);
  const int e1 = kc ? e0 + kPlane : e0 + 2 * kPlane;
  const int op = second ? 1 : 0;

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  float x[16];
  auto load = [&](int k0) {
    if (second) load_entry2<BKC>(g.B, g.ldb, g.N, n0, k0, tt, x);
    else load_entry2<AKC>(g.A, g.lda, g.M, m0, k0, tt, x);
  };
  auto store = [&](int buf) {
    float lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) lo[i] = x[i], hi[i] = x[8 + i];
    const Pieces p = split8(lo), q = split8(hi);
    uint4* dst = lds32 + buf * kStage32 + op * (3 * 4 * kPlane);
    dst[0 * 4 * kPlane + e0] = p.h, dst[1 * 4 * kPlane + e0] = p.m, dst[2 * 4 * kPlane + e0] = p.l;
    dst[0 * 4 * kPlane + e1] = q.h, dst[1 * 4 * kPlane + e1] = q.m, dst[2 * 4 * kPlane + e1] = q.l;
  };
  auto frag_a = [&](int buf, int p, int u) {
    return __builtin_bit_cast(bf16x8, lds32[buf * kStage32 + (p * 4 + 2 * u + half) * kPlane + wr * 32 + col]);
  };
  auto frag_b = [&](int buf, int p, int u, int j) {
    return __builtin_bit_cast(bf16x8, lds32[buf * kStage32 + 3 * 4 * kPlane + (p * 4 + 2 * u + half) * kPlane + wc * 64 + j * 32 + col]);
  };
  bf16x8 pa[3], pb[3][2];                      // prefetched fragments of sub-step 0 (PREFETCH only)
  auto mfma12 = [&](const bf16x8 (&a)[3], const bf16x8 (&b)[3][2]) {
#pragma unroll
    for (int term = 0; term < 6; ++term) {
      const int ia = term == 0 ? 2 : ((term == 1 || term == 3) ? 1 : 0);
      const int ib = term == 2 ? 2 : ((term == 1 || term == 4) ? 1 : 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ia], b[ib][j], acc[j], 0, 0, 0);
    }
  };
  // have_a / have_b: sub-step 0 fragments of that operand were prefetched
  auto multiply = [&](int cur, bool have_a, bool have_b) {
    bf16x8 a[3], b[3][2], a1[3], b1[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      a[p] = (PREFETCH && have_a) ? pa[p] : frag_a(cur, p, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[p][j] = (PREFETCH && have_b) ? pb[p][j] : frag_b(cur, p, 0, j);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      a1[p] = frag_a(cur, p, 1);
#pragma unroll
      for (int j = 0; j < 2; ++j) b1[p][j] = frag_b(cur, p, 1, j);
    }
    mfma12(a, b);
    mfma12(a1, b1);
  };

  const int stages = g.K / 32;
  load(0);
  store(0);
  if (stages > 1) load(32);
  __syncthreads();

  for (int s = 0; s < stages; ++s) {
    const int cur = s & 1;
    const bool more = s + 1 < stages, more2 = s + 2 < stages;
    // even phase: waves 0-3 multiply stage s | waves 4-7 stage B of s + 1, then prefetch their fragments of stage s
    if (!second) {
      multiply(cur, false, PREFETCH && s > 0);
    } else {
      if (more) {
        store(cur ^ 1);
        if (more2) load((s + 2) * 32);
      }
      if (PREFETCH) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          pa[p] = frag_a(cur, p, 0);
#pragma unroll
          for (int j = 0; j < 2; ++j) pb[p][j] = frag_b(cur, p, 0, j);
        }
      }
    }
    __syncthreads();
    // odd phase: waves 4-7 multiply stage s | waves 0-3 stage A of s + 1, then prefetch B fragments of stage s + 1
    if (second) {
      multiply(cur, PREFETCH, PREFETCH);
    } else if (more) {
      store(cur ^ 1);
      if (more2) load((s + 2) * 32);
      if (PREFETCH) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int j = 0; j < 2; ++j) pb[p][j] = frag_b(cur ^ 1, p, 0, j);
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wc * 64 + j * 32 + col;
    if (n >= g.N) continue;
    const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (m < g.M) g.C[(long long)m * g.ldc + n] = acc[j][r] + bv;
    }
  }
}

template <bool PREFETCH>
void launch_pp32(const float* A, long long lda, int akc, const float* B, long long ldb, int bkc, float* C, long long ldc, int M,
                 int N, int K, hipStream_t stream) {
  GemmArgs g;
  g.A = A, g.B = B, g.C = C, g.bias = nullptr;
  g.lda = lda, g.ldb = ldb, g.ldc = ldc;
  g.M = M, g.N = N, g.K = K;
  g.tiles_n = (N + kBN - 1) / kBN;
  const unsigned blocks = (unsigned)(((M + kBM - 1) / kBM) * g.tiles_n);
  const size_t lds = sizeof(uint4) * 2 * kStage32;
#define PP32(a, b)                                                                                                     \
  do {                                                                                                                 \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pp32_kernel<a, b, PREFETCH>),                          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                   \
    hipLaunchKernelGGL((gemm_pp32_kernel<a, b, PREFETCH>), dim3(blocks), dim3(512), lds, stream, g);                    \
  } while (0)
  if (akc && bkc) PP32(true, true);
  else if (akc) PP32(true, false);
  else if (bkc) PP32(false, true);
  else PP32(false, false);
#undef PP32
}

template <int VARIANT>
void launch_pp(const float* A, long long lda, int akc, const float* B, long long ldb, int bkc, float* C, long long ldc, int M,
               int N, int K, hipStream_t stream) {
  GemmArgs g;
  g.A = A, g.B = B, g.C = C, g.bias = nullptr;
  g.lda = lda, g.ldb = ldb, g.ldc = ldc;
  g.M = M, g.N = N, g.K = K;
  g.tiles_n = (N + kBN - 1) / kBN;
  const unsigned blocks = (unsigned)(((M + kBM - 1) / kBM) * g.tiles_n);
  if (akc && bkc) hipLaunchKernelGGL((gemm_pp_kernel<true, true, VARIANT>), dim3(blocks), dim3(512), 0, stream, g);
  else if (akc) hipLaunchKernelGGL((gemm_pp_kernel<true, false, VARIANT>), dim3(blocks), dim3(512), 0, stream, g);
  else if (bkc) hipLaunchKernelGGL((gemm_pp_kernel<false, true, VARIANT>), dim3(blocks), dim3(512), 0, stream, g);
  else hipLaunchKernelGGL((gemm_pp_kernel<false, false, VARIANT>), dim3(blocks), dim3(512), 0, stream, g);
}

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

float median(std::vector<float> v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

}  // namespace

int main() {
  struct Shape {
    int M, N, K, akc, bkc;
    const char* what;
  };
  // forward x.W^T: A [M,K] kc, B [N,K] kc.  input gradient dy.W: A [M,K'] kc, B = W [K', N'] stored [K, ldb] (k strided).
  // weight-gradient-like (both strided) is timed too, without its split-K.
  const Shape shapes[] = {
      {18432, 384, 384, 1, 1, "fwd 18432x384x384"},    {18432, 1152, 384, 1, 1, "fwd 18432x1152x384"},
      {18432, 768, 384, 1, 1, "fwd 18432x768x384"},    {18432, 384, 768, 1, 1, "fwd 18432x384x768"},
      {18432, 1024, 1024, 1, 1, "fwd 18432x1024x1024"}, {65536, 512, 512, 1, 1, "fwd 65536x512x512"},
      {8192, 384, 384, 1, 1, "fwd 8192x384x384"},      {8192, 1152, 384, 1, 1, "fwd 8192x1152x384"},
      {18432, 384, 384, 1, 0, "dgrad 18432x384x384"},  {18432, 384, 1152, 1, 0, "dgrad 18432x384x1152"},
      {18432, 1024, 1024, 1, 0, "dgrad 18432x1024x1024"}, {8192, 384, 384, 1, 0, "dgrad 8192x384x384"},
  };
  hipStream_t stream;
  CK(hipStreamCreate(&stream));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%-24s %10s %10s %10s %10s   %s\n", "shape", "split3 us", "base occ2", "pp32 us", "base occ3",
         "TFLOP/s (fp32 product) library kernel / probe copy occ 2 / pp32 / probe copy occ 3; bitwise");
  for (const Shape& s : shapes) {
    const size_t na = (size_t)s.M * s.K, nb = (size_t)s.N * s.K, nc = (size_t)s.M * s.N;
    std::vector<float> ha(na), hb(nb);
    uint32_t seed = 12345u + s.M + 3 * s.N + 7 * s.K;
    auto rnd = [&]() {
      seed = seed * 1664525u + 1013904223u;
      return ((seed >> 8) & 0xFFFF) / 32768.0f - 1.0f;
    };
    for (auto& v : ha) v = rnd();
    for (auto& v : hb) v = rnd() * 0.05f;
    float *A, *B, *C0, *C1, *C2, *C3;
    CK(hipMalloc(&A, na * 4));
    CK(hipMalloc(&B, nb * 4));
    CK(hipMalloc(&C0, nc * 4));
    CK(hipMalloc(&C1, nc * 4));
    CK(hipMalloc(&C2, nc * 4));
    CK(hipMalloc(&C3, nc * 4));
    CK(hipMemcpy(A, ha.data(), na * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hb.data(), nb * 4, hipMemcpyHostToDevice));
    const long long lda = s.akc ? s.K : s.M, ldb = s.bkc ? s.K : s.N;
    auto run = [&](int which) {
      if (which == 0) {
        if (ptops_gemm_f32_split3(A, lda, s.akc, B, ldb, s.bkc, C0, s.N, s.M, s.N, s.K, nullptr, stream) != 0) {
          fprintf(stderr, "ptops_gemm_f32_split3 failed\n");
          exit(1);
        }
      } else if (which == 1) {
        launch_base<2>(A, lda, s.akc, B, ldb, s.bkc, C1, s.N, s.M, s.N, s.K, stream);
      } else if (which == 2) {
        launch_pp32<false>(A, lda, s.akc, B, ldb, s.bkc, C2, s.N, s.M, s.N, s.K, stream);
      } else {
        launch_base<3>(A, lda, s.akc, B, ldb, s.bkc, C3, s.N, s.M, s.N, s.K, stream);
      }
    };
    std::vector<float> ts[4];
    for (int w = 0; w < 4; ++w) run(w);
    CK(hipStreamSynchronize(stream));
    for (int round = 0; round < 7; ++round)
      for (int w = 0; w < 4; ++w) {
        CK(hipEventRecord(e0, stream));
        for (int i = 0; i < 5; ++i) run(w);
        CK(hipEventRecord(e1, stream));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ts[w].push_back(ms * 1000.f / 5);
      }
    CK(hipGetLastError());
    std::vector<float> h0(nc), h1(nc), h2(nc), h3(nc);
    CK(hipMemcpy(h3.data(), C3, nc * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h0.data(), C0, nc * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1.data(), C1, nc * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), C2, nc * 4, hipMemcpyDeviceToHost));
    const bool same = memcmp(h0.data(), h1.data(), nc * 4) == 0 && memcmp(h0.data(), h2.data(), nc * 4) == 0 &&
                      memcmp(h0.data(), h3.data(), nc * 4) == 0;
    const double fl = 2.0 * s.M * s.N * s.K;
    const float m0 = median(ts[0]), m1 = median(ts[1]), m2 = median(ts[2]), m3 = median(ts[3]);
    printf("%-24s %10.1f %10.1f %10.1f %10.1f   %6.1f / %6.1f / %6.1f / %6.1f   %s\n", s.what, m0, m1, m2, m3, fl / m0 * 1e-6,
           fl / m1 * 1e-6, fl / m2 * 1e-6, fl / m3 * 1e-6, same ? "identical" : "DIFFERENT");
    fflush(stdout);
    CK(hipFree(A));
    CK(hipFree(B));
    CK(hipFree(C0));
    CK(hipFree(C1));
    CK(hipFree(C2));
    CK(hipFree(C3));
  }
  return 0;
}
