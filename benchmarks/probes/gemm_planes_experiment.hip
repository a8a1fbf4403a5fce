// fp32 GEMM on the bf16 matrix pipe from PRE-SPLIT operands: C[M,N] = A[M,K] . B[N,K]^T (+ bias[N]), gfx950 (MI355X).
//
// Same arithmetic as gemm_split3.hip (x = h + m + l in bf16, six piece products per k-step, fp32 accumulation), but the
// split is done once per operand by ptops_split3_pack instead of by every tile that touches the operand (in
// gemm_split3_kernel an activation is split N/128 times and a weight M/128 times), and the pack writes the pieces in
// the order the matrix pipe consumes them ("fragment-major"):
//
//   block (rb = row / 32, kc = k / 16) = 3072 contiguous bytes = [piece 0..2][k-half 0..1][row % 32] entries of 8 bf16
//   operand = [rb][kc] blocks, rows padded with zeros to a multiple of 32
//
// so that (a) one global_load_lds_dwordx4 wave-instruction copies 1 KB of global memory that is contiguous to 1 KB of
// LDS that is contiguous (no VGPRs, no ds_write, no vector ALU work in the loop), and (b) a wave's ds_read_b128 of one
// MFMA operand (32 rows x 2 k-halves) is again 1 KB contiguous: conflict-free without padding or swizzle.
//
// Tile 256 x 128 per workgroup, 8 waves as 4 (M) x 2 (N), 64 x 64 per wave (2 x 2 MFMA blocks); stage = 32 k (two
// 16-deep sub-steps): 72 KB of LDS per stage, two stages, one workgroup per CU.  Per stage and wave: 9 LDS-DMA
// instructions, 24 ds_read_b128, 48 MFMAs (1536 matrix cycles per wave, 3072 per SIMD) and ONE barrier — the matrix phase
// is ~10x longer than in the 128 x 128 / k-step-16 loop of gemm_split3.hip, whose k-step is two thirds LDS bursts and
// barrier skew (profiles/round2/gemm_split3_notes.txt).
#include "ptops_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kTM = 256, kTN = 128;
constexpr int kBlockEntries = 192;                 // 16-byte entries of one (32 rows x 16 k) block: 3 x 2 x 32
constexpr int kStageA = (kTM / 32) * 2 * kBlockEntries;   // 3072 entries = 48 KB
constexpr int kStageB = (kTN / 32) * 2 * kBlockEntries;   // 1536 entries = 24 KB
constexpr int kStage = kStageA + kStageB;

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

// ---- pack: fp32 [rows, K] (element (r, k) at x[r * rs + k * ks]) -> fragment-major bf16 pieces ---------------------
// One workgroup = one row block x four k-chunks (32 rows x 64 k): coalesced reads when ks == 1 (8 lanes = 256 bytes of
// a row), pieces to LDS in output order, 12 KB written back linearly.
__global__ __launch_bounds__(256) void split3_pack_kernel(const float* __restrict__ x, long long rs, long long ks, int rows,
                                                          int K, uint4* __restrict__ out) {
  __shared__ uint4 sOut[4 * kBlockEntries];
  const int t = threadIdx.x;
  const int KC = K >> 4;
  const int rb = blockIdx.y, kc0 = blockIdx.x * 4;
  const int row = t >> 3, oct = t & 7;
  const int r = rb * 32 + row;
  const int kc = kc0 + (oct >> 1), half = oct & 1;
  float v[8];
  if (r < rows && kc < KC) {
    const float* p = x + (long long)r * rs + (long long)(kc * 16 + half * 8) * ks;
    if (ks == 1 && (((uintptr_t)p) & 15) == 0) {
      const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
      v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = p[(long long)i * ks];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
  }
  const Pieces pc = split8(v);
  const int base = (oct >> 1) * kBlockEntries + half * 32 + row;
  sOut[base] = pc.h;
  sOut[base + 64] = pc.m;
  sOut[base + 128] = pc.l;
  __syncthreads();
  uint4* dst = out + ((long long)rb * KC + kc0) * kBlockEntries;
  const int valid = min(4, KC - kc0) * kBlockEntries;
#pragma unroll
  for (int e = t; e < 4 * kBlockEntries; e += 256)
    if (e < valid) dst[e] = sOut[e];
}

struct PlanesArgs {
  const uint4* A;   // packed [ceil(M/32)][K/16][192]
  const uint4* B;   // packed [ceil(N/32)][K/16][192]
  float* C;
  const float* bias;
  long long ldc;
  int M, N, K;
  int tiles_n;
};

__device__ __forceinline__ void lds_dma_1k(const uint4* gsrc_lane, uint4* lds_wave_base) {
  // 64 lanes x 16 bytes: lane l's 16 bytes land at lds_wave_base + l
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__global__ __launch_bounds__(512) void gemm_planes_kernel(PlanesArgs g) {
  extern __shared__ uint4 lds[];                   // [2][kStage]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int col = lane & 31, half = lane >> 5;
  int bid;
  {
    const int orig = blockIdx.x, nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  }
  const int tile_m = bid / g.tiles_n, tile_n = bid - tile_m * g.tiles_n;
  const int KC = g.K >> 4;
  const int rbA_last = (g.M + 31) / 32 - 1, rbB_last = (g.N + 31) / 32 - 1;

  // the 9 DMA instructions of this wave per stage: instruction j = wave + 8 i covers 1 KB (one sixth of a row block's
  // 6 KB); j < 48: A row block j / 6, else B row block (j - 48) / 6
  const uint4* src[9];
  int dst[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int j = wave + 8 * i;
    const bool isB = j >= 48;
    const int jj = isB ? j - 48 : j;
    const int rbl = jj / 6, part = jj - rbl * 6;
    const int rb = isB ? min(tile_n * (kTN / 32) + rbl, rbB_last) : min(tile_m * (kTM / 32) + rbl, rbA_last);
    src[i] = (isB ? g.B : g.A) + (long long)rb * KC * kBlockEntries + part * 64 + lane;
    dst[i] = (isB ? kStageA : 0) + rbl * 2 * kBlockEntries + part * 64;
  }
  auto issue = [&](int s, int buf) {
    const int koff = s * 2 * kBlockEntries;        // two k-chunks per stage
#pragma unroll
    for (int i = 0; i < 9; ++i) lds_dma_1k(src[i] + koff, lds + buf * kStage + dst[i]);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int stages = g.K / 32;
  issue(0, 0);
  for (int s = 0; s < stages; ++s) {
    const int buf = s & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                               // stage s has landed for every wave; the other buffer is free
    if (s + 1 < stages) issue(s + 1, buf ^ 1);
    const uint4* sA = lds + buf * kStage;
    const uint4* sB = sA + kStageA;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[i][p] = __builtin_bit_cast(bf16x8, sA[((wm * 2 + i) * 2 + c) * kBlockEntries + (p * 2 + half) * 32 + col]);
          b[i][p] = __builtin_bit_cast(bf16x8, sB[((wn * 2 + i) * 2 + c) * kBlockEntries + (p * 2 + half) * 32 + col]);
        }
      // six piece products per block, smallest first; four independent accumulators alternate
#pragma unroll
      for (int term = 0; term < 6; ++term) {
        const int pa = term == 0 ? 2 : ((term == 1 || term == 3) ? 1 : 0);
        const int pb = term == 2 ? 2 : ((term == 1 || term == 4) ? 1 : 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[j][pb], acc[i][j], 0, 0, 0);
      }
    }
  }

  // epilogue: block (i, j) register r holds C[m0 + (r&3) + 8 (r>>2) + 4 half][n0 + col]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = tile_n * kTN + (wn * 2 + j) * 32 + col;
    if (n >= g.N) continue;
    const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = tile_m * kTM + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < g.M) g.C[(long long)m * g.ldc + n] = acc[i][j][r] + bv;
      }
    }
  }
}

}  // namespace

// bytes of the packed form of a [rows, K] operand (K % 16 == 0)
PTOPS_API long long ptops_split3_pack_bytes(int rows, int K) {
  if (rows <= 0 || K <= 0) return 0;
  return (long long)((rows + 31) / 32) * (K / 16) * kBlockEntries * 16;
}

// x: fp32, element (r, k) at x[r * row_stride + k * k_stride]  ->  out (16-byte aligned, ptops_split3_pack_bytes)
PTOPS_API int ptops_split3_pack(const float* x, long long row_stride, long long k_stride, int rows, int K, void* out,
                                void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(rows >= 0 && K >= 0, "split3_pack: negative size");
  if (rows == 0 || K == 0) return PTOPS_OK;
  PTOPS_REQUIRE(x && out, "split3_pack: null pointer");
  PTOPS_REQUIRE(K % 16 == 0, "split3_pack: K must be a multiple of 16 (K=%d)", K);
  PTOPS_REQUIRE(((uintptr_t)out & 15) == 0, "split3_pack: out must be 16-byte aligned");
  dim3 grid((unsigned)ptops::ceil_div(K / 16, 4), (unsigned)ptops::ceil_div(rows, 32));
  hipLaunchKernelGGL(split3_pack_kernel, grid, dim3(256), 0, stream, x, row_stride, k_stride, rows, K, (uint4*)out);
  return ptops::check_launch("split3_pack_kernel");
}

// C[M,N] = A . B^T (+ bias) from packed operands (ptops_split3_pack of A [M,K] and B [N,K]).  K % 32 == 0.
PTOPS_API int ptops_gemm_f32_planes(const void* A_packed, const void* B_packed, float* C, long long ldc, int M, int N, int K,
                                    const float* bias, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_f32_planes: negative size");
  if (M == 0 || N == 0) return PTOPS_OK;
  PTOPS_REQUIRE(A_packed && B_packed && C, "gemm_f32_planes: null pointer");
  PTOPS_REQUIRE(K >= 32 && K % 32 == 0, "gemm_f32_planes: K must be a positive multiple of 32 (K=%d)", K);
  PTOPS_REQUIRE((((uintptr_t)A_packed | (uintptr_t)B_packed) & 15) == 0, "gemm_f32_planes: operands must be 16-byte aligned");
  static bool attr_set = false;
  constexpr int lds_bytes = 2 * kStage * 16;
  if (!attr_set) {
    PTOPS_REQUIRE(hipFuncSetAttribute((const void*)gemm_planes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) ==
                      hipSuccess,
                  "gemm_f32_planes: cannot reserve %d bytes of LDS", lds_bytes);
    attr_set = true;
  }
  PlanesArgs g;
  g.A = (const uint4*)A_packed, g.B = (const uint4*)B_packed, g.C = C, g.bias = bias, g.ldc = ldc;
  g.M = M, g.N = N, g.K = K;
  g.tiles_n = ptops::ceil_div(N, kTN);
  const long long blocks = (long long)ptops::ceil_div(M, kTM) * g.tiles_n;
  PTOPS_REQUIRE(blocks < (1LL << 31), "gemm_f32_planes: grid too large");
  hipLaunchKernelGGL(gemm_planes_kernel, dim3((unsigned)blocks), dim3(512), lds_bytes, stream, g);
  return ptops::check_launch("gemm_planes_kernel");
}
