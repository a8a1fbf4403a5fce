// fp32 GEMM on the bf16 matrix pipe: C[M,N] = A[M,K] . B[N,K]^T (+ bias[N]), fp32 in, fp32 out, gfx950 (MI355X).
//
// The dense layers of the path (models/Transformer_utils.py:71-159 of the reference: nn.Linear forward and backward)
// are fp32 GEMMs; gfx950 has no TF32-like mode and its f32 MFMA runs at the f32 vector rate (157 TFLOP/s), where the
// library kernels already sit at 70-90 %.  The bf16 MFMA is 16x faster, so each f32 operand is split in the kernel into
// three bf16 pieces, x = h + m + l (round to nearest; exact to 2^-27 |x|), and every k-step issues the six piece
// products down to 2^-18: Ah.Bh, Ah.Bm, Am.Bh, Ah.Bl, Am.Bm, Al.Bh.  bf16 x bf16 products are exact in f32 and the
// MFMA accumulates in f32 (measured in chamfer_mfma.hip: like a single-rounding f32 sum), the dropped products are
// <= 3 * 2^-26 |a||b| per term: the result is as close to the exact product as an f32 FMA chain is (tests compare both
// with float64).  6 MFMA per k-step at 16x the rate = 2.7x the f32 matrix peak at equal clocks.
//
// Tiling: 128 x 128 output tile per workgroup (8 waves, 32 x 64 each = 1 x 2 MFMA blocks of 32 x 32), k-step 16.
// Staging: every thread loads the 8 f32 of ONE operand entry (row, k-half), splits them and writes the three 16-byte
// bf16 entries to LDS ([piece][k-half][row], conflict-free ds_read_b128); the next k-step's loads are in flight while
// the current one computes; double-buffered LDS, one barrier per k-step.  Either operand may be stored with k
// contiguous ([rows, ld]) or with k strided ([K, ld], rows contiguous): the strided form reads 8 coalesced dwords
// instead of two dwordx4, so A.B^T (forward), A.B (input gradient) and A^T.B (weight gradient) use one kernel.
#include <string.h>

#include <algorithm>
#include <vector>

#include "ptops_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kBM = 128, kBN = 128, kBK = 16;
// entries per k-half plane: 128 rows + 4 of padding, so that the 8 lanes of a ds_write_b128 group (rows r .. r+3 of both
// k-halves when k is contiguous) fall on 8 different 16-byte bank groups; reads stay consecutive within a half
constexpr int kPlane = 132;

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;   // [N] or null
  float* G;            // optional second output: gelu(C) (exact erf form), leading dimension ldg
  long long lda, ldb, ldc, ldg;
  int M, N, K;
  int tiles_n;
  int debug_no_store;  // PTOPS_GEMM_DEBUG_NO_STORE=1: the epilogue writes nothing (timing experiments only)
};

struct Pieces {
  uint4 h, m, l;
};

// 8 f32 -> three 8 x bf16 entries
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

// One operand entry of the tile that starts at row r0 / k-step k0: 8 consecutive k of one row.
//   KC (k contiguous, [rows, ld]): thread t -> row t >> 1, k-half t & 1
//   else (k strided,  [K, ld])   : thread t -> row t & 127, k-half t >> 7
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

// 8 waves per 128 x 128 tile: wave w owns rows (w >> 1) * 32 .. + 31 and columns (w & 1) * 64 .. + 63 (two MFMA blocks).
// Thread t < 256 stages A entry t, thread t >= 256 stages B entry t - 256 (one entry = 8 f32 of one row -> 3 x 16 bytes).
// With 4 waves per tile and 432 tiles (M = 18432, N = 384) a SIMD mostly hosts ONE wave, whose LDS / staging / barrier
// phases leave the matrix pipe idle; 8 waves put two waves of every tile on each SIMD (107 -> ? TFLOP/s).
// Round 3 (benchmarks/probes/gemm_pingpong_probe.hip, profiles/round3/gemm_pingpong_v*.log): a "ping-pong" role split of
// the two wave halves (one half multiplies while the other splits / stores, a barrier per phase) is 3-12 % SLOWER than this
// loop with 16-deep steps and 20-30 % slower with 32-deep stages; three workgroups per CU instead of two (OCC = 3, <= 80
// VGPRs) is +1..4 % in the kernel alone but -0.08 ms in the train step (profiles/round3/step_ab_v1.log): not adopted.
template <bool AKC, bool BKC, int OCC = 2>
__global__ __launch_bounds__(512, OCC) void gemm_split3_kernel(GemmArgs g) {
  // [buffer][operand][piece][k-half * kPlane + row]
  __shared__ uint4 lds[2][2][3][2 * kPlane];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int col = lane & 31, half = lane >> 5;
  // Workgroups b and b + 8 share an XCD (round-robin dispatch): give every XCD one contiguous range of tiles, so that
  // the tiles_n tiles that read the same 128 rows of A do so through one L2 (bijective for any grid; speed only).
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

  // (Tried: waves w and w + 4 — SIMD partners — staggered, one splitting / storing the next k-step before its MFMAs
  // and one after, with loads two k-steps ahead: 112 VGPRs instead of 79 and 3-30 % slower.)
  // Tried on this loop, none faster (profiles/round2/gemm_split3_notes.txt): global loads two k-steps ahead with a raw
  // s_barrier so that they stay in flight (0-8 % slower: 9 more VGPRs, 8 v_mov per step); SIMD partners (waves w, w + 4)
  // staggered, one splitting / storing before its MFMAs and one after (3-30 % slower); 4 waves of 64 x 64 (10 % slower:
  // one wave per SIMD on most CUs at 432 tiles).
  const int ksteps = g.K / kBK;
  for (int ks = 0; ks < ksteps; ++ks) {
    const int cur = ks & 1;
    const bool more = ks + 1 < ksteps;
    if (more) load((ks + 1) * kBK);          // next k-step: global -> registers now, split + LDS after the math
    bf16x8 a[3], b[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      a[p] = __builtin_bit_cast(bf16x8, lds[cur][0][p][half * kPlane + wr * 32 + col]);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[p][j] = __builtin_bit_cast(bf16x8, lds[cur][1][p][half * kPlane + wc * 64 + j * 32 + col]);
    }
    // six piece products per block, smallest first; the two blocks alternate so that no MFMA waits on the one before
#pragma unroll
    for (int term = 0; term < 6; ++term) {
      // (piece of A, piece of B): (l,h) (m,m) (h,l) (m,h) (h,m) (h,h)
      const int pa = term == 0 ? 2 : ((term == 1 || term == 3) ? 1 : 0);
      const int pb = term == 2 ? 2 : ((term == 1 || term == 4) ? 1 : 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b[pb][j], acc[j], 0, 0, 0);
    }
    if (more) store(cur ^ 1);
    __syncthreads();
  }

  // epilogue: block j register r holds C[m0 + wr*32 + (r&3) + 8 (r>>2) + 4 half][n0 + wc*64 + j*32 + col]
  if (g.debug_no_store) {                      // keep the accumulators alive without the 64 KB of stores
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 1.2345e-30f) g.C[0] = s;
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wc * 64 + j * 32 + col;
    if (n >= g.N) continue;
    const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (m < g.M) {
        const float u = acc[j][r] + bv;
        g.C[(long long)m * g.ldc + n] = u;
        // nn.GELU() of the pre-activation in the same pass (x * 0.5 * (1 + erf(x / sqrt 2)), as aten's kernel)
        if (g.G) g.G[(long long)m * g.ldg + n] = (u * 0.5f) * (1.0f + erff(u * 0.70710678118654752440f));
      }
    }
  }
}


// ---- weight gradient: dW[N,K] = dY[M,N]^T . X[M,K] (+ db[N] = column sums of dY), contraction over the M rows -------
// Same tile, staging and arithmetic as gemm_split3_kernel<false, false> (both operands "k strided": an operand entry is
// 8 consecutive rows m of one column, read as 8 dwords that are coalesced across the lanes), but the output is small
// (N x K of a layer: 9 .. 108 tiles) and the contraction long (M = 8192 .. 65536), so the k-steps are split over
// gridDim.y workgroups per tile: each writes its 128 x 128 partial to the workspace and wgrad_reduce_kernel adds the
// partials in a fixed order (deterministic, no atomics).  The column sums of dY (the layer's bias gradient) fall out of
// the A staging: every A entry passes through one thread's registers exactly once.
struct WgradArgs {
  const float* dY;   // [M, ldy]
  const float* X;    // [M, ldx]
  float* part;       // [splits][N][K] partial products
  float* bpart;      // [splits][N] partial column sums, or null
  long long ldy, ldx;
  int M, N, K;       // M: contraction; output N x K
  int tiles_k;       // tiles along K (the output's columns)
  int steps_per_split;
};

// One 128 x 128 tile of dY^T . X over the k-steps [ks_begin, ks_end) (16 rows of the contraction each): the product goes
// to part[N][K], the column sums of the dY rows it staged to bpart[N] (when not null; the caller passes it to ONE of the
// tiles that share these dY columns).
__device__ __forceinline__ void wgrad_tile(const float* __restrict__ dY, const float* __restrict__ X, long long ldy,
                                           long long ldx, int N, int K, int n0, int c0, int ks_begin, int ks_end,
                                           float* __restrict__ part, float* __restrict__ bpart,
                                           uint4 (&lds)[2][2][3][2 * kPlane], float (&sB)[256]) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int col = lane & 31, half = lane >> 5;
  const bool stage_b = t >= 256;
  const int tt = t & 255;
  const int e = entry_index<false>(tt);

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  float x[8];
  // A stagers: sum of this thread's dY entries, in two levels (16 k-steps go to `blo`, which is then added to `bhi`):
  // a job stages up to a few thousand rows, and one running sum would carry that many roundings
  float blo = 0.f, bhi = 0.f;
  int bcount = 0;
  auto load = [&](int ks) {
    if (stage_b) load_entry<false>(X, ldx, K, c0, ks * kBK, tt, x);
    else load_entry<false>(dY, ldy, N, n0, ks * kBK, tt, x);
  };
  auto store = [&](int buf) {
    blo += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
    if (++bcount == 16) bhi += blo, blo = 0.f, bcount = 0;
    const Pieces p = split8(x);
    uint4(*dst)[2 * kPlane] = lds[buf][stage_b ? 1 : 0];
    dst[0][e] = p.h, dst[1][e] = p.m, dst[2][e] = p.l;
  };
  if (ks_begin < ks_end) {
    load(ks_begin);
    store(0);
  }
  __syncthreads();
  for (int ks = ks_begin; ks < ks_end; ++ks) {
    const int cur = (ks - ks_begin) & 1;
    const bool more = ks + 1 < ks_end;
    if (more) load(ks + 1);
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
    const int c = c0 + wc * 64 + j * 32 + col;
    if (c >= K) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (n < N) part[(long long)n * K + c] = acc[j][r];
    }
  }
  if (bpart) {                                                 // uniform per workgroup
    if (!stage_b) sB[tt] = bhi + blo;                          // entry tt: row tt & 127, k-half tt >> 7
    __syncthreads();
    if (t < 128 && n0 + t < N) bpart[n0 + t] = sB[t] + sB[t + 128];
  }
}

__global__ __launch_bounds__(512, 2) void gemm_split3_wgrad_kernel(WgradArgs g) {
  __shared__ uint4 lds[2][2][3][2 * kPlane];
  __shared__ float sB[256];
  const int bid = blockIdx.x, split = blockIdx.y;
  const int tile_n = bid / g.tiles_k, tile_k = bid - tile_n * g.tiles_k;
  const int ks_total = g.M / kBK;
  const int ks_begin = split * g.steps_per_split, ks_end = min(ks_begin + g.steps_per_split, ks_total);
  // every tile_k sees the same dY rows: one of them reports the column sums
  wgrad_tile(g.dY, g.X, g.ldy, g.ldx, g.N, g.K, tile_n * kBM, tile_k * kBN, ks_begin, ks_end,
             g.part + (long long)split * g.N * g.K,
             (g.bpart && tile_k == 0) ? g.bpart + (long long)split * g.N : nullptr, lds, sB);
}

// ---- grouped weight gradients: the dW (and db) of MANY layers in one launch ------------------------------------------
// A 384 x 384 layer has 9 output tiles: on its own it needs a 50-fold split of the contraction to fill 256 CUs, and the
// 9 tiles that share its operand columns land on 8 different XCDs (8 L2s), so each operand is fetched from HBM three
// times.  The layers' weight gradients do not depend on each other, so the host queues them during the backward pass and
// one launch runs them all: a job = (layer, contraction split, tile); the planner gives every (layer, split) group of
// tiles to ONE XCD (workgroup b runs on XCD b % 8; the job table is laid out for that) so the operand slabs are shared
// through that XCD's L2, balances the XCD queues by work, and keeps the splits few (long steady-state loops, little
// partial traffic).  A second launch adds the partials of all layers in a fixed order (deterministic, no atomics).
struct WgradProblem {     // == PtopsWgradProblem (include/ptops.h), 72 bytes
  const float* dY;
  const float* X;
  float* dW;
  float* db;              // or null
  long long ldy, ldx, lddw;
  int M, N, K, pad;
};
struct WgradPlanHeader {  // 32 bytes
  int magic, n_problems, n_jobs, n_rjobs;
  long long workspace_bytes, reserved;
};
struct WgradPlanProblem { // 32 bytes; bases in floats from the workspace start
  long long part_base, bpart_base;
  int splits, steps_per_split, tiles_k, pad;
};
struct WgradJob {         // 16 bytes; problem < 0: padding of a shorter XCD queue
  int problem, tile, split, pad;
};
struct WgradRJob {        // 16 bytes: 1024-float chunk `chunk` of dW (kind 0) or db (kind 1) of a problem
  int problem, kind, chunk, pad;
};
constexpr int kWgradMagic = 0x57475233;

template <int OCC>   // waves per SIMD the register budget must allow: 2 = the default (two workgroups per CU need four)
__global__ __launch_bounds__(512, OCC) void gemm_split3_wgrad_grouped_kernel(const WgradProblem* __restrict__ probs,
                                                                           const WgradPlanProblem* __restrict__ plan,
                                                                           const WgradJob* __restrict__ jobs,
                                                                           float* __restrict__ ws) {
  __shared__ uint4 lds[2][2][3][2 * kPlane];
  __shared__ float sB[256];
  const WgradJob job = jobs[blockIdx.x];
  if (job.problem < 0) return;
  const WgradProblem P = probs[job.problem];
  const WgradPlanProblem Q = plan[job.problem];
  const int tile_n = job.tile / Q.tiles_k, tile_k = job.tile - tile_n * Q.tiles_k;
  const int ks_begin = job.split * Q.steps_per_split, ks_end = min(ks_begin + Q.steps_per_split, P.M / kBK);
  wgrad_tile(P.dY, P.X, P.ldy, P.ldx, P.N, P.K, tile_n * kBM, tile_k * kBN, ks_begin, ks_end,
             ws + Q.part_base + (long long)job.split * P.N * P.K,
             (P.db && tile_k == 0) ? ws + Q.bpart_base + (long long)job.split * P.N : nullptr, lds, sB);
}

__global__ __launch_bounds__(256) void wgrad_grouped_reduce_kernel(const WgradProblem* __restrict__ probs,
                                                                   const WgradPlanProblem* __restrict__ plan,
                                                                   const WgradRJob* __restrict__ rjobs,
                                                                   const float* __restrict__ ws) {
  const WgradRJob rj = rjobs[blockIdx.x];
  const WgradProblem P = probs[rj.problem];
  const WgradPlanProblem Q = plan[rj.problem];
  const long long count = rj.kind == 0 ? (long long)P.N * P.K : (long long)P.N;
  const long long i = (long long)rj.chunk * 1024 + threadIdx.x * 4;
  if (i >= count) return;
  const float* part = ws + (rj.kind == 0 ? Q.part_base : Q.bpart_base);
  float4 s = *reinterpret_cast<const float4*>(part + i);
  for (int k = 1; k < Q.splits; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(part + (long long)k * count + i);
    s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
  }
  if (rj.kind == 0) {
    const long long row = i / P.K;
    const int c = (int)(i - row * P.K);
    *reinterpret_cast<float4*>(P.dW + row * P.lddw + c) = s;
  } else {
    *reinterpret_cast<float4*>(P.db + i) = s;
  }
}

// out[i] = sum over the splits, in order; 4 elements per thread
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, long long count, int splits,
                                                           float* __restrict__ out, long long ld_out, int cols) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= count) return;
  float4 s = *reinterpret_cast<const float4*>(part + i);
  for (int k = 1; k < splits; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(part + (long long)k * count + i);
    s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
  }
  const long long row = i / cols;
  const int c = (int)(i - row * cols);
  *reinterpret_cast<float4*>(out + row * ld_out + c) = s;
}

// splits of the contraction: about two workgroups per CU in total, at least 8 k-steps each
static int wgrad_splits(int M, int N, int K) {
  const int tiles = ptops::ceil_div(N, kBM) * ptops::ceil_div(K, kBN);
  const int ksteps = M / kBK;
  int s = ptops::ceil_div(512, tiles);
  s = s < 1 ? 1 : s;
  const int max_s = ksteps / 8 > 0 ? ksteps / 8 : 1;
  s = s > max_s ? max_s : s;
  s = s > 64 ? 64 : s;
  const int per = ptops::ceil_div(ksteps, s);
  return ptops::ceil_div(ksteps, per);                         // no empty split
}

}  // namespace

// C[M,N] = opA . opB^T + bias.  a_k_contiguous: A is stored [M, lda] (k contiguous) else [K, lda] (m contiguous);
// b_k_contiguous: B is stored [N, ldb] else [K, ldb].  K % 16 == 0; a k-contiguous operand needs 16-byte aligned rows.
static int gemm_split3_launch(const float* A, long long lda, int a_k_contiguous, const float* B, long long ldb,
                              int b_k_contiguous, float* C, long long ldc, float* G, long long ldg, int M, int N, int K,
                              const float* bias, hipStream_t stream) {
  PTOPS_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_f32_split3: negative size");
  if (M == 0 || N == 0) return PTOPS_OK;
  PTOPS_REQUIRE(A && B && C, "gemm_f32_split3: null pointer");
  PTOPS_REQUIRE(K >= 16 && K % 16 == 0, "gemm_f32_split3: K must be a positive multiple of 16 (K=%d)", K);
  if (a_k_contiguous) PTOPS_REQUIRE(lda % 4 == 0 && ((uintptr_t)A & 15) == 0, "gemm_f32_split3: A rows must be 16-byte aligned");
  if (b_k_contiguous) PTOPS_REQUIRE(ldb % 4 == 0 && ((uintptr_t)B & 15) == 0, "gemm_f32_split3: B rows must be 16-byte aligned");
  GemmArgs g;
  g.A = A, g.B = B, g.C = C, g.bias = bias, g.G = G;
  g.lda = lda, g.ldb = ldb, g.ldc = ldc, g.ldg = ldg;
  g.M = M, g.N = N, g.K = K;
  g.tiles_n = ptops::ceil_div(N, kBN);
  g.debug_no_store = ptops::option("PTOPS_GEMM_DEBUG_NO_STORE");
  const long long blocks = (long long)ptops::ceil_div(M, kBM) * g.tiles_n;
  PTOPS_REQUIRE(blocks < (1LL << 31), "gemm_f32_split3: grid too large");
  dim3 grid((unsigned)blocks), block(512);
  if (a_k_contiguous && b_k_contiguous) {
    if (ptops::option("PTOPS_GEMM_OCC3")) hipLaunchKernelGGL((gemm_split3_kernel<true, true, 3>), grid, block, 0, stream, g);   // A/B
    else hipLaunchKernelGGL((gemm_split3_kernel<true, true>), grid, block, 0, stream, g);
  }
  else if (a_k_contiguous) hipLaunchKernelGGL((gemm_split3_kernel<true, false>), grid, block, 0, stream, g);
  else if (b_k_contiguous) hipLaunchKernelGGL((gemm_split3_kernel<false, true>), grid, block, 0, stream, g);
  else hipLaunchKernelGGL((gemm_split3_kernel<false, false>), grid, block, 0, stream, g);
  return ptops::check_launch("gemm_split3_kernel");
}

PTOPS_API int ptops_gemm_f32_split3(const float* A, long long lda, int a_k_contiguous, const float* B, long long ldb,
                                    int b_k_contiguous, float* C, long long ldc, int M, int N, int K, const float* bias,
                                    void* stream) {
  return gemm_split3_launch(A, lda, a_k_contiguous, B, ldb, b_k_contiguous, C, ldc, nullptr, 0, M, N, K, bias,
                            (hipStream_t)stream);
}

// Forward of Linear + nn.GELU(): C = A . B^T + bias (the pre-activation, kept for backward) and G = gelu(C) from the
// same epilogue (A [M, lda], B [N, ldb], both k contiguous).
PTOPS_API int ptops_gemm_f32_split3_gelu(const float* A, long long lda, const float* B, long long ldb, float* C,
                                         long long ldc, float* G, long long ldg, int M, int N, int K, const float* bias,
                                         void* stream) {
  PTOPS_REQUIRE(G != nullptr || M == 0 || N == 0, "gemm_f32_split3_gelu: null output");
  return gemm_split3_launch(A, lda, 1, B, ldb, 1, C, ldc, G, ldg, M, N, K, bias, (hipStream_t)stream);
}

PTOPS_API long long ptops_gemm_f32_split3_wgrad_workspace(int M, int N, int K) {
  if (M < 16 || N < 1 || K < 1) return 0;
  const long long s = wgrad_splits(M, N, K);
  return s * ((long long)N * K + N) * (long long)sizeof(float);
}

// dW[N,K] = dY[M,N]^T . X[M,K];  db[N] = column sums of dY when db != null.  M % 16 == 0, K % 4 == 0, lddw % 4 == 0.
PTOPS_API int ptops_gemm_f32_split3_wgrad(const float* dY, long long ldy, const float* X, long long ldx, int M, int N,
                                          int K, float* dW, long long lddw, float* db, void* workspace,
                                          long long workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_f32_split3_wgrad: negative size");
  if (N == 0 || K == 0) return PTOPS_OK;
  PTOPS_REQUIRE(dY && X && dW && workspace, "gemm_f32_split3_wgrad: null pointer");
  PTOPS_REQUIRE(M >= 16 && M % 16 == 0, "gemm_f32_split3_wgrad: M must be a positive multiple of 16 (M=%d)", M);
  PTOPS_REQUIRE(K % 4 == 0 && lddw % 4 == 0 && ((uintptr_t)dW & 15) == 0 && ((uintptr_t)workspace & 15) == 0,
                "gemm_f32_split3_wgrad: output rows must be 16-byte aligned");
  PTOPS_REQUIRE(workspace_bytes >= ptops_gemm_f32_split3_wgrad_workspace(M, N, K), "gemm_f32_split3_wgrad: workspace too small");
  if (db) PTOPS_REQUIRE(N % 4 == 0 && ((uintptr_t)db & 15) == 0, "gemm_f32_split3_wgrad: db must be 16-byte aligned, N %% 4 == 0");
  const int splits = wgrad_splits(M, N, K);
  WgradArgs g;
  g.dY = dY, g.X = X, g.ldy = ldy, g.ldx = ldx;
  g.M = M, g.N = N, g.K = K;
  g.tiles_k = ptops::ceil_div(K, kBN);
  g.steps_per_split = ptops::ceil_div(M / kBK, splits);
  g.part = (float*)workspace;
  g.bpart = db ? g.part + (long long)splits * N * K : nullptr;
  dim3 grid((unsigned)(ptops::ceil_div(N, kBM) * g.tiles_k), (unsigned)splits);
  hipLaunchKernelGGL(gemm_split3_wgrad_kernel, grid, dim3(512), 0, stream, g);
  int rc = ptops::check_launch("gemm_split3_wgrad_kernel");
  if (rc != PTOPS_OK) return rc;
  const long long count = (long long)N * K;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ptops::ceil_div_ll(count, 1024)), dim3(256), 0, stream, g.part, count,
                     splits, dW, lddw, K);
  rc = ptops::check_launch("wgrad_reduce_kernel");
  if (rc != PTOPS_OK || !db) return rc;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ptops::ceil_div_ll(N, 1024)), dim3(256), 0, stream, g.bpart,
                     (long long)N, splits, db, (long long)N, N);
  return ptops::check_launch("wgrad_reduce_kernel");
}

// ---- grouped weight gradients: planner + launch ---------------------------------------------------------------------
namespace {

struct WgradGroup {          // the tiles of one (problem, split): one XCD runs them all
  int problem, split, tiles;
  long long cost;            // tiles x k-steps
};

// Job length in 16-row steps: explicit, or PTOPS_WGRAD_GROUP_STEPS, or chosen from the work: about 4096 jobs per launch
// (eight per workgroup slot: enough to balance the XCD queues), never shorter than 288 steps (measured flat from 96 to
// 384 at bs = 32) - so a 12-view batch (six times the rows) gets six times longer jobs instead of six
// times the partial sums.
int wgrad_group_target_steps(int steps_target, const int* mnk, int n) {
  if (steps_target > 0) return steps_target;
  const int opt = ptops::option("PTOPS_WGRAD_GROUP_STEPS");
  if (opt > 0) return opt;
  long long units = 0;                                   // tile x k-step units of the whole launch
  for (int i = 0; i < n; ++i)
    units += (long long)ptops::ceil_div(mnk[3 * i + 1], kBM) * ptops::ceil_div(mnk[3 * i + 2], kBN) * (mnk[3 * i] / kBK);
  const long long t = units / 4096;
  return (int)(t < 288 ? 288 : (t > 8192 ? 8192 : t));
}

// splits / steps per split of one problem for a target job length
void wgrad_group_split(int M, int target, int* splits, int* per) {
  const int ksteps = M / kBK;
  int s = (ksteps + target / 2) / target;
  s = s < 1 ? 1 : s;
  *per = ptops::ceil_div(ksteps, s);
  *splits = ptops::ceil_div(ksteps, *per);
}

}  // namespace

// Plans the grouped launch for n problems given as {M, N, K} triples (want_db[i] != 0: problem i also delivers db).
// Returns the size of the plan in bytes; writes it when `plan` holds at least that many (same bytes go to the device).
// The plan depends on the shapes only: a training loop builds it once.
PTOPS_API long long ptops_wgrad_group_plan(const int* mnk, const unsigned char* want_db, int n, int steps_target, void* plan,
                                           long long capacity) {
  if (!mnk || n <= 0) return 0;
  for (int i = 0; i < n; ++i)
    if (mnk[3 * i] < kBK || mnk[3 * i + 1] < 1 || mnk[3 * i + 2] < 1) return -1;
  const int target = wgrad_group_target_steps(steps_target, mnk, n);
  std::vector<WgradPlanProblem> pp((size_t)n);
  std::vector<WgradGroup> groups;
  long long ws_floats = 0, n_rjobs = 0;
  for (int i = 0; i < n; ++i) {
    const int M = mnk[3 * i], N = mnk[3 * i + 1], K = mnk[3 * i + 2];
    if (M < kBK || M % kBK || N < 1 || K < 1 || K % 4) return -1;
    if (want_db && want_db[i] && N % 4) return -1;
    int splits, per;
    wgrad_group_split(M, target, &splits, &per);
    WgradPlanProblem& q = pp[(size_t)i];
    q.splits = splits, q.steps_per_split = per, q.tiles_k = ptops::ceil_div(K, kBN), q.pad = 0;
    q.part_base = ws_floats;
    ws_floats += (long long)splits * N * K;
    q.bpart_base = ws_floats;
    if (want_db && want_db[i]) ws_floats += ((long long)splits * N + 3) / 4 * 4;
    const int tiles = ptops::ceil_div(N, kBM) * q.tiles_k;
    const int ksteps = M / kBK;
    for (int s = 0; s < splits; ++s) {
      const int len = std::min(per, ksteps - s * per);
      groups.push_back({i, s, tiles, (long long)tiles * len});
    }
    n_rjobs += ptops::ceil_div_ll((long long)N * K, 1024) + ((want_db && want_db[i]) ? ptops::ceil_div(N, 1024) : 0);
  }
  // longest groups first, each to the XCD queue with the least work so far
  std::stable_sort(groups.begin(), groups.end(), [](const WgradGroup& a, const WgradGroup& b) { return a.cost > b.cost; });
  std::vector<WgradJob> queue[8];
  long long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (const WgradGroup& g : groups) {
    int x = 0;
    for (int j = 1; j < 8; ++j)
      if (load[j] < load[x]) x = j;
    load[x] += g.cost;
    for (int t = 0; t < g.tiles; ++t) queue[x].push_back({g.problem, t, g.split, 0});
  }
  size_t rounds = 0;
  for (int x = 0; x < 8; ++x) rounds = std::max(rounds, queue[x].size());
  const long long n_jobs = (long long)rounds * 8;
  const long long bytes = (long long)sizeof(WgradPlanHeader) + (long long)n * sizeof(WgradPlanProblem) +
                          n_jobs * (long long)sizeof(WgradJob) + n_rjobs * (long long)sizeof(WgradRJob);
  if (!plan || capacity < bytes) return bytes;
  if (n_jobs >= (1LL << 31) || n_rjobs >= (1LL << 31)) return -1;
  unsigned char* out = (unsigned char*)plan;
  WgradPlanHeader h;
  h.magic = kWgradMagic, h.n_problems = n, h.n_jobs = (int)n_jobs, h.n_rjobs = (int)n_rjobs;
  h.workspace_bytes = std::max<long long>(ws_floats, 4) * (long long)sizeof(float), h.reserved = target;
  memcpy(out, &h, sizeof h);
  out += sizeof h;
  memcpy(out, pp.data(), (size_t)n * sizeof(WgradPlanProblem));
  out += (size_t)n * sizeof(WgradPlanProblem);
  WgradJob* jobs = (WgradJob*)out;
  for (size_t r = 0; r < rounds; ++r)
    for (int x = 0; x < 8; ++x) jobs[r * 8 + x] = r < queue[x].size() ? queue[x][r] : WgradJob{-1, 0, 0, 0};
  out += (size_t)n_jobs * sizeof(WgradJob);
  WgradRJob* rj = (WgradRJob*)out;
  for (int i = 0; i < n; ++i) {
    const int N = mnk[3 * i + 1], K = mnk[3 * i + 2];
    const int chunks = (int)ptops::ceil_div_ll((long long)N * K, 1024);
    for (int c = 0; c < chunks; ++c) *rj++ = {i, 0, c, 0};
    if (want_db && want_db[i])
      for (int c = 0; c < ptops::ceil_div(N, 1024); ++c) *rj++ = {i, 1, c, 0};
  }
  return bytes;
}

PTOPS_API long long ptops_wgrad_group_workspace_bytes(const void* plan_host) {
  if (!plan_host) return 0;
  WgradPlanHeader h;
  memcpy(&h, plan_host, sizeof h);
  return h.magic == kWgradMagic ? h.workspace_bytes : -1;
}

PTOPS_API int ptops_wgrad_problem_bytes(void) { return (int)sizeof(WgradProblem); }

// Runs the plan.  problems_host / problems_dev: the same n rows of PtopsWgradProblem (the host copy is only validated
// against the plan); plan_host / plan_dev: the bytes ptops_wgrad_group_plan wrote; workspace: device scratch of
// ptops_wgrad_group_workspace_bytes(plan_host), 16-byte aligned.  Two launches, whatever n is.
PTOPS_API int ptops_gemm_f32_split3_wgrad_grouped(const void* problems_host, const void* problems_dev, int n,
                                                  const void* plan_host, const void* plan_dev, void* workspace,
                                                  long long workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n == 0) return PTOPS_OK;
  PTOPS_REQUIRE(problems_host && problems_dev && plan_host && plan_dev && workspace, "wgrad_grouped: null pointer");
  WgradPlanHeader h;
  memcpy(&h, plan_host, sizeof h);
  PTOPS_REQUIRE(h.magic == kWgradMagic && h.n_problems == n, "wgrad_grouped: the plan is not for these %d problems", n);
  PTOPS_REQUIRE(workspace_bytes >= h.workspace_bytes && ((uintptr_t)workspace & 15) == 0, "wgrad_grouped: workspace too small or misaligned");
  const WgradPlanProblem* qh = (const WgradPlanProblem*)((const unsigned char*)plan_host + sizeof h);
  const WgradProblem* ph = (const WgradProblem*)problems_host;
  for (int i = 0; i < n; ++i) {
    const WgradProblem& p = ph[i];
    PTOPS_REQUIRE(p.dY && p.X && p.dW, "wgrad_grouped: null operand in problem %d", i);
    PTOPS_REQUIRE(p.M >= kBK && p.M % kBK == 0 && p.K % 4 == 0 && p.lddw % 4 == 0 && ((uintptr_t)p.dW & 15) == 0,
                  "wgrad_grouped: problem %d: M %% 16, K %% 4, 16-byte aligned dW rows required", i);
    PTOPS_REQUIRE(!p.db || (p.N % 4 == 0 && ((uintptr_t)p.db & 15) == 0), "wgrad_grouped: problem %d: db needs N %% 4 == 0 and 16-byte alignment", i);
    int splits, per;
    wgrad_group_split(p.M, (int)h.reserved, &splits, &per);
    PTOPS_REQUIRE(splits == qh[i].splits && per == qh[i].steps_per_split && ptops::ceil_div(p.K, kBN) == qh[i].tiles_k,
                  "wgrad_grouped: problem %d does not match the plan", i);
    const long long room = (i + 1 < n ? qh[i + 1].part_base : h.workspace_bytes / (long long)sizeof(float)) - qh[i].part_base;
    PTOPS_REQUIRE(room >= (long long)splits * p.N * p.K + (p.db ? (long long)splits * p.N : 0), "wgrad_grouped: problem %d does not match the plan", i);
  }
  const unsigned char* pd = (const unsigned char*)plan_dev + sizeof h;
  const WgradPlanProblem* qd = (const WgradPlanProblem*)pd;
  const WgradJob* jobs = (const WgradJob*)(pd + (size_t)n * sizeof(WgradPlanProblem));
  const WgradRJob* rjobs = (const WgradRJob*)((const unsigned char*)jobs + (size_t)h.n_jobs * sizeof(WgradJob));
  if (ptops::option("PTOPS_WGRAD_OCC3"))      // A/B: three workgroups per CU = six waves per SIMD (<= 80 VGPRs; LDS 3 x 51.7 KB fits)
    hipLaunchKernelGGL(gemm_split3_wgrad_grouped_kernel<6>, dim3((unsigned)h.n_jobs), dim3(512), 0, stream,
                       (const WgradProblem*)problems_dev, qd, jobs, (float*)workspace);
  else
    hipLaunchKernelGGL(gemm_split3_wgrad_grouped_kernel<2>, dim3((unsigned)h.n_jobs), dim3(512), 0, stream,
                       (const WgradProblem*)problems_dev, qd, jobs, (float*)workspace);
  int rc = ptops::check_launch("gemm_split3_wgrad_grouped_kernel");
  if (rc != PTOPS_OK) return rc;
  hipLaunchKernelGGL(wgrad_grouped_reduce_kernel, dim3((unsigned)h.n_rjobs), dim3(256), 0, stream,
                     (const WgradProblem*)problems_dev, qd, rjobs, (const float*)workspace);
  return ptops::check_launch("wgrad_grouped_reduce_kernel");
}
