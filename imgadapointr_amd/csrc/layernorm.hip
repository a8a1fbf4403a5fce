// LayerNorm over the last dimension for gfx950 (MI355X): forward and a single-pass backward.
//
// The transformer blocks of the path (models/base_blocks.py:88-118, 154-207 of the reference: norm1 /
// norm2 / norm_q / norm_v of every encoder and decoder block, embed dim 384) call nn.LayerNorm 44 times per
// step on [B*N, 384] rows.  The op is a pure stream (read x, write y; backward: read x and dy, write dx), so
// it is laid out for HBM: one wave owns a row at a time, a lane holds its float4 slices of the row in
// registers (C <= 1024), both moments are wave reductions (two-pass, so the variance is the centred one), and
// the backward produces dx and the per-row terms of dgamma / dbeta in ONE pass over x and dy:
//   xh = (x - mean) rstd,  g = dy gamma,  dx = rstd (g - mean_c(g) - xh mean_c(g xh)),
//   dgamma = sum_rows dy xh,  dbeta = sum_rows dy
// Each workgroup keeps its dgamma / dbeta partial in registers over all its rows, writes one partial row, and
// the column-sum kernels of reduce.hip fold the partials in a fixed order (deterministic, no atomics).
// PyTorch-ROCm's backward takes 68 us per [18432, 384] call (three kernels, 4x the stream time).
#include <math.h>

#include "ptops_common.h"

namespace {

constexpr int kWaves = 4;

// Sum over the wave (all lanes receive it).  `__shfl_xor` compiles to ds_bpermute_b32, an LDS round trip per step: six
// dependent ones per reduction and two or three dependent reductions per row made the row latency, not HBM, the bound
// (3.6 TB/s forward).  Here: an inclusive DPP scan written as ISA (row_shr 1, 2, 4, 8 inside each row of 16, row_bcast:15
// into rows 1 and 3, row_bcast:31 into rows 2 and 3; one v_add_f32_dpp per step; a lane whose source does not exist keeps
// its value), lane 63 holds the total.  The assembler adds no wait states inside an asm block: a DPP read of a VGPR
// written by the previous VALU instruction needs two (s_nop 1).  A fixed order, so results stay run-to-run identical.
__device__ __forceinline__ float wave_sum(float v) {
#define PTOPS_DPP_STEP(ctrl) "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 " ctrl " bank_mask:0xf\n\t"
  asm volatile(PTOPS_DPP_STEP("row_shr:1 row_mask:0xf") PTOPS_DPP_STEP("row_shr:2 row_mask:0xf")
               PTOPS_DPP_STEP("row_shr:4 row_mask:0xf") PTOPS_DPP_STEP("row_shr:8 row_mask:0xf")
               PTOPS_DPP_STEP("row_bcast:15 row_mask:0xa") PTOPS_DPP_STEP("row_bcast:31 row_mask:0xc") "s_nop 1"
               : "+v"(v));
#undef PTOPS_DPP_STEP
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// NV = number of 256-column slices a lane may hold (C <= 256 NV)
// ADD: the row is x + res and the sum is also written out (residual add fused into the following norm)
template <int NV, bool ADD>
__global__ __launch_bounds__(kWaves * 64) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                             float* __restrict__ sum_out,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, long long rows, int C,
                                                             float eps, int rows_per_wave, float* __restrict__ y,
                                                             float* __restrict__ mean, float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long r0 = ((long long)blockIdx.x * kWaves + wave) * rows_per_wave;
  const long long r1 = min(rows, r0 + rows_per_wave);
  float4 g[NV], b[NV];
  bool in[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int col = 4 * lane + 256 * i;
    in[i] = col < C;
    g[i] = in[i] ? ld4(gamma + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    b[i] = in[i] ? ld4(beta + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float inv_c = 1.0f / (float)C;
  // the loads of row r+1 are issued before the reductions of row r (two dependent wave reductions per row would
  // otherwise leave the memory pipe idle)
  float4 nx[NV], nr[NV];
  auto fetch = [&](long long r) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      nx[i] = (in[i] && r < r1) ? ld4(x + r * C + 4 * lane + 256 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (ADD) nr[i] = (in[i] && r < r1) ? ld4(res + r * C + 4 * lane + 256 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  fetch(r0);
  for (long long r = r0; r < r1; ++r) {
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i] = nx[i];
      if constexpr (ADD) {
        v[i].x += nr[i].x; v[i].y += nr[i].y; v[i].z += nr[i].z; v[i].w += nr[i].w;
        if (in[i]) st4(sum_out + r * C + 4 * lane + 256 * i, v[i]);
      }
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    fetch(r + 1);
    const float mu = wave_sum(s) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i].x -= mu; v[i].y -= mu; v[i].z -= mu; v[i].w -= mu;
      if (in[i]) q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rs = 1.0f / sqrtf(wave_sum(q) * inv_c + eps);
    float* yr = y + r * C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (in[i]) {
        float4 o;
        o.x = fmaf(v[i].x * rs, g[i].x, b[i].x);
        o.y = fmaf(v[i].y * rs, g[i].y, b[i].y);
        o.z = fmaf(v[i].z * rs, g[i].z, b[i].z);
        o.w = fmaf(v[i].w * rs, g[i].w, b[i].w);
        st4(yr + 4 * lane + 256 * i, o);
      }
    }
    if (lane == 0) {
      mean[r] = mu;
      rstd[r] = rs;
    }
  }
}

// ADD: dx += extra (the gradient that reaches the normalised tensor through its other consumers)
template <int NV, bool ADD>
__global__ __launch_bounds__(kWaves * 64) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ extra,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, long long rows, int C,
                                                             int rows_per_wave, float* __restrict__ dx,
                                                             float* __restrict__ partial /* [grid, 3, C] */) {
  __shared__ float4 red[kWaves][3 * NV][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long r0 = ((long long)blockIdx.x * kWaves + wave) * rows_per_wave;
  const long long r1 = min(rows, r0 + rows_per_wave);
  float4 g[NV], ag[NV], ab[NV], ax[NV];   // partial dgamma, dbeta and column sums of the dx this kernel writes
  bool in[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int col = 4 * lane + 256 * i;
    in[i] = col < C;
    g[i] = in[i] ? ld4(gamma + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ax[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float inv_c = 1.0f / (float)C;
  float4 nxv[NV], ndv[NV], nev[NV];
  float nmu = 0.f, nrs = 0.f;
  auto fetch = [&](long long r) {   // row r+1 is requested before the reductions of row r
    const bool ok = r < r1;
    nmu = ok ? mean[r] : 0.f;
    nrs = ok ? rstd[r] : 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      nxv[i] = (in[i] && ok) ? ld4(x + r * C + 4 * lane + 256 * i) : make_float4(nmu, nmu, nmu, nmu);
      ndv[i] = (in[i] && ok) ? ld4(dy + r * C + 4 * lane + 256 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (ADD) nev[i] = (in[i] && ok) ? ld4(extra + r * C + 4 * lane + 256 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  fetch(r0);
  for (long long r = r0; r < r1; ++r) {
    const float mu = nmu, rs = nrs;
    float4 xh[NV], gy[NV], ev[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float4 xv = nxv[i];
      const float4 dv = ndv[i];
      if constexpr (ADD) ev[i] = nev[i];
      xh[i].x = (xv.x - mu) * rs; xh[i].y = (xv.y - mu) * rs; xh[i].z = (xv.z - mu) * rs; xh[i].w = (xv.w - mu) * rs;
      gy[i].x = dv.x * g[i].x; gy[i].y = dv.y * g[i].y; gy[i].z = dv.z * g[i].z; gy[i].w = dv.w * g[i].w;
      s1 += (gy[i].x + gy[i].y) + (gy[i].z + gy[i].w);
      s2 += (gy[i].x * xh[i].x + gy[i].y * xh[i].y) + (gy[i].z * xh[i].z + gy[i].w * xh[i].w);
      ag[i].x = fmaf(dv.x, xh[i].x, ag[i].x); ag[i].y = fmaf(dv.y, xh[i].y, ag[i].y);
      ag[i].z = fmaf(dv.z, xh[i].z, ag[i].z); ag[i].w = fmaf(dv.w, xh[i].w, ag[i].w);
      ab[i].x += dv.x; ab[i].y += dv.y; ab[i].z += dv.z; ab[i].w += dv.w;
    }
    fetch(r + 1);
    const float m1 = wave_sum(s1) * inv_c, m2 = wave_sum(s2) * inv_c;
    float* dxr = dx + r * C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (in[i]) {
        float4 o;
        o.x = rs * (gy[i].x - m1 - xh[i].x * m2);
        o.y = rs * (gy[i].y - m1 - xh[i].y * m2);
        o.z = rs * (gy[i].z - m1 - xh[i].z * m2);
        o.w = rs * (gy[i].w - m1 - xh[i].w * m2);
        if constexpr (ADD) {
          o.x += ev[i].x; o.y += ev[i].y; o.z += ev[i].z; o.w += ev[i].w;
        }
        st4(dxr + 4 * lane + 256 * i, o);
        ax[i].x += o.x; ax[i].y += o.y; ax[i].z += o.z; ax[i].w += o.w;
      }
    }
  }
  // fold the four waves' dgamma / dbeta / column-sum partials in a fixed order, one partial row per workgroup
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    red[wave][3 * i][lane] = ag[i];
    red[wave][3 * i + 1][lane] = ab[i];
    red[wave][3 * i + 2][lane] = ax[i];
  }
  __syncthreads();
  if (wave == 0) {
    float* pg = partial + (size_t)blockIdx.x * 3 * C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (in[i]) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          float4 sg = red[0][3 * i + q][lane];
#pragma unroll
          for (int w = 1; w < kWaves; ++w) {
            const float4 a = red[w][3 * i + q][lane];
            sg.x += a.x; sg.y += a.y; sg.z += a.z; sg.w += a.w;
          }
          st4(pg + q * C + 4 * lane + 256 * i, sg);
        }
      }
    }
  }
}

// rows handled by one wave: enough workgroups to fill the chip, few enough that the partial fold stays small
int pick_rows_per_wave(long long rows, bool backward) {
  const long long target_blocks = backward ? 1024 : 2048;
  long long rpw = (rows + target_blocks * kWaves - 1) / (target_blocks * kWaves);
  if (rpw < 1) rpw = 1;
  return (int)rpw;
}

bool ln_args_ok(long long rows, int C) { return rows >= 0 && C >= 4 && C % 4 == 0 && C <= 1024; }

}  // namespace

PTOPS_API int ptops_layer_norm_max_dim(void) { return 1024; }

static int ln_forward_impl(const float* x, const float* res, float* sum_out, const float* gamma, const float* beta,
                           long long rows, int C, float eps, float* y, float* mean, float* rstd, hipStream_t stream) {
  PTOPS_REQUIRE(ln_args_ok(rows, C), "layer_norm_forward: need rows >= 0, C %% 4 == 0, 4 <= C <= 1024 (C=%d)", C);
  if (rows == 0) return PTOPS_OK;
  PTOPS_REQUIRE(x && gamma && beta && y && mean && rstd, "layer_norm_forward: null pointer");
  PTOPS_REQUIRE((((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y | (uintptr_t)res | (uintptr_t)sum_out) & 15) == 0,
                "layer_norm_forward: pointers must be 16-byte aligned");
  const int rpw = pick_rows_per_wave(rows, false);
  const long long blocks = (rows + (long long)rpw * kWaves - 1) / ((long long)rpw * kWaves);
  PTOPS_REQUIRE(blocks <= 0x7fffffffLL, "layer_norm_forward: too many rows");
  const dim3 grid((unsigned)blocks), block(kWaves * 64);
#define PTOPS_LN_FWD(NV)                                                                                              \
  if (res) hipLaunchKernelGGL((ln_fwd_kernel<NV, true>), grid, block, 0, stream, x, res, sum_out, gamma, beta, rows, C, eps, rpw, y, mean, rstd); \
  else hipLaunchKernelGGL((ln_fwd_kernel<NV, false>), grid, block, 0, stream, x, res, sum_out, gamma, beta, rows, C, eps, rpw, y, mean, rstd)
  switch ((C + 255) / 256) {
    case 1: PTOPS_LN_FWD(1); break;
    case 2: PTOPS_LN_FWD(2); break;
    case 3: PTOPS_LN_FWD(3); break;
    default: PTOPS_LN_FWD(4); break;
  }
#undef PTOPS_LN_FWD
  return ptops::check_launch("ln_fwd_kernel");
}

PTOPS_API int ptops_layer_norm_forward(const float* x, const float* gamma, const float* beta, long long rows, int C,
                                       float eps, float* y, float* mean, float* rstd, void* stream_) {
  return ln_forward_impl(x, nullptr, nullptr, gamma, beta, rows, C, eps, y, mean, rstd, (hipStream_t)stream_);
}

PTOPS_API int ptops_add_layer_norm_forward(const float* x, const float* res, const float* gamma, const float* beta,
                                           long long rows, int C, float eps, float* sum_out, float* y, float* mean,
                                           float* rstd, void* stream_) {
  PTOPS_REQUIRE(rows == 0 || (res && sum_out), "add_layer_norm_forward: null pointer");
  return ln_forward_impl(x, res, sum_out, gamma, beta, rows, C, eps, y, mean, rstd, (hipStream_t)stream_);
}

static long long ln_bwd_blocks(long long rows) {
  const int rpw = pick_rows_per_wave(rows, true);
  return (rows + (long long)rpw * kWaves - 1) / ((long long)rpw * kWaves);
}

PTOPS_API size_t ptops_layer_norm_backward_workspace_bytes(long long rows, int C) {
  if (!ln_args_ok(rows, C) || rows == 0) return 0;
  const long long blocks = ln_bwd_blocks(rows);
  return sizeof(float) * (size_t)blocks * 3 * C;
}

PTOPS_API int ptops_layer_norm_backward(const float* dy, const float* x, const float* mean, const float* rstd,
                                        const float* gamma, const float* extra, long long rows, int C, float* dx,
                                        float* dgamma_dbeta, void* workspace, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(ln_args_ok(rows, C), "layer_norm_backward: need rows >= 0, C %% 4 == 0, 4 <= C <= 1024 (C=%d)", C);
  PTOPS_REQUIRE(dgamma_dbeta != nullptr, "layer_norm_backward: null output");
  if (rows == 0) {
    (void)ptops::fill_async(dgamma_dbeta, 0, sizeof(float) * 3 * (size_t)C, stream);
    return ptops::check_launch("layer_norm_backward(memset)");
  }
  PTOPS_REQUIRE(dy && x && mean && rstd && gamma && dx && workspace, "layer_norm_backward: null pointer");
  PTOPS_REQUIRE((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)gamma | (uintptr_t)dx | (uintptr_t)workspace |
                  (uintptr_t)dgamma_dbeta | (uintptr_t)extra) & 15) == 0,
                "layer_norm_backward: pointers must be 16-byte aligned");
  const int rpw = pick_rows_per_wave(rows, true);
  const long long blocks = ln_bwd_blocks(rows);
  PTOPS_REQUIRE(blocks <= 0x7fffffffLL, "layer_norm_backward: too many rows");
  float* partial = (float*)workspace;
  const dim3 grid((unsigned)blocks), block(kWaves * 64);
#define PTOPS_LN_BWD(NV)                                                                                                \
  if (extra) hipLaunchKernelGGL((ln_bwd_kernel<NV, true>), grid, block, 0, stream, dy, x, extra, mean, rstd, gamma, rows, C, rpw, dx, partial); \
  else hipLaunchKernelGGL((ln_bwd_kernel<NV, false>), grid, block, 0, stream, dy, x, extra, mean, rstd, gamma, rows, C, rpw, dx, partial)
  switch ((C + 255) / 256) {
    case 1: PTOPS_LN_BWD(1); break;
    case 2: PTOPS_LN_BWD(2); break;
    case 3: PTOPS_LN_BWD(3); break;
    default: PTOPS_LN_BWD(4); break;
  }
#undef PTOPS_LN_BWD
  int rc = ptops::check_launch("ln_bwd_kernel");
  if (rc != PTOPS_OK) return rc;
  // dgamma | dbeta | colsum(dx)  [3C] = column sums of the [blocks <= 1024, 3C] partial rows: one fold launch
  return ptops_reduce::fold_rows(partial, (int)blocks, 3 * C, dgamma_dbeta, stream, true);
}
