// Training-mode BatchNorm over the rows of a token-major matrix, fused with LeakyReLU, for gfx950 (MI355X).
//
// AdaPoinTr.increase_dim (models/AdaPoinTr.py:222-224 of the reference: Conv1d 1x1 -> BatchNorm1d(1024) ->
// LeakyReLU(0.2) -> Conv1d 1x1 on [B, C, M]) runs here token-major on [B*M, C] rows, where the batch statistics are
// column moments.  PyTorch-ROCm's channels-last batch-norm kernels take 0.33 ms to collect the statistics of one
// [18432, 1024] matrix (230 GB/s); these are plain streams laid out like the column sums of reduce.hip:
//   forward : column partials of {sum x, sum x^2} per 128-row slab -> fold -> y = lrelu(gamma (x - mean) rstd + beta)
//   backward: dz = dy lrelu'(z); column partials of {sum dz, sum dz xhat} -> fold ->
//             dx = gamma rstd (dz - mean_rows(dz) - xhat mean_rows(dz xhat)),  dgamma = sum dz xhat,  dbeta = sum dz
// Moments are combined in double from the folded fp32 sums (fixed order: deterministic).
#include <math.h>

#include "ptops_common.h"

namespace {

constexpr int kRows = 128;  // rows per slab

// partial[slab][0..N) = sum_rows a, partial[slab][N..2N) = sum_rows b  with (a, b) produced per element by F
template <typename F>
__device__ __forceinline__ void column_pair_partial(int M, int N, float* __restrict__ partial, F f) {
  __shared__ float4 red[2][16][16];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int col = (blockIdx.x * 16 + cx) * 4;
  const bool in = col < N;
  const int r0 = blockIdx.y * kRows, r1 = min(M, r0 + kRows);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (in) {
    for (int r = r0 + ry; r < r1; r += 16) {
      float4 u, v;
      f((size_t)r * N + col, col, u, v);
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
      b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
    }
  }
  red[0][ry][cx] = a;
  red[1][ry][cx] = b;
  __syncthreads();
  if (ry < 2 && in) {
    float4 s = red[ry][0][cx];
#pragma unroll
    for (int t = 1; t < 16; ++t) {
      const float4 v = red[ry][t][cx];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(partial + (size_t)blockIdx.y * 2 * N + (size_t)ry * N + col) = s;
  }
}

__global__ __launch_bounds__(256) void bn_moments_kernel(const float* __restrict__ x, int M, int N,
                                                         float* __restrict__ partial) {
  column_pair_partial(M, N, partial, [&](size_t o, int, float4& u, float4& v) {
    u = *reinterpret_cast<const float4*>(x + o);
    v = make_float4(u.x * u.x, u.y * u.y, u.z * u.z, u.w * u.w);
  });
}

// sums [2N] = {sum x, sum x^2}  ->  stats [2N] = {mean, rstd}, var_out [N] = biased variance
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ sums, int M, int N, float eps,
                                                          float* __restrict__ stats, float* __restrict__ var_out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  const double mean = (double)sums[c] / M;
  double var = (double)sums[N + c] / M - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[c] = (float)mean;
  stats[N + c] = (float)(1.0 / sqrt(var + (double)eps));
  var_out[c] = (float)var;
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       long long total4, int N, float slope, float* __restrict__ y) {
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
    const int col = (int)((e * 4) % N);
    const float4 v = *reinterpret_cast<const float4*>(x + e * 4);
    const float4 mu = *reinterpret_cast<const float4*>(stats + col), rs = *reinterpret_cast<const float4*>(stats + N + col);
    const float4 g = *reinterpret_cast<const float4*>(gamma + col), b = *reinterpret_cast<const float4*>(beta + col);
    float4 z;
    z.x = fmaf((v.x - mu.x) * rs.x, g.x, b.x);
    z.y = fmaf((v.y - mu.y) * rs.y, g.y, b.y);
    z.z = fmaf((v.z - mu.z) * rs.z, g.z, b.z);
    z.w = fmaf((v.w - mu.w) * rs.w, g.w, b.w);
    z.x = z.x > 0.f ? z.x : z.x * slope;
    z.y = z.y > 0.f ? z.y : z.y * slope;
    z.z = z.z > 0.f ? z.z : z.z * slope;
    z.w = z.w > 0.f ? z.w : z.w * slope;
    *reinterpret_cast<float4*>(y + e * 4) = z;
  }
}

__device__ __forceinline__ void bn_dz_xhat(float dy, float x, float mu, float rs, float g, float b, float slope, float& dz,
                                           float& xh) {
  xh = (x - mu) * rs;
  const float z = fmaf(xh, g, b);
  dz = z > 0.f ? dy : dy * slope;
}

__global__ __launch_bounds__(256) void bn_bwd_moments_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int M, int N, float slope,
                                                             float* __restrict__ partial) {
  column_pair_partial(M, N, partial, [&](size_t o, int col, float4& u, float4& v) {
    const float4 d = *reinterpret_cast<const float4*>(dy + o), xv = *reinterpret_cast<const float4*>(x + o);
    const float4 mu = *reinterpret_cast<const float4*>(stats + col), rs = *reinterpret_cast<const float4*>(stats + N + col);
    const float4 g = *reinterpret_cast<const float4*>(gamma + col), b = *reinterpret_cast<const float4*>(beta + col);
    float4 xh;
    bn_dz_xhat(d.x, xv.x, mu.x, rs.x, g.x, b.x, slope, u.x, xh.x);
    bn_dz_xhat(d.y, xv.y, mu.y, rs.y, g.y, b.y, slope, u.y, xh.y);
    bn_dz_xhat(d.z, xv.z, mu.z, rs.z, g.z, b.z, slope, u.z, xh.z);
    bn_dz_xhat(d.w, xv.w, mu.w, rs.w, g.w, b.w, slope, u.w, xh.w);
    v = make_float4(u.x * xh.x, u.y * xh.y, u.z * xh.z, u.w * xh.w);
  });
}

// sums [2N] = {sum dz (= dbeta), sum dz xhat (= dgamma)}
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ sums, long long total4, int M, int N,
                                                           float slope, float* __restrict__ dx) {
  const float inv_m = 1.0f / (float)M;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
    const int col = (int)((e * 4) % N);
    const float4 d = *reinterpret_cast<const float4*>(dy + e * 4), xv = *reinterpret_cast<const float4*>(x + e * 4);
    const float4 mu = *reinterpret_cast<const float4*>(stats + col), rs = *reinterpret_cast<const float4*>(stats + N + col);
    const float4 g = *reinterpret_cast<const float4*>(gamma + col), b = *reinterpret_cast<const float4*>(beta + col);
    const float4 s1 = *reinterpret_cast<const float4*>(sums + col), s2 = *reinterpret_cast<const float4*>(sums + N + col);
    float4 dz, xh, o;
    bn_dz_xhat(d.x, xv.x, mu.x, rs.x, g.x, b.x, slope, dz.x, xh.x);
    bn_dz_xhat(d.y, xv.y, mu.y, rs.y, g.y, b.y, slope, dz.y, xh.y);
    bn_dz_xhat(d.z, xv.z, mu.z, rs.z, g.z, b.z, slope, dz.z, xh.z);
    bn_dz_xhat(d.w, xv.w, mu.w, rs.w, g.w, b.w, slope, dz.w, xh.w);
    o.x = g.x * rs.x * (dz.x - s1.x * inv_m - xh.x * s2.x * inv_m);
    o.y = g.y * rs.y * (dz.y - s1.y * inv_m - xh.y * s2.y * inv_m);
    o.z = g.z * rs.z * (dz.z - s1.z * inv_m - xh.z * s2.z * inv_m);
    o.w = g.w * rs.w * (dz.w - s1.w * inv_m - xh.w * s2.w * inv_m);
    *reinterpret_cast<float4*>(dx + e * 4) = o;
  }
}

int bn_check(const char* who, int M, int N, const void* a, const void* b, const void* c) {
  PTOPS_REQUIRE(M >= 1 && N >= 4 && N % 4 == 0, "%s: need M >= 1 and N %% 4 == 0 (M=%d N=%d)", who, M, N);
  PTOPS_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0, "%s: pointers must be 16-byte aligned", who);
  PTOPS_REQUIRE(ptops::ceil_div(M, kRows) <= 65535, "%s: M=%d exceeds the grid limit", who, M);
  return PTOPS_OK;
}

unsigned stream_grid(long long total4) {
  const long long b = (total4 + 255) / 256;
  return (unsigned)(b < 8192 ? b : 8192);
}

}  // namespace

PTOPS_API size_t ptops_batch_norm_rows_workspace_bytes(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  return sizeof(float) * ((size_t)ptops::ceil_div(M, kRows) * 2 * N + 2 * (size_t)N);
}

PTOPS_API int ptops_batch_norm_rows_forward(const float* x, const float* gamma, const float* beta, int M, int N, float eps,
                                            float slope, float* y, float* stats, float* var_out, void* workspace,
                                            void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(x && gamma && beta && y && stats && var_out && workspace, "batch_norm_rows_forward: null pointer");
  int rc = bn_check("batch_norm_rows_forward", M, N, x, y, workspace);
  if (rc != PTOPS_OK) return rc;
  const int S = ptops::ceil_div(M, kRows);
  float* partial = (float*)workspace;
  float* sums = partial + (size_t)S * 2 * N;
  hipLaunchKernelGGL(bn_moments_kernel, dim3(ptops::ceil_div(N, 64), S), dim3(256), 0, stream, x, M, N, partial);
  rc = ptops_reduce::fold_rows(partial, S, 2 * N, sums, stream);
  if (rc != PTOPS_OK) return rc;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ptops::ceil_div(N, 256)), dim3(256), 0, stream, sums, M, N, eps, stats, var_out);
  const long long total4 = (long long)M * N / 4;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(stream_grid(total4)), dim3(256), 0, stream, x, stats, gamma, beta, total4, N,
                     slope, y);
  return ptops::check_launch("batch_norm_rows_forward");
}

PTOPS_API int ptops_batch_norm_rows_backward(const float* dy, const float* x, const float* stats, const float* gamma,
                                             const float* beta, int M, int N, float slope, float* dx,
                                             float* dbeta_dgamma, void* workspace, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(dy && x && stats && gamma && beta && dx && dbeta_dgamma && workspace, "batch_norm_rows_backward: null pointer");
  int rc = bn_check("batch_norm_rows_backward", M, N, x, dy, dx);
  if (rc != PTOPS_OK) return rc;
  PTOPS_REQUIRE((((uintptr_t)workspace | (uintptr_t)dbeta_dgamma) & 15) == 0, "batch_norm_rows_backward: pointers must be 16-byte aligned");
  const int S = ptops::ceil_div(M, kRows);
  float* partial = (float*)workspace;
  hipLaunchKernelGGL(bn_bwd_moments_kernel, dim3(ptops::ceil_div(N, 64), S), dim3(256), 0, stream, dy, x, stats, gamma, beta,
                     M, N, slope, partial);
  rc = ptops_reduce::fold_rows(partial, S, 2 * N, dbeta_dgamma, stream);
  if (rc != PTOPS_OK) return rc;
  const long long total4 = (long long)M * N / 4;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(total4)), dim3(256), 0, stream, dy, x, stats, gamma, beta,
                     dbeta_dgamma, total4, M, N, slope, dx);
  return ptops::check_launch("batch_norm_rows_backward");
}
