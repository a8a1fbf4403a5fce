// GELU (exact, erf form) backward fused with the column sums of its result, for gfx950 (MI355X).
//
// In every MLP of the path (models/Transformer_utils.py: Mlp = fc1 -> GELU -> fc2) the tensor dpre = dy * gelu'(x)
// is at once the input gradient of the activation and the output gradient of fc1, whose bias gradient is its column
// sum.  One pass reads x and dy, writes dpre and leaves per-slab column partials that a fixed-order fold turns into
// the bias gradient — instead of aten::gelu_backward followed by a separate column-sum pass over dpre.
#include <math.h>

#include "ptops_common.h"

namespace {

constexpr int kRows = 128;

__device__ __forceinline__ float gelu_grad(float x) {
  // d/dx [ x * Phi(x) ] = Phi(x) + x * phi(x)   (same expression as ATen's GeluBackwardKernel, "none" approximation)
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = expf(-0.5f * x * x) * 0.39894228040143267794f;
  return cdf + x * pdf;
}

__global__ __launch_bounds__(256) void gelu_bwd_colsum_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              int M, int N, float* __restrict__ dpre,
                                                              float* __restrict__ partial) {
  __shared__ float4 red[16][16];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int col = (blockIdx.x * 16 + cx) * 4;
  const bool in = col < N;
  const int r0 = blockIdx.y * kRows, r1 = min(M, r0 + kRows);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (in) {
    for (int r = r0 + ry; r < r1; r += 16) {
      const size_t o = (size_t)r * N + col;
      const float4 xv = *reinterpret_cast<const float4*>(x + o), dv = *reinterpret_cast<const float4*>(dy + o);
      float4 g;
      g.x = dv.x * gelu_grad(xv.x);
      g.y = dv.y * gelu_grad(xv.y);
      g.z = dv.z * gelu_grad(xv.z);
      g.w = dv.w * gelu_grad(xv.w);
      *reinterpret_cast<float4*>(dpre + o) = g;
      acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
    }
  }
  red[ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && in) {
    float4 s = red[0][cx];
#pragma unroll
    for (int t = 1; t < 16; ++t) {
      const float4 v = red[t][cx];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(partial + (size_t)blockIdx.y * N + col) = s;
  }
}

}  // namespace

PTOPS_API size_t ptops_gelu_backward_colsum_workspace_bytes(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  return sizeof(float) * (size_t)ptops::ceil_div(M, kRows) * N;
}

PTOPS_API int ptops_gelu_backward_colsum(const float* x, const float* dy, int M, int N, float* dpre, float* colsum,
                                         void* workspace, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(M >= 1 && N >= 4 && N % 4 == 0, "gelu_backward_colsum: need M >= 1 and N %% 4 == 0 (M=%d N=%d)", M, N);
  PTOPS_REQUIRE(x && dy && dpre && colsum && workspace, "gelu_backward_colsum: null pointer");
  PTOPS_REQUIRE((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dpre | (uintptr_t)colsum | (uintptr_t)workspace) & 15) == 0,
                "gelu_backward_colsum: pointers must be 16-byte aligned");
  const int S = ptops::ceil_div(M, kRows);
  PTOPS_REQUIRE(S <= 65535, "gelu_backward_colsum: M=%d exceeds the grid limit", M);
  hipLaunchKernelGGL(gelu_bwd_colsum_kernel, dim3(ptops::ceil_div(N, 64), S), dim3(256), 0, stream, x, dy, M, N, dpre,
                     (float*)workspace);
  int rc = ptops::check_launch("gelu_bwd_colsum_kernel");
  if (rc != PTOPS_OK) return rc;
  return ptops_reduce::fold_rows((const float*)workspace, S, N, colsum, stream, true);
}
