// Fused DGCNN edge-conv stage for gfx950 (MI355X): gather + GroupNorm statistics + neighbour max/min.
//
// One stage of DGCNN_Grouper (models/base_blocks.py:565-623 of the reference) is
//   max_j LeakyReLU(GroupNorm(Conv1x1(cat(x_k[idx_j] - x_q, x_q)))).
// The 1x1 convolution is linear, so its value on edge (q, j) is  y = P[idx[q,j]] + R[q]  with
// P = W1.x_k (one small GEMM per source cloud) and R = (W2 - W1).x_q; LeakyReLU and the GroupNorm affine
// are monotone per channel, so the max over the k neighbours is taken on y (max if the channel's scale
// gamma*rstd is >= 0, else min) and the [B, C', Nq, k] tensors of the reference (134 MB per stage at
// B = 32, written and re-read by conv, norm, activation, max and all their backwards) never exist.
//
//  forward   edge_select: per (b, c', q) gathers the k values of P, keeps the selected value and the
//            source index that produced it, their sum (needed by the R gradient) and per-workgroup partial
//            sums of y and y^2 for the GroupNorm statistics (reduced deterministically by the caller).
//  backward  edge_scatter: d/dP of everything that reaches P through the edges,
//            T[c', n] = sum_{edges e -> n} dense[c', q_e] + sum_{selected edges e -> n} sel[c', q_e],
//            accumulated in an LDS tile (ds_add_f32) per (batch, channel tile); the remaining terms of the
//            GroupNorm backward are element-wise in [B, C', N] and stay on the host side.
#include <math.h>

#include "ptops_common.h"

namespace {

constexpr int kTPB = 256;
constexpr int kCT = 4;      // channels per workgroup in edge_select (index registers are reused)
constexpr int kMaxK = 32;

__global__ __launch_bounds__(kTPB) void edge_select_kernel(
    const float* __restrict__ P, const float* __restrict__ R, const int32_t* __restrict__ idx,
    const unsigned char* __restrict__ take_max, int C, int Nk, int Nq, int k, float* __restrict__ ysel,
    int32_t* __restrict__ nsel, float* __restrict__ ysum, float* __restrict__ partial) {
  __shared__ float red[2][kTPB / 64];
  const int b = blockIdx.z, c0 = blockIdx.y * kCT;
  const int q = blockIdx.x * kTPB + threadIdx.x;
  const bool in = q < Nq;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int id[kMaxK];
  const int32_t* ip = idx + ((size_t)b * Nq + (in ? q : 0)) * k;
#pragma unroll
  for (int j = 0; j < kMaxK; ++j) id[j] = j < k ? ip[j] : 0;
  const int nblk = gridDim.x;
  for (int ct = 0; ct < kCT; ++ct) {
    const int c = c0 + ct;
    if (c >= C) break;  // uniform
    const float* prow = P + ((size_t)b * C + c) * Nk;
    const float r = in ? R[((size_t)b * C + c) * Nq + q] : 0.0f;
    const bool mx = take_max[c] != 0;
    float best = 0.0f, s1 = 0.0f, s2 = 0.0f;
    int bn = 0;
#pragma unroll
    for (int j = 0; j < kMaxK; ++j) {
      if (j < k) {
        const float y = prow[id[j]] + r;
        const bool better = j == 0 || (mx ? y > best : y < best);  // first extremum wins
        best = better ? y : best;
        bn = better ? id[j] : bn;
        s1 += y;
        s2 += y * y;
      }
    }
    if (in) {
      const size_t o = ((size_t)b * C + c) * Nq + q;
      ysel[o] = best;
      nsel[o] = bn;
      ysum[o] = s1;
    } else {
      s1 = 0.0f;
      s2 = 0.0f;
    }
    // workgroup partial of (sum y, sum y^2): wave shuffles, then 4 values through LDS
    for (int off = 32; off >= 1; off >>= 1) {
      s1 += __shfl_xor(s1, off);
      s2 += __shfl_xor(s2, off);
    }
    if (lane == 0) {
      red[0][wave] = s1;
      red[1][wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float a = 0.0f, d = 0.0f;
#pragma unroll
      for (int w = 0; w < kTPB / 64; ++w) {
        a += red[0][w];
        d += red[1][w];
      }
      float* pp = partial + (((size_t)b * C + c) * nblk + blockIdx.x) * 2;
      pp[0] = a;
      pp[1] = d;
    }
    __syncthreads();
  }
}

template <int CT>
__global__ __launch_bounds__(kTPB) void edge_scatter_kernel(
    const float* __restrict__ dense, const float* __restrict__ sel, const int32_t* __restrict__ idx,
    const int32_t* __restrict__ nsel, int C, int Nk, int Nq, int k, float* __restrict__ T) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [CT][Nk]
  const int b = blockIdx.y, c0 = blockIdx.x * CT;
  for (int t = threadIdx.x; t < CT * Nk; t += kTPB) tile[t] = 0.0f;
  __syncthreads();
  const int32_t* ib = idx + (size_t)b * Nq * k;
  for (int q = threadIdx.x; q < Nq; q += kTPB) {
    int id[kMaxK];
#pragma unroll
    for (int j = 0; j < kMaxK; ++j) id[j] = j < k ? ib[(size_t)q * k + j] : 0;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int c = c0 + ct;
      if (c >= C) break;
      const size_t o = ((size_t)b * C + c) * Nq + q;
      const float dv = dense[o];
#pragma unroll
      for (int j = 0; j < kMaxK; ++j)
        if (j < k) atomicAdd(&tile[ct * Nk + id[j]], dv);
      atomicAdd(&tile[ct * Nk + nsel[o]], sel[o]);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < CT * Nk; t += kTPB) {
    const int ct = t / Nk, n = t - ct * Nk;
    if (c0 + ct < C) T[((size_t)b * C + c0 + ct) * Nk + n] = tile[t];
  }
}

}  // namespace

PTOPS_API int ptops_edge_select_blocks(int Nq) { return ptops::ceil_div(Nq, kTPB); }

PTOPS_API int ptops_edge_select(const float* P, const float* R, const int32_t* idx, const unsigned char* take_max,
                                int B, int C, int Nk, int Nq, int k, float* ysel, int32_t* nsel, float* ysum,
                                float* partial, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && C >= 1 && Nk >= 1 && Nq >= 1 && k >= 1 && k <= kMaxK, "edge_select: bad size (k <= %d)", kMaxK);
  if (B == 0) return PTOPS_OK;
  PTOPS_REQUIRE(P && R && idx && take_max && ysel && nsel && ysum && partial, "edge_select: null pointer");
  PTOPS_REQUIRE(B <= 65535, "edge_select: B exceeds grid limit");
  dim3 grid(ptops::ceil_div(Nq, kTPB), ptops::ceil_div(C, kCT), B);
  hipLaunchKernelGGL(edge_select_kernel, grid, dim3(kTPB), 0, stream, P, R, idx, take_max, C, Nk, Nq, k, ysel, nsel,
                     ysum, partial);
  return ptops::check_launch("edge_select_kernel");
}

PTOPS_API int ptops_edge_scatter(const float* dense, const float* sel, const int32_t* idx, const int32_t* nsel,
                                 int B, int C, int Nk, int Nq, int k, float* T, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && C >= 1 && Nk >= 1 && Nq >= 1 && k >= 1 && k <= kMaxK, "edge_scatter: bad size (k <= %d)", kMaxK);
  if (B == 0) return PTOPS_OK;
  PTOPS_REQUIRE(dense && sel && idx && nsel && T, "edge_scatter: null pointer");
  PTOPS_REQUIRE((size_t)Nk * sizeof(float) <= 64 * 1024, "edge_scatter: Nk=%d does not fit an LDS tile", Nk);
  int CT = 8;
  while (CT > 1 && ((long long)B * ptops::ceil_div(C, CT) < 512 || (size_t)CT * Nk * 4 > 64 * 1024)) CT >>= 1;
  dim3 grid(ptops::ceil_div(C, CT), B);
  const size_t lds = (size_t)CT * Nk * sizeof(float);
  switch (CT) {
    case 8: hipLaunchKernelGGL(edge_scatter_kernel<8>, grid, dim3(kTPB), lds, stream, dense, sel, idx, nsel, C, Nk, Nq, k, T); break;
    case 4: hipLaunchKernelGGL(edge_scatter_kernel<4>, grid, dim3(kTPB), lds, stream, dense, sel, idx, nsel, C, Nk, Nq, k, T); break;
    case 2: hipLaunchKernelGGL(edge_scatter_kernel<2>, grid, dim3(kTPB), lds, stream, dense, sel, idx, nsel, C, Nk, Nq, k, T); break;
    default: hipLaunchKernelGGL(edge_scatter_kernel<1>, grid, dim3(kTPB), lds, stream, dense, sel, idx, nsel, C, Nk, Nq, k, T); break;
  }
  return ptops::check_launch("edge_scatter_kernel");
}
