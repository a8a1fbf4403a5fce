// Column sums of a row-major fp32 matrix: out[n] = sum_m x[m, n]  (bias gradients of the dense layers).
//
// PyTorch's generic reduce kernel runs this shape (M = 8192..18432 rows, N = 96..1536 columns) on 2-5
// workgroups at ~1.4 TB/s; it is a pure streaming read, so it is laid out for the memory system instead:
// 16 lanes x float4 cover 64 contiguous columns (256 B per row), 16 row-lanes per workgroup walk a slab
// of 128 rows with eight independent loads in flight per lane; the grid has (N/64) x S workgroups, each
// writing one partial row.  A second small launch adds the S partials in a fixed order (deterministic, no
// atomics, no device-wide fences: a single-launch variant with a "last workgroup folds" ticket was 5x
// slower because every workgroup's release fence writes back its XCD's L2).
#include "ptops_common.h"

namespace {

constexpr int kRowsPerBlock = 128;

// sum over the 16 row-lanes of a 256-thread workgroup (4 per wave: two shuffles; 4 waves: LDS), fixed order
__device__ __forceinline__ float4 fold16(float4 (*red)[16], float4 acc, int cx, int ry) {
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off);
    acc.y += __shfl_xor(acc.y, off);
    acc.z += __shfl_xor(acc.z, off);
    acc.w += __shfl_xor(acc.w, off);
  }
  if ((threadIdx.x & 63) < 16) red[threadIdx.x >> 6][cx] = acc;
  __syncthreads();
  float4 s = red[0][cx];
#pragma unroll
  for (int t = 1; t < 4; ++t) {
    const float4 v = red[t][cx];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  return s;
}

__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int M, int N,
                                                             float* __restrict__ partial) {
  __shared__ float4 red[16][16];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int col = (blockIdx.x * 16 + cx) * 4;
  const bool in = col < N;
  const int r0 = blockIdx.y * kRowsPerBlock;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (in) {
    const float* base = x + (size_t)r0 * N + col;
    if (r0 + kRowsPerBlock <= M) {
      float4 v[kRowsPerBlock / 16];
#pragma unroll
      for (int u = 0; u < kRowsPerBlock / 16; ++u)
        v[u] = *reinterpret_cast<const float4*>(base + (size_t)(ry + 16 * u) * N);
#pragma unroll
      for (int u = 0; u < kRowsPerBlock / 16; ++u) {
        acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
      }
    } else {
      for (int r = r0 + ry; r < M; r += 16) {
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * N + col);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
  }
  const float4 s = fold16(red, acc, cx, ry);
  if (ry == 0 && in) *reinterpret_cast<float4*>(partial + (size_t)blockIdx.y * N + col) = s;
}

// one workgroup per 64 columns; row-lane ry adds partials ry, ry+RY, ... and the RY lanes are folded in order
// (RY = 16 for the few slabs of a column sum, 64 when up to ~1000 partial rows have to be folded in one launch)
template <int RY>
__global__ __launch_bounds__(16 * RY) void colsum_final_kernel(const float* __restrict__ partial, int S, int N,
                                                               float* __restrict__ out) {
  __shared__ float4 red[RY / 4][16];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int col = (blockIdx.x * 16 + cx) * 4;
  const bool in = col < N;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (in) {
    for (int t = ry; t < S; t += RY) {
      const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)t * N + col);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  // a wave holds four row-lanes of every column lane: fold them with two shuffles (fixed order), then the
  // RY/4 waves through LDS
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off);
    acc.y += __shfl_xor(acc.y, off);
    acc.z += __shfl_xor(acc.z, off);
    acc.w += __shfl_xor(acc.w, off);
  }
  if ((threadIdx.x & 63) < 16) red[threadIdx.x >> 6][cx] = acc;
  __syncthreads();
  if (ry == 0 && in) {
    float4 s = red[0][cx];
#pragma unroll
    for (int t = 1; t < RY / 4; ++t) {
      const float4 v = red[t][cx];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + col) = s;
  }
}

}  // namespace

int ptops_reduce::fold_rows(const float* partial, int S, int N, float* out, hipStream_t stream) {
  if (S > 64)
    hipLaunchKernelGGL(colsum_final_kernel<64>, dim3(ptops::ceil_div(N, 64)), dim3(1024), 0, stream, partial, S, N, out);
  else
    hipLaunchKernelGGL(colsum_final_kernel<16>, dim3(ptops::ceil_div(N, 64)), dim3(256), 0, stream, partial, S, N, out);
  return ptops::check_launch("colsum_final_kernel");
}

PTOPS_API size_t ptops_colsum_workspace_bytes(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  return sizeof(float) * (size_t)ptops::ceil_div(M, kRowsPerBlock) * N;
}

PTOPS_API int ptops_colsum(const float* x, int M, int N, float* out, void* workspace, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(M >= 0 && N >= 1, "colsum: bad size");
  PTOPS_REQUIRE(out != nullptr, "colsum: null output");
  if (M == 0) {
    (void)hipMemsetAsync(out, 0, sizeof(float) * (size_t)N, stream);
    return ptops::check_launch("colsum(memset)");
  }
  PTOPS_REQUIRE(x && workspace, "colsum: null pointer");
  PTOPS_REQUIRE(N % 4 == 0 && (((uintptr_t)x | (uintptr_t)workspace | (uintptr_t)out) & 15) == 0,
                "colsum: N must be a multiple of 4 and pointers 16-byte aligned");
  const int S = ptops::ceil_div(M, kRowsPerBlock);
  PTOPS_REQUIRE(S <= 65535, "colsum: M=%d exceeds the grid limit", M);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(ptops::ceil_div(N, 64), S), dim3(256), 0, stream, x, M, N,
                     (float*)workspace);
  return ptops_reduce::fold_rows((const float*)workspace, S, N, out, stream);
}
