// Probe: split-K weight-gradient GEMM  dW[N,K] = sum_m dY[m,n] X[m,k]  on v_mfma_f32_32x32x2_f32 (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o dw_gemm_probe dw_gemm_probe.hip && ./dw_gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TN = 128, TK = 128, RM = 16;

__global__ __launch_bounds__(256, 2) void dw_kernel(const float* __restrict__ dY, const float* __restrict__ X, int M, int N, int K,
                                                    int rows_per_split, float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float sY[2][RM * TN];
  __shared__ __attribute__((aligned(16))) float sX[2][RM * TK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int wn = wave & 1, wk = wave >> 1;
  const int n0 = blockIdx.x * TN, k0 = blockIdx.y * TK;
  const int m_begin = blockIdx.z * rows_per_split, m_end = min(M, m_begin + rows_per_split);
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x16{0};
  const int sr = tid >> 5, sc = (tid & 31) * 4;  // 8 rows x 32 float4 per pass, 2 passes
  f32x4 yr[2], xr[2];
  auto fetch = [&](int m0) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = m0 + sr + 8 * u;
      if (m < m_end) {
        yr[u] = *reinterpret_cast<const f32x4*>(dY + (size_t)m * N + n0 + sc);
        xr[u] = *reinterpret_cast<const f32x4*>(X + (size_t)m * K + k0 + sc);
      } else {
        yr[u] = f32x4{0, 0, 0, 0};
        xr[u] = f32x4{0, 0, 0, 0};
      }
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      *reinterpret_cast<f32x4*>(&sY[buf][(sr + 8 * u) * TN + sc]) = yr[u];
      *reinterpret_cast<f32x4*>(&sX[buf][(sr + 8 * u) * TK + sc]) = xr[u];
    }
  };
  const int ntiles = (m_end - m_begin + RM - 1) / RM;
  fetch(m_begin);
  store(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) fetch(m_begin + (t + 1) * RM);
    const float* ya = &sY[buf][h * TN + wn * 64 + j];
    const float* xb = &sX[buf][h * TK + wk * 64 + j];
#pragma unroll
    for (int s = 0; s < RM / 2; ++s) {
      const float a0 = ya[2 * s * TN], a1 = ya[2 * s * TN + 32];
      const float b0 = xb[2 * s * TK], b1 = xb[2 * s * TK + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (t + 1 < ntiles) store(buf ^ 1);
    __syncthreads();
  }
  float* out = partial + (size_t)blockIdx.z * N * K;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int k = k0 + wk * 64 + b * 32 + j;
        out[(size_t)n * K + k] = acc[a][b][r];
      }
}

__global__ void reduce_kernel(const float* __restrict__ partial, int S, long long NK, float* __restrict__ out) {
  const long long e = (blockIdx.x * 256LL + threadIdx.x) * 4;
  if (e >= NK) return;
  f32x4 a = {0, 0, 0, 0};
  for (int s = 0; s < S; ++s) a += *reinterpret_cast<const f32x4*>(partial + (size_t)s * NK + e);
  *reinterpret_cast<f32x4*>(out + e) = a;
}

int main() {
  const int shapes[][3] = {{18432, 384, 384}, {18432, 1152, 384}, {18432, 768, 384}, {18432, 384, 768}, {8192, 384, 384}, {18432, 1024, 1024}};
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    float *dY, *X, *partial, *out;
    hipMalloc(&dY, sizeof(float) * (size_t)M * N);
    hipMalloc(&X, sizeof(float) * (size_t)M * K);
    std::vector<float> hy((size_t)M * N), hx((size_t)M * K);
    for (size_t i = 0; i < hy.size(); ++i) hy[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 40503u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(dY, hy.data(), hy.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, sizeof(float) * (size_t)N * K);
    for (int target : {256, 512, 1024}) {
      const int tiles = (N / TN) * (K / TK);
      int S = (target + tiles - 1) / tiles;
      int rps = ((M + S - 1) / S + RM - 1) / RM * RM;
      S = (M + rps - 1) / rps;
      hipMalloc(&partial, sizeof(float) * (size_t)S * N * K);
      dim3 grid(N / TN, K / TK, S);
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      for (int it = 0; it < 3; ++it) {
        dw_kernel<<<grid, 256>>>(dY, X, M, N, K, rps, partial);
        reduce_kernel<<<(unsigned)(((long long)N * K / 4 + 255) / 256), 256>>>(partial, S, (long long)N * K, out);
      }
      hipDeviceSynchronize();
      hipEventRecord(e0);
      const int iters = 20;
      for (int it = 0; it < iters; ++it) {
        dw_kernel<<<grid, 256>>>(dY, X, M, N, K, rps, partial);
        reduce_kernel<<<(unsigned)(((long long)N * K / 4 + 255) / 256), 256>>>(partial, S, (long long)N * K, out);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= iters;
      // spot check one element
      std::vector<float> ho(4);
      hipMemcpy(ho.data(), out, 16, hipMemcpyDeviceToHost);
      double ref = 0;
      for (int m = 0; m < M; ++m) ref += (double)hy[(size_t)m * N] * hx[(size_t)m * K];
      printf("M=%d N=%d K=%d  blocks=%d (S=%d)  %.1f us  %.1f TF/s   out[0]=%.4f ref=%.4f\n", M, N, K, tiles * S, S, ms * 1e3,
             2.0 * M * N * K / ms / 1e9, ho[0], ref);
      hipFree(partial);
    }
    hipFree(dY); hipFree(X); hipFree(out);
  }
  return 0;
}
