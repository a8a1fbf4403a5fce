// Ceiling probe: v_mfma_f32_32x32x2_f32 issue rate on gfx950 with register operands.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ACC>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a0, float b0) {
  f32x16 acc[ACC];
#pragma unroll
  for (int i = 0; i < ACC; ++i) acc[i] = f32x16{0};
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < ACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ACC>
void run(int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;  // 4 waves per block -> one block per SIMD quad slot
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  const int iters = 4000;
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  probe<ACC><<<blocks, 256>>>(out, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(s);
  probe<ACC><<<blocks, 256>>>(out, iters, 1.f, 2.f);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  const double flops = 2.0 * 32 * 32 * 2 * 8.0 * ACC * iters * (double)blocks * 4;
  printf("independent accumulators %d, waves/SIMD %d: %.1f TFLOP/s\n", ACC, waves_per_simd, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int w = 1; w <= 4; ++w) run<1>(w);
  for (int w = 1; w <= 4; ++w) run<2>(w);
  for (int w = 1; w <= 2; ++w) run<4>(w);
  return 0;
}
