// Probe: LDS-free split-bf16 fp32 GEMM (forward form C = A . W^T, both k contiguous), gfx950.
//
// gemm_split3_kernel (csrc/gemm_split3.hip) stages both operands through LDS with one barrier per 16-deep k-step; its
// matrix pipe is ~49 % busy and every re-arrangement of that loop landed within 0.7x .. 1.04x of it.
// This probe removes LDS and barriers altogether:
//   * the WEIGHT operand is split ONCE into its three bf16 pieces and stored in MFMA fragment order
//     (pack_weights_kernel: [n-block of 32][k-step][piece][lane] x 16 bytes), so a wave fetches a B fragment as one
//     coalesced 1 KB load and spends no vector work on it;
//   * every wave loads its A rows straight from global memory in fragment order (lane = (row, k-half): 8 consecutive
//     fp32 = two dwordx4), splits them in registers and multiplies; waves are independent (no __syncthreads), the
//     reuse of A between the two waves of a row block and of W between the row blocks is left to L1 / L2;
//   * per-wave tile 64 x 64 (TM x TN blocks of 32 x 32): 24 MFMAs per k-step against 4 A loads + 6 W loads.
// Same arithmetic and summation order as gemm_split3_kernel: results must be BIT-IDENTICAL.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include -I../../imgadapointr_amd/csrc \
//        gemm_direct_probe.hip -L../../imgadapointr_amd -lptops -Wl,-rpath,'$ORIGIN/../../imgadapointr_amd' -o gemm_direct_probe
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

// W [N, ldw] fp32 -> packed[((nb * ksteps + ks) * 3 + piece) * 64 + lane] (uint4): lane = (col = lane & 31, half = lane >> 5)
// holds the 8 bf16 of W[nb * 32 + col][ks * 16 + half * 8 ..].  Rows beyond N are clamped (their products are never stored).
__global__ __launch_bounds__(64) void pack_weights_kernel(const float* __restrict__ W, long long ldw, int N, int K,
                                                          uint4* __restrict__ packed) {
  const int ksteps = K / 16;
  const int nb = blockIdx.x / ksteps, ks = blockIdx.x - nb * ksteps;
  const int lane = threadIdx.x, col = lane & 31, half = lane >> 5;
  const int row = min(nb * 32 + col, N - 1);
  const float4* p = reinterpret_cast<const float4*>(W + (long long)row * ldw + ks * 16 + half * 8);
  const float4 u = p[0], v = p[1];
  float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
  const Pieces q = split8(x);
  uint4* dst = packed + ((long long)blockIdx.x * 3) * 64 + lane;
  dst[0] = q.h, dst[64] = q.m, dst[128] = q.l;
}

struct DirectArgs {
  const float* A;
  const uint4* Wp;     // packed weights
  float* C;
  const float* bias;
  long long lda, ldc;
  int M, N, K;
  int tiles_n;         // workgroup tiles along N
};

// TM x TN blocks of 32 x 32 per wave; WM x WN waves per workgroup (sharing a CU's L1, nothing else).
template <int TM, int TN, int WM, int WN, int OCC>
__global__ __launch_bounds__(WM* WN * 64, OCC) void gemm_direct_kernel(DirectArgs g) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave / WN, wc = wave - wr * WN;
  const int col = lane & 31, half = lane >> 5;
  int bid;
  {
    const int orig = blockIdx.x, nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  }
  const int tile_m = bid / g.tiles_n, tile_n = bid - tile_m * g.tiles_n;
  const int m0 = tile_m * (WM * TM * 32) + wr * (TM * 32);
  const int n0 = tile_n * (WN * TN * 32) + wc * (TN * 32);
  const int ksteps = g.K / 16;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // this lane's A rows (one per row block) and W fragment bases (one per column block)
  const float* arow[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) arow[i] = g.A + (long long)min(m0 + i * 32 + col, g.M - 1) * g.lda + half * 8;
  const uint4* wfrag[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nb = min((n0 >> 5) + j, (g.N + 31) / 32 - 1);
    wfrag[j] = g.Wp + ((long long)nb * ksteps * 3) * 64 + lane;
  }

  float4 araw[TM][2];
  uint4 wraw[TN][3];
  auto load = [&](int ks) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float4* p = reinterpret_cast<const float4*>(arow[i] + ks * 16);
      araw[i][0] = p[0], araw[i][1] = p[1];
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) wraw[j][p] = wfrag[j][((long long)ks * 3 + p) * 64];
  };
  load(0);
  for (int ks = 0; ks < ksteps; ++ks) {
    bf16x8 a[TM][3], b[TN][3];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float x[8] = {araw[i][0].x, araw[i][0].y, araw[i][0].z, araw[i][0].w, araw[i][1].x, araw[i][1].y, araw[i][1].z, araw[i][1].w};
      const Pieces q = split8(x);
      a[i][0] = __builtin_bit_cast(bf16x8, q.h), a[i][1] = __builtin_bit_cast(bf16x8, q.m), a[i][2] = __builtin_bit_cast(bf16x8, q.l);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) b[j][p] = __builtin_bit_cast(bf16x8, wraw[j][p]);
    if (ks + 1 < ksteps) load(ks + 1);       // next step's operands are in flight under this step's MFMAs
#pragma unroll
    for (int term = 0; term < 6; ++term) {
      const int pa = term == 0 ? 2 : ((term == 1 || term == 3) ? 1 : 0);
      const int pb = term == 2 ? 2 : ((term == 1 || term == 4) ? 1 : 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[j][pb], acc[i][j], 0, 0, 0);
    }
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + j * 32 + col;
      if (n >= g.N) continue;
      const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < g.M) g.C[(long long)m * g.ldc + n] = acc[i][j][r] + bv;
      }
    }
}

template <int TM, int TN, int WM, int WN, int OCC>
void launch_direct(const float* A, long long lda, const uint4* Wp, float* C, long long ldc, int M, int N, int K, hipStream_t s) {
  DirectArgs g;
  g.A = A, g.Wp = Wp, g.C = C, g.bias = nullptr, g.lda = lda, g.ldc = ldc, g.M = M, g.N = N, g.K = K;
  const int bm = WM * TM * 32, bn = WN * TN * 32;
  g.tiles_n = (N + bn - 1) / bn;
  const int blocks = ((M + bm - 1) / bm) * g.tiles_n;
  hipLaunchKernelGGL((gemm_direct_kernel<TM, TN, WM, WN, OCC>), dim3(blocks), dim3(WM * WN * 64), 0, s, g);
}

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
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
    int M, N, K;
    const char* what;
  };
  const Shape shapes[] = {{18432, 384, 384, "fwd 18432x384x384"},   {18432, 1152, 384, "fwd 18432x1152x384"},
                          {18432, 768, 384, "fwd 18432x768x384"},   {18432, 384, 768, "fwd 18432x384x768"},
                          {18432, 1536, 384, "fwd 18432x1536x384"}, {18432, 384, 1536, "fwd 18432x384x1536"},
                          {18432, 1024, 1024, "fwd 18432x1024x1024"}, {8192, 384, 384, "fwd 8192x384x384"},
                          {8192, 1152, 384, "fwd 8192x1152x384"},   {8192, 384, 1536, "fwd 8192x384x1536"}};
  hipStream_t stream;
  CK(hipStreamCreate(&stream));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  constexpr int NV = 6;
  printf("%-22s %9s | %9s %9s %9s %9s %9s | pack us | TFLOP/s (fp32 product) same order; bitwise vs library kernel\n", "shape",
         "split3 us", "2x2 w2x2", "2x2 w2x2o3", "1x2 w4x2", "2x2 w4x1", "2x4 w2x1");
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
    float *A, *B, *C[NV];
    uint4* Wp;
    CK(hipMalloc(&A, na * 4));
    CK(hipMalloc(&B, nb * 4));
    for (int w = 0; w < NV; ++w) CK(hipMalloc(&C[w], nc * 4));
    const int nblocks = (s.N + 31) / 32, ksteps = s.K / 16;
    CK(hipMalloc(&Wp, (size_t)nblocks * ksteps * 3 * 64 * 16));
    CK(hipMemcpy(A, ha.data(), na * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hb.data(), nb * 4, hipMemcpyHostToDevice));
    auto pack = [&]() { hipLaunchKernelGGL(pack_weights_kernel, dim3(nblocks * ksteps), dim3(64), 0, stream, B, (long long)s.K, s.N, s.K, Wp); };
    pack();
    auto run = [&](int which) {
      switch (which) {
        case 0:
          if (ptops_gemm_f32_split3(A, s.K, 1, B, s.K, 1, C[0], s.N, s.M, s.N, s.K, nullptr, stream) != 0) exit(1);
          break;
        case 1: launch_direct<2, 2, 2, 2, 2>(A, s.K, Wp, C[1], s.N, s.M, s.N, s.K, stream); break;
        case 2: launch_direct<2, 2, 2, 2, 3>(A, s.K, Wp, C[2], s.N, s.M, s.N, s.K, stream); break;
        case 3: launch_direct<1, 2, 4, 2, 2>(A, s.K, Wp, C[3], s.N, s.M, s.N, s.K, stream); break;
        case 4: launch_direct<2, 2, 4, 1, 2>(A, s.K, Wp, C[4], s.N, s.M, s.N, s.K, stream); break;
        default: launch_direct<2, 4, 2, 1, 1>(A, s.K, Wp, C[5], s.N, s.M, s.N, s.K, stream); break;
      }
    };
    std::vector<float> ts[NV], tp;
    for (int w = 0; w < NV; ++w) run(w);
    CK(hipStreamSynchronize(stream));
    CK(hipGetLastError());
    for (int round = 0; round < 7; ++round) {
      for (int w = 0; w < NV; ++w) {
        CK(hipEventRecord(e0, stream));
        for (int i = 0; i < 5; ++i) run(w);
        CK(hipEventRecord(e1, stream));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ts[w].push_back(ms * 1000.f / 5);
      }
      CK(hipEventRecord(e0, stream));
      for (int i = 0; i < 5; ++i) pack();
      CK(hipEventRecord(e1, stream));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      tp.push_back(ms * 1000.f / 5);
    }
    CK(hipGetLastError());
    std::vector<float> h0(nc), h(nc);
    CK(hipMemcpy(h0.data(), C[0], nc * 4, hipMemcpyDeviceToHost));
    char same[NV + 1] = {0};
    for (int w = 1; w < NV; ++w) {
      CK(hipMemcpy(h.data(), C[w], nc * 4, hipMemcpyDeviceToHost));
      same[w - 1] = memcmp(h0.data(), h.data(), nc * 4) == 0 ? '=' : 'X';
    }
    const double fl = 2.0 * s.M * s.N * s.K;
    printf("%-22s %9.1f |", s.what, median(ts[0]));
    for (int w = 1; w < NV; ++w) printf(" %9.1f", median(ts[w]));
    printf(" | %7.1f |", median(tp));
    for (int w = 0; w < NV; ++w) printf(" %6.1f", fl / median(ts[w]) * 1e-6);
    printf("  %s\n", same);
    fflush(stdout);
    CK(hipFree(A));
    CK(hipFree(B));
    CK(hipFree(Wp));
    for (int w = 0; w < NV; ++w) CK(hipFree(C[w]));
  }
  return 0;
}
