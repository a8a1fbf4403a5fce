// Farthest point sampling for gfx950 (MI355X).
//
// Replaces pointnet2_ops.pointnet2_utils.furthest_point_sample (third-party CUDA, call sites
// models/base_blocks.py:550 and utils/misc.py:37 of the reference).  Semantics are the ones
// restated in oracle/ptops_ref.c::ref_furthest_point_sample.
//
// The algorithm is serial in npoint (each pick depends on the previous one), so the design
// minimises the latency of one iteration rather than HBM traffic (the cloud is read once):
//  * one workgroup per cloud; the whole cloud and its running min-distance live in VGPRs
//    (P points per lane), nothing is re-read from memory inside the loop;
//  * the per-iteration arg-max is a two-step wave reduction on DPP (max of the value, then min
//    of the index among lanes holding that value — lowest index wins ties), one LDS slot per
//    wave, ONE barrier per iteration (slots are double-buffered by iteration parity), and every
//    wave redundantly reduces the <=16 slots so no second barrier is needed;
//  * the coordinates of the selected point come from an LDS copy of the cloud (wave-uniform
//    address -> broadcast read) when it fits, else from a scalar global load.
//  * points with |p|^2 <= 1e-3 (zero padding) get min-distance -2 at load time, which the
//    update min(d, -2) keeps forever and the arg-max (start value -1) can never pick.
#include <math.h>
#include <stdlib.h>

#include "ptops_common.h"

namespace {

// max over the wave (all lanes receive it) — DPP row shifts + row broadcasts, then readlane 63.
__device__ __forceinline__ float wave_max_f32(float v) {
  // row_shr:1,2,4,8 within each row of 16, then row_bcast:15 and row_bcast:31 (GFX9 DPP).
  float t;
  t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
  v = fmaxf(v, t);
  t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, false));
  v = fmaxf(v, t);
  t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, false));
  v = fmaxf(v, t);
  t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, false));
  v = fmaxf(v, t);
  t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));
  v = fmaxf(v, t);
  t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));
  v = fmaxf(v, t);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  unsigned t;
  t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false);
  v = min(v, t);
  t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false);
  v = min(v, t);
  t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false);
  v = min(v, t);
  t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xf, 0xf, false);
  v = min(v, t);
  t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false);
  v = min(v, t);
  t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false);
  v = min(v, t);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

template <int T, int P, bool LDS_XYZ>
__global__ __launch_bounds__(T) void fps_kernel(const float* __restrict__ xyz, int N, int npoint,
                                                int32_t* __restrict__ idx_out) {
  constexpr int W = T / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // layout: [2][W] float slot_val | [2][W] int slot_idx | (LDS_XYZ ? N*3 floats : 0)
  float* slot_val = smem;
  int* slot_idx = reinterpret_cast<int*>(smem + 2 * W);
  float* sxyz = smem + 4 * W;

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const float* __restrict__ p = xyz + (size_t)b * N * 3;
  int32_t* __restrict__ out = idx_out + (size_t)b * npoint;

  float x[P], y[P], z[P], mind[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int k = tid + i * T;  // strided ownership: coalesced load, ascending k per lane
    const bool in = k < N;
    const int kk = in ? k : 0;
    x[i] = p[kk * 3 + 0];
    y[i] = p[kk * 3 + 1];
    z[i] = p[kk * 3 + 2];
    const float mag = __builtin_fmaf(z[i], z[i], __builtin_fmaf(y[i], y[i], x[i] * x[i]));
    mind[i] = (in && !(mag <= 1e-3f)) ? 1e10f : -2.0f;
    if (LDS_XYZ && in) {
      sxyz[k * 3 + 0] = x[i];
      sxyz[k * 3 + 1] = y[i];
      sxyz[k * 3 + 2] = z[i];
    }
  }
  if (tid == 0 && npoint > 0) out[0] = 0;
  if (LDS_XYZ) __syncthreads();

  int old = 0;
  for (int j = 1; j < npoint; ++j) {
    float x1, y1, z1;
    if (LDS_XYZ) {
      x1 = sxyz[old * 3 + 0];
      y1 = sxyz[old * 3 + 1];
      z1 = sxyz[old * 3 + 2];
    } else {
      x1 = p[(size_t)old * 3 + 0];
      y1 = p[(size_t)old * 3 + 1];
      z1 = p[(size_t)old * 3 + 2];
    }
    float best = -1.0f;
    int besti = 0;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const float dx = x[i] - x1, dy = y[i] - y1, dz = z[i] - z1;
      const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
      const float d2 = d < mind[i] ? d : mind[i];
      mind[i] = d2;
      const bool gt = d2 > best;  // strict: lowest k (ascending i) wins lane-local ties
      besti = gt ? tid + i * T : besti;
      best = gt ? d2 : best;
    }
    // wave arg-max: value first, then the lowest index among the lanes that hold it
    const float wmax = wave_max_f32(best);
    const unsigned cand = (best == wmax) ? (unsigned)besti : 0xFFFFFFFFu;
    const unsigned widx = wave_min_u32(cand);
    const int par = j & 1;
    if (lane == 0) {
      slot_val[par * W + wave] = wmax;
      slot_idx[par * W + wave] = (int)widx;
    }
    __syncthreads();
    // every wave reduces the W slots itself (W <= 16: lanes >= W contribute neutral values)
    float v = lane < W ? slot_val[par * W + lane] : -3.0f;
    unsigned vi = lane < W ? (unsigned)slot_idx[par * W + lane] : 0xFFFFFFFFu;
    const float bmax = wave_max_f32(v);
    const unsigned bidx = wave_min_u32(v == bmax ? vi : 0xFFFFFFFFu);
    // all points invalid (every value -2 < -1): the reference keeps besti = 0
    old = bmax > -1.0f ? (int)bidx : 0;
    if (tid == 0) out[j] = old;
  }
}

template <int T, int P>
int launch(const float* xyz, int B, int N, int npoint, int32_t* idx, hipStream_t stream) {
  constexpr int W = T / 64;
  const size_t base = sizeof(float) * 4 * W;
  const size_t with_xyz = base + sizeof(float) * 3 * (size_t)N;
  if (with_xyz <= 64 * 1024) {
    hipLaunchKernelGGL((fps_kernel<T, P, true>), dim3(B), dim3(T), with_xyz, stream, xyz, N, npoint, idx);
  } else {
    hipLaunchKernelGGL((fps_kernel<T, P, false>), dim3(B), dim3(T), base, stream, xyz, N, npoint, idx);
  }
  return ptops::check_launch("fps_kernel");
}

}  // namespace

PTOPS_API int ptops_furthest_point_sample(const float* xyz, int B, int N, int npoint,
                                          int32_t* idx, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && N >= 1 && npoint >= 0, "furthest_point_sample: bad size B=%d N=%d npoint=%d", B, N, npoint);
  if (B == 0 || npoint == 0) return PTOPS_OK;
  PTOPS_REQUIRE(xyz && idx, "furthest_point_sample: null pointer");
  PTOPS_REQUIRE(N <= 16384, "furthest_point_sample: N=%d exceeds the in-register limit 16384", N);
#ifdef PTOPS_TUNING  // benchmarks/fps_tune.py: PTOPS_TUNE_FPS = threads * 100 + points per lane
  if (const char* t = getenv("PTOPS_TUNE_FPS")) {
    const int code = atoi(t);
    if (code == 25608 && N <= 2048) return launch<256, 8>(xyz, B, N, npoint, idx, stream);
    if (code == 51204 && N <= 2048) return launch<512, 4>(xyz, B, N, npoint, idx, stream);
    if (code == 102402 && N <= 2048) return launch<1024, 2>(xyz, B, N, npoint, idx, stream);
    if (code == 12816 && N <= 2048) return launch<128, 16>(xyz, B, N, npoint, idx, stream);
    if (code == 12804 && N <= 512) return launch<128, 4>(xyz, B, N, npoint, idx, stream);
    if (code == 6408 && N <= 512) return launch<64, 8>(xyz, B, N, npoint, idx, stream);
    if (code == 51201 && N <= 512) return launch<512, 1>(xyz, B, N, npoint, idx, stream);
    if (code == 25602 && N <= 512) return launch<256, 2>(xyz, B, N, npoint, idx, stream);
  }
#endif
  if (N <= 256) return launch<256, 1>(xyz, B, N, npoint, idx, stream);
  if (N <= 512) return launch<256, 2>(xyz, B, N, npoint, idx, stream);
  if (N <= 1024) return launch<512, 2>(xyz, B, N, npoint, idx, stream);
  if (N <= 2048) return launch<256, 8>(xyz, B, N, npoint, idx, stream);   // 330 vs 346 us for 512 picks (fps_tune.py)
  if (N <= 4096) return launch<1024, 4>(xyz, B, N, npoint, idx, stream);
  if (N <= 8192) return launch<1024, 8>(xyz, B, N, npoint, idx, stream);
  return launch<1024, 16>(xyz, B, N, npoint, idx, stream);
}
