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
//  * a lane owns P CONSECUTIVE points, so "lowest index wins ties" is "lowest lane wins" inside a wave and "lowest wave
//    wins" across waves: the per-iteration arg-max is ONE DPP max chain (6 steps) + a ballot of the lanes holding the
//    maximum + s_ff1 + v_readlane of that lane's index, per wave; one 64-bit LDS slot per wave, ONE barrier per
//    iteration (slots are double-buffered by iteration parity); every wave then reduces the <= 16 slots itself with a
//    4-step DPP chain inside one row and the same ballot trick.  (Round 3 ran a max chain and a min-index chain at both
//    levels: 24 DPP steps per iteration, 0.64 us per pick; now 10.)
//  * the coordinates of the selected point come from an LDS copy of the cloud (wave-uniform
//    address -> broadcast read) when it fits, else from a scalar global load.
//  * points with |p|^2 <= 1e-3 (zero padding) get min-distance -2 at load time, which the
//    update min(d, -2) keeps forever and the arg-max (start value -1) can never pick.
#include <math.h>
#include <stdlib.h>

#include "ptops_common.h"

namespace {

// Max reductions on DPP, written as ISA: the compiler lowers fmaxf(v, update_dpp(v)) to v_mov_b32 + v_mov_b32_dpp + a
// canonicalising v_max + v_max (five instructions and a nop per step, 22 cycles); `v_max_f32_dpp` does a step in one
// instruction.  A lane whose source lane does not exist (bound_ctrl off) or whose row is masked keeps its value.  The
// assembler does not see hazards inside an asm block: a DPP read of a VGPR written by the previous VALU instruction needs
// two wait states (s_nop 1), also in front of the first step and after the last one (v_readlane follows).
// The reduced values are KEYS (see fps_kernel): int32, ordered like the distances they encode.
#define PTOPS_DPP_STEP(ctrl) "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 " ctrl " bank_mask:0xf\n\t"

// max over the wave (all lanes receive it): row_shr 1, 2, 4, 8 inside each row of 16, row_bcast:15 into rows 1 and 3,
// row_bcast:31 into rows 2 and 3; lane 63 then holds the maximum
__device__ __forceinline__ int wave_max_i32(int v) {
  asm volatile(PTOPS_DPP_STEP("row_shr:1 row_mask:0xf") PTOPS_DPP_STEP("row_shr:2 row_mask:0xf")
               PTOPS_DPP_STEP("row_shr:4 row_mask:0xf") PTOPS_DPP_STEP("row_shr:8 row_mask:0xf")
               PTOPS_DPP_STEP("row_bcast:15 row_mask:0xa") PTOPS_DPP_STEP("row_bcast:31 row_mask:0xc") "s_nop 1"
               : "+v"(v));
  return __builtin_amdgcn_readlane(v, 63);
}

// max over lanes 0..15 (one DPP row); all lanes receive it
__device__ __forceinline__ int row0_max_i32(int v) {
  asm volatile(PTOPS_DPP_STEP("row_shr:1 row_mask:0xf") PTOPS_DPP_STEP("row_shr:2 row_mask:0xf")
               PTOPS_DPP_STEP("row_shr:4 row_mask:0xf") PTOPS_DPP_STEP("row_shr:8 row_mask:0xf") "s_nop 1"
               : "+v"(v));
  return __builtin_amdgcn_readlane(v, 15);
}

// the value `x` of the lowest lane whose `v` equals the wave-uniform `m` (some lane does)
__device__ __forceinline__ int value_of_first_lane_with(int v, int m, int x) {
  const unsigned long long mask = __ballot(v == m);
  return __builtin_amdgcn_readlane(x, (int)__builtin_ctzll(mask));
}

template <int T, int P, bool LDS_XYZ>
__global__ __launch_bounds__(T) void fps_kernel(const float* __restrict__ xyz, int N, int npoint,
                                                int32_t* __restrict__ idx_out) {
  constexpr int W = T / 64;
  static_assert(W >= 1 && W <= 16, "the slot reduction runs inside one DPP row");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // layout: [2][16] (value, index) slots as 64-bit words | (LDS_XYZ ? N float4 : 0)
  int2* slots = reinterpret_cast<int2*>(smem);
  float4* sxyz = reinterpret_cast<float4*>(smem + 64);

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const float* __restrict__ p = xyz + (size_t)b * N * 3;
  int32_t* __restrict__ out = idx_out + (size_t)b * npoint;

  // Running minimum distances are kept as int32 KEYS: the bit pattern of the (non-negative) float for a valid point -
  // ordered like the floats -, kInvalid for zero padding / out-of-range slots, and kNone as the arg-max start value.
  // kInvalid < kNone < every valid key, so "min(d, -2) stays -2" and "start value -1 never loses to -2" of the float
  // formulation (oracle/ptops_ref.c) hold, and min / compare / DPP max are plain integer instructions (the float forms
  // drag a canonicalising v_max per loop-carried operand along).  A NaN distance (bits above every finite key) never
  // wins the min against a running minimum, as `d < mind` never did.
  constexpr int kInvalid = -2, kNone = -1;
  float x[P], y[P], z[P];
  int mind[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int k = tid * P + i;  // blocked ownership: ascending k per lane, per wave and across waves
    const bool in = k < N;
    const int kk = in ? k : 0;
    x[i] = p[kk * 3 + 0];
    y[i] = p[kk * 3 + 1];
    z[i] = p[kk * 3 + 2];
    const float mag = __builtin_fmaf(z[i], z[i], __builtin_fmaf(y[i], y[i], x[i] * x[i]));
    mind[i] = (in && !(mag <= 1e-3f)) ? __builtin_bit_cast(int, 1e10f) : kInvalid;
    if (LDS_XYZ && in) sxyz[k] = make_float4(x[i], y[i], z[i], 0.f);
  }
  if (tid == 0 && npoint > 0) out[0] = 0;
  if (LDS_XYZ) __syncthreads();

  int old = 0;
  for (int j = 1; j < npoint; ++j) {
    float x1, y1, z1;
    if (LDS_XYZ) {
      const float4 c = sxyz[old];       // wave-uniform address: one broadcast read
      x1 = c.x;
      y1 = c.y;
      z1 = c.z;
    } else {
      x1 = p[(size_t)old * 3 + 0];
      y1 = p[(size_t)old * 3 + 1];
      z1 = p[(size_t)old * 3 + 2];
    }
    int best = kNone;
    int besti = 0;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const float dx = x[i] - x1, dy = y[i] - y1, dz = z[i] - z1;
      const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
      const int d2 = min(__builtin_bit_cast(int, d), mind[i]);
      mind[i] = d2;
      const bool gt = d2 > best;  // strict: lowest k (ascending i) wins lane-local ties
      besti = gt ? tid * P + i : besti;
      best = gt ? d2 : best;
    }
    // wave arg-max: the maximum, then the index held by the LOWEST lane that has it (= the lowest index)
    const int wmax = wave_max_i32(best);
    const int widx = value_of_first_lane_with(best, wmax, besti);
    if (W == 1) {
      old = wmax > kNone ? widx : 0;   // all points invalid: the reference keeps besti = 0
    } else {
      const int par = j & 1;
      if (lane == 0) slots[par * 16 + wave] = make_int2(wmax, widx);
      __syncthreads();
      // every wave reduces the W slots itself (lanes >= W contribute a key no slot can hold)
      const int2 s = slots[par * 16 + (lane & 15)];
      const int v = lane < W ? s.x : kInvalid - 1;
      const int bmax = row0_max_i32(v);
      const int bidx = value_of_first_lane_with(v, bmax, s.y);     // lowest wave = lowest index
      old = bmax > kNone ? bidx : 0;
    }
    if (tid == 0) out[j] = old;
  }
}

template <int T, int P>
int launch(const float* xyz, int B, int N, int npoint, int32_t* idx, hipStream_t stream) {
  const size_t base = sizeof(float) * 64;                    // 2 x 16 slots of 8 bytes
  const size_t with_xyz = base + sizeof(float) * 4 * (size_t)N;
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
