// Row-wise top-k of a dense matrix for gfx950 (MI355X): the selection step of GDANet's feature-space kNN.
//
// Replaces `pairwise_distance.topk(k=k, dim=-1)[1]` of models/segmentation/GDANet_util.py:5-11 of the reference (called
// from local_operator :14, local_operator_withnorm :40 and GDM :73 with k = 30 / 64 on [B,512,512] matrices).  ATen's
// top-k is a radix select + a segmented sort (110-170 us for 8192 rows of 512); here one wavefront owns a row:
//  * the row lives in VGPRs as int32 KEYS ordered like the floats (P consecutive entries per lane, -0 folded into +0);
//  * each of the k rounds is one DPP max chain over the lanes' current best (6 steps), a ballot of the lanes holding the
//    maximum, s_ff1 + v_readlane of the winner's index - the lowest lane, i.e. the lowest column, wins ties, which is the
//    order oracle/ptops_ref.c::ref_topk_rows defines: (value descending, column ascending) - and the winner lane
//    retires that entry and rescans its P keys;
//  * result j is parked in lane j % 64 and stored 64 at a time, coalesced, as int64 (what torch.gather consumes).
// The matrix is read exactly once (algorithmic bytes = 4 N per row + 8 k out); the bound is the serial chain of k rounds.
#include <limits.h>
#include <stdint.h>

#include "ptops_common.h"

namespace {

#define PTOPS_DPP_STEP(ctrl) "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 " ctrl " bank_mask:0xf\n\t"

// max over the wave, see fps.hip::wave_max_i32 (one v_max_i32_dpp per step; two wait states around DPP reads)
__device__ __forceinline__ int wave_max_i32(int v) {
  asm volatile(PTOPS_DPP_STEP("row_shr:1 row_mask:0xf") PTOPS_DPP_STEP("row_shr:2 row_mask:0xf")
               PTOPS_DPP_STEP("row_shr:4 row_mask:0xf") PTOPS_DPP_STEP("row_shr:8 row_mask:0xf")
               PTOPS_DPP_STEP("row_bcast:15 row_mask:0xa") PTOPS_DPP_STEP("row_bcast:31 row_mask:0xc") "s_nop 1"
               : "+v"(v));
  return __builtin_amdgcn_readlane(v, 63);
}

// float -> int32 with the same order (finite values and infinities; -0 == +0)
__device__ __forceinline__ int ordered_key(float x) {
  const int b = __builtin_bit_cast(int, x + 0.0f);
  return b ^ ((b >> 31) & 0x7fffffff);
}

constexpr int kGone = INT_MIN;   // a retired / out-of-range entry: below the key of -inf (INT_MIN + 0x7fffff)

template <int P, bool VEC>
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ x, long long R, int N, int k,
                                                        long long* __restrict__ idx_out) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;                                   // whole waves leave together; no barrier in this kernel
  const float* __restrict__ xr = x + row * (long long)N;
  long long* __restrict__ out = idx_out + row * (long long)k;

  // all loads are issued before the first use (out-of-range lanes read entry 0 and discard it)
  int key[P];
  if constexpr (VEC) {                                    // N % 4 == 0 and a 16-byte aligned matrix: P / 4 x 128-bit loads
    float4 v[P / 4];
#pragma unroll
    for (int i = 0; i < P / 4; ++i) {
      const int c = lane * P + 4 * i;
      v[i] = *reinterpret_cast<const float4*>(xr + (c < N ? c : 0));
    }
#pragma unroll
    for (int i = 0; i < P / 4; ++i) {
      const bool in = lane * P + 4 * i < N;
      key[4 * i + 0] = in ? ordered_key(v[i].x) : kGone;
      key[4 * i + 1] = in ? ordered_key(v[i].y) : kGone;
      key[4 * i + 2] = in ? ordered_key(v[i].z) : kGone;
      key[4 * i + 3] = in ? ordered_key(v[i].w) : kGone;
    }
  } else {
    float v[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int c = lane * P + i;
      v[i] = xr[c < N ? c : 0];
    }
#pragma unroll
    for (int i = 0; i < P; ++i) key[i] = lane * P + i < N ? ordered_key(v[i]) : kGone;
  }
  int best = key[0], pos = 0;
#pragma unroll
  for (int i = 1; i < P; ++i) {
    const bool gt = key[i] > best;                        // strict: the lowest column wins lane-local ties
    pos = gt ? i : pos;
    best = gt ? key[i] : best;
  }
  int parked = 0;
  for (int j = 0; j < k; ++j) {
    const int wmax = wave_max_i32(best);
    const unsigned long long holders = __ballot(best == wmax);
    const int win = (int)__builtin_ctzll(holders);        // lowest lane = lowest column (blocked ownership)
    const int col = __builtin_amdgcn_readlane(lane * P + pos, win);
    if (lane == (j & 63)) parked = col;
    if ((j & 63) == 63 || j == k - 1) {
      const int base = j & ~63;
      if (base + lane <= j) out[base + lane] = parked;
    }
    const bool mine = lane == win;
    int nb = kGone, np = 0;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      key[i] = (mine && pos == i) ? kGone : key[i];
      const bool gt = key[i] > nb;
      np = gt ? i : np;
      nb = gt ? key[i] : nb;
    }
    best = nb;
    pos = np;
  }
}

template <int P>
int launch(const float* x, long long R, int N, int k, long long* idx, hipStream_t stream) {
  const long long blocks = (R + 3) / 4;
  if (P >= 4 && N % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0) {
    hipLaunchKernelGGL((topk_rows_kernel<P, (P >= 4)>), dim3((unsigned)blocks), dim3(256), 0, stream, x, R, N, k, idx);
  } else {
    hipLaunchKernelGGL((topk_rows_kernel<P, false>), dim3((unsigned)blocks), dim3(256), 0, stream, x, R, N, k, idx);
  }
  return ptops::check_launch("topk_rows_kernel");
}

}  // namespace

PTOPS_API int ptops_topk_rows(const float* x, long long R, int N, int k, long long* idx, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(R >= 0 && N >= 1 && k >= 1 && k <= N, "topk_rows: bad size R=%lld N=%d k=%d", R, N, k);
  PTOPS_REQUIRE(N <= 2048, "topk_rows: N=%d exceeds the in-register limit 2048", N);
  PTOPS_REQUIRE(R <= 4ll * 0x7fffffff, "topk_rows: R=%lld rows exceed the grid limit", R);
  if (R == 0) return PTOPS_OK;
  PTOPS_REQUIRE(x && idx, "topk_rows: null pointer");
  if (N <= 64) return launch<1>(x, R, N, k, idx, stream);
  if (N <= 128) return launch<2>(x, R, N, k, idx, stream);
  if (N <= 256) return launch<4>(x, R, N, k, idx, stream);
  if (N <= 512) return launch<8>(x, R, N, k, idx, stream);
  if (N <= 1024) return launch<16>(x, R, N, k, idx, stream);
  return launch<32>(x, R, N, k, idx, stream);
}
