// Column sums of a row-major fp32 matrix: out[n] = sum_m x[m, n]  (bias gradients of the dense layers).
//
// PyTorch's generic reduce kernel runs this shape (M = 8192..18432 rows, N = 96..1536 columns) on 2-5
// workgroups at ~1.4 TB/s; it is a pure streaming read, so it is laid out for the memory system instead:
// 16 lanes x float4 cover 64 contiguous columns (256 B per row), 16 row-lanes per workgroup walk a slab
// of 128 rows with eight independent loads in flight per lane; the grid has (N/64) x S workgroups, each
// writing one partial row.  A second small launch adds the S partials in a fixed order (deterministic, no
// atomics, no device-wide fences: a single-launch variant with a "last workgroup folds" ticket was 5x
// slower because every workgroup's release fence writes back its XCD's L2).
#include <mutex>
#include <vector>

#include <unordered_map>

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

// ---- deferred folds: the ~115 tiny fold launches of a backward pass (bias, LayerNorm and GELU-bias gradients) as a few
// batched ones.  The queues are keyed by stream (autograd runs backward nodes on its own thread, but on the stream of
// the forward), guarded by one mutex.
namespace {

constexpr int kMaxBatch = 96;
struct FoldBatch {
  const float* partial[kMaxBatch];
  float* out[kMaxBatch];
  int S[kMaxBatch];
  int N[kMaxBatch];
  int first_block[kMaxBatch + 1];   // prefix sums of ceil(N / 64)
  int count;
};

// the body of colsum_final_kernel<RY> for item `it`, column block `cb`
template <int RY>
__global__ __launch_bounds__(16 * RY) void colsum_final_batched_kernel(FoldBatch b) {
  __shared__ float4 red[RY / 4][16];
  int it = 0;
  while (it + 1 < b.count && (int)blockIdx.x >= b.first_block[it + 1]) ++it;   // uniform, <= 96 steps
  const float* __restrict__ partial = b.partial[it];
  const int S = b.S[it], N = b.N[it];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int col = (((int)blockIdx.x - b.first_block[it]) * 16 + cx) * 4;
  const bool in = col < N;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (in) {
    for (int t = ry; t < S; t += RY) {
      const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)t * N + col);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
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
    *reinterpret_cast<float4*>(b.out[it] + col) = s;
  }
}

struct PendingFold {
  const float* partial;
  float* out;
  int S, N;
};
// One queue per stream: deferral is switched on for ONE stream (the stream backward runs on) and only folds that are
// launched on that stream are queued; a fold on any other stream runs at once.  The queues are the only state the
// library keeps between calls (include/ptops.h); they hold no device memory.
struct FoldQueue {
  bool defer = false;
  bool poison = false;
  std::vector<PendingFold> pending;
};
std::mutex g_fold_mutex;
std::unordered_map<hipStream_t, FoldQueue> g_queues;

// debug aid (ptops_fold_defer(stream, 2)): the output of a queued fold is filled with NaN until the flush writes it
__global__ __launch_bounds__(256) void fold_poison_kernel(float* out, int N) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < N) out[i] = __builtin_nanf("");
}

template <int RY>
int launch_batches(const std::vector<PendingFold>& items, hipStream_t stream) {
  size_t i = 0;
  while (i < items.size()) {
    FoldBatch b;
    b.count = 0;
    b.first_block[0] = 0;
    while (i < items.size() && b.count < kMaxBatch) {
      const PendingFold& f = items[i++];
      b.partial[b.count] = f.partial, b.out[b.count] = f.out, b.S[b.count] = f.S, b.N[b.count] = f.N;
      b.first_block[b.count + 1] = b.first_block[b.count] + ptops::ceil_div(f.N, 64);
      ++b.count;
    }
    hipLaunchKernelGGL(colsum_final_batched_kernel<RY>, dim3(b.first_block[b.count]), dim3(16 * RY), 0, stream, b);
    const int rc = ptops::check_launch("colsum_final_batched_kernel");
    if (rc != PTOPS_OK) return rc;
  }
  return PTOPS_OK;
}

}  // namespace

int ptops_reduce::fold_rows(const float* partial, int S, int N, float* out, hipStream_t stream, bool deferrable) {
  if (deferrable) {
    bool queued = false, poison = false;
    {
      std::lock_guard<std::mutex> lock(g_fold_mutex);
      auto it = g_queues.find(stream);
      if (it != g_queues.end() && it->second.defer) {
        it->second.pending.push_back({partial, out, S, N});
        queued = true;
        poison = it->second.poison;
      }
    }
    if (queued) {
      if (poison) {
        hipLaunchKernelGGL(fold_poison_kernel, dim3(ptops::ceil_div(N, 256)), dim3(256), 0, stream, out, N);
        return ptops::check_launch("fold_poison_kernel");
      }
      return PTOPS_OK;
    }
  }
  if (S > 64)
    hipLaunchKernelGGL(colsum_final_kernel<64>, dim3(ptops::ceil_div(N, 64)), dim3(1024), 0, stream, partial, S, N, out);
  else
    hipLaunchKernelGGL(colsum_final_kernel<16>, dim3(ptops::ceil_div(N, 64)), dim3(256), 0, stream, partial, S, N, out);
  return ptops::check_launch("colsum_final_kernel");
}

// mode != 0: deferrable folds (ptops_colsum, ptops_layer_norm_backward, ptops_gelu_backward_colsum) launched on `stream`
// are queued until ptops_fold_flush(stream); the caller keeps their workspaces and outputs alive and unread until then.
// mode 2 additionally fills every queued output with NaN (debug: a premature read shows).  Returns the number of folds
// still queued on that stream (switching off does not flush).
PTOPS_API int ptops_fold_defer(void* stream_, int mode) {
  std::lock_guard<std::mutex> lock(g_fold_mutex);
  FoldQueue& q = g_queues[(hipStream_t)stream_];
  q.defer = mode != 0;
  q.poison = mode == 2;
  const int n = (int)q.pending.size();
  if (!q.defer && q.pending.empty()) g_queues.erase((hipStream_t)stream_);
  return n;
}

// Launches every fold queued on `stream`: one launch per 96 folds and size class.
PTOPS_API int ptops_fold_flush(void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  std::vector<PendingFold> small, large;
  {
    std::lock_guard<std::mutex> lock(g_fold_mutex);
    auto it = g_queues.find(stream);
    if (it == g_queues.end()) return PTOPS_OK;
    for (const PendingFold& f : it->second.pending) (f.S > 64 ? large : small).push_back(f);
    it->second.pending.clear();
    if (!it->second.defer) g_queues.erase(it);
  }
  int rc = launch_batches<16>(small, stream);
  if (rc != PTOPS_OK) return rc;
  return launch_batches<64>(large, stream);
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
    (void)ptops::fill_async(out, 0, sizeof(float) * (size_t)N, stream);
    return ptops::check_launch("colsum(memset)");
  }
  PTOPS_REQUIRE(x && workspace, "colsum: null pointer");
  PTOPS_REQUIRE(N % 4 == 0 && (((uintptr_t)x | (uintptr_t)workspace | (uintptr_t)out) & 15) == 0,
                "colsum: N must be a multiple of 4 and pointers 16-byte aligned");
  const int S = ptops::ceil_div(M, kRowsPerBlock);
  PTOPS_REQUIRE(S <= 65535, "colsum: M=%d exceeds the grid limit", M);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(ptops::ceil_div(N, 64), S), dim3(256), 0, stream, x, M, N,
                     (float*)workspace);
  return ptops_reduce::fold_rows((const float*)workspace, S, N, out, stream, true);
}


// ---------------------------------------------------------------------------------------------------------------
// Max over the tokens of [B,M,C] -> [B,C] with the arg-max token (torch.max(x, dim=1) of the global-feature pools,
// models/external/AdaPoinTr.py:110,225 and models/base_blocks.py:759 of the reference).  One launch, no scratch and no
// memset: ATen's strided reduction stages partial results behind semaphores that it zeroes with a MEMSET node, which
// does not survive in a replayed HIP graph (core.hip).  16 lanes x float4 cover 64 channels of a row; the 16 row
// groups of a workgroup fold through LDS.  Lowest token wins ties; a NaN wins over everything, as in torch.
namespace {

__global__ __launch_bounds__(256) void max_tokens_fwd_kernel(const float* __restrict__ x, int M, int C,
                                                             float* __restrict__ out, int32_t* __restrict__ arg) {
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c0 = (blockIdx.x * 16 + tx) * 4;
  const int b = blockIdx.y;
  float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int bi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};   // a row group without rows (M < 16) never wins a tie
  if (c0 < C) {
    const float* xb = x + (size_t)b * M * C + c0;
    for (int m = ty; m < M; m += 16) {
      const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)m * C);
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool take = e[i] > best[i] || (e[i] != e[i] && best[i] == best[i]) || m == ty;   // first row always; rows ascend
        best[i] = take ? e[i] : best[i];
        bi[i] = take ? m : bi[i];
      }
    }
  }
  __shared__ float sv[16][16][4];
  __shared__ int si[16][16][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sv[ty][tx][i] = best[i];
    si[ty][tx][i] = bi[i];
  }
  __syncthreads();
  if (ty == 0 && c0 < C) {
    float r[4];
    int ri[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      r[i] = sv[0][tx][i];
      ri[i] = si[0][tx][i];
    }
    for (int g = 1; g < 16; ++g) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = sv[g][tx][i];
        const int vi = si[g][tx][i];
        const bool r_nan = r[i] != r[i], v_nan = v != v;
        // larger value, or equal value at a lower token, or the first NaN
        const bool take = (!r_nan && (v_nan || v > r[i] || (v == r[i] && vi < ri[i]))) || (r_nan && v_nan && vi < ri[i]);
        r[i] = take ? v : r[i];
        ri[i] = take ? vi : ri[i];
      }
    }
    *reinterpret_cast<float4*>(out + (size_t)b * C + c0) = make_float4(r[0], r[1], r[2], r[3]);
    *reinterpret_cast<int4*>(arg + (size_t)b * C + c0) = make_int4(ri[0], ri[1], ri[2], ri[3]);
  }
}

// grad_x[b,m,c] = (arg[b,c] == m) ? g[b,c] : 0, every element written once (no zero fill, no scatter)
__global__ __launch_bounds__(256) void max_tokens_bwd_kernel(const float* __restrict__ g, const int32_t* __restrict__ arg,
                                                             int M, int C4, long long total4, float* __restrict__ gx) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += gridDim.x * 256LL) {
    const int c4 = (int)(i % C4);
    const long long bm = i / C4;
    const int m = (int)(bm % M);
    const long long b = bm / M;
    const float4 gv = reinterpret_cast<const float4*>(g)[b * C4 + c4];
    const int4 av = reinterpret_cast<const int4*>(arg)[b * C4 + c4];
    reinterpret_cast<float4*>(gx)[i] = make_float4(av.x == m ? gv.x : 0.f, av.y == m ? gv.y : 0.f,
                                                   av.z == m ? gv.z : 0.f, av.w == m ? gv.w : 0.f);
  }
}

}  // namespace

PTOPS_API int ptops_max_tokens_forward(const float* x, int B, int M, int C, float* out, int32_t* arg, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && M >= 1 && C >= 4 && C % 4 == 0, "max_tokens_forward: need M >= 1 and C a multiple of 4");
  if (B == 0) return PTOPS_OK;
  PTOPS_REQUIRE(x && out && arg, "max_tokens_forward: null pointer");
  PTOPS_REQUIRE(B <= 65535, "max_tokens_forward: B too large");
  hipLaunchKernelGGL(max_tokens_fwd_kernel, dim3((unsigned)ptops::ceil_div(C / 4, 16), (unsigned)B), dim3(256), 0, stream,
                     x, M, C, out, arg);
  return ptops::check_launch("max_tokens_fwd_kernel");
}

PTOPS_API int ptops_max_tokens_backward(const float* grad_out, const int32_t* arg, int B, int M, int C, float* grad_x,
                                        void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && M >= 1 && C >= 4 && C % 4 == 0, "max_tokens_backward: need M >= 1 and C a multiple of 4");
  if (B == 0) return PTOPS_OK;
  PTOPS_REQUIRE(grad_out && arg && grad_x, "max_tokens_backward: null pointer");
  const long long total4 = (long long)B * M * (C / 4);
  const long long blocks = ptops::ceil_div_ll(total4, 256 * 4);
  hipLaunchKernelGGL(max_tokens_bwd_kernel, dim3((unsigned)(blocks < 8192 ? (blocks > 0 ? blocks : 1) : 8192)), dim3(256), 0,
                     stream, grad_out, arg, M, C / 4, total4, grad_x);
  return ptops::check_launch("max_tokens_bwd_kernel");
}
