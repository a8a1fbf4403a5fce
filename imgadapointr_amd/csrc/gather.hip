// Index-driven data movement for gfx950 (MI355X): gather_operation, index_points, the DGCNN
// edge grouping and the graph gather-max, each with its backward.
//
// All of these are HBM/L2-bound byte movers; the rules followed are the ones that matter for
// that regime: 16 B per lane wherever the layout allows, consecutive lanes on consecutive
// addresses for every streaming side, gathers only on the side that is small enough to stay
// in L2, scatter-adds through LDS (ds_add_f32) when the destination tile fits in LDS and
// through row-shaped global atomics otherwise.
#include "ptops_common.h"

namespace {

constexpr int kTPB = 256;

inline int grid_for(long long work_items, int cap = 8 * ptops::kNumCU * 4) {
  long long g = ptops::ceil_div_ll(work_items, kTPB);
  if (g < 1) g = 1;
  return (int)(g < cap ? g : cap);
}

// ------------------------------------------------------------ gather_operation
// pointnet2_utils.gather_operation (call sites models/base_blocks.py:555, utils/misc.py:38)
__global__ __launch_bounds__(kTPB) void gather_fwd_kernel(const float* __restrict__ feat,
                                                          const int32_t* __restrict__ idx, int C, int N,
                                                          int npoint, long long total,
                                                          float* __restrict__ out) {
  for (long long e = blockIdx.x * (long long)kTPB + threadIdx.x; e < total;
       e += (long long)gridDim.x * kTPB) {
    const int j = (int)(e % npoint);
    const long long bc = e / npoint;
    const int b = (int)(bc / C);
    out[e] = feat[bc * N + idx[(long long)b * npoint + j]];
  }
}

__global__ __launch_bounds__(kTPB) void gather_bwd_kernel(const float* __restrict__ grad_out,
                                                          const int32_t* __restrict__ idx, int C, int N,
                                                          int npoint, long long total,
                                                          float* __restrict__ grad_feat) {
  for (long long e = blockIdx.x * (long long)kTPB + threadIdx.x; e < total;
       e += (long long)gridDim.x * kTPB) {
    const int j = (int)(e % npoint);
    const long long bc = e / npoint;
    const int b = (int)(bc / C);
    atomicAdd(&grad_feat[bc * N + idx[(long long)b * npoint + j]], grad_out[e]);
  }
}

// ---------------------------------------------------------------- index_points
// models/Transformer_utils.py:53-69 for idx [B,S,k]; V = vector width in floats (4 or 1).
template <int V>
__global__ __launch_bounds__(kTPB) void index_points_fwd_kernel(const float* __restrict__ points,
                                                                const int32_t* __restrict__ idx, int N,
                                                                int C, int SK, long long total_vec,
                                                                float* __restrict__ out) {
  const int CV = C / V;
  for (long long e = blockIdx.x * (long long)kTPB + threadIdx.x; e < total_vec;
       e += (long long)gridDim.x * kTPB) {
    const int cv = (int)(e % CV);
    const long long row = e / CV;  // (b*S + s)*k + j
    const int b = (int)(row / SK);
    const float* src = points + ((long long)b * N + idx[row]) * C + (long long)cv * V;
    float* dst = out + row * C + (long long)cv * V;
    if (V == 4) {
      *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
    } else {
      *dst = *src;
    }
  }
}

__global__ __launch_bounds__(kTPB) void index_points_bwd_kernel(const float* __restrict__ grad_out,
                                                                const int32_t* __restrict__ idx, int N,
                                                                int C, int SK, long long total,
                                                                float* __restrict__ grad_points) {
  // consecutive lanes -> consecutive channels of one destination row (row-shaped atomics)
  for (long long e = blockIdx.x * (long long)kTPB + threadIdx.x; e < total;
       e += (long long)gridDim.x * kTPB) {
    const int c = (int)(e % C);
    const long long row = e / C;
    const int b = (int)(row / SK);
    atomicAdd(&grad_points[((long long)b * N + idx[row]) * C + c], grad_out[e]);
  }
}

// ------------------------------------------------------------------ edge group
// DGCNN_Grouper.get_graph_feature, models/base_blocks.py:565-588 (gather + concat half).
// One lane per (b, c, q, 4 neighbours): int4 index load, 4 gathers from the L2-resident row of
// x_k, two float4 stores; consecutive lanes store consecutive 16-byte pieces.
__global__ __launch_bounds__(kTPB) void edge_group_fwd_v4_kernel(
    const float* __restrict__ x_k, const float* __restrict__ x_q, const int32_t* __restrict__ idx, int C,
    int Nk, int Nq, int k, long long total_vec, float* __restrict__ out) {
  const int KV = k / 4;
  const long long per_c = (long long)Nq * KV;
  for (long long e = blockIdx.x * (long long)kTPB + threadIdx.x; e < total_vec;
       e += (long long)gridDim.x * kTPB) {
    const long long r = e % per_c;  // q*KV + jv
    const long long bc = e / per_c;
    const int b = (int)(bc / C), c = (int)(bc % C);
    const int q = (int)(r / KV);
    const int4 id = *reinterpret_cast<const int4*>(idx + ((long long)b * Nq * k + r * 4));
    const float* xk = x_k + bc * Nk;
    const float xq = x_q[bc * Nq + q];
    float4 o0, o1;
    o0.x = xk[id.x] - xq;
    o0.y = xk[id.y] - xq;
    o0.z = xk[id.z] - xq;
    o0.w = xk[id.w] - xq;
    o1.x = o1.y = o1.z = o1.w = xq;
    float* base0 = out + (((long long)b * 2 * C + c) * Nq) * k + r * 4;
    float* base1 = out + (((long long)b * 2 * C + C + c) * Nq) * k + r * 4;
    *reinterpret_cast<float4*>(base0) = o0;
    *reinterpret_cast<float4*>(base1) = o1;
  }
}

__global__ __launch_bounds__(kTPB) void edge_group_fwd_kernel(
    const float* __restrict__ x_k, const float* __restrict__ x_q, const int32_t* __restrict__ idx, int C,
    int Nk, int Nq, int k, long long total, float* __restrict__ out) {
  const long long per_c = (long long)Nq * k;
  for (long long e = blockIdx.x * (long long)kTPB + threadIdx.x; e < total;
       e += (long long)gridDim.x * kTPB) {
    const long long r = e % per_c;
    const long long bc = e / per_c;
    const int b = (int)(bc / C), c = (int)(bc % C);
    const int q = (int)(r / k);
    const float xq = x_q[bc * Nq + q];
    out[(((long long)b * 2 * C + c) * Nq) * k + r] = x_k[bc * Nk + idx[(long long)b * Nq * k + r]] - xq;
    out[(((long long)b * 2 * C + C + c) * Nq) * k + r] = xq;
  }
}

// Backward.  One workgroup per (batch, tile of CT channels): the destination rows of grad_xk
// (CT * Nk floats) live in LDS, the edges stream through coalesced float4 loads, the scatter is
// ds_add_f32, and grad_xq (sum over the k neighbours of g1 - g0) is reduced across the k/4
// adjacent lanes with DPP-able shuffles.  No global atomics.
template <int CT>
__global__ __launch_bounds__(kTPB) void edge_group_bwd_lds_kernel(
    const float* __restrict__ grad_out, const int32_t* __restrict__ idx, int C, int Nk, int Nq, int k,
    float* __restrict__ grad_xk, float* __restrict__ grad_xq) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [CT][Nk]
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * CT;
  const int tid = threadIdx.x;
  for (int t = tid; t < CT * Nk; t += kTPB) tile[t] = 0.0f;
  __syncthreads();
  const int KV = k / 4;              // lanes per query (power of two, <= 16)
  const long long nvec = (long long)Nq * KV;
  const int32_t* idb = idx + (long long)b * Nq * k;
  // every lane runs the same trip count so that the shuffles below see full groups
  const long long trips = ptops::ceil_div_ll(nvec, kTPB);
  for (long long it = 0; it < trips; ++it) {
    const long long r = it * kTPB + tid;
    const bool in = r < nvec;
    const long long rr = in ? r : 0;
    const int4 id = *reinterpret_cast<const int4*>(idb + rr * 4);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int c = c0 + ct;
      if (c >= C) break;  // uniform
      const float4 g0 = *reinterpret_cast<const float4*>(
          grad_out + (((long long)b * 2 * C + c) * Nq) * k + rr * 4);
      const float4 g1 = *reinterpret_cast<const float4*>(
          grad_out + (((long long)b * 2 * C + C + c) * Nq) * k + rr * 4);
      float part = in ? ((g1.x - g0.x) + (g1.y - g0.y)) + ((g1.z - g0.z) + (g1.w - g0.w)) : 0.0f;
      if (in) {
        atomicAdd(&tile[ct * Nk + id.x], g0.x);
        atomicAdd(&tile[ct * Nk + id.y], g0.y);
        atomicAdd(&tile[ct * Nk + id.z], g0.z);
        atomicAdd(&tile[ct * Nk + id.w], g0.w);
      }
      for (int off = 1; off < KV; off <<= 1) part += __shfl_xor(part, off);
      if (in && (r % KV) == 0) grad_xq[((long long)b * C + c) * Nq + r / KV] = part;
    }
  }
  __syncthreads();
  for (int t = tid; t < CT * Nk; t += kTPB) {
    const int ct = t / Nk, n = t - ct * Nk;
    if (c0 + ct < C) grad_xk[((long long)b * C + c0 + ct) * Nk + n] = tile[t];
  }
}

// Generic fallback (any k, any Nk): global atomics for grad_xk, serial sum for grad_xq.
__global__ __launch_bounds__(kTPB) void edge_group_bwd_generic_kernel(
    const float* __restrict__ grad_out, const int32_t* __restrict__ idx, int C, int Nk, int Nq, int k,
    long long total, float* __restrict__ grad_xk, float* __restrict__ grad_xq) {
  for (long long e = blockIdx.x * (long long)kTPB + threadIdx.x; e < total;
       e += (long long)gridDim.x * kTPB) {  // e over (b, c, q)
    const int q = (int)(e % Nq);
    const long long bc = e / Nq;
    const int b = (int)(bc / C), c = (int)(bc % C);
    const float* g0 = grad_out + ((((long long)b * 2 * C + c) * Nq) + q) * k;
    const float* g1 = grad_out + ((((long long)b * 2 * C + C + c) * Nq) + q) * k;
    const int32_t* id = idx + ((long long)b * Nq + q) * k;
    float acc = 0.0f;
    for (int j = 0; j < k; ++j) {
      atomicAdd(&grad_xk[bc * Nk + id[j]], g0[j]);
      acc += g1[j] - g0[j];
    }
    grad_xq[e] = acc;
  }
}

// ------------------------------------------------------------------ gather-max
// Neighbour half of DynamicGraphAttention (models/Transformer_utils.py:811-817) after the
// linear split: out[b,s,c] = max_j p[b, idx[b,s,j], c]; consecutive lanes own consecutive
// float4 channel groups of one (b,s), so every neighbour row is read as one coalesced burst.
// ACT: out = LeakyReLU(max + add[b,s,c]) - the query half and the activation of DynamicGraphAttention in the same pass
template <int V, bool ACT = false>
__global__ __launch_bounds__(kTPB) void gather_max_fwd_kernel(const float* __restrict__ p,
                                                              const int32_t* __restrict__ idx, int N, int C,
                                                              int S, int k, long long total_vec,
                                                              float* __restrict__ out,
                                                              int32_t* __restrict__ arg,
                                                              const float* __restrict__ add = nullptr,
                                                              float slope = 0.f) {
  const int CV = C / V;
  for (long long e = blockIdx.x * (long long)kTPB + threadIdx.x; e < total_vec;
       e += (long long)gridDim.x * kTPB) {
    const int cv = (int)(e % CV);
    const long long bs = e / CV;
    const int b = (int)(bs / S);
    const int32_t* id = idx + bs * k;
    const float* pb = p + (long long)b * N * C + (long long)cv * V;
    float best[V];
    int bi[V];
    {
      const int r = id[0];
      if (V == 4) {
        const float4 v = *reinterpret_cast<const float4*>(pb + (long long)r * C);
        best[0] = v.x; best[1 % V] = v.y; best[2 % V] = v.z; best[3 % V] = v.w;
      } else {
        best[0] = pb[(long long)r * C];
      }
#pragma unroll
      for (int t = 0; t < V; ++t) bi[t] = r;
    }
    for (int j = 1; j < k; ++j) {
      const int r = id[j];
      float v[V];
      if (V == 4) {
        const float4 t4 = *reinterpret_cast<const float4*>(pb + (long long)r * C);
        v[0] = t4.x; v[1 % V] = t4.y; v[2 % V] = t4.z; v[3 % V] = t4.w;
      } else {
        v[0] = pb[(long long)r * C];
      }
#pragma unroll
      for (int t = 0; t < V; ++t) {
        const bool gt = v[t] > best[t];
        best[t] = gt ? v[t] : best[t];
        bi[t] = gt ? r : bi[t];
      }
    }
    float* o = out + bs * C + (long long)cv * V;
    int32_t* a = arg + bs * C + (long long)cv * V;
    if (ACT) {
      const float* q = add + bs * C + (long long)cv * V;
#pragma unroll
      for (int t = 0; t < V; ++t) {
        const float v = best[t] + q[t];
        best[t] = v > 0.f ? v : v * slope;
      }
    }
    if (V == 4) {
      *reinterpret_cast<float4*>(o) = make_float4(best[0], best[1 % V], best[2 % V], best[3 % V]);
      *reinterpret_cast<int4*>(a) = make_int4(bi[0], bi[1 % V], bi[2 % V], bi[3 % V]);
    } else {
      o[0] = best[0];
      a[0] = bi[0];
    }
  }
}

__global__ __launch_bounds__(kTPB) void gather_max_bwd_kernel(const float* __restrict__ grad_out,
                                                              const int32_t* __restrict__ arg, int N, int C,
                                                              int S, long long total,
                                                              float* __restrict__ grad_p) {
  for (long long e = blockIdx.x * (long long)kTPB + threadIdx.x; e < total;
       e += (long long)gridDim.x * kTPB) {
    const int c = (int)(e % C);
    const int b = (int)(e / ((long long)S * C));
    atomicAdd(&grad_p[((long long)b * N + arg[e]) * C + c], grad_out[e]);
  }
}

// Same scatter through an LDS tile, DETERMINISTIC (round 3): a workgroup owns (b, 16 channels) and accumulates every
// query's contribution into tile[channel][source point] - as 64-bit FIXED-POINT integers (ds_add_u64), whose sum does
// not depend on the order in which the atomics arrive.  The scale is a power of two chosen from the largest |gradient|
// of the workgroup's own slice (so it is a pure function of the inputs): max |g| maps to [2^46, 2^47), up to 2^16
// addends fit 63 bits, and a value keeps all its 24 mantissa bits unless it is below 2^-23 of that maximum - an
// accumulation error far below the fp32 rounding of the sum itself.  No global atomics, no memset, N <= 1024.
constexpr int kGmCT = 16;

// ACT: grad_out is the gradient of LeakyReLU(max + add); `out` is that result; the gradient through the activation,
// g' = out > 0 ? g : g * slope (aten's leaky_relu_backward), is scattered and also written to grad_add.
template <bool ACT>
__global__ __launch_bounds__(kTPB) void gather_max_bwd_lds_kernel(const float* __restrict__ grad_out,
                                                                  const int32_t* __restrict__ arg, int N, int C,
                                                                  int S, float* __restrict__ grad_p,
                                                                  const float* __restrict__ out = nullptr,
                                                                  float slope = 0.f,
                                                                  float* __restrict__ grad_add = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long gm_tile[];  // [kGmCT][N + 1]
  __shared__ float wmax[kTPB / 64];
  const int b = blockIdx.y, c0 = blockIdx.x * kGmCT;
  const int ld = N + 1;
  for (int t = threadIdx.x; t < kGmCT * ld; t += kTPB) gm_tile[t] = 0ull;
  const int ct = threadIdx.x & (kGmCT - 1), sl = threadIdx.x / kGmCT;  // 16 channels x 16 query lanes
  const int c = c0 + ct;
  float mx = 0.0f;
  int bad = 0;                                               // a NaN or infinity in the slice
  if (c < C)
    for (int s = sl; s < S; s += kTPB / kGmCT) {
      const long long e = ((long long)b * S + s) * C + c;
      float g = grad_out[e];
      if (ACT) g = out[e] > 0.f ? g : g * slope;
      const float a = fabsf(g);
      const bool fin = a < INFINITY;
      mx = fin ? fmaxf(mx, a) : mx;
      bad |= !fin;
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
  bad = __syncthreads_or(bad);
  mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  // an infinite or NaN gradient must poison the outputs it touches: those terms are added through the float path at
  // the end (their sum is NaN / infinite in any order); the finite ones go through the fixed-point tile
  int e2 = 0;
  (void)frexpf(mx > 0.0f ? mx : 1.0f, &e2);                 // mx = f * 2^e2, f in [0.5, 1)
  const double scale = ldexp(1.0, 47 - e2), inv = ldexp(1.0, e2 - 47);
  if (c < C) {
    for (int s = sl; s < S; s += kTPB / kGmCT) {
      const long long e = ((long long)b * S + s) * C + c;
      float g = grad_out[e];
      if (ACT) {
        g = out[e] > 0.f ? g : g * slope;
        grad_add[e] = g;
      }
      const long long q = fabsf(g) < INFINITY ? __double2ll_rn((double)g * scale) : 0ll;
      atomicAdd(&gm_tile[ct * ld + arg[e]], (unsigned long long)q);
    }
  }
  __syncthreads();
  // write-out: lanes over channels (contiguous in grad_p), rows over source points
  for (int t = threadIdx.x; t < kGmCT * N; t += kTPB) {
    const int cc = t & (kGmCT - 1), n = t / kGmCT;
    if (c0 + cc < C) grad_p[((long long)b * N + n) * C + c0 + cc] = (float)((double)(long long)gm_tile[cc * ld + n] * inv);
  }
  if (bad) {                                                 // rare: add the non-finite terms on top (order is irrelevant)
    __syncthreads();
    if (c < C)
      for (int s = sl; s < S; s += kTPB / kGmCT) {
        const long long e = ((long long)b * S + s) * C + c;
        float g = grad_out[e];
        if (ACT) g = out[e] > 0.f ? g : g * slope;
        if (!(fabsf(g) < INFINITY)) atomicAdd(&grad_p[((long long)b * N + arg[e]) * C + c], g);
      }
  }
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

PTOPS_API int ptops_gather_forward(const float* feat, const int32_t* idx, int B, int C, int N,
                                   int npoint, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && C >= 0 && N >= 1 && npoint >= 0, "gather_forward: bad size");
  const long long total = (long long)B * C * npoint;
  if (total == 0) return PTOPS_OK;
  PTOPS_REQUIRE(feat && idx && out, "gather_forward: null pointer");
  hipLaunchKernelGGL(gather_fwd_kernel, dim3(grid_for(total)), dim3(kTPB), 0, stream, feat, idx, C, N,
                     npoint, total, out);
  return ptops::check_launch("gather_fwd_kernel");
}

PTOPS_API int ptops_gather_backward(const float* grad_out, const int32_t* idx, int B, int C, int N,
                                    int npoint, float* grad_feat, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && C >= 0 && N >= 1 && npoint >= 0, "gather_backward: bad size");
  if ((long long)B * C * N == 0) return PTOPS_OK;
  PTOPS_REQUIRE(grad_feat, "gather_backward: null pointer");
  (void)ptops::fill_async(grad_feat, 0, sizeof(float) * (size_t)B * C * N, stream);
  const long long total = (long long)B * C * npoint;
  if (total == 0) return ptops::check_launch("gather_backward(memset)");
  PTOPS_REQUIRE(grad_out && idx, "gather_backward: null pointer");
  hipLaunchKernelGGL(gather_bwd_kernel, dim3(grid_for(total)), dim3(kTPB), 0, stream, grad_out, idx, C,
                     N, npoint, total, grad_feat);
  return ptops::check_launch("gather_bwd_kernel");
}

PTOPS_API int ptops_index_points_forward(const float* points, const int32_t* idx, int B, int N,
                                         int C, int S, int k, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && N >= 1 && C >= 1 && S >= 0 && k >= 0, "index_points_forward: bad size");
  const long long rows = (long long)B * S * k;
  if (rows == 0) return PTOPS_OK;
  PTOPS_REQUIRE(points && idx && out, "index_points_forward: null pointer");
  if (C % 4 == 0 && aligned16(points) && aligned16(out)) {
    const long long tv = rows * (C / 4);
    hipLaunchKernelGGL(index_points_fwd_kernel<4>, dim3(grid_for(tv)), dim3(kTPB), 0, stream, points, idx,
                       N, C, S * k, tv, out);
  } else {
    const long long tv = rows * C;
    hipLaunchKernelGGL(index_points_fwd_kernel<1>, dim3(grid_for(tv)), dim3(kTPB), 0, stream, points, idx,
                       N, C, S * k, tv, out);
  }
  return ptops::check_launch("index_points_fwd_kernel");
}

PTOPS_API int ptops_index_points_backward(const float* grad_out, const int32_t* idx, int B, int N,
                                          int C, int S, int k, float* grad_points, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && N >= 1 && C >= 1 && S >= 0 && k >= 0, "index_points_backward: bad size");
  if (B == 0) return PTOPS_OK;
  PTOPS_REQUIRE(grad_points, "index_points_backward: null pointer");
  (void)ptops::fill_async(grad_points, 0, sizeof(float) * (size_t)B * N * C, stream);
  const long long total = (long long)B * S * k * C;
  if (total == 0) return ptops::check_launch("index_points_backward(memset)");
  PTOPS_REQUIRE(grad_out && idx, "index_points_backward: null pointer");
  hipLaunchKernelGGL(index_points_bwd_kernel, dim3(grid_for(total)), dim3(kTPB), 0, stream, grad_out, idx,
                     N, C, S * k, total, grad_points);
  return ptops::check_launch("index_points_bwd_kernel");
}

PTOPS_API int ptops_edge_group_forward(const float* x_k, const float* x_q, const int32_t* idx, int B,
                                       int C, int Nk, int Nq, int k, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && C >= 1 && Nk >= 1 && Nq >= 0 && k >= 1, "edge_group_forward: bad size");
  const long long total = (long long)B * C * Nq * k;
  if (total == 0) return PTOPS_OK;
  PTOPS_REQUIRE(x_k && x_q && idx && out, "edge_group_forward: null pointer");
  if (k % 4 == 0 && aligned16(idx) && aligned16(out)) {
    const long long tv = total / 4;
    hipLaunchKernelGGL(edge_group_fwd_v4_kernel, dim3(grid_for(tv)), dim3(kTPB), 0, stream, x_k, x_q, idx,
                       C, Nk, Nq, k, tv, out);
  } else {
    hipLaunchKernelGGL(edge_group_fwd_kernel, dim3(grid_for(total)), dim3(kTPB), 0, stream, x_k, x_q, idx,
                       C, Nk, Nq, k, total, out);
  }
  return ptops::check_launch("edge_group_fwd_kernel");
}

PTOPS_API int ptops_edge_group_backward(const float* grad_out, const int32_t* idx, int B, int C,
                                        int Nk, int Nq, int k, float* grad_xk, float* grad_xq,
                                        void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && C >= 1 && Nk >= 1 && Nq >= 0 && k >= 1, "edge_group_backward: bad size");
  if (B == 0) return PTOPS_OK;
  PTOPS_REQUIRE(grad_xk && (grad_xq || Nq == 0), "edge_group_backward: null pointer");
  if (Nq == 0) {
    (void)ptops::fill_async(grad_xk, 0, sizeof(float) * (size_t)B * C * Nk, stream);
    return ptops::check_launch("edge_group_backward(memset)");
  }
  PTOPS_REQUIRE(grad_out && idx, "edge_group_backward: null pointer");
  const int KV = k / 4;
  const bool pow2 = KV > 0 && (KV & (KV - 1)) == 0 && KV <= 16;
  if (k % 4 == 0 && pow2 && (kTPB % KV) == 0 && aligned16(idx) && aligned16(grad_out) &&
      (size_t)Nk * sizeof(float) <= 64 * 1024) {
    // widest channel tile that still leaves >= 512 workgroups and fits 64 KB of LDS
    int CT = 8;
    while (CT > 1 && ((long long)B * ptops::ceil_div(C, CT) < 512 || (size_t)CT * Nk * 4 > 64 * 1024)) CT >>= 1;
    dim3 grid(ptops::ceil_div(C, CT), B), block(kTPB);
    const size_t lds = (size_t)CT * Nk * sizeof(float);
    switch (CT) {
      case 8: hipLaunchKernelGGL(edge_group_bwd_lds_kernel<8>, grid, block, lds, stream, grad_out, idx, C, Nk, Nq, k, grad_xk, grad_xq); break;
      case 4: hipLaunchKernelGGL(edge_group_bwd_lds_kernel<4>, grid, block, lds, stream, grad_out, idx, C, Nk, Nq, k, grad_xk, grad_xq); break;
      case 2: hipLaunchKernelGGL(edge_group_bwd_lds_kernel<2>, grid, block, lds, stream, grad_out, idx, C, Nk, Nq, k, grad_xk, grad_xq); break;
      default: hipLaunchKernelGGL(edge_group_bwd_lds_kernel<1>, grid, block, lds, stream, grad_out, idx, C, Nk, Nq, k, grad_xk, grad_xq); break;
    }
    return ptops::check_launch("edge_group_bwd_lds_kernel");
  }
  (void)ptops::fill_async(grad_xk, 0, sizeof(float) * (size_t)B * C * Nk, stream);
  const long long total = (long long)B * C * Nq;
  hipLaunchKernelGGL(edge_group_bwd_generic_kernel, dim3(grid_for(total)), dim3(kTPB), 0, stream, grad_out,
                     idx, C, Nk, Nq, k, total, grad_xk, grad_xq);
  return ptops::check_launch("edge_group_bwd_generic_kernel");
}

PTOPS_API int ptops_gather_max_forward(const float* p, const int32_t* idx, int B, int N, int C, int S,
                                       int k, float* out, int32_t* arg, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && N >= 1 && C >= 1 && S >= 0 && k >= 1, "gather_max_forward: bad size");
  const long long total = (long long)B * S * C;
  if (total == 0) return PTOPS_OK;
  PTOPS_REQUIRE(p && idx && out && arg, "gather_max_forward: null pointer");
  if (C % 4 == 0 && aligned16(p) && aligned16(out) && aligned16(arg)) {
    const long long tv = total / 4;
    hipLaunchKernelGGL((gather_max_fwd_kernel<4, false>), dim3(grid_for(tv)), dim3(kTPB), 0, stream, p, idx, N, C, S,
                       k, tv, out, arg, (const float*)nullptr, 0.f);
  } else {
    hipLaunchKernelGGL((gather_max_fwd_kernel<1, false>), dim3(grid_for(total)), dim3(kTPB), 0, stream, p, idx, N, C,
                       S, k, total, out, arg, (const float*)nullptr, 0.f);
  }
  return ptops::check_launch("gather_max_fwd_kernel");
}

// out = LeakyReLU(max_j p[b, idx[b,s,j], :] + add[b,s,:], slope): the whole neighbour aggregation of
// DynamicGraphAttention after the linear split (models/Transformer_utils.py:811-817) in one launch.
PTOPS_API int ptops_gather_max_act_forward(const float* p, const int32_t* idx, const float* add, float slope, int B, int N,
                                           int C, int S, int k, float* out, int32_t* arg, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && N >= 1 && C >= 1 && S >= 0 && k >= 1, "gather_max_act_forward: bad size");
  const long long total = (long long)B * S * C;
  if (total == 0) return PTOPS_OK;
  PTOPS_REQUIRE(p && idx && add && out && arg, "gather_max_act_forward: null pointer");
  if (C % 4 == 0 && aligned16(p) && aligned16(out) && aligned16(arg)) {
    const long long tv = total / 4;
    hipLaunchKernelGGL((gather_max_fwd_kernel<4, true>), dim3(grid_for(tv)), dim3(kTPB), 0, stream, p, idx, N, C, S, k, tv,
                       out, arg, add, slope);
  } else {
    hipLaunchKernelGGL((gather_max_fwd_kernel<1, true>), dim3(grid_for(total)), dim3(kTPB), 0, stream, p, idx, N, C, S, k,
                       total, out, arg, add, slope);
  }
  return ptops::check_launch("gather_max_fwd_kernel");
}

// Backward of the above: g' = out > 0 ? g : g * slope -> grad_add [B,S,C]; grad_p [B,N,C] = scatter of g' by arg
// (deterministic fixed-point tile).  N <= 1024 (PTOPS_EINVAL otherwise: use the separate steps).
PTOPS_API int ptops_gather_max_act_backward(const float* grad_out, const float* out, const int32_t* arg, float slope, int B,
                                            int N, int C, int S, float* grad_p, float* grad_add, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && N >= 1 && C >= 1 && S >= 0, "gather_max_act_backward: bad size");
  // S <= 65536: the fixed-point sums hold 2^16 addends of up to 2^47 in 63 bits (a source row can be the arg-max of every query)
  PTOPS_REQUIRE(N <= 1024 && B <= 65535 && S <= 65536,
                "gather_max_act_backward: at most 1024 source rows, 65535 clouds, 65536 queries (N=%d B=%d S=%d)", N, B, S);
  if (B == 0) return PTOPS_OK;
  PTOPS_REQUIRE(grad_p, "gather_max_act_backward: null pointer");
  if ((long long)B * S * C == 0) {
    (void)ptops::fill_async(grad_p, 0, sizeof(float) * (size_t)B * N * C, stream);
    return ptops::check_launch("gather_max_act_backward(memset)");
  }
  PTOPS_REQUIRE(grad_out && out && arg && grad_add, "gather_max_act_backward: null pointer");
  const size_t lds = sizeof(unsigned long long) * (size_t)kGmCT * (N + 1);
  PTOPS_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(gather_max_bwd_lds_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(sizeof(unsigned long long) * kGmCT * 1025)) == hipSuccess,
                "gather_max_act_backward: the device refused %zu bytes of dynamic LDS", sizeof(unsigned long long) * kGmCT * 1025);
  hipLaunchKernelGGL(gather_max_bwd_lds_kernel<true>, dim3(ptops::ceil_div(C, kGmCT), B), dim3(kTPB), lds, stream, grad_out, arg,
                     N, C, S, grad_p, out, slope, grad_add);
  return ptops::check_launch("gather_max_bwd_lds_kernel");
}

PTOPS_API int ptops_gather_max_backward(const float* grad_out, const int32_t* arg, int B, int N, int C,
                                        int S, float* grad_p, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && N >= 1 && C >= 1 && S >= 0, "gather_max_backward: bad size");
  if (B == 0) return PTOPS_OK;
  PTOPS_REQUIRE(grad_p, "gather_max_backward: null pointer");
  const long long total = (long long)B * S * C;
  // the fixed-point tile needs N <= 1024 (LDS) and S <= 65536 (2^16 addends of up to 2^47 fit 63 bits); anything else, or a
  // device that refuses the 131 KB of dynamic LDS, takes the atomic kernel below
  if (total > 0 && N <= 1024 && B <= 65535 && S <= 65536 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(gather_max_bwd_lds_kernel<false>),
                          hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)(sizeof(unsigned long long) * kGmCT * 1025)) == hipSuccess) {
    PTOPS_REQUIRE(grad_out && arg, "gather_max_backward: null pointer");
    const size_t lds = sizeof(unsigned long long) * (size_t)kGmCT * (N + 1);   // <= 131 KB
    hipLaunchKernelGGL(gather_max_bwd_lds_kernel<false>, dim3(ptops::ceil_div(C, kGmCT), B), dim3(kTPB), lds, stream, grad_out,
                       arg, N, C, S, grad_p, (const float*)nullptr, 0.f, (float*)nullptr);
    return ptops::check_launch("gather_max_bwd_lds_kernel");
  }
  (void)ptops::fill_async(grad_p, 0, sizeof(float) * (size_t)B * N * C, stream);
  if (total == 0) return ptops::check_launch("gather_max_backward(memset)");
  PTOPS_REQUIRE(grad_out && arg, "gather_max_backward: null pointer");
  hipLaunchKernelGGL(gather_max_bwd_kernel, dim3(grid_for(total)), dim3(kTPB), 0, stream, grad_out, arg, N,
                     C, S, total, grad_p);
  return ptops::check_launch("gather_max_bwd_kernel");
}
