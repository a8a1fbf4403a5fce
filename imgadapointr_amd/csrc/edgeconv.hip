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
    const float* __restrict__ gamma, int C, int Nk, int Nq, int k, float* __restrict__ ysel,
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
    const bool mx = gamma[c] >= 0.0f;       // the affine + LeakyReLU that follow are increasing in y iff gamma >= 0
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


// ---- GroupNorm statistics, affine + LeakyReLU, and the element-wise parts of the backward --------------------
// All tensors here are [B, C, Nq] (<= 8 MB): these kernels exist to replace ~35 tiny PyTorch launches per stage.

__device__ __forceinline__ double block_sum_double(double v, double* red) {
  // fixed-order tree over 256 threads
  red[threadIdx.x] = v;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// one workgroup per (b, group): mean / rstd of all edge values from the per-workgroup partials of edge_select
__global__ __launch_bounds__(256) void edge_stats_kernel(const float* __restrict__ partial, int C, int G, int nblk,
                                                         double M, float eps, float* __restrict__ mean_rstd) {
  __shared__ double red[256];
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  const int cpg = C / G;
  const float* base = partial + ((size_t)b * C + (size_t)g * cpg) * nblk * 2;
  const int n = cpg * nblk;
  double s1 = 0.0, s2 = 0.0;
  for (int t = threadIdx.x; t < n; t += 256) {
    s1 += (double)base[2 * t];
    s2 += (double)base[2 * t + 1];
  }
  s1 = block_sum_double(s1, red);
  s2 = block_sum_double(s2, red);
  if (threadIdx.x == 0) {
    const double mu = s1 / M;
    double var = s2 / M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean_rstd[2 * blockIdx.x] = (float)mu;
    mean_rstd[2 * blockIdx.x + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

__device__ __forceinline__ float edge_z(float y, float mu, float rs, float ga, float be, float* yhat) {
  *yhat = (y - mu) * rs;
  return fmaf(*yhat, ga, be);
}

// out = LeakyReLU(gamma * (ysel - mean) * rstd + beta)
__global__ __launch_bounds__(256) void edge_affine_kernel(const float* __restrict__ ysel, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ mean_rstd, int C, int G, int Nq,
                                                          float slope, float* __restrict__ out) {
  const int bc = blockIdx.y, b = bc / C, c = bc % C;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= Nq) return;
  const int g = c / (C / G);
  const float mu = mean_rstd[2 * (b * G + g)], rs = mean_rstd[2 * (b * G + g) + 1];
  float yh;
  const float z = edge_z(ysel[(size_t)bc * Nq + q], mu, rs, gamma[c], beta[c], &yh);
  out[(size_t)bc * Nq + q] = z > 0.0f ? z : z * slope;
}

// per (b, c): sum_q dz and sum_q dz * yhat, dz = go * LeakyReLU'(z)
__global__ __launch_bounds__(256) void edge_bwd_reduce_kernel(const float* __restrict__ go, const float* __restrict__ ysel,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              const float* __restrict__ mean_rstd, int C, int G, int Nq,
                                                              float slope, float* __restrict__ sums) {
  __shared__ double red[256];
  const int bc = blockIdx.x, b = bc / C, c = bc % C;
  const int g = c / (C / G);
  const float mu = mean_rstd[2 * (b * G + g)], rs = mean_rstd[2 * (b * G + g) + 1];
  const float ga = gamma[c], be = beta[c];
  double s1 = 0.0, s2 = 0.0;
  for (int q = threadIdx.x; q < Nq; q += 256) {
    float yh;
    const float z = edge_z(ysel[(size_t)bc * Nq + q], mu, rs, ga, be, &yh);
    const float dz = go[(size_t)bc * Nq + q] * (z > 0.0f ? 1.0f : slope);
    s1 += (double)dz;
    s2 += (double)dz * (double)yh;
  }
  s1 = block_sum_double(s1, red);
  s2 = block_sum_double(s2, red);
  if (threadIdx.x == 0) {
    sums[2 * bc] = (float)s1;
    sums[2 * bc + 1] = (float)s2;
  }
}

// per (b, group): the two coefficients of dy_e = [selected] S + A + Bc * y_e
__global__ __launch_bounds__(64) void edge_bwd_group_kernel(const float* __restrict__ sums, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean_rstd, int total, int C,
                                                            int G, double M, float* __restrict__ ab) {
  const int bg = blockIdx.x * 64 + threadIdx.x;
  if (bg >= total) return;  // total = B * G
  const int b = bg / G, g = bg % G;
  const int cpg = C / G;
  double m1 = 0.0, m2 = 0.0;
  for (int t = 0; t < cpg; ++t) {
    const int c = g * cpg + t;
    m1 += (double)gamma[c] * (double)sums[2 * ((size_t)b * C + c)];
    m2 += (double)gamma[c] * (double)sums[2 * ((size_t)b * C + c) + 1];
  }
  m1 /= M;
  m2 /= M;
  const double mu = mean_rstd[2 * bg], rs = mean_rstd[2 * bg + 1];
  const double Bc = -(rs * rs) * m2;
  ab[2 * bg] = (float)(-rs * m1 - Bc * mu);
  ab[2 * bg + 1] = (float)Bc;
}

// S = rstd * gamma * dz (selected-edge term) and dR = S + k A + Bc ysum
__global__ __launch_bounds__(256) void edge_bwd_apply_kernel(const float* __restrict__ go, const float* __restrict__ ysel,
                                                             const float* __restrict__ ysum,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ mean_rstd,
                                                             const float* __restrict__ ab, int C, int G, int Nq, int k,
                                                             float slope, float* __restrict__ S, float* __restrict__ dR) {
  const int bc = blockIdx.y, b = bc / C, c = bc % C;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= Nq) return;
  const int g = c / (C / G);
  const float mu = mean_rstd[2 * (b * G + g)], rs = mean_rstd[2 * (b * G + g) + 1];
  const float A = ab[2 * (b * G + g)], Bc = ab[2 * (b * G + g) + 1];
  const size_t o = (size_t)bc * Nq + q;
  float yh;
  const float z = edge_z(ysel[o], mu, rs, gamma[c], beta[c], &yh);
  const float dz = go[o] * (z > 0.0f ? 1.0f : slope);
  const float s = rs * gamma[c] * dz;
  S[o] = s;
  dR[o] = s + (float)k * A + Bc * ysum[o];
}

// dP[c, n] = sum_{e -> n} (A + Bc y_e) + sum_{selected e -> n} S[c, q_e]
//          = cnt[n] (A + Bc P[c, n]) + Bc sum_{e -> n} R[c, q_e] + sum_{selected e -> n} S[c, q_e]
template <int CT>
__global__ __launch_bounds__(kTPB) void edge_scatter_dp_kernel(
    const float* __restrict__ P, const float* __restrict__ R, const float* __restrict__ S,
    const int32_t* __restrict__ idx, const int32_t* __restrict__ nsel, const float* __restrict__ ab, int C, int G,
    int Nk, int Nq, int k, float* __restrict__ dP) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [CT + 1][Nk]: CT channels + the in-degree row
  const int b = blockIdx.y, c0 = blockIdx.x * CT;
  for (int t = threadIdx.x; t < (CT + 1) * Nk; t += kTPB) tile[t] = 0.0f;
  __syncthreads();
  const int32_t* ib = idx + (size_t)b * Nq * k;
  float* cnt = tile + CT * Nk;
  const int cpg = C / G;
  for (int q = threadIdx.x; q < Nq; q += kTPB) {
    int id[kMaxK];
#pragma unroll
    for (int j = 0; j < kMaxK; ++j) id[j] = j < k ? ib[(size_t)q * k + j] : 0;
#pragma unroll
    for (int j = 0; j < kMaxK; ++j)
      if (j < k) atomicAdd(&cnt[id[j]], 1.0f);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int c = c0 + ct;
      if (c >= C) break;
      const size_t o = ((size_t)b * C + c) * Nq + q;
      const float dv = ab[2 * (b * G + c / cpg) + 1] * R[o];
#pragma unroll
      for (int j = 0; j < kMaxK; ++j)
        if (j < k) atomicAdd(&tile[ct * Nk + id[j]], dv);
      atomicAdd(&tile[ct * Nk + nsel[o]], S[o]);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < CT * Nk; t += kTPB) {
    const int ct = t / Nk, n = t - ct * Nk;
    const int c = c0 + ct;
    if (c < C) {
      const float A = ab[2 * (b * G + c / cpg)], Bc = ab[2 * (b * G + c / cpg) + 1];
      const size_t o = ((size_t)b * C + c) * Nk + n;
      dP[o] = tile[t] + cnt[n] * (A + Bc * P[o]);
    }
  }
}

// ---- the same dP without the k float atomics per (channel, query): transpose the neighbour graph once ----------
// edge_csr_kernel: for every key point n the queries q that list it (ascending), as rowptr [B][Nk+1] / col [B][Nq*k]
// (uint16).  One workgroup per cloud; counting sort in LDS with integer atomics, rows sorted afterwards so that the
// summation order of the gather - and with it dP - does not depend on atomic arrival order.
constexpr int kCsrT = 1024;
__global__ __launch_bounds__(kCsrT) void edge_csr_kernel(const int32_t* __restrict__ idx, int Nk, int Nq, int k,
                                                         int32_t* __restrict__ rowptr, uint16_t* __restrict__ col) {
  extern __shared__ __attribute__((aligned(16))) int csm[];  // cnt[Nk] | cur[Nk] | part[kCsrT] | col16[Nq*k] (uint16)
  int* cnt = csm;
  int* cur = cnt + Nk;
  int* part = cur + Nk;
  uint16_t* lcol = reinterpret_cast<uint16_t*>(part + kCsrT);
  const int b = blockIdx.x, tid = threadIdx.x;
  const int E = Nq * k;
  const int32_t* ib = idx + (size_t)b * E;
  for (int n = tid; n < Nk; n += kCsrT) cnt[n] = 0;
  __syncthreads();
  for (int e = tid; e < E; e += kCsrT) atomicAdd(&cnt[ib[e]], 1);
  __syncthreads();
  // exclusive scan of cnt: thread t owns the consecutive items [t*per, (t+1)*per)
  const int per = (Nk + kCsrT - 1) / kCsrT;
  int local = 0;
  for (int i = 0; i < per; ++i) {
    const int n = tid * per + i;
    if (n < Nk) local += cnt[n];
  }
  part[tid] = local;
  __syncthreads();
  for (int off = 1; off < kCsrT; off <<= 1) {  // Hillis-Steele inclusive scan of the per-thread totals
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - local;
  int32_t* rp = rowptr + (size_t)b * (Nk + 1);
  for (int i = 0; i < per; ++i) {
    const int n = tid * per + i;
    if (n < Nk) {
      cur[n] = run;
      rp[n] = run;
      run += cnt[n];
    }
  }
  if (tid == kCsrT - 1) rp[Nk] = part[tid];
  __syncthreads();
  for (int e = tid; e < E; e += kCsrT) {
    const int pos = atomicAdd(&cur[ib[e]], 1);
    lcol[pos] = (uint16_t)(e / k);
  }
  __syncthreads();
  for (int n = tid; n < Nk; n += kCsrT) {  // insertion sort of each (short) row
    const int end = cur[n], beg = end - cnt[n];
    for (int i = beg + 1; i < end; ++i) {
      const uint16_t v = lcol[i];
      int j = i - 1;
      while (j >= beg && lcol[j] > v) {
        lcol[j + 1] = lcol[j];
        --j;
      }
      lcol[j + 1] = v;
    }
  }
  __syncthreads();
  uint16_t* gc = col + (size_t)b * E;
  for (int e = tid; e < E; e += kCsrT) gc[e] = lcol[e];
}

// dP[c, n] = cnt[n] (A + Bc P[c, n]) + Bc sum_{q in row n} R[c, q] + sum_{q : nsel[c, q] = n} S[c, q]
// Both sums are gathers over the transposed graph from LDS copies of the per-query values: the selected source of a
// query is one of its k (distinct) neighbours, so every q with nsel[c, q] = n is in row n.  No atomics: the summation
// order is the row order (ascending q) and dP is bit-identical from run to run (round 3; the selected-edge term used
// to go through LDS float atomics).
template <int CT>
__global__ __launch_bounds__(kTPB) void edge_gather_dp_kernel(
    const float* __restrict__ P, const float* __restrict__ R, const float* __restrict__ S,
    const int32_t* __restrict__ rowptr, const uint16_t* __restrict__ col, const int32_t* __restrict__ nsel,
    const float* __restrict__ ab, int C, int G, int Nk, int Nq, int k, float* __restrict__ dP) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // rt[CT][Nq] | st[CT][Nq] | sn[CT][Nq] (uint16)
  float* rt = tile;
  float* st = tile + (size_t)CT * Nq;
  uint16_t* sn = reinterpret_cast<uint16_t*>(st + (size_t)CT * Nq);
  const int b = blockIdx.y, c0 = blockIdx.x * CT;
  const int cpg = C / G;
  for (int t = threadIdx.x; t < CT * Nq; t += kTPB) {
    const int ct = t / Nq, q = t - ct * Nq, c = c0 + ct;
    const size_t o = ((size_t)b * C + c) * Nq + q;
    rt[t] = c < C ? ab[2 * (b * G + c / cpg) + 1] * R[o] : 0.0f;
    st[t] = c < C ? S[o] : 0.0f;
    sn[t] = c < C ? (uint16_t)nsel[o] : (uint16_t)0xFFFF;
  }
  __syncthreads();
  const int32_t* rp = rowptr + (size_t)b * (Nk + 1);
  const uint16_t* gc = col + (size_t)b * Nq * k;
  for (int n = threadIdx.x; n < Nk; n += kTPB) {
    const int beg = rp[n], end = rp[n + 1];
    float acc[CT], sel[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = 0.0f, sel[ct] = 0.0f;
    int prev = -1;
    for (int e = beg; e < end; ++e) {
      const int q = gc[e];
      // a query that lists this source point more than once (never a kNN graph, but legal input) contributes one edge
      // per listing to the dense term and its selected value ONCE; the row is sorted, so repeats are adjacent
      const bool first = q != prev;
      prev = q;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        acc[ct] += rt[ct * Nq + q];
        sel[ct] += (first && sn[ct * Nq + q] == (uint16_t)n) ? st[ct * Nq + q] : 0.0f;
      }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int c = c0 + ct;
      if (c < C) {
        const float A = ab[2 * (b * G + c / cpg)], Bc = ab[2 * (b * G + c / cpg) + 1];
        const size_t o = ((size_t)b * C + c) * Nk + n;
        dP[o] = (sel[ct] + acc[ct]) + (float)(end - beg) * (A + Bc * P[o]);
      }
    }
  }
}

// LDS bytes of the two kernels; the CSR path is used when both fit (else the atomic scatter above)
size_t csr_lds_bytes(int Nk, int Nq, int k) { return sizeof(int) * (2 * (size_t)Nk + kCsrT) + sizeof(uint16_t) * (size_t)Nq * k + 16; }
// bytes of LDS per channel of edge_gather_dp_kernel: two floats and one uint16 per query
inline size_t dp_lds_per_channel(int Nq) { return (2 * sizeof(float) + sizeof(uint16_t)) * (size_t)Nq; }
bool csr_eligible(int Nk, int Nq, int k) {
  return Nq <= 65535 && Nk <= 65535 && csr_lds_bytes(Nk, Nq, k) <= 150 * 1024 && dp_lds_per_channel(Nq) <= 96 * 1024;
}
size_t csr_workspace_bytes(int B, int Nk, int Nq, int k) {
  const size_t rowptr = (sizeof(int32_t) * (size_t)B * (Nk + 1) + 15) / 16 * 16;
  return rowptr + (sizeof(uint16_t) * (size_t)B * Nq * k + 15) / 16 * 16;
}

}  // namespace

PTOPS_API int ptops_edge_select_blocks(int Nq) { return ptops::ceil_div(Nq, kTPB); }

PTOPS_API int ptops_edge_select(const float* P, const float* R, const int32_t* idx, const float* gamma,
                                int B, int C, int Nk, int Nq, int k, float* ysel, int32_t* nsel, float* ysum,
                                float* partial, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && C >= 1 && Nk >= 1 && Nq >= 1 && k >= 1 && k <= kMaxK, "edge_select: bad size (k <= %d)", kMaxK);
  if (B == 0) return PTOPS_OK;
  PTOPS_REQUIRE(P && R && idx && gamma && ysel && nsel && ysum && partial, "edge_select: null pointer");
  PTOPS_REQUIRE(B <= 65535, "edge_select: B exceeds grid limit");
  dim3 grid(ptops::ceil_div(Nq, kTPB), ptops::ceil_div(C, kCT), B);
  hipLaunchKernelGGL(edge_select_kernel, grid, dim3(kTPB), 0, stream, P, R, idx, gamma, C, Nk, Nq, k, ysel, nsel,
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

PTOPS_API int ptops_edge_norm_forward(const float* ysel, const float* partial, const float* gamma, const float* beta,
                                      int B, int C, int Nq, int nblk, int groups, int k, float eps, float slope,
                                      float* out, float* mean_rstd, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && C >= 1 && Nq >= 1 && nblk >= 1 && groups >= 1 && C % groups == 0 && k >= 1,
                "edge_norm_forward: bad size");
  if (B == 0) return PTOPS_OK;
  PTOPS_REQUIRE(ysel && partial && gamma && beta && out && mean_rstd, "edge_norm_forward: null pointer");
  PTOPS_REQUIRE((long long)B * C <= 65535, "edge_norm_forward: B*C exceeds grid limit");
  const double M = (double)(C / groups) * Nq * k;
  hipLaunchKernelGGL(edge_stats_kernel, dim3(B * groups), dim3(256), 0, stream, partial, C, groups, nblk, M, eps,
                     mean_rstd);
  hipLaunchKernelGGL(edge_affine_kernel, dim3(ptops::ceil_div(Nq, 256), B * C), dim3(256), 0, stream, ysel, gamma, beta,
                     mean_rstd, C, groups, Nq, slope, out);
  return ptops::check_launch("edge_norm_forward");
}

static size_t edge_bwd_float_ws(int B, int C, int Nq, int groups) {
  return ((size_t)B * C * Nq + 2 * (size_t)B * groups + 4 + 3) / 4 * 4;   // floats, 16-byte multiple
}

PTOPS_API size_t ptops_edge_norm_backward_workspace_bytes(int B, int C, int Nk, int Nq, int k, int groups) {
  if (B <= 0 || C <= 0 || Nk <= 0 || Nq <= 0 || k <= 0 || groups <= 0) return 0;
  return sizeof(float) * edge_bwd_float_ws(B, C, Nq, groups) + (csr_eligible(Nk, Nq, k) ? csr_workspace_bytes(B, Nk, Nq, k) : 0);
}

PTOPS_API int ptops_edge_norm_backward(const float* grad_out, const float* ysel, const float* ysum, const float* P,
                                       const float* R, const int32_t* idx, const int32_t* nsel, const float* gamma,
                                       const float* beta, const float* mean_rstd, int B, int C, int Nk, int Nq, int k,
                                       int groups, float slope, float* dP, float* dR, float* sums, void* workspace,
                                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && C >= 1 && Nk >= 1 && Nq >= 1 && k >= 1 && k <= kMaxK && groups >= 1 && C % groups == 0,
                "edge_norm_backward: bad size (k <= %d)", kMaxK);
  if (B == 0) return PTOPS_OK;
  PTOPS_REQUIRE(grad_out && ysel && ysum && P && R && idx && nsel && gamma && beta && mean_rstd && dP && dR && sums &&
                    workspace,
                "edge_norm_backward: null pointer");
  PTOPS_REQUIRE((long long)B * C <= 65535, "edge_norm_backward: B*C exceeds grid limit");
  PTOPS_REQUIRE((size_t)2 * Nk * sizeof(float) <= 64 * 1024, "edge_norm_backward: Nk=%d does not fit an LDS tile", Nk);
  float* S = (float*)workspace;
  float* ab = S + (size_t)B * C * Nq;
  const double M = (double)(C / groups) * Nq * k;
  hipLaunchKernelGGL(edge_bwd_reduce_kernel, dim3(B * C), dim3(256), 0, stream, grad_out, ysel, gamma, beta, mean_rstd, C,
                     groups, Nq, slope, sums);
  hipLaunchKernelGGL(edge_bwd_group_kernel, dim3(ptops::ceil_div(B * groups, 64)), dim3(64), 0, stream, sums, gamma,
                     mean_rstd, B * groups, C, groups, M, ab);
  hipLaunchKernelGGL(edge_bwd_apply_kernel, dim3(ptops::ceil_div(Nq, 256), B * C), dim3(256), 0, stream, grad_out, ysel,
                     ysum, gamma, beta, mean_rstd, ab, C, groups, Nq, k, slope, S, dR);
  if (csr_eligible(Nk, Nq, k)) {
    int32_t* rowptr = reinterpret_cast<int32_t*>((float*)workspace + edge_bwd_float_ws(B, C, Nq, groups));
    uint16_t* col = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(rowptr) +
                                                (sizeof(int32_t) * (size_t)B * (Nk + 1) + 15) / 16 * 16);
    const size_t csr_lds = csr_lds_bytes(Nk, Nq, k);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(edge_csr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)csr_lds);
    hipLaunchKernelGGL(edge_csr_kernel, dim3(B), dim3(kCsrT), csr_lds, stream, idx, Nk, Nq, k, rowptr, col);
    int CT = 8;
    while (CT > 1 && ((long long)B * ptops::ceil_div(C, CT) < 512 || CT * dp_lds_per_channel(Nq) > 96 * 1024)) CT >>= 1;
    dim3 grid(ptops::ceil_div(C, CT), B);
    const size_t lds = (CT * dp_lds_per_channel(Nq) + 15) / 16 * 16;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(edge_gather_dp_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(edge_gather_dp_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(edge_gather_dp_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(edge_gather_dp_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    switch (CT) {
      case 8: hipLaunchKernelGGL(edge_gather_dp_kernel<8>, grid, dim3(kTPB), lds, stream, P, R, S, rowptr, col, nsel, ab, C, groups, Nk, Nq, k, dP); break;
      case 4: hipLaunchKernelGGL(edge_gather_dp_kernel<4>, grid, dim3(kTPB), lds, stream, P, R, S, rowptr, col, nsel, ab, C, groups, Nk, Nq, k, dP); break;
      case 2: hipLaunchKernelGGL(edge_gather_dp_kernel<2>, grid, dim3(kTPB), lds, stream, P, R, S, rowptr, col, nsel, ab, C, groups, Nk, Nq, k, dP); break;
      default: hipLaunchKernelGGL(edge_gather_dp_kernel<1>, grid, dim3(kTPB), lds, stream, P, R, S, rowptr, col, nsel, ab, C, groups, Nk, Nq, k, dP); break;
    }
    return ptops::check_launch("edge_norm_backward");
  }
  int CT = 8;
  while (CT > 1 && ((long long)B * ptops::ceil_div(C, CT) < 512 || (size_t)(CT + 1) * Nk * 4 > 64 * 1024)) CT >>= 1;
  dim3 grid(ptops::ceil_div(C, CT), B);
  const size_t lds = (size_t)(CT + 1) * Nk * sizeof(float);
  switch (CT) {
    case 8: hipLaunchKernelGGL(edge_scatter_dp_kernel<8>, grid, dim3(kTPB), lds, stream, P, R, S, idx, nsel, ab, C, groups, Nk, Nq, k, dP); break;
    case 4: hipLaunchKernelGGL(edge_scatter_dp_kernel<4>, grid, dim3(kTPB), lds, stream, P, R, S, idx, nsel, ab, C, groups, Nk, Nq, k, dP); break;
    case 2: hipLaunchKernelGGL(edge_scatter_dp_kernel<2>, grid, dim3(kTPB), lds, stream, P, R, S, idx, nsel, ab, C, groups, Nk, Nq, k, dP); break;
    default: hipLaunchKernelGGL(edge_scatter_dp_kernel<1>, grid, dim3(kTPB), lds, stream, P, R, S, idx, nsel, ab, C, groups, Nk, Nq, k, dP); break;
  }
  return ptops::check_launch("edge_norm_backward");
}
