/*
 * ptops.h — C ABI of libptops, the MI355X (gfx950) point-cloud primitive library
 * that sits under the AdaPoinTr completion hot path.
 *
 * Every entry point takes raw DEVICE pointers, explicit sizes and an explicit
 * hipStream_t (passed as void* so that this header needs no HIP include), does no
 * hidden allocation, launches on the given stream and returns an int status
 * (PTOPS_OK or a negative PTOPS_E* code).  Scratch memory is caller-provided; the
 * *_workspace_bytes helpers say how much.  The library keeps no state between calls
 * with one documented exception: the per-stream queues of deferred folds
 * (ptops_fold_defer / ptops_fold_flush below), which exist only while a caller has
 * switched deferral on for a stream.
 *
 * Each function names the reference interface it replaces (paths relative to the
 * reference repository mmkuznecov/ImgAdaPoinTr).
 *
 * All point data is fp32, all indices int32, all tensors dense row-major.
 */
#ifndef PTOPS_H_
#define PTOPS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTOPS_OK 0
#define PTOPS_EINVAL (-1)   /* bad size / null pointer / unsupported shape */
#define PTOPS_ELAUNCH (-2)  /* the HIP runtime reported a launch error      */
#define PTOPS_EALIGN (-3)   /* a pointer does not meet the stated alignment  */

/* Library identification; ptops_last_error() returns a thread-local message for
 * the most recent non-OK status on the calling thread. */
int ptops_abi_version(void);
const char* ptops_last_error(void);

/* ---------------------------------------------------------------- Chamfer --
 * Replaces chamfer.forward  (extensions/chamfer_dist/chamfer_cuda.cpp:22-25,
 * kernel extensions/chamfer_dist/chamfer.cu:15-171).
 *   xyz1 [B,n,3], xyz2 [B,m,3]  ->  dist1 [B,n], idx1 [B,n] (nearest point of
 *   xyz2 for every point of xyz1: squared distance, index), dist2/idx2 [B,m] the
 *   other way round.  Squared distance = fma(dz,dz, fma(dy,dy, dx*dx)) with
 *   d* = candidate - query; ties resolve to the lowest candidate index.
 * workspace: ptops_chamfer_forward_workspace_bytes(B,n,m) bytes, 8-byte aligned
 *   (may be NULL when that function returns 0).                               */
size_t ptops_chamfer_forward_workspace_bytes(int B, int n, int m);
int ptops_chamfer_forward(const float* xyz1, const float* xyz2, int B, int n, int m,
                          float* dist1, float* dist2, int32_t* idx1, int32_t* idx2,
                          void* workspace, void* stream);

/* Same operator with an explicit algorithm per direction (algorithm1: xyz1 -> xyz2, dist1/idx1;
 * algorithm2: xyz2 -> xyz1).  All algorithms return bit-identical results:
 *   PTOPS_CHAMFER_BRUTE  every pair is evaluated (VALU-issue bound);
 *   PTOPS_CHAMFER_SWEEP  clouds are sorted by x and every query tile sweeps outwards until no farther
 *                        candidate can win or tie (exact pruning; needs ptops_chamfer_sweep_available);
 *   PTOPS_CHAMFER_MFMA   the matrix pipe searches, the vector pipe only keeps minima: coordinates are split into three
 *                        bf16 pieces and two v_mfma_f32_32x32x16_bf16 evaluate |c|^2 - 2 c.q for 32 x 32 pairs as a
 *                        FILTER with a proven error bound; per query only the one or two 128-candidate superblocks that
 *                        can hold the nearest neighbour are re-evaluated with the exact direct-difference arithmetic
 *                        (every pair is seen; about twice the speed of brute force on large clouds whatever the
 *                        geometry; masses of exactly tied distances make it slower);
 *   PTOPS_CHAMFER_AUTO   the sweep whenever it is available, else the MFMA filter for directions with at least
 *                        PTOPS_CHAMFER_MFMA_MIN_PAIRS pair evaluations (B * queries * candidates), else brute force.
 * stats (optional, 4 x u64, device memory, caller-zeroed): a swept direction d adds {16-candidate chunks
 * evaluated, chunks a brute force would evaluate} to stats[2d], stats[2d+1]; an MFMA-filtered direction adds
 * {128-candidate superblocks re-evaluated exactly beyond the one every query needs, superblocks visited} (per lane), so
 * that a caller can steer the choice for the following calls. */
#define PTOPS_CHAMFER_AUTO 0
#define PTOPS_CHAMFER_BRUTE 1
#define PTOPS_CHAMFER_SWEEP 2
#define PTOPS_CHAMFER_MFMA 3
#define PTOPS_CHAMFER_MFMA_MIN_PAIRS (1LL << 32)
int ptops_chamfer_sweep_available(int B, int n, int m);
int ptops_chamfer_forward_ex(const float* xyz1, const float* xyz2, int B, int n, int m,
                             float* dist1, float* dist2, int32_t* idx1, int32_t* idx2,
                             void* workspace, int algorithm1, int algorithm2,
                             unsigned long long* stats, void* stream);

/* Replaces chamfer.backward (extensions/chamfer_dist/chamfer_cuda.cpp:27-34,
 * kernel extensions/chamfer_dist/chamfer.cu:173-229).
 *   grad_xyz1 [B,n,3], grad_xyz2 [B,m,3] are fully overwritten (zeroed first).  */
int ptops_chamfer_backward(const float* xyz1, const float* xyz2,
                           const int32_t* idx1, const int32_t* idx2,
                           const float* grad_dist1, const float* grad_dist2,
                           int B, int n, int m,
                           float* grad_xyz1, float* grad_xyz2, void* stream);

/* The same gradients with a FIXED summation order (that of a scalar loop over cloud 1, then cloud 2, each adding its
 * own term and the negated term of its neighbour) instead of the reference's atomicAdd arrival order (chamfer.cu:193-198):
 * bit-identical from run to run, no atomics, no zero-fill.  The nearest-neighbour map is transposed once per call in
 * LDS, hence max(n, m) <= 16384; workspace of ptops_chamfer_backward_det_workspace_bytes(B, n, m) bytes, 16-byte
 * aligned.  Values agree with ptops_chamfer_backward to the rounding of that reordering. */
size_t ptops_chamfer_backward_det_workspace_bytes(int B, int n, int m);
int ptops_chamfer_backward_det(const float* xyz1, const float* xyz2,
                               const int32_t* idx1, const int32_t* idx2,
                               const float* grad_dist1, const float* grad_dist2,
                               int B, int n, int m, float* grad_xyz1, float* grad_xyz2,
                               void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------- farthest point sampling --
 * Replaces pointnet2_ops.pointnet2_utils.furthest_point_sample (third-party,
 * un-vendored; call sites models/base_blocks.py:550, utils/misc.py:37).
 *   xyz [B,N,3] -> idx [B,npoint]; starts at index 0, running min-distance
 *   initialised to 1e10, candidates with |p|^2 <= 1e-3 never selected, arg-max
 *   ties resolve to the lowest index.  Because the selection order is
 *   prefix-consistent, idx[:, :p] is the answer for npoint = p.               */
int ptops_furthest_point_sample(const float* xyz, int B, int N, int npoint,
                                int32_t* idx, void* stream);

/* Replaces pointnet2_utils.gather_operation forward / backward
 * (call sites models/base_blocks.py:555, utils/misc.py:38).
 *   out[b,c,j] = feat[b,c,idx[b,j]];  feat [B,C,N], idx [B,npoint].
 *   backward: grad_feat [B,C,N] is zeroed, then scatter-added.                */
int ptops_gather_forward(const float* feat, const int32_t* idx, int B, int C, int N,
                         int npoint, float* out, void* stream);
int ptops_gather_backward(const float* grad_out, const int32_t* idx, int B, int C,
                          int N, int npoint, float* grad_feat, void* stream);

/* --------------------------------------------------------------------- kNN --
 * Replaces knn_point + square_distance (models/Transformer_utils.py:19-51)
 * without materialising the [B,S,N] matrix.
 *   xyz [B,N,3] (candidates), new_xyz [B,S,3] (queries) -> idx [B,S,k]:
 *   the k candidates with the smallest expansion-form squared distance
 *   ((-2*(q.c) + |q|^2) + |c|^2, q.c = fma(qz,cz, fma(qy,cy, qx*cx))),
 *   ordered by (distance, index) ascending.  Requires 1 <= k <= min(N, 64).   */
int ptops_knn(const float* xyz, const float* new_xyz, int B, int N, int S, int k,
              int32_t* idx, void* stream);

/* Replaces the selection step of GDANet's feature-space kNN,
 * `pairwise_distance.topk(k=k, dim=-1)[1]` (models/segmentation/GDANet_util.py:5-11;
 * callers local_operator :14, local_operator_withnorm :40, GDM :73).
 *   x [R,N] row-major -> idx [R,k] (int64): the k largest entries of each row,
 *   ordered by (value descending, column ascending); -0 equals +0; NaN is not
 *   supported.  Requires 1 <= k <= N <= 2048.                                   */
int ptops_topk_rows(const float* x, long long R, int N, int k, long long* idx,
                    void* stream);

/* Replaces index_points (models/Transformer_utils.py:53-69) for idx [B,S,k]:
 *   out[b,s,j,:] = points[b, idx[b,s,j], :];  points [B,N,C] -> out [B,S,k,C].
 *   backward: grad_points [B,N,C] zeroed then scatter-added.                   */
int ptops_index_points_forward(const float* points, const int32_t* idx, int B, int N,
                               int C, int S, int k, float* out, void* stream);
int ptops_index_points_backward(const float* grad_out, const int32_t* idx, int B,
                                int N, int C, int S, int k, float* grad_points,
                                void* stream);

/* --------------------------------------------------- DGCNN edge grouping ---
 * Replaces the gather+concat half of DGCNN_Grouper.get_graph_feature
 * (models/base_blocks.py:565-588).
 *   x_k [B,C,Nk], x_q [B,C,Nq], idx [B,Nq,k] (neighbour ids into Nk) ->
 *   out [B,2C,Nq,k]: channels [0,C) = x_k[:, c, idx] - x_q, [C,2C) = x_q.
 *   backward: grad_xk [B,C,Nk] (zeroed, scatter-added), grad_xq [B,C,Nq].      */
int ptops_edge_group_forward(const float* x_k, const float* x_q, const int32_t* idx,
                             int B, int C, int Nk, int Nq, int k, float* out,
                             void* stream);
int ptops_edge_group_backward(const float* grad_out, const int32_t* idx, int B, int C,
                              int Nk, int Nq, int k, float* grad_xk, float* grad_xq,
                              void* stream);

/* ----------------------------------------------- graph gather-max (fused) ---
 * The neighbour half of DynamicGraphAttention (models/Transformer_utils.py:
 * 811-817) after the exact split W.cat(v_j - q, q) = W1.v_j + (W2 - W1).q:
 *   p [B,N,C] (= v @ W1^T), idx [B,S,k] -> out[b,s,c] = max_j p[b,idx[b,s,j],c],
 *   arg[b,s,c] = the winning row id (first maximum).  backward routes grad_out
 *   to grad_p[b,arg,c] (grad_p zeroed then scatter-added).                     */
int ptops_gather_max_forward(const float* p, const int32_t* idx, int B, int N, int C,
                             int S, int k, float* out, int32_t* arg, void* stream);
int ptops_gather_max_backward(const float* grad_out, const int32_t* arg, int B, int N,
                              int C, int S, float* grad_p, void* stream);

/* The same aggregation with the query half and the activation of DynamicGraphAttention folded in:
 * out = LeakyReLU(max_j p[b, idx[b,s,j], :] + add[b,s,:], slope)  (Transformer_utils.py:811-817 after the linear split:
 * the `+ q-term` and nn.LeakyReLU(0.2) that follow the max).  Backward: g' = out > 0 ? g : g * slope (aten's
 * leaky_relu_backward) goes to grad_add [B,S,C] and, scattered by arg, to grad_p [B,N,C] (deterministic; N <= 1024). */
int ptops_gather_max_act_forward(const float* p, const int32_t* idx, const float* add, float slope, int B, int N, int C,
                                 int S, int k, float* out, int32_t* arg, void* stream);
int ptops_gather_max_act_backward(const float* grad_out, const float* out, const int32_t* arg, float slope, int B, int N,
                                  int C, int S, float* grad_p, float* grad_add, void* stream);

/* --------------------------------------------------------------------- EMD --
 * Replaces emd.forward / emd.backward (extensions/emd/emd.cpp:6-31, kernels
 * extensions/emd/emd_cuda.cu:23-316).  Same caller-allocates contract as the
 * reference (extensions/emd/emd_module.py:45-56): every output and scratch array is
 * passed in — dist [B,n] f32, assignment [B,n] i32 pre-filled with -1, price [B,m] f32
 * zeros, assignment_inv [B,m] i32 pre-filled with -1, bid [B,n] i32, bid_increments
 * [B,n] f32, max_increments [B,m] f32 zeros, unass_idx [B*n] i32, unass_cnt /
 * unass_cnt_sum / cnt_tmp [512] i32, max_idx [B*m] i32.  Requires B <= 512, n == m,
 * n % 1024 == 0 (emd_cuda.cu:236-249; violations return PTOPS_EINVAL).  Ties, which
 * the CUDA code resolves by race, resolve by lowest index here (csrc/emd.hip).
 * backward accumulates 2*graddist*(xyz1 - xyz2[idx]) INTO gradxyz (caller zeroes it). */
int ptops_emd_forward(const float* xyz1, const float* xyz2, int B, int n, int m,
                      float* dist, int32_t* assignment, float* price,
                      int32_t* assignment_inv, int32_t* bid, float* bid_increments,
                      float* max_increments, int32_t* unass_idx, int32_t* unass_cnt,
                      int32_t* unass_cnt_sum, int32_t* cnt_tmp, int32_t* max_idx,
                      float eps, int iters, void* stream);
int ptops_emd_backward(const float* xyz1, const float* xyz2, float* gradxyz,
                       const float* graddist, const int32_t* idx, int B, int n, int m,
                       void* stream);

/* --------------------------------------------------------------- attention --
 * softmax(q k^T * scale [+ block mask]) v of Attention / CrossAttention
 * (models/Transformer_utils.py:89-159), fp32, head_dim 64, on the f32 MFMA path.
 *   q [B,H,Nq,64], k/v [B,H,Nk,64] given by base pointer + element strides
 *   {batch, head, token} (innermost dim contiguous), Nq and Nk multiples of 32.
 *   Mask (base_blocks.py:224-229): queries < split_q attend to keys < limit_k, the
 *   rest to all keys; split_q = 0, limit_k = Nk means no mask; both multiples of 32.
 *   out [B,Nq,H,64] contiguous; lse [B,H,Nq] (base-2 log-sum-exp, kept for backward). */
int ptops_attention_forward(const float* q, const float* k, const float* v, int B, int H,
                            int Nq, int Nk, const long long* q_strides,
                            const long long* k_strides, const long long* v_strides,
                            int split_q, int limit_k, float scale, float* out, float* lse,
                            void* stream);

/* Backward of ptops_attention_forward (replaces the autograd chain through softmax / matmul of
 * models/Transformer_utils.py:89-159).  out, dout [B,Nq,H,64] contiguous; lse as returned by the
 * forward; dq/dk/dv are written through their own (batch, head, token) strides (rows of 64 contiguous
 * floats, e.g. the three slices of a packed [B,N,3,H,64] gradient); delta [B,H,Nq] is scratch
 * (sum_c dout*out per query).  Deterministic: no atomics. */
int ptops_attention_backward(const float* q, const float* k, const float* v, const float* out,
                             const float* dout, const float* lse, int B, int H, int Nq, int Nk,
                             const long long* q_strides, const long long* k_strides,
                             const long long* v_strides, const long long* dq_strides,
                             const long long* dk_strides, const long long* dv_strides,
                             int split_q, int limit_k, float scale, float* dq, float* dk,
                             float* dv, float* delta, void* stream);

/* Single-pass variant: dk, dv and dq from one evaluation of S and dP (5 GEMMs instead of 7).  Eight
 * stationary key tiles per workgroup; the dq tile is formed from all of them through LDS (no atomics) and,
 * when Nk > 256, folded across 256-key blocks through `workspace`
 * (ptops_attention_backward_fused_workspace_bytes bytes, 16-byte aligned) in a fixed order.  Same
 * arguments and results as ptops_attention_backward (deterministic as well); no delta buffer. */
size_t ptops_attention_backward_fused_workspace_bytes(int B, int H, int Nq, int Nk);
int ptops_attention_backward_fused(const float* q, const float* k, const float* v, const float* out,
                                   const float* dout, const float* lse, int B, int H, int Nq, int Nk,
                                   const long long* q_strides, const long long* k_strides,
                                   const long long* v_strides, const long long* dq_strides,
                                   const long long* dk_strides, const long long* dv_strides,
                                   int split_q, int limit_k, float scale, float* dq, float* dk,
                                   float* dv, void* workspace, void* stream);

/* ---------------------------------------------------------------- LayerNorm --
 * nn.LayerNorm over the last dimension (norm1 / norm2 / norm_q / norm_v of every transformer block,
 * models/base_blocks.py:88-118, 154-207) on row-major fp32 x [rows, C], C % 4 == 0, C <= 1024.
 * forward : y = (x - mean) * rstd * gamma + beta; mean, rstd [rows] are kept for the backward.
 * add_layer_norm_forward: the same on x + res, which is also written to sum_out (the residual add
 *           `x = x + f(...)` of a block fused into the norm that follows it).
 * backward: dx [rows, C] (+ extra, if given: the gradient reaching the normalised tensor through its other
 *           consumers) and dgamma_dbeta [3*C] = dgamma | dbeta | column sums of the dx written (the bias
 *           gradient of a Linear whose output was the `res` of add_layer_norm_forward) in one pass over x
 *           and dy plus a deterministic fold; workspace of
 *           ptops_layer_norm_backward_workspace_bytes(rows, C) bytes, 16-byte aligned. */
int ptops_layer_norm_max_dim(void);
int ptops_layer_norm_forward(const float* x, const float* gamma, const float* beta, long long rows,
                             int C, float eps, float* y, float* mean, float* rstd, void* stream);
int ptops_add_layer_norm_forward(const float* x, const float* res, const float* gamma,
                                 const float* beta, long long rows, int C, float eps, float* sum_out,
                                 float* y, float* mean, float* rstd, void* stream);
size_t ptops_layer_norm_backward_workspace_bytes(long long rows, int C);
int ptops_layer_norm_backward(const float* dy, const float* x, const float* mean, const float* rstd,
                              const float* gamma, const float* extra, long long rows, int C, float* dx,
                              float* dgamma_dbeta, void* workspace, void* stream);

/* ------------------------------------------- BatchNorm over rows + LeakyReLU --
 * Training-mode nn.BatchNorm1d followed by LeakyReLU(slope) for token-major data: x [M, N] row-major
 * (N % 4 == 0), statistics per COLUMN over the M rows (what BatchNorm1d computes over (B, L) of a
 * [B, N, L] tensor) — AdaPoinTr.increase_dim, models/AdaPoinTr.py:222-224.  slope = 1 disables the
 * activation.  forward: y, stats [2N] = {mean, rstd} (kept for the backward), var_out [N] = biased
 * batch variance (the caller updates the running statistics).  backward: dx and dbeta_dgamma [2N] =
 * {dbeta, dgamma}.  workspace: ptops_batch_norm_rows_workspace_bytes(M, N) bytes, 16-byte aligned. */
size_t ptops_batch_norm_rows_workspace_bytes(int M, int N);
int ptops_batch_norm_rows_forward(const float* x, const float* gamma, const float* beta, int M, int N,
                                  float eps, float slope, float* y, float* stats, float* var_out,
                                  void* workspace, void* stream);
int ptops_batch_norm_rows_backward(const float* dy, const float* x, const float* stats,
                                   const float* gamma, const float* beta, int M, int N, float slope,
                                   float* dx, float* dbeta_dgamma, void* workspace, void* stream);

/* ------------------------------------------- GELU backward + column sums ----
 * dpre = dy * gelu'(x) (exact erf GELU, as nn.GELU()) for row-major x, dy [M, N] (N % 4 == 0), together with
 * colsum [N] = sum over rows of dpre — the bias gradient of the Linear that feeds the activation (Mlp.fc1,
 * models/Transformer_utils.py:58-75).  workspace: ptops_gelu_backward_colsum_workspace_bytes(M, N) bytes. */
size_t ptops_gelu_backward_colsum_workspace_bytes(int M, int N);
int ptops_gelu_backward_colsum(const float* x, const float* dy, int M, int N, float* dpre,
                               float* colsum, void* workspace, void* stream);

/* ------------------------------------------------- fused edge-conv stage ----
 * One stage of DGCNN_Grouper (models/base_blocks.py:565-623): with P = W1.x_k [B,C,Nk] and
 * R = (W2-W1).x_q [B,C,Nq] (the 1x1 convolution split over its two input halves) the value on edge
 * (q,j) is y = P[idx[q,j]] + R[q].  ptops_edge_select returns per (b,c,q) the max (gamma[c] >= 0, gamma [C] = the
 * GroupNorm scale: the affine + LeakyReLU after it are monotone in y) or min of y over the k neighbours (ysel), the source index that produced it (nsel), sum_j y (ysum) and
 * per-workgroup partial sums {sum y, sum y^2} (partial [B,C,ptops_edge_select_blocks(Nq),2]) for the
 * GroupNorm statistics.  ptops_edge_scatter is the edge part of the backward:
 * T[b,c,n] = sum_{(q,j): idx[q,j]=n} dense[b,c,q] + sum_{q: nsel[b,c,q]=n} sel[b,c,q].   k <= 32. */
int ptops_edge_select_blocks(int Nq);
int ptops_edge_select(const float* P, const float* R, const int32_t* idx,
                      const float* gamma, int B, int C, int Nk, int Nq, int k,
                      float* ysel, int32_t* nsel, float* ysum, float* partial, void* stream);
int ptops_edge_scatter(const float* dense, const float* sel, const int32_t* idx,
                       const int32_t* nsel, int B, int C, int Nk, int Nq, int k, float* T,
                       void* stream);

/* The rest of the stage, so that no [B,C,Nq] element-wise op is left to the caller:
 * ptops_edge_norm_forward : GroupNorm statistics over ALL edges from `partial` (float64, fixed order),
 *   mean_rstd [B,groups,2], and out = LeakyReLU(gamma * (ysel - mean) * rstd + beta)  [B,C,Nq].
 * ptops_edge_norm_backward: from grad_out [B,C,Nq] the gradients dP [B,C,Nk], dR [B,C,Nq] of the two
 *   conv halves and sums [B,C,2] = {sum_q dz, sum_q dz*yhat} (dbeta / dgamma are their sums over B);
 *   workspace of ptops_edge_norm_backward_workspace_bytes(B, C, Nk, Nq, k, groups) bytes (16-byte aligned).
 *   The neighbour graph is transposed once per call (counting sort, rows ordered) and dP gathered from it;
 *   only the selected-edge term (one value per channel and query) goes through LDS float adds. */
int ptops_edge_norm_forward(const float* ysel, const float* partial, const float* gamma,
                            const float* beta, int B, int C, int Nq, int nblk, int groups, int k,
                            float eps, float slope, float* out, float* mean_rstd, void* stream);
size_t ptops_edge_norm_backward_workspace_bytes(int B, int C, int Nk, int Nq, int k, int groups);
int ptops_edge_norm_backward(const float* grad_out, const float* ysel, const float* ysum,
                             const float* P, const float* R, const int32_t* idx, const int32_t* nsel,
                             const float* gamma, const float* beta, const float* mean_rstd, int B, int C,
                             int Nk, int Nq, int k, int groups, float slope, float* dP, float* dR,
                             float* sums, void* workspace, void* stream);

/* -------------------------------------------------------------- column sums --
 * out[n] = sum_m x[m,n] for row-major fp32 x [M,N] (N % 4 == 0): the bias gradient of the dense
 * layers (the `grad.sum(0)` autograd performs for torch.nn.Linear).  Deterministic; workspace of
 * ptops_colsum_workspace_bytes(M, N) bytes, 16-byte aligned. */
size_t ptops_colsum_workspace_bytes(int M, int N);
int ptops_colsum(const float* x, int M, int N, float* out, void* workspace, void* stream);

/* A/B switches of the library (PTOPS_ATTN_FWD_F32, PTOPS_ATTN_BWD_F32, PTOPS_ATTN_FWD_WAVES, PTOPS_ATTN_DKV_WAVES,
 * PTOPS_ATTN_QSPLIT, PTOPS_ATTN_NO_XCD_REMAP, PTOPS_CHAMFER_MFMA_PREFETCH, PTOPS_GEMM_OCC3): each is initialised ONCE from the environment variable of the same name
 * and afterwards changes only through ptops_set_option (returns the previous value, -1 for an unknown name), so that
 * a benchmark's label and the kernel that ran agree.  Benchmarking / test aid, not a tuning interface. */
int ptops_set_option(const char* name, int value);
int ptops_get_option(const char* name);

/* Deferred folds.  ptops_colsum, ptops_layer_norm_backward and ptops_gelu_backward_colsum end with a tiny launch that
 * adds their partial rows (115 such launches per AdaPoinTr backward, 5.5 us each).  Their results are leaf gradients
 * (bias / LayerNorm weight gradients) nobody reads before the end of backward, so a training loop may switch
 * deferral on around backward: the folds are queued and ptops_fold_flush runs all of them in one or two launches,
 * with the summation order of the immediate launch.  Deferral is per STREAM: ptops_fold_defer(stream, mode) affects only
 * folds launched on that stream (mode 0 off, 1 on, 2 on + every queued output is filled with NaN until the flush
 * writes it - a debug aid that makes a premature read visible); ptops_fold_flush(stream) runs that stream's queue.
 * While deferral is on the CALLER must keep every such call's workspace and output alive and unread until the flush
 * (so give each call its own workspace).  ptops_fold_defer returns the number of folds still queued on the stream;
 * switching off does not flush. */
int ptops_fold_defer(void* stream, int mode);
int ptops_fold_flush(void* stream);

/* ---------------------------------------------------------- max over tokens --
 * out[b,c] = max_m x[b,m,c], arg[b,c] = the (lowest) token that attains it, for row-major fp32 x [B,M,C]
 * (C % 4 == 0): torch.max(x, dim=1) of the global-feature pools (models/external/AdaPoinTr.py:110,225,
 * models/base_blocks.py:759).  backward: grad_x[b,m,c] = grad_out[b,c] if m == arg[b,c] else 0, fully written.
 * One kernel each, no scratch memory, no memset (safe under HIP-graph capture). */
int ptops_max_tokens_forward(const float* x, int B, int M, int C, float* out, int32_t* arg, void* stream);
int ptops_max_tokens_backward(const float* grad_out, const int32_t* arg, int B, int M, int C, float* grad_x,
                              void* stream);

/* ------------------------------------------------- fp32 GEMM on the bf16 matrix pipe --
 * C[M,N] = A . B^T (+ bias[N]) for fp32 operands, the product every nn.Linear of the path evaluates forward and
 * backward (models/Transformer_utils.py:71-159).  Each operand is split in the kernel into three bf16 pieces and the
 * six piece products down to 2^-18 are accumulated in fp32 on v_mfma_f32_32x32x16_bf16: as accurate as an fp32 FMA
 * chain, 2.7x the fp32 matrix peak.  A is logically [M,K], B logically [N,K]; either may be stored with k contiguous
 * ([rows, ld], *_k_contiguous = 1) or with k strided ([K, ld], rows contiguous, = 0), which covers x.W^T (forward),
 * dy.W (input gradient) and dy^T.x (weight gradient).  K % 16 == 0; k-contiguous operands need 16-byte aligned rows. */
int ptops_gemm_f32_split3(const float* A, long long lda, int a_k_contiguous, const float* B, long long ldb,
                          int b_k_contiguous, float* C, long long ldc, int M, int N, int K, const float* bias,
                          void* stream);

/* Linear + nn.GELU() forward in one pass (Mlp.fc1 + act, models/Transformer_utils.py:146-159): C = A . B^T + bias is the
 * pre-activation the backward needs, G = gelu(C) (exact erf form) comes out of the same epilogue. */
int ptops_gemm_f32_split3_gelu(const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc,
                               float* G, long long ldg, int M, int N, int K, const float* bias, void* stream);

/* Weight gradient of the same layers: dW[N,K] = dY[M,N]^T . X[M,K] and, when db != NULL, db[N] = the column sums of dY
 * (the bias gradient; the rows of dY pass through the staging registers anyway).  The long contraction (M rows) is
 * split over several workgroups per 128 x 128 output tile, whose partial products go to `workspace` and are added in
 * a fixed order by a second kernel: deterministic, no atomics, no memset.  M % 16 == 0; K % 4 == 0 and N % 4 == 0
 * (with db); dW rows, db and the workspace 16-byte aligned; *_workspace gives the bytes needed for a shape. */
long long ptops_gemm_f32_split3_wgrad_workspace(int M, int N, int K);
int ptops_gemm_f32_split3_wgrad(const float* dY, long long ldy, const float* X, long long ldx, int M, int N, int K,
                                float* dW, long long lddw, float* db, void* workspace, long long workspace_bytes,
                                void* stream);

/* Reductions of the two Chamfer distance vectors - ChamferDistanceL2 / L2_split / L1 / L1_PM
 * (extensions/chamfer_dist/__init__.py:28-103: torch.mean / torch.sqrt / + / 2) - one launch forward, one backward.
 * forward: out[0] = mean(f(dist1)), out[1] = mean(f(dist2)), out[2] = out[0] + out[1], out[3] = out[2] / 2 with
 * f = sqrt when take_sqrt else identity (n1, n2: element counts; mean of nothing is NaN as in torch).  Deterministic
 * (block partials added in a fixed order by the last block).  workspace: *_workspace_bytes(n1, n2) bytes whose first 4
 * bytes are zero before the first use; the kernel leaves them zero (one workspace per stream).
 * backward: grad_dist1 / grad_dist2 for the gradient of out[which], which = 2 or 3 (grad_out1: one float on the
 * device), or which = 0: of out[0] and out[1] separately (grad_out1 / grad_out2, NULL = zero), by autograd's own
 * formulas (mean: g / N, sqrt: g / (2 result), "/ 2": g / 2) in autograd's order: aten's values. */
size_t ptops_chamfer_reduce_workspace_bytes(long long n1, long long n2);
int ptops_chamfer_reduce(const float* dist1, long long n1, const float* dist2, long long n2, int take_sqrt, float* out,
                         void* workspace, void* stream);
int ptops_chamfer_reduce_backward(const float* dist1, long long n1, const float* dist2, long long n2, int take_sqrt,
                                  int which, const float* grad_out1, const float* grad_out2, float* grad_dist1,
                                  float* grad_dist2, void* stream);

/* The weight (and bias) gradients of MANY layers in two launches.  A layer's dW has few output tiles and a long
 * contraction, and nothing in a backward pass reads it before the optimizer: the host side queues the (dY, X) pairs
 * while autograd runs (torch's Linear backward, models/Transformer_utils.py:71-159 of the reference, computes each
 * dY^T.X at once) and runs them together.  Per problem the arithmetic and the fixed-order reduction are those of
 * ptops_gemm_f32_split3_wgrad; the planner gives all tiles of a (problem, contraction split) to one XCD.
 *   problems: n rows of PtopsWgradProblem, the same bytes on the host (validated) and on the device (read by the kernels);
 *   plan:     ptops_wgrad_group_plan(mnk = n x {M,N,K}, want_db = n flags or NULL, steps_target = 0 for the default
 *             job length in 16-row steps, plan, capacity) returns the plan's size in bytes (-1: unsupported shape) and
 *             writes it when capacity suffices; it depends on the shapes only, the caller copies it to the device once;
 *   workspace: ptops_wgrad_group_workspace_bytes(plan) bytes of device scratch, 16-byte aligned.
 * M % 16 == 0, K % 4 == 0, dW rows 16-byte aligned; with db: N % 4 == 0, db 16-byte aligned. */
typedef struct PtopsWgradProblem {
  const float* dY;          /* [M, ldy] */
  const float* X;           /* [M, ldx] */
  float* dW;                /* [N, lddw] out */
  float* db;                /* [N] out, or NULL */
  long long ldy, ldx, lddw;
  int M, N, K, reserved;
} PtopsWgradProblem;
int ptops_wgrad_problem_bytes(void);                    /* sizeof(PtopsWgradProblem) = 72 */
long long ptops_wgrad_group_plan(const int* mnk, const unsigned char* want_db, int n, int steps_target, void* plan,
                                 long long capacity);
long long ptops_wgrad_group_workspace_bytes(const void* plan_host);
int ptops_gemm_f32_split3_wgrad_grouped(const void* problems_host, const void* problems_dev, int n, const void* plan_host,
                                        const void* plan_dev, void* workspace, long long workspace_bytes, void* stream);

/* ------------------------------------------------------------------- clip + AdamW --
 * torch.nn.utils.clip_grad_norm_(params, max_norm, 2) followed by torch.optim.AdamW.step() (tools/runner.py:208-212,
 * tools/builder.py:37-52) for ALL parameters in three launches: |g|^2 per 4096-element chunk, the norm and the clip
 * coefficient min(1, max_norm / (norm + 1e-6)), the update (decoupled weight decay, lerp / addcmul moments,
 * bias-corrected step; the arithmetic of torch/optim/adamw.py::_single_tensor_adamw in fp32).  The gradients are read,
 * not scaled in place.
 * rows: device table of n_tensors rows of ptops_adamw_row_bytes() = 64 bytes {float* param, const float* grad,
 * float* exp_avg, float* exp_avg_sq, int64 numel, float lr, float weight_decay, float 1 - beta1^step,
 * float sqrt(1 - beta2^step), 8 bytes padding}; chunks: device table of n_chunks rows {int32 tensor, int32 0,
 * int64 offset}, one per started ptops_adamw_chunk_elements() of every tensor; partial: n_chunks floats of scratch;
 * norm_coef: 2 floats out = {gradient norm before clipping, coefficient applied}.  max_norm <= 0: no clipping. */
int ptops_adamw_row_bytes(void);
int ptops_adamw_chunk_elements(void);
int ptops_adamw_clip_step(const void* rows, int n_tensors, const void* chunks, int n_chunks, float max_norm, float beta1,
                          float beta2, float eps, float* partial, float* norm_coef, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PTOPS_H_ */
