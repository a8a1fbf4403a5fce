// Gradient clipping + AdamW for ALL parameters of the model in three launches.
//
// The reference clips the global gradient norm (torch.nn.utils.clip_grad_norm_, tools/runner.py:208-212) and then steps
// torch.optim.AdamW (tools/builder.py:37-52: weight decay 5e-4, none for 1-D parameters and biases).  On 335 tensors /
// 32.5 M parameters PyTorch-ROCm runs that as ~25 launches (for-each norm, scale, ~20 multi-tensor AdamW chunks): 0.55 ms
// of a 33 ms step for work that is three passes over memory: |g|^2, the norm, the update.
//
//   adamw_sumsq_kernel   one workgroup per 4096-element chunk of a gradient -> partial[chunk]   (reads the gradients once)
//   adamw_norm_kernel    one workgroup adds the partials in a fixed order -> {norm, clip coefficient}
//   adamw_update_kernel  one workgroup per chunk: g * coef, decoupled weight decay, moments, bias-corrected update
//
// Per-tensor data (pointers, element count, lr, weight decay, bias corrections) is a device-resident table the host
// refreshes per step (the gradient tensors are new every step); the chunk -> (tensor, offset) map is fixed.
// Arithmetic follows torch/optim/adamw.py (_single_tensor_adamw) term by term in fp32; fp contraction is off.
#include "ptops_common.h"

namespace {

constexpr int kChunk = 4096;

struct TensorRow {        // 64 bytes
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  long long numel;
  float lr, weight_decay;
  float bias_correction1, bias_correction2_sqrt;
  long long pad;
};

struct ChunkRow {         // 16 bytes
  int tensor;
  int pad;
  long long offset;
};

__global__ __launch_bounds__(256) void adamw_sumsq_kernel(const TensorRow* __restrict__ rows, const ChunkRow* __restrict__ chunks,
                                                          float* __restrict__ partial) {
  __shared__ float red[4];
  const ChunkRow c = chunks[blockIdx.x];
  const TensorRow r = rows[c.tensor];
  const long long end = min(c.offset + kChunk, r.numel);
  float acc = 0.f;
  for (long long i = c.offset + threadIdx.x; i < end; i += 256) {
    const float g = r.grad[i];
    acc += g * g;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = total norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6)), NaN if the norm is NaN
// (max_norm <= 0: no clipping)
__global__ __launch_bounds__(1024) void adamw_norm_kernel(const float* __restrict__ partial, int n, float max_norm,
                                                          float* __restrict__ out) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) acc += partial[i];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w];
    const float norm = sqrtf(s);
    out[0] = norm;
    // torch.clamp(max_norm / (norm + 1e-6), max=1.0): a NaN norm gives a NaN coefficient that poisons every gradient
    // and parameter, exactly as clip_grad_norm_ + step() of the reference does (fminf would hide it behind a 1)
    const float c = max_norm / (norm + 1e-6f);
    out[1] = max_norm > 0.f ? (c < 1.0f ? c : (c != c ? c : 1.0f)) : 1.0f;
  }
}

__global__ __launch_bounds__(256) void adamw_update_kernel(const TensorRow* __restrict__ rows, const ChunkRow* __restrict__ chunks,
                                                           const float* __restrict__ norm_coef, float beta1, float beta2,
                                                           float eps) {
  const ChunkRow c = chunks[blockIdx.x];
  const TensorRow r = rows[c.tensor];
  const long long end = min(c.offset + kChunk, r.numel);
  const float coef = norm_coef[1];
  const float decay = 1.0f - r.lr * r.weight_decay;
  const float step_size = r.lr / r.bias_correction1;
  for (long long i = c.offset + threadIdx.x; i < end; i += 256) {
    const float g = r.grad[i] * coef;
    float p = r.param[i] * decay;
    float m = r.exp_avg[i], v = r.exp_avg_sq[i];
    m = m + (g - m) * (1.0f - beta1);                       // exp_avg.lerp_(grad, 1 - beta1)
    v = v * beta2 + (1.0f - beta2) * g * g;                 // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) / r.bias_correction2_sqrt + eps;
    p = p - step_size * (m / denom);                        // param.addcdiv_(exp_avg, denom, value=-step_size)
    r.param[i] = p;
    r.exp_avg[i] = m;
    r.exp_avg_sq[i] = v;
  }
}

}  // namespace

PTOPS_API int ptops_adamw_row_bytes(void) { return (int)sizeof(TensorRow); }
PTOPS_API int ptops_adamw_chunk_elements(void) { return kChunk; }

// rows: device table of n_tensors 64-byte rows {param*, grad*, exp_avg*, exp_avg_sq*, int64 numel, f32 lr, f32 weight_decay,
// f32 bias_correction1, f32 sqrt(bias_correction2), 8 bytes padding}; chunks: device table of n_chunks 16-byte rows
// {int32 tensor, int32 0, int64 offset}, one per started kChunk elements of every tensor; partial: n_chunks floats;
// norm_coef: 2 floats (out: the gradient norm before clipping and the coefficient applied).
PTOPS_API int ptops_adamw_clip_step(const void* rows, int n_tensors, const void* chunks, int n_chunks, float max_norm,
                                    float beta1, float beta2, float eps, float* partial, float* norm_coef, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(n_tensors >= 0 && n_chunks >= 0, "adamw_clip_step: negative count");
  if (n_tensors == 0 || n_chunks == 0) return PTOPS_OK;
  PTOPS_REQUIRE(rows && chunks && partial && norm_coef, "adamw_clip_step: null pointer");
  hipLaunchKernelGGL(adamw_sumsq_kernel, dim3(n_chunks), dim3(256), 0, stream, (const TensorRow*)rows, (const ChunkRow*)chunks,
                     partial);
  int rc = ptops::check_launch("adamw_sumsq_kernel");
  if (rc != PTOPS_OK) return rc;
  hipLaunchKernelGGL(adamw_norm_kernel, dim3(1), dim3(1024), 0, stream, partial, n_chunks, max_norm, norm_coef);
  rc = ptops::check_launch("adamw_norm_kernel");
  if (rc != PTOPS_OK) return rc;
  hipLaunchKernelGGL(adamw_update_kernel, dim3(n_chunks), dim3(256), 0, stream, (const TensorRow*)rows, (const ChunkRow*)chunks,
                     norm_coef, beta1, beta2, eps);
  return ptops::check_launch("adamw_update_kernel");
}
