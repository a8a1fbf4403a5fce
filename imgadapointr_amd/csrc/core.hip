// libptops: ABI version, the thread-local error message and the fill kernel used instead of hipMemsetAsync.
#include <stdarg.h>
#include <stdio.h>

#include "ptops_common.h"

namespace ptops {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
namespace {
__global__ __launch_bounds__(256) void fill_bytes_kernel(unsigned char* __restrict__ p, unsigned int word, size_t bytes) {
  // 16-byte stores over the aligned middle, single bytes at the two ragged ends
  const size_t head = ((16 - ((uintptr_t)p & 15)) & 15) < bytes ? ((16 - ((uintptr_t)p & 15)) & 15) : bytes;
  const size_t vecs = (bytes - head) / 16;
  const size_t tid = blockIdx.x * 256ull + threadIdx.x, nth = gridDim.x * 256ull;
  uint4* v = reinterpret_cast<uint4*>(p + head);
  const uint4 w4 = {word, word, word, word};
  for (size_t i = tid; i < vecs; i += nth) v[i] = w4;
  const unsigned char b = (unsigned char)(word & 0xFF);
  if (tid < head) p[tid] = b;
  const size_t tail0 = head + vecs * 16;
  if (tid < bytes - tail0) p[tail0 + tid] = b;
}
}  // namespace

// hipMemsetAsync replacement.  A MEMSET node captured into a HIP graph stops taking effect on replays that follow a
// host synchronisation (ROCm 7.2 / PyTorch 2.10+rocm7.0, benchmarks/graph_diag3.py: every graph containing one —
// ours or ATen's reduction semaphores — returns stale results, graphs of kernels only never do), so nothing in this
// library that can run under stream capture issues one.  `value` is a byte, as for memset.
int fill_async(void* ptr, int value, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return PTOPS_OK;
  const unsigned int b = (unsigned int)(value & 0xFF);
  const unsigned int word = b | (b << 8) | (b << 16) | (b << 24);
  const size_t vecs = bytes / 16 + 2;
  size_t blocks = (vecs + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (unsigned char*)ptr, word, bytes);
  return check_launch("fill_bytes_kernel");
}
}  // namespace ptops

PTOPS_API int ptops_abi_version(void) { return 1; }
PTOPS_API const char* ptops_last_error(void) { return ptops::g_err; }
