// libptops: ABI version, the thread-local error message and the fill kernel used instead of hipMemsetAsync.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

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

// ---- A/B switches: every switch is read from the environment ONCE (first use) and can be changed afterwards only
// through ptops_set_option, so that a label printed by a benchmark and the kernel that ran cannot disagree.
namespace ptops {
namespace {
struct Option {
  const char* name;
  int value;
};
Option g_options[] = {{"PTOPS_ATTN_FWD_F32", 0},  {"PTOPS_ATTN_BWD_F32", 0}, {"PTOPS_ATTN_FWD_WAVES", 0},
                      {"PTOPS_ATTN_DKV_WAVES", 0}, {"PTOPS_ATTN_FWD_ABL", 0}, {"PTOPS_ATTN_STAGE_EARLY", 0}, {"PTOPS_ATTN_PRIO", 0}, {"PTOPS_ATTN_QSPLIT", 0}, {"PTOPS_ATTN_NO_XCD_REMAP", 0}, {"PTOPS_CHAMFER_MFMA_PREFETCH", 0}, {"PTOPS_CHAMFER_MFMA_PREP", 0}, {"PTOPS_GEMM_OCC3", 0},
                      {"PTOPS_WGRAD_GROUP_STEPS", 0}, {"PTOPS_WGRAD_OCC3", 0}, {"PTOPS_GEMM_DEBUG_NO_STORE", 0}};
std::once_flag g_options_once;
void read_options() {
  for (Option& o : g_options)
    if (const char* v = getenv(o.name)) o.value = (v[0] >= '0' && v[0] <= '9') ? atoi(v) : 1;
}
}  // namespace
int option(const char* name) {
  std::call_once(g_options_once, read_options);
  for (const Option& o : g_options)
    if (strcmp(o.name, name) == 0) return o.value;
  return 0;
}
}  // namespace ptops

// Sets an A/B switch (names = the environment variables that initialise them).  Returns the previous value or -1 for an
// unknown name.  Not a tuning interface: the defaults are what is measured and tested.
PTOPS_API int ptops_set_option(const char* name, int value) {
  std::call_once(ptops::g_options_once, ptops::read_options);
  for (ptops::Option& o : ptops::g_options)
    if (name && strcmp(o.name, name) == 0) {
      const int old = o.value;
      o.value = value;
      return old;
    }
  ptops::set_error("set_option: unknown option");
  return -1;
}
PTOPS_API int ptops_get_option(const char* name) { return name ? ptops::option(name) : 0; }

PTOPS_API int ptops_abi_version(void) { return 1; }
PTOPS_API const char* ptops_last_error(void) { return ptops::g_err; }
