// Shared host-side helpers for libptops (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "ptops.h"

#define PTOPS_API extern "C" __attribute__((visibility("default")))

namespace ptops {

// Thread-local message behind ptops_last_error().
void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return PTOPS_ELAUNCH;
  }
  return PTOPS_OK;
}

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline long long ceil_div_ll(long long a, long long b) { return (a + b - 1) / b; }

// memset(ptr, value, bytes) as a kernel on `stream` (never a hipMemsetAsync: see core.hip)
int fill_async(void* ptr, int value, size_t bytes, hipStream_t stream);

// core.hip: A/B switch `name` (an environment variable read once at first use; ptops_set_option changes it)
int option(const char* name);

constexpr int kNumCU = 256;   // MI355X
constexpr int kWave = 64;

}  // namespace ptops

// reduce.hip: out[n] = sum over the S rows of partial [S, N] in a fixed order, one launch (N % 4 == 0).
// `deferrable`: the result is a leaf gradient nobody reads before the end of backward - while ptops_fold_defer(1) is
// in force the fold is queued instead of launched and ptops_fold_flush() runs all queued folds in one or two launches
// (same summation order as the immediate launch).
namespace ptops_reduce {
int fold_rows(const float* partial, int S, int N, float* out, hipStream_t stream, bool deferrable = false);
}

// chamfer_sweep.hip: exact pruned Chamfer search (sort by x + bounded sweep)
namespace ptops_sweep {
bool eligible(int B, int n, int m);
size_t workspace_bytes(int B, int n, int m);
int forward(const float* xyz1, const float* xyz2, int B, int n, int m, float* dist1, float* dist2,
            int32_t* idx1, int32_t* idx2, void* workspace, unsigned long long* stats, int dirmask,
            hipStream_t stream);
}  // namespace ptops_sweep

// chamfer_mfma.hip: exact Chamfer search with an fp32 MFMA filter + exact re-evaluation
namespace ptops_mfma {
bool eligible(int B, int n, int m);
size_t workspace_bytes(int B, int n, int m);
int forward(const float* xyz1, const float* xyz2, int B, int n, int m, float* dist1, float* dist2,
            int32_t* idx1, int32_t* idx2, void* workspace, unsigned long long* stats, int dirmask,
            hipStream_t stream);
}  // namespace ptops_mfma

#define PTOPS_REQUIRE(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      ptops::set_error(__VA_ARGS__);    \
      return PTOPS_EINVAL;              \
    }                                   \
  } while (0)
