/*
 * emd_ref.c — CPU ORACLE for the EMD (auction) operator.  TEST INFRASTRUCTURE ONLY.
 *
 * Restates extensions/emd/emd_cuda.cu:23-316 of the reference (host loop :256-269, Bid :95-179,
 * GetMax :181-194, Assign :196-215, CalcDist :217-226, backward :284-316) in scalar C.
 *
 * The CUDA reference is not deterministic where several candidates tie: the winner of an exact
 * tie in the bid value depends on how its Bid kernel happens to partition candidates over threads
 * (emd_cuda.cu:118-121,139-160), and two bidders within 1e-6 of an object's best increment race
 * on max_idx (emd_cuda.cu:188-190).  This restatement (and the HIP kernel) fix both by index:
 * the LOWEST candidate index wins a bid-value tie, the LOWEST bidder index wins an increment tie.
 * Everywhere else the arithmetic is the reference's, including the double-precision evaluation of
 * `3.0 - sqrtf(.) - price` (emd_cuda.cu:146) and of the +-1e-6 window (emd_cuda.cu:188).
 * PARITY UNPINNED beyond that: the reference holds no known-answer vector for EMD (its only
 * self-test, emd_module.py:81-97, prints values for random input) and cannot be built here.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REF_API __attribute__((visibility("default")))

static inline float sq3(float dx, float dy, float dz) { return fmaf(dz, dz, fmaf(dy, dy, dx * dx)); }

/* One batch element.  All arrays are that element's slices. */
static void emd_one(const float* a, const float* b, int n, float eps, int iters, float* dist,
                    int32_t* assignment, int32_t* assignment_inv, float* price, int32_t* bid,
                    float* bid_inc, float* max_inc, int32_t* max_idx) {
  int32_t* unass = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  for (int it = 0; it < iters; ++it) {
    const int last = (it == iters - 1);
    int cnt = 0;
    for (int j = 0; j < n; ++j)
      if (assignment[j] == -1) unass[cnt++] = j;
    /* Bid (emd_cuda.cu:95-179) */
    for (int u = 0; u < cnt; ++u) {
      const int j = unass[u];
      const float x1 = a[j * 3], y1 = a[j * 3 + 1], z1 = a[j * 3 + 2];
      float best = -1e9f, better = -1e9f;
      int best_i = -1;
      for (int k = 0; k < n; ++k) {
        const float s = sqrtf(sq3(b[k * 3] - x1, b[k * 3 + 1] - y1, b[k * 3 + 2] - z1));
        const float d = (float)(3.0 - (double)s - (double)price[k]);
        if (d > best) {
          better = best;
          best = d;
          best_i = k;
        } else if (d > better) {
          better = d;
        }
      }
      const float inc = best - better + eps;
      bid[j] = best_i;
      bid_inc[j] = inc;
      if (inc > max_inc[best_i]) max_inc[best_i] = inc; /* atomicMax, emd_cuda.cu:177 */
    }
    /* GetMax (emd_cuda.cu:181-194): lowest bidder inside the +-1e-6 window of the best increment */
    for (int u = cnt - 1; u >= 0; --u) {
      const int j = unass[u];
      const int o = bid[j];
      const float bi = bid_inc[j], mi = max_inc[o];
      if ((double)bi - 1e-6 <= (double)mi && (double)mi <= (double)bi + 1e-6) max_idx[o] = j;
    }
    /* Assign (emd_cuda.cu:196-215) */
    for (int u = 0; u < cnt; ++u) {
      const int j = unass[u];
      const int o = bid[j];
      if (last || max_idx[o] == j) {
        const int prev = assignment_inv[o];
        if (!last && prev != -1) assignment[prev] = -1;
        assignment_inv[o] = j;
        assignment[j] = o;
        price[o] += bid_inc[j];
        max_inc[o] = -1e9f;
      }
    }
  }
  /* CalcDist (emd_cuda.cu:217-226) */
  for (int j = 0; j < n; ++j) {
    const int k = assignment[j];
    dist[j] = sq3(a[j * 3] - b[k * 3], a[j * 3 + 1] - b[k * 3 + 1], a[j * 3 + 2] - b[k * 3 + 2]);
  }
  free(unass);
}

/* emd_cuda_forward.  Outputs dist [B,n], assignment [B,n]; also returns assignment_inv and price
 * for inspection.  Scratch is internal.  Returns 1 (ok) or -1 (bad shape), like emd_cuda.cu:236-249. */
REF_API int ref_emd_forward(const float* xyz1, const float* xyz2, int B, int n, int m, float eps,
                            int iters, float* dist, int32_t* assignment, int32_t* assignment_inv,
                            float* price) {
  if (n != m || B > 512 || n % 1024 != 0) return -1;
#pragma omp parallel for schedule(dynamic, 1)
  for (int i = 0; i < B; ++i) {
    size_t o = (size_t)i * n;
    int32_t* bid = (int32_t*)calloc((size_t)n, sizeof(int32_t));
    float* bid_inc = (float*)calloc((size_t)n, sizeof(float));
    float* max_inc = (float*)calloc((size_t)n, sizeof(float)); /* zeros, emd_module.py:51 */
    int32_t* max_idx = (int32_t*)calloc((size_t)n, sizeof(int32_t));
    for (int j = 0; j < n; ++j) {
      assignment[o + j] = -1;
      assignment_inv[o + j] = -1;
      price[o + j] = 0.0f;
    }
    emd_one(xyz1 + o * 3, xyz2 + o * 3, n, eps, iters, dist + o, assignment + o, assignment_inv + o,
            price + o, bid, bid_inc, max_inc, max_idx);
    free(bid);
    free(bid_inc);
    free(max_inc);
    free(max_idx);
  }
  return 1;
}

/* NmDistanceGradKernel (emd_cuda.cu:284-301): grad_xyz1[j] += 2*graddist[j]*(xyz1[j]-xyz2[idx[j]]). */
REF_API void ref_emd_backward(const float* xyz1, const float* xyz2, const float* graddist,
                              const int32_t* idx, int B, int n, float* gradxyz1) {
  for (int i = 0; i < B; ++i)
    for (int j = 0; j < n; ++j) {
      size_t e = (size_t)i * n + j;
      const int k = idx[e];
      const float g = graddist[e] * 2.0f;
      for (int d = 0; d < 3; ++d)
        gradxyz1[e * 3 + d] += g * (xyz1[e * 3 + d] - xyz2[((size_t)i * n + k) * 3 + d]);
    }
}
