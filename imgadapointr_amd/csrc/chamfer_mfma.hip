// Chamfer nearest-neighbour search with an fp32 MFMA filter, gfx950 (MI355X).
//
// Same operator and the same bit-exact results as chamfer.hip (reference: extensions/chamfer_dist/chamfer.cu:15-171,
// squared distance by direct differences, lowest index wins ties), a different way to get there.
//
// The brute-force kernel is at the VALU issue ceiling: the reference's arithmetic costs 6.8 vector instructions per
// pair.  Here the matrix pipe does the searching and the vector pipe only the bookkeeping:
//   * prep pass: every candidate c becomes the row (-2cx, -2cy, -2cz, |c|^2) of an MFMA A operand;
//   * a query q is the column (qx, qy, qz, 1) of the B operand, so ONE v_mfma_f32_16x16x4_f32 yields
//     D[c][q] = |c|^2 - 2 c.q = |c - q|^2 - |q|^2 for 16 candidates x 16 queries (8 FLOP per pair, the algorithmic
//     count of the direct form), exact f32 fma chain, 32 cycles per instruction per SIMD = 8 pairs/cycle/SIMD against
//     4.7 for the VALU kernel;
//   * D is only a FILTER.  Its error against the exact direct-difference value is bounded by E = 20 u (|c|max + |q|)^2
//     (u = 2^-24, derivation in DESIGN.md 4.1c), so the true nearest neighbour (and every candidate that ties with
//     it) has D <= min D + 2E.  Each lane sees 4 of the 16 candidate rows of every MFMA; per 64-candidate superblock
//     it folds its 16 values with v_min3 and keeps the three smallest superblock minima and the ids of the first
//     two (v_med3 insertion, 4 vector instructions per MFMA in all).  At the end the one or two superblocks whose
//     minimum lies within 2E of the lane's best are re-evaluated with the reference's exact arithmetic, lowest index
//     first; if a third one qualifies too (duplicated points, lattices) the lane scans its whole subset exactly.
//     The four lanes that share a query merge (distance, index) lexicographically.
// Both directions run in one launch; a direction with few queries splits its candidates over workgroups that merge
// through atomicMin on (dist_bits << 32 | index), as in chamfer.hip.
#include <math.h>
#include <stdlib.h>

#include "ptops_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTPB = 256;    // 4 waves
constexpr int kQG = 4;       // 16-query groups per wave
constexpr int kQW = 16 * kQG;            // queries per wave
constexpr int kQB = kQW * (kTPB / 64);   // queries per workgroup
constexpr int kSBC = 64;     // candidates per superblock (4 MFMA row blocks)
constexpr float kBig = 3.0e38f;          // |c|^2 of a padding candidate: never the minimum, never overflows
constexpr float kU = 5.9604645e-8f;      // 2^-24

struct DirM {
  const float* q;          // [B,nq,3] queries
  const float* c;          // [B,nc,3] candidates (exact re-evaluation)
  const float* prep;       // [B,ncp/16,4,16] MFMA A operands of the candidates
  const unsigned* cmax2;   // [B] bit pattern of max |c|^2
  float* dist;             // [B,nq]
  int32_t* idx;            // [B,nq]
  unsigned long long* ws;  // [B,nq] packed minima (splits > 1 only)
  int nq, nc, ncp;         // ncp = nc rounded up to 64
  int qtiles, splits, sb_per_split, first_block;
};

struct ArgsM {
  DirM d[2];
  unsigned long long* stats;   // optional {EXTRA exact superblock evaluations, superblock visits} per direction
};

__device__ __forceinline__ float sqdist(float qx, float qy, float qz, float cx, float cy, float cz) {
  float dx = cx - qx, dy = cy - qy, dz = cz - qz;
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

__device__ __forceinline__ float min3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }

// (-2x, -2y, -2z, |c|^2) per candidate in A-operand order: 64 floats per 16 candidates, element l = component l>>4
// of candidate l&15.  Also the per-cloud maximum of |c|^2 (atomicMax on the bit pattern, values are >= 0).
__global__ __launch_bounds__(256) void chamfer_prep_kernel(const float* __restrict__ xyz, int B, int n, int np,
                                                           float* __restrict__ prep, unsigned* __restrict__ cmax2) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  float cn = 0.f;
  if (j < np) {
    float x = 0.f, y = 0.f, z = 0.f, w = kBig;
    if (j < n) {
      const float* p = xyz + ((size_t)b * n + j) * 3;
      x = p[0], y = p[1], z = p[2];
      cn = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
      w = cn;
    }
    float* o = prep + ((size_t)b * np + (size_t)(j >> 4) * 16) * 4 + (j & 15);
    o[0] = -2.0f * x;
    o[16] = -2.0f * y;
    o[32] = -2.0f * z;
    o[48] = w;
  }
  if (!(cn == cn) || cn > 3.0e38f) cn = INFINITY;   // NaN / overflow: an infinite bound sends every lane to the exact scan
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cn = __builtin_fmaxf(cn, __shfl_xor(cn, off));
  if ((threadIdx.x & 63) == 0) atomicMax(cmax2 + b, __float_as_uint(cn));
}

__global__ __launch_bounds__(256) void chamfer_mfma_unpack_kernel(const unsigned long long* __restrict__ ws,
                                                                  float* __restrict__ dist, int32_t* __restrict__ idx,
                                                                  long long total) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
    unsigned long long key = ws[i];
    dist[i] = __uint_as_float((unsigned int)(key >> 32));
    idx[i] = (int32_t)(unsigned int)key;
  }
}

// (kTPB, 4): at most 128 registers, which also makes hipcc write MFMA results to VGPRs instead of AGPRs + v_accvgpr_read
__global__ __launch_bounds__(kTPB, 4) void chamfer_mfma_kernel(ArgsM a) {
  const int bid = blockIdx.x;
  const bool second = bid >= a.d[1].first_block;
  const float* __restrict__ qptr = second ? a.d[1].q : a.d[0].q;
  const float* __restrict__ cptr = second ? a.d[1].c : a.d[0].c;
  const float* __restrict__ pptr = second ? a.d[1].prep : a.d[0].prep;
  const unsigned* __restrict__ cmax2 = second ? a.d[1].cmax2 : a.d[0].cmax2;
  float* __restrict__ dist = second ? a.d[1].dist : a.d[0].dist;
  int32_t* __restrict__ idx = second ? a.d[1].idx : a.d[0].idx;
  unsigned long long* __restrict__ ws = second ? a.d[1].ws : a.d[0].ws;
  const int nq = second ? a.d[1].nq : a.d[0].nq;
  const int nc = second ? a.d[1].nc : a.d[0].nc;
  const int ncp = second ? a.d[1].ncp : a.d[0].ncp;
  const int qtiles = second ? a.d[1].qtiles : a.d[0].qtiles;
  const int splits = second ? a.d[1].splits : a.d[0].splits;
  const int sps = second ? a.d[1].sb_per_split : a.d[0].sb_per_split;
  int local = bid - (second ? a.d[1].first_block : a.d[0].first_block);
  const int s = local % splits;
  local /= splits;
  const int qt = local % qtiles;
  const int b = local / qtiles;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, grp = lane >> 4;
  const float* __restrict__ qb = qptr + (size_t)b * nq * 3;
  const float* __restrict__ cb = cptr + (size_t)b * nc * 3;
  const int nsb = ncp / kSBC;
  const int sb_begin = s * sps;
  const int sb_end = min(nsb, sb_begin + sps);
  const int qbase = qt * kQB + wave * kQW;

  // B operands: lane (col, grp) holds component grp of query col; component 3 is the constant 1
  float bq[kQG];
#pragma unroll
  for (int g = 0; g < kQG; ++g) {
    const int j = min(qbase + g * 16 + col, nq - 1);
    bq[g] = grp < 3 ? qb[(size_t)j * 3 + grp] : 1.0f;
  }
  float m1[kQG], m2[kQG], m3[kQG];
  int b1[kQG], b2[kQG];
#pragma unroll
  for (int g = 0; g < kQG; ++g) {
    m1[g] = m2[g] = m3[g] = INFINITY;
    b1[g] = b2[g] = sb_begin;
  }

  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ ap = pptr + ((size_t)b * ncp + (size_t)sb_begin * kSBC) * 4 + lane;
  float a0 = ap[0], a1 = ap[64], a2 = ap[128], a3 = ap[192];
  for (int sb = sb_begin; sb < sb_end; ++sb) {
    const float* __restrict__ an = ap + (sb + 1 < sb_end ? 256 : 0);   // next superblock's operands, one step ahead
    const float n0 = an[0], n1 = an[64], n2 = an[128], n3 = an[192];
    f32x4 r[kQG][4];
#pragma unroll
    for (int g = 0; g < kQG; ++g) {                 // all 16 MFMAs first: independent, back to back on the matrix pipe
      r[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bq[g], zero, 0, 0, 0);
      r[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bq[g], zero, 0, 0, 0);
      r[g][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, bq[g], zero, 0, 0, 0);
      r[g][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, bq[g], zero, 0, 0, 0);
    }
    // Keep the 16 MFMAs together: hipcc otherwise interleaves each one with the v_min3 that consume the previous result,
    // every MFMA then waits out the full result latency of its predecessor and the matrix pipe idles 60 % of the time
    // (profiles/round2: SQ_VALU_MFMA_BUSY 38 %).  Batched, one wave's vector phase overlaps the other waves' MFMAs.
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < kQG; ++g) {
      // 16 values -> one: eight v_min3 (the +inf seed keeps the compiler from canonicalising raw MFMA outputs)
      float v = min3(INFINITY, r[g][0].x, r[g][0].y);
      v = min3(v, r[g][0].z, r[g][0].w);
      v = min3(v, r[g][1].x, r[g][1].y);
      v = min3(v, r[g][1].z, r[g][1].w);
      v = min3(v, r[g][2].x, r[g][2].y);
      v = min3(v, r[g][2].z, r[g][2].w);
      v = min3(v, r[g][3].x, r[g][3].y);
      v = min3(v, r[g][3].z, r[g][3].w);
      // sorted insertion of (v, sb) into the three smallest superblock minima (v_med3 x2, 2 compares, 4 selects)
      const float o1 = m1[g], o2 = m2[g], o3 = m3[g];
      const int ob1 = b1[g], ob2 = b2[g];
      const bool lt1 = v < o1, lt2 = v < o2;
      const int t2 = lt2 ? sb : ob2;
      m3[g] = __builtin_amdgcn_fmed3f(o2, o3, v);
      m2[g] = __builtin_amdgcn_fmed3f(o1, o2, v);
      m1[g] = lt1 ? v : o1;
      b2[g] = lt1 ? ob1 : t2;
      b1[g] = lt1 ? sb : ob1;
    }
    __builtin_amdgcn_sched_barrier(0);
    a0 = n0, a1 = n1, a2 = n2, a3 = n3;
    ap = an;
  }

  // ---- exact phase: the reference's arithmetic on the superblocks that can hold the nearest neighbour
  const float cmax = sqrtf(__uint_as_float(cmax2[b]));
  unsigned long long n_exact = 0;
#pragma unroll 1
  for (int g = 0; g < kQG; ++g) {
    const int jq = qbase + g * 16 + col;
    const int j = min(jq, nq - 1);
    const float qx = qb[(size_t)j * 3 + 0], qy = qb[(size_t)j * 3 + 1], qz = qb[(size_t)j * 3 + 2];
    const float R = sqrtf(__builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx))) + cmax;
    const float thr = (40.0f * kU) * R * R;                 // >= 2 E with margin for its own rounding
    float best = INFINITY;
    int besti = 0x7fffffff;
    auto rescan = [&](int sbi) {
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        const int base = sbi * kSBC + blk * 16 + 4 * grp;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = base + r;
          if (k < nc) {
            const float d = sqdist(qx, qy, qz, cb[(size_t)k * 3 + 0], cb[(size_t)k * 3 + 1], cb[(size_t)k * 3 + 2]);
            const bool better = d < best || (d == best && k < besti);
            best = better ? d : best;
            besti = better ? k : besti;
          }
        }
      }
    };
    const bool full = !(m3[g] > m1[g] + thr);               // written so that a NaN / infinite bound means "scan all"
    if (!full) {
      rescan(b1[g]);                                          // the one every lane pays: not counted
      if (!(m2[g] > m1[g] + thr)) {
        rescan(b2[g]);
        ++n_exact;
      }
    } else {
      for (int sb = sb_begin; sb < sb_end; ++sb) rescan(sb);
      n_exact += (unsigned long long)(sb_end - sb_begin);
    }
    // the four lanes of a query: lexicographic minimum of (distance, index)
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      const float od = __shfl_xor(best, off);
      const int oi = __shfl_xor(besti, off);
      const bool better = od < best || (od == best && oi < besti);
      best = better ? od : best;
      besti = better ? oi : besti;
    }
    if (grp == 0 && jq < nq) {
      const size_t o = (size_t)b * nq + jq;
      if (splits == 1) {
        dist[o] = best;
        idx[o] = besti;
      } else {
        unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned int)besti;
        atomicMin(&ws[o], key);
      }
    }
  }
  if (a.stats != nullptr) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n_exact += __shfl_xor(n_exact, off);
    if (lane == 0) {
      atomicAdd(a.stats + (second ? 2 : 0), n_exact);
      atomicAdd(a.stats + (second ? 3 : 1), (unsigned long long)(sb_end - sb_begin) * 64ull * kQG);
    }
  }
}

struct PlanM {
  int qtiles[2], splits[2], sps[2];
  bool need_ws;
};

PlanM make_plan(int B, int n, int m) {
  PlanM p;
  const int nq[2] = {n, m}, nc[2] = {m, n};
  p.need_ws = false;
  for (int d = 0; d < 2; ++d) {
    p.qtiles[d] = ptops::ceil_div(nq[d], kQB);
    const long long tiles = (long long)B * p.qtiles[d];
    const int nsb = ptops::ceil_div(nc[d], kSBC);
    const int target = 8 * ptops::kNumCU;                        // 8 workgroups (32 waves) per CU
    long long s = tiles >= target / 2 ? 1 : target / (tiles > 0 ? tiles : 1);
    const int max_s = nsb / 4 > 0 ? nsb / 4 : 1;                 // >= 256 candidates per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    int sps = ptops::ceil_div(nsb, (int)s);
    s = ptops::ceil_div(nsb, sps);                               // drop empty splits
    p.splits[d] = (int)s;
    p.sps[d] = sps;
    if (s > 1) p.need_ws = true;
  }
  return p;
}

inline size_t pad64(int n) { return ((size_t)n + 63) / 64 * 64; }
inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

namespace ptops_mfma {

bool eligible(int B, int n, int m) { return B >= 1 && n >= 1 && m >= 1; }

// [prep cloud 1 | prep cloud 2 | cmax2[2][B] | packed minima B*(n+m) u64]
size_t workspace_bytes(int B, int n, int m) {
  return align256(sizeof(float) * 4 * (size_t)B * pad64(n)) + align256(sizeof(float) * 4 * (size_t)B * pad64(m)) +
         align256(sizeof(unsigned) * 2 * (size_t)B) + sizeof(unsigned long long) * (size_t)B * ((size_t)n + m);
}

int forward(const float* xyz1, const float* xyz2, int B, int n, int m, float* dist1, float* dist2, int32_t* idx1,
            int32_t* idx2, void* workspace, unsigned long long* stats, int dirmask, hipStream_t stream) {
  PTOPS_REQUIRE(workspace != nullptr, "chamfer_forward(mfma): workspace required");
  if (((uintptr_t)workspace & 15) != 0) {
    ptops::set_error("chamfer_forward(mfma): workspace must be 16-byte aligned");
    return PTOPS_EALIGN;
  }
  const bool on[2] = {(dirmask & 1) != 0, (dirmask & 2) != 0};
  if (!on[0] && !on[1]) return PTOPS_OK;
  char* w = (char*)workspace;
  float* prep1 = (float*)w;
  w += align256(sizeof(float) * 4 * (size_t)B * pad64(n));
  float* prep2 = (float*)w;
  w += align256(sizeof(float) * 4 * (size_t)B * pad64(m));
  unsigned* cmax = (unsigned*)w;
  w += align256(sizeof(unsigned) * 2 * (size_t)B);
  unsigned long long* packed = (unsigned long long*)w;

  const PlanM p = make_plan(B, n, m);
  (void)ptops::fill_async(cmax, 0, sizeof(unsigned) * 2 * (size_t)B, stream);
  // direction 0 searches cloud 2, direction 1 searches cloud 1
  if (on[1]) {
    hipLaunchKernelGGL(chamfer_prep_kernel, dim3((unsigned)(pad64(n) / 256 + (pad64(n) % 256 != 0)), B), dim3(256), 0,
                       stream, xyz1, B, n, (int)pad64(n), prep1, cmax);
  }
  if (on[0]) {
    hipLaunchKernelGGL(chamfer_prep_kernel, dim3((unsigned)(pad64(m) / 256 + (pad64(m) % 256 != 0)), B), dim3(256), 0,
                       stream, xyz2, B, m, (int)pad64(m), prep2, cmax + B);
  }
  int rc = ptops::check_launch("chamfer_prep_kernel");
  if (rc != PTOPS_OK) return rc;
  const bool split[2] = {on[0] && p.splits[0] > 1, on[1] && p.splits[1] > 1};
  if (split[0]) (void)ptops::fill_async(packed, 0xFF, sizeof(unsigned long long) * (size_t)B * n, stream);
  if (split[1]) (void)ptops::fill_async(packed + (size_t)B * n, 0xFF, sizeof(unsigned long long) * (size_t)B * m, stream);

  ArgsM a;
  const long long blocks0 = on[0] ? (long long)B * p.qtiles[0] * p.splits[0] : 0;
  const long long blocks1 = on[1] ? (long long)B * p.qtiles[1] * p.splits[1] : 0;
  a.d[0] = DirM{xyz1, xyz2, prep2, cmax + B, dist1, idx1, packed, n, m, (int)pad64(m), p.qtiles[0], p.splits[0], p.sps[0], 0};
  a.d[1] = DirM{xyz2, xyz1, prep1, cmax, dist2, idx2, packed + (size_t)B * n, m, n, (int)pad64(n), p.qtiles[1],
                p.splits[1], p.sps[1], (int)blocks0};
  a.stats = stats;
  const long long blocks = blocks0 + blocks1;
  PTOPS_REQUIRE(blocks < (1LL << 31), "chamfer_forward(mfma): grid too large");
  hipLaunchKernelGGL(chamfer_mfma_kernel, dim3((unsigned)blocks), dim3(kTPB), 0, stream, a);
  rc = ptops::check_launch("chamfer_mfma_kernel");
  if (rc != PTOPS_OK) return rc;
  for (int d = 0; d < 2; ++d) {
    if (!split[d]) continue;
    const long long total = (long long)B * (d ? m : n);
    const int g = (int)(ptops::ceil_div_ll(total, 256) < 2048 ? ptops::ceil_div_ll(total, 256) : 2048);
    hipLaunchKernelGGL(chamfer_mfma_unpack_kernel, dim3(g), dim3(256), 0, stream, d ? packed + (size_t)B * n : packed,
                       d ? dist2 : dist1, d ? idx2 : idx1, total);
    rc = ptops::check_launch("chamfer_mfma_unpack_kernel");
  }
  return rc;
}

}  // namespace ptops_mfma
