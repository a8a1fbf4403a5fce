// Chamfer nearest-neighbour search with a bf16x3 MFMA filter, gfx950 (MI355X).
//
// Same operator and the same bit-exact results as chamfer.hip (reference: extensions/chamfer_dist/chamfer.cu:15-171,
// squared distance by direct differences, lowest index wins ties), a different way to get there.
//
// The brute-force kernel is at the VALU issue ceiling: the reference's arithmetic costs 6.8 vector instructions per
// pair.  Here the matrix pipe does the searching and the vector pipe only the bookkeeping:
//   * every coordinate is split into three bf16 pieces (x = h + m + l, exact to 2^-27); candidate c becomes one row of
//     an MFMA A operand holding the pieces of -2c and of |c|^2 (built by the search kernel itself while it stages a tile;
//     until late round 4 by a prep pass, kept behind PTOPS_CHAMFER_MFMA_PREP=1), a query q is one column of the B
//     operand holding its pieces and ones, arranged so that 21 k-slots sum every piece product down to
//     2^-18 |c||q|:  D[c][q] = |c|^2 - 2 c.q = |c - q|^2 - |q|^2 (products of bf16 pieces are exact in f32, the
//     accumulation is f32) at 16 pairs per cycle and SIMD on the matrix pipe, against 4.7 for the VALU kernel;
//   * D is only a FILTER.  Its error against the exact direct-difference value is bounded by E = 51 u (|c|max + |q|)^2
//     (u = 2^-24), so the true nearest neighbour (and every candidate that ties with it) has
//     D <= min D + 2E.  The MFMA shape is 32x32x16, two per 32 candidates x 32 queries (k-slots 0-15, then 16-31 on top
//     of the first result): a quarter of the vector-issue cost of the 16x16x32 shape per pair, which is what bounds
//     this kernel.  Each lane sees 16 of the 32 candidate rows; per 256-candidate superblock it folds its 128 values
//     with v_min3 (two values per instruction) and keeps the three smallest superblock minima and the ids of the first
//     two (v_med3 insertion).  At the end a lane re-evaluates, with the reference's exact arithmetic, only the one or
//     two superblocks whose minimum lies within 2E of the QUERY's best (the two lanes of a query compare notes first);
//     if a third one qualifies too (duplicated points, lattices) the lane scans its whole subset exactly.  The two lanes
//     that share a query merge (distance, index) lexicographically;
//   * the four waves of a workgroup search the same candidates for different queries: operand tiles go global -> LDS
//     once per workgroup (double buffered, one barrier per superblock), and workgroups of one cloud are placed on one
//     XCD so that its operand array stays in that XCD's L2.
// History (profiles/round2/chamfer_mfma_notes.txt): v_mfma_f32_16x16x4_f32 on (-2c, |c|^2) x (q, 1) is an exact f32
// chain, but MFMA and VALU time added up instead of overlapping (38 % matrix-pipe busy) and it was slower than brute
// force; 16x16x32 bf16 was bound by vector issue (8 of 16 cycles per MFMA + 2 v_min3) and by lane-divergent loads of
// the exact phase (every lane of a query re-evaluating its own best superblock).
// Both directions run in one launch; a direction with few queries splits its candidates over workgroups that merge
// through atomicMin on (dist_bits << 32 | index), as in chamfer.hip.
#include <math.h>
#include <stdlib.h>

#include "ptops_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kTPB = 256;    // 4 waves
constexpr int kQG = 2;       // 32-query groups per wave
constexpr int kQW = 32 * kQG;            // queries per wave
constexpr int kQB = kQW * (kTPB / 64);   // queries per workgroup
constexpr int kBlk = 32;     // candidates per MFMA pair
constexpr int kSBB = 4;      // blocks per superblock (the unit whose minimum is tracked and re-evaluated)
constexpr int kSBC = kBlk * kSBB;        // candidates per superblock (128)
constexpr int kTileSB = 2;   // superblocks per LDS tile (one barrier per tile)
constexpr int kTileC = kSBC * kTileSB;   // candidates per tile (256): clouds are padded to whole tiles
constexpr int kEntB = 96;    // 16-byte operand entries per block: 3 k-groups x 32 rows
constexpr int kEntT = kEntB * kSBB * kTileSB;   // 768 entries = 12 KB per tile: 3 per thread
constexpr float kBig = 3.0e38f;          // |c|^2 of a padding candidate: never the minimum, never overflows
constexpr float kU = 5.9604645e-8f;      // 2^-24
constexpr float kThrUlps = 128.0f;       // 2 (E + F) = 2 (51 + 5.01) u (|c|max + |q|)^2, rounded up

struct DirM {
  const float* q;          // [B,nq,3] queries
  const float* c;          // [B,nc,3] candidates (exact re-evaluation)
  const uint4* prep;       // [B,ncp/32,96] MFMA A operand entries of the candidates (8 bf16 each)
  float* dist;             // [B,nq]
  int32_t* idx;            // [B,nq]
  unsigned long long* ws;  // [B,nq] packed minima (splits > 1 only)
  int nq, nc, ncp;         // ncp = nc rounded up to kTileC
  int qtiles, splits, tiles_per_split, first_block;
  int vec_ok;              // candidate rows may be fetched four at a time as three 16-byte loads
};

struct ArgsM {
  DirM d[2];
  unsigned long long* stats;   // optional {EXTRA exact superblock evaluations, superblock visits} per direction
  int nblocks;
};

__device__ __forceinline__ float sqdist(float qx, float qy, float qz, float cx, float cy, float cz) {
  float dx = cx - qx, dy = cy - qy, dz = cz - qz;
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

__device__ __forceinline__ float min3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }

// x = h + m + l with bf16 pieces (round to nearest each time; the two subtractions are exact in f32)
struct Split3 {
  __bf16 h, m, l;
};
__device__ __forceinline__ Split3 split3(float x) {
  Split3 s;
  s.h = (__bf16)x;
  const float r1 = x - (float)s.h;
  s.m = (__bf16)r1;
  s.l = (__bf16)(r1 - (float)s.m);
  return s;
}

// The six piece products of one coordinate kept per k-group: (h,h) (h,m) (m,h) (h,l) (m,m) (l,h); slots 6 and 7 carry
// the pieces of |c|^2 against ones.  k-group 0: x and two pieces of |c|^2, 1: y and the third, 2: z.
__device__ __forceinline__ bf16x8 cand_operand(Split3 c, __bf16 n0, __bf16 n1) {
  bf16x8 v = {c.h, c.h, c.m, c.h, c.m, c.l, n0, n1};
  return v;
}
__device__ __forceinline__ bf16x8 query_operand(Split3 q, __bf16 one0, __bf16 one1) {
  bf16x8 v = {q.h, q.m, q.h, q.l, q.m, q.h, one0, one1};
  return v;
}

// A operand entries of all candidates: block of 32 candidates = 96 x 16 bytes, entry kg * 32 + row = the 8 bf16 of
// k-group kg of candidate row.  One thread per entry (coalesced 16-byte stores); blockIdx.z selects the cloud set.
struct PrepArgs {
  const float* xyz[2];
  uint4* prep[2];
  int n[2], np[2];
};

__global__ __launch_bounds__(256) void chamfer_prep_kernel(PrepArgs pa) {
  const int which = blockIdx.z, b = blockIdx.y;
  const float* __restrict__ xyz = which ? pa.xyz[1] : pa.xyz[0];
  uint4* __restrict__ prep = which ? pa.prep[1] : pa.prep[0];
  const int n = which ? pa.n[1] : pa.n[0], np = which ? pa.np[1] : pa.np[0];
  const int e = blockIdx.x * 256 + threadIdx.x;          // entry within the cloud
  const int per_cloud = np / kBlk * kEntB;
  if (prep == nullptr || e >= per_cloud) return;
  const int blk = e / kEntB, r = e - blk * kEntB;
  const int kg = r >> 5, j = blk * kBlk + (r & 31);
  float x = 0.f, y = 0.f, z = 0.f, w = kBig;
  if (j < n) {
    const float* p = xyz + ((size_t)b * n + j) * 3;
    x = p[0], y = p[1], z = p[2];
    w = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
  }
  const Split3 sn = split3(w);
  const __bf16 zero = (__bf16)0.0f;
  const Split3 sc = split3(-2.0f * (kg == 0 ? x : (kg == 1 ? y : z)));
  const bf16x8 v = cand_operand(sc, kg == 0 ? sn.h : (kg == 1 ? sn.l : zero), kg == 0 ? sn.m : zero);
  prep[(size_t)b * per_cloud + e] = __builtin_bit_cast(uint4, v);
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

// (kTPB, 4): at most 128 registers (4 workgroups per CU); it also makes hipcc write MFMA results to VGPRs.
// INLINE: no prep pass - thread t of the workgroup owns candidate t of a 256-candidate tile: it fetches the three
// coordinates (12 bytes instead of the 48 of a pre-split row), keeps them in three registers under the tile's math and splits
// them into the three operand entries right before the LDS store.  The entries are bit for bit those of chamfer_prep_kernel.
template <bool PREFETCH, bool INLINE>
__global__ __launch_bounds__(kTPB, 4) void chamfer_mfma_kernel(ArgsM a) {
  __shared__ uint4 tile[2][kEntT];   // two tiles of A operand entries, 12 KB each

  // Workgroups b and b + 8 share an XCD (round-robin dispatch): give every XCD one contiguous range of the job list,
  // so that the workgroups of one cloud (consecutive jobs) read its operand array through one L2.  Bijective for any
  // grid size; placement is a speed matter only.
  int bid;
  {
    const int orig = blockIdx.x, nwg = a.nblocks;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  }
  const bool second = bid >= a.d[1].first_block;
  const float* __restrict__ qptr = second ? a.d[1].q : a.d[0].q;
  const float* __restrict__ cptr = second ? a.d[1].c : a.d[0].c;
  const uint4* __restrict__ pptr = second ? a.d[1].prep : a.d[0].prep;
  float* __restrict__ dist = second ? a.d[1].dist : a.d[0].dist;
  int32_t* __restrict__ idx = second ? a.d[1].idx : a.d[0].idx;
  unsigned long long* __restrict__ ws = second ? a.d[1].ws : a.d[0].ws;
  const int nq = second ? a.d[1].nq : a.d[0].nq;
  const int nc = second ? a.d[1].nc : a.d[0].nc;
  const int ncp = second ? a.d[1].ncp : a.d[0].ncp;
  const int qtiles = second ? a.d[1].qtiles : a.d[0].qtiles;
  const int splits = second ? a.d[1].splits : a.d[0].splits;
  const int tps = second ? a.d[1].tiles_per_split : a.d[0].tiles_per_split;
  const bool vec_ok = (second ? a.d[1].vec_ok : a.d[0].vec_ok) != 0;
  int local = bid - (second ? a.d[1].first_block : a.d[0].first_block);
  const int s = local % splits;
  local /= splits;
  const int qt = local % qtiles;
  const int b = local / qtiles;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;
  const float* __restrict__ qb = qptr + (size_t)b * nq * 3;
  const float* __restrict__ cb = cptr + (size_t)b * nc * 3;
  const int ntiles = ncp / kTileC;
  const int tl_begin = s * tps;
  const int tl_end = min(ntiles, tl_begin + tps);
  const int sb_begin = tl_begin * kTileSB, sb_end = tl_end * kTileSB;
  const int qbase = qt * kQB + wave * kQW;

  // B operands: lane (col, half) holds k-group `half` of query col for the first MFMA and k-group 2 + half for the
  // second one (k-group 3 does not exist: zeros)
  bf16x8 bq1[kQG], bq2[kQG];
  {
    const __bf16 one = (__bf16)1.0f, zero = (__bf16)0.0f;
#pragma unroll
    for (int g = 0; g < kQG; ++g) {
      const int j = min(qbase + g * 32 + col, nq - 1);
      const float v1 = qb[(size_t)j * 3 + half];          // x (k-group 0) or y (k-group 1)
      const float v2 = qb[(size_t)j * 3 + 2];             // z (k-group 2)
      bq1[g] = query_operand(split3(v1), one, half == 0 ? one : zero);
      bq2[g] = query_operand(split3(v2), zero, zero);
      if (half == 1) bq2[g] = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
    }
  }
  float m1[kQG], m2[kQG], m3[kQG];
  int b1[kQG], b2[kQG];
#pragma unroll
  for (int g = 0; g < kQG; ++g) {
    m1[g] = m2[g] = m3[g] = INFINITY;
    b1[g] = b2[g] = sb_begin;
  }

  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const uint4 zero_u4 = make_uint4(0u, 0u, 0u, 0u);
  // thread t stages entries t, t + 256 and t + 512 of a tile (INLINE: the three k-group entries of candidate t of the tile)
  const uint4* __restrict__ gp = INLINE ? nullptr : pptr + (size_t)b * (ncp / kBlk * kEntB) + (size_t)tl_begin * kEntT + threadIdx.x;
  // INLINE: candidate t of a tile sits in block t / 32, row t % 32: entries (t / 32) * 96 + kg * 32 + t % 32
  const int own = (threadIdx.x >> 5) * kEntB + (threadIdx.x & 31);
  auto fetch_xyz = [&](int tl, float& x, float& y, float& z) {
    const int j = tl * kTileC + threadIdx.x;
    const int jj = j < nc ? j : 0;
    x = cb[(size_t)jj * 3 + 0];
    y = cb[(size_t)jj * 3 + 1];
    z = cb[(size_t)jj * 3 + 2];
  };
  auto split_store = [&](int buf, int tl, float x, float y, float z) {
    const bool pad = tl * kTileC + (int)threadIdx.x >= nc;
    if (pad) x = y = z = 0.f;
    const float w = pad ? kBig : __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
    const Split3 sn = split3(w);
    const __bf16 zero = (__bf16)0.0f;
    tile[buf][own] = __builtin_bit_cast(uint4, cand_operand(split3(-2.0f * x), sn.h, sn.m));
    tile[buf][own + 32] = __builtin_bit_cast(uint4, cand_operand(split3(-2.0f * y), sn.l, zero));
    tile[buf][own + 64] = __builtin_bit_cast(uint4, cand_operand(split3(-2.0f * z), zero, zero));
  };
  if (INLINE) {
    float x, y, z;
    fetch_xyz(tl_begin, x, y, z);
    split_store(0, tl_begin, x, y, z);
  } else {
    const uint4 e0 = gp[0], e1 = gp[256], e2 = gp[512];
    tile[0][threadIdx.x] = e0;
    tile[0][threadIdx.x + 256] = e1;
    tile[0][threadIdx.x + 512] = e2;
  }
  __syncthreads();
  for (int tl = tl_begin; tl < tl_end; ++tl) {
    const int cur = (tl - tl_begin) & 1;
    const bool more = tl + 1 < tl_end;
    uint4 e0 = zero_u4, e1 = zero_u4, e2 = zero_u4;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (more) {                                       // next tile: global -> registers now, -> LDS after the math
      if (INLINE) {
        fetch_xyz(tl + 1, nx, ny, nz);
      } else {
        gp += kEntT;
        e0 = gp[0];
        e1 = gp[256];
        e2 = gp[512];
      }
    }
    if (PREFETCH) {
      // The same work with the fragments of the NEXT block pair requested right behind the MFMAs of the current one, so
      // that their LDS latency passes under the 32 v_min3 of the fold instead of in front of the next MFMAs (two named
      // register sets, the two block pairs of a superblock unrolled by hand: no copies).  Only the first pair of a tile
      // is read unprefetched (its tile became visible at the barrier just passed).
      auto frags = [&](int pair, bf16x8& x10, bf16x8& x20, bf16x8& x11, bf16x8& x21) {
        const uint4* t0 = &tile[cur][pair * 2 * kEntB];
        const uint4* t1 = t0 + kEntB;
        x10 = __builtin_bit_cast(bf16x8, t0[lane]);
        x20 = __builtin_bit_cast(bf16x8, t0[64 + col]);
        x11 = __builtin_bit_cast(bf16x8, t1[lane]);
        x21 = __builtin_bit_cast(bf16x8, t1[64 + col]);
      };
      auto products = [&](const bf16x8& x10, const bf16x8& x20, const bf16x8& x11, const bf16x8& x21, f32x16 (&r)[2][kQG]) {
#pragma unroll
        for (int g = 0; g < kQG; ++g) {
          r[0][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x10, bq1[g], zero16, 0, 0, 0);
          r[1][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x11, bq1[g], zero16, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < kQG; ++g) {
          r[0][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x20, bq2[g], r[0][g], 0, 0, 0);
          r[1][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x21, bq2[g], r[1][g], 0, 0, 0);
        }
      };
      auto fold = [&](const f32x16 (&r)[2][kQG], float (&v)[kQG]) {
#pragma unroll
        for (int g = 0; g < kQG; ++g) {
          float wa = min3(r[0][g][0], r[0][g][1], r[0][g][2]);
          float wb = min3(r[1][g][0], r[1][g][1], r[1][g][2]);
#pragma unroll
          for (int i = 3; i < 15; i += 2) {
            wa = min3(wa, r[0][g][i], r[0][g][i + 1]);
            wb = min3(wb, r[1][g][i], r[1][g][i + 1]);
          }
          v[g] = min3(v[g], wa, wb);
          v[g] = min3(v[g], r[0][g][15], r[1][g][15]);
        }
      };
      bf16x8 a10, a20, a11, a21, b10, b20, b11, b21;
      frags(0, a10, a20, a11, a21);
#pragma unroll 1
      for (int h = 0; h < kTileSB; ++h) {
        const int sb = tl * kTileSB + h;
        float v[kQG];
#pragma unroll
        for (int g = 0; g < kQG; ++g) v[g] = INFINITY;
        f32x16 r[2][kQG];
        products(a10, a20, a11, a21, r);
        frags(2 * h + 1, b10, b20, b11, b21);                 // second pair of this superblock
        __builtin_amdgcn_sched_barrier(0);
        fold(r, v);
        // hipcc would hoist the next eight MFMAs above this fold (second result set, spills in the loop): make their
        // operands nominally depend on the fold's result.  Costs no instruction.
        asm volatile("" : "+v"(b10), "+v"(b20), "+v"(b11), "+v"(b21) : "v"(v[0]), "v"(v[1]));
        products(b10, b20, b11, b21, r);
        if (h + 1 < kTileSB) frags(2 * h + 2, a10, a20, a11, a21);   // first pair of the next superblock of this tile
        __builtin_amdgcn_sched_barrier(0);
        fold(r, v);
        asm volatile("" : "+v"(a10), "+v"(a20), "+v"(a11), "+v"(a21) : "v"(v[0]), "v"(v[1]));
#pragma unroll
        for (int g = 0; g < kQG; ++g) {
          const float o1 = m1[g], o2 = m2[g], o3 = m3[g];
          const int ob1 = b1[g], ob2 = b2[g];
          const bool lt1 = v[g] < o1, lt2 = v[g] < o2;
          const int t2 = lt2 ? sb : ob2;
          m3[g] = __builtin_amdgcn_fmed3f(o2, o3, v[g]);
          m2[g] = __builtin_amdgcn_fmed3f(o1, o2, v[g]);
          m1[g] = lt1 ? v[g] : o1;
          b2[g] = lt1 ? ob1 : t2;
          b1[g] = lt1 ? sb : ob1;
        }
      }
    } else {
  #pragma unroll 1
      for (int h = 0; h < kTileSB; ++h) {
        const int sb = tl * kTileSB + h;
        float v[kQG];
  #pragma unroll
        for (int g = 0; g < kQG; ++g) v[g] = INFINITY;
  #pragma unroll 1
        for (int blk = 0; blk < kSBB; blk += 2) {   // not unrolled: unrolled, hipcc hoists all the MFMAs and spills their results
          // Two blocks x two query groups = eight MFMAs issued back to back: the four that start an accumulation, then
          // the four that finish one (each 128 cycles behind the MFMA it depends on).  Left to itself hipcc reuses ONE
          // result register block and serialises MFMA -> MFMA -> 8 dependent v_min3 (matrix pipe 57 % busy).
          const uint4* t0 = &tile[cur][(h * kSBB + blk) * kEntB];
          const uint4* t1 = t0 + kEntB;
          const bf16x8 a10 = __builtin_bit_cast(bf16x8, t0[lane]);           // k-groups 0 | 1
          const bf16x8 a20 = __builtin_bit_cast(bf16x8, t0[64 + col]);       // k-group 2 for both halves: the upper half
          const bf16x8 a11 = __builtin_bit_cast(bf16x8, t1[lane]);           //   meets zeros in bq2
          const bf16x8 a21 = __builtin_bit_cast(bf16x8, t1[64 + col]);
          f32x16 r[2][kQG];
  #pragma unroll
          for (int g = 0; g < kQG; ++g) {
            r[0][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a10, bq1[g], zero16, 0, 0, 0);
            r[1][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a11, bq1[g], zero16, 0, 0, 0);
          }
  #pragma unroll
          for (int g = 0; g < kQG; ++g) {
            r[0][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a20, bq2[g], r[0][g], 0, 0, 0);
            r[1][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a21, bq2[g], r[1][g], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
          for (int g = 0; g < kQG; ++g) {
            float wa = min3(r[0][g][0], r[0][g][1], r[0][g][2]);
            float wb = min3(r[1][g][0], r[1][g][1], r[1][g][2]);
  #pragma unroll
            for (int i = 3; i < 15; i += 2) {
              wa = min3(wa, r[0][g][i], r[0][g][i + 1]);
              wb = min3(wb, r[1][g][i], r[1][g][i + 1]);
            }
            v[g] = min3(v[g], wa, wb);
            v[g] = min3(v[g], r[0][g][15], r[1][g][15]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
  #pragma unroll
        for (int g = 0; g < kQG; ++g) {
          // sorted insertion of (v, sb) into the three smallest superblock minima (v_med3 x2, 2 compares, 4 selects)
          const float o1 = m1[g], o2 = m2[g], o3 = m3[g];
          const int ob1 = b1[g], ob2 = b2[g];
          const bool lt1 = v[g] < o1, lt2 = v[g] < o2;
          const int t2 = lt2 ? sb : ob2;
          m3[g] = __builtin_amdgcn_fmed3f(o2, o3, v[g]);
          m2[g] = __builtin_amdgcn_fmed3f(o1, o2, v[g]);
          m1[g] = lt1 ? v[g] : o1;
          b2[g] = lt1 ? ob1 : t2;
          b1[g] = lt1 ? sb : ob1;
        }
      }
    }
    if (more) {
      if (INLINE) {
        split_store(cur ^ 1, tl + 1, nx, ny, nz);
      } else {
        tile[cur ^ 1][threadIdx.x] = e0;
        tile[cur ^ 1][threadIdx.x + 256] = e1;
        tile[cur ^ 1][threadIdx.x + 512] = e2;
      }
    }
    __syncthreads();
  }

  // ---- exact phase: the reference's arithmetic on the superblocks that can hold the nearest neighbour
  unsigned long long n_exact = 0;
#pragma unroll
  for (int g = 0; g < kQG; ++g) {      // unrolled: m1[g] ... b2[g] stay in registers
    const int jq = qbase + g * 32 + col;
    const int j = min(jq, nq - 1);
    const float qx = qb[(size_t)j * 3 + 0], qy = qb[(size_t)j * 3 + 1], qz = qb[(size_t)j * 3 + 2];
    // Error bound of the filter for THIS query.  Only two candidates enter the argument: the true
    // nearest neighbour j* and the filter's minimiser jm; both lie within d(q, jm) of q, and d(q, jm)^2 is the filter
    // minimum plus |q|^2 up to the filter error itself, so |c| + |q| <= R := 1.01 (2 |q| + sqrt(max(gmin + |q|^2, 0))).
    const float q2 = __builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx));
    const float gmin = __builtin_fminf(m1[g], __shfl_xor(m1[g], 32));          // over BOTH lanes of the query
    const float R = 1.01f * (2.0f * sqrtf(q2) + sqrtf(__builtin_fmaxf(gmin + q2, 0.0f)));
    const float thr = (kThrUlps * kU) * R * R;
    // a superblock whose filter minimum is above the query's best by more than the bound holds nothing
    const float lim = gmin + thr;
    float best = INFINITY;
    int besti = 0x7fffffff;
    // candidates of one rescan come in ascending index: strict < keeps the lowest index inside it (as in chamfer.hip);
    // only the merge with what an earlier rescan found compares (distance, index) lexicographically
    auto rescan = [&](int sbi) {   // this lane's rows of superblock sbi: per block, rows 8 t + 4 half + (0..3), t = 0..3
      float ld = INFINITY;
      int lk = 0x7fffffff;
      auto consider = [&](int k, float cx, float cy, float cz) {
        const float d = sqdist(qx, qy, qz, cx, cy, cz);
        const bool better = d < ld;
        ld = better ? d : ld;
        lk = better ? k : lk;
      };
#pragma unroll 2
      for (int blk = 0; blk < kSBB; ++blk) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int k0 = sbi * kSBC + blk * kBlk + 8 * t + 4 * half;
          if (vec_ok && k0 + 3 < nc) {
            const float4* p = reinterpret_cast<const float4*>(cb + (size_t)k0 * 3);
            const float4 u0 = p[0], u1 = p[1], u2 = p[2];
            consider(k0, u0.x, u0.y, u0.z);
            consider(k0 + 1, u0.w, u1.x, u1.y);
            consider(k0 + 2, u1.z, u1.w, u2.x);
            consider(k0 + 3, u2.y, u2.z, u2.w);
          } else {
            for (int r = 0; r < 4; ++r) {
              const int k = k0 + r;
              if (k < nc) consider(k, cb[(size_t)k * 3 + 0], cb[(size_t)k * 3 + 1], cb[(size_t)k * 3 + 2]);
            }
          }
        }
      }
      const bool better = ld < best || (ld == best && lk < besti);
      best = better ? ld : best;
      besti = better ? lk : besti;
    };
    // A third superblock within the bound (duplicated points, lattices, a handful of queries in any large cloud):
    // the whole wave scans the query's candidate range exactly, 64 candidates per step with coalesced loads — a lane
    // doing it alone would keep its wave alive for a millisecond.  Written so that a NaN / infinite bound lands here.
    const bool mine = !(m3[g] > lim);
    const int partner_mine = __shfl_xor((int)mine, 32);   // outside the ||: a short-circuited shuffle would read an inactive lane
    const bool need_full = mine || partner_mine != 0;
    {
      if (!need_full) {
        if (!(m1[g] > lim)) rescan(b1[g]);
        if (!(m2[g] > lim)) {
          rescan(b2[g]);
          ++n_exact;
        }
      }
      const int c_begin = tl_begin * kTileC, c_end = min(nc, tl_end * kTileC);
      const unsigned long long todo = __ballot(need_full);
      for (int src = 0; src < 32; ++src) {                         // lanes 0..31 are the half == 0 lanes, one per query
        if (((todo >> src) & 1ull) == 0) continue;                  // wave-uniform
        const float fx = __shfl(qx, src), fy = __shfl(qy, src), fz = __shfl(qz, src);
        float bd = INFINITY;
        int bi = 0x7fffffff;
        for (int k = c_begin + lane; k < c_end; k += 64) {         // ascending k per lane: strict < keeps the lowest index
          const float d = sqdist(fx, fy, fz, cb[(size_t)k * 3 + 0], cb[(size_t)k * 3 + 1], cb[(size_t)k * 3 + 2]);
          const bool better = d < bd;
          bd = better ? d : bd;
          bi = better ? k : bi;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const float od = __shfl_xor(bd, off);
          const int oi = __shfl_xor(bi, off);
          const bool better = od < bd || (od == bd && oi < bi);
          bd = better ? od : bd;
          bi = better ? oi : bi;
        }
        if (lane == src) {
          best = bd;
          besti = bi;
        }
        if (lane == 0) n_exact += (unsigned long long)(sb_end - sb_begin);
      }
    }
    // the two lanes of a query: lexicographic minimum of (distance, index)
    {
      const float od = __shfl_xor(best, 32);
      const int oi = __shfl_xor(besti, 32);
      const bool better = od < best || (od == best && oi < besti);
      best = better ? od : best;
      besti = better ? oi : besti;
    }
    if (half == 0 && jq < nq) {
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
  int qtiles[2], splits[2], tps[2];
  bool need_ws;
};

PlanM make_plan(int B, int n, int m) {
  PlanM p;
  const int nq[2] = {n, m}, nc[2] = {m, n};
  p.need_ws = false;
  for (int d = 0; d < 2; ++d) {
    p.qtiles[d] = ptops::ceil_div(nq[d], kQB);
    const long long tiles = (long long)B * p.qtiles[d];
    const int ntl = ptops::ceil_div(nc[d], kTileC);
    const int target = 8 * ptops::kNumCU;                        // 8 workgroups (32 waves) per CU
    long long s = tiles >= target / 2 ? 1 : target / (tiles > 0 ? tiles : 1);
    const int max_s = ntl > 0 ? ntl : 1;                         // >= 256 candidates (one tile) per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    int tps = ptops::ceil_div(ntl, (int)s);
    s = ptops::ceil_div(ntl, tps);                               // drop empty splits
    p.splits[d] = (int)s;
    p.tps[d] = tps;
    if (s > 1) p.need_ws = true;
  }
  return p;
}

inline size_t padsb(int n) { return ((size_t)n + kTileC - 1) / kTileC * kTileC; }   // candidates padded to whole tiles
constexpr size_t kPrepBytesPerCand = 48;   // 3 k-groups x 8 bf16
inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

namespace ptops_mfma {

bool eligible(int B, int n, int m) { return B >= 1 && n >= 1 && m >= 1; }

// [prep cloud 1 | prep cloud 2 | packed minima B*(n+m) u64]
size_t workspace_bytes(int B, int n, int m) {
  return align256(kPrepBytesPerCand * (size_t)B * padsb(n)) + align256(kPrepBytesPerCand * (size_t)B * padsb(m)) +
         sizeof(unsigned long long) * (size_t)B * ((size_t)n + m);
}

int forward(const float* xyz1, const float* xyz2, int B, int n, int m, float* dist1, float* dist2, int32_t* idx1,
            int32_t* idx2, void* workspace, unsigned long long* stats, int dirmask, hipStream_t stream) {
  PTOPS_REQUIRE(workspace != nullptr, "chamfer_forward(mfma): workspace required");
  if (((uintptr_t)workspace & 15) != 0) {
    ptops::set_error("chamfer_forward(mfma): workspace must be 16-byte aligned");
    return PTOPS_EALIGN;
  }
  PTOPS_REQUIRE(B <= 65535, "chamfer_forward(mfma): B too large");
  const bool on[2] = {(dirmask & 1) != 0, (dirmask & 2) != 0};
  if (!on[0] && !on[1]) return PTOPS_OK;
  char* w = (char*)workspace;
  uint4* prep1 = (uint4*)w;
  w += align256(kPrepBytesPerCand * (size_t)B * padsb(n));
  uint4* prep2 = (uint4*)w;
  w += align256(kPrepBytesPerCand * (size_t)B * padsb(m));
  unsigned long long* packed = (unsigned long long*)w;

  const PlanM p = make_plan(B, n, m);
  // PTOPS_CHAMFER_MFMA_PREP=1: the round-2 form with the candidates pre-split by a pass of their own (50 MB written and read
  // back per 16384 x 16384 x 32 launch); default: the search kernel splits the coordinates itself
  const bool inline_split = !ptops::option("PTOPS_CHAMFER_MFMA_PREP") && !ptops::option("PTOPS_CHAMFER_MFMA_PREFETCH");
  int rc = PTOPS_OK;
  if (!inline_split) {  // direction 0 searches cloud 2, direction 1 searches cloud 1: operands of the clouds that are searched, one launch
    PrepArgs pa;
    pa.xyz[0] = xyz1, pa.xyz[1] = xyz2;
    pa.prep[0] = on[1] ? prep1 : nullptr, pa.prep[1] = on[0] ? prep2 : nullptr;
    pa.n[0] = n, pa.n[1] = m;
    pa.np[0] = (int)padsb(n), pa.np[1] = (int)padsb(m);
    const size_t e0 = on[1] ? padsb(n) / kBlk * kEntB : 0, e1 = on[0] ? padsb(m) / kBlk * kEntB : 0;
    const size_t emax = e0 > e1 ? e0 : e1;
    hipLaunchKernelGGL(chamfer_prep_kernel, dim3((unsigned)((emax + 255) / 256), B, 2), dim3(256), 0, stream, pa);
    rc = ptops::check_launch("chamfer_prep_kernel");
    if (rc != PTOPS_OK) return rc;
  }
  const bool split[2] = {on[0] && p.splits[0] > 1, on[1] && p.splits[1] > 1};
  if (split[0]) (void)ptops::fill_async(packed, 0xFF, sizeof(unsigned long long) * (size_t)B * n, stream);
  if (split[1]) (void)ptops::fill_async(packed + (size_t)B * n, 0xFF, sizeof(unsigned long long) * (size_t)B * m, stream);

  ArgsM a;
  const long long blocks0 = on[0] ? (long long)B * p.qtiles[0] * p.splits[0] : 0;
  const long long blocks1 = on[1] ? (long long)B * p.qtiles[1] * p.splits[1] : 0;
  // four candidate rows as three 16-byte loads: every cloud of the batch must start on a 16-byte boundary
  const int vec2 = (((uintptr_t)xyz2 & 15) == 0 && m % 4 == 0) ? 1 : 0, vec1 = (((uintptr_t)xyz1 & 15) == 0 && n % 4 == 0) ? 1 : 0;
  a.d[0] = DirM{xyz1, xyz2, prep2, dist1, idx1, packed, n, m, (int)padsb(m), p.qtiles[0], p.splits[0], p.tps[0], 0, vec2};
  a.d[1] = DirM{xyz2, xyz1, prep1, dist2, idx2, packed + (size_t)B * n, m, n, (int)padsb(n), p.qtiles[1],
                p.splits[1], p.tps[1], (int)blocks0, vec1};
  a.stats = stats;
  a.nblocks = (int)(blocks0 + blocks1);
  const long long blocks = blocks0 + blocks1;
  PTOPS_REQUIRE(blocks < (1LL << 31), "chamfer_forward(mfma): grid too large");
  // PTOPS_CHAMFER_MFMA_PREFETCH=1: the fragment-prefetch variant of the inner loop.  Measured SLOWER (B = 32, 16384^2:
  // 1.62 vs 0.94 ms, profiles/round3/chamfer_mfma_ab_v1.log): the second fragment set does not fit the 128 registers
  // of four workgroups per CU and hipcc spills inside the loop; kept switchable for the record.
  if (ptops::option("PTOPS_CHAMFER_MFMA_PREFETCH"))
    hipLaunchKernelGGL((chamfer_mfma_kernel<true, false>), dim3((unsigned)blocks), dim3(kTPB), 0, stream, a);
  else if (inline_split)
    hipLaunchKernelGGL((chamfer_mfma_kernel<false, true>), dim3((unsigned)blocks), dim3(kTPB), 0, stream, a);
  else
    hipLaunchKernelGGL((chamfer_mfma_kernel<false, false>), dim3((unsigned)blocks), dim3(kTPB), 0, stream, a);
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
