// fp32 multi-head attention (forward + backward) on the f32 MFMA path of gfx950 (MI355X).
//
// Replaces the softmax(q k^T * scale [+ block mask]) v chain of Attention / CrossAttention
// (models/Transformer_utils.py:89-159 of the reference) for the shapes of the AdaPoinTr path:
// head_dim 64, token counts that are multiples of 32 (256 / 512 / 576), and the denoise mask of
// base_blocks.py:224-229, which is block-aligned (the first `split_q` queries see only the first
// `limit_k` keys; the remaining queries see every key) and therefore needs no per-element masking.
//
// Everything is exact fp32: v_mfma_f32_32x32x2_f32 is bit-for-bit an fmaf chain, the softmax is the
// usual online (running max / running sum) form.
//
// Forward, one wave per 32-query tile, W waves per workgroup sharing LDS K/V tiles of 32 keys:
//   S^T = K Q^T   : A = K tile from LDS (lane (i,h): key i, dims 32h..32h+31 -> 8 ds_read_b128, rows
//                   padded to 68 dwords so that the 16-lane read groups hit 16 distinct 16-byte slots),
//                   B = Q^T held in 32 VGPRs for the whole kernel (pre-scaled by scale*log2 e).
//                   The contraction order over the 64 dims is permuted (half h of the wave covers dims
//                   32h+s) so that both operands are contiguous per lane.
//   softmax       : with S^T the query sits on the lane and its 32 keys in 16 registers x 2 lane
//                   halves, so row max / row sum are register reductions plus ONE cross-half exchange.
//   O^T = V^T P   : B = P is the S^T accumulator itself (k-step r consumes the keys that register r
//                   holds in the two halves: no data movement), A = V^T from LDS (conflict-free
//                   ds_read_b32 across dims).
#include <math.h>

#include <stdlib.h>

#include "ptops_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kD = 64;        // head dim
constexpr int kTK = 32;       // keys per LDS tile
constexpr int kLd = 68;       // padded row stride (dwords) of the LDS tiles

struct AttnArgs {
  const float* q;  // element (b,h,n,c) at q + b*q_sb + h*q_sh + n*q_sn + c
  const float* k;
  const float* v;
  float* o;        // [B, Nq, H, 64] contiguous
  float* lse;      // [B, H, Nq]  (base-2 log-sum-exp of the scaled scores)
  long long q_sb, q_sh, q_sn, k_sb, k_sh, k_sn, v_sb, v_sh, v_sn;
  int H, Nq, Nk;
  int split_q, limit_k;  // queries < split_q attend to keys < limit_k; the others to all Nk
  float scale_log2e;
  int xcd_remap;         // split3 kernels: XCD-contiguous job order (xcd_job below); 0 = plain blockIdx order (A/B)
};

__device__ __forceinline__ float other_half(float x) {
  // value held by lane ^ 32
  return __shfl_xor(x, 32);
}

// __launch_bounds__(.., 3): at most 168 VGPR+AGPR, i.e. three waves per SIMD.
template <int W>
__global__ __launch_bounds__(W * 64, 3) void attn_fwd_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float sK[2][kTK * kLd];
  __shared__ __attribute__((aligned(16))) float sV[2][kTK * kLd];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.H, head = bh % a.H;
  const int q0 = (blockIdx.x * W + wave) * 32;
  const bool active = q0 < a.Nq;  // wave-uniform
  const float* qb = a.q + b * a.q_sb + head * a.q_sh;
  const float* kb = a.k + b * a.k_sb + head * a.k_sh;
  const float* vb = a.v + b * a.v_sb + head * a.v_sh;

  // keys this workgroup has to walk: the maximum over its waves' tiles (block-uniform)
  const int blk_q0 = blockIdx.x * W * 32;
  const int blk_q1 = min(blk_q0 + W * 32, a.Nq);
  const int blk_keys = (blk_q1 > a.split_q) ? a.Nk : a.limit_k;
  const int my_keys = active ? ((q0 >= a.split_q) ? a.Nk : a.limit_k) : 0;

  // B operand of S^T = K Q^T: lane (j,h) holds Q[q0+j][32h + s], s = 0..31, pre-scaled
  float qf[32];
  if (active) {
    const float* qp = qb + (long long)(q0 + j) * a.q_sn + 32 * h;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(qp + 4 * t);
      qf[4 * t + 0] = v4.x * a.scale_log2e;
      qf[4 * t + 1] = v4.y * a.scale_log2e;
      qf[4 * t + 2] = v4.z * a.scale_log2e;
      qf[4 * t + 3] = v4.w * a.scale_log2e;
    }
  } else {
#pragma unroll
    for (int t = 0; t < 32; ++t) qf[t] = 0.0f;
  }

  f32x16 o0 = {0}, o1 = {0};  // O^T accumulators: dims 0..31 and 32..63, query on the lane
  float m_run = -INFINITY, l_run = 0.0f;

  // Cooperative tile staging, split in two halves so that the global loads of tile t+1 are in
  // flight while tile t is being multiplied: fetch -> registers before the MFMAs, registers -> LDS after.
  constexpr int kPerThread = (kTK * (kD / 4) + W * 64 - 1) / (W * 64);  // float4 per thread per matrix
  f32x4 kreg[kPerThread], vreg[kPerThread];
  auto fetch_tile = [&](int key0) {
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
      const int e = tid + u * W * 64;
      if (e < kTK * (kD / 4)) {
        const int r = e >> 4, c4 = (e & 15) * 4;
        kreg[u] = *reinterpret_cast<const f32x4*>(kb + (long long)(key0 + r) * a.k_sn + c4);
        vreg[u] = *reinterpret_cast<const f32x4*>(vb + (long long)(key0 + r) * a.v_sn + c4);
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
      const int e = tid + u * W * 64;
      if (e < kTK * (kD / 4)) {
        const int r = e >> 4, c4 = (e & 15) * 4;
        *reinterpret_cast<f32x4*>(&sK[buf][r * kLd + c4]) = kreg[u];
        *reinterpret_cast<f32x4*>(&sV[buf][r * kLd + c4]) = vreg[u];
      }
    }
  };

  const int ntiles = blk_keys / kTK;
  if (ntiles > 0) {
    fetch_tile(0);
    store_tile(0);
  }
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) fetch_tile((t + 1) * kTK);  // in flight during the MFMAs below
    if (t * kTK < my_keys) {
      // ---- S^T = K Q^T ------------------------------------------------------------------
      f32x16 s = {0};
      const float* kr = &sK[buf][j * kLd + 32 * h];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const f32x4 k4 = *reinterpret_cast<const f32x4*>(kr + 4 * g);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.x, qf[4 * g + 0], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.y, qf[4 * g + 1], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.z, qf[4 * g + 2], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.w, qf[4 * g + 3], s, 0, 0, 0);
      }
      // ---- online softmax (query = lane&31; this half holds 16 of the tile's 32 keys) ------
      float mt = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
      mt = fmaxf(mt, other_half(mt));
      const float m_new = fmaxf(m_run, mt);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float psum = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
        psum += s[r];
      }
      l_run = l_run * alpha + psum;  // per-half partial sum; halves are added at the end
      m_run = m_new;
      o0 *= alpha;
      o1 *= alpha;
      // ---- O^T += V^T P : k-step r uses keys (r&3)+8(r>>2) [half 0] and +4 [half 1] ---------
      const float* vr = &sV[buf][(4 * h) * kLd + j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2);
        const float v0 = vr[key * kLd], v1 = vr[key * kLd + 32];
        o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[r], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[r], o1, 0, 0, 0);
      }
    }
    if (t + 1 < ntiles) store_tile(buf ^ 1);
    __syncthreads();
  }

  if (active) {
    const float l_tot = l_run + other_half(l_run);
    const float inv = 1.0f / l_tot;
    // accumulator layout: column = query (lane&31), row(reg) = (reg&3) + 8*(reg>>2) + 4*h = dim
    float* op = a.o + (((long long)b * a.Nq + q0 + j) * a.H + head) * kD;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 w0 = {o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
      f32x4 w1 = {o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
      *reinterpret_cast<f32x4*>(op + 8 * g + 4 * h) = w0;
      *reinterpret_cast<f32x4*>(op + 32 + 8 * g + 4 * h) = w1;
    }
    if (h == 0) a.lse[((long long)b * a.H + head) * a.Nq + q0 + j] = m_run + __builtin_amdgcn_logf(l_tot);
  }
}


// ------------------------------------------------------------------------------------------------
// Forward on the bf16 matrix pipe with fp32 accuracy (round 2).  Same algorithm, tiling and outputs as attn_fwd_kernel;
// every fp32 operand is split into three bf16 pieces (x = h + m + l, exact to 2^-27) and each contraction issues the six
// piece products down to 2^-18 on v_mfma_f32_32x32x16_bf16 (see csrc/gemm_split3.hip for the arithmetic argument):
// 48 MFMAs of 32 cycles per 32 x 32 tile instead of 64 of 64 cycles.
//   S^T = K Q^T : A = K pieces from LDS ([piece][16-dim chunk][k-half][key], rows padded to 33 entries), B = Q pieces
//                 (pre-scaled, split once, 48 VGPRs);
//   O^T = V^T P : B = the P pieces, split on the fly from the S^T accumulator: k-step s takes registers 8s..8s+7, whose
//                 element jj of lane half h is key 16s + 8 (jj >> 2) + 4h + (jj & 3); A = V^T pieces from LDS, staged in
//                 exactly that key order ([piece][k-step][k-half][dim]).
// K and V are split by the 256 staging threads (one 8-value operand entry each) on their way into LDS.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Pieces3 {
  uint4 h, m, l;
};

__device__ __forceinline__ Pieces3 split8(const float (&x)[8]) {
  bf16x8 h, m, l;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 a = (__bf16)x[i];
    const float r1 = x[i] - (float)a;
    const __bf16 b = (__bf16)r1;
    const __bf16 c = (__bf16)(r1 - (float)b);
    h[i] = a;
    m[i] = b;
    l[i] = c;
  }
  Pieces3 p;
  p.h = __builtin_bit_cast(uint4, h);
  p.m = __builtin_bit_cast(uint4, m);
  p.l = __builtin_bit_cast(uint4, l);
  return p;
}

// Workgroups b and b + 8 share an XCD (round-robin dispatch, speed only): hand every XCD one contiguous range of the job
// list (job = head-major, tile-minor), so that the workgroups of one (batch, head) read its K / V (or Q / dO) tiles through
// ONE L2 instead of up to five.  Bijective for any grid; placement never affects results.
__device__ __forceinline__ void xcd_job(int remap, int& bx, int& bh) {
  if (!remap) {
    bx = blockIdx.x, bh = blockIdx.y;
    return;
  }
  const int gx = gridDim.x, nwg = gx * gridDim.y;
  const int orig = blockIdx.x + gx * blockIdx.y;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int job = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  bh = job / gx;
  bx = job - bh * gx;
}

// the six piece products, smallest first: (piece of A, piece of B)
#define PTOPS_SPLIT3_TERMS(F) F(2, 0) F(1, 1) F(0, 2) F(1, 0) F(0, 1) F(0, 0)

// ABL != 0: timing experiments only (benchmarks/attention_ablation.py; results are WRONG): bit 0 = K / V staged with one
// conversion per pair instead of the three-piece split, bit 1 = P likewise, bit 2 = no softmax arithmetic, bit 3 = no
// S^T MFMAs, bit 4 = no O^T MFMAs, bit 5 = tiles after the first neither loaded nor staged.
__device__ __forceinline__ Pieces3 cheap8(const float (&x)[8]) {
  bf16x8 h;
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = (__bf16)x[i];
  Pieces3 p;
  p.h = p.m = p.l = __builtin_bit_cast(uint4, h);
  return p;
}

template <int ABL, bool LATE = true, bool PRIO = false>
__global__ __launch_bounds__(256, 3) void attn_fwd_split3_kernel(AttnArgs a) {
  constexpr int W = 4;
  constexpr int kKRow = 33;                                  // padded: conflict-free ds_write_b128 of 8 lanes per key
  __shared__ uint4 sK[2][3 * 4 * 2][kKRow];                  // [buf][(piece*4 + chunk)*2 + half][key]
  __shared__ uint4 sV[2][3 * 2 * 2][64];                     // [buf][(piece*2 + kstep)*2 + half][dim]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  int bx, bh;
  xcd_job(a.xcd_remap, bx, bh);
  const int b = bh / a.H, head = bh % a.H;
  const int q0 = (bx * W + wave) * 32;
  const bool active = q0 < a.Nq;  // wave-uniform
  const float* qb = a.q + b * a.q_sb + head * a.q_sh;
  const float* kb = a.k + b * a.k_sb + head * a.k_sh;
  const float* vb = a.v + b * a.v_sb + head * a.v_sh;

  const int blk_q0 = bx * W * 32;
  const int blk_q1 = min(blk_q0 + W * 32, a.Nq);
  const int blk_keys = (blk_q1 > a.split_q) ? a.Nk : a.limit_k;
  const int my_keys = active ? ((q0 >= a.split_q) ? a.Nk : a.limit_k) : 0;

  // B operands of S^T = K Q^T: lane (j,h), chunk c holds Q[q0+j][16c + 8h + 0..7], pre-scaled, as three pieces
  bf16x8 bq[3][4];
  {
    const float* qp = qb + (long long)(active ? q0 + j : 0) * a.q_sn + 8 * h;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float x[8];
      const f32x4 u = *reinterpret_cast<const f32x4*>(qp + 16 * c), w = *reinterpret_cast<const f32x4*>(qp + 16 * c + 4);
      x[0] = u.x, x[1] = u.y, x[2] = u.z, x[3] = u.w, x[4] = w.x, x[5] = w.y, x[6] = w.z, x[7] = w.w;
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = active ? x[i] * a.scale_log2e : 0.0f;
      const Pieces3 p = split8(x);
      bq[0][c] = __builtin_bit_cast(bf16x8, p.h);
      bq[1][c] = __builtin_bit_cast(bf16x8, p.m);
      bq[2][c] = __builtin_bit_cast(bf16x8, p.l);
    }
  }

  f32x16 o0 = {0}, o1 = {0};  // O^T accumulators: dims 0..31 and 32..63, query on the lane
  float m_run = -INFINITY, l_run = 0.0f;

  // staging: thread e owns one K entry (key e >> 3, dims 8 (e & 7) ..) and one V entry (dim e & 63, k-step / half e >> 6)
  const int k_row = tid >> 3, k_oct = tid & 7;
  const int k_slot = (k_oct >> 1) * 2 + (k_oct & 1);          // chunk * 2 + half
  const int v_dim = tid & 63, v_sh = tid >> 6;                  // v_sh = kstep * 2 + half
  float kx[8], vx[8];
  auto fetch_tile = [&](int key0) {
    const float* kp = kb + (long long)(key0 + k_row) * a.k_sn + 8 * k_oct;
    const f32x4 u = *reinterpret_cast<const f32x4*>(kp), w = *reinterpret_cast<const f32x4*>(kp + 4);
    kx[0] = u.x, kx[1] = u.y, kx[2] = u.z, kx[3] = u.w, kx[4] = w.x, kx[5] = w.y, kx[6] = w.z, kx[7] = w.w;
    const int s = v_sh >> 1, hh = v_sh & 1;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int key = 16 * s + 8 * (jj >> 2) + 4 * hh + (jj & 3);
      vx[jj] = vb[(long long)(key0 + key) * a.v_sn + v_dim];
    }
  };
  auto store_tile = [&](int buf) {
    const Pieces3 pk = (ABL & 1) ? cheap8(kx) : split8(kx), pv = (ABL & 1) ? cheap8(vx) : split8(vx);
    sK[buf][0 * 8 + k_slot][k_row] = pk.h;
    sK[buf][1 * 8 + k_slot][k_row] = pk.m;
    sK[buf][2 * 8 + k_slot][k_row] = pk.l;
    sV[buf][0 * 4 + v_sh][v_dim] = pv.h;
    sV[buf][1 * 4 + v_sh][v_dim] = pv.m;
    sV[buf][2 * 4 + v_sh][v_dim] = pv.l;
  };

  // Staging order (round 4): a tile's global loads are issued a whole iteration before they are needed, and its split +
  // LDS writes come right AFTER the barrier that frees the other buffer, not in front of the next one - the writes'
  // latency then passes under this iteration's fragment reads and MFMAs instead of under a wait for lgkmcnt(0) that every
  // wave of the workgroup sits through at the barrier (the guide's issue-early / write-late split).
  const int ntiles = blk_keys / kTK;
  if (ntiles > 0) {
    fetch_tile(0);
    store_tile(0);
    if (LATE && !(ABL & 32) && ntiles > 1) fetch_tile(kTK);
  }
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = (ABL & 32) ? 0 : (t & 1);
    if (LATE && !(ABL & 32) && t + 1 < ntiles) {
      store_tile(buf ^ 1);                                   // tile t + 1: its buffer was last read before the barrier just passed
      if (t + 2 < ntiles) fetch_tile((t + 2) * kTK);         // in flight during the MFMAs below and the next store's barrier
    }
    if (!LATE && !(ABL & 32) && t + 1 < ntiles) fetch_tile((t + 1) * kTK);
    if (t * kTK < my_keys) {
      // ---- S^T = K Q^T: 4 chunks of 16 dims x 6 piece products --------------------------------
      f32x16 s = {0};
      if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        bf16x8 ak[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) ak[p] = __builtin_bit_cast(bf16x8, sK[buf][p * 8 + c * 2 + h][j]);
#define PTOPS_QK(PA, PB) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ak[PA], bq[PB][c], s, 0, 0, 0);
        if (!(ABL & 8)) { PTOPS_SPLIT3_TERMS(PTOPS_QK) } else { s[c] += __builtin_bit_cast(float, __builtin_bit_cast(uint4, ak[0]).x & 0x3f800000u); }
#undef PTOPS_QK
      }
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      // ---- online softmax (query = lane&31; this half holds 16 of the tile's 32 keys) ------
      if (!(ABL & 4)) {
      float mt = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
      mt = fmaxf(mt, other_half(mt));
      // Deferred rescaling: the running reference m_run only has to be an upper bound "within reach" of every score, not
      // the exact maximum - exp2(s - m_run) of a tile whose maximum exceeds m_run by <= 6 is at most 64, far from any
      // fp32 trouble, and O and l are accumulated against the same reference, so the quotient at the end is the same
      // function.  The wave rescales O (32 multiplies, an exp2) only when some query's maximum outgrows the reference by
      // more than that (wave-uniform branch); on trained or random scores that is the first tile and a handful after it.
      if (__builtin_amdgcn_ballot_w64(mt > m_run + 6.0f)) {
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
        m_run = m_new;
        o0 *= alpha;
        o1 *= alpha;
      }
      float psum = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r] - m_run);
        psum += s[r];
      }
      l_run += psum;  // per-half partial sum; halves are added at the end
      } else {
        l_run += s[0];
        m_run = 0.0f;
      }
      // ---- O^T += V^T P: 2 k-steps of 16 keys x 2 dim blocks x 6 piece products -----------------
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float px[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) px[i] = s[8 * ks + i];
        const Pieces3 pp = (ABL & 2) ? cheap8(px) : split8(px);
        const bf16x8 bp[3] = {__builtin_bit_cast(bf16x8, pp.h), __builtin_bit_cast(bf16x8, pp.m), __builtin_bit_cast(bf16x8, pp.l)};
        bf16x8 av0[3], av1[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          av0[p] = __builtin_bit_cast(bf16x8, sV[buf][p * 4 + ks * 2 + h][j]);
          av1[p] = __builtin_bit_cast(bf16x8, sV[buf][p * 4 + ks * 2 + h][32 + j]);
        }
#define PTOPS_PV(PA, PB)                                                          \
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av0[PA], bp[PB], o0, 0, 0, 0); \
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av1[PA], bp[PB], o1, 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        if (!(ABL & 16)) { PTOPS_SPLIT3_TERMS(PTOPS_PV) } else {
          o0[ks] += __builtin_bit_cast(float, (__builtin_bit_cast(uint4, av0[0]).x ^ __builtin_bit_cast(uint4, bp[1]).y) & 0x3f800000u);
          o1[ks] += __builtin_bit_cast(float, (__builtin_bit_cast(uint4, av1[2]).x ^ __builtin_bit_cast(uint4, bp[2]).y) & 0x3f800000u);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
#undef PTOPS_PV
      }
    }
    if (!LATE && !(ABL & 32) && t + 1 < ntiles) store_tile(buf ^ 1);
    __syncthreads();
  }

  if (active) {
    const float l_tot = l_run + other_half(l_run);
    const float inv = 1.0f / l_tot;
    // accumulator layout: column = query (lane&31), row(reg) = (reg&3) + 8*(reg>>2) + 4*h = dim
    float* op = a.o + (((long long)b * a.Nq + q0 + j) * a.H + head) * kD;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 w0 = {o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
      f32x4 w1 = {o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
      *reinterpret_cast<f32x4*>(op + 8 * g + 4 * h) = w0;
      *reinterpret_cast<f32x4*>(op + 32 + 8 * g + 4 * h) = w1;
    }
    if (h == 0) a.lse[((long long)b * a.H + head) * a.Nq + q0 + j] = m_run + __builtin_amdgcn_logf(l_tot);
  }
}


// ------------------------------------------------------------------------------------------------
// Backward.  With P = softmax(scale q k^T), D_i = sum_c dO_ic O_ic and dS = P o (dP - D), dP = dO v^T:
//   dq = scale dS k,   dk = scale dS^T q,   dv = P^T dO.
// Two kernels, both deterministic (no atomics) and both built from the forward's two MFMA patterns
// "row-major LDS tile x register-resident operand" and "accumulator reused as the B operand":
//   dq  : query tile stationary (Q^T, dO^T in registers), K/V tiles stream through LDS.
//         S^T = K Q^T, dP^T = V dO^T (queries on lanes, so lse / D are per-lane scalars),
//         dQ^T += K^T dS^T with B = the dS^T accumulator.  Also emits D for the second kernel.
//   dkv : key tile stationary (K^T, V^T in registers), Q/dO tiles stream through LDS.
//         S = Q K^T, dP = dO V^T (keys on lanes), dV^T += dO^T P and dK^T += Q^T dS with B = P / dS.
// S and dP are recomputed in both (7 GEMMs instead of 5): the price of not reducing dq across key tiles.
struct AttnBwdArgs {
  const float* q;
  const float* k;
  const float* v;
  const float* o;     // [B, Nq, H, 64] contiguous (forward output)
  const float* dout;  // [B, Nq, H, 64] contiguous
  const float* lse;   // [B, H, Nq] base-2
  float* delta;       // [B, H, Nq]  D_i, written by the dq kernel, read by the dkv kernel
  float* dq;
  float* dk;
  float* dv;
  long long q_sb, q_sh, q_sn, k_sb, k_sh, k_sn, v_sb, v_sh, v_sn;
  long long dq_sb, dq_sh, dq_sn, dk_sb, dk_sh, dk_sn, dv_sb, dv_sh, dv_sn;
  int H, Nq, Nk;
  int split_q, limit_k;
  float scale, scale_log2e;
  int xcd_remap;
};

// accumulator (rows = dims, column = lane&31) -> 64 contiguous floats of row `dst`, times `mul`
__device__ __forceinline__ void store_dim_major(float* dst, const f32x16& a0, const f32x16& a1, int h, float mul) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 w0 = {a0[4 * g] * mul, a0[4 * g + 1] * mul, a0[4 * g + 2] * mul, a0[4 * g + 3] * mul};
    f32x4 w1 = {a1[4 * g] * mul, a1[4 * g + 1] * mul, a1[4 * g + 2] * mul, a1[4 * g + 3] * mul};
    *reinterpret_cast<f32x4*>(dst + 8 * g + 4 * h) = w0;
    *reinterpret_cast<f32x4*>(dst + 32 + 8 * g + 4 * h) = w1;
  }
}

template <int W>
__global__ __launch_bounds__(W * 64, 2) void attn_bwd_dq_kernel(AttnBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float sK[2][kTK * kLd];
  __shared__ __attribute__((aligned(16))) float sV[2][kTK * kLd];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.H, head = bh % a.H;
  const int q0 = (blockIdx.x * W + wave) * 32;
  const bool active = q0 < a.Nq;
  const float* kb = a.k + b * a.k_sb + head * a.k_sh;
  const float* vb = a.v + b * a.v_sb + head * a.v_sh;

  const int blk_q0 = blockIdx.x * W * 32;
  const int blk_q1 = min(blk_q0 + W * 32, a.Nq);
  const int blk_keys = (blk_q1 > a.split_q) ? a.Nk : a.limit_k;
  const int my_keys = active ? ((q0 >= a.split_q) ? a.Nk : a.limit_k) : 0;

  float qf[32], dof[32];
  float lse = 0.0f, dlt = 0.0f;
  if (active) {
    const float* qp = a.q + b * a.q_sb + head * a.q_sh + (long long)(q0 + j) * a.q_sn + 32 * h;
    const long long orow = (((long long)b * a.Nq + q0 + j) * a.H + head) * kD + 32 * h;
    const float* dop = a.dout + orow;
    const float* op = a.o + orow;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const f32x4 q4 = *reinterpret_cast<const f32x4*>(qp + 4 * t);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(dop + 4 * t);
      const f32x4 o4 = *reinterpret_cast<const f32x4*>(op + 4 * t);
      qf[4 * t + 0] = q4.x * a.scale_log2e;
      qf[4 * t + 1] = q4.y * a.scale_log2e;
      qf[4 * t + 2] = q4.z * a.scale_log2e;
      qf[4 * t + 3] = q4.w * a.scale_log2e;
      dof[4 * t + 0] = d4.x;
      dof[4 * t + 1] = d4.y;
      dof[4 * t + 2] = d4.z;
      dof[4 * t + 3] = d4.w;
      dlt = fmaf(d4.x, o4.x, dlt);
      dlt = fmaf(d4.y, o4.y, dlt);
      dlt = fmaf(d4.z, o4.z, dlt);
      dlt = fmaf(d4.w, o4.w, dlt);
    }
    dlt += other_half(dlt);
    const long long li = ((long long)b * a.H + head) * a.Nq + q0 + j;
    lse = a.lse[li];
    if (h == 0) a.delta[li] = dlt;
  } else {
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      qf[t] = 0.0f;
      dof[t] = 0.0f;
    }
  }

  f32x16 g0 = {0}, g1 = {0};  // dQ^T accumulators: dims 0..31 / 32..63, query on the lane

  constexpr int kPerThread = (kTK * (kD / 4) + W * 64 - 1) / (W * 64);
  f32x4 kreg[kPerThread], vreg[kPerThread];
  auto fetch_tile = [&](int key0) {
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
      const int e = tid + u * W * 64;
      if (e < kTK * (kD / 4)) {
        const int r = e >> 4, c4 = (e & 15) * 4;
        kreg[u] = *reinterpret_cast<const f32x4*>(kb + (long long)(key0 + r) * a.k_sn + c4);
        vreg[u] = *reinterpret_cast<const f32x4*>(vb + (long long)(key0 + r) * a.v_sn + c4);
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
      const int e = tid + u * W * 64;
      if (e < kTK * (kD / 4)) {
        const int r = e >> 4, c4 = (e & 15) * 4;
        *reinterpret_cast<f32x4*>(&sK[buf][r * kLd + c4]) = kreg[u];
        *reinterpret_cast<f32x4*>(&sV[buf][r * kLd + c4]) = vreg[u];
      }
    }
  };

  const int ntiles = blk_keys / kTK;
  if (ntiles > 0) {
    fetch_tile(0);
    store_tile(0);
  }
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) fetch_tile((t + 1) * kTK);
    if (t * kTK < my_keys) {
      f32x16 s = {0}, dp = {0};
      const float* kr = &sK[buf][j * kLd + 32 * h];
      const float* vr = &sV[buf][j * kLd + 32 * h];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const f32x4 k4 = *reinterpret_cast<const f32x4*>(kr + 4 * g);
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(vr + 4 * g);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.x, qf[4 * g + 0], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(v4.x, dof[4 * g + 0], dp, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.y, qf[4 * g + 1], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(v4.y, dof[4 * g + 1], dp, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.z, qf[4 * g + 2], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(v4.z, dof[4 * g + 2], dp, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.w, qf[4 * g + 3], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(v4.w, dof[4 * g + 3], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r] - lse) * (dp[r] - dlt);  // dS^T (unscaled)
      const float* kc = &sK[buf][(4 * h) * kLd + j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2);
        const float k0v = kc[key * kLd], k1v = kc[key * kLd + 32];
        g0 = __builtin_amdgcn_mfma_f32_32x32x2f32(k0v, s[r], g0, 0, 0, 0);
        g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(k1v, s[r], g1, 0, 0, 0);
      }
    }
    if (t + 1 < ntiles) store_tile(buf ^ 1);
    __syncthreads();
  }

  if (active)
    store_dim_major(a.dq + b * a.dq_sb + head * a.dq_sh + (long long)(q0 + j) * a.dq_sn, g0, g1, h, a.scale);
}

template <int W>
__global__ __launch_bounds__(W * 64, 2) void attn_bwd_dkv_kernel(AttnBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float sQ[2][kTK * kLd];
  __shared__ __attribute__((aligned(16))) float sO[2][kTK * kLd];   // dO tile
  __shared__ __attribute__((aligned(16))) float sL[2][2 * kTK];     // lse[32], delta[32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.H, head = bh % a.H;
  const int k0 = (blockIdx.x * W + wave) * 32;
  const bool active = k0 < a.Nk;
  const float* qb = a.q + b * a.q_sb + head * a.q_sh;
  const float* dob = a.dout + ((long long)b * a.Nq * a.H + head) * kD;  // row stride H*64
  const float* lb = a.lse + ((long long)b * a.H + head) * a.Nq;
  const float* db = a.delta + ((long long)b * a.H + head) * a.Nq;
  const long long do_sn = (long long)a.H * kD;

  // query tiles this workgroup walks: keys >= limit_k are seen only by queries >= split_q
  const int blk_k0 = blockIdx.x * W * 32;
  const int blk_qstart = (blk_k0 >= a.limit_k) ? a.split_q : 0;
  const int my_qstart = active ? ((k0 >= a.limit_k) ? a.split_q : 0) : a.Nq;

  float kf[32], vf[32];
  if (active) {
    const float* kp = a.k + b * a.k_sb + head * a.k_sh + (long long)(k0 + j) * a.k_sn + 32 * h;
    const float* vp = a.v + b * a.v_sb + head * a.v_sh + (long long)(k0 + j) * a.v_sn + 32 * h;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const f32x4 k4 = *reinterpret_cast<const f32x4*>(kp + 4 * t);
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(vp + 4 * t);
      kf[4 * t + 0] = k4.x * a.scale_log2e;
      kf[4 * t + 1] = k4.y * a.scale_log2e;
      kf[4 * t + 2] = k4.z * a.scale_log2e;
      kf[4 * t + 3] = k4.w * a.scale_log2e;
      vf[4 * t + 0] = v4.x;
      vf[4 * t + 1] = v4.y;
      vf[4 * t + 2] = v4.z;
      vf[4 * t + 3] = v4.w;
    }
  } else {
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      kf[t] = 0.0f;
      vf[t] = 0.0f;
    }
  }

  f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};  // dK^T / dV^T: rows = dims, key on the lane

  constexpr int kPerThread = (kTK * (kD / 4) + W * 64 - 1) / (W * 64);
  f32x4 qreg[kPerThread], oreg[kPerThread];
  float lreg = 0.0f;
  auto fetch_tile = [&](int q0) {
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
      const int e = tid + u * W * 64;
      if (e < kTK * (kD / 4)) {
        const int r = e >> 4, c4 = (e & 15) * 4;
        qreg[u] = *reinterpret_cast<const f32x4*>(qb + (long long)(q0 + r) * a.q_sn + c4);
        oreg[u] = *reinterpret_cast<const f32x4*>(dob + (long long)(q0 + r) * do_sn + c4);
      }
    }
    if (tid < 64) lreg = (tid < 32) ? lb[q0 + tid] : db[q0 + tid - 32];
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
      const int e = tid + u * W * 64;
      if (e < kTK * (kD / 4)) {
        const int r = e >> 4, c4 = (e & 15) * 4;
        *reinterpret_cast<f32x4*>(&sQ[buf][r * kLd + c4]) = qreg[u];
        *reinterpret_cast<f32x4*>(&sO[buf][r * kLd + c4]) = oreg[u];
      }
    }
    if (tid < 64) sL[buf][tid] = lreg;
  };

  const int t_begin = blk_qstart / kTK, t_end = a.Nq / kTK;
  if (t_begin < t_end) {
    fetch_tile(t_begin * kTK);
    store_tile(t_begin & 1);
  }
  __syncthreads();
  for (int t = t_begin; t < t_end; ++t) {
    const int buf = t & 1;
    if (t + 1 < t_end) fetch_tile((t + 1) * kTK);
    if (t * kTK >= my_qstart) {
      // ---- S = Q K^T, dP = dO V^T  (rows = queries (r&3)+8(r>>2)+4h, column = key j) --------------
      f32x16 s = {0}, dp = {0};
      const float* qr = &sQ[buf][j * kLd + 32 * h];
      const float* orow = &sO[buf][j * kLd + 32 * h];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const f32x4 q4 = *reinterpret_cast<const f32x4*>(qr + 4 * g);
        const f32x4 o4 = *reinterpret_cast<const f32x4*>(orow + 4 * g);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.x, kf[4 * g + 0], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(o4.x, vf[4 * g + 0], dp, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.y, kf[4 * g + 1], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(o4.y, vf[4 * g + 1], dp, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.z, kf[4 * g + 2], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(o4.z, vf[4 * g + 2], dp, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.w, kf[4 * g + 3], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(o4.w, vf[4 * g + 3], dp, 0, 0, 0);
      }
      // ---- P and dS; lse / D of the 16 query rows this half holds: 4 broadcast b128 reads each --------
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(&sL[buf][8 * g + 4 * h]);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(&sL[buf][kTK + 8 * g + 4 * h]);
        const float p0 = __builtin_amdgcn_exp2f(s[4 * g + 0] - l4.x);
        const float p1 = __builtin_amdgcn_exp2f(s[4 * g + 1] - l4.y);
        const float p2 = __builtin_amdgcn_exp2f(s[4 * g + 2] - l4.z);
        const float p3 = __builtin_amdgcn_exp2f(s[4 * g + 3] - l4.w);
        s[4 * g + 0] = p0;
        s[4 * g + 1] = p1;
        s[4 * g + 2] = p2;
        s[4 * g + 3] = p3;
        dp[4 * g + 0] = p0 * (dp[4 * g + 0] - d4.x);
        dp[4 * g + 1] = p1 * (dp[4 * g + 1] - d4.y);
        dp[4 * g + 2] = p2 * (dp[4 * g + 2] - d4.z);
        dp[4 * g + 3] = p3 * (dp[4 * g + 3] - d4.w);
      }
      // ---- dV^T += dO^T P, dK^T += Q^T dS : k-step r uses query rows (r&3)+8(r>>2) [+4 in half 1] ----
      const float* oc = &sO[buf][(4 * h) * kLd + j];
      const float* qc = &sQ[buf][(4 * h) * kLd + j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        const float o0v = oc[row * kLd], o1v = oc[row * kLd + 32];
        const float q0v = qc[row * kLd], q1v = qc[row * kLd + 32];
        dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(o0v, s[r], dv0, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(q0v, dp[r], dk0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(o1v, s[r], dv1, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(q1v, dp[r], dk1, 0, 0, 0);
      }
    }
    if (t + 1 < t_end) store_tile(buf ^ 1);
    __syncthreads();
  }

  if (active) {
    store_dim_major(a.dk + b * a.dk_sb + head * a.dk_sh + (long long)(k0 + j) * a.dk_sn, dk0, dk1, h, a.scale);
    store_dim_major(a.dv + b * a.dv_sb + head * a.dv_sh + (long long)(k0 + j) * a.dv_sn, dv0, dv1, h, 1.0f);
  }
}


// ------------------------------------------------------------------------------------------------
// The two-kernel backward on the bf16 matrix pipe with fp32 accuracy (round 2): the arithmetic of attn_fwd_split3_kernel
// (three bf16 pieces per fp32 operand, six piece products per contraction on v_mfma_f32_32x32x16_bf16) applied to the
// seven GEMMs of attn_bwd_dq_kernel / attn_bwd_dkv_kernel.  Operands meet the matrix pipe in two LDS layouts, both
// split by the staging threads on the way in:
//   row entries         [piece][16-dim chunk][k-half][row], rows padded to 33: A operand of "tile rows x head dims"
//                       contractions over the head dimension (S, dP);
//   transposed entries  [piece][k-step][k-half][dim], the 8 values of an entry being the tile rows
//                       16 s + 8 (jj >> 2) + 4 h + (jj & 3): A operand of contractions over the tile rows (dQ^T, dK^T,
//                       dV^T), in exactly the order in which an S / dP accumulator hands its rows over as the B operand.
// The stationary operand (Q, dO for dq; K, V for dkv) is split once into 2 x 48 VGPRs.
template <bool LATE>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_split3_kernel(AttnBwdArgs a) {
  constexpr int W = 4;
  constexpr int kRow = 33;
  __shared__ uint4 sR[2][2][3 * 4 * 2][kRow];                // [buf][K | V][(piece*4 + chunk)*2 + half][key]
  __shared__ uint4 sT[2][3 * 2 * 2][64];                     // [buf][(piece*2 + kstep)*2 + half][dim]   (K^T)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  int bx, bh;
  xcd_job(a.xcd_remap, bx, bh);
  const int b = bh / a.H, head = bh % a.H;
  const int q0 = (bx * W + wave) * 32;
  const bool active = q0 < a.Nq;
  const float* kb = a.k + b * a.k_sb + head * a.k_sh;
  const float* vb = a.v + b * a.v_sb + head * a.v_sh;

  const int blk_q0 = bx * W * 32;
  const int blk_q1 = min(blk_q0 + W * 32, a.Nq);
  const int blk_keys = (blk_q1 > a.split_q) ? a.Nk : a.limit_k;
  const int my_keys = active ? ((q0 >= a.split_q) ? a.Nk : a.limit_k) : 0;

  // stationary B operands: lane (j,h), chunk c holds X[q0+j][16c + 8h + 0..7] as three pieces (Q pre-scaled)
  bf16x8 bq[3][4], bdo[3][4];
  float lse = 0.0f, dlt = 0.0f;
  {
    const int qrow = active ? q0 + j : 0;
    const float* qp = a.q + b * a.q_sb + head * a.q_sh + (long long)qrow * a.q_sn + 8 * h;
    const long long orow = (((long long)b * a.Nq + qrow) * a.H + head) * kD + 8 * h;
    const float* dop = a.dout + orow;
    const float* op = a.o + orow;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float x[8], y[8];
      const f32x4 u = *reinterpret_cast<const f32x4*>(qp + 16 * c), w = *reinterpret_cast<const f32x4*>(qp + 16 * c + 4);
      const f32x4 d0 = *reinterpret_cast<const f32x4*>(dop + 16 * c), d1 = *reinterpret_cast<const f32x4*>(dop + 16 * c + 4);
      const f32x4 o0 = *reinterpret_cast<const f32x4*>(op + 16 * c), o1 = *reinterpret_cast<const f32x4*>(op + 16 * c + 4);
      x[0] = u.x, x[1] = u.y, x[2] = u.z, x[3] = u.w, x[4] = w.x, x[5] = w.y, x[6] = w.z, x[7] = w.w;
      y[0] = d0.x, y[1] = d0.y, y[2] = d0.z, y[3] = d0.w, y[4] = d1.x, y[5] = d1.y, y[6] = d1.z, y[7] = d1.w;
      dlt = fmaf(d0.x, o0.x, dlt);
      dlt = fmaf(d0.y, o0.y, dlt);
      dlt = fmaf(d0.z, o0.z, dlt);
      dlt = fmaf(d0.w, o0.w, dlt);
      dlt = fmaf(d1.x, o1.x, dlt);
      dlt = fmaf(d1.y, o1.y, dlt);
      dlt = fmaf(d1.z, o1.z, dlt);
      dlt = fmaf(d1.w, o1.w, dlt);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        x[i] = active ? x[i] * a.scale_log2e : 0.0f;
        y[i] = active ? y[i] : 0.0f;
      }
      const Pieces3 p = split8(x), d = split8(y);
      bq[0][c] = __builtin_bit_cast(bf16x8, p.h);
      bq[1][c] = __builtin_bit_cast(bf16x8, p.m);
      bq[2][c] = __builtin_bit_cast(bf16x8, p.l);
      bdo[0][c] = __builtin_bit_cast(bf16x8, d.h);
      bdo[1][c] = __builtin_bit_cast(bf16x8, d.m);
      bdo[2][c] = __builtin_bit_cast(bf16x8, d.l);
    }
    dlt += other_half(dlt);
    if (active) {
      const long long li = ((long long)b * a.H + head) * a.Nq + q0 + j;
      lse = a.lse[li];
      if (h == 0) a.delta[li] = dlt;
    }
  }

  f32x16 g0 = {0}, g1 = {0};  // dQ^T accumulators: dims 0..31 / 32..63, query on the lane

  // staging: thread e owns the K and the V row entry (key e >> 3, dims 8 (e & 7) ..) and one K^T entry
  // (dim e & 63, k-step / half e >> 6)
  const int r_row = tid >> 3, r_oct = tid & 7;
  const int r_slot = (r_oct >> 1) * 2 + (r_oct & 1);          // chunk * 2 + half
  const int t_dim = tid & 63, t_sh = tid >> 6;                  // t_sh = kstep * 2 + half
  float kx[8], vx[8], tx[8];
  auto fetch_tile = [&](int key0) {
    const float* kp = kb + (long long)(key0 + r_row) * a.k_sn + 8 * r_oct;
    const float* vp = vb + (long long)(key0 + r_row) * a.v_sn + 8 * r_oct;
    const f32x4 u = *reinterpret_cast<const f32x4*>(kp), w = *reinterpret_cast<const f32x4*>(kp + 4);
    const f32x4 u2 = *reinterpret_cast<const f32x4*>(vp), w2 = *reinterpret_cast<const f32x4*>(vp + 4);
    kx[0] = u.x, kx[1] = u.y, kx[2] = u.z, kx[3] = u.w, kx[4] = w.x, kx[5] = w.y, kx[6] = w.z, kx[7] = w.w;
    vx[0] = u2.x, vx[1] = u2.y, vx[2] = u2.z, vx[3] = u2.w, vx[4] = w2.x, vx[5] = w2.y, vx[6] = w2.z, vx[7] = w2.w;
    const int s = t_sh >> 1, hh = t_sh & 1;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int key = 16 * s + 8 * (jj >> 2) + 4 * hh + (jj & 3);
      tx[jj] = kb[(long long)(key0 + key) * a.k_sn + t_dim];
    }
  };
  auto store_tile = [&](int buf) {
    const Pieces3 pk = split8(kx), pv = split8(vx), pt = split8(tx);
    sR[buf][0][0 * 8 + r_slot][r_row] = pk.h;
    sR[buf][0][1 * 8 + r_slot][r_row] = pk.m;
    sR[buf][0][2 * 8 + r_slot][r_row] = pk.l;
    sR[buf][1][0 * 8 + r_slot][r_row] = pv.h;
    sR[buf][1][1 * 8 + r_slot][r_row] = pv.m;
    sR[buf][1][2 * 8 + r_slot][r_row] = pv.l;
    sT[buf][0 * 4 + t_sh][t_dim] = pt.h;
    sT[buf][1 * 4 + t_sh][t_dim] = pt.m;
    sT[buf][2 * 4 + t_sh][t_dim] = pt.l;
  };

  // staging order as in attn_fwd_split3_kernel: loads a whole iteration ahead, split + LDS writes right after the barrier
  const int ntiles = blk_keys / kTK;
  if (ntiles > 0) {
    fetch_tile(0);
    store_tile(0);
    if (LATE && ntiles > 1) fetch_tile(kTK);
  }
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (LATE && t + 1 < ntiles) {
      store_tile(buf ^ 1);
      if (t + 2 < ntiles) fetch_tile((t + 2) * kTK);
    }
    if (!LATE && t + 1 < ntiles) fetch_tile((t + 1) * kTK);
    if (t * kTK < my_keys) {
      // ---- S^T = K Q^T, dP^T = V dO^T: 4 chunks of 16 dims x 6 piece products each ---------------------
      f32x16 s = {0}, dp = {0};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        bf16x8 ak[3], av[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          ak[p] = __builtin_bit_cast(bf16x8, sR[buf][0][p * 8 + c * 2 + h][j]);
          av[p] = __builtin_bit_cast(bf16x8, sR[buf][1][p * 8 + c * 2 + h][j]);
        }
#define PTOPS_SDP(PA, PB)                                                          \
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ak[PA], bq[PB][c], s, 0, 0, 0);  \
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA], bdo[PB][c], dp, 0, 0, 0);
        PTOPS_SPLIT3_TERMS(PTOPS_SDP)
#undef PTOPS_SDP
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r] - lse) * (dp[r] - dlt);  // dS^T (unscaled)
      // ---- dQ^T += K^T dS^T: 2 k-steps of 16 keys x 2 dim blocks x 6 piece products ---------------------
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float px[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) px[i] = s[8 * ks + i];
        const Pieces3 pp = split8(px);
        const bf16x8 bp[3] = {__builtin_bit_cast(bf16x8, pp.h), __builtin_bit_cast(bf16x8, pp.m), __builtin_bit_cast(bf16x8, pp.l)};
        bf16x8 at0[3], at1[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          at0[p] = __builtin_bit_cast(bf16x8, sT[buf][p * 4 + ks * 2 + h][j]);
          at1[p] = __builtin_bit_cast(bf16x8, sT[buf][p * 4 + ks * 2 + h][32 + j]);
        }
#define PTOPS_DQ(PA, PB)                                                           \
        g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at0[PA], bp[PB], g0, 0, 0, 0);  \
        g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at1[PA], bp[PB], g1, 0, 0, 0);
        PTOPS_SPLIT3_TERMS(PTOPS_DQ)
#undef PTOPS_DQ
      }
    }
    if (!LATE && t + 1 < ntiles) store_tile(buf ^ 1);
    __syncthreads();
  }

  if (active)
    store_dim_major(a.dq + b * a.dq_sb + head * a.dq_sh + (long long)(q0 + j) * a.dq_sn, g0, g1, h, a.scale);
}

template <int W, bool LATE>
__global__ __launch_bounds__(W * 64) void attn_bwd_dkv_split3_kernel(AttnBwdArgs a) {
  constexpr int kRow = 33;
  __shared__ uint4 sR[2][2][3 * 4 * 2][kRow];                // [buf][Q | dO][(piece*4 + chunk)*2 + half][query]
  __shared__ uint4 sT[2][2][3 * 2 * 2][64];                  // [buf][Q^T | dO^T][(piece*2 + kstep)*2 + half][dim]
  __shared__ __attribute__((aligned(16))) float sL[2][2 * kTK];  // lse[32], delta[32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  int bx, bh;
  xcd_job(a.xcd_remap, bx, bh);
  const int b = bh / a.H, head = bh % a.H;
  const int k0 = (bx * W + wave) * 32;
  const bool active = k0 < a.Nk;
  const float* qb = a.q + b * a.q_sb + head * a.q_sh;
  const float* dob = a.dout + ((long long)b * a.Nq * a.H + head) * kD;  // row stride H*64
  const float* lb = a.lse + ((long long)b * a.H + head) * a.Nq;
  const float* db = a.delta + ((long long)b * a.H + head) * a.Nq;
  const long long do_sn = (long long)a.H * kD;

  // query tiles this workgroup walks: keys >= limit_k are seen only by queries >= split_q
  const int blk_k0 = bx * W * 32;
  const int blk_qstart = (blk_k0 >= a.limit_k) ? a.split_q : 0;
  const int my_qstart = active ? ((k0 >= a.limit_k) ? a.split_q : 0) : a.Nq;

  // stationary B operands: lane (j,h), chunk c holds X[k0+j][16c + 8h + 0..7] as three pieces (K pre-scaled)
  bf16x8 bk[3][4], bv[3][4];
  {
    const int krow = active ? k0 + j : 0;
    const float* kp = a.k + b * a.k_sb + head * a.k_sh + (long long)krow * a.k_sn + 8 * h;
    const float* vp = a.v + b * a.v_sb + head * a.v_sh + (long long)krow * a.v_sn + 8 * h;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float x[8], y[8];
      const f32x4 u = *reinterpret_cast<const f32x4*>(kp + 16 * c), w = *reinterpret_cast<const f32x4*>(kp + 16 * c + 4);
      const f32x4 u2 = *reinterpret_cast<const f32x4*>(vp + 16 * c), w2 = *reinterpret_cast<const f32x4*>(vp + 16 * c + 4);
      x[0] = u.x, x[1] = u.y, x[2] = u.z, x[3] = u.w, x[4] = w.x, x[5] = w.y, x[6] = w.z, x[7] = w.w;
      y[0] = u2.x, y[1] = u2.y, y[2] = u2.z, y[3] = u2.w, y[4] = w2.x, y[5] = w2.y, y[6] = w2.z, y[7] = w2.w;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        x[i] = active ? x[i] * a.scale_log2e : 0.0f;
        y[i] = active ? y[i] : 0.0f;
      }
      const Pieces3 p = split8(x), d = split8(y);
      bk[0][c] = __builtin_bit_cast(bf16x8, p.h);
      bk[1][c] = __builtin_bit_cast(bf16x8, p.m);
      bk[2][c] = __builtin_bit_cast(bf16x8, p.l);
      bv[0][c] = __builtin_bit_cast(bf16x8, d.h);
      bv[1][c] = __builtin_bit_cast(bf16x8, d.m);
      bv[2][c] = __builtin_bit_cast(bf16x8, d.l);
    }
  }

  f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};  // dK^T / dV^T: rows = dims, key on the lane

  // staging: the tile's 1024 operand entries (256 each of Q rows, dO rows, Q^T, dO^T) over W*64 threads
  constexpr int kPer = 1024 / (W * 64);
  float xr[kPer][8];
  float lreg = 0.0f;
  auto fetch_tile = [&](int q0) {
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int e = tid + u * W * 64;
      const int which = (e >> 8) & 1, idx = e & 255;          // e >> 9: 0 = row entries, 1 = transposed entries
      const float* src = which ? dob : qb;
      const long long sn = which ? do_sn : a.q_sn;
      if (u < kPer / 2) {
        const float* p = src + (long long)(q0 + (idx >> 3)) * sn + 8 * (idx & 7);
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(p), c1 = *reinterpret_cast<const f32x4*>(p + 4);
        xr[u][0] = c0.x, xr[u][1] = c0.y, xr[u][2] = c0.z, xr[u][3] = c0.w;
        xr[u][4] = c1.x, xr[u][5] = c1.y, xr[u][6] = c1.z, xr[u][7] = c1.w;
      } else {
        const int dim = idx & 63, sh = idx >> 6, s = sh >> 1, hh = sh & 1;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int row = 16 * s + 8 * (jj >> 2) + 4 * hh + (jj & 3);
          xr[u][jj] = src[(long long)(q0 + row) * sn + dim];
        }
      }
    }
    if (tid < 64) lreg = (tid < 32) ? lb[q0 + tid] : db[q0 + tid - 32];
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int e = tid + u * W * 64;
      const int which = (e >> 8) & 1, idx = e & 255;
      const Pieces3 p = split8(xr[u]);
      if (u < kPer / 2) {
        const int row = idx >> 3, oct = idx & 7, slot = (oct >> 1) * 2 + (oct & 1);
        sR[buf][which][0 * 8 + slot][row] = p.h;
        sR[buf][which][1 * 8 + slot][row] = p.m;
        sR[buf][which][2 * 8 + slot][row] = p.l;
      } else {
        const int dim = idx & 63, sh = idx >> 6;
        sT[buf][which][0 * 4 + sh][dim] = p.h;
        sT[buf][which][1 * 4 + sh][dim] = p.m;
        sT[buf][which][2 * 4 + sh][dim] = p.l;
      }
    }
    if (tid < 64) sL[buf][tid] = lreg;
  };

  // staging order as in attn_fwd_split3_kernel: loads a whole iteration ahead, split + LDS writes right after the barrier
  const int t_begin = blk_qstart / kTK, t_end = a.Nq / kTK;
  if (t_begin < t_end) {
    fetch_tile(t_begin * kTK);
    store_tile(t_begin & 1);
    if (LATE && t_begin + 1 < t_end) fetch_tile((t_begin + 1) * kTK);
  }
  __syncthreads();
  for (int t = t_begin; t < t_end; ++t) {
    const int buf = t & 1;
    if (LATE && t + 1 < t_end) {
      store_tile(buf ^ 1);
      if (t + 2 < t_end) fetch_tile((t + 2) * kTK);
    }
    if (!LATE && t + 1 < t_end) fetch_tile((t + 1) * kTK);
    if (t * kTK >= my_qstart) {
      // ---- S = Q K^T, dP = dO V^T  (rows = queries (r&3)+8(r>>2)+4h, column = key j) --------------
      f32x16 s = {0}, dp = {0};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        bf16x8 aq[3], ao[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          aq[p] = __builtin_bit_cast(bf16x8, sR[buf][0][p * 8 + c * 2 + h][j]);
          ao[p] = __builtin_bit_cast(bf16x8, sR[buf][1][p * 8 + c * 2 + h][j]);
        }
#define PTOPS_SDP(PA, PB)                                                          \
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[PA], bk[PB][c], s, 0, 0, 0);  \
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ao[PA], bv[PB][c], dp, 0, 0, 0);
        PTOPS_SPLIT3_TERMS(PTOPS_SDP)
#undef PTOPS_SDP
      }
      // ---- P and dS; lse / D of the 16 query rows this half holds: 4 broadcast b128 reads each --------
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(&sL[buf][8 * g + 4 * h]);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(&sL[buf][kTK + 8 * g + 4 * h]);
        const float p0 = __builtin_amdgcn_exp2f(s[4 * g + 0] - l4.x);
        const float p1 = __builtin_amdgcn_exp2f(s[4 * g + 1] - l4.y);
        const float p2 = __builtin_amdgcn_exp2f(s[4 * g + 2] - l4.z);
        const float p3 = __builtin_amdgcn_exp2f(s[4 * g + 3] - l4.w);
        s[4 * g + 0] = p0;
        s[4 * g + 1] = p1;
        s[4 * g + 2] = p2;
        s[4 * g + 3] = p3;
        dp[4 * g + 0] = p0 * (dp[4 * g + 0] - d4.x);
        dp[4 * g + 1] = p1 * (dp[4 * g + 1] - d4.y);
        dp[4 * g + 2] = p2 * (dp[4 * g + 2] - d4.z);
        dp[4 * g + 3] = p3 * (dp[4 * g + 3] - d4.w);
      }
      // ---- dV^T += dO^T P, dK^T += Q^T dS: 2 k-steps of 16 queries x 2 dim blocks x 6 piece products each ----
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float px[8], dx[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          px[i] = s[8 * ks + i];
          dx[i] = dp[8 * ks + i];
        }
        {
          const Pieces3 pp = split8(px);
          const bf16x8 bp[3] = {__builtin_bit_cast(bf16x8, pp.h), __builtin_bit_cast(bf16x8, pp.m), __builtin_bit_cast(bf16x8, pp.l)};
          bf16x8 at0[3], at1[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            at0[p] = __builtin_bit_cast(bf16x8, sT[buf][1][p * 4 + ks * 2 + h][j]);
            at1[p] = __builtin_bit_cast(bf16x8, sT[buf][1][p * 4 + ks * 2 + h][32 + j]);
          }
#define PTOPS_DV(PA, PB)                                                             \
          dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at0[PA], bp[PB], dv0, 0, 0, 0); \
          dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at1[PA], bp[PB], dv1, 0, 0, 0);
          PTOPS_SPLIT3_TERMS(PTOPS_DV)
#undef PTOPS_DV
        }
        {
          const Pieces3 pd = split8(dx);
          const bf16x8 bd[3] = {__builtin_bit_cast(bf16x8, pd.h), __builtin_bit_cast(bf16x8, pd.m), __builtin_bit_cast(bf16x8, pd.l)};
          bf16x8 at0[3], at1[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            at0[p] = __builtin_bit_cast(bf16x8, sT[buf][0][p * 4 + ks * 2 + h][j]);
            at1[p] = __builtin_bit_cast(bf16x8, sT[buf][0][p * 4 + ks * 2 + h][32 + j]);
          }
#define PTOPS_DK(PA, PB)                                                             \
          dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at0[PA], bd[PB], dk0, 0, 0, 0); \
          dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at1[PA], bd[PB], dk1, 0, 0, 0);
          PTOPS_SPLIT3_TERMS(PTOPS_DK)
#undef PTOPS_DK
        }
      }
    }
    if (!LATE && t + 1 < t_end) store_tile(buf ^ 1);
    __syncthreads();
  }

  if (active) {
    store_dim_major(a.dk + b * a.dk_sb + head * a.dk_sh + (long long)(k0 + j) * a.dk_sn, dk0, dk1, h, a.scale);
    store_dim_major(a.dv + b * a.dv_sb + head * a.dv_sh + (long long)(k0 + j) * a.dv_sn, dv0, dv1, h, 1.0f);
  }
}


// ---- single-pass backward: dk, dv and dq from ONE evaluation of S and dP (5 GEMMs) ----------------------------
// Eight waves per workgroup, each with a stationary 32-key tile (K^T, V^T in registers, dK^T / dV^T accumulators),
// walk the query tiles together.  dQ[32 q x 64 dims] = dS[q x 256 keys] K[keys x dims] needs dS with the QUERY as the
// row: every wave drops its dS tile (key on the lane) into a query-major LDS scratch, and after one barrier the
// eight waves each compute one 16x16 block of the dQ tile over ALL keys of the workgroup with
// v_mfma_f32_16x16x4_f32 (A = dS from the scratches, B = K from LDS copies of the key tiles): no reduction across
// waves, no atomics, deterministic.  The workgroup writes dq directly when it holds all keys of the head
// (Nk <= 256), else one partial per 256-key block that attn_dq_reduce_kernel folds in a fixed order.
// D = rowsum(dO o O) is computed while the dO tile is staged.  Both 16x16x4 operands are read as two ds_read_b128
// per lane and key tile: MFMA step u of lane group k (= lane / 16) contracts key 8k + u, so a lane's eight values
// are contiguous in the query-major dS scratch and in the dim-major K^T copy (row stride 36 floats: 16-byte
// aligned rows, eight lanes of a group land on eight different bank quads).
typedef float f32x4acc __attribute__((ext_vector_type(4)));
// FW = waves per workgroup (8: 256 keys, one workgroup per CU; 4: 128 keys, two per CU)
constexpr int kTs = 36;                      // row stride of the dS scratch     [32 q][32 keys]
constexpr int kKs = 36;                      // row stride of the K^T copies     [64 dims][32 keys]
constexpr int kWaveLds = kTK * kTs + kD * kKs;
constexpr int fused_lds_floats(int fw) { return 2 * kTK * kLd + 2 * kTK + fw * kWaveLds; }

struct AttnFusedArgs {
  AttnBwdArgs a;
  float* dq_partial;   // [n_kb, B*H, Nq, 64] when n_kb > 1
  float* dkv_partial;  // [n_qs, B*H, Nk, 2, 64] when the query tiles of a key block are split over n_qs workgroups
  int n_kb;
};

template <int kFW>
__global__ __launch_bounds__(kFW * 64, 2) void attn_bwd_fused_kernel(AttnFusedArgs fa) {
  constexpr int kFusedKeys = kFW * 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const AttnBwdArgs& a = fa.a;
  float* sQ = lds;                         // [32][68]
  float* sO = sQ + kTK * kLd;              // [32][68]  dO tile
  float* sL = sO + kTK * kLd;              // lse[32], delta[32]
  float* sW = sL + 2 * kTK;                // per wave: dS scratch [32 q][36], K^T copy [64 dims][36]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* sT = sW + wave * kWaveLds;
  float* sK = sT + kTK * kTs;
  const int j = lane & 31, h = lane >> 5;
  // grid = (B*H, n_kb, query splits): key-block-major dispatch order, so the short workgroups of the last block
  // (keys past limit_k, seen by the denoise queries only) start after every long one and fill the tail of the launch
  const int bh = blockIdx.x, b = bh / a.H, head = bh % a.H;
  const int kb = blockIdx.y;
  const int n_bh = gridDim.x;
  const int k0 = (kb * kFW + wave) * 32;
  const bool active = k0 < a.Nk;
  const float* qb = a.q + b * a.q_sb + head * a.q_sh;
  const long long orow0 = ((long long)b * a.Nq * a.H + head) * kD;  // row stride H*64 (out and dout)
  const float* dob = a.dout + orow0;
  const float* ob = a.o + orow0;
  const float* lb = a.lse + ((long long)b * a.H + head) * a.Nq;
  const long long do_sn = (long long)a.H * kD;

  const int blk_k0 = kb * kFusedKeys;
  const int blk_qstart = (blk_k0 >= a.limit_k) ? a.split_q : 0;
  const int my_qstart = active ? ((k0 >= a.limit_k) ? a.split_q : 0) : a.Nq;

  float kf[32], vf[32];
  if (active) {
    const float* kp = a.k + b * a.k_sb + head * a.k_sh + (long long)(k0 + j) * a.k_sn + 32 * h;
    const float* vp = a.v + b * a.v_sb + head * a.v_sh + (long long)(k0 + j) * a.v_sn + 32 * h;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const f32x4 k4 = *reinterpret_cast<const f32x4*>(kp + 4 * t);
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(vp + 4 * t);
      sK[(32 * h + 4 * t + 0) * kKs + j] = k4.x;   // K^T copy: [dim][key]
      sK[(32 * h + 4 * t + 1) * kKs + j] = k4.y;
      sK[(32 * h + 4 * t + 2) * kKs + j] = k4.z;
      sK[(32 * h + 4 * t + 3) * kKs + j] = k4.w;
      kf[4 * t + 0] = k4.x * a.scale_log2e;
      kf[4 * t + 1] = k4.y * a.scale_log2e;
      kf[4 * t + 2] = k4.z * a.scale_log2e;
      kf[4 * t + 3] = k4.w * a.scale_log2e;
      vf[4 * t + 0] = v4.x;
      vf[4 * t + 1] = v4.y;
      vf[4 * t + 2] = v4.z;
      vf[4 * t + 3] = v4.w;
    }
  } else {
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      kf[t] = 0.0f;
      vf[t] = 0.0f;
    }
  }

  f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};

  // staging: thread e covers row e>>4 (+ 16 kFW per pass), float4 column e&15 of the 32x64 tiles
  constexpr int kPass = 512 / (kFW * 64);   // 1 (512 threads) or 2 (256 threads)
  const int sr = tid >> 4, sc4 = (tid & 15) * 4;
  f32x4 qreg[kPass], oreg[kPass];
  float dpart[kPass], lreg = 0.0f;
  auto fetch_tile = [&](int q0) {
#pragma unroll
    for (int u = 0; u < kPass; ++u) {
      const int row = sr + u * (kFW * 4);
      qreg[u] = *reinterpret_cast<const f32x4*>(qb + (long long)(q0 + row) * a.q_sn + sc4);
      oreg[u] = *reinterpret_cast<const f32x4*>(dob + (long long)(q0 + row) * do_sn + sc4);
      const f32x4 o4 = *reinterpret_cast<const f32x4*>(ob + (long long)(q0 + row) * do_sn + sc4);
      dpart[u] = fmaf(oreg[u].x, o4.x, fmaf(oreg[u].y, o4.y, fmaf(oreg[u].z, o4.z, oreg[u].w * o4.w)));
    }
    if (tid < kTK) lreg = lb[q0 + tid];
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int u = 0; u < kPass; ++u) {
      const int row = sr + u * (kFW * 4);
      *reinterpret_cast<f32x4*>(&sQ[row * kLd + sc4]) = qreg[u];
      *reinterpret_cast<f32x4*>(&sO[row * kLd + sc4]) = oreg[u];
      float d = dpart[u];  // D of the row: sum over its 16 consecutive lanes
      d += __shfl_xor(d, 1);
      d += __shfl_xor(d, 2);
      d += __shfl_xor(d, 4);
      d += __shfl_xor(d, 8);
      if ((tid & 15) == 0) sL[kTK + row] = d;
    }
    if (tid < kTK) sL[tid] = lreg;
  };

  // this wave's 16x16 block(s) of the dQ tile: blocks wave, wave + kFW, ... of the 8
  const int li = lane & 15, lk = lane >> 4;

  // gridDim.z workgroups share the query tiles of a key block (each owns a contiguous range and its own dk/dv partial)
  const int n_qs = gridDim.z;
  const int t_first = blk_qstart / kTK, t_last = a.Nq / kTK;
  const int t_per = (t_last - t_first + n_qs - 1) / n_qs;
  const int t_begin = t_first + (int)blockIdx.z * t_per;
  const int t_end = min(t_last, t_begin + t_per);
  if (t_begin < t_end) {
    fetch_tile(t_begin * kTK);
    store_tile();
  }
  __syncthreads();
  for (int t = t_begin; t < t_end; ++t) {
    if (t + 1 < t_end) fetch_tile((t + 1) * kTK);
    if (t * kTK >= my_qstart) {
      f32x16 s = {0}, dp = {0};
      const float* qr = &sQ[j * kLd + 32 * h];
      const float* orow = &sO[j * kLd + 32 * h];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const f32x4 q4 = *reinterpret_cast<const f32x4*>(qr + 4 * g);
        const f32x4 o4 = *reinterpret_cast<const f32x4*>(orow + 4 * g);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.x, kf[4 * g + 0], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(o4.x, vf[4 * g + 0], dp, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.y, kf[4 * g + 1], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(o4.y, vf[4 * g + 1], dp, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.z, kf[4 * g + 2], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(o4.z, vf[4 * g + 2], dp, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.w, kf[4 * g + 3], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(o4.w, vf[4 * g + 3], dp, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(&sL[8 * g + 4 * h]);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(&sL[kTK + 8 * g + 4 * h]);
        const float p0 = __builtin_amdgcn_exp2f(s[4 * g + 0] - l4.x);
        const float p1 = __builtin_amdgcn_exp2f(s[4 * g + 1] - l4.y);
        const float p2 = __builtin_amdgcn_exp2f(s[4 * g + 2] - l4.z);
        const float p3 = __builtin_amdgcn_exp2f(s[4 * g + 3] - l4.w);
        s[4 * g + 0] = p0;
        s[4 * g + 1] = p1;
        s[4 * g + 2] = p2;
        s[4 * g + 3] = p3;
        dp[4 * g + 0] = p0 * (dp[4 * g + 0] - d4.x);
        dp[4 * g + 1] = p1 * (dp[4 * g + 1] - d4.y);
        dp[4 * g + 2] = p2 * (dp[4 * g + 2] - d4.z);
        dp[4 * g + 3] = p3 * (dp[4 * g + 3] - d4.w);
      }
      // dS (rows = queries (r&3)+8(r>>2)+4h, column = key j) -> scratch T[query][key]
#pragma unroll
      for (int r = 0; r < 16; ++r) sT[((r & 3) + 8 * (r >> 2) + 4 * h) * kTs + j] = dp[r];
      const float* oc = &sO[(4 * h) * kLd + j];
      const float* qc = &sQ[(4 * h) * kLd + j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        const float o0v = oc[row * kLd], o1v = oc[row * kLd + 32];
        const float q0v = qc[row * kLd], q1v = qc[row * kLd + 32];
        dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(o0v, s[r], dv0, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(q0v, dp[r], dk0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(o1v, s[r], dv1, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(q1v, dp[r], dk1, 0, 0, 0);
      }
    }
    __syncthreads();  // every wave's dS tile is in its scratch
#pragma unroll
    for (int blk = wave; blk < 8; blk += kFW) {
      // dQ block [bq..bq+16) x [bd..bd+16) over the keys of every wave that took part in this query tile
      const int bq = (blk & 1) * 16, bd = (blk >> 1) * 16;
      f32x4acc g = {0.f, 0.f, 0.f, 0.f};
      for (int w = 0; w < kFW; ++w) {
        const int wk0 = (kb * kFW + w) * 32;
        if (wk0 >= a.Nk || (wk0 >= a.limit_k && t * kTK < a.split_q)) continue;  // workgroup-uniform
        // step u of lane group lk contracts key 8*lk + u:  A[i][.] = dS[q = bq+i][8lk+u],  B[.][n] = K[8lk+u][bd+n]
        const float* tw = sW + w * kWaveLds + (bq + li) * kTs + 8 * lk;
        const float* kw = sW + w * kWaveLds + kTK * kTs + (bd + li) * kKs + 8 * lk;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(tw), a1 = *reinterpret_cast<const f32x4*>(tw + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(kw), b1 = *reinterpret_cast<const f32x4*>(kw + 4);
        g = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, g, 0, 0, 0);
        g = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, g, 0, 0, 0);
        g = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, g, 0, 0, 0);
        g = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, g, 0, 0, 0);
        g = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, g, 0, 0, 0);
        g = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, g, 0, 0, 0);
        g = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, g, 0, 0, 0);
        g = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, g, 0, 0, 0);
      }
      // g[r]: query bq + 4*lk + r, dim bd + li
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = t * kTK + bq + 4 * lk + r;
        if (fa.n_kb == 1)
          a.dq[b * a.dq_sb + head * a.dq_sh + (long long)q * a.dq_sn + bd + li] = g[r] * a.scale;
        else
          fa.dq_partial[(((long long)kb * n_bh + bh) * a.Nq + q) * kD + bd + li] = g[r];
      }
    }
    __syncthreads();  // all reads of sQ / sO / sL / the scratches done
    if (t + 1 < t_end) store_tile();
    __syncthreads();
  }

  if (active) {
    if (n_qs == 1) {
      store_dim_major(a.dk + b * a.dk_sb + head * a.dk_sh + (long long)(k0 + j) * a.dk_sn, dk0, dk1, h, a.scale);
      store_dim_major(a.dv + b * a.dv_sb + head * a.dv_sh + (long long)(k0 + j) * a.dv_sn, dv0, dv1, h, 1.0f);
    } else {
      float* pk = fa.dkv_partial + ((((long long)blockIdx.z * n_bh + bh) * a.Nk + (k0 + j)) * 2) * kD;
      store_dim_major(pk, dk0, dk1, h, a.scale);
      store_dim_major(pk + kD, dv0, dv1, h, 1.0f);
    }
  }
}

// dk / dv = sum over the query ranges (fixed order); one float4 of one (bh, key, k|v) row per thread
__global__ __launch_bounds__(256) void attn_dkv_reduce_kernel(const float* __restrict__ partial, int n_qs, long long BH, int H,
                                                              int Nk, float* __restrict__ dk, float* __restrict__ dv,
                                                              long long dk_sb, long long dk_sh, long long dk_sn,
                                                              long long dv_sb, long long dv_sh, long long dv_sn) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= BH * Nk * 32) return;
  const int c4 = (int)(e & 15) * 4, which = (int)(e >> 4) & 1;
  const long long row = e >> 5;  // bh * Nk + key
  const int key = (int)(row % Nk);
  const long long bh = row / Nk;
  const long long stride = BH * Nk * 2 * kD;
  const float* src = partial + (row * 2 + which) * kD + c4;
  f32x4 acc = *reinterpret_cast<const f32x4*>(src);
  for (int s = 1; s < n_qs; ++s) acc += *reinterpret_cast<const f32x4*>(src + s * stride);
  const long long b = bh / H, head = bh % H;
  float* dst = which ? dv + b * dv_sb + head * dv_sh + (long long)key * dv_sn : dk + b * dk_sb + head * dk_sh + (long long)key * dk_sn;
  *reinterpret_cast<f32x4*>(dst + c4) = acc;
}

// dq = scale * sum over the key blocks that reach the query (fixed order)
__global__ __launch_bounds__(256) void attn_dq_reduce_kernel(const float* __restrict__ partial, int n_kb, int keys_per_block,
                                                             long long BH, int H, int Nq, int split_q, int limit_k,
                                                             float scale,
                                                             float* __restrict__ dq, long long dq_sb, long long dq_sh,
                                                             long long dq_sn) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;  // one float4 of one (bh, q) row
  const long long rows = BH * Nq;
  if (e >= rows * 16) return;
  const long long row = e >> 4;
  const int c4 = (int)(e & 15) * 4;
  const long long bh = row / Nq;
  const int q = (int)(row - bh * Nq);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int kb = 0; kb < n_kb; ++kb) {
    if (q < split_q && kb * keys_per_block >= limit_k) continue;  // that block never visited this query
    const f32x4 v = *reinterpret_cast<const f32x4*>(partial + (((long long)kb * BH + bh) * Nq + q) * kD + c4);
    acc += v;
  }
  const long long b = bh / H, head = bh % H;
  *reinterpret_cast<f32x4*>(dq + b * dq_sb + head * dq_sh + (long long)q * dq_sn + c4) = acc * scale;
}

}  // namespace

PTOPS_API int ptops_attention_forward(const float* q, const float* k, const float* v, int B, int H, int Nq,
                                      int Nk, const long long* q_strides, const long long* k_strides,
                                      const long long* v_strides, int split_q, int limit_k, float scale,
                                      float* out, float* lse, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && H >= 1 && Nq >= 0 && Nk >= 1, "attention_forward: bad size");
  if (B == 0 || Nq == 0) return PTOPS_OK;
  PTOPS_REQUIRE(q && k && v && out && lse && q_strides && k_strides && v_strides, "attention_forward: null pointer");
  PTOPS_REQUIRE(Nq % 32 == 0 && Nk % 32 == 0, "attention_forward: token counts must be multiples of 32 (Nq=%d Nk=%d)", Nq, Nk);
  PTOPS_REQUIRE(split_q >= 0 && split_q <= Nq && split_q % 32 == 0 && limit_k >= 32 && limit_k <= Nk && limit_k % 32 == 0,
                "attention_forward: mask split (%d, %d) must be multiples of 32 within range", split_q, limit_k);
  for (int i = 0; i < 3; ++i)
    PTOPS_REQUIRE(q_strides[i] % 4 == 0 && k_strides[i] % 4 == 0 && v_strides[i] % 4 == 0,
                  "attention_forward: strides must keep rows 16-byte aligned");
  PTOPS_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0,
                "attention_forward: pointers must be 16-byte aligned");
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.o = out; a.lse = lse;
  a.q_sb = q_strides[0]; a.q_sh = q_strides[1]; a.q_sn = q_strides[2];
  a.k_sb = k_strides[0]; a.k_sh = k_strides[1]; a.k_sn = k_strides[2];
  a.v_sb = v_strides[0]; a.v_sh = v_strides[1]; a.v_sn = v_strides[2];
  a.H = H; a.Nq = Nq; a.Nk = Nk; a.split_q = split_q; a.limit_k = limit_k;
  a.scale_log2e = scale * 1.4426950408889634f;
  a.xcd_remap = !ptops::option("PTOPS_ATTN_NO_XCD_REMAP");
  const int tiles = Nq / 32;
  // Four waves per workgroup whenever there are four query tiles, a last workgroup with idle waves included
  // (576 queries = 18 tiles -> 5 workgroups per head): the 4-wave instantiation fits 128 VGPRs, so four workgroups
  // (16 waves) are resident per CU and e.g. all 960 workgroups of the 576-query training shape run in ONE round,
  // where the 3-wave form (occupancy 3, 1152 workgroups on 1024 slots) needed a second, nearly empty one.
  // Measured at B=32 (benchmarks/attention_bench.py, waves 1/2/3/4): 227/180/183/146 us (576x576 masked),
  // 108/89/90/77 us (576x256), 58/47/58/43 us (256x256).
  // >= 4 query tiles: the bf16-matrix-pipe kernel (fp32-accurate three-piece split, attn_fwd_split3_kernel);
  // PTOPS_ATTN_FWD_F32=1 keeps the f32-MFMA kernel
  if (tiles >= 4 && !ptops::option("PTOPS_ATTN_FWD_F32")) {
    const dim3 grid(ptops::ceil_div(tiles, 4), B * H);
    switch (ptops::option("PTOPS_ATTN_FWD_ABL")) {
#ifdef PTOPS_TUNING                                        // `make TUNING=1`: timing experiments (wrong results), see the kernel's header
#define PTOPS_ABL_CASE(N) case N: hipLaunchKernelGGL(attn_fwd_split3_kernel<N>, grid, dim3(256), 0, stream, a); break;
      PTOPS_ABL_CASE(1) PTOPS_ABL_CASE(2) PTOPS_ABL_CASE(3) PTOPS_ABL_CASE(4) PTOPS_ABL_CASE(7) PTOPS_ABL_CASE(8)
      PTOPS_ABL_CASE(16) PTOPS_ABL_CASE(24) PTOPS_ABL_CASE(32) PTOPS_ABL_CASE(35) PTOPS_ABL_CASE(39) PTOPS_ABL_CASE(63)
#undef PTOPS_ABL_CASE
#endif
      default:
        if (ptops::option("PTOPS_ATTN_PRIO")) hipLaunchKernelGGL((attn_fwd_split3_kernel<0, true, true>), grid, dim3(256), 0, stream, a);
        else if (ptops::option("PTOPS_ATTN_STAGE_EARLY")) hipLaunchKernelGGL((attn_fwd_split3_kernel<0, false>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((attn_fwd_split3_kernel<0, true>), grid, dim3(256), 0, stream, a);
        break;
    }
    return ptops::check_launch("attn_fwd_split3_kernel");
  }
  int W = tiles < 4 ? tiles : 4;
  if (const int force = ptops::option("PTOPS_ATTN_FWD_WAVES")) {            // benchmarking knob
    if (force >= 1 && force <= 4) W = force;
  }
  dim3 grid(ptops::ceil_div(tiles, W), B * H);
  switch (W) {
    case 4: hipLaunchKernelGGL(attn_fwd_kernel<4>, grid, dim3(256), 0, stream, a); break;
    case 3: hipLaunchKernelGGL(attn_fwd_kernel<3>, grid, dim3(192), 0, stream, a); break;
    case 2: hipLaunchKernelGGL(attn_fwd_kernel<2>, grid, dim3(128), 0, stream, a); break;
    default: hipLaunchKernelGGL(attn_fwd_kernel<1>, grid, dim3(64), 0, stream, a); break;
  }
  return ptops::check_launch("attn_fwd_kernel");
}

PTOPS_API int ptops_attention_backward(const float* q, const float* k, const float* v, const float* out,
                                       const float* dout, const float* lse, int B, int H, int Nq, int Nk,
                                       const long long* q_strides, const long long* k_strides,
                                       const long long* v_strides, const long long* dq_strides,
                                       const long long* dk_strides, const long long* dv_strides, int split_q,
                                       int limit_k, float scale, float* dq, float* dk, float* dv, float* delta,
                                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && H >= 1 && Nq >= 0 && Nk >= 1, "attention_backward: bad size");
  if (B == 0 || Nq == 0) return PTOPS_OK;
  PTOPS_REQUIRE(q && k && v && out && dout && lse && dq && dk && dv && delta && q_strides && k_strides && v_strides &&
                    dq_strides && dk_strides && dv_strides,
                "attention_backward: null pointer");
  PTOPS_REQUIRE(Nq % 32 == 0 && Nk % 32 == 0, "attention_backward: token counts must be multiples of 32 (Nq=%d Nk=%d)", Nq, Nk);
  PTOPS_REQUIRE(split_q >= 0 && split_q <= Nq && split_q % 32 == 0 && limit_k >= 32 && limit_k <= Nk && limit_k % 32 == 0,
                "attention_backward: mask split (%d, %d) must be multiples of 32 within range", split_q, limit_k);
  for (int i = 0; i < 3; ++i)
    PTOPS_REQUIRE(q_strides[i] % 4 == 0 && k_strides[i] % 4 == 0 && v_strides[i] % 4 == 0 && dq_strides[i] % 4 == 0 &&
                      dk_strides[i] % 4 == 0 && dv_strides[i] % 4 == 0,
                  "attention_backward: strides must keep rows 16-byte aligned");
  PTOPS_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dq |
                  (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
                "attention_backward: pointers must be 16-byte aligned");
  AttnBwdArgs a;
  a.q = q; a.k = k; a.v = v; a.o = out; a.dout = dout; a.lse = lse; a.delta = delta; a.dq = dq; a.dk = dk; a.dv = dv;
  a.q_sb = q_strides[0]; a.q_sh = q_strides[1]; a.q_sn = q_strides[2];
  a.k_sb = k_strides[0]; a.k_sh = k_strides[1]; a.k_sn = k_strides[2];
  a.v_sb = v_strides[0]; a.v_sh = v_strides[1]; a.v_sn = v_strides[2];
  a.dq_sb = dq_strides[0]; a.dq_sh = dq_strides[1]; a.dq_sn = dq_strides[2];
  a.dk_sb = dk_strides[0]; a.dk_sh = dk_strides[1]; a.dk_sn = dk_strides[2];
  a.dv_sb = dv_strides[0]; a.dv_sh = dv_strides[1]; a.dv_sn = dv_strides[2];
  a.H = H; a.Nq = Nq; a.Nk = Nk; a.split_q = split_q; a.limit_k = limit_k;
  a.scale = scale; a.scale_log2e = scale * 1.4426950408889634f;
  a.xcd_remap = !ptops::option("PTOPS_ATTN_NO_XCD_REMAP");
  // >= 4 tiles on both sides: the bf16-matrix-pipe pair (fp32-accurate three-piece split); PTOPS_ATTN_BWD_F32=1 keeps
  // the f32-MFMA kernels for A/B runs
  const bool f32_only = ptops::option("PTOPS_ATTN_BWD_F32") != 0;
  if (!f32_only && Nq >= 128 && Nk >= 128) {
    const bool late = !ptops::option("PTOPS_ATTN_STAGE_EARLY");     // A/B switch: split + LDS writes before the barrier
    if (late) hipLaunchKernelGGL(attn_bwd_dq_split3_kernel<true>, dim3(ptops::ceil_div(Nq / 32, 4), B * H), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(attn_bwd_dq_split3_kernel<false>, dim3(ptops::ceil_div(Nq / 32, 4), B * H), dim3(256), 0, stream, a);
    int rc = ptops::check_launch("attn_bwd_dq_split3_kernel");
    if (rc != PTOPS_OK) return rc;
    const int ktiles = Nk / 32;
    const int dkv_waves = ptops::option("PTOPS_ATTN_DKV_WAVES") == 4 ? 4 : 8;
    if (dkv_waves == 4 && late)
      hipLaunchKernelGGL((attn_bwd_dkv_split3_kernel<4, true>), dim3(ptops::ceil_div(ktiles, 4), B * H), dim3(256), 0, stream, a);
    else if (dkv_waves == 4)
      hipLaunchKernelGGL((attn_bwd_dkv_split3_kernel<4, false>), dim3(ptops::ceil_div(ktiles, 4), B * H), dim3(256), 0, stream, a);
    else if (late)
      hipLaunchKernelGGL((attn_bwd_dkv_split3_kernel<8, true>), dim3(ptops::ceil_div(ktiles, 8), B * H), dim3(512), 0, stream, a);
    else
      hipLaunchKernelGGL((attn_bwd_dkv_split3_kernel<8, false>), dim3(ptops::ceil_div(ktiles, 8), B * H), dim3(512), 0, stream, a);
    return ptops::check_launch("attn_bwd_dkv_split3_kernel");
  }
  {  // dq (+ delta): one wave per 32-query tile; two waves per SIMD fit, so 4-wave workgroups (two per CU)
     // fill a CU where 3-wave ones would leave a quarter of it empty; the last workgroup may be partly idle
    const int tiles = Nq / 32;
    const int W = tiles >= 4 ? 4 : (tiles >= 2 ? 2 : 1);
    dim3 grid(ptops::ceil_div(tiles, W), B * H);
    switch (W) {
      case 4: hipLaunchKernelGGL(attn_bwd_dq_kernel<4>, grid, dim3(256), 0, stream, a); break;
      case 2: hipLaunchKernelGGL(attn_bwd_dq_kernel<2>, grid, dim3(128), 0, stream, a); break;
      default: hipLaunchKernelGGL(attn_bwd_dq_kernel<1>, grid, dim3(64), 0, stream, a); break;
    }
    int rc = ptops::check_launch("attn_bwd_dq_kernel");
    if (rc != PTOPS_OK) return rc;
  }
  {  // dk, dv: one wave per 32-key tile, 4 per workgroup (the last workgroup may be partly idle)
    const int tiles = Nk / 32;
    const int W = tiles >= 4 ? 4 : (tiles >= 2 ? 2 : 1);
    dim3 grid(ptops::ceil_div(tiles, W), B * H);
    switch (W) {
      case 4: hipLaunchKernelGGL(attn_bwd_dkv_kernel<4>, grid, dim3(256), 0, stream, a); break;
      case 2: hipLaunchKernelGGL(attn_bwd_dkv_kernel<2>, grid, dim3(128), 0, stream, a); break;
      default: hipLaunchKernelGGL(attn_bwd_dkv_kernel<1>, grid, dim3(64), 0, stream, a); break;
    }
  }
  return ptops::check_launch("attn_bwd_dkv_kernel");
}

// 8 waves (256 keys) per workgroup: one workgroup holds every key of a head up to 256 keys (no dq partials) and
// the 576-key training shape needs 3 instead of 5 partials (465 vs 476 us at B=32, H=6); 4 waves (128 keys, two
// workgroups per CU) only where they tile the keys exactly and balance the CUs better (512 keys: 323 vs 385 us).
static int fused_waves(int Nk) { return (Nk > 256 && Nk % 256 == 0) ? 4 : 8; }

// Splitting the query tiles of every key block over two workgroups (each with its own dk/dv partial, summed by
// attn_dkv_reduce_kernel, ~40 us at the training shape) pays in two situations, both measured on the 8-wave kernel
// (benchmarks/attention_bench.py, attention_occupancy_probe.py; one workgroup per CU, launches run in whole rounds of
// 256 workgroups):
//  * the unsplit launch leaves at least half of the CU slots empty - the batch-16 model variants
//    (96 blocks: 201 -> 124 us for the 576 x 256 cross-attention);
//  * it needs >= 2 rounds and halves fill them better - the batch-32 decoder self-attention
//    (2 x 192 long blocks = 1.5 rounds -> 3 half rounds: 424 -> 394 us).
// It loses where the halves fill no better (192 blocks: 207 -> 229 us) and with the 4-wave kernel, whose workgroups
// share CUs in pairs.  PTOPS_ATTN_QSPLIT=1|2 overrides the choice for benchmarking.
static int fused_query_splits(long long BH, int Nq, int Nk, int limit_k) {
  const int fw = fused_waves(Nk);
  const long long slots = (long long)ptops::kNumCU * (fw == 8 ? 1 : 2);
  const long long blocks = BH * ptops::ceil_div(limit_k < Nk ? limit_k : Nk, fw * 32);   // the others see few queries
  if (Nq / kTK < 8) return 1;
  if (const int force = ptops::option("PTOPS_ATTN_QSPLIT")) {
    if (force == 1 || force == 2) return force;
  }
  if (Nq / kTK < 16) return 1;
  if (2 * blocks <= slots) return 2;
  const long long r1 = (blocks + slots - 1) / slots, r2x2 = (2 * blocks + slots - 1) / slots;   // rounds, 2 x half rounds
  return (fw == 8 && r1 >= 2 && 5 * r2x2 <= 8 * r1) ? 2 : 1;
}

static size_t fused_dq_partial_floats(int B, int H, int Nq, int Nk) {
  const int n_kb = ptops::ceil_div(Nk, fused_waves(Nk) * 32);
  return n_kb > 1 ? (size_t)n_kb * B * H * Nq * kD : 64;
}

PTOPS_API size_t ptops_attention_backward_fused_workspace_bytes(int B, int H, int Nq, int Nk) {
  if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return 0;
  // dq partials + room for two dk/dv partials (whether they are used depends on the mask split of the call)
  return sizeof(float) * (fused_dq_partial_floats(B, H, Nq, Nk) + (size_t)2 * B * H * Nk * 2 * kD);
}

PTOPS_API int ptops_attention_backward_fused(const float* q, const float* k, const float* v, const float* out,
                                             const float* dout, const float* lse, int B, int H, int Nq, int Nk,
                                             const long long* q_strides, const long long* k_strides,
                                             const long long* v_strides, const long long* dq_strides,
                                             const long long* dk_strides, const long long* dv_strides, int split_q,
                                             int limit_k, float scale, float* dq, float* dk, float* dv,
                                             void* workspace, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PTOPS_REQUIRE(B >= 0 && H >= 1 && Nq >= 0 && Nk >= 1, "attention_backward_fused: bad size");
  if (B == 0 || Nq == 0) return PTOPS_OK;
  PTOPS_REQUIRE(q && k && v && out && dout && lse && dq && dk && dv && workspace && q_strides && k_strides && v_strides &&
                    dq_strides && dk_strides && dv_strides,
                "attention_backward_fused: null pointer");
  PTOPS_REQUIRE(Nq % 32 == 0 && Nk % 32 == 0, "attention_backward_fused: token counts must be multiples of 32 (Nq=%d Nk=%d)", Nq, Nk);
  PTOPS_REQUIRE(split_q >= 0 && split_q <= Nq && split_q % 32 == 0 && limit_k >= 32 && limit_k <= Nk && limit_k % 32 == 0,
                "attention_backward_fused: mask split (%d, %d) must be multiples of 32 within range", split_q, limit_k);
  for (int i = 0; i < 3; ++i)
    PTOPS_REQUIRE(q_strides[i] % 4 == 0 && k_strides[i] % 4 == 0 && v_strides[i] % 4 == 0 && dq_strides[i] % 4 == 0 &&
                      dk_strides[i] % 4 == 0 && dv_strides[i] % 4 == 0,
                  "attention_backward_fused: strides must keep rows 16-byte aligned");
  PTOPS_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dq |
                  (uintptr_t)dk | (uintptr_t)dv | (uintptr_t)workspace) & 15) == 0,
                "attention_backward_fused: pointers must be 16-byte aligned");
  PTOPS_REQUIRE((long long)B * H <= 65535, "attention_backward_fused: B*H exceeds the grid limit");
  AttnFusedArgs fa;
  AttnBwdArgs& a = fa.a;
  a.q = q; a.k = k; a.v = v; a.o = out; a.dout = dout; a.lse = lse; a.delta = nullptr; a.dq = dq; a.dk = dk; a.dv = dv;
  a.xcd_remap = 0;
  a.q_sb = q_strides[0]; a.q_sh = q_strides[1]; a.q_sn = q_strides[2];
  a.k_sb = k_strides[0]; a.k_sh = k_strides[1]; a.k_sn = k_strides[2];
  a.v_sb = v_strides[0]; a.v_sh = v_strides[1]; a.v_sn = v_strides[2];
  a.dq_sb = dq_strides[0]; a.dq_sh = dq_strides[1]; a.dq_sn = dq_strides[2];
  a.dk_sb = dk_strides[0]; a.dk_sh = dk_strides[1]; a.dk_sn = dk_strides[2];
  a.dv_sb = dv_strides[0]; a.dv_sh = dv_strides[1]; a.dv_sn = dv_strides[2];
  a.H = H; a.Nq = Nq; a.Nk = Nk; a.split_q = split_q; a.limit_k = limit_k;
  a.scale = scale; a.scale_log2e = scale * 1.4426950408889634f;
  const int fw = fused_waves(Nk);
  fa.n_kb = ptops::ceil_div(Nk, fw * 32);
  fa.dq_partial = (float*)workspace;
  fa.dkv_partial = (float*)workspace + fused_dq_partial_floats(B, H, Nq, Nk);
  const int n_qs = fused_query_splits((long long)B * H, Nq, Nk, (split_q > 0 && limit_k < Nk) ? limit_k : Nk);
  const size_t lds = sizeof(float) * (size_t)fused_lds_floats(fw);
  // up to 128 KB of dynamic LDS: above the 64 KB default cap (idempotent per-function setting)
  if (fw == 8) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_kernel<8>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_bwd_fused_kernel<8>, dim3(B * H, fa.n_kb, n_qs), dim3(512), lds, stream, fa);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_kernel<4>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_bwd_fused_kernel<4>, dim3(B * H, fa.n_kb, n_qs), dim3(256), lds, stream, fa);
  }
  int rc = ptops::check_launch("attn_bwd_fused_kernel");
  if (rc == PTOPS_OK && n_qs > 1) {
    const long long vec2 = (long long)B * H * Nk * 32;
    hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3((unsigned)((vec2 + 255) / 256)), dim3(256), 0, stream,
                       (const float*)fa.dkv_partial, n_qs, (long long)B * H, H, Nk, dk, dv, a.dk_sb, a.dk_sh, a.dk_sn, a.dv_sb,
                       a.dv_sh, a.dv_sn);
    rc = ptops::check_launch("attn_dkv_reduce_kernel");
  }
  if (rc != PTOPS_OK || fa.n_kb == 1) return rc;
  const long long vec = (long long)B * H * Nq * 16;
  hipLaunchKernelGGL(attn_dq_reduce_kernel, dim3((unsigned)((vec + 255) / 256)), dim3(256), 0, stream,
                     (const float*)workspace, fa.n_kb, fw * 32, (long long)B * H, H, Nq, split_q, limit_k, scale, dq, a.dq_sb,
                     a.dq_sh, a.dq_sn);
  return ptops::check_launch("attn_dq_reduce_kernel");
}
