// attention.hip -- fused multi-head attention core for GroupFree3D's decoder layers
// (reference: detection/GroupFree3D/models/transformer.py:36-76 calling
// models/multi_head_attention.py:  softmax(q k^T / sqrt(d)) -> dropout -> . v  per head):
// one forward launch and two backward launches per attention instead of the bmm / softmax /
// dropout / bmm / transpose string, with nothing of size Lq x Lk in HBM (the backward rebuilds
// the probabilities from the saved log-sum-exp).
//
// Shapes at the reference's sizes: 8 heads of d = 36 (d_model 288), Lq = 256 queries,
// Lk = 256 (self) or 1024 (cross) keys, B = 4 scenes: 0.15-0.6 GFLOP per call, i.e. latency- and
// occupancy-bound, not FLOP-bound.  f32 MFMA (v_mfma_f32_32x32x2_f32) kernels: a wave owns 32
// queries (forward, dQ) or 32 keys (dK / dV), the four waves of a block split the OTHER axis and
// merge through LDS, so 1 024 waves exist at these sizes (one per SIMD).  MI355X, kernel time
// self / cross: forward 14 / 46 us, dQ 21 / 49, dK+dV 22 / 85 (rocprofv3); the GroupFree3D step
// takes 14.4 ms with them and 15.0 ms with the torch string they replace (scaling, bmm, softmax,
// dropout, bmm, head transposes, the head-averaged weights nobody reads).  A first version on
// the vector ALU (8 threads per query row, K / V tiles in LDS) was 3-4x slower than these
// (183 / 237 / 168 us cross): every FMA read an LDS operand, the MFMA form reads 1/32 as many.
// Ablation of the forward at the cross size: MFMAs 16 us, tile loads 10 us after switching the
// 32 single-float row loads per tile to 5 float4 loads (40 us before), softmax / LDS traffic /
// merge 20 us.
//
// q, k, v are read IN PLACE from the projection outputs: q[l][b][h*d + c] at q + l*q_sl + b*q_sb,
// same for k / v with their own strides, so packed (L, B, 3E) / (L, B, 2E) projections need no
// transposes; the output is (Lq, B, E), what the output projection consumes.
// Dropout: keep(b*H + h, i, j) is a counter-based hash of (seed, *step, element), identical in
// the three kernels; `seed` is a host value unique per call, `step` an optional device counter
// (so a replayed HIP graph draws new masks every replay).
#include "common.hpp"
#include "internal.hpp"

namespace btr {
namespace {

struct AttnArgs {
  int lq, lk, b, h, d;
  const float *q;
  long long q_sl, q_sb;
  const float *k, *v;
  long long kv_sl, kv_sb;
  float *out;          // out[l][b][h*d + c] at out + l*o_sl + b*o_sb (dout alike)
  long long o_sl, o_sb;
  float *lse;          // (b*h, lq)
  const float *dout;   // (lq, b, h*d)
  float *dsum;         // (b*h, lq): rowsum(dO * O)
  float *dq;
  long long dq_sl, dq_sb;
  float *dk, *dv;
  long long dkv_sl, dkv_sb;
  float scale, keep_inv;
  unsigned drop_threshold;  // keep iff hash >= threshold (0: no dropout)
  unsigned long long seed;
  const long long *step;
};


__device__ __forceinline__ unsigned mix32(unsigned long long x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return (unsigned)x;
}

// keep-mask of element (bh, i, j); key = seed mixed with the device step counter
__device__ __forceinline__ bool keep_elem(unsigned long long key, int bh, int i, int j, int lq,
                                          int lk, unsigned threshold) {
  if (threshold == 0u) return true;
  const unsigned long long e = ((unsigned long long)bh * lq + i) * lk + j;
  return mix32(key + e * 0x9e3779b97f4a7c15ull) >= threshold;
}

__device__ __forceinline__ unsigned long long drop_key(const AttnArgs &a) {
  unsigned long long key = a.seed;
  if (a.step) key ^= (unsigned long long)(*a.step) * 0xd6e8feb86659fd93ull;
  return key;
}

// =================================================================== f32-MFMA kernels
// v_mfma_f32_32x32x2_f32: lane l supplies A[row = l & 31][k = l >> 5] and B[k = l >> 5][col =
// l & 31]; register v of the result holds D[row = (v & 3) + 8 * (v >> 2) + 4 * (l >> 5)][col =
// l & 31].  One wave owns 32 queries (forward, dQ) or 32 keys (dK / dV).
//
// The score tile is computed TRANSPOSED, S^T = K Q^T (A = the K tile, B = Q): a lane then holds
// 16 keys of ONE query, so the softmax statistics over the keys are a reduction over the
// lane's own registers plus one exchange with lane ^ 32.  The second product needs
// P[query][key] as the A operand, one k slot per lane half -- exactly what the lane already
// holds, in a permuted key order: step s takes key (s & 3) + 8 * (s >> 2) + 4 * (l >> 5) from
// register s, and the B operand reads the V row of the same key.  A sum over keys does not
// care about the order, so P never moves.
constexpr int kMT = 32;            // tile edge

__device__ __forceinline__ int mfma_row(int v, int hh) { return (v & 3) + 8 * (v >> 2) + 4 * hh; }

// a wave's private 32 x d tile: global -> registers (every load of the tile in flight at once,
// issued a tile AHEAD of its use) -> LDS rows of stride ld; rows >= limit are zero.
// VEC: 16-byte loads (d, the strides and the base pointers are multiples of 4 floats): 32 * d/4
// float4 per tile = 5 instructions per lane at d = 36 instead of 32 single-float row loads
// (measured: the row loads alone were 40 of the forward kernel's 83 us).
template <bool VEC>
struct TileRegs {
  float4 r[8];   // VEC: element e = lane + 64 i -> row e / d4, float4 column e % d4
};
struct TileMap {           // per-lane (row, column) of the VEC elements, computed once
  short row[8], col[8];
};
__device__ __forceinline__ TileMap tile_map(int d, int lane) {
  TileMap tm;
  const int d4 = d >> 2;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = lane + 64 * i;
    const int r = d4 > 0 ? e / d4 : kMT;
    tm.row[i] = (short)(r < kMT ? r : -1);
    tm.col[i] = (short)(d4 > 0 ? 4 * (e - r * d4) : 0);
  }
  return tm;
}
template <bool VEC>
__device__ __forceinline__ void tile_fetch(TileRegs<VEC> &t, const TileMap &tm, const float *src,
                                           long long sl, int r0, int limit, int d, int lane) {
  if (VEC) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      t.r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tm.row[i] >= 0 && r0 + tm.row[i] < limit)
        t.r[i] = *reinterpret_cast<const float4 *>(src + (long long)(r0 + tm.row[i]) * sl +
                                                   tm.col[i]);
    }
  } else {   // 32 row loads of one float, 4 rows per register quad
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 4 * i + q;
        v[q] = (lane < d && r0 + r < limit) ? src[(long long)(r0 + r) * sl + lane] : 0.f;
      }
      t.r[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}
// ALIGNED: the LDS row stride is a multiple of 4 floats (16-byte stores)
template <bool VEC, bool ALIGNED>
__device__ __forceinline__ void tile_store(const TileRegs<VEC> &t, const TileMap &tm, float *tile,
                                           int ld, int d, int lane) {
  if (VEC) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (tm.row[i] >= 0) {
        float *p = tile + tm.row[i] * ld + tm.col[i];
        if (ALIGNED) {
          *reinterpret_cast<float4 *>(p) = t.r[i];
        } else {
          p[0] = t.r[i].x; p[1] = t.r[i].y; p[2] = t.r[i].z; p[3] = t.r[i].w;
        }
      }
  } else if (lane < d) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      tile[(4 * i + 0) * ld + lane] = t.r[i].x;
      tile[(4 * i + 1) * ld + lane] = t.r[i].y;
      tile[(4 * i + 2) * ld + lane] = t.r[i].z;
      tile[(4 * i + 3) * ld + lane] = t.r[i].w;
    }
  }
}

// KS: k-steps of the d reduction (2 * KS >= d); NT: 32-wide tiles over the d output columns
// NW: waves of the workgroup (they split the keys).  4: one wave per SIMD; 8: two -- with 256
// workgroups or fewer on 256 CUs a SIMD's only wave waited 63 % of its cycles (profiles/
// r06_attn_pmc.md: loads, LDS round trips, the MFMA results its own VALU work needs) and nothing
// else ran on its matrix unit meanwhile.
template <int KS, int NT, bool VEC, int NW>
__global__ __launch_bounds__(NW * 64) void attn_fwd_mfma_kernel(AttnArgs a) {
  constexpr int LDK = 2 * KS + 1;   // odd: fragment reads walk down the rows without conflicts
  constexpr int LDV = 32 * NT;
  constexpr int WAVE_F = kMT * LDK + kMT * LDV + 64;   // K tile, V tile, per-query exchange
  __shared__ __attribute__((aligned(16))) float smem[NW * WAVE_F];
  typedef float acc16 __attribute__((ext_vector_type(16)));
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.h, h = bh - b * a.h;
  const int q0 = blockIdx.x * kMT;
  const float *qp = a.q + (long long)b * a.q_sb + (long long)h * a.d;
  const float *kp = a.k + (long long)b * a.kv_sb + (long long)h * a.d;
  const float *vp = a.v + (long long)b * a.kv_sb + (long long)h * a.d;
  const unsigned long long key = drop_key(a);
  float *Kt = smem + w * WAVE_F, *Vt = Kt + kMT * LDK, *xch = Vt + kMT * LDV;
  for (int e = lane; e < kMT * LDK; e += 64) Kt[e] = 0.f;   // padding columns stay zero
  for (int e = lane; e < kMT * LDV; e += 64) Vt[e] = 0.f;

  // B operand of S^T = K Q^T: Q[query = l31][d index 2 kk + hh]
  float qf[KS];
  const int qi = q0 + l31;
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    const int c = 2 * kk + hh;
    qf[kk] = (qi < a.lq && c < a.d) ? qp[(long long)qi * a.q_sl + c] : 0.f;
  }
  acc16 o[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int v = 0; v < 16; ++v) o[nt][v] = 0.f;
  float m = -INFINITY, l = 0.f;

  const int ntiles = (a.lk + kMT - 1) / kMT;
  TileRegs<VEC> kt, vt;
  const TileMap tm = tile_map(a.d, lane);
  if (w < ntiles) {
    tile_fetch<VEC>(kt, tm, kp, a.kv_sl, w * kMT, a.lk, a.d, lane);
    tile_fetch<VEC>(vt, tm, vp, a.kv_sl, w * kMT, a.lk, a.d, lane);
  }
  for (int tile = w; tile < ntiles; tile += NW) {   // the waves split the keys
    const int k0 = tile * kMT;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();               // previous tile's LDS reads are done
    tile_store<VEC, false>(kt, tm, Kt, LDK, a.d, lane);
    tile_store<VEC, true>(vt, tm, Vt, LDV, a.d, lane);
    if (tile + NW < ntiles) {       // next tile's loads fly under this one's math
      tile_fetch<VEC>(kt, tm, kp, a.kv_sl, k0 + NW * kMT, a.lk, a.d, lane);
      tile_fetch<VEC>(vt, tm, vp, a.kv_sl, k0 + NW * kMT, a.lk, a.d, lane);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    acc16 st;
#pragma unroll
    for (int v = 0; v < 16; ++v) st[v] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(Kt[l31 * LDK + 2 * kk + hh], qf[kk], st, 0, 0, 0);
    // st[v] = S^T[key = mfma_row(v, hh)][query = l31]
    float tmax = -INFINITY;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int kj = k0 + mfma_row(v, hh);
      st[v] = kj < a.lk ? st[v] * a.scale : -INFINITY;
      tmax = fmaxf(tmax, st[v]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float mn = fmaxf(m, tmax);
    const float alpha = __expf(m - mn);
    m = mn;
    float psum = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const float p = __expf(st[v] - mn);
      psum += p;
      const bool kp_ = keep_elem(key, bh, qi, k0 + mfma_row(v, hh), a.lq, a.lk, a.drop_threshold);
      st[v] = kp_ ? p * a.keep_inv : 0.f;
    }
    l = l * alpha + psum;
    // the O accumulators hold query rows mfma_row(v, hh): fetch those queries' alpha
    if (hh == 0) xch[l31] = alpha;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 al = *reinterpret_cast<const float4 *>(xch + 8 * g + 4 * hh);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        o[nt][4 * g + 0] *= al.x;
        o[nt][4 * g + 1] *= al.y;
        o[nt][4 * g + 2] *= al.z;
        o[nt][4 * g + 3] *= al.w;
      }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int kr = mfma_row(s, hh);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        o[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(st[s], Vt[kr * LDV + 32 * nt + l31], o[nt],
                                                     0, 0, 0);
    }
  }
  // ---- merge the waves' partial (m, l, O) and write
  l += __shfl_xor(l, 32);
  __syncthreads();   // every wave is done with its tiles: the LDS is reused
  constexpr int LDO = 32 * NT + 1;
  float *Om = smem;                       // [NW][32][LDO]
  float *Mm = smem + NW * kMT * LDO;      // [NW][32]
  float *Lm = Mm + NW * kMT;              // [NW][32]
  static_assert(NW * kMT * LDO + 2 * NW * kMT <= NW * WAVE_F, "merge buffers fit the tile storage");
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int v = 0; v < 16; ++v)
      Om[(w * kMT + mfma_row(v, hh)) * LDO + 32 * nt + l31] = o[nt][v];
  if (hh == 0) {
    Mm[w * kMT + l31] = m;
    Lm[w * kMT + l31] = l;
  }
  __syncthreads();
  if (tid >= 256) return;   // (32 rows x 8 column slices write)
  const int row = tid >> 3, sub = tid & 7;
  const int i = q0 + row;
  float M = -INFINITY;
#pragma unroll
  for (int ww = 0; ww < NW; ++ww) M = fmaxf(M, Mm[ww * kMT + row]);
  float f[NW], L = 0.f;
#pragma unroll
  for (int ww = 0; ww < NW; ++ww) {
    f[ww] = __expf(Mm[ww * kMT + row] - M);
    L = fmaf(Lm[ww * kMT + row], f[ww], L);
  }
  if (i < a.lq) {
    const float inv = 1.f / L;
    float *op = a.out + (long long)i * a.o_sl + (long long)b * a.o_sb + (long long)h * a.d;
    for (int c = sub; c < a.d; c += 8) {
      float acc = 0.f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) acc = fmaf(Om[(ww * kMT + row) * LDO + c], f[ww], acc);
      op[c] = acc * inv;
    }
    if (sub == 0) a.lse[(long long)bh * a.lq + i] = M + __logf(L);
  }
}


// ---- backward, dQ (and D = rowsum(dO * O)): a wave owns 32 queries, the four waves of a block
// split the keys and their partial dQ are summed through LDS.  Per key tile:
//   S^T = K Q^T,  dP^T = V dO^T  (both in the forward's transposed layout: a lane = one query)
//   dS = P * (keep ? dP / (1 - p) : 0  -  D),   dQ += dS K   (dS as the A operand in place)
template <int KS, int NT, bool VEC, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_mfma_kernel(AttnArgs a) {
  constexpr int LDT = 2 * KS + 1;
  constexpr int WAVE_F = 2 * kMT * LDT + 32;
  constexpr int LDO = 32 * NT + 1;
  constexpr int SMEM = (NW * WAVE_F > NW * kMT * LDO) ? NW * WAVE_F : NW * kMT * LDO;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];
  typedef float acc16 __attribute__((ext_vector_type(16)));
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.h, h = bh - b * a.h;
  const int q0 = blockIdx.x * kMT;
  const float *qp = a.q + (long long)b * a.q_sb + (long long)h * a.d;
  const float *kp = a.k + (long long)b * a.kv_sb + (long long)h * a.d;
  const float *vp = a.v + (long long)b * a.kv_sb + (long long)h * a.d;
  const unsigned long long key = drop_key(a);
  float *Kt = smem + w * WAVE_F, *Vt = Kt + kMT * LDT;
  for (int i = lane; i < 2 * kMT * LDT; i += 64) Kt[i] = 0.f;

  const int qi = q0 + l31;
  const bool live = qi < a.lq;
  float qf[KS], dof[KS];
  float dsum = 0.f;
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    const int c = 2 * kk + hh;
    const bool ok = live && c < a.d;
    const long long o_at = (long long)qi * a.o_sl + (long long)b * a.o_sb + (long long)h * a.d + c;
    qf[kk] = ok ? qp[(long long)qi * a.q_sl + c] : 0.f;
    dof[kk] = ok ? a.dout[o_at] : 0.f;
    dsum = fmaf(dof[kk], ok ? a.out[o_at] : 0.f, dsum);
  }
  dsum += __shfl_xor(dsum, 32);
  const float lse = live ? a.lse[(long long)bh * a.lq + qi] : INFINITY;
  if (live && w == 0 && hh == 0) a.dsum[(long long)bh * a.lq + qi] = dsum;
  acc16 dq[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int v = 0; v < 16; ++v) dq[nt][v] = 0.f;

  const int ntiles = (a.lk + kMT - 1) / kMT;
  TileRegs<VEC> kt, vt;
  const TileMap tm = tile_map(a.d, lane);
  if (w < ntiles) {
    tile_fetch<VEC>(kt, tm, kp, a.kv_sl, w * kMT, a.lk, a.d, lane);
    tile_fetch<VEC>(vt, tm, vp, a.kv_sl, w * kMT, a.lk, a.d, lane);
  }
  for (int tile = w; tile < ntiles; tile += NW) {
    const int k0 = tile * kMT;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    tile_store<VEC, false>(kt, tm, Kt, LDT, a.d, lane);
    tile_store<VEC, false>(vt, tm, Vt, LDT, a.d, lane);
    if (tile + NW < ntiles) {
      tile_fetch<VEC>(kt, tm, kp, a.kv_sl, k0 + NW * kMT, a.lk, a.d, lane);
      tile_fetch<VEC>(vt, tm, vp, a.kv_sl, k0 + NW * kMT, a.lk, a.d, lane);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    acc16 st, dp;
#pragma unroll
    for (int v = 0; v < 16; ++v) st[v] = dp[v] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(Kt[l31 * LDT + 2 * kk + hh], qf[kk], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Vt[l31 * LDT + 2 * kk + hh], dof[kk], dp, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) {   // [key = mfma_row(v, hh)][query = l31]
      const int kj = k0 + mfma_row(v, hh);
      const float p = kj < a.lk ? __expf(st[v] * a.scale - lse) : 0.f;
      const bool kp_ = keep_elem(key, bh, qi, kj, a.lq, a.lk, a.drop_threshold);
      st[v] = p * ((kp_ ? dp[v] * a.keep_inv : 0.f) - dsum);
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int kr = mfma_row(s, hh);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c = 32 * nt + l31;
        const float bv = c < a.d ? Kt[kr * LDT + c] : 0.f;
        dq[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(st[s], bv, dq[nt], 0, 0, 0);
      }
    }
  }
  __syncthreads();
  float *Om = smem;   // [NW][32][LDO]
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int v = 0; v < 16; ++v)
      Om[(w * kMT + mfma_row(v, hh)) * LDO + 32 * nt + l31] = dq[nt][v];
  __syncthreads();
  if (tid >= 256) return;
  const int row = tid >> 3, sub = tid & 7;
  const int i = q0 + row;
  if (i < a.lq) {
    float *dqp = a.dq + (long long)i * a.dq_sl + (long long)b * a.dq_sb + (long long)h * a.d;
    for (int c = sub; c < a.d; c += 8) {
      float acc = (Om[row * LDO + c] + Om[(kMT + row) * LDO + c]) +
                  (Om[(2 * kMT + row) * LDO + c] + Om[(3 * kMT + row) * LDO + c]);
      if (NW == 8)
        acc += (Om[(4 * kMT + row) * LDO + c] + Om[(5 * kMT + row) * LDO + c]) +
               (Om[(6 * kMT + row) * LDO + c] + Om[(7 * kMT + row) * LDO + c]);
      dqp[c] = acc * a.scale;
    }
  }
}

// ---- backward, dK and dV: a block owns 32 keys, its four waves split the QUERY tiles and their
// partial sums are added through LDS.  Here a lane holds ONE key and 16 queries:
//   S = Q K^T,  dP = dO V^T   (A = the Q / dO tile, B = this lane's K / V row)
//   dV += P~^T dO,  dK += dS^T Q   (P~ / dS as the A operand in place, queries permuted)
template <int KS, int NT, bool VEC>
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma_kernel(AttnArgs a) {
  constexpr int LDT = 2 * KS + 1;
  constexpr int WAVE_F = 2 * kMT * LDT + 64;
  constexpr int LDO = 32 * NT + 1;
  constexpr int SMEM = (4 * WAVE_F > 4 * kMT * LDO) ? 4 * WAVE_F : 4 * kMT * LDO;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];
  typedef float acc16 __attribute__((ext_vector_type(16)));
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.h, h = bh - b * a.h;
  const int j0 = blockIdx.x * kMT;
  const float *qp = a.q + (long long)b * a.q_sb + (long long)h * a.d;
  const float *kp = a.k + (long long)b * a.kv_sb + (long long)h * a.d;
  const float *vp = a.v + (long long)b * a.kv_sb + (long long)h * a.d;
  const float *dop = a.dout + (long long)b * a.o_sb + (long long)h * a.d;   // row stride o_sl
  const unsigned long long key = drop_key(a);
  float *Qt = smem + w * WAVE_F, *Ot = Qt + kMT * LDT, *Ls = Ot + kMT * LDT, *Ds = Ls + 32;
  for (int i = lane; i < 2 * kMT * LDT; i += 64) Qt[i] = 0.f;

  const int kj = j0 + l31;
  const bool live = kj < a.lk;
  float kf[KS], vf[KS];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    const int c = 2 * kk + hh;
    const bool ok = live && c < a.d;
    kf[kk] = ok ? kp[(long long)kj * a.kv_sl + c] : 0.f;
    vf[kk] = ok ? vp[(long long)kj * a.kv_sl + c] : 0.f;
  }
  acc16 dk[NT], dv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int v = 0; v < 16; ++v) dk[nt][v] = dv[nt][v] = 0.f;

  const int ntiles = (a.lq + kMT - 1) / kMT;
  TileRegs<VEC> qt, ot;
  const TileMap tm = tile_map(a.d, lane);
  if (w < ntiles) {
    tile_fetch<VEC>(qt, tm, qp, a.q_sl, w * kMT, a.lq, a.d, lane);
    tile_fetch<VEC>(ot, tm, dop, a.o_sl, w * kMT, a.lq, a.d, lane);
  }
  for (int tile = w; tile < ntiles; tile += 4) {
    const int q0 = tile * kMT;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    tile_store<VEC, false>(qt, tm, Qt, LDT, a.d, lane);
    tile_store<VEC, false>(ot, tm, Ot, LDT, a.d, lane);
    if (lane < 32) {
      const bool ok = q0 + lane < a.lq;
      Ls[lane] = ok ? a.lse[(long long)bh * a.lq + q0 + lane] : INFINITY;
      Ds[lane] = ok ? a.dsum[(long long)bh * a.lq + q0 + lane] : 0.f;
    }
    if (tile + 4 < ntiles) {
      tile_fetch<VEC>(qt, tm, qp, a.q_sl, q0 + 4 * kMT, a.lq, a.d, lane);
      tile_fetch<VEC>(ot, tm, dop, a.o_sl, q0 + 4 * kMT, a.lq, a.d, lane);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    acc16 st, dp;
#pragma unroll
    for (int v = 0; v < 16; ++v) st[v] = dp[v] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(Qt[l31 * LDT + 2 * kk + hh], kf[kk], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Ot[l31 * LDT + 2 * kk + hh], vf[kk], dp, 0, 0, 0);
    }
    // [query = mfma_row(v, hh)][key = l31]: st <- dS, dp <- P~
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 ls = *reinterpret_cast<const float4 *>(Ls + 8 * g + 4 * hh);
      const float4 dd = *reinterpret_cast<const float4 *>(Ds + 8 * g + 4 * hh);
      const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, ddv[4] = {dd.x, dd.y, dd.z, dd.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int v = 4 * g + q;
        const int qi = q0 + mfma_row(v, hh);
        const float p = (live && qi < a.lq) ? __expf(st[v] * a.scale - lsv[q]) : 0.f;
        const bool kp_ = keep_elem(key, bh, qi, kj, a.lq, a.lk, a.drop_threshold);
        st[v] = p * ((kp_ ? dp[v] * a.keep_inv : 0.f) - ddv[q]);
        dp[v] = kp_ ? p * a.keep_inv : 0.f;
      }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int qr = mfma_row(s, hh);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c = 32 * nt + l31;
        const float bo = c < a.d ? Ot[qr * LDT + c] : 0.f;
        const float bq = c < a.d ? Qt[qr * LDT + c] : 0.f;
        dv[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[s], bo, dv[nt], 0, 0, 0);
        dk[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(st[s], bq, dk[nt], 0, 0, 0);
      }
    }
  }
  float *Om = smem;   // [4][32][LDO], used for dK, then for dV
  const int row = tid >> 3, sub = tid & 7;
  const int jj = j0 + row;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int v = 0; v < 16; ++v)
        Om[(w * kMT + mfma_row(v, hh)) * LDO + 32 * nt + l31] = which ? dv[nt][v] : dk[nt][v];
    __syncthreads();
    if (jj < a.lk) {
      float *dst = (which ? a.dv : a.dk) + (long long)jj * a.dkv_sl + (long long)b * a.dkv_sb +
                   (long long)h * a.d;
      const float f = which ? 1.f : a.scale;
      for (int c = sub; c < a.d; c += 8)
        dst[c] = ((Om[row * LDO + c] + Om[(kMT + row) * LDO + c]) +
                  (Om[(2 * kMT + row) * LDO + c] + Om[(3 * kMT + row) * LDO + c])) * f;
    }
  }
}

// BTR_ATTN_WAVES=4: never the eight-wave form (read per call: the tests toggle it).  Measured at
// the decoder's shapes (4 x 8 heads, d = 36; tools/attn_bench.py): 256 queries x 1 024 keys forward
// 45.3 -> 37.3 us, 256 x 256 16.8 -> 17.5 (one key tile per wave: nothing to overlap)
bool wide_waves(long long workgroups, int lk) {
  const char *e = getenv("BTR_ATTN_WAVES");
  if (e && e[0] == '4') return false;
  return workgroups <= 512 && lk >= 16 * kMT;
}

// 16-byte loads of the q / k / v / dout rows are possible
bool vec_ok(const AttnArgs &a) {
  auto al = [](const void *p) { return p == nullptr || ((size_t)p & 15) == 0; };
  return a.d % 4 == 0 && a.q_sl % 4 == 0 && a.q_sb % 4 == 0 && a.kv_sl % 4 == 0 &&
         a.kv_sb % 4 == 0 && a.o_sl % 4 == 0 && a.o_sb % 4 == 0 && al(a.q) && al(a.k) &&
         al(a.v) && al(a.dout) && al(a.out);
}

int fill_dropout(AttnArgs &a, float p) {
  if (!(p >= 0.f && p < 1.f)) return fail(BTR_ERR_INVALID_ARGUMENT, "attention: dropout %g", p);
  if (p > 0.f) {
    a.drop_threshold = (unsigned)((double)p * 4294967296.0);
    if (a.drop_threshold == 0u) a.drop_threshold = 1u;
    a.keep_inv = 1.f / (1.f - p);
  } else {
    a.drop_threshold = 0u;
    a.keep_inv = 1.f;
  }
  return BTR_OK;
}

}  // namespace
}  // namespace btr

using namespace btr;

extern "C" {

int btr_attention_supported(int d) { return d >= 1 && d <= 64; }

int btr_attention_fwd(int lq, int lk, int b, int h, int d, const float *q, long long q_sl,
                      long long q_sb, const float *k, const float *v, long long kv_sl,
                      long long kv_sb, float *out, float *lse, float scale, float dropout_p,
                      unsigned long long seed, const long long *step, btr_stream_t stream) {
  return attention_fwd_strided(lq, lk, b, h, d, q, q_sl, q_sb, k, v, kv_sl, kv_sb, out,
                               (long long)b * h * d, (long long)h * d, lse, scale, dropout_p,
                               seed, step, stream);
}
}  // extern "C"

int btr::attention_fwd_strided(int lq, int lk, int b, int h, int d, const float *q,
                               long long q_sl, long long q_sb, const float *k, const float *v,
                               long long kv_sl, long long kv_sb, float *out, long long o_sl,
                               long long o_sb, float *lse, float scale, float dropout_p,
                               unsigned long long seed, const long long *step,
                               btr_stream_t stream) {
  if (lq <= 0 || b <= 0 || h <= 0) return BTR_OK;
  BTR_REQUIRE(q && k && v && out && lse && lk > 0 && btr_attention_supported(d),
              "attention_fwd: null pointer, no keys or head width %d not in 1..64", d);
  BTR_REQUIRE((long long)b * h <= 65535, "attention_fwd: %d x %d heads", b, h);
  AttnArgs a{};
  a.lq = lq; a.lk = lk; a.b = b; a.h = h; a.d = d;
  a.q = q; a.q_sl = q_sl; a.q_sb = q_sb;
  a.k = k; a.v = v; a.kv_sl = kv_sl; a.kv_sb = kv_sb;
  a.out = out; a.o_sl = o_sl; a.o_sb = o_sb; a.lse = lse; a.scale = scale; a.seed = seed; a.step = step;
  if (int rc = fill_dropout(a, dropout_p)) return rc;
  const dim3 grid(cdiv(lq, kMT), b * h);
  hipStream_t s = as_stream(stream);
  {
    const bool vec = vec_ok(a);
    // eight waves where the grid leaves CUs a single workgroup (<= 2 per CU) and every wave
    // still gets a key tile
    const bool wide = wide_waves((long long)grid.x * grid.y, lk);
#define BTR_ATTN_M(KS, NT)                                                                   \
  do {                                                                                       \
    if (vec && wide)                                                                         \
      hipLaunchKernelGGL((attn_fwd_mfma_kernel<KS, NT, true, 8>), grid, dim3(512), 0, s, a); \
    else if (vec)                                                                            \
      hipLaunchKernelGGL((attn_fwd_mfma_kernel<KS, NT, true, 4>), grid, dim3(256), 0, s, a); \
    else                                                                                     \
      hipLaunchKernelGGL((attn_fwd_mfma_kernel<KS, NT, false, 4>), grid, dim3(256), 0, s, a); \
  } while (0)
    if (d <= 16) BTR_ATTN_M(8, 1);
    else if (d <= 32) BTR_ATTN_M(16, 1);
    else if (d <= 36) BTR_ATTN_M(18, 2);
    else if (d <= 48) BTR_ATTN_M(24, 2);
    else BTR_ATTN_M(32, 2);
#undef BTR_ATTN_M
    return check_launch("attention_fwd");
  }
}

extern "C" {

int btr_attention_bwd(int lq, int lk, int b, int h, int d, const float *q, long long q_sl,
                      long long q_sb, const float *k, const float *v, long long kv_sl,
                      long long kv_sb, const float *out, const float *dout, const float *lse,
                      float *dsum, float *dq, long long dq_sl, long long dq_sb, float *dk,
                      float *dv, long long dkv_sl, long long dkv_sb, float scale, float dropout_p,
                      unsigned long long seed, const long long *step, btr_stream_t stream) {
  return attention_bwd_strided(lq, lk, b, h, d, q, q_sl, q_sb, k, v, kv_sl, kv_sb, out, dout,
                               (long long)b * h * d, (long long)h * d, lse, dsum, dq, dq_sl,
                               dq_sb, dk, dv, dkv_sl, dkv_sb, scale, dropout_p, seed, step,
                               stream);
}
}  // extern "C"

int btr::attention_bwd_strided(int lq, int lk, int b, int h, int d, const float *q,
                               long long q_sl, long long q_sb, const float *k, const float *v,
                               long long kv_sl, long long kv_sb, const float *out,
                               const float *dout, long long o_sl, long long o_sb,
                               const float *lse, float *dsum, float *dq, long long dq_sl,
                               long long dq_sb, float *dk, float *dv, long long dkv_sl,
                               long long dkv_sb, float scale, float dropout_p,
                               unsigned long long seed, const long long *step,
                               btr_stream_t stream) {
  return attention_bwd_strided_parts(lq, lk, b, h, d, q, q_sl, q_sb, k, v, kv_sl, kv_sb, out,
                                     dout, o_sl, o_sb, lse, dsum, dq, dq_sl, dq_sb, dk, dv,
                                     dkv_sl, dkv_sb, scale, dropout_p, seed, step,
                                     kAttnBwdQ | kAttnBwdKV, stream);
}

// parts: kAttnBwdQ = the dQ kernel (it also writes dsum = rowsum(dO * O), which the other reads);
// kAttnBwdKV = the dK / dV kernel -- after the first, on the same stream or one ordered behind it
int btr::attention_bwd_strided_parts(int lq, int lk, int b, int h, int d, const float *q,
                                     long long q_sl, long long q_sb, const float *k,
                                     const float *v, long long kv_sl, long long kv_sb,
                                     const float *out, const float *dout, long long o_sl,
                                     long long o_sb, const float *lse, float *dsum, float *dq,
                                     long long dq_sl, long long dq_sb, float *dk, float *dv,
                                     long long dkv_sl, long long dkv_sb, float scale,
                                     float dropout_p, unsigned long long seed,
                                     const long long *step, int parts, btr_stream_t stream) {
  if (lq <= 0 || b <= 0 || h <= 0) return BTR_OK;
  const bool do_q = (parts & kAttnBwdQ) != 0, do_kv = (parts & kAttnBwdKV) != 0;
  BTR_REQUIRE(q && k && v && out && dout && lse && dsum && dq && dk && dv && lk > 0 &&
                  btr_attention_supported(d),
              "attention_bwd: null pointer, no keys or head width %d not in 1..64", d);
  BTR_REQUIRE((long long)b * h <= 65535, "attention_bwd: %d x %d heads", b, h);
  AttnArgs a{};
  a.lq = lq; a.lk = lk; a.b = b; a.h = h; a.d = d;
  a.q = q; a.q_sl = q_sl; a.q_sb = q_sb;
  a.k = k; a.v = v; a.kv_sl = kv_sl; a.kv_sb = kv_sb;
  a.out = const_cast<float *>(out); a.o_sl = o_sl; a.o_sb = o_sb; a.lse = const_cast<float *>(lse); a.dout = dout;
  a.dsum = dsum; a.dq = dq; a.dq_sl = dq_sl; a.dq_sb = dq_sb;
  a.dk = dk; a.dv = dv; a.dkv_sl = dkv_sl; a.dkv_sb = dkv_sb;
  a.scale = scale; a.seed = seed; a.step = step;
  if (int rc = fill_dropout(a, dropout_p)) return rc;
  hipStream_t s = as_stream(stream);
  const dim3 gq(cdiv(lq, kMT), b * h), gk(cdiv(lk, kMT), b * h);
  {
    const bool vec = vec_ok(a);
    // (the eight-wave form of the dQ kernel measured no faster inside the decoder stack: 3.59
    // against 3.56 ms per backward; BTR_ATTN_WAVES=8 selects it)
    const char *we = getenv("BTR_ATTN_WAVES");
    const bool wide = we && we[0] == '8' && wide_waves((long long)gq.x * gq.y, lk);
#define BTR_ATTN_M(KS, NT)                                                                     \
  do {                                                                                         \
    if (vec) {                                                                                 \
      if (do_q && wide)                                                                        \
        hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<KS, NT, true, 8>), gq, dim3(512), 0, s, a); \
      else if (do_q)                                                                           \
        hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<KS, NT, true, 4>), gq, dim3(256), 0, s, a); \
      if (do_kv)                                                                               \
        hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<KS, NT, true>), gk, dim3(256), 0, s, a);  \
    } else {                                                                                   \
      if (do_q)                                                                                \
        hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<KS, NT, false, 4>), gq, dim3(256), 0, s, a); \
      if (do_kv)                                                                               \
        hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<KS, NT, false>), gk, dim3(256), 0, s, a); \
    }                                                                                          \
  } while (0)
    if (d <= 16) BTR_ATTN_M(8, 1);
    else if (d <= 32) BTR_ATTN_M(16, 1);
    else if (d <= 36) BTR_ATTN_M(18, 2);
    else if (d <= 48) BTR_ATTN_M(24, 2);
    else BTR_ATTN_M(32, 2);
#undef BTR_ATTN_M
    return check_launch("attention_bwd");
  }
}
