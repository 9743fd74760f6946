// sa_mlp.hip -- the per-group shared MLP + max-pool of a PointNet++ set-abstraction layer
// (reference: pointnet2_utils.QueryAndGroup :317-376 + pytorch_utils.SharedMLP :11-36 +
// F.max_pool2d, pointnet2_modules.py:243-267) as hand-written gfx950 kernels.
//
// The reference materialises the grouped tensor (B, 3+C, npoint, nsample) in NCHW and runs
// cuDNN 1x1 conv / BatchNorm / ReLU / max-pool as separate passes over it.  Here activations
// live channel-last, one row per (b, centre, sample): r = (b*M + m)*S + s, so that
//   * the 1x1 conv is a plain (rows x C_in) . (C_out x C_in)^T contraction on the f32 MFMA
//     (v_mfma_f32_32x32x2_f32, exact f32, 157 TF peak) with LDS-staged 128-row tiles;
//   * train-mode BatchNorm splits into (a) per-channel sum / sum-of-squares accumulated in
//     the GEMM epilogue (per-workgroup partials, reduced in f64: deterministic, no atomics),
//     (b) the affine+ReLU applied in the NEXT GEMM's prologue while the tile is staged to
//     LDS -- post-activation tensors are never written to HBM;
//   * max-pool + BN + ReLU of the last layer is one pass that also records the arg-max.
// Backward mirrors it: BN/ReLU/max-pool backward are folded into small elementwise passes,
// dgrad re-uses the NT GEMM (with W^T), wgrad is a TN GEMM reducing over rows with
// per-chunk partials (deterministic).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include <atomic>
#include <mutex>
#include <vector>

#include "internal.hpp"

namespace btr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ---- "bf16x6": an f32 product on the bf16 matrix pipe ----------------------------------------
// v_mfma_f32_32x32x2_f32 retires 2 k per 64 cycles, v_mfma_f32_32x32x16_bf16 16 k per 32: the
// bf16 pipe is 16x the f32-input one.  Every f32 operand is split into three bf16 pieces,
// a = ah + am + al, each the round-to-nearest of what the previous ones left -- 3 x 8 significant
// bits carry the 24 of an f32, the split is exact.  A bf16 x bf16 product is exact in f32, so
//   a*b = ah*bh + (ah*bm + am*bh) + (ah*bl + al*bh + am*bm) + [am*bl + al*bm + al*bl]
// where the bracket is <= 2^-23 |a*b| (the size of one f32 rounding) and is dropped; the six
// kept terms are accumulated in f32 by the MFMA, smallest first.  Six bf16 instructions per
// 16 k = 2.67x the f32-input rate at the same accuracy: against a float64 evaluation both
// forms sit at 2e-7 (max) / 2e-8 (mean) of max|C| on the layer shapes of the benchmark step
// (tools/probe/gemm_lab.hip).  BTR_GEMM=f32 selects the f32-input MFMA kernels instead.
struct Split4 {
  bf16x4 h, m, l;
};
// Two elements at a time: gfx950 converts a PAIR of f32 to packed bf16 in one instruction
// (v_cvt_pk_bf16_f32, round to nearest even -- what the scalar conversions compile to as well,
// one pair slot wasted each), and a packed pair widens back to two f32 with a shift and a mask.
// 11 VALU ops per pair instead of ~17, results bit-identical to the element-wise form, and the
// planes come out packed (no v_or to assemble the 8-byte LDS stores).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 widen2(const bf16x2 b) {
  const unsigned u = __builtin_bit_cast(unsigned, b);
  f32x2 r;
  r.x = __uint_as_float(u << 16);
  r.y = __uint_as_float(u & 0xffff0000u);
  return r;
}
__device__ __forceinline__ Split4 split4(const float4 v) {
  Split4 s;
  const f32x2 f[2] = {{v.x, v.y}, {v.z, v.w}};
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    // (the differences element by element: packed f32 arithmetic beside MFMAs is an anti-lever)
    const bf16x2 h = __builtin_convertvector(f[p], bf16x2);
    const f32x2 wh = widen2(h);
    f32x2 r1;
    r1.x = f[p].x - wh.x;
    r1.y = f[p].y - wh.y;
    const bf16x2 m = __builtin_convertvector(r1, bf16x2);
    const f32x2 wm = widen2(m);
    f32x2 r2;
    r2.x = r1.x - wm.x;
    r2.y = r1.y - wm.y;
    const bf16x2 l = __builtin_convertvector(r2, bf16x2);
    s.h[2 * p] = h.x; s.h[2 * p + 1] = h.y;
    s.m[2 * p] = m.x; s.m[2 * p + 1] = m.y;
    s.l[2 * p] = l.x; s.l[2 * p + 1] = l.y;
  }
  return s;
}
__device__ __forceinline__ void split1(float v, __bf16 &h, __bf16 &m, __bf16 &l) {
  const f32x2 f = {v, 0.f};
  const bf16x2 bh = __builtin_convertvector(f, bf16x2);
  const f32x2 r1 = {v - widen2(bh).x, 0.f};
  const bf16x2 bm = __builtin_convertvector(r1, bf16x2);
  const f32x2 r2 = {r1.x - widen2(bm).x, 0.f};
  const bf16x2 bl = __builtin_convertvector(r2, bf16x2);
  h = bh.x; m = bm.x; l = bl.x;
}
constexpr int kLp = 40;   // row pitch of a bf16 plane (80 B: conflict-free ds_read_b128)

// "Compact rows" (csrc comment block further down, btr_sac_plan): device-side description of
// a row array in which every group keeps only ceil8(#distinct neighbours) rows.
//   dims[0] = number of rows, dims[1] = number of 8-row blocks
//   bw[block]   = weight of the block's FIRST row (1 + S - len for a group's first block, else 1)
//   bgrp[block] = group (b * M + m) the block belongs to,  goff[g] = first row of group g
struct Compact {
  const int *dims = nullptr;
  const float *bw = nullptr;
  const int *bgrp = nullptr;
  const int *goff = nullptr;
  // split-K (pm_gemm_nt_splitk): workgroups of blockIdx.z = z reduce k in [z*kz, (z+1)*kz) and
  // write the plane C + z*czs; kz == 0: off
  int kz = 0;
  long long czs = 0;
};

constexpr int kBM = 128;      // rows per workgroup tile
constexpr int kBK = 32;       // reduction chunk staged in LDS
constexpr int kLd = kBK + 4;  // padded LDS row (floats): conflict-free ds_read_b128
constexpr int kMaxK = 512;    // largest reduction length of a GEMM with a prologue (LDS tables)

// ----------------------------------------------------------------------------- group gather
// X0[r][c], r = (b*M + m)*S + s:  c < 3: (xyz[b, idx, c] - new_xyz[b, m, c]) * inv_radius
// (pointnet2_utils.py:348-352; torch's GPU `x / scalar` is a multiply by the f32 reciprocal),
// 3 <= c < 3+C: features_cl[b, idx, c-3]; columns up to ldx are zero padding.
__global__ __launch_bounds__(256) void sa_gather_kernel(
    int N, int M, int S, int C, int ldx, int use_xyz, float inv_radius,
    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ feats_cl, const int *__restrict__ idx, float *__restrict__ X) {
  const long long rows = (long long)gridDim.y * M * S;  // gridDim.y = B
  const int bi = blockIdx.y;
  const int cw = ldx;  // columns written per row
  const int tpr = min(64, cw);  // threads cooperating on a row (power of two not required)
  const int rows_per_block = 256 / tpr;
  const int lr = threadIdx.x / tpr, lc = threadIdx.x % tpr;
  if (lr >= rows_per_block) return;
  const long long ms = (long long)M * S;
  for (long long jk = (long long)blockIdx.x * rows_per_block + lr; jk < ms;
       jk += (long long)gridDim.x * rows_per_block) {
    const int m = (int)(jk / S);
    const int ii = idx[(size_t)bi * ms + jk];
    float *out = X + ((size_t)bi * ms + jk) * ldx;
    const float *f = feats_cl ? feats_cl + ((size_t)bi * N + ii) * C : nullptr;
    const int xoff = use_xyz ? 3 : 0;
    for (int c = lc; c < cw; c += tpr) {
      float v = 0.f;
      if (c < xoff) {
        v = (xyz[((size_t)bi * N + ii) * 3 + c] - new_xyz[((size_t)bi * M + m) * 3 + c]) *
            inv_radius;
      } else if (c - xoff < C) {
        v = f[c - xoff];
      }
      out[c] = v;
    }
  }
  (void)rows;
}

// First-layer recompute ("RC"): a set-abstraction layer whose input has <= 4 columns (SA1:
// xyz + height) does not store its first pre-BN output y0 = X0 . W0^T (rows x 64 floats, 268 MB
// at the benchmark shape): every consumer rebuilds the 4 values it needs from the 16-byte input
// row and 4 weight rows.  Same expression everywhere, so the ReLU masks agree.
__device__ __forceinline__ float rc_dot4(const float4 x, const float4 w) {
  return fmaf(x.w, w.w, fmaf(x.z, w.z, fmaf(x.y, w.y, x.x * w.x)));
}
__device__ __forceinline__ float4 rc_y4(const float4 x, const float *__restrict__ w0, int k) {
  const float4 *w = reinterpret_cast<const float4 *>(w0) + k;  // W0[k..k+3][0..3]
  return make_float4(rc_dot4(x, w[0]), rc_dot4(x, w[1]), rc_dot4(x, w[2]), rc_dot4(x, w[3]));
}

// ---- live timing of the GEMM family (btr_gemm_trace_begin / _end, include/btr_pointnet2.h) ------
// bench.py's mlp_roofline used to time these launches from Python, which only the Python-sequenced
// form of a layer can do -- a form that lacks what csrc/sa_layer.hip does beyond it (Gram-form
// backward, per-point first layer), so the line's figure drifted from the rocprofv3 one.  While a
// trace is open on the host thread, every entry point of the family records a HIP event pair on
// its own stream around its launches (the outermost scope only: entry points call each other).
// (Process-wide, not per thread: autograd runs the backward's calls on its own thread.  A scope is
// "outermost" per thread.)
// Work accounting (round 6): every entry point also declares the work of ITS launches --
// flops and operand bytes per row and the row count -- so that the roofline figure divides the
// work that was executed by the time it took (the Python-sequenced formulation the line used
// to price has more products and more bytes than the native step: Gram form, per-point first
// layers).  Compact rows: the row count is a device value (Compact::dims[0]); the outermost
// scope copies it to a pinned host slot on its stream BEFORE its start event, _end reads it.
struct GemmWork {
  double flops_per_row, bytes_per_row, bytes_fixed;
  int rows;   // the host's count (dense bound for compact rows)
  int slot;   // pinned slot holding the device row count, -1: rows is exact
};
struct GemmTraceState {
  std::atomic<bool> on{false};
  std::mutex mu;
  std::vector<hipEvent_t> ev;   // pairs
  std::vector<GemmWork> work;
  int *pinned = nullptr;        // kGemmTraceSlots ints, hipHostMalloc
  int nslot = 0;
  double flops = 0.0, bytes = 0.0, dense_flops = 0.0;   // totals of the last closed trace
};
constexpr int kGemmTraceSlots = 8192;
inline GemmTraceState &gemm_trace_state() {
  static GemmTraceState st;
  return st;
}
const int *trace_compact_dims();   // Compact::dims bound on this host thread (or nullptr)
inline int &gemm_trace_slot() {
  static thread_local int slot = -1;   // the outermost open scope's pinned slot
  return slot;
}
bool gemm_trace_active() { return gemm_trace_state().on.load(std::memory_order_relaxed); }
struct GemmTrace {
  hipStream_t s;
  hipEvent_t stop;
  explicit GemmTrace(hipStream_t stream) : s(stream), stop(nullptr) {
    GemmTraceState &st = gemm_trace_state();
    if (!st.on.load(std::memory_order_relaxed)) return;
    static thread_local int depth = 0;
    depth_ = &depth;
    if (depth++ == 0) {
      gemm_trace_slot() = -1;
      if (const int *dims = trace_compact_dims()) {
        int slot = -1;
        {
          std::lock_guard<std::mutex> lock(st.mu);
          if (st.pinned && st.nslot < kGemmTraceSlots) slot = st.nslot++;
        }
        if (slot >= 0 &&
            hipMemcpyAsync(st.pinned + slot, dims, sizeof(int), hipMemcpyDeviceToHost, s) ==
                hipSuccess)
          gemm_trace_slot() = slot;
      }
      hipEvent_t a, b;
      if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) {
        {
          std::lock_guard<std::mutex> lock(st.mu);
          st.ev.push_back(a);
          st.ev.push_back(b);
        }
        (void)hipEventRecord(a, s);
        stop = b;
      }
    }
  }
  ~GemmTrace() {
    if (!depth_) return;
    --*depth_;
    if (stop) (void)hipEventRecord(stop, s);
  }
  // the work of this entry point's own launches (entry points that only sequence others add
  // nothing); compact: the launches take their row count from the bound compact description
  void work(int rows, double flops_per_row, double bytes_per_row, double bytes_fixed,
            bool compact) const {
    if (!depth_) return;
    GemmTraceState &st = gemm_trace_state();
    const int slot = (compact && trace_compact_dims()) ? gemm_trace_slot() : -1;
    std::lock_guard<std::mutex> lock(st.mu);
    st.work.push_back(GemmWork{flops_per_row, bytes_per_row, bytes_fixed, rows, slot});
  }
  int *depth_ = nullptr;
};

// ---- the BatchNorm ticket (BnFin): which workgroup of a column block finished last
// __threadfence() is `buffer_wbl2 sc1` + `buffer_inv sc1` on gfx950: EVERY wave of EVERY workgroup
// walks its XCD's L2 to write the just-stored C tile back before the ticket may be taken, and the
// statistics GEMMs of the 2 048-row chains ran 22 - 29 us where the same kernel without the ticket
// takes 8 (r05 timeline).  The only data another workgroup reads are the `part` rows: they are
// stored write-through (agent-scope stores, `sc1`) and read with agent-scope loads, so the order
// partial sums -> ticket needs the stores' completion (s_waitcnt vmcnt(0): gfx9 counts stores) and
// the workgroup barrier, not a cache walk.  C, scale, shift, ... reach their readers through the
// kernel boundary as always.  fin.fence == 1 keeps the full fences beside it (same results).
__device__ __forceinline__ void store_agent(float *p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float load_agent(const float *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// true in every thread of the workgroup that took the last ticket of its column block
__device__ __forceinline__ bool bn_ticket_last(const BnFin &fin, int *s_last) {
  // The fence-free order below is outside the HIP memory model: it holds where stores are counted
  // by vmcnt and sc1 stores write through (gfx9: validated on gfx942 / gfx950).  Any other target
  // gets the full fences whatever fin.fence says.
#if defined(__gfx942__) || defined(__gfx950__)
  if (fin.fence) __threadfence();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
  __threadfence();
#endif
  __syncthreads();
  if (threadIdx.x == 0)
    *s_last = __hip_atomic_fetch_add(&fin.ticket[blockIdx.y], 1u, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  if (fin.fence && *s_last) __threadfence();
  return *s_last != 0;
}

// The finalisation itself, by the 256 threads of that workgroup: column block [n_blk, n_blk + BN),
// `fr` = [2][8][BN] doubles of LDS.  reduce_partials' order: 64 slices of the workgroup axis, 8 groups
// of 8, the 8 groups; thread (col, q) adds the groups q, q + G, ... on its own, thread (col, 0) the 8
// groups.  With <= 64 row tiles (every launch the ticket accepts by default) a slice holds one
// partial: all of a thread's loads are issued before the first add -- one trip to memory, where the
// loop below makes eight dependent ones per group (~1 us each: they are write-through lines).
template <int BN>
__device__ __forceinline__ void bn_ticket_finalize(const BnFin &fin, const float *part, int N,
                                                   int n_blk, double *fr) {
  constexpr int G = 256 / BN;   // threads per column
  const int tid = threadIdx.x;
  const int col = tid % BN, q = tid / BN, n = n_blk + col;
  const int nblk = gridDim.x;
  if (nblk <= 64) {
    float v1[8 / G][8], v2[8 / G][8];
#pragma unroll
    for (int gi = 0; gi < 8 / G; ++gi)
#pragma unroll
      for (int y = 0; y < 8; ++y) {
        const int b = (q + gi * G) * 8 + y;
        const bool on = n < N && b < nblk;
        v1[gi][y] = on ? load_agent(&part[((size_t)b * 2 + 0) * N + n]) : 0.f;
        v2[gi][y] = on ? load_agent(&part[((size_t)b * 2 + 1) * N + n]) : 0.f;
      }
#pragma unroll
    for (int gi = 0; gi < 8 / G; ++gi) {
      double a1 = 0.0, a2 = 0.0;
#pragma unroll
      for (int y = 0; y < 8; ++y) {
        a1 += (double)v1[gi][y];
        a2 += (double)v2[gi][y];
      }
      fr[(0 * 8 + q + gi * G) * BN + col] = a1;
      fr[(1 * 8 + q + gi * G) * BN + col] = a2;
    }
  } else {
    for (int g = q; g < 8; g += G) {
      double a1 = 0.0, a2 = 0.0;
      if (n < N)
        for (int y = 0; y < 8; ++y) {
          double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
          for (int b = g * 8 + y; b < nblk; b += 64) {
            s1 += (double)load_agent(&part[((size_t)b * 2 + 0) * N + n]);
            s2 += (double)load_agent(&part[((size_t)b * 2 + 1) * N + n]);
          }
          a1 += s1;
          a2 += s2;
        }
      fr[(0 * 8 + g) * BN + col] = a1;
      fr[(1 * 8 + g) * BN + col] = a2;
    }
  }
  __syncthreads();
  if (q == 0 && n < N) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      s1 += fr[(0 * 8 + g) * BN + col];
      s2 += fr[(1 * 8 + g) * BN + col];
    }
    const double mean = s1 / fin.count;
    double var = s2 / fin.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)fin.eps));
    const float a = fin.gamma[n] * invstd;
    fin.scale[n] = a;
    fin.shift[n] = fin.beta[n] - (float)mean * a;
    fin.mean[n] = (float)mean;
    fin.invstd[n] = invstd;
    if (fin.running_mean) {
      const double unbiased = fin.count > 1.0 ? var * fin.count / (fin.count - 1.0) : var;
      float rm = (1.f - fin.momentum) * fin.running_mean[n] + fin.momentum * (float)mean;
      if (fin.rbias && n < fin.nbias) rm += fin.momentum * fin.rbias[n];
      fin.running_mean[n] = rm;
      fin.running_var[n] = (1.f - fin.momentum) * fin.running_var[n] +
                           fin.momentum * (float)unbiased;
    }
  }
}

// ------------------------------------------------------------------------ NT GEMM (MFMA f32)
// C[r][n] = sum_k f(A[r][k]) * W[n][k],  f(y) = PRO ? max(pa[k]*y + pb[k], 0) : y.
// Workgroup = 4 waves, tile 128 rows x BN columns, K staged BK=32 at a time.  Lanes 0-31 of
// a wave feed the first half of each staged K chunk to the MFMA, lanes 32-63 the second half
// (the MFMA's two k slots), so every lane reads 4 consecutive k with one ds_read_b128.
// STATS: per-column sum and sum of squares of C over the rows this workgroup processed are
// written to part[blockIdx.x][0..1][n] (grid-stride over row tiles, reduced later in f64).
// PRO == 2 ("pooled gradient"): A is the pooled layer's pre-BN output Y and the operand is the
// gradient w.r.t. it, formed on the fly (the dense dY never exists in HBM):
//   dY[r][k] = pa[k]*Y[r][k] + pb[k] + (r % S == parg[r/S][k] ? pdcl[r/S][k] : 0)
// (pa = alpha, pb = beta, pdcl = scale*dOut where the pooled output is > 0; see
// btr_sa_pool_bwd_coef).
// PS > 0 ("pooling epilogue", BN == 128 only: a wave owns 64 consecutive rows): per group of PS
// rows and column the epilogue also emits the extremum of C that the max-pool will select and
// the row where it occurs first -- gext [R/PS][N] f32, aext [R/PS][N] u8.  BatchNorm + ReLU
// are monotone per channel, non-decreasing for scale > 0 and non-increasing for scale < 0, and
// sign(scale) = sign(gamma) (`gsign` = the layer's gamma) is known before the statistics are:
// the maximum is tracked where gamma >= 0, the minimum elsewhere, and the max-pool of the
// activated layer is relu(scale * gext + shift).  The pool kernel's pass over the whole pre-BN
// tensor (537 MB for SA1) becomes a pass over 1/PS of it (sa_pool_fin_kernel).
// MM == 1: the products run as bf16x6 (see split4); LDS then holds three bf16 planes per operand
// and the C tile leaves through a per-wave LDS transpose as 16-byte row stores (with the matrix
// pipe 2.67x faster the 4-byte-per-lane stores of the accumulator layout were the bound).
// (Few-workgroup launches -- the 1024-row decoder / head products, 48 - 256 workgroups, 14 us for a
// 1024 x 288 x 288 product whose arithmetic is 3 us: a variant with three k chunks in flight
// (three register sets, chunk loop unrolled by three) measured the same 16.4 us as this one.
// With one workgroup per CU there is one wave per SIMD and its split / LDS / MFMA phases simply
// run one after the other, ~1.3 us per chunk; global-load latency is not the bound.)
template <int BN, int PRO, bool STATS, int PS = 0, int BM = kBM, bool BIAS = false, int MM = 0>
// (second launch bound: at least 2 waves per SIMD, i.e. <= 256 VGPRs -- two workgroups per
// CU; without it the PRO == 2 / BN = 128 variant allocates 292 and runs alone on its CU)
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(
    const float *__restrict__ A, int lda, const float *__restrict__ W, int ldw,
    float *__restrict__ C, int ldc, int R, int N, int K, const float *__restrict__ pa,
    const float *__restrict__ pb, float *__restrict__ part,
    const unsigned char *__restrict__ parg, const float *__restrict__ pdcl, int SSH,
    const float *__restrict__ gsign = nullptr, float *__restrict__ gext = nullptr,
    unsigned char *__restrict__ aext = nullptr, Compact cm = Compact{}, BnFin fin = BnFin{}) {
  static_assert(PS == 0 || (BN == 128 && (PS == 8 || PS == 16 || PS == 32 || PS == 64)),
                "pooling epilogue: 128-column tiles, groups of 8 / 16 / 32 / 64 rows");
  if (cm.dims) R = cm.dims[0];  // compact rows: the row count lives on the device
  if (cm.kz) {
    const int k0 = (int)blockIdx.z * cm.kz;
    A += k0;
    W += k0;
    C += (size_t)blockIdx.z * cm.czs;
    K = min(cm.kz, K - k0);
  }
  constexpr int WN = BN / 64;      // waves along N
  constexpr int WM = 4 / WN;       // waves along M
  constexpr int MI = BM / WM / 32;  // 32-row MFMA tiles per wave
  constexpr int NJ = 2;            // 32-col MFMA tiles per wave
  // PRO == 2 with bf16x6: the sparse part is added by the thread that stages the element, BEFORE
  // the split, from a per-chunk table (slot = group / 8-row block of the tile, k): local row of
  // the arg-max (-1: none in this tile) and the value to add
  __shared__ int sLr[(MM && PRO == 2) ? 16 * 32 : 4];
  __shared__ float sDv[(MM && PRO == 2) ? 16 * 32 : 4];
  __shared__ __attribute__((aligned(16))) float As[MM ? 4 : BM * kLd];
  __shared__ __attribute__((aligned(16))) float Bs[MM ? 4 : BN * kLd];
  __shared__ __attribute__((aligned(16))) __bf16 Pl[MM ? 3 * (BM + BN) * kLp : 8];
  static_assert(!MM || 3 * (BM + BN) * kLp * 2 >= 4 * 32 * 68 * 4, "transpose regions");
  __shared__ double red[STATS ? 2 * WM * BN : 1];
  // per-k prologue coefficients (and, PRO == 3, the first layer's weight rows) live in LDS for
  // the whole kernel: read from global memory inside stage() each of them was a cache round
  // trip in front of the LDS writes of every chunk
  __shared__ __attribute__((aligned(16))) float sPa[PRO ? kMaxK : 4], sPb[PRO ? kMaxK : 4];
  __shared__ __attribute__((aligned(16))) float sW0[PRO == 3 ? kMaxK * 4 : 4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;
  const int n_blk = blockIdx.y * BN;

  // BatchNorm statistics: per tile a lane adds its 16 * MI values in f32, across the tiles of
  // this workgroup the per-lane partials accumulate in f64 (a workgroup covers up to 2048
  // rows: in f32 that chain alone put ~1e-6 of relative error into mean / variance)
  float s1[NJ], s2[NJ];
  double d1[NJ], d2[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    s1[j] = s2[j] = 0.f;
    d1[j] = d2[j] = 0.0;
  }

  const int ntiles = (R + BM - 1) / BM;
  const int nkc = (K + kBK - 1) / kBK;
  const int kq = (tid & 7) * 4;  // this thread's 4 consecutive k inside a staged chunk
  const int srow = tid >> 3;     // ... and its row (+32*p)

  // Register-staged software pipeline over the flattened (tile, k-chunk) sequence: the global
  // loads of step q+1 are in flight while the MFMAs of step q run (T14 "issue early, write
  // late"); the BN+ReLU prologue is applied when the registers are written to LDS.
  float4 ra[BM / 32], rb[BN / 32];
  // PRO == 2: the sparse part of dY (one entry per group and channel) is added to the staged
  // tile in LDS by one thread per (group of the tile, k column): 2 registers of prefetch
  // instead of an arg word + a float4 per staged row (which cost a wave of occupancy).
  const int sp_gi = tid >> 5, sp_k = tid & 31;  // S >= 16 and S | 128: at most 8 groups/tile
  unsigned sp_arg = 0;
  float sp_d = 0.f;
  bool sp_on = false;
  // compact rows (cm.bgrp): groups are runs of 8-row blocks, a tile holds 16 blocks -> two
  // block slots per thread; sp_lr = local row of the group's arg-max if it lies in that block
  int sp_g[2] = {0, 0}, sp_ng[2] = {0, 0}, sp_off[2] = {0, 0};
  unsigned sp_a[2] = {0, 0};
  bool sp_ok[2] = {false, false};
  float sp_dv[2] = {0.f, 0.f};
  float rwt[BM / 32];  // PRO == 2: weight of the dense part for the rows this thread stages
  auto fetch = [&](int tile, int kc) {
    const int r0 = tile * BM, kk = kc * kBK + kq;
#pragma unroll
    for (int p = 0; p < BM / 32; ++p) {
      const int row = srow + 32 * p;
      ra[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      rwt[p] = 1.f;
      if (r0 + row < R && kk < K) {
        ra[p] = PRO == 3 ? *reinterpret_cast<const float4 *>(A + (size_t)(r0 + row) * 4)
                         : *reinterpret_cast<const float4 *>(A + (size_t)(r0 + row) * lda + kk);
        if (PRO == 2 && cm.bw && (row & 7) == 0) rwt[p] = cm.bw[(r0 + row) >> 3];
      }
    }
    if (PRO == 2 && cm.bgrp) {
      // group of each block slot: fixed per TILE, loaded one tile ahead (with the last chunk of
      // the previous tile) so that the loads below depend on registers only -- a chain
      // bgrp -> goff / arg inside this prefetch would stall the wave in front of its MFMAs
      const int k1 = kc * kBK + sp_k;
      if (kc == 0 && tile != (int)blockIdx.x) {
        sp_g[0] = sp_ng[0];
        sp_g[1] = sp_ng[1];
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int slot = sp_gi + 8 * q, blk = (r0 >> 3) + slot;
        sp_ok[q] = (blk << 3) < R && k1 < K;
        if (sp_ok[q]) {
          const int g = sp_g[q];
          sp_off[q] = cm.goff[g] - r0;
          sp_a[q] = parg[(size_t)g * K + k1];
          sp_dv[q] = pdcl[(size_t)g * K + k1];
        }
      }
      if (kc + 1 >= nkc) {  // last chunk of this tile: the next tile's groups
        const int nr0 = (tile + (int)gridDim.x) * BM;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int blk = (nr0 >> 3) + sp_gi + 8 * q;
          sp_ng[q] = (blk << 3) < R ? cm.bgrp[blk] : 0;
        }
      }
    } else if (PRO == 2) {
      const int k1 = kc * kBK + sp_k;  // (group size S = 1 << SSH: shifts, no divisions)
      const int g = (r0 >> SSH) + sp_gi;
      sp_on = (sp_gi << SSH) < BM && (g << SSH) < R && k1 < K;
      if (sp_on) {
        sp_arg = parg[(size_t)g * K + k1];
        sp_d = pdcl[(size_t)g * K + k1];
      }
    }
#pragma unroll
    for (int p = 0; p < BN / 32; ++p) {
      const int row = srow + 32 * p;
      rb[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n_blk + row < N && kk < K)
        rb[p] = *reinterpret_cast<const float4 *>(W + (size_t)(n_blk + row) * ldw + kk);
    }
  };
  auto stage = [&](int tile, int kc) {
    const int r0 = tile * BM, kk = kc * kBK + kq;
    float4 fa = make_float4(1.f, 1.f, 1.f, 1.f), fb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PRO && kk < K) {  // K is padded to a multiple of 4 by the caller
      fa = *reinterpret_cast<const float4 *>(&sPa[kk]);
      fb = *reinterpret_cast<const float4 *>(&sPb[kk]);
    }
#pragma unroll
    for (int p = 0; p < BM / 32; ++p) {
      const int row = srow + 32 * p;
      float4 v = ra[p];
      if (PRO == 1 && r0 + row < R && kk < K) {  // padded rows / columns must stay exactly 0
        v.x = fmaxf(fmaf(fa.x, v.x, fb.x), 0.f);
        v.y = fmaxf(fmaf(fa.y, v.y, fb.y), 0.f);
        v.z = fmaxf(fmaf(fa.z, v.z, fb.z), 0.f);
        v.w = fmaxf(fmaf(fa.w, v.w, fb.w), 0.f);
      }
      if (PRO == 3 && r0 + row < R && kk < K) {  // y0 rebuilt from the 4-column input row
        const float4 y = rc_y4(v, sW0, kk);       // (pdcl carries W0 [K][4] in this mode)
        v.x = fmaxf(fmaf(fa.x, y.x, fb.x), 0.f);
        v.y = fmaxf(fmaf(fa.y, y.y, fb.y), 0.f);
        v.z = fmaxf(fmaf(fa.z, y.z, fb.z), 0.f);
        v.w = fmaxf(fmaf(fa.w, y.w, fb.w), 0.f);
      }
      if (PRO == 2 && r0 + row < R && kk < K) {  // dense part: w * (alpha*y + beta)
        const float wr = rwt[p];                  // (w = 1 except a compact group's first row)
        v.x = wr * fmaf(fa.x, v.x, fb.x);
        v.y = wr * fmaf(fa.y, v.y, fb.y);
        v.z = wr * fmaf(fa.z, v.z, fb.z);
        v.w = wr * fmaf(fa.w, v.w, fb.w);
        if constexpr (MM != 0) {   // + the sparse part (tables written right before this stage)
          const int slot = cm.bgrp ? (row >> 3) : (row >> SSH);
          const int4 lr = *reinterpret_cast<const int4 *>(&sLr[slot * 32 + kq]);
          const float4 dv = *reinterpret_cast<const float4 *>(&sDv[slot * 32 + kq]);
          v.x += lr.x == row ? dv.x : 0.f;
          v.y += lr.y == row ? dv.y : 0.f;
          v.z += lr.z == row ? dv.z : 0.f;
          v.w += lr.w == row ? dv.w : 0.f;
        }
      }
      if constexpr (MM) {
        const Split4 sp = split4(v);
        *reinterpret_cast<bf16x4 *>(&Pl[(0 * BM + row) * kLp + kq]) = sp.h;
        *reinterpret_cast<bf16x4 *>(&Pl[(1 * BM + row) * kLp + kq]) = sp.m;
        *reinterpret_cast<bf16x4 *>(&Pl[(2 * BM + row) * kLp + kq]) = sp.l;
      } else {
        *reinterpret_cast<float4 *>(&As[row * kLd + kq]) = v;
      }
    }
#pragma unroll
    for (int p = 0; p < BN / 32; ++p) {
      if constexpr (MM) {
        const Split4 sp = split4(rb[p]);
        const int row = srow + 32 * p;
        *reinterpret_cast<bf16x4 *>(&Pl[(3 * BM + 0 * BN + row) * kLp + kq]) = sp.h;
        *reinterpret_cast<bf16x4 *>(&Pl[(3 * BM + 1 * BN + row) * kLp + kq]) = sp.m;
        *reinterpret_cast<bf16x4 *>(&Pl[(3 * BM + 2 * BN + row) * kLp + kq]) = sp.l;
      } else {
        *reinterpret_cast<float4 *>(&Bs[(srow + 32 * p) * kLd + kq]) = rb[p];
      }
    }
  };

  if (PRO == 2 && cm.bgrp) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int blk = ((int)blockIdx.x * BM >> 3) + sp_gi + 8 * q;
      sp_g[q] = (blk << 3) < R ? cm.bgrp[blk] : 0;
    }
  }
  if (PRO) {
    for (int i = tid; i < K; i += 256) {
      sPa[i] = pa[i];
      sPb[i] = pb[i];
    }
    if (PRO == 3)
      for (int i = tid; i < K * 4; i += 256) sW0[i] = pdcl[i];
    __syncthreads();
  }
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x, 0);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int r0 = tile * BM;
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    float bwx[MI][4];  // STATS on compact rows: (weight - 1) of the wave's 8-row blocks,
    if (STATS && cm.bw) {  // fetched here so that the epilogue does not wait for them
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = r0 + wm * (BM / WM) + i * 32 + 8 * q;
          bwx[i][q] = row < R ? cm.bw[row >> 3] - 1.f : 0.f;
        }
    }

    for (int kc = 0; kc < nkc; ++kc) {
      if constexpr (PRO == 2 && MM != 0) {   // the chunk's sparse table (see sLr)
        if (cm.bgrp) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int slot = sp_gi + 8 * q;
            const int lr = sp_off[q] + (int)sp_a[q];
            sLr[slot * 32 + sp_k] = (sp_ok[q] && lr >= 0 && (lr >> 3) == slot) ? lr : -1;
            sDv[slot * 32 + sp_k] = sp_dv[q];
          }
        } else {
          sLr[sp_gi * 32 + sp_k] = sp_on ? (sp_gi << SSH) + (int)sp_arg : -1;
          sDv[sp_gi * 32 + sp_k] = sp_d;
          sLr[(sp_gi + 8) * 32 + sp_k] = -1;
        }
        __syncthreads();
      }
      stage(tile, kc);
      __syncthreads();
      if (PRO == 2 && MM == 0) {  // sparse part: the arg-max row of every (group, channel) of this tile
        if (cm.bgrp) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int lr = sp_off[q] + (int)sp_a[q];  // local row of the group's arg-max
            if (sp_ok[q] && lr >= 0 && (lr >> 3) == sp_gi + 8 * q) As[lr * kLd + sp_k] += sp_dv[q];
          }
        } else if (sp_on) {
          As[((sp_gi << SSH) + (int)sp_arg) * kLd + sp_k] += sp_d;
        }
        __syncthreads();
      }
      // issue the next step's global loads before the MFMAs of this one
      if (kc + 1 < nkc)
        fetch(tile, kc + 1);
      else if (tile + (int)gridDim.x < ntiles)
        fetch(tile + gridDim.x, 0);
      // ---- 16 MFMA k-steps on the staged chunk (group t4 holds k = t4..t4+3 in lanes 0-31
      // and 16+t4.. in lanes 32-63: a chunk with <= t4 valid columns has nothing left -- the
      // K = 4 first layer runs one group instead of four)
      const int kleft = K - kc * kBK;
      if constexpr (MM) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          if (ks * 16 >= kleft) break;
          bf16x8 af[3][MI], bf[3][NJ];
#pragma unroll
          for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
              af[q][i] = *reinterpret_cast<const bf16x8 *>(
                  &Pl[(q * BM + wm * (BM / WM) + i * 32 + l31) * kLp + ks * 16 + h * 8]);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              bf[q][j] = *reinterpret_cast<const bf16x8 *>(
                  &Pl[(3 * BM + q * BN + wn * 64 + j * 32 + l31) * kLp + ks * 16 + h * 8]);
          }
          // smallest terms first: (l,h) (h,l) (m,m) | (m,h) (h,m) | (h,h)
#define BTR_X6(QA, QB)                                                                          \
  _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[QA][i], bf[QB][j], acc[i][j], 0, 0, 0);
          BTR_X6(2, 0)
          BTR_X6(0, 2)
          BTR_X6(1, 1)
          BTR_X6(1, 0)
          BTR_X6(0, 1)
          BTR_X6(0, 0)
#undef BTR_X6
        }
      } else {
#pragma unroll
      for (int t4 = 0; t4 < kBK / 2; t4 += 4) {
        if (t4 >= kleft) break;
        float4 af[MI], bf[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
          af[i] = *reinterpret_cast<const float4 *>(
              &As[(wm * (BM / WM) + i * 32 + l31) * kLd + h * (kBK / 2) + t4]);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          bf[j] = *reinterpret_cast<const float4 *>(
              &Bs[(wn * 64 + j * 32 + l31) * kLd + h * (kBK / 2) + t4]);
        // k component outermost: consecutive MFMAs go to DIFFERENT accumulators
#define BTR_MFMA_STEP(C)                                                                        \
  _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].C, bf[j].C, acc[i][j], 0, 0, 0);
        BTR_MFMA_STEP(x)
        BTR_MFMA_STEP(y)
        BTR_MFMA_STEP(z)
        BTR_MFMA_STEP(w)
#undef BTR_MFMA_STEP
      }
      }
      __syncthreads();
    }
    // ---- epilogue: D layout col = lane&31, row = (v&3) + 8*(v>>2) + 4*(lane>>5)
    constexpr bool kWide = MM != 0;   // C through the per-wave LDS transpose, 16-byte stores
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int col = n_blk + wn * 64 + j * 32 + l31;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int row = r0 + wm * (BM / WM) + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
          float c = acc[i][j][v];
          if (BIAS && col < N && row < R) c += gsign[col];  // (BIAS: gsign carries the bias row)
          if (BIAS && kWide) acc[i][j][v] = c;
          if (!kWide && C != nullptr && row < R && col < N) C[(size_t)row * ldc + col] = c;
          if (STATS) {  // rows >= R hold exact zeros (A staged as 0): no masking needed
            s1[j] += c;
            s2[j] = fmaf(c, c, s2[j]);
            // compact rows: the first row of a group stands for 1 + S - len copies of itself
            // (bwx: block weights - 1, loaded at the top of the tile)
            if (cm.bw && h == 0 && (v & 3) == 0) {
              const float wx = bwx[i][v >> 2];
              s1[j] = fmaf(wx, c, s1[j]);
              s2[j] = fmaf(wx * c, c, s2[j]);
            }
          }
        }
      }
      if (kWide && C != nullptr) {
        // the wave's 32 x 64 block: written to its own LDS region in the accumulator layout
        // (lanes along the columns), read back as rows: each store instruction puts 4 rows x
        // 256 B.  The regions overlay the staging planes: every wave is past its last MFMA
        // (the barrier that ended the chunk loop), and the barrier below ends the overlay.
        float *T = reinterpret_cast<float *>(Pl) + wave * (32 * 68);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int v = 0; v < 16; ++v)
            T[((v & 3) + 8 * (v >> 2) + 4 * h) * 68 + j * 32 + l31] = acc[i][j][v];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the region is private to the wave)
        __builtin_amdgcn_wave_barrier();
        const int rl = lane >> 4, c4 = (lane & 15) * 4;
        const int colw = n_blk + wn * 64 + c4;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const float4 q4 = *reinterpret_cast<const float4 *>(&T[(it * 4 + rl) * 68 + c4]);
          const int row = r0 + wm * (BM / WM) + i * 32 + it * 4 + rl;
          if (row < R && colw < N) *reinterpret_cast<float4 *>(C + (size_t)row * ldc + colw) = q4;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (kWide && C != nullptr) __syncthreads();
    if (STATS) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        d1[j] += (double)s1[j];
        d2[j] += (double)s2[j];
        s1[j] = s2[j] = 0.f;
      }
    }
    if constexpr (PS > 0) {
      // this wave's 64 rows = 64 / PS whole groups; lanes l and l ^ 32 hold the same column
      // (rows interleaved in blocks of 4), v ascending = rows ascending within a lane
      constexpr int G = 64 / PS;
      const int wrow0 = r0 + wm * (BM / WM);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int col = n_blk + wn * 64 + j * 32 + l31;
        const float sg = (col < N && gsign[col] < 0.f) ? -1.f : 1.f;
        float vmx[G];
        int imx[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          vmx[g] = -3.0e38f;
          imx[g] = 0;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int rw = i * 32 + (v & 3) + 8 * (v >> 2);   // + 4 * h (same group)
            const int g = rw / PS;
            const float c = acc[i][j][v] * sg;
            if (c > vmx[g]) {
              vmx[g] = c;
              imx[g] = rw - g * PS;
            }
          }
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int mine = imx[g] + 4 * h;
          const float omx = __shfl_xor(vmx[g], 32);
          const int oix = __shfl_xor(mine, 32);
          const bool take = omx > vmx[g] || (omx == vmx[g] && oix < mine);
          const float best = take ? omx : vmx[g];
          const int bidx = take ? oix : mine;
          const int grow = wrow0 + g * PS;
          if (h == 0 && grow < R && col < N) {
            const size_t o = (size_t)(grow / PS) * N + col;
            gext[o] = best * sg;
            aext[o] = (unsigned char)bidx;
          }
        }
      }
    }
  }
  if (STATS) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      d1[j] += __shfl_xor(d1[j], 32);
      d2[j] += __shfl_xor(d2[j], 32);
      if (h == 0) {
        red[(0 * WM + wm) * BN + wn * 64 + j * 32 + l31] = d1[j];
        red[(1 * WM + wm) * BN + wn * 64 + j * 32 + l31] = d2[j];
      }
    }
    __syncthreads();
    for (int c = tid; c < 2 * BN; c += 256) {
      const int which = c / BN, col = c % BN;
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < WM; ++w) s += red[(which * WM + w) * BN + col];
      if (n_blk + col < N)
        store_agent(&part[((size_t)blockIdx.x * 2 + which) * N + n_blk + col], (float)s);
    }
    if constexpr (MM != 0) {
      if (fin.ticket) {
        // ---- the last workgroup of this column block finalises the BatchNorm (see BnFin)
        __shared__ int s_last;
        if (!bn_ticket_last(fin, &s_last)) return;
        double *fr = reinterpret_cast<double *>(Pl);   // [2][8][BN], over the staging planes
        static_assert(3 * (BM + BN) * kLp * 2 >= 2 * 8 * BN * 8, "finalize scratch");
        bn_ticket_finalize<BN>(fin, part, N, n_blk, fr);
      }
    }
  }
}

// ---- small-M NT GEMM: 32-row tiles, weight fragments straight from pre-split planes -------------
// The point-wise chains (feature propagation, vote generator, proposal head: 2 048 - 16 384 rows x
// 128 - 512 columns) and GroupFree3D's 1 024-row layers ran on gemm_nt_kernel's 64 x 128 tiles:
// 48 - 256 workgroups that each walk k in 32-wide chunks, ~1.3 us per chunk whatever it holds
// (stage -> barrier -> MFMA -> barrier with one wave per SIMD): 14 - 17 us for 3 us of
// arithmetic.  Here
//   * a workgroup takes a 32 x 64 tile (>= 4x as many workgroups), and up to 288 columns of k in ONE
//     staged chunk: two barriers per chunk, not per 32 columns;
//   * the weights are split into their three bf16 planes ONCE per training step (by the weight
//     preparation kernel, [3][n][ceil16(k)] next to W2 / W^T) instead of by every row tile: a wave
//     loads its B fragments for the whole chunk straight from L2 into registers, before the A rows
//     are staged, so every global load of the tile is in flight at once;
//   * the four waves are (column tile 0 / 1) x (k half 0 / 1); the halves meet in LDS.
// Same products (bf16x6, smallest terms first per 16-wide step) as gemm_nt_kernel; the k halves
// are added in another order, so the results differ from it in the last bits.  Statistics
// partials: one `part` row per 32-row tile, which is what btr_pm_gemm_grid() counts for these row
// counts; BnFin (the last workgroup of a column block finalises the BatchNorm) as there.
struct SmArgs {
  const float *A;
  int lda;
  const __bf16 *Wp;    // planes [3][N][kp], kp = ceil16(K); rows n < N, columns k >= K are zero
  int kp;              // (a sub-block of a wider matrix: kp = its row pitch, ps = its plane stride)
  long long ps;        // elements between two planes
  float *C;
  int ldc;
  int R, N, K;
  const float *pa, *pb;
  float *part;
  const float *bias;
};
constexpr int kSmKC = 288;   // columns of k per staged chunk
constexpr int kSmLX = kSmKC + 8;

template <int PRO, bool STATS, bool BIAS>
__global__ __launch_bounds__(256, 2) void gemm_nt_sm_kernel(SmArgs a, BnFin fin) {
  constexpr int BM = 32, KC = kSmKC, LX = kSmLX, KS = KC / 32;   // KS: 16-wide steps per k half
  __shared__ __attribute__((aligned(16))) __bf16 Xp[3 * BM * LX];
  __shared__ __attribute__((aligned(16))) float Cs[2 * 32 * 36];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = wave & 1, kh = wave >> 1;
  const int l31 = lane & 31, h = lane >> 5;
  const int r0 = blockIdx.x * BM, n_blk = blockIdx.y * 64;
  const int R = a.R, N = a.N, K = a.K;
  const int ncol = n_blk + t * 32 + l31;   // this lane's weight row / output column
  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  const int srow = tid >> 6;               // staging: rows srow + 4 p, columns skq + 256 j
  const int skq = (tid & 63) * 4;
  for (int k0 = 0; k0 < K; k0 += KC) {
    const int kc = min(KC, K - k0);
    // ---- B fragments of this wave's k half: [ks][plane], all in flight before anything waits
    bf16x8 bfr[KS][3];
    const int kb0 = k0 + kh * (KC / 2) + h * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int kk = kb0 + ks * 16;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bfr[ks][q][e] = (__bf16)0.f;
        if (ncol < N && kk < a.kp && kh * (KC / 2) + ks * 16 < kc)
          bfr[ks][q] = *reinterpret_cast<const bf16x8 *>(
              a.Wp + (size_t)q * a.ps + (size_t)ncol * a.kp + kk);
      }
    }
    // ---- A rows of the chunk: registers -> prologue -> three bf16 planes in LDS
    float4 ra[8][2];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = r0 + srow + 4 * p, kk = skq + 256 * j;
        ra[p][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < R && kk < kc && (j == 0 || skq < KC - 256))
          ra[p][j] = *reinterpret_cast<const float4 *>(a.A + (size_t)row * a.lda + k0 + kk);
      }
    if (k0 > 0) __syncthreads();   // (the previous chunk's fragment reads are done)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kk = skq + 256 * j;
      if (j == 1 && skq >= KC - 256) break;
      float4 fa = make_float4(1.f, 1.f, 1.f, 1.f), fb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (PRO && kk < kc) {
        fa = *reinterpret_cast<const float4 *>(a.pa + k0 + kk);
        fb = *reinterpret_cast<const float4 *>(a.pb + k0 + kk);
      }
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int row = srow + 4 * p;
        float4 v = ra[p][j];
        if (PRO && r0 + row < R && kk < kc) {   // padded rows / columns stay exactly 0
          v.x = fmaxf(fmaf(fa.x, v.x, fb.x), 0.f);
          v.y = fmaxf(fmaf(fa.y, v.y, fb.y), 0.f);
          v.z = fmaxf(fmaf(fa.z, v.z, fb.z), 0.f);
          v.w = fmaxf(fmaf(fa.w, v.w, fb.w), 0.f);
        }
        const Split4 sp = split4(v);
        *reinterpret_cast<bf16x4 *>(&Xp[(0 * BM + row) * LX + kk]) = sp.h;
        *reinterpret_cast<bf16x4 *>(&Xp[(1 * BM + row) * LX + kk]) = sp.m;
        *reinterpret_cast<bf16x4 *>(&Xp[(2 * BM + row) * LX + kk]) = sp.l;
      }
    }
    __syncthreads();
    // ---- this wave's k half of the chunk
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (kh * (KC / 2) + ks * 16 >= kc) break;
      bf16x8 af[3];
#pragma unroll
      for (int q = 0; q < 3; ++q)
        af[q] = *reinterpret_cast<const bf16x8 *>(
            &Xp[(q * BM + l31) * LX + kh * (KC / 2) + ks * 16 + h * 8]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bfr[ks][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[ks][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[ks][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[ks][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[ks][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[ks][0], acc, 0, 0, 0);
    }
  }
  // ---- the k halves meet: waves 2, 3 hand their tiles to waves 0, 1 (D layout kept)
  float *T = Cs + t * (32 * 36);
  if (kh == 1) {
#pragma unroll
    for (int v = 0; v < 16; ++v) T[((v & 3) + 8 * (v >> 2) + 4 * h) * 36 + l31] = acc[v];
  }
  __syncthreads();
  if (kh == 0) {
    const float bcol = (BIAS && ncol < N) ? a.bias[ncol] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int rl = (v & 3) + 8 * (v >> 2) + 4 * h;
      float c = acc[v] + T[rl * 36 + l31];
      if (BIAS && r0 + rl < R) c += bcol;
      acc[v] = c;
      if (STATS) {   // rows >= R hold exact zeros
        s1 += c;
        s2 = fmaf(c, c, s2);
      }
    }
    if (STATS) {
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (h == 0 && ncol < N) {
        store_agent(&a.part[((size_t)blockIdx.x * 2 + 0) * N + ncol], s1);
        store_agent(&a.part[((size_t)blockIdx.x * 2 + 1) * N + ncol], s2);
      }
    }
    if (a.C != nullptr) {
      // the tile back through LDS in the accumulator layout, out as 16-byte row stores
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();   // (this wave's reads of T above are done)
#pragma unroll
      for (int v = 0; v < 16; ++v) T[((v & 3) + 8 * (v >> 2) + 4 * h) * 36 + l31] = acc[v];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      const int rl = lane >> 3, c4 = (lane & 7) * 4;
      const int colw = n_blk + t * 32 + c4;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const float4 q4 = *reinterpret_cast<const float4 *>(&T[(it * 8 + rl) * 36 + c4]);
        const int row = r0 + it * 8 + rl;
        if (row < R && colw < N)
          *reinterpret_cast<float4 *>(a.C + (size_t)row * a.ldc + colw) = q4;
      }
    }
  }
  if constexpr (STATS) {
    if (fin.ticket) {
      // ---- the last workgroup of this column block finalises the BatchNorm (see gemm_nt_kernel)
      __shared__ int s_last;
      if (!bn_ticket_last(fin, &s_last)) return;
      bn_ticket_finalize<64>(fin, a.part, N, n_blk, reinterpret_cast<double *>(Xp));
    }
  }
}

// planes[q][r][c] (q = 0..2, r < n, c < ceil16(k)): the three bf16 pieces of w[r][c] (zero for
// c >= k) -- the stand-alone form of what the weight preparation kernel writes beside W2 / W^T
__global__ __launch_bounds__(256) void pm_weight_planes_kernel(int n, int k, int kp,
                                                               const float *__restrict__ w, int ldw,
                                                               __bf16 *__restrict__ planes) {
  const int e = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (e >= n * kp) return;
  const int r = e / kp, c = e - r * kp;
  __bf16 ph, pm, pl;
  split1(c < k ? w[(size_t)r * ldw + c] : 0.f, ph, pm, pl);
  planes[((size_t)0 * n + r) * kp + c] = ph;
  planes[((size_t)1 * n + r) * kp + c] = pm;
  planes[((size_t)2 * n + r) * kp + c] = pl;
}

// ------------------------------------------------- parallel reduction of per-workgroup partials
// part[nblk][2][N] -> (s1, s2) for channel c in f64.  A 256-thread block covers 4 channels x
// 64 slices of the nblk axis (these launches are latency-bound: 8 loads per thread instead of
// 32 took each of them from ~10 us to ~4 us); the result is valid in the threads with ty == 0.
constexpr int kRedCh = 4, kRedSl = 64;
__device__ __forceinline__ void reduce_partials(const float *__restrict__ part, int nblk, int N,
                                                int c, int tx, int ty, double &s1, double &s2) {
  __shared__ double red[2][kRedSl][kRedCh + 1];
  __shared__ double red2[2][8][kRedCh + 1];
  s1 = 0.0;
  s2 = 0.0;
  if (c < N)
#pragma unroll 8
    for (int b = ty; b < nblk; b += kRedSl) {
      s1 += (double)part[((size_t)b * 2 + 0) * N + c];
      s2 += (double)part[((size_t)b * 2 + 1) * N + c];
    }
  red[0][ty][tx] = s1;
  red[1][ty][tx] = s2;
  __syncthreads();
  if (ty < 8) {  // fixed-order tree: 8 slices per thread, then 8 threads
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      a += red[0][ty * 8 + y][tx];
      b += red[1][ty * 8 + y][tx];
    }
    red2[0][ty][tx] = a;
    red2[1][ty][tx] = b;
  }
  __syncthreads();
  if (ty == 0) {
    s1 = 0.0;
    s2 = 0.0;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      s1 += red2[0][y][tx];
      s2 += red2[1][y][tx];
    }
  }
}

// --------------------------------------------------------------- BN statistics -> affine form
// part[nblk][2][N] -> mean, biased var; scale a = gamma*invstd, shift b = beta - mean*a; also
// saves mean / invstd for backward and updates the running statistics exactly like
// nn.BatchNorm2d in training mode (unbiased variance, momentum).
__global__ __launch_bounds__(256) void bn_finalize_kernel(
    int N, int nblk, double count, float eps, float momentum, const float *__restrict__ part,
    const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ scale,
    float *__restrict__ shift, float *__restrict__ mean_out, float *__restrict__ invstd_out,
    float *__restrict__ running_mean, float *__restrict__ running_var,
    const float *__restrict__ rbias = nullptr, int nbias = 0) {
  const int tx = threadIdx.x & (kRedCh - 1), ty = threadIdx.x / kRedCh;
  const int n = blockIdx.x * kRedCh + tx;
  double s1, s2;
  reduce_partials(part, nblk, N, n, tx, ty, s1, s2);
  if (ty != 0 || n >= N) return;
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float a = gamma[n] * invstd;
  scale[n] = a;
  shift[n] = beta[n] - (float)mean * a;
  mean_out[n] = (float)mean;
  invstd_out[n] = invstd;
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    float rm = (1.f - momentum) * running_mean[n] + momentum * (float)mean;
    // rbias: the convolution bias skipped in front of this BatchNorm only moves the running mean
    // (was a launch of its own behind the chain: bias_running_mean_kernel)
    if (rbias && n < nbias) rm += momentum * rbias[n];
    running_mean[n] = rm;
    running_var[n] = (1.f - momentum) * running_var[n] + momentum * (float)unbiased;
  }
}

// -------------------------------------------------------------- BN + ReLU + max-pool (fwd)
// out[b][c][m] = max_s relu(a[c]*Y[r][c] + b[c]) (first maximal s, like F.max_pool2d);
// arg[(b*M+m)*C + c] = that s;  out_cl[b][m][c] = same values channel-last (next layer's
// gather source).  One thread per (row-group, channel); lanes along c -> coalesced Y reads.
__global__ __launch_bounds__(256) void sa_pool_kernel(int M, int S, int C, int ldy,
                                                      const float *__restrict__ Y,
                                                      const float *__restrict__ scale,
                                                      const float *__restrict__ shift,
                                                      float *__restrict__ out,
                                                      float *__restrict__ out_cl,
                                                      unsigned char *__restrict__ arg,
                                                      long long groups) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= groups * C) return;
  const long long g = t / C;  // g = b*M + m
  const int c = (int)(t - g * C);
  const float a = scale[c], b = shift[c];
  const float *y = Y + (size_t)g * S * ldy + c;
  float best = -1.f;
  int bs = 0;
  for (int s = 0; s < S; ++s) {
    const float v = fmaxf(fmaf(a, y[(size_t)s * ldy], b), 0.f);
    if (v > best) {
      best = v;
      bs = s;
    }
  }
  const long long bi = g / M;
  const int m = (int)(g - bi * M);
  out[((size_t)bi * C + c) * M + m] = best;
  if (out_cl) out_cl[(size_t)g * C + c] = best;
  arg[(size_t)g * C + c] = (unsigned char)bs;
}

// Max-pool of relu(bn(Y)) from the per-group extremum the pooling epilogue of the last layer's
// GEMM emitted (see gemm_nt_kernel, PS): out / out_cl / arg as sa_pool4_kernel writes them.
// relu(fma(a, y, b)) is monotone in y, so the pooled value is the one of the extremum; its row
// (first occurrence) is the arg-max unless the pooled value is 0, where sa_pool4_kernel keeps
// row 0 and no gradient flows anyway.
__global__ __launch_bounds__(256) void sa_pool_fin_kernel(
    int M, int C, const float *__restrict__ gext, const unsigned char *__restrict__ aext,
    const float *__restrict__ scale, const float *__restrict__ shift, float *__restrict__ out,
    float *__restrict__ out_cl, unsigned char *__restrict__ arg, long long groups,
    float *__restrict__ ywin = nullptr) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= groups * C) return;
  const long long g = t / C;
  const int c = (int)(t - g * C);
  const float a = scale[c], b = shift[c];
  const float v = fmaxf(fmaf(a, gext[t], b), 0.f);
  const long long bi = g / M;
  const int m = (int)(g - bi * M);
  out[((size_t)bi * C + c) * M + m] = v;
  if (out_cl) out_cl[t] = v;
  arg[t] = v > 0.f ? aext[t] : (unsigned char)0;
  if (ywin) ywin[t] = gext[t];   // the pre-BN value behind `out` (only read where out > 0)
}

// Same with four channels per thread (C % 4 == 0, ldy % 4 == 0): 16-byte loads, four
// independent max chains per thread.
__global__ __launch_bounds__(256) void sa_pool4_kernel(int M, int S, int C, int ldy,
                                                       const float *__restrict__ Y,
                                                       const float *__restrict__ scale,
                                                       const float *__restrict__ shift,
                                                       float *__restrict__ out,
                                                       float *__restrict__ out_cl,
                                                       unsigned char *__restrict__ arg,
                                                       long long groups) {
  const int cq = C >> 2;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= groups * cq) return;
  const long long g = t / cq;
  const int c = (int)(t - g * cq) * 4;
  const float4 a = *reinterpret_cast<const float4 *>(scale + c);
  const float4 b = *reinterpret_cast<const float4 *>(shift + c);
  const float *y = Y + (size_t)g * S * ldy + c;
  float best[4] = {-1.f, -1.f, -1.f, -1.f};
  int bs[4] = {0, 0, 0, 0};
#pragma unroll 4
  for (int s = 0; s < S; ++s) {
    const float4 q = *reinterpret_cast<const float4 *>(y + (size_t)s * ldy);
    const float v[4] = {fmaxf(fmaf(a.x, q.x, b.x), 0.f), fmaxf(fmaf(a.y, q.y, b.y), 0.f),
                        fmaxf(fmaf(a.z, q.z, b.z), 0.f), fmaxf(fmaf(a.w, q.w, b.w), 0.f)};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (v[i] > best[i]) {
        best[i] = v[i];
        bs[i] = s;
      }
  }
  const long long bi = g / M;
  const int m = (int)(g - bi * M);
#pragma unroll
  for (int i = 0; i < 4; ++i) out[((size_t)bi * C + c + i) * M + m] = best[i];
  if (out_cl)
    *reinterpret_cast<float4 *>(out_cl + (size_t)g * C + c) =
        make_float4(best[0], best[1], best[2], best[3]);
  *reinterpret_cast<unsigned *>(arg + (size_t)g * C + c) =
      (unsigned)bs[0] | ((unsigned)bs[1] << 8) | ((unsigned)bs[2] << 16) | ((unsigned)bs[3] << 24);
}

// ------------------------------------------- max-pool + ReLU + BN backward, statistics pass
// g = dOut[b][c][m] where out > 0 (ReLU), routed to row s* = arg;  accumulates per channel
// sum(g) and sum(g * xhat) with xhat = (Y[r*][c] - mean)*invstd  -> part[blk][2][C].
__global__ __launch_bounds__(256) void sa_pool_bwd_stats_kernel(
    int M, int S, int C, int ldy, const float *__restrict__ Y, const float *__restrict__ dout,
    const float *__restrict__ out, const unsigned char *__restrict__ arg,
    const float *__restrict__ mean, const float *__restrict__ invstd, long long groups,
    float *__restrict__ part) {
  // block handles all channels (c = threadIdx.x + 256*i) for a strided set of groups
  for (int c = threadIdx.x; c < C; c += 256) {
    const float mu = mean[c], is = invstd[c];
    float s1 = 0.f, s2 = 0.f;
    for (long long g = blockIdx.x; g < groups; g += gridDim.x) {
      const long long bi = g / M;
      const int m = (int)(g - bi * M);
      const size_t o = ((size_t)bi * C + c) * M + m;
      if (out[o] > 0.f) {
        const float gr = dout[o];
        const int s = arg[(size_t)g * C + c];
        const float xh = (Y[((size_t)g * S + s) * ldy + c] - mu) * is;
        s1 += gr;
        s2 = fmaf(gr, xh, s2);
      }
    }
    part[((size_t)blockIdx.x * 2 + 0) * C + c] = s1;
    part[((size_t)blockIdx.x * 2 + 1) * C + c] = s2;
  }
}

// part[nblk][2][C] -> m1 = sum(g)/count, m2 = sum(g*xhat)/count, dgamma = sum(g*xhat),
// dbeta = sum(g).  alpha != NULL (the pooled layer): also the coefficients of its dense gradient
// part, what sa_pool_ab_kernel computes from the stored m1 / m2 (one launch less per level).
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(
    int C, int nblk, double count, const float *__restrict__ part, float *__restrict__ m1,
    float *__restrict__ m2, float *__restrict__ dgamma, float *__restrict__ dbeta,
    const float *__restrict__ scale = nullptr, const float *__restrict__ mean = nullptr,
    const float *__restrict__ invstd = nullptr, float *__restrict__ alpha = nullptr,
    float *__restrict__ beta = nullptr) {
  const int tx = threadIdx.x & (kRedCh - 1), ty = threadIdx.x / kRedCh;
  const int c = blockIdx.x * kRedCh + tx;
  double s1, s2;
  reduce_partials(part, nblk, C, c, tx, ty, s1, s2);
  if (ty != 0 || c >= C) return;
  const float f1 = (float)(s1 / count), f2 = (float)(s2 / count);
  m1[c] = f1;
  m2[c] = f2;
  dgamma[c] = (float)s2;
  dbeta[c] = (float)s1;
  if (alpha) {
    const float a = scale[c];
    const float al = -a * invstd[c] * f2;
    alpha[c] = al;
    beta[c] = -a * f1 - al * mean[c];
  }
}

// Coefficients of the pooled layer's gradient for the GEMM prologues (no dense dY pass):
//   dY[r][c] = alpha[c]*Y[r][c] + beta[c] + (r % S == arg[g][c] ? dcl[g][c] : 0),  g = r / S
// with alpha = -a*invstd*m2, beta = -a*m1 - alpha*mean, dcl = a*dOut where out > 0 (else 0),
// which is a[c]*(g - m1 - xhat*m2) regrouped.
__global__ __launch_bounds__(256) void sa_pool_coef_kernel(
    int M, int C, long long groups, const float *__restrict__ dout,
    const float *__restrict__ out, const float *__restrict__ scale,
    const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ m1, const float *__restrict__ m2, float *__restrict__ dcl,
    float *__restrict__ alpha, float *__restrict__ beta) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= groups * C) return;
  const long long g = t / C;
  const int c = (int)(t - g * C);
  const long long bi = g / M;
  const int m = (int)(g - bi * M);
  const size_t o = ((size_t)bi * C + c) * M + m;
  const float a = scale[c];
  dcl[t] = out[o] > 0.f ? a * dout[o] : 0.f;
  if (g == 0) {
    const float al = -a * invstd[c] * m2[c];
    alpha[c] = al;
    beta[c] = -a * m1[c] - al * mean[c];
  }
}

// Statistics + dcl of the pooled layer in one pass over (dout, out) only: tile = 16 channels x
// 64 centres of one batch element, lanes along the centre axis (coalesced reads of the
// (B, C, M) tensors); xhat at the arg-max row is recovered from the pooled output
// (out = a*y + b, out > 0) instead of gathered from Y (|a| ~ 0: gathered).  part row =
// b * ceil(M/64) + centre tile; dcl is written channel-last through an LDS transpose.
__global__ __launch_bounds__(256) void sa_pool_bwd_tile_kernel(
    int M, int S, int C, int ldy, const float *__restrict__ Y, const float *__restrict__ dout,
    const float *__restrict__ out, const unsigned char *__restrict__ arg,
    const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ scale, const float *__restrict__ shift, float *__restrict__ part,
    float *__restrict__ dcl, const int *__restrict__ goff = nullptr) {
  __shared__ float tile[64][17];
  const int bi = blockIdx.z, c0 = blockIdx.y * 16, m0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = m0 + lane;
  const size_t prow = ((size_t)bi * gridDim.x + blockIdx.x) * 2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cl = wave + 4 * i, c = c0 + cl;
    float g = 0.f, gx = 0.f, dv = 0.f;
    if (c < C && m < M) {
      const size_t o = ((size_t)bi * C + c) * M + m;
      const float ov = out[o];
      if (ov > 0.f) {
        g = dout[o];
        const float a = scale[c];
        float y;
        if (fabsf(a) > 1e-20f) {
          y = (ov - shift[c]) / a;
        } else if (ldy == 0) {   // Y = the arg-max rows' values (btr_sa_pool_fin_y / _sac_pool_y)
          y = Y[((size_t)bi * M + m) * C + c];
        } else {
          const size_t grp = (size_t)bi * M + m;
          const size_t row = goff ? (size_t)goff[grp] + arg[grp * C + c]
                                  : grp * S + arg[grp * C + c];
          y = Y[row * ldy + c];
        }
        gx = g * ((y - mean[c]) * invstd[c]);
        dv = a * g;
      }
    }
    tile[lane][cl] = dv;
    float s1 = g, s2 = gx;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      s1 += __shfl_xor(s1, off);
      s2 += __shfl_xor(s2, off);
    }
    if (lane == 0 && c < C) {
      part[(prow + 0) * C + c] = s1;
      part[(prow + 1) * C + c] = s2;
    }
  }
  __syncthreads();
  const int tm = threadIdx.x >> 2, tc = (threadIdx.x & 3) * 4;  // 64 centres x 4 float4
  if (m0 + tm < M && c0 + tc < C) {
    float *dst = dcl + ((size_t)bi * M + m0 + tm) * C + c0 + tc;
    if (c0 + tc + 3 < C) {
      *reinterpret_cast<float4 *>(dst) =
          make_float4(tile[tm][tc], tile[tm][tc + 1], tile[tm][tc + 2], tile[tm][tc + 3]);
    } else {
      for (int q = 0; q < 4 && c0 + tc + q < C; ++q) dst[q] = tile[tm][tc + q];
    }
  }
}

// alpha / beta of the pooled layer's dense gradient part (see sa_pool_coef_kernel)
__global__ __launch_bounds__(256) void sa_pool_ab_kernel(
    int C, const float *__restrict__ scale, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ m1,
    const float *__restrict__ m2, float *__restrict__ alpha, float *__restrict__ beta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float a = scale[c];
  const float al = -a * invstd[c] * m2[c];
  alpha[c] = al;
  beta[c] = -a * m1[c] - al * mean[c];
}

// dY[r][c] = a[c] * (g[r][c] - m1[c] - xhat[r][c]*m2[c]) written IN PLACE over Y, for the
// pooled (last) layer: g[r][c] = dOut if (s == arg && out > 0) else 0.
__global__ __launch_bounds__(256) void sa_pool_bwd_apply_kernel(
    int M, int S, int C, int ldy, float *__restrict__ Y, const float *__restrict__ dout,
    const float *__restrict__ out, const unsigned char *__restrict__ arg,
    const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ scale, const float *__restrict__ m1, const float *__restrict__ m2,
    long long groups) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= groups * C) return;
  const long long g = t / C;
  const int c = (int)(t - g * C);
  const long long bi = g / M;
  const int m = (int)(g - bi * M);
  const size_t o = ((size_t)bi * C + c) * M + m;
  const float gr = out[o] > 0.f ? dout[o] : 0.f;
  const int sa = arg[(size_t)g * C + c];
  const float mu = mean[c], is = invstd[c], a = scale[c], c1 = m1[c], c2 = m2[c];
  float *y = Y + (size_t)g * S * ldy + c;
  for (int s = 0; s < S; ++s) {
    const float xh = (y[(size_t)s * ldy] - mu) * is;
    const float gg = s == sa ? gr : 0.f;
    y[(size_t)s * ldy] = a * (gg - c1 - xh * c2);
  }
}

// Hidden layers: G holds dX (gradient w.r.t. the post-ReLU activation).  Pass 1 (stats):
// g = G * (a*Y + b > 0);  part <- sum(g), sum(g*xhat).   Pass 2 (apply, in place on G):
// G <- a * (g - m1 - xhat*m2).  Rows x channels elementwise, lanes along c.
constexpr int kBnBwdMaxC = 512;

// Statistics pass: every thread owns 4 consecutive channels (one float4 per row) and walks
// the rows assigned to its row slot; C % 4 == 0, C <= 512, so a row is covered by C/4 <= 128
// lanes and a 256-thread block streams 256/(C/4) rows per iteration with 16-byte loads
// (threads beyond the last whole row slot idle: C = 288 uses 216 of 256).
__global__ __launch_bounds__(256) void bn_relu_bwd_stats_kernel(
    long long R, int C, int ld, const float *__restrict__ G, const float *__restrict__ Y,
    const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ mean, const float *__restrict__ invstd, float *__restrict__ part,
    Compact cm = Compact{}) {
  __shared__ float red[2][256 * 4];
  if (cm.dims) R = cm.dims[0];
  const int tpr = C >> 2;             // threads per row
  const int slots = 256 / tpr;        // rows per block iteration
  const int slot = threadIdx.x / tpr, c4 = (threadIdx.x % tpr) * 4;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (slot < slots) {
    const float4 a = *reinterpret_cast<const float4 *>(scale + c4);
    const float4 b = *reinterpret_cast<const float4 *>(shift + c4);
    const float4 mu = *reinterpret_cast<const float4 *>(mean + c4);
    const float4 is = *reinterpret_cast<const float4 *>(invstd + c4);
    // four rows in flight per thread (a streaming pass: the loads of a row must not wait for
    // the arithmetic of the previous one)
    const long long stride = (long long)gridDim.x * slots;
    long long r = (long long)blockIdx.x * slots + slot;
    auto one = [&](const float4 y, float4 g) {
      g.x = fmaf(a.x, y.x, b.x) > 0.f ? g.x : 0.f;
      g.y = fmaf(a.y, y.y, b.y) > 0.f ? g.y : 0.f;
      g.z = fmaf(a.z, y.z, b.z) > 0.f ? g.z : 0.f;
      g.w = fmaf(a.w, y.w, b.w) > 0.f ? g.w : 0.f;
      s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
      s2.x = fmaf(g.x, (y.x - mu.x) * is.x, s2.x);
      s2.y = fmaf(g.y, (y.y - mu.y) * is.y, s2.y);
      s2.z = fmaf(g.z, (y.z - mu.z) * is.z, s2.z);
      s2.w = fmaf(g.w, (y.w - mu.w) * is.w, s2.w);
    };
    for (; r + 3 * stride < R; r += 4 * stride) {
      float4 y[4], g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        y[u] = *reinterpret_cast<const float4 *>(Y + (size_t)(r + u * stride) * ld + c4);
        g[u] = *reinterpret_cast<const float4 *>(G + (size_t)(r + u * stride) * ld + c4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(y[u], g[u]);
    }
    for (; r < R; r += stride)
      one(*reinterpret_cast<const float4 *>(Y + (size_t)r * ld + c4),
          *reinterpret_cast<const float4 *>(G + (size_t)r * ld + c4));
  }
  *reinterpret_cast<float4 *>(&red[0][threadIdx.x * 4]) = s1;
  *reinterpret_cast<float4 *>(&red[1][threadIdx.x * 4]) = s2;
  __syncthreads();
  for (int q = threadIdx.x; q < 2 * C; q += 256) {
    const int which = q / C, c = q - which * C;
    float acc = 0.f;
    for (int sl = 0; sl < slots; ++sl) acc += red[which][(sl * tpr + (c >> 2)) * 4 + (c & 3)];
    part[((size_t)blockIdx.x * 2 + which) * C + c] = acc;
  }
}

// RC layer 0, statistics: as bn_relu_bwd_stats_kernel with y rebuilt from X0 (ld 4) and W0.
__global__ __launch_bounds__(256) void bn_relu_bwd_stats_rc_kernel(
    long long R, int C, int ld, const float *__restrict__ G, const float *__restrict__ X0,
    const float *__restrict__ W0, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ mean,
    const float *__restrict__ invstd, float *__restrict__ part, Compact cm = Compact{}) {
  __shared__ float red[2][256 * 4];
  if (cm.dims) R = cm.dims[0];
  const int tpr = C >> 2;
  const int slots = 256 / tpr;
  const int slot = threadIdx.x / tpr, c4 = (threadIdx.x % tpr) * 4;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (slot < slots) {
    const float4 a = *reinterpret_cast<const float4 *>(scale + c4);
    const float4 b = *reinterpret_cast<const float4 *>(shift + c4);
    const float4 mu = *reinterpret_cast<const float4 *>(mean + c4);
    const float4 is = *reinterpret_cast<const float4 *>(invstd + c4);
    const long long stride = (long long)gridDim.x * slots;
    long long r = (long long)blockIdx.x * slots + slot;
    auto one = [&](const float4 x, float4 g) {
      const float4 y = rc_y4(x, W0, c4);
      g.x = fmaf(a.x, y.x, b.x) > 0.f ? g.x : 0.f;
      g.y = fmaf(a.y, y.y, b.y) > 0.f ? g.y : 0.f;
      g.z = fmaf(a.z, y.z, b.z) > 0.f ? g.z : 0.f;
      g.w = fmaf(a.w, y.w, b.w) > 0.f ? g.w : 0.f;
      s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
      s2.x = fmaf(g.x, (y.x - mu.x) * is.x, s2.x);
      s2.y = fmaf(g.y, (y.y - mu.y) * is.y, s2.y);
      s2.z = fmaf(g.z, (y.z - mu.z) * is.z, s2.z);
      s2.w = fmaf(g.w, (y.w - mu.w) * is.w, s2.w);
    };
    for (; r + 3 * stride < R; r += 4 * stride) {   // four rows in flight per thread
      float4 x[4], g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        x[u] = *reinterpret_cast<const float4 *>(X0 + (size_t)(r + u * stride) * 4);
        g[u] = *reinterpret_cast<const float4 *>(G + (size_t)(r + u * stride) * ld + c4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(x[u], g[u]);
    }
    for (; r < R; r += stride)
      one(*reinterpret_cast<const float4 *>(X0 + (size_t)r * 4),
          *reinterpret_cast<const float4 *>(G + (size_t)r * ld + c4));
  }
  *reinterpret_cast<float4 *>(&red[0][threadIdx.x * 4]) = s1;
  *reinterpret_cast<float4 *>(&red[1][threadIdx.x * 4]) = s2;
  __syncthreads();
  for (int q = threadIdx.x; q < 2 * C; q += 256) {
    const int which = q / C, c = q - which * C;
    float acc = 0.f;
    for (int sl = 0; sl < slots; ++sl) acc += red[which][(sl * tpr + (c >> 2)) * 4 + (c & 3)];
    part[((size_t)blockIdx.x * 2 + which) * C + c] = acc;
  }
}

// RC layer 0, apply + weight gradient in one pass: dY0 = a*(mask*g - m1 - xhat*m2) is formed per
// row and immediately contracted with the input row, dW0[c][0..3] += dY0[c] * X0[r][0..3]; dY0
// itself is never written (layer 0 has no other consumer when the inputs need no gradient).
// pw[blk][C][4] per-block partials, reduced by reduce_chunks_kernel.
__global__ __launch_bounds__(256) void bn_relu_bwd_wgrad0_rc_kernel(
    long long R, int C, int ld, const float *__restrict__ G, const float *__restrict__ X0,
    const float *__restrict__ W0, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ m1,
    const float *__restrict__ m2, float *__restrict__ pw, Compact cm = Compact{}) {
  __shared__ float red[256 * 16];
  if (cm.dims) R = cm.dims[0];
  const int tpr = C >> 2;
  const int slots = 256 / tpr;
  const int slot = threadIdx.x / tpr, c4 = (threadIdx.x % tpr) * 4;
  float4 w[4];  // w[i] = partial dW0[c4 + i][0..3]
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (slot < slots) {
    const float4 a = *reinterpret_cast<const float4 *>(scale + c4);
    const float4 b = *reinterpret_cast<const float4 *>(shift + c4);
    const float4 mu = *reinterpret_cast<const float4 *>(mean + c4);
    const float4 is = *reinterpret_cast<const float4 *>(invstd + c4);
    const float4 c1 = *reinterpret_cast<const float4 *>(m1 + c4);
    const float4 c2 = *reinterpret_cast<const float4 *>(m2 + c4);
    for (long long r = (long long)blockIdx.x * slots + slot; r < R;
         r += (long long)gridDim.x * slots) {
      const float4 x = *reinterpret_cast<const float4 *>(X0 + (size_t)r * 4);
      const float4 y = rc_y4(x, W0, c4);
      const float4 g = *reinterpret_cast<const float4 *>(G + (size_t)r * ld + c4);
      // compact rows: G is the gradient summed over the copies a row stands for, the
      // statistics term counts once per copy (w copies)
      const float cw = (cm.bw && (r & 7) == 0) ? cm.bw[r >> 3] : 1.f;
      float d[4];
      d[0] = a.x * ((fmaf(a.x, y.x, b.x) > 0.f ? g.x : 0.f) - cw * (c1.x + (y.x - mu.x) * is.x * c2.x));
      d[1] = a.y * ((fmaf(a.y, y.y, b.y) > 0.f ? g.y : 0.f) - cw * (c1.y + (y.y - mu.y) * is.y * c2.y));
      d[2] = a.z * ((fmaf(a.z, y.z, b.z) > 0.f ? g.z : 0.f) - cw * (c1.z + (y.z - mu.z) * is.z * c2.z));
      d[3] = a.w * ((fmaf(a.w, y.w, b.w) > 0.f ? g.w : 0.f) - cw * (c1.w + (y.w - mu.w) * is.w * c2.w));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        w[i].x = fmaf(d[i], x.x, w[i].x);
        w[i].y = fmaf(d[i], x.y, w[i].y);
        w[i].z = fmaf(d[i], x.z, w[i].z);
        w[i].w = fmaf(d[i], x.w, w[i].w);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    *reinterpret_cast<float4 *>(&red[(threadIdx.x * 4 + i) * 4]) = w[i];
  __syncthreads();
  // red[(thread*4 + i)*4 + j]: thread = slot*tpr + c/4, i = c%4 -> sum over slots
  for (int q = threadIdx.x; q < C * 4; q += 256) {
    const int c = q >> 2, j = q & 3;
    float acc = 0.f;
    for (int sl = 0; sl < slots; ++sl) acc += red[((sl * tpr + (c >> 2)) * 4 + (c & 3)) * 4 + j];
    pw[(size_t)blockIdx.x * C * 4 + q] = acc;
  }
}

// In-place apply, float4 per thread (C % 4 == 0).
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(
    long long R, int C, int ld, float *__restrict__ G, const float *__restrict__ Y,
    const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ m1, const float *__restrict__ m2, Compact cm = Compact{}) {
  if (cm.dims) R = cm.dims[0];
  const int cq = C >> 2;
  const long long total = R * (long long)cq;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total;
       t += (long long)gridDim.x * 256) {
    const long long r = t / cq;
    const int c4 = (int)(t - r * cq) * 4;
    const float4 y = *reinterpret_cast<const float4 *>(Y + (size_t)r * ld + c4);
    float4 g = *reinterpret_cast<float4 *>(G + (size_t)r * ld + c4);
    const float4 a = *reinterpret_cast<const float4 *>(scale + c4);
    const float4 b = *reinterpret_cast<const float4 *>(shift + c4);
    const float4 mu = *reinterpret_cast<const float4 *>(mean + c4);
    const float4 is = *reinterpret_cast<const float4 *>(invstd + c4);
    const float4 c1 = *reinterpret_cast<const float4 *>(m1 + c4);
    const float4 c2 = *reinterpret_cast<const float4 *>(m2 + c4);
    const float w = (cm.bw && (r & 7) == 0) ? cm.bw[r >> 3] : 1.f;  // see wgrad0_rc
    g.x = a.x * ((fmaf(a.x, y.x, b.x) > 0.f ? g.x : 0.f) - w * (c1.x + (y.x - mu.x) * is.x * c2.x));
    g.y = a.y * ((fmaf(a.y, y.y, b.y) > 0.f ? g.y : 0.f) - w * (c1.y + (y.y - mu.y) * is.y * c2.y));
    g.z = a.z * ((fmaf(a.z, y.z, b.z) > 0.f ? g.z : 0.f) - w * (c1.z + (y.z - mu.z) * is.z * c2.z));
    g.w = a.w * ((fmaf(a.w, y.w, b.w) > 0.f ? g.w : 0.f) - w * (c1.w + (y.w - mu.w) * is.w * c2.w));
    *reinterpret_cast<float4 *>(G + (size_t)r * ld + c4) = g;
  }
}

// ------------------------------------------------------------------------ TN GEMM (wgrad)
// dW[n][k] = sum_r G[r][n] * f(X[r][k]),  f as in the NT GEMM (the layer's input is the
// previous layer's pre-BN output).  Workgroup tile (32*TNW)(n) x 64(k): wave w < TNW owns the
// 32 n-rows w*32.. and both 32-wide k tiles (TNW = 4), or with TNW = 2 the four waves form a
// 2x2 grid of 32x32 tiles.  The reduction runs over a chunk of rows staged 32 at a time with
// the next rows prefetched into registers; per-chunk partials -> pw[chunk][N][K], reduced by
// reduce_chunks_kernel in a fixed order (deterministic).
// GPOOL: G is the pooled layer's pre-BN output Y and the gradient operand is formed on the fly
// exactly as in gemm_nt_kernel's PRO == 2 (galpha/gbeta per n column, garg/gdcl per group).
// XRC: the X operand is relu(bn(y0)) with y0 rebuilt from the 4-column input rows (X = X0,
// xw0 = W0 [K][4]); implies PRO.
template <int TNW, bool PRO, bool GPOOL = false, bool XRC = false>
__global__ __launch_bounds__(256) void gemm_tn_kernel(
    const float *__restrict__ G, int ldg, const float *__restrict__ X, int ldx, int R, int N,
    int K, const float *__restrict__ pa, const float *__restrict__ pb, int rows_per_chunk,
    float *__restrict__ pw, const unsigned char *__restrict__ garg = nullptr,
    const float *__restrict__ gdcl = nullptr, const float *__restrict__ galpha = nullptr,
    const float *__restrict__ gbeta = nullptr, int SSH = 0,
    const float *__restrict__ xw0 = nullptr, Compact cm = Compact{}) {
  if (cm.dims) R = cm.dims[0];  // compact rows: the row count lives on the device
  constexpr int BR = 32;
  constexpr int TN = 32 * TNW;      // n columns of G staged per step
  constexpr int LG = TN + 4, LX = 68;
  constexpr int KT = TNW == 4 ? 2 : 1;  // 32-wide k tiles per wave
  __shared__ __attribute__((aligned(16))) float Gs[BR * LG];
  __shared__ __attribute__((aligned(16))) float Xs[BR * LX];
  __shared__ __attribute__((aligned(16))) float sXw[XRC ? 64 * 4 : 4];  // W0 rows k0 .. k0+63
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = TNW == 4 ? wave : (wave >> 1);
  const int wk = TNW == 4 ? 0 : (wave & 1);
  const int l31 = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * TN, k0 = blockIdx.y * 64;
  const int chunk = blockIdx.z;
  const int rbeg = chunk * rows_per_chunk;
  const int rend = min(R, rbeg + rows_per_chunk);

  f32x16 acc[KT];
#pragma unroll
  for (int q = 0; q < KT; ++q)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[q][v] = 0.f;

  // X: 16 threads x float4 cover 64 k columns, 16 rows per pass, 2 passes
  const int xc4 = (tid & 15) * 4, xr = tid >> 4;
  // G: TN/4 threads x float4 cover TN n columns, 256/(TN/4) rows per pass
  constexpr int GT = TN / 4, GR = 256 / GT, GPASS = BR / GR;
  const int gc4 = (tid % GT) * 4, gr = tid / GT;
  float4 fa = make_float4(1.f, 1.f, 1.f, 1.f), fb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (PRO && k0 + xc4 < K) {
    fa = *reinterpret_cast<const float4 *>(pa + k0 + xc4);
    fb = *reinterpret_cast<const float4 *>(pb + k0 + xc4);
  }
  float4 ga = make_float4(0.f, 0.f, 0.f, 0.f), gb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (GPOOL && n0 + gc4 < N) {
    ga = *reinterpret_cast<const float4 *>(galpha + n0 + gc4);
    gb = *reinterpret_cast<const float4 *>(gbeta + n0 + gc4);
  }
  float4 rg[GPASS], rx[2];
  // GPOOL: sparse part added in LDS by one thread per (group of the 32-row step, n column)
  const int sp_gi = tid / TN, sp_n = tid % TN;
  unsigned sp_arg = 0;
  float sp_d = 0.f;
  bool sp_on = false;
  // compact rows: a 32-row step holds 4 blocks of 8 rows; 256 / TN block slots per pass
  constexpr int SPQ = 4 * TN / 256;  // passes over the block slots (2 for TN = 128, 1 for 64)
  int sp_g[SPQ], sp_ng[SPQ], sp_off[SPQ];
  unsigned sp_a[SPQ];
  bool sp_ok[SPQ];
  float sp_dv[SPQ];
#pragma unroll
  for (int q = 0; q < SPQ; ++q) {
    sp_g[q] = sp_ng[q] = sp_off[q] = 0;
    sp_a[q] = 0u;
    sp_ok[q] = false;
    sp_dv[q] = 0.f;
  }
  float gwt[GPASS];  // weight of the dense part for the rows this thread stages
  auto fetch = [&](int r0) {
#pragma unroll
    for (int p = 0; p < GPASS; ++p) {
      const int row = gr + GR * p;
      rg[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      gwt[p] = 1.f;
      if (r0 + row < rend && n0 + gc4 < N) {
        rg[p] = *reinterpret_cast<const float4 *>(G + (size_t)(r0 + row) * ldg + n0 + gc4);
        if (GPOOL && cm.bw && ((r0 + row) & 7) == 0) gwt[p] = cm.bw[(r0 + row) >> 3];
      }
    }
    if (GPOOL && cm.bgrp) {
      // groups of this step's four blocks: loaded one step ahead (see gemm_nt_kernel)
#pragma unroll
      for (int q = 0; q < SPQ; ++q) {
        const int slot = sp_gi + (256 / TN) * q, blk = (r0 >> 3) + slot;
        if (r0 != rbeg) sp_g[q] = sp_ng[q];
        sp_ok[q] = slot < 4 && (blk << 3) < rend && n0 + sp_n < N;
        if (sp_ok[q]) {
          const int g = sp_g[q];
          sp_off[q] = cm.goff[g] - r0;
          sp_a[q] = garg[(size_t)g * N + n0 + sp_n];
          sp_dv[q] = gdcl[(size_t)g * N + n0 + sp_n];
        }
        const int nblk = ((r0 + BR) >> 3) + slot;
        sp_ng[q] = (slot < 4 && (nblk << 3) < rend) ? cm.bgrp[nblk] : 0;
      }
    } else if (GPOOL) {
      const int g = (r0 >> SSH) + sp_gi;  // group size S = 1 << SSH
      sp_on = (sp_gi << SSH) < BR && (g << SSH) < rend && n0 + sp_n < N;
      if (sp_on) {
        sp_arg = garg[(size_t)g * N + n0 + sp_n];
        sp_d = gdcl[(size_t)g * N + n0 + sp_n];
        const int lr = (g << SSH) + (int)sp_arg - r0;  // local row of the arg-max in this step
        sp_on = lr >= 0 && lr < BR && (g << SSH) + (int)sp_arg < rend;
        sp_arg = (unsigned)lr;
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = xr + 16 * p;
      rx[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + row < rend && k0 + xc4 < K)
        rx[p] = XRC ? *reinterpret_cast<const float4 *>(X + (size_t)(r0 + row) * 4)
                    : *reinterpret_cast<const float4 *>(X + (size_t)(r0 + row) * ldx + k0 + xc4);
    }
  };
  if (GPOOL && cm.bgrp && rbeg < rend) {
#pragma unroll
    for (int q = 0; q < SPQ; ++q) {
      const int slot = sp_gi + (256 / TN) * q, blk = (rbeg >> 3) + slot;
      sp_g[q] = (slot < 4 && (blk << 3) < rend) ? cm.bgrp[blk] : 0;
    }
  }
  if (XRC) {  // 64 k rows x 4 input columns of the first layer's weight
    if (k0 + (tid >> 2) < K) sXw[tid] = xw0[(size_t)k0 * 4 + tid];
    __syncthreads();
  }
  if (rbeg < rend) fetch(rbeg);
  for (int r0 = rbeg; r0 < rend; r0 += BR) {
#pragma unroll
    for (int p = 0; p < GPASS; ++p) {
      float4 v = rg[p];
      if (GPOOL && r0 + gr + GR * p < rend && n0 + gc4 < N) {  // dense part, weighted
        const float wr = gwt[p];
        v.x = wr * fmaf(ga.x, v.x, gb.x);
        v.y = wr * fmaf(ga.y, v.y, gb.y);
        v.z = wr * fmaf(ga.z, v.z, gb.z);
        v.w = wr * fmaf(ga.w, v.w, gb.w);
      }
      *reinterpret_cast<float4 *>(&Gs[(gr + GR * p) * LG + gc4]) = v;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = xr + 16 * p;
      float4 x = rx[p];
      if (XRC && r0 + row < rend && k0 + xc4 < K) x = rc_y4(x, sXw, xc4);
      if (PRO && r0 + row < rend && k0 + xc4 < K) {
        x.x = fmaxf(fmaf(fa.x, x.x, fb.x), 0.f);
        x.y = fmaxf(fmaf(fa.y, x.y, fb.y), 0.f);
        x.z = fmaxf(fmaf(fa.z, x.z, fb.z), 0.f);
        x.w = fmaxf(fmaf(fa.w, x.w, fb.w), 0.f);
      }
      *reinterpret_cast<float4 *>(&Xs[row * LX + xc4]) = x;
    }
    __syncthreads();
    if (GPOOL) {
      if (cm.bgrp) {
#pragma unroll
        for (int q = 0; q < SPQ; ++q) {
          const int lr = sp_off[q] + (int)sp_a[q];
          if (sp_ok[q] && lr >= 0 && (lr >> 3) == sp_gi + (256 / TN) * q)
            Gs[lr * LG + sp_n] += sp_dv[q];
        }
      } else if (sp_on) {
        Gs[(int)sp_arg * LG + sp_n] += sp_d;
      }
      __syncthreads();
    }
    if (r0 + BR < rend) fetch(r0 + BR);  // next rows in flight during the MFMAs
#pragma unroll
    for (int t = 0; t < BR / 2; ++t) {
      const float a = Gs[(2 * t + h) * LG + wn * 32 + l31];
#pragma unroll
      for (int q = 0; q < KT; ++q) {
        const float b = Xs[(2 * t + h) * LX + (wk + q) * 32 + l31];
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  float *out = pw + (size_t)chunk * N * K;
#pragma unroll
  for (int q = 0; q < KT; ++q) {
    const int col = k0 + (wk + q) * 32 + l31;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int row = n0 + wn * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
      if (row < N && col < K) out[(size_t)row * K + col] = acc[q][v];
    }
  }
}

// ---- the plain TN GEMM (no pooled gradient, no first-layer recompute) in bf16x6 ---------------
// The reduction runs over ROWS, so an MFMA operand fragment is 8 consecutive rows of ONE column
// of a row-major tile.  gfx950's LDS transpose read delivers exactly that: within a 16-lane
// group the lanes address a [4 rows][16 columns] block of bf16 (lane p: row p/4, columns
// 4*(p%4)..+3, 8 bytes) and ds_read_b64_tr_b16 hands lane i column i of the block, rows 0..3
// (tools/probe/tr_probe.hip prints the mapping).  So the tiles are staged exactly like the NT
// kernel's -- coalesced float4 loads, exact 3-way bf16 split, 8-byte plane stores, row-major --
// and two transpose reads per plane give the 8-row fragment.  Row pitches of 320 B / 192 B put
// the four rows of a block 16 / 48 banks apart: the 32 lanes of a read cycle hit 64 distinct banks.
// (A first bf16x6 form staged the tiles transposed with one column per thread: plain 16-byte
// LDS traffic but 4-byte global loads, 96 instead of 24 vector-memory instructions per 32-row
// step, address-unit bound and slower than the f32-input kernel: 78 -> 136 us on
// 114 624 x 128 x 132.)
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 tr_read8(const __bf16 *p, int pitch) {
  typedef __attribute__((address_space(3))) s16x4 *lds_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * pitch));
  union {
    s16x4 s[2];
    bf16x8 b;
  } u;
  u.s[0] = lo;
  u.s[1] = hi;
  return u.b;
}

// GPOOL / XRC as in gemm_tn_kernel.  GPOOL's sparse part cannot be added to the planes after the
// split, so -- as in gemm_nt_kernel's PRO == 2 -- the threads that hold the (slot, n) entries of
// the step write a small table (local row of the arg-max or -1, value) and the staging thread adds
// its entries before splitting.
template <int TNW, bool PRO, bool GPOOL = false, bool XRC = false>
__global__ __launch_bounds__(256) void gemm_tn_x6_kernel(
    const float *__restrict__ G, int ldg, const float *__restrict__ X, int ldx, int R, int N,
    int K, const float *__restrict__ pa, const float *__restrict__ pb, int rows_per_chunk,
    float *__restrict__ pw, const unsigned char *__restrict__ garg,
    const float *__restrict__ gdcl, const float *__restrict__ galpha,
    const float *__restrict__ gbeta, int SSH, const float *__restrict__ xw0, Compact cm) {
  if (cm.dims) {  // compact rows: the row count lives on the device, and so does the split of
    R = cm.dims[0];  // the rows over the chunks (by the host's dense bound half the grid idles)
    rows_per_chunk = ((R + (int)gridDim.z - 1) / (int)gridDim.z + 31) / 32 * 32;
  }
  constexpr int BR = 32;
  constexpr int TN = 32 * TNW;
  constexpr int LG = TN == 128 ? 160 : 96, LX = 96;   // bf16 row pitches: 320 B / 192 B
  constexpr int KT = TNW == 4 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) __bf16 Gp[3 * BR * LG];
  __shared__ __attribute__((aligned(16))) __bf16 Xp[3 * BR * LX];
  __shared__ __attribute__((aligned(16))) float sXw[XRC ? 64 * 4 : 4];  // W0 rows k0 .. k0+63
  // GPOOL: per step, slot = group (plain rows) / 8-row block (compact rows) of the 32 rows
  __shared__ __attribute__((aligned(16))) int sLr[GPOOL ? 4 * TN : 4];
  __shared__ __attribute__((aligned(16))) float sDv[GPOOL ? 4 * TN : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = TNW == 4 ? wave : (wave >> 1);
  const int wk = TNW == 4 ? 0 : (wave & 1);
  const int l31 = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * TN, k0 = blockIdx.y * 64;
  const int chunk = blockIdx.z;
  const int rbeg = chunk * rows_per_chunk;
  const int rend = min(R, rbeg + rows_per_chunk);

  f32x16 acc[KT];
#pragma unroll
  for (int q = 0; q < KT; ++q)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[q][v] = 0.f;

  const int xc4 = (tid & 15) * 4, xr = tid >> 4;
  constexpr int GT = TN / 4, GR = 256 / GT, GPASS = BR / GR;
  const int gc4 = (tid % GT) * 4, gr = tid / GT;
  float4 fa = make_float4(1.f, 1.f, 1.f, 1.f), fb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (PRO && k0 + xc4 < K) {
    fa = *reinterpret_cast<const float4 *>(pa + k0 + xc4);
    fb = *reinterpret_cast<const float4 *>(pb + k0 + xc4);
  }
  float4 ga = make_float4(0.f, 0.f, 0.f, 0.f), gb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (GPOOL && n0 + gc4 < N) {
    ga = *reinterpret_cast<const float4 *>(galpha + n0 + gc4);
    gb = *reinterpret_cast<const float4 *>(gbeta + n0 + gc4);
  }
  float4 rg[GPASS], rx[2];
  // GPOOL: this thread's (slot, n column) entries of the step's sparse table
  const int sp_gi = tid / TN, sp_n = tid % TN;
  constexpr int SPQ = 4 * TN / 256;  // passes over the 4 slots (2 for TN = 128, 1 for 64)
  int sp_g[SPQ], sp_ng[SPQ], sp_lr[SPQ];
  float sp_dv[SPQ];
#pragma unroll
  for (int q = 0; q < SPQ; ++q) {
    sp_g[q] = sp_ng[q] = 0;
    sp_lr[q] = -1;
    sp_dv[q] = 0.f;
  }
  float gwt[GPASS];  // weight of the dense part for the rows this thread stages
  auto fetch = [&](int r0) {
#pragma unroll
    for (int p = 0; p < GPASS; ++p) {
      const int row = gr + GR * p;
      rg[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      gwt[p] = 1.f;
      if (r0 + row < rend && n0 + gc4 < N) {
        rg[p] = *reinterpret_cast<const float4 *>(G + (size_t)(r0 + row) * ldg + n0 + gc4);
        if (GPOOL && cm.bw && ((r0 + row) & 7) == 0) gwt[p] = cm.bw[(r0 + row) >> 3];
      }
    }
    if (GPOOL && cm.bgrp) {
      // groups of this step's four blocks: loaded one step ahead (see gemm_nt_kernel)
#pragma unroll
      for (int q = 0; q < SPQ; ++q) {
        const int slot = sp_gi + (256 / TN) * q, blk = (r0 >> 3) + slot;
        if (r0 != rbeg) sp_g[q] = sp_ng[q];
        sp_lr[q] = -1;
        if (slot < 4 && (blk << 3) < rend && n0 + sp_n < N) {
          const int g = sp_g[q];
          const int lr = cm.goff[g] - r0 + (int)garg[(size_t)g * N + n0 + sp_n];
          sp_dv[q] = gdcl[(size_t)g * N + n0 + sp_n];
          sp_lr[q] = (lr >= 0 && (lr >> 3) == slot) ? lr : -1;
        }
        const int nblk = ((r0 + BR) >> 3) + slot;
        sp_ng[q] = (slot < 4 && (nblk << 3) < rend) ? cm.bgrp[nblk] : 0;
      }
    } else if (GPOOL) {
#pragma unroll
      for (int q = 0; q < SPQ; ++q) {
        const int slot = sp_gi + (256 / TN) * q;
        const int g = (r0 >> SSH) + slot;  // group size S = 1 << SSH
        sp_lr[q] = -1;
        if ((slot << SSH) < BR && (g << SSH) < rend && n0 + sp_n < N) {
          const int arow = (g << SSH) + (int)garg[(size_t)g * N + n0 + sp_n];
          sp_dv[q] = gdcl[(size_t)g * N + n0 + sp_n];
          const int lr = arow - r0;  // local row of the arg-max in this step
          sp_lr[q] = (lr >= 0 && lr < BR && arow < rend) ? lr : -1;
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = xr + 16 * p;
      rx[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + row < rend && k0 + xc4 < K)
        rx[p] = XRC ? *reinterpret_cast<const float4 *>(X + (size_t)(r0 + row) * 4)
                    : *reinterpret_cast<const float4 *>(X + (size_t)(r0 + row) * ldx + k0 + xc4);
    }
  };
  if (GPOOL && cm.bgrp && rbeg < rend) {
#pragma unroll
    for (int q = 0; q < SPQ; ++q) {
      const int slot = sp_gi + (256 / TN) * q, blk = (rbeg >> 3) + slot;
      sp_g[q] = (slot < 4 && (blk << 3) < rend) ? cm.bgrp[blk] : 0;
    }
  }
  if (XRC) {  // 64 k rows x 4 input columns of the first layer's weight
    if (k0 + (tid >> 2) < K) sXw[tid] = xw0[(size_t)k0 * 4 + tid];
    __syncthreads();
  }
  // fragment addressing of the transpose reads: lane -> (row, column) of its 8-byte piece
  const int p16 = lane & 15, grp = lane >> 4;
  const int frow = 8 * (grp >> 1) + (p16 >> 2), fcol = 16 * (grp & 1) + 4 * (p16 & 3);
  if (rbeg < rend) fetch(rbeg);
  for (int r0 = rbeg; r0 < rend; r0 += BR) {
    if constexpr (GPOOL) {   // the step's sparse table
#pragma unroll
      for (int q = 0; q < SPQ; ++q) {
        const int slot = sp_gi + (256 / TN) * q;
        sLr[slot * TN + sp_n] = sp_lr[q];
        sDv[slot * TN + sp_n] = sp_dv[q];
      }
      __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < GPASS; ++p) {
      float4 v = rg[p];
      const int row = gr + GR * p;
      if (GPOOL && r0 + row < rend && n0 + gc4 < N) {  // dense part, weighted, + sparse part
        const float wr = gwt[p];
        v.x = wr * fmaf(ga.x, v.x, gb.x);
        v.y = wr * fmaf(ga.y, v.y, gb.y);
        v.z = wr * fmaf(ga.z, v.z, gb.z);
        v.w = wr * fmaf(ga.w, v.w, gb.w);
        const int slot = cm.bgrp ? (row >> 3) : (row >> SSH);
        const int4 lr = *reinterpret_cast<const int4 *>(&sLr[slot * TN + gc4]);
        const float4 dv = *reinterpret_cast<const float4 *>(&sDv[slot * TN + gc4]);
        v.x += lr.x == row ? dv.x : 0.f;
        v.y += lr.y == row ? dv.y : 0.f;
        v.z += lr.z == row ? dv.z : 0.f;
        v.w += lr.w == row ? dv.w : 0.f;
      }
      const Split4 sp = split4(v);
      const int at = row * LG + gc4;
      *reinterpret_cast<bf16x4 *>(&Gp[0 * BR * LG + at]) = sp.h;
      *reinterpret_cast<bf16x4 *>(&Gp[1 * BR * LG + at]) = sp.m;
      *reinterpret_cast<bf16x4 *>(&Gp[2 * BR * LG + at]) = sp.l;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = xr + 16 * p;
      float4 x = rx[p];
      if (XRC && r0 + row < rend && k0 + xc4 < K) x = rc_y4(x, sXw, xc4);
      if (PRO && r0 + row < rend && k0 + xc4 < K) {
        x.x = fmaxf(fmaf(fa.x, x.x, fb.x), 0.f);
        x.y = fmaxf(fmaf(fa.y, x.y, fb.y), 0.f);
        x.z = fmaxf(fmaf(fa.z, x.z, fb.z), 0.f);
        x.w = fmaxf(fmaf(fa.w, x.w, fb.w), 0.f);
      }
      const Split4 sp = split4(x);
      const int at = row * LX + xc4;
      *reinterpret_cast<bf16x4 *>(&Xp[0 * BR * LX + at]) = sp.h;
      *reinterpret_cast<bf16x4 *>(&Xp[1 * BR * LX + at]) = sp.m;
      *reinterpret_cast<bf16x4 *>(&Xp[2 * BR * LX + at]) = sp.l;
    }
    __syncthreads();
    if (r0 + BR < rend) fetch(r0 + BR);  // next rows in flight during the MFMAs
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[3], bf[3][KT];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        af[q] = tr_read8(&Gp[(q * BR + ks * 16 + frow) * LG + wn * 32 + fcol], LG);
#pragma unroll
        for (int t = 0; t < KT; ++t)
          bf[q][t] = tr_read8(&Xp[(q * BR + ks * 16 + frow) * LX + (wk + t) * 32 + fcol], LX);
      }
      // smallest terms first (see gemm_nt_kernel)
#define BTR_X6(QA, QB)                       \
  _Pragma("unroll") for (int t = 0; t < KT; ++t) acc[t] = \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[QA], bf[QB][t], acc[t], 0, 0, 0);
      BTR_X6(2, 0)
      BTR_X6(0, 2)
      BTR_X6(1, 1)
      BTR_X6(1, 0)
      BTR_X6(0, 1)
      BTR_X6(0, 0)
#undef BTR_X6
    }
    __syncthreads();
  }
  float *out = pw + (size_t)chunk * N * K;
#pragma unroll
  for (int q = 0; q < KT; ++q) {
    const int col = k0 + (wk + q) * 32 + l31;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int row = n0 + wn * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
      if (row < N && col < K) out[(size_t)row * K + col] = acc[q][v];
    }
  }
}

// ---- streaming forward / input-gradient GEMM for the narrow layers ----------------------------
// gemm_nt_kernel re-stages and re-splits its weight tile for every 128-row tile -- for SA1's
// 128 x 64 last layer that is as much staging work as the activations themselves -- and its
// epilogue (statistics, pooling extrema, transposed store) runs with nothing in flight.  For
// n <= 128, k <= 128 the sa_bwd_fused_kernel layout does better: 32 (n <= 128) or 64 (n <= 64)
// rows per step, the whole k extent staged once per step as bf16 planes (double-buffered: one
// barrier per step), every wave owns one 32 x 32 output tile whose B fragments -- its 32 weight
// rows, split once -- stay in registers for the whole kernel, the next step's rows are in flight
// during the MFMAs, and the epilogue works on the accumulators: BatchNorm statistics (compact-row
// weights as in gemm_nt_kernel), the pooling extrema of 8-row blocks, the C tile through a
// per-wave LDS transpose as 16-byte row stores.  Same arithmetic as gemm_nt_kernel<.., MM = 1>
// (bf16x6, smallest terms first); the statistics partials keep its layout (one row of `part`
// per workgroup, gridDim.x = btr_sa_gemm_grid(rows)), so bn_finalize_kernel does not change.
// PRO: 0 plain, 1 relu(pa * a + pb), 3 first-layer recompute (a = x0 rows of 4, w0 [k][4]).
struct StreamArgs {
  const float *A;
  int lda;
  const float *W;   // [N][ldw]
  int ldw;
  float *C;
  int ldc;
  int R, N, K, rows_per_chunk;
  const float *pa, *pb, *w0;
  float *part;
  const float *gsign;   // PS > 0: the layer's gamma
  float *gext;
  unsigned char *aext;
  int hs;               // log2 of the 128-column slabs of n (n > 128: two workgroups per row chunk)
};

template <int BR, int KMAX, int PRO, bool STATS, int PS>
__global__ __launch_bounds__(256, 2) void sa_fwd_stream_kernel(StreamArgs a, Compact cm) {
  static_assert((BR == 32 || BR == 64) && (KMAX == 64 || KMAX == 128), "tile shapes");
  static_assert(PS == 0 || PS == 8 || PS == 16,
                "pooling epilogue: 8-row blocks (compact rows) or groups of 16 rows");
  constexpr int WR = BR / 32;           // row tiles (waves along the rows)
  constexpr int WC = 4 / WR;            // column tiles: n <= 32 * WC
  constexpr int LX = KMAX + 8;          // bf16 pitch: 144 / 272 B (conflict-free 16-byte row reads)
  constexpr int KS = KMAX / 16;         // reduction steps of 16
  constexpr int TPR = KMAX / 4;         // staging threads per row
  constexpr int RP = 256 / TPR;         // rows per staging pass
  constexpr int NP = BR / RP;           // staging passes
  constexpr int LT = 36;                // f32 pitch of the per-wave transpose tile
  int R = a.R, rows_per_chunk = a.rows_per_chunk;
  if (cm.dims) {   // compact rows: count and split on the device (see sa_bwd_fused_kernel)
    R = cm.dims[0];
    const int nchunks = (int)gridDim.x >> a.hs;
    rows_per_chunk = ((R + nchunks - 1) / nchunks + BR - 1) / BR * BR;
  }
  const int N = a.N, K = a.K;
  // n > 128: neighbouring workgroups share a row chunk and take one 128-column slab each
  const int chunk = (int)blockIdx.x >> a.hs, cb = ((int)blockIdx.x & ((1 << a.hs) - 1)) * 128;
  __shared__ __attribute__((aligned(16))) __bf16 Xp[2][3 * BR * LX];
  __shared__ __attribute__((aligned(16))) float Ts[4][32 * LT];
  __shared__ double red[STATS ? 2 * WR * 32 * WC : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int l31 = lane & 31, h = lane >> 5;
  const int rbeg = chunk * rows_per_chunk;
  const int rend = min(R, rbeg + rows_per_chunk);

  // ---- this wave's 32 weight rows, split once
  bf16x8 bdr[KS][3];
  {
    const int nr = cb + wc * 32 + l31;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int kb = ks * 16 + h * 8;
      float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
      if (nr < N && kb < K) w0 = *reinterpret_cast<const float4 *>(a.W + (size_t)nr * a.ldw + kb);
      if (nr < N && kb + 4 < K)
        w1 = *reinterpret_cast<const float4 *>(a.W + (size_t)nr * a.ldw + kb + 4);
      const Split4 s0 = split4(w0), s1 = split4(w1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bdr[ks][0][e] = s0.h[e]; bdr[ks][0][4 + e] = s1.h[e];
        bdr[ks][1][e] = s0.m[e]; bdr[ks][1][4 + e] = s1.m[e];
        bdr[ks][2][e] = s0.l[e]; bdr[ks][2][4 + e] = s1.l[e];
      }
    }
  }
  const int xc4 = (tid % TPR) * 4, xr = tid / TPR;
  float4 fa = make_float4(1.f, 1.f, 1.f, 1.f), fb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (PRO && xc4 < K) {
    fa = *reinterpret_cast<const float4 *>(a.pa + xc4);
    fb = *reinterpret_cast<const float4 *>(a.pb + xc4);
  }
  float4 w0r[PRO == 3 ? 4 : 1];
  if constexpr (PRO == 3) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      w0r[e] = (xc4 + e < K) ? *reinterpret_cast<const float4 *>(a.w0 + (size_t)(xc4 + e) * 4)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 rx[NP];
  auto fetch = [&](int r0) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int row = xr + RP * p;
      rx[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + row < rend && xc4 < K)
        rx[p] = PRO == 3 ? *reinterpret_cast<const float4 *>(a.A + (size_t)(r0 + row) * 4)
                         : *reinterpret_cast<const float4 *>(a.A + (size_t)(r0 + row) * a.lda + xc4);
    }
  };
  const float sg =
      (PS > 0 && cb + wc * 32 + l31 < N && a.gsign[cb + wc * 32 + l31] < 0.f) ? -1.f : 1.f;
  double d1 = 0.0, d2 = 0.0;
  if (rbeg < rend) fetch(rbeg);
  int buf = 0;
  for (int r0 = rbeg; r0 < rend; r0 += BR, buf ^= 1) {
    __bf16 *xp = Xp[buf];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int row = xr + RP * p;
      float4 x = rx[p];
      const bool live = r0 + row < rend && xc4 < K;
      if constexpr (PRO == 3) {
        if (live)
          x = make_float4(rc_dot4(x, w0r[0]), rc_dot4(x, w0r[1]), rc_dot4(x, w0r[2]),
                          rc_dot4(x, w0r[3]));
      }
      if (PRO != 0 && live) {
        x.x = fmaxf(fmaf(fa.x, x.x, fb.x), 0.f);
        x.y = fmaxf(fmaf(fa.y, x.y, fb.y), 0.f);
        x.z = fmaxf(fmaf(fa.z, x.z, fb.z), 0.f);
        x.w = fmaxf(fmaf(fa.w, x.w, fb.w), 0.f);
      }
      if (!live) x = make_float4(0.f, 0.f, 0.f, 0.f);   // padded rows / columns stay exactly 0
      const Split4 sp = split4(x);
      const int at = row * LX + xc4;
      *reinterpret_cast<bf16x4 *>(&xp[0 * BR * LX + at]) = sp.h;
      *reinterpret_cast<bf16x4 *>(&xp[1 * BR * LX + at]) = sp.m;
      *reinterpret_cast<bf16x4 *>(&xp[2 * BR * LX + at]) = sp.l;
    }
    float bwx[4];   // STATS on compact rows: (weight - 1) of this wave's four 8-row blocks
    if (STATS && cm.bw) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = r0 + wr * 32 + 8 * q;
        bwx[q] = row < rend ? cm.bw[row >> 3] - 1.f : 0.f;
      }
    }
    __syncthreads();   // (the other buffer is free: every wave finished its MFMAs of the step
                       // before last when it arrived here)
    if (r0 + BR < rend) fetch(r0 + BR);
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks * 16 >= K) break;
      bf16x8 af[3];
#pragma unroll
      for (int q = 0; q < 3; ++q)
        af[q] = *reinterpret_cast<const bf16x8 *>(
            &xp[(q * BR + wr * 32 + l31) * LX + ks * 16 + h * 8]);
      const bf16x8 *bd = bdr[ks];
#define BTR_X6S(QA, QB) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[QA], bd[QB], acc, 0, 0, 0);
      BTR_X6S(2, 0)
      BTR_X6S(0, 2)
      BTR_X6S(1, 1)
      BTR_X6S(1, 0)
      BTR_X6S(0, 1)
      BTR_X6S(0, 0)
#undef BTR_X6S
    }
    // ---- epilogue on the accumulators: D layout col = lane & 31, row = (v&3) + 8*(v>>2) + 4*h
    const int col = cb + wc * 32 + l31;
    const int wrow0 = r0 + wr * 32;
    if constexpr (STATS) {   // rows >= rend hold exact zeros
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const float c = acc[v];
        s1 += c;
        s2 = fmaf(c, c, s2);
        if (cm.bw && h == 0 && (v & 3) == 0) {
          const float wx = bwx[v >> 2];
          s1 = fmaf(wx, c, s1);
          s2 = fmaf(wx * c, c, s2);
        }
      }
      d1 += (double)s1;
      d2 += (double)s2;
    }
    if constexpr (PS > 0) {
      float bestv[4];
      int bestr[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        float vmx = -3.0e38f;
        int imx = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float c = acc[4 * b + q] * sg;
          if (c > vmx) {
            vmx = c;
            imx = q;
          }
        }
        const int mine = imx + 4 * h;
        const float omx = __shfl_xor(vmx, 32);
        const int oix = __shfl_xor(mine, 32);
        const bool take = omx > vmx || (omx == vmx && oix < mine);
        bestv[b] = take ? omx : vmx;
        bestr[b] = take ? oix : mine;
      }
      if constexpr (PS == 8) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int grow = wrow0 + b * 8;
          if (h == 0 && grow < rend && col < N) {
            const size_t o = (size_t)(grow >> 3) * N + col;
            a.gext[o] = bestv[b] * sg;
            a.aext[o] = (unsigned char)bestr[b];
          }
        }
      } else {   // groups of 16 rows: two blocks, the first occurrence wins a tie
#pragma unroll
        for (int b = 0; b < 4; b += 2) {
          const bool second = bestv[b + 1] > bestv[b];
          const int grow = wrow0 + b * 8;
          if (h == 0 && grow < rend && col < N) {
            const size_t o = (size_t)(grow >> 4) * N + col;
            a.gext[o] = (second ? bestv[b + 1] : bestv[b]) * sg;
            a.aext[o] = (unsigned char)(second ? 8 + bestr[b + 1] : bestr[b]);
          }
        }
      }
    }
    if (a.C != nullptr) {
      float *T = Ts[wave];
#pragma unroll
      for (int v = 0; v < 16; ++v) T[((v & 3) + 8 * (v >> 2) + 4 * h) * LT + l31] = acc[v];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the tile is private to the wave)
      __builtin_amdgcn_wave_barrier();
      const int rl = lane >> 3, c4 = (lane & 7) * 4;   // 8 rows x 8 float4 per pass
      const int colw = cb + wc * 32 + c4;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const float4 q4 = *reinterpret_cast<const float4 *>(&T[(it * 8 + rl) * LT + c4]);
        const int row = wrow0 + it * 8 + rl;
        if (row < rend && colw < N)
          *reinterpret_cast<float4 *>(a.C + (size_t)row * a.ldc + colw) = q4;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if constexpr (STATS) {
    d1 += __shfl_xor(d1, 32);
    d2 += __shfl_xor(d2, 32);
    if (h == 0) {
      red[(0 * WR + wr) * 32 * WC + wc * 32 + l31] = d1;
      red[(1 * WR + wr) * 32 * WC + wc * 32 + l31] = d2;
    }
    __syncthreads();
    for (int c = tid; c < 2 * 32 * WC; c += 256) {
      const int which = c / (32 * WC), col = c % (32 * WC);
      double sum = 0.0;
#pragma unroll
      for (int w = 0; w < WR; ++w) sum += red[(which * WR + w) * 32 * WC + col];
      if (cb + col < N) a.part[((size_t)chunk * 2 + which) * N + cb + col] = (float)sum;
    }
  }
}

// A producer / consumer form of this forward kernel (512 threads: four waves stage, four multiply
// and run the previous step's epilogue; bit-identical results) was built in round 5 and measured
// SLOWER on every layer but one (tools/fwd_ws_ab.py, alone on the chip; single-role / two chunks
// per workgroup / one chunk per workgroup): SA1's pooled layer 706 560 x 128 x 64 with the Y store
// 160 / 258 / 182 us, without 100 / 172 / 118; SA1's 64-wide layer 72 / 73 / 74; SA2's hidden layer
// 40 / 37 / 37; SA2's pooled layer 68 / 83 / 90; SA3's 46 / 53 / 71 (profiles/r05_*_fwd_ws_ab.txt).
// The forward step carries 24 (k = 64) or 48 MFMAs per wave against ~250 VALU instructions of
// epilogue: it is bound by issue latency with two waves per SIMD either way, and splitting the
// roles does not add waves -- unlike the pooled layer's backward (72 MFMAs per step), where it
// does pay (sa_bwd_gram_ws_kernel).  Removed in round 6.

// ---- single-launch inference set-abstraction layer ---------------------------------------------
// eval mode (module.eval(), pointnet2_modules.py:210-272 under the evaluation pass of
// train_Votenet_FSB.py:246-293): BatchNorm uses its RUNNING statistics, so no batch reduction
// separates the layers -- gather -> conv -> BN -> ReLU -> conv -> BN -> ReLU -> conv -> BN -> ReLU ->
// max over the group can run per 64 rows without anything of size rows x channels touching HBM
// (the three-launch eval path writes and re-reads 268 + 268 + 537 MB for SA1 at the benchmark
// shape).  For the layer shape that carries those bytes: input of <= 4 columns (xyz + one
// feature), widths <= 64, <= 64, <= 128, nsample 16 / 32 / 64.
//   stage 0  a thread gathers its row's 4 input values (neighbour index -> coordinates relative to
//            the centre, scaled; the feature) -- loaded one step ahead -- and evaluates 16 of the
//            64 first-layer outputs (k = 4: plain FMAs), BN + ReLU, bf16 split -> planes X1
//   stage 1  64 x 64 x 64 on the matrix pipe (2 x 2 waves; W1 fragments resident in registers),
//            BN + ReLU on the accumulators, split -> planes X2 (2-byte LDS stores in row layout)
//   stage 2  64 x 128 x 64 (a wave owns 32 columns of all 64 rows; W2 fragments resident),
//            BN + ReLU, max over the rows of each group -- all of a group's rows of a column sit
//            in one wave -- and the pooled values are the only thing written.
// Same arithmetic as the multi-launch path (bf16x6 products, relu(fma(a, y, b))), so the results
// agree to rounding; tests/test_eval_fused_gpu.py and the votenet_eval golden.
struct EvalArgs {
  const float *xyz, *new_xyz, *feat_cl;   // (B, N, 3), (B, M, 3), (B, N, C) or NULL
  const int *idx;                          // (B, M, S)
  const float *w0, *w1, *w2;               // [c1][4], [c2][ld1], [c3][ld2]
  int ld1, ld2;
  const float *a0, *b0, *a1, *b1, *a2, *b2;   // eval-mode BatchNorm as scale / shift per layer
  float *out, *out_cl;                     // (B, c3, M), (B, M, c3)
  int B, N, M, S, C, use_xyz, c1, c2, c3, steps_per_wg;
  float inv_radius;
};

__global__ __launch_bounds__(256, 2) void sa_eval_fused_kernel(EvalArgs a) {
  constexpr int BR = 64, LX = 72;   // rows per step; bf16 pitch (144 B: conflict-free row reads)
  __shared__ __attribute__((aligned(16))) __bf16 X1[3 * BR * LX];
  __shared__ __attribute__((aligned(16))) __bf16 X2[3 * BR * LX];
  __shared__ __attribute__((aligned(16))) float sW0[64 * 4], sA0[64], sB0[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const long long rows = (long long)a.B * a.M * a.S;
  const long long nsteps = (rows + BR - 1) / BR;
  const long long s_beg = (long long)blockIdx.x * a.steps_per_wg;
  const long long s_end = min(nsteps, s_beg + a.steps_per_wg);
  for (int i = tid; i < 64 * 4; i += 256) sW0[i] = (i >> 2) < a.c1 ? a.w0[i] : 0.f;
  if (tid < 64) {
    sA0[tid] = tid < a.c1 ? a.a0[tid] : 0.f;
    sB0[tid] = tid < a.c1 ? a.b0[tid] : 0.f;
  }
  // weight fragments: stage 1 tile (wr1, wc1) of 32 x 32, stage 2 columns 32 * wave
  const int wr1 = wave >> 1, wc1 = wave & 1;
  bf16x8 f1[4][3], f2[4][3];
  float a1c = 0.f, b1c = 0.f, a2c = 0.f, b2c = 0.f;
  {
    const int n1 = wc1 * 32 + l31, n2 = wave * 32 + l31;
    if (n1 < a.c2) { a1c = a.a1[n1]; b1c = a.b1[n1]; }
    if (n2 < a.c3) { a2c = a.a2[n2]; b2c = a.b2[n2]; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kb = ks * 16 + h * 8;
      float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), u1 = u0, v0 = u0, v1 = u0;
      if (n1 < a.c2 && kb < a.c1) u0 = *reinterpret_cast<const float4 *>(a.w1 + (size_t)n1 * a.ld1 + kb);
      if (n1 < a.c2 && kb + 4 < a.c1) u1 = *reinterpret_cast<const float4 *>(a.w1 + (size_t)n1 * a.ld1 + kb + 4);
      if (n2 < a.c3 && kb < a.c2) v0 = *reinterpret_cast<const float4 *>(a.w2 + (size_t)n2 * a.ld2 + kb);
      if (n2 < a.c3 && kb + 4 < a.c2) v1 = *reinterpret_cast<const float4 *>(a.w2 + (size_t)n2 * a.ld2 + kb + 4);
      const Split4 su0 = split4(u0), su1 = split4(u1), sv0 = split4(v0), sv1 = split4(v1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f1[ks][0][e] = su0.h[e]; f1[ks][0][4 + e] = su1.h[e];
        f1[ks][1][e] = su0.m[e]; f1[ks][1][4 + e] = su1.m[e];
        f1[ks][2][e] = su0.l[e]; f1[ks][2][4 + e] = su1.l[e];
        f2[ks][0][e] = sv0.h[e]; f2[ks][0][4 + e] = sv1.h[e];
        f2[ks][1][e] = sv0.m[e]; f2[ks][1][4 + e] = sv1.m[e];
        f2[ks][2][e] = sv0.l[e]; f2[ks][2][4 + e] = sv1.l[e];
      }
    }
  }
  __syncthreads();
  // stage 0 mapping: thread -> (row = tid / 4, columns 16 * (tid % 4) .. + 15)
  const int grow = tid >> 2, gq = tid & 3;
  const long long ms = (long long)a.M * a.S;
  float4 x0n = make_float4(0.f, 0.f, 0.f, 0.f);
  auto gather = [&](long long step) {
    const long long r = step * BR + grow;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) {
      const long long bi = r / ms;
      const long long g = r / a.S;   // b * M + m
      const int ii = a.idx[r];
      int c = 0;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.use_xyz) {
        const float *p = a.xyz + ((size_t)bi * a.N + ii) * 3;
        const float *q = a.new_xyz + (size_t)g * 3;
        v[0] = (p[0] - q[0]) * a.inv_radius;
        v[1] = (p[1] - q[1]) * a.inv_radius;
        v[2] = (p[2] - q[2]) * a.inv_radius;
        c = 3;
      }
      for (int j = 0; j < a.C && c < 4; ++j, ++c) v[c] = a.feat_cl[((size_t)bi * a.N + ii) * a.C + j];
      x = make_float4(v[0], v[1], v[2], v[3]);
    }
    return x;
  };
  if (s_beg < s_end) x0n = gather(s_beg);
  for (long long step = s_beg; step < s_end; ++step) {
    const float4 x0 = x0n;
    // ---- stage 0: first layer on the VALU, BN + ReLU, split
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int c4 = gq * 16 + p * 4;
      const float4 *w = reinterpret_cast<const float4 *>(sW0) + c4;
      const float4 fa = *reinterpret_cast<const float4 *>(&sA0[c4]);
      const float4 fb = *reinterpret_cast<const float4 *>(&sB0[c4]);
      float4 y = make_float4(rc_dot4(x0, w[0]), rc_dot4(x0, w[1]), rc_dot4(x0, w[2]),
                             rc_dot4(x0, w[3]));
      y.x = fmaxf(fmaf(fa.x, y.x, fb.x), 0.f);
      y.y = fmaxf(fmaf(fa.y, y.y, fb.y), 0.f);
      y.z = fmaxf(fmaf(fa.z, y.z, fb.z), 0.f);
      y.w = fmaxf(fmaf(fa.w, y.w, fb.w), 0.f);
      const Split4 sp = split4(y);
      const int at = grow * LX + c4;
      *reinterpret_cast<bf16x4 *>(&X1[0 * BR * LX + at]) = sp.h;
      *reinterpret_cast<bf16x4 *>(&X1[1 * BR * LX + at]) = sp.m;
      *reinterpret_cast<bf16x4 *>(&X1[2 * BR * LX + at]) = sp.l;
    }
    __syncthreads();   // X1 complete; every wave is past stage 2 of the step before (X2 free)
    if (step + 1 < s_end) x0n = gather(step + 1);   // the next rows' loads fly under the MFMAs
    // ---- stage 1: X1 (64 x c1) . W1^T -> 32 x 32 tile (wr1, wc1)
    {
      f32x16 acc;
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks * 16 >= a.c1) break;
        bf16x8 af[3];
#pragma unroll
        for (int q = 0; q < 3; ++q)
          af[q] = *reinterpret_cast<const bf16x8 *>(
              &X1[(q * BR + wr1 * 32 + l31) * LX + ks * 16 + h * 8]);
#define BTR_X6E(F, QA, QB) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[QA], F[ks][QB], acc, 0, 0, 0);
        BTR_X6E(f1, 2, 0)
        BTR_X6E(f1, 0, 2)
        BTR_X6E(f1, 1, 1)
        BTR_X6E(f1, 1, 0)
        BTR_X6E(f1, 0, 1)
        BTR_X6E(f1, 0, 0)
      }
      // BN + ReLU, split, row-layout planes: element (row, col = wc1 * 32 + l31)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int row = wr1 * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
        const float y = fmaxf(fmaf(a1c, acc[v], b1c), 0.f);
        const __bf16 bh = (__bf16)y;
        const float r1 = y - (float)bh;
        const __bf16 bm = (__bf16)r1;
        const __bf16 bl = (__bf16)(r1 - (float)bm);
        const int at = row * LX + wc1 * 32 + l31;
        X2[0 * BR * LX + at] = bh;
        X2[1 * BR * LX + at] = bm;
        X2[2 * BR * LX + at] = bl;
      }
    }
    __syncthreads();   // X2 complete (and every wave is done reading X1)
    // ---- stage 2: X2 (64 x c2) . W2^T -> columns 32 * wave, both row tiles; BN + ReLU + max
    {
      f32x16 acc2[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc2[t][v] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks * 16 >= a.c2) break;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          bf16x8 af[3];
#pragma unroll
          for (int q = 0; q < 3; ++q)
            af[q] = *reinterpret_cast<const bf16x8 *>(
                &X2[(q * BR + t * 32 + l31) * LX + ks * 16 + h * 8]);
          f32x16 acc = acc2[t];
          BTR_X6E(f2, 2, 0)
          BTR_X6E(f2, 0, 2)
          BTR_X6E(f2, 1, 1)
          BTR_X6E(f2, 1, 0)
          BTR_X6E(f2, 0, 1)
          BTR_X6E(f2, 0, 0)
          acc2[t] = acc;
        }
      }
#undef BTR_X6E
      // max over the rows of each group: 8-row blocks first (v >> 2 = block of the tile, both h)
      float bmax[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          float mx = 0.f;   // (relu output: >= 0)
#pragma unroll
          for (int q = 0; q < 4; ++q) mx = fmaxf(mx, fmaxf(fmaf(a2c, acc2[t][4 * b + q], b2c), 0.f));
          mx = fmaxf(mx, __shfl_xor(mx, 32));
          bmax[t][b] = mx;
        }
      const int col = wave * 32 + l31;
      const int gps = BR / a.S;   // groups per step: 1, 2, 4
      if (h == 0 && col < a.c3) {
        for (int g = 0; g < gps; ++g) {
          const long long grp = step * gps + g;
          if (grp * a.S >= rows) break;
          float mx = 0.f;
          const int b0 = g * (a.S / 8), b1 = b0 + a.S / 8;   // 8-row blocks [b0, b1) of the step
#pragma unroll
          for (int bb = 0; bb < 8; ++bb)
            if (bb >= b0 && bb < b1) mx = fmaxf(mx, bmax[bb >> 2][bb & 3]);
          const long long bi = grp / a.M;
          const int m = (int)(grp - bi * a.M);
          a.out[((size_t)bi * a.c3 + col) * a.M + m] = mx;
          a.out_cl[(size_t)grp * a.c3 + col] = mx;
        }
      }
    }
  }
}

// ---- one pass over a layer's backward operands: dgrad + wgrad + the next BatchNorm's sums ------
// The backward of a hidden layer l used to be five passes over its big tensors: the weight
// gradient dW_l = dY_l^T . X_{l-1} and the input gradient dZ_{l-1} = dY_l . W_l each read dY_l,
// then BatchNorm_{l-1}'s backward read (dZ_{l-1}, Y_{l-1}) once for its sums and once more to
// apply them (writing dY_{l-1}), which the next two GEMMs read again: 9 row passes per layer (16
// over SA1's backward: 2.9 GB at the benchmark shape).  All of them walk the same rows.  This
// kernel is gemm_tn_x6_kernel's streaming loop (32 rows per step, operands split into bf16
// planes in LDS, the next rows prefetched into registers) with everything else hung onto it:
//   * dY_l is FORMED while staging -- GM == 1: from the pooled layer's pre-BN output and the
//     (alpha, beta, arg-max) coefficients, exactly as GPOOL; GM == 2: BatchNorm_l's backward
//     applied on the fly, dY = a (m G - w (c1 + xhat c2)) from (G = dZ_l, Y_l) and the finalised
//     sums of layer l (what bn_relu_bwd_apply_kernel wrote in place);
//   * the weight gradient accumulates over the workgroup's row chunk as before (transpose reads);
//   * the SAME dY planes are the A operand of the input gradient: 32 rows x 64 columns per step
//     against W_l^T, whose three planes stay in LDS for the whole kernel.  Rows of a plane are
//     read as rows here and as columns by the transpose reads: with the tr-friendly pitches
//     (320 / 192 B) sixteen row reads would share four bank groups, so the 16-byte units of a
//     row are XOR-swizzled by (row / 4) % 4 -- constant over the four rows of a transpose block,
//     a permutation inside its 64-byte segment, so both access patterns stay conflict-free;
//   * the four waves split the input gradient's reduction in halves, the halves meet in LDS in
//     row layout, and the thread that staged X_{l-1}[row][4 k] (and still holds the raw
//     Y_{l-1} values) adds them, stores dZ_{l-1} as a 16-byte row store and accumulates
//     BatchNorm_{l-1}'s backward sums  s1 += m dz,  s2 += m dz xhat  -- per-workgroup partials,
//     reduced in a fixed order by bn_bwd_finalize_kernel like the stand-alone statistics pass.
// Per layer: read dZ_l (or Y_l), Y_l, Y_{l-1}; write dZ_{l-1}: 4 passes.
// XRC: X_{l-1} = relu(bn(y0)), y0 rebuilt from the 4-column input rows (first-layer recompute).
constexpr int kFusedMaxChunks = 512;
__device__ __forceinline__ int swz(int row, int byte_off) { return byte_off ^ (((row >> 2) & 3) << 4); }

struct FusedArgs {
  const float *G;      // GM 1: pooled layer's pre-BN output Y_l;  GM 2: dZ_l
  const float *Yl;     // GM 2: Y_l (same leading dimension as G)
  int ldg;
  const float *X;      // Y_{l-1} [R][ldx], or X0 [R][4] (XRC)
  int ldx;
  int R, N, K, rows_per_chunk;
  const float *pa, *pb, *mu_p, *is_p;    // layer l-1: scale, shift, mean, invstd
  const float *xw0;                      // XRC: first-layer weight [K][4]
  const float *Wt;     // W_l^T [K][ldw]
  int ldw;
  float *Z;            // out: dZ_{l-1} [R][ldz]
  int ldz;
  float *pw;           // out: weight-gradient partials [chunk][N][K]
  float *spart;        // out: [chunk][2][K] sums for BatchNorm_{l-1}'s backward
  // GM 1 (garg / gdcl: [groups][ldt], this launch's columns from the pointer on)
  const unsigned char *garg;
  const float *gdcl, *galpha, *gbeta;
  int SSH, ldt;
  // GM 2: layer l's scale, shift, mean, invstd and finalised m1, m2
  const float *sc, *sh, *mu, *is, *m1, *m2;
};

// TNW = 2 / 4: n <= 64 / 128, two workgroups per CU.  TNW = 8: n <= 256 (the pooled layers of
// SA2-SA4, the 256-wide point-wise chains) -- every wave owns two n tiles of the weight gradient
// and eight reduction steps of W^T fragments, ~380 VGPRs and 91 KB of LDS: one workgroup per CU.
// Those layers have 8 000 - 115 000 rows; what they gain is the passes and launches they lose.
template <int TNW, int GM, bool XRC>
__global__ __launch_bounds__(256, TNW == 8 ? 1 : 2) void sa_bwd_fused_kernel(FusedArgs a,
                                                                             Compact cm) {
  constexpr int BR = 32;
  constexpr int TN = 32 * TNW;
  // bf16 row pitches: 576 / 320 / 192 B -- (pitch / 4) % 64 in {16, 48}: transpose-read friendly
  constexpr int LG = TN == 256 ? 288 : (TN == 128 ? 160 : 96), LX = 96;
  constexpr int LC = 68;                              // f32 pitch of the dgrad exchange tile
  constexpr int KT = TNW >= 4 ? 2 : 1;                // 32-wide k tiles per wave (weight gradient)
  constexpr int NTW = TNW >= 4 ? TNW / 4 : 1;         // 32-wide n tiles per wave
  constexpr bool GPOOL = GM == 1;
  // compact rows: the row count lives on the device, and so does the split of the rows over
  // the workgroups -- sized by the host's dense bound, a third of the grid would find no rows and
  // the rest would run 1.5x as long (SA1 keeps 67 % of its rows, SA2 44 %)
  int R = a.R, rows_per_chunk = a.rows_per_chunk;
  if (cm.dims) {
    R = cm.dims[0];
    rows_per_chunk = ((R + (int)gridDim.z - 1) / (int)gridDim.z + 31) / 32 * 32;
  }
  const int N = a.N, K = a.K;
  __shared__ __attribute__((aligned(16))) __bf16 Gp[3 * BR * LG];
  __shared__ __attribute__((aligned(16))) __bf16 Xp[3 * BR * LX];
  __shared__ __attribute__((aligned(16))) float Cs[2 * BR * LC];
  __shared__ __attribute__((aligned(16))) int sLr[GPOOL ? 4 * TN : 4];
  __shared__ __attribute__((aligned(16))) float sDv[GPOOL ? 4 * TN : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = TNW >= 4 ? wave * NTW : (wave >> 1);   // first n tile of this wave
  const int wk = TNW >= 4 ? 0 : (wave & 1);
  const int l31 = lane & 31, h = lane >> 5;
  const int k0 = blockIdx.y * 64;
  const int chunk = blockIdx.z;
  const int rbeg = chunk * rows_per_chunk;
  const int rend = min(R, rbeg + rows_per_chunk);
  const float *__restrict__ G = a.G;
  const float *__restrict__ X = a.X;

  // ---- input gradient: wave -> (k half dj, n half dnh) of the step's 32 x 64 tile.  Its B
  // operand -- W_l^T rows k0 + 32 dj + (lane % 32), this lane's 8 n of each 16-wide reduction
  // step of the wave's n half -- never changes: split once, kept in registers for the whole
  // kernel (TN / 32 steps x 3 planes x 4 VGPRs; as LDS planes they cost 52 KB and the second
  // workgroup per CU that hides the load latency of this streaming loop)
  const int dj = wave & 1, dnh = wave >> 1;
  bf16x8 bdr[TN / 32][3];
  {
    const int kr = k0 + dj * 32 + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < TN / 32; ++kk) {
      const int nb = (dnh * (TN / 32) + kk) * 16 + (lane >> 5) * 8;
      float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
      if (kr < K && nb < N) w0 = *reinterpret_cast<const float4 *>(a.Wt + (size_t)kr * a.ldw + nb);
      if (kr < K && nb + 4 < N)
        w1 = *reinterpret_cast<const float4 *>(a.Wt + (size_t)kr * a.ldw + nb + 4);
      const Split4 s0 = split4(w0), s1 = split4(w1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bdr[kk][0][e] = s0.h[e]; bdr[kk][0][4 + e] = s1.h[e];
        bdr[kk][1][e] = s0.m[e]; bdr[kk][1][4 + e] = s1.m[e];
        bdr[kk][2][e] = s0.l[e]; bdr[kk][2][4 + e] = s1.l[e];
      }
    }
  }
  // XRC: the first layer's weight rows of this thread's four k columns, in registers (read
  // from an LDS table [k][4] the sixteen lanes of a row group hit four banks: 36 % of the LDS
  // cycles of this variant were conflicts)
  float4 w0r[XRC ? 4 : 1];
  if constexpr (XRC) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      w0r[e] = (k0 + (tid & 15) * 4 + e < K)
                   ? *reinterpret_cast<const float4 *>(a.xw0 + (size_t)(k0 + (tid & 15) * 4 + e) * 4)
                   : make_float4(0.f, 0.f, 0.f, 0.f);
  }

  f32x16 acc[NTW][KT];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int q = 0; q < KT; ++q)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[t][q][v] = 0.f;

  const int xc4 = (tid & 15) * 4, xr = tid >> 4;
  constexpr int GT = TN / 4, GR = 256 / GT, GPASS = BR / GR;
  const int gc4 = (tid % GT) * 4, gr = tid / GT;
  // layer l-1 per-column coefficients of this thread's four k columns
  float4 fa = make_float4(0.f, 0.f, 0.f, 0.f), fb = fa, fmu = fa, fis = fa;
  if (k0 + xc4 < K) {
    fa = *reinterpret_cast<const float4 *>(a.pa + k0 + xc4);
    fb = *reinterpret_cast<const float4 *>(a.pb + k0 + xc4);
    fmu = *reinterpret_cast<const float4 *>(a.mu_p + k0 + xc4);
    fis = *reinterpret_cast<const float4 *>(a.is_p + k0 + xc4);
  }
  // layer l per-column coefficients of this thread's four n columns
  float4 ga = make_float4(0.f, 0.f, 0.f, 0.f), gb = ga, gA1 = ga, gA2 = ga;
  if (gc4 < N) {
    if (GPOOL) {
      ga = *reinterpret_cast<const float4 *>(a.galpha + gc4);
      gb = *reinterpret_cast<const float4 *>(a.gbeta + gc4);
    } else {
      // dY = a (m G - w (c1 + xhat c2)) = m (a G) - w (A1 + A2 y),  A2 = a c2 invstd,
      // A1 = a c1 - A2 mean;  m = (a y + shift > 0)
      ga = *reinterpret_cast<const float4 *>(a.sc + gc4);
      gb = *reinterpret_cast<const float4 *>(a.sh + gc4);
      const float4 mu = *reinterpret_cast<const float4 *>(a.mu + gc4);
      const float4 is = *reinterpret_cast<const float4 *>(a.is + gc4);
      const float4 c1 = *reinterpret_cast<const float4 *>(a.m1 + gc4);
      const float4 c2 = *reinterpret_cast<const float4 *>(a.m2 + gc4);
      gA2 = make_float4(ga.x * c2.x * is.x, ga.y * c2.y * is.y, ga.z * c2.z * is.z,
                        ga.w * c2.w * is.w);
      gA1 = make_float4(ga.x * c1.x - gA2.x * mu.x, ga.y * c1.y - gA2.y * mu.y,
                        ga.z * c1.z - gA2.z * mu.z, ga.w * c1.w - gA2.w * mu.w);
    }
  }
  float4 rg[GPASS], ry[GM == 2 ? GPASS : 1], rx[2], ykeep[2];
  const int sp_gi = tid / TN, sp_n = tid % TN;
  constexpr int SPQ = 4 * TN / 256;
  // (the entries' loads are kept RAW -- first row of the group relative to the step, arg-max
  // byte, value -- and turned into local rows in put_table(), behind the MFMAs: a comparison
  // formed from a load inside fetch() parks the wave on that load, one L2 round trip per entry
  // and step with nothing in flight to hide it)
  int sp_g[SPQ], sp_ng[SPQ], sp_base[SPQ], sp_arg[SPQ];
  float sp_dv[SPQ];
#pragma unroll
  for (int q = 0; q < SPQ; ++q) {
    sp_g[q] = sp_ng[q] = 0;
    sp_base[q] = -0x40000000;   // (no entry)
    sp_arg[q] = 0;
    sp_dv[q] = 0.f;
  }
  int tb_r0 = 0;   // the step the raw entries belong to
  float gwt[GPASS];
  auto fetch = [&](int r0) {
#pragma unroll
    for (int p = 0; p < GPASS; ++p) {
      const int row = gr + GR * p;
      rg[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (GM == 2) ry[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      gwt[p] = 1.f;
      if (r0 + row < rend && gc4 < N) {
        rg[p] = *reinterpret_cast<const float4 *>(G + (size_t)(r0 + row) * a.ldg + gc4);
        if constexpr (GM == 2)
          ry[p] = *reinterpret_cast<const float4 *>(a.Yl + (size_t)(r0 + row) * a.ldg + gc4);
        if (cm.bw && ((r0 + row) & 7) == 0) gwt[p] = cm.bw[(r0 + row) >> 3];
      }
    }
    if (GPOOL && cm.bgrp) {
      tb_r0 = r0;
#pragma unroll
      for (int q = 0; q < SPQ; ++q) {
        const int slot = sp_gi + (256 / TN) * q, blk = (r0 >> 3) + slot;
        if (r0 != rbeg) sp_g[q] = sp_ng[q];
        sp_base[q] = -0x40000000;
        if (slot < 4 && (blk << 3) < rend && sp_n < N) {
          const int g = sp_g[q];
          sp_base[q] = cm.goff[g] - r0;
          sp_arg[q] = (int)a.garg[(size_t)g * a.ldt + sp_n];
          sp_dv[q] = a.gdcl[(size_t)g * a.ldt + sp_n];
        }
        const int nblk = ((r0 + BR) >> 3) + slot;
        sp_ng[q] = (slot < 4 && (nblk << 3) < rend) ? cm.bgrp[nblk] : 0;
      }
    } else if (GPOOL) {
      tb_r0 = r0;
#pragma unroll
      for (int q = 0; q < SPQ; ++q) {
        const int slot = sp_gi + (256 / TN) * q;
        const int g = (r0 >> a.SSH) + slot;
        sp_base[q] = -0x40000000;
        if ((slot << a.SSH) < BR && (g << a.SSH) < rend && sp_n < N) {
          sp_base[q] = (g << a.SSH) - r0;
          sp_arg[q] = (int)a.garg[(size_t)g * a.ldt + sp_n];
          sp_dv[q] = a.gdcl[(size_t)g * a.ldt + sp_n];
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = xr + 16 * p;
      rx[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + row < rend && k0 + xc4 < K)
        rx[p] = XRC ? *reinterpret_cast<const float4 *>(X + (size_t)(r0 + row) * 4)
                    : *reinterpret_cast<const float4 *>(X + (size_t)(r0 + row) * a.ldx + k0 + xc4);
    }
  };
  if (GPOOL && cm.bgrp && rbeg < rend) {
#pragma unroll
    for (int q = 0; q < SPQ; ++q) {
      const int slot = sp_gi + (256 / TN) * q, blk = (rbeg >> 3) + slot;
      sp_g[q] = (slot < 4 && (blk << 3) < rend) ? cm.bgrp[blk] : 0;
    }
  }
  const int p16 = lane & 15, grp = lane >> 4;
  const int frow = 8 * (grp >> 1) + (p16 >> 2), fcol = 16 * (grp & 1) + 4 * (p16 & 3);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (rbeg < rend) fetch(rbeg);
  // GPOOL: the step's sparse table.  Written for the FIRST step here; for every later step
  // right behind the fetch that loaded its entries, i.e. between the staging barrier and the
  // exchange barrier of the step before (every wave is past its table reads, none has started
  // the next staging) -- no barrier of its own.
  auto put_table = [&]() {
    if constexpr (GPOOL) {
#pragma unroll
      for (int q = 0; q < SPQ; ++q) {
        const int slot = sp_gi + (256 / TN) * q;
        const int lr = sp_base[q] + sp_arg[q];
        const bool hit = cm.bgrp ? (lr >= 0 && (lr >> 3) == slot)
                                 : (lr >= 0 && lr < BR && tb_r0 + lr < rend);
        sLr[slot * TN + sp_n] = hit ? lr : -1;
        sDv[slot * TN + sp_n] = sp_dv[q];
      }
    }
  };
  if constexpr (GPOOL) {
    put_table();
    __syncthreads();
  }
  for (int r0 = rbeg; r0 < rend; r0 += BR) {
#pragma unroll
    for (int p = 0; p < GPASS; ++p) {
      float4 v = rg[p];
      const int row = gr + GR * p;
      if (r0 + row < rend && gc4 < N) {
        const float wr = gwt[p];
        if constexpr (GPOOL) {
          v.x = wr * fmaf(ga.x, v.x, gb.x);
          v.y = wr * fmaf(ga.y, v.y, gb.y);
          v.z = wr * fmaf(ga.z, v.z, gb.z);
          v.w = wr * fmaf(ga.w, v.w, gb.w);
          const int slot = cm.bgrp ? (row >> 3) : (row >> a.SSH);
          const int4 lr = *reinterpret_cast<const int4 *>(&sLr[slot * TN + gc4]);
          const float4 dv = *reinterpret_cast<const float4 *>(&sDv[slot * TN + gc4]);
          v.x += lr.x == row ? dv.x : 0.f;
          v.y += lr.y == row ? dv.y : 0.f;
          v.z += lr.z == row ? dv.z : 0.f;
          v.w += lr.w == row ? dv.w : 0.f;
        } else if constexpr (GM == 2) {
          const float4 y = ry[p];
          v.x = fmaf(-wr, fmaf(gA2.x, y.x, gA1.x), fmaf(ga.x, y.x, gb.x) > 0.f ? ga.x * v.x : 0.f);
          v.y = fmaf(-wr, fmaf(gA2.y, y.y, gA1.y), fmaf(ga.y, y.y, gb.y) > 0.f ? ga.y * v.y : 0.f);
          v.z = fmaf(-wr, fmaf(gA2.z, y.z, gA1.z), fmaf(ga.z, y.z, gb.z) > 0.f ? ga.z * v.z : 0.f);
          v.w = fmaf(-wr, fmaf(gA2.w, y.w, gA1.w), fmaf(ga.w, y.w, gb.w) > 0.f ? ga.w * v.w : 0.f);
        }
      } else {
        v = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const Split4 sp = split4(v);
      const int at = row * LG + (swz(row, gc4 * 2) >> 1);
      *reinterpret_cast<bf16x4 *>(&Gp[0 * BR * LG + at]) = sp.h;
      *reinterpret_cast<bf16x4 *>(&Gp[1 * BR * LG + at]) = sp.m;
      *reinterpret_cast<bf16x4 *>(&Gp[2 * BR * LG + at]) = sp.l;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = xr + 16 * p;
      float4 x = rx[p];
      const bool live = r0 + row < rend && k0 + xc4 < K;
      if constexpr (XRC) {
        if (live)
          x = make_float4(rc_dot4(x, w0r[0]), rc_dot4(x, w0r[1]), rc_dot4(x, w0r[2]),
                          rc_dot4(x, w0r[3]));
      }
      ykeep[p] = x;   // the raw pre-BN values of layer l-1: the epilogue's mask and xhat
      if (live) {
        x.x = fmaxf(fmaf(fa.x, x.x, fb.x), 0.f);
        x.y = fmaxf(fmaf(fa.y, x.y, fb.y), 0.f);
        x.z = fmaxf(fmaf(fa.z, x.z, fb.z), 0.f);
        x.w = fmaxf(fmaf(fa.w, x.w, fb.w), 0.f);
      } else {
        x = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const Split4 sp = split4(x);
      const int at = row * LX + xc4;
      *reinterpret_cast<bf16x4 *>(&Xp[0 * BR * LX + at]) = sp.h;
      *reinterpret_cast<bf16x4 *>(&Xp[1 * BR * LX + at]) = sp.m;
      *reinterpret_cast<bf16x4 *>(&Xp[2 * BR * LX + at]) = sp.l;
    }
    __syncthreads();
    if (r0 + BR < rend) fetch(r0 + BR);  // next rows in flight during the MFMAs
    // ---- weight gradient: reduction over the 32 rows (transpose reads of both operands)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[3][NTW], bf[3][KT];
      const int trow = ks * 16 + frow;   // (+4 for the second half: same swizzle class + 1)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          typedef __attribute__((address_space(3))) s16x4 *lds_ptr;
          const int c2 = ((wn + t) * 32 + fcol) * 2;
          const __bf16 *b0 = &Gp[(q * BR + trow) * LG + (swz(trow, c2) >> 1)];
          const __bf16 *b1 = &Gp[(q * BR + trow + 4) * LG + (swz(trow + 4, c2) >> 1)];
          union {
            s16x4 s[2];
            bf16x8 b;
          } u;
          u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(b0));
          u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(b1));
          af[q][t] = u.b;
        }
#pragma unroll
        for (int t = 0; t < KT; ++t)
          bf[q][t] = tr_read8(&Xp[(q * BR + trow) * LX + (wk + t) * 32 + fcol], LX);
      }
#define BTR_X6(QA, QB)                                                                       \
  _Pragma("unroll") for (int u = 0; u < NTW; ++u) _Pragma("unroll") for (int t = 0; t < KT; ++t) \
      acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[QA][u], bf[QB][t], acc[u][t], 0, 0, 0);
      BTR_X6(2, 0)
      BTR_X6(0, 2)
      BTR_X6(1, 1)
      BTR_X6(1, 0)
      BTR_X6(0, 1)
      BTR_X6(0, 0)
#undef BTR_X6
    }
    // ---- input gradient: C[32 rows][32 k of half dj] over the n half dnh (row reads)
    {
      f32x16 cd;
#pragma unroll
      for (int v = 0; v < 16; ++v) cd[v] = 0.f;
#pragma unroll
      for (int kk = 0; kk < TN / 32; ++kk) {
        const int nb = (dnh * (TN / 32) + kk) * 16 + h * 8;   // first of this lane's 8 n
        bf16x8 ad[3];
        const bf16x8 *bd = bdr[kk];
#pragma unroll
        for (int q = 0; q < 3; ++q)
          ad[q] = *reinterpret_cast<const bf16x8 *>(
              &Gp[(q * BR + l31) * LG + (swz(l31, nb * 2) >> 1)]);
#define BTR_X6D(QA, QB) cd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad[QA], bd[QB], cd, 0, 0, 0);
        BTR_X6D(2, 0)
        BTR_X6D(0, 2)
        BTR_X6D(1, 1)
        BTR_X6D(1, 0)
        BTR_X6D(0, 1)
        BTR_X6D(0, 0)
#undef BTR_X6D
      }
      float *T = Cs + dnh * (BR * LC);
#pragma unroll
      for (int v = 0; v < 16; ++v)
        T[((v & 3) + 8 * (v >> 2) + 4 * h) * LC + dj * 32 + l31] = cd[v];
    }
    if (r0 + BR < rend) put_table();   // (the entries the fetch above loaded)
    __syncthreads();
    // ---- epilogue: the thread that staged X[row][xc4..] owns dZ[row][k0 + xc4..]
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = xr + 16 * p;
      if (r0 + row < rend && k0 + xc4 < K) {
        const float4 c0 = *reinterpret_cast<const float4 *>(&Cs[row * LC + xc4]);
        const float4 c1 = *reinterpret_cast<const float4 *>(&Cs[BR * LC + row * LC + xc4]);
        const float4 c = make_float4(c0.x + c1.x, c0.y + c1.y, c0.z + c1.z, c0.w + c1.w);
        float *zp = a.Z + (size_t)(r0 + row) * a.ldz + k0 + xc4;
        *reinterpret_cast<float4 *>(zp) = c;
        const float4 y = ykeep[p];
        const float gx = fmaf(fa.x, y.x, fb.x) > 0.f ? c.x : 0.f;
        const float gy = fmaf(fa.y, y.y, fb.y) > 0.f ? c.y : 0.f;
        const float gz = fmaf(fa.z, y.z, fb.z) > 0.f ? c.z : 0.f;
        const float gw = fmaf(fa.w, y.w, fb.w) > 0.f ? c.w : 0.f;
        s1.x += gx; s1.y += gy; s1.z += gz; s1.w += gw;
        s2.x = fmaf(gx, (y.x - fmu.x) * fis.x, s2.x);
        s2.y = fmaf(gy, (y.y - fmu.y) * fis.y, s2.y);
        s2.z = fmaf(gz, (y.z - fmu.z) * fis.z, s2.z);
        s2.w = fmaf(gw, (y.w - fmu.w) * fis.w, s2.w);
      }
    }
    // (the next step's staging writes Gp / Xp: every wave is past its MFMA reads -- the barrier
    // above -- and Cs is rewritten only after the next staging barrier)
  }
  // ---- weight-gradient partial of this chunk
  float *out = a.pw + (size_t)chunk * N * K;
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int q = 0; q < KT; ++q) {
      const int col = k0 + (wk + q) * 32 + l31;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int row = (wn + t) * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
        if (row < N && col < K) out[(size_t)row * K + col] = acc[t][q][v];
      }
    }
  // ---- BatchNorm_{l-1} sums of this chunk: 16 row-threads per k column group, fixed order
  __syncthreads();
  float *red = Cs;   // [2][16 row threads][64 k]
  *reinterpret_cast<float4 *>(&red[(0 * 16 + xr) * 64 + xc4]) = s1;
  *reinterpret_cast<float4 *>(&red[(1 * 16 + xr) * 64 + xc4]) = s2;
  __syncthreads();
  if (tid < 128) {
    const int which = tid >> 6, c = tid & 63;
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[(which * 16 + q) * 64 + c];
    if (k0 + c < K) a.spart[((size_t)chunk * 2 + which) * K + k0 + c] = t;
  }
}

// ---- the pooled layer's backward in Gram form --------------------------------------------------
// sa_bwd_fused_kernel<., 1, .> reads the pooled layer's whole pre-BN output Y_l [rows][n] -- the
// largest tensor of a set-abstraction level, twice as wide as its input -- to form
//   dY[r][n] = w_r (alpha[n] Y[r][n] + beta[n]) + sparse[r][n]
// (sparse: dcl[g][n] on the arg-max row of (group g, channel n); w_r: copies a compact row stands
// for).  The dense part is AFFINE in Y_l = X W^T, X = relu(bn(Y_{l-1})) [rows][k], so it never
// has to be formed per element:
//   dZ_{l-1}[r] = dY[r] W        = w_r (x_r M + c) + sparse[r] W,   M = W^T diag(alpha) W [k][k],
//                                                                   c = beta^T W [k]
//   dW_l        = dY^T X         = diag(alpha) W G + beta (x) sx + sparse^T X,
//                                  G = sum_r w_r x_r x_r^T [k][k],  sx = sum_r w_r x_r [k]
// Y_l is not read -- and therefore not written by the forward either (sa_fwd_stream_kernel with
// C == NULL: its pooling epilogue keeps the group extrema; the arg-max row's pre-BN value, the
// one thing the pooled BatchNorm backward needs of Y_l, is left by the pool kernel: `ywin`).
// Per row the kernel reads X (k floats) and writes dZ (k floats): 512 B at SA1's 128 x 64 layer
// instead of 1 024 B here plus 512 B of Y_l written by the forward; the dense products shrink
// from n x k to k x k per row.  One streaming pass of 32-row steps, laid out like
// sa_bwd_fused_kernel (bf16x6 planes in LDS, the next rows in flight during the MFMAs):
//   * X planes hold the layer's WHOLE k extent (k <= 128), swizzled for row and transpose reads;
//     the first row of a compact group (the only row with w != 1) has a second, weighted copy in
//     four side rows that the B operand of G and the A operand of x M select per lane;
//   * the sparse operand is a plane tile that stays zero: the thread that fetched entry (slot,
//     n) of the step writes its three bf16 pieces, and clears them again behind the MFMAs;
//   * a wave accumulates  sparse^T X  (its n tiles x the workgroup's 64 k),  G  (its share of the
//     k x 64 block) and, in ONE accumulator, its half of  sparse W  and of  (w x) M;
//   * epilogue as in sa_bwd_fused_kernel: the halves meet in LDS, + w_r c, dZ_{l-1} leaves as
//     16-byte row stores and BatchNorm_{l-1}'s backward sums accumulate on the way.
// gram_reduce_kernel / gram_finish_kernel turn the per-chunk partials into dW_l (float64 for the
// dense part: G and sx are sums of ~10^5..10^6 terms whose combination with alpha / beta cancels).
struct GramArgs {
  const float *X;      // Y_{l-1} [R][ldx]
  int ldx;
  int R, N, K, rows_per_chunk;
  const float *pa, *pb, *mu_p, *is_p;    // layer l-1: scale, shift, mean, invstd
  const float *Wt;     // W_l^T [K][ldw]
  int ldw;
  const float *M;      // [K][K] = W^T diag(alpha) W (symmetric)
  const float *cvec;   // [K] = beta^T W
  float *Z;            // out: dZ_{l-1} [R][ldz]
  int ldz;
  float *pw;           // out: [chunk][N][K]  sparse^T X
  float *gp;           // out: [chunk][K][K]  G partial
  float *sxp;          // out: [chunk][K]     sx partial
  float *spart;        // out: [chunk][2][K]  sums for BatchNorm_{l-1}'s backward
  const unsigned char *garg;
  const float *gdcl;
  int SSH, ldt;
};


// 8 consecutive reduction indices of one column out of a swizzled row-major plane (transpose
// read; lane -> (row trow (+4), 4-column chunk at col), see sa_bwd_fused_kernel)
__device__ __forceinline__ bf16x8 tr_read8_swz(const __bf16 *plane, int pitch, int trow, int col) {
  typedef __attribute__((address_space(3))) s16x4 *lds_ptr;
  const __bf16 *b0 = &plane[trow * pitch + (swz(trow, col * 2) >> 1)];
  const __bf16 *b1 = &plane[(trow + 4) * pitch + (swz(trow + 4, col * 2) >> 1)];
  union {
    s16x4 s[2];
    bf16x8 b;
  } u;
  u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(b0));
  u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(b1));
  return u.b;
}

// (A first, single-role form of the kernel -- 256 threads, every wave staging, multiplying and
// finishing in turn, n <= 256 and k <= 128 -- equalled sa_bwd_fused_kernel at every shape: SA1's
// pooled layer 266 us in Gram form against 270 us reading twice as much, 198 / 133 / 92 us against
// 185 / 114 / 70 us at SA2 - SA4's 256 x 128, where its helper launches made it a net loss
// (profiles/r05_h_bwd_gram_ab.txt).  Removed in round 6; the form below is the one that ships.)

// ---- ... with producer and consumer waves -------------------------------------------------------
// The single-role form, like sa_bwd_fused_kernel, ran its three phases -- staging (VALU),
// products (LDS reads + MFMA), epilogue (VALU + stores) -- one after the other in every wave, two
// barriers per 32-row step, and leaves it to the second workgroup of the CU to fill the gaps:
// both forms need ~12 500 cycles per pair of steps for 4 600 cycles of MFMA issue, whatever the
// bytes (SA1's pooled layer: 266 us in Gram form against 270 us reading twice as much).  Here the
// phases run BESIDE each other: a 512-thread workgroup (one per CU) whose waves 0-3 only stage and
// finish rows and whose waves 4-7 only multiply, one of each kind per SIMD, so that a SIMD's
// VALU work and its matrix work come from different waves by construction.
//   iteration i:  producers   epilogue of step i-1 (C tile of buffer (i-1) % 2: dZ rows out, the
//                             BatchNorm sums), staging of step i+1 into plane buffer (i+1) % 2,
//                             global loads of step i+3 (two steps ahead of their use)
//                 consumers   products of step i out of plane buffer i % 2 into C tile i % 2
//                 ONE barrier
// The planes, the weighted side rows and the C tile are double-buffered (137 KB at n = 128); the
// producers keep the raw rows of the two steps in flight in registers for the epilogue.
// n <= 128, k <= 64 (SA1's pooled layer: a third of all set-abstraction rows of the step).
constexpr int kGramWsMaxBlocks = 2560;   // 8-row blocks per chunk (20 KB LDS table): 20 480 rows
template <int TNW>
__global__ __launch_bounds__(512, 1) void sa_bwd_gram_ws_kernel(GramArgs a, Compact cm) {
  constexpr int BR = 32, KF = 64;
  constexpr int TN = 32 * TNW;
  constexpr int LG = TN == 128 ? 160 : 96;
  constexpr int LXF = 96, LC = 68;
  constexpr int XR = BR + 4;   // X plane rows: 32 + the weighted copies of rows 0, 8, 16, 24
  constexpr int KT = TNW >= 4 ? 2 : 1;
  constexpr int SPB = 3 * BR * LG, XPB = 3 * XR * LXF, CSB = 2 * BR * LC;
  int R = a.R, rows_per_chunk = a.rows_per_chunk;
  if (cm.dims) {
    R = cm.dims[0];
    rows_per_chunk = ((R + (int)gridDim.z - 1) / (int)gridDim.z + 31) / 32 * 32;
  }
  const int N = a.N, K = a.K;
  __shared__ __attribute__((aligned(16))) __bf16 Sp[2 * SPB];
  __shared__ __attribute__((aligned(16))) __bf16 Xp[2 * XPB];
  __shared__ __attribute__((aligned(16))) float Cs[2 * CSB];
  __shared__ int2 Bg[kGramWsMaxBlocks];   // per 8-row block of the chunk: (group, its first row)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = blockIdx.z;
  const int rbeg = chunk * rows_per_chunk;
  const int rend = min(R, rbeg + rows_per_chunk);
  const int nsteps = rbeg < rend ? (rend - rbeg + BR - 1) / BR : 0;
  const int nsteps2 = (nsteps + 1) & ~1;   // (pairs: the buffers alternate with static indices)
  const bool cmw = cm.bw != nullptr;

  for (int i = tid; i < 2 * SPB / 8; i += 512)
    reinterpret_cast<float4 *>(Sp)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  // compact rows: group of every 8-row block of this chunk, first row of every group it touches
  // (the chunk's blocks are consecutive, and so are their groups)
  if (cm.bgrp && rbeg < rend) {
    const int nblk = (rend - rbeg + 7) >> 3;
    for (int i = tid; i < nblk; i += 512) {
      const int g = cm.bgrp[(rbeg >> 3) + i];
      Bg[i] = make_int2(g, cm.goff[g]);
    }
  }
  __syncthreads();

  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, sxa = s1;
  const int xc4 = (tid & 15) * 4, xr = (tid & 255) >> 4;
  const std::integral_constant<int, 0> P0{};
  const std::integral_constant<int, 1> P1{};
  if (wave < 4) {
    // ================================================================== producers
    const float *__restrict__ X = a.X;
    float4 fa = make_float4(0.f, 0.f, 0.f, 0.f), fb = fa, fmu = fa, fis = fa, cv = fa;
    if (xc4 < K) {
      fa = *reinterpret_cast<const float4 *>(a.pa + xc4);
      fb = *reinterpret_cast<const float4 *>(a.pb + xc4);
      fmu = *reinterpret_cast<const float4 *>(a.mu_p + xc4);
      fis = *reinterpret_cast<const float4 *>(a.is_p + xc4);
      cv = *reinterpret_cast<const float4 *>(a.cvec + xc4);
    }
    const int sp_gi = tid / TN, sp_n = tid % TN;
    constexpr int SPQ = 4 * TN / 256;
    float4 rx[2][2], yk[2][2];
    float wx[2][2], wk[2][2];
    // sparse entries: what the loads return is kept RAW (arg-max byte, value, the group's first
    // row) and turned into a local row only when the step is staged, two iterations later -- an
    // address or a comparison formed from a load inside fetch() would park the wave on that load
    // (and on every load in front of it) once per step
    int sp_arg[2][SPQ], sp_base[2][SPQ], cl_lr[2][SPQ];
    float sp_dv[2][SPQ];
#pragma unroll
    for (int q = 0; q < SPQ; ++q) {
      sp_arg[0][q] = sp_arg[1][q] = 0;
      sp_base[0][q] = sp_base[1][q] = -0x40000000;   // (no entry)
      cl_lr[0][q] = cl_lr[1][q] = -1;
      sp_dv[0][q] = sp_dv[1][q] = 0.f;
    }
    // loads of the step at r0 into register set P
    auto fetch = [&](auto P, int r0) {
      constexpr int p = decltype(P)::value;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = xr + 16 * j;
        wx[p][j] = 1.f;
        rx[p][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + row < rend) {
          if (cmw && (row & 7) == 0) wx[p][j] = cm.bw[(r0 + row) >> 3];
          if (xc4 < K)
            rx[p][j] = *reinterpret_cast<const float4 *>(X + (size_t)(r0 + row) * a.ldx + xc4);
        }
      }
#pragma unroll
      for (int q = 0; q < SPQ; ++q) {
        const int slot = sp_gi + (256 / TN) * q;
        sp_base[p][q] = -0x40000000;
        if (cm.bgrp) {
          // (group of the block and its first row: LDS tables of this chunk, filled at the start)
          const int blk = (r0 >> 3) + slot;
          if (slot < 4 && (blk << 3) < rend && sp_n < N) {
            const int2 gg = Bg[blk - (rbeg >> 3)];
            const int g = gg.x;
            sp_base[p][q] = gg.y - r0;
            sp_arg[p][q] = (int)a.garg[(size_t)g * a.ldt + sp_n];
            sp_dv[p][q] = a.gdcl[(size_t)g * a.ldt + sp_n];
          }
        } else {
          const int g = (r0 >> a.SSH) + slot;
          if ((slot << a.SSH) < BR && (g << a.SSH) < rend && sp_n < N) {
            sp_base[p][q] = (g << a.SSH) - r0;
            sp_arg[p][q] = (int)a.garg[(size_t)g * a.ldt + sp_n];
            sp_dv[p][q] = a.gdcl[(size_t)g * a.ldt + sp_n];
          }
        }
      }
    };
    // local row of entry q of register set p (-1: none in this step)
    auto entry_row = [&](int r0, int slot, int base, int arg) {
      const int lr = base + arg;
      if (cm.bgrp) return (lr >= 0 && (lr >> 3) == slot) ? lr : -1;
      return (lr >= 0 && lr < BR && r0 + lr < rend) ? lr : -1;
    };
    // register set P -> plane buffer P (the step's sparse entries replace those of the step before
    // last; X = relu(bn(.)) planes incl. the weighted copies of rows 0 (mod 8), column sums)
    auto stage = [&](auto P, int r0) {
      constexpr int p = decltype(P)::value;
      __bf16 *sp = Sp + p * SPB, *xp = Xp + p * XPB;
#pragma unroll
      for (int q = 0; q < SPQ; ++q) {
        if (cl_lr[p][q] >= 0) {
          const int at = cl_lr[p][q] * LG + (swz(cl_lr[p][q], sp_n * 2) >> 1);
          const __bf16 z = (__bf16)0.f;
          sp[0 * BR * LG + at] = z;
          sp[1 * BR * LG + at] = z;
          sp[2 * BR * LG + at] = z;
        }
        const int lr = entry_row(r0, sp_gi + (256 / TN) * q, sp_base[p][q], sp_arg[p][q]);
        cl_lr[p][q] = lr;
        if (lr >= 0) {
          __bf16 eh, em, el;
          split1(sp_dv[p][q], eh, em, el);
          const int at = lr * LG + (swz(lr, sp_n * 2) >> 1);
          sp[0 * BR * LG + at] = eh;
          sp[1 * BR * LG + at] = em;
          sp[2 * BR * LG + at] = el;
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = xr + 16 * j;
        float4 x = rx[p][j];
        const bool live = r0 + row < rend && xc4 < K;
        yk[p][j] = x;
        wk[p][j] = wx[p][j];
        if (live) {
          x.x = fmaxf(fmaf(fa.x, x.x, fb.x), 0.f);
          x.y = fmaxf(fmaf(fa.y, x.y, fb.y), 0.f);
          x.z = fmaxf(fmaf(fa.z, x.z, fb.z), 0.f);
          x.w = fmaxf(fmaf(fa.w, x.w, fb.w), 0.f);
        } else {
          x = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const Split4 spl = split4(x);
        const int at = row * LXF + (swz(row, xc4 * 2) >> 1);
        *reinterpret_cast<bf16x4 *>(&xp[0 * XR * LXF + at]) = spl.h;
        *reinterpret_cast<bf16x4 *>(&xp[1 * XR * LXF + at]) = spl.m;
        *reinterpret_cast<bf16x4 *>(&xp[2 * XR * LXF + at]) = spl.l;
        const float w = wx[p][j];
        sxa.x = fmaf(w, x.x, sxa.x); sxa.y = fmaf(w, x.y, sxa.y);
        sxa.z = fmaf(w, x.z, sxa.z); sxa.w = fmaf(w, x.w, sxa.w);
        if ((row & 7) == 0) {   // (dense rows: the copy is the row itself -- w = 1)
          const Split4 sw = split4(make_float4(w * x.x, w * x.y, w * x.z, w * x.w));
          const int aw = (BR + (row >> 3)) * LXF + xc4;   // rows 32 .. 35: (row >> 2) & 3 == 0
          *reinterpret_cast<bf16x4 *>(&xp[0 * XR * LXF + aw]) = sw.h;
          *reinterpret_cast<bf16x4 *>(&xp[1 * XR * LXF + aw]) = sw.m;
          *reinterpret_cast<bf16x4 *>(&xp[2 * XR * LXF + aw]) = sw.l;
        }
      }
    };
    // C tile P -> dZ rows of the step at r0, BatchNorm_{l-1}'s sums
    auto epilogue = [&](auto P, int r0) {
      constexpr int p = decltype(P)::value;
      const float *cs = Cs + p * CSB;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = xr + 16 * j;
        if (r0 + row < rend && xc4 < K) {
          const float wr = wk[p][j];
          const float4 c0 = *reinterpret_cast<const float4 *>(&cs[row * LC + xc4]);
          const float4 c1 = *reinterpret_cast<const float4 *>(&cs[BR * LC + row * LC + xc4]);
          const float4 c = make_float4(fmaf(wr, cv.x, c0.x + c1.x), fmaf(wr, cv.y, c0.y + c1.y),
                                       fmaf(wr, cv.z, c0.z + c1.z), fmaf(wr, cv.w, c0.w + c1.w));
          *reinterpret_cast<float4 *>(a.Z + (size_t)(r0 + row) * a.ldz + xc4) = c;
          const float4 y = yk[p][j];
          const float gx = fmaf(fa.x, y.x, fb.x) > 0.f ? c.x : 0.f;
          const float gy = fmaf(fa.y, y.y, fb.y) > 0.f ? c.y : 0.f;
          const float gz = fmaf(fa.z, y.z, fb.z) > 0.f ? c.z : 0.f;
          const float gw = fmaf(fa.w, y.w, fb.w) > 0.f ? c.w : 0.f;
          s1.x += gx; s1.y += gy; s1.z += gz; s1.w += gw;
          s2.x = fmaf(gx, (y.x - fmu.x) * fis.x, s2.x);
          s2.y = fmaf(gy, (y.y - fmu.y) * fis.y, s2.y);
          s2.z = fmaf(gz, (y.z - fmu.z) * fis.z, s2.z);
          s2.w = fmaf(gw, (y.w - fmu.w) * fis.w, s2.w);
        }
      }
    };
    fetch(P0, rbeg);
    fetch(P1, rbeg + BR);
    stage(P0, rbeg);
    fetch(P0, rbeg + 2 * BR);
    __syncthreads();
    for (int i = 0; i < nsteps2; i += 2) {
      const int r0 = rbeg + i * BR;
      // iteration i (even): consumers work on buffer 0
      if (i >= 1) epilogue(P1, r0 - BR);
      stage(P1, r0 + BR);
      fetch(P1, r0 + 3 * BR);
      __syncthreads();
      // iteration i + 1: consumers work on buffer 1
      epilogue(P0, r0);
      stage(P0, r0 + 2 * BR);
      fetch(P0, r0 + 4 * BR);
      __syncthreads();
    }
    if (nsteps2 >= 1) epilogue(P1, rbeg + (nsteps2 - 1) * BR);
  } else {
    // ================================================================== consumers
    const int mw = wave - 4;
    const int wn = TNW >= 4 ? mw : (mw >> 1);
    const int wk_ = TNW >= 4 ? 0 : (mw & 1);
    const int l31 = lane & 31, h = lane >> 5;
    const int dj = mw & 1, dnh = mw >> 1;
    bf16x8 bdr[TN / 32][3], mdr[KF / 32][3];
    {
      const int kr = dj * 32 + (lane & 31);
#pragma unroll
      for (int kk = 0; kk < TN / 32; ++kk) {
        const int nb = (dnh * (TN / 32) + kk) * 16 + (lane >> 5) * 8;
        float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
        if (kr < K && nb < N) w0 = *reinterpret_cast<const float4 *>(a.Wt + (size_t)kr * a.ldw + nb);
        if (kr < K && nb + 4 < N)
          w1 = *reinterpret_cast<const float4 *>(a.Wt + (size_t)kr * a.ldw + nb + 4);
        const Split4 t0 = split4(w0), t1 = split4(w1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bdr[kk][0][e] = t0.h[e]; bdr[kk][0][4 + e] = t1.h[e];
          bdr[kk][1][e] = t0.m[e]; bdr[kk][1][4 + e] = t1.m[e];
          bdr[kk][2][e] = t0.l[e]; bdr[kk][2][4 + e] = t1.l[e];
        }
      }
#pragma unroll
      for (int kk = 0; kk < KF / 32; ++kk) {
        const int xb = (dnh * (KF / 32) + kk) * 16 + (lane >> 5) * 8;
        float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
        if (kr < K && xb < K) w0 = *reinterpret_cast<const float4 *>(a.M + (size_t)kr * K + xb);
        if (kr < K && xb + 4 < K)
          w1 = *reinterpret_cast<const float4 *>(a.M + (size_t)kr * K + xb + 4);
        const Split4 t0 = split4(w0), t1 = split4(w1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          mdr[kk][0][e] = t0.h[e]; mdr[kk][0][4 + e] = t1.h[e];
          mdr[kk][1][e] = t0.m[e]; mdr[kk][1][4 + e] = t1.m[e];
          mdr[kk][2][e] = t0.l[e]; mdr[kk][2][4 + e] = t1.l[e];
        }
      }
    }
    f32x16 acc[KT], gacc;
#pragma unroll
    for (int q = 0; q < KT; ++q)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[q][v] = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) gacc[v] = 0.f;
    // ---- every LDS address of the loop = one of these lane constants + a compile-time offset
    // (plane, buffer, ks, k tile: the swizzle only touches bits 4-5 of the byte offset and is the
    // same for rows 16 apart).  Element (bf16) offsets inside buffer 0:
    const int p16 = lane & 15, grp = lane >> 4;
    const int frow = 8 * (grp >> 1) + (p16 >> 2), fcol = 16 * (grp & 1) + 4 * (p16 & 3);
    const bool wl_tr = cmw && (p16 >> 2) == 0;   // this lane's first transpose row is 0 (mod 8)
    const bool wl_row = cmw && (l31 & 7) == 0;
    const int oS0 = frow * LG + (swz(frow, fcol * 2) >> 1) + wn * 32;        // Sp, rows frow ..
    const int oS1 = (frow + 4) * LG + (swz(frow + 4, fcol * 2) >> 1) + wn * 32;
    const int oX0 = frow * LXF + (swz(frow, fcol * 2) >> 1);                  // Xp, columns fcol ..
    const int oX1 = (frow + 4) * LXF + (swz(frow + 4, fcol * 2) >> 1);
    const int oB = wk_ * 32, oGa = (mw >> 1) * 32, oGb = (mw & 1) * 32;       // column tiles
    // G's B operand: the weighted copy (rows 32 + trow / 8, unswizzled) where trow = 0 (mod 8)
    const int oW0 = wl_tr ? (BR + (frow >> 3)) * LXF + fcol + oGb : oX0 + oGb;            // ks = 0
    const int oW1 = wl_tr ? (BR + 2 + (frow >> 3)) * LXF + fcol + oGb : oX0 + 16 * LXF + oGb;
    int oDs[TN / 32], oDx[KF / 32];   // row reads of the input-gradient products
#pragma unroll
    for (int kk = 0; kk < TN / 32; ++kk) {
      const int nb = (dnh * (TN / 32) + kk) * 16 + h * 8;
      oDs[kk] = l31 * LG + (swz(l31, nb * 2) >> 1);
    }
#pragma unroll
    for (int kk = 0; kk < KF / 32; ++kk) {
      const int xb = (dnh * (KF / 32) + kk) * 16 + h * 8;
      oDx[kk] = wl_row ? (BR + (l31 >> 3)) * LXF + xb : l31 * LXF + (swz(l31, xb * 2) >> 1);
    }
    const int oC = dnh * (BR * LC) + 4 * h * LC + dj * 32 + l31;
    typedef __attribute__((address_space(3))) s16x4 *lds_ptr;
    auto tr2 = [&](const __bf16 *p0, const __bf16 *p1) {
      union {
        s16x4 s[2];
        bf16x8 b;
      } u;
      u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p0));
      u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p1));
      return u.b;
    };
    // The step's products as a chain of groups -- fragments out of LDS, then six (twelve) MFMAs --
    // with a group's fragments requested before the MFMAs of the group in front of it:
    //   groups 0 .. 3: (ks, what) = (0, sparse^T X), (0, G), (1, sparse^T X), (1, G)
    //   groups 4 .. 4 + TN/32 - 1: sparse W, one 16-wide n step each;  then KF/32 groups (w x) M
    struct Frag {
      bf16x8 a[3], b[KT][3];
    };
    constexpr int NG = 4 + TN / 32 + KF / 32;
#define BTR_X6G(ACC, A, B)                                                    \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A)[2], (B)[0], ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A)[0], (B)[2], ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A)[1], (B)[1], ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A)[1], (B)[0], ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A)[0], (B)[1], ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A)[0], (B)[0], ACC, 0, 0, 0);
    auto step = [&](auto PB) {
      constexpr int pb = decltype(PB)::value;
      const __bf16 *sp = Sp + pb * SPB, *xp = Xp + pb * XPB;
      f32x16 cd;
#pragma unroll
      for (int v = 0; v < 16; ++v) cd[v] = 0.f;
      auto load = [&](int g, Frag &f) {
        if (g < 4) {
          const int ro = (g >> 1) * 16;   // ks * 16 rows
          if ((g & 1) == 0) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              f.a[q] = tr2(sp + q * BR * LG + ro * LG + oS0, sp + q * BR * LG + ro * LG + oS1);
#pragma unroll
              for (int t = 0; t < KT; ++t)
                f.b[t][q] = tr2(xp + q * XR * LXF + ro * LXF + t * 32 + oB + oX0,
                                xp + q * XR * LXF + ro * LXF + t * 32 + oB + oX1);
            }
          } else {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              f.a[q] = tr2(xp + q * XR * LXF + ro * LXF + oGa + oX0,
                           xp + q * XR * LXF + ro * LXF + oGa + oX1);
              f.b[0][q] = tr2(xp + q * XR * LXF + ((g >> 1) ? oW1 : oW0),
                              xp + q * XR * LXF + ro * LXF + oGb + oX1);
            }
          }
        } else if (g < 4 + TN / 32) {
#pragma unroll
          for (int q = 0; q < 3; ++q)
            f.a[q] = *reinterpret_cast<const bf16x8 *>(sp + q * BR * LG + oDs[g - 4]);
        } else {
#pragma unroll
          for (int q = 0; q < 3; ++q)
            f.a[q] = *reinterpret_cast<const bf16x8 *>(xp + q * XR * LXF + oDx[g - 4 - TN / 32]);
        }
      };
      auto mma = [&](int g, const Frag &f) {
        if (g < 4) {
          if ((g & 1) == 0) {
#pragma unroll
            for (int t = 0; t < KT; ++t) { BTR_X6G(acc[t], f.a, f.b[t]) }
          } else {
            BTR_X6G(gacc, f.a, f.b[0])
          }
        } else if (g < 4 + TN / 32) {
          BTR_X6G(cd, f.a, bdr[g - 4])
        } else {
          BTR_X6G(cd, f.a, mdr[g - 4 - TN / 32])
        }
      };
      Frag fr[2];
      load(0, fr[0]);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        // (scheduling fences: left alone, the compiler sinks a group's LDS reads down to the MFMAs
        // that consume them and the wave waits out an LDS round trip per group)
        if (g + 1 < NG) load(g + 1, fr[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        mma(g, fr[g & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      float *T = Cs + pb * CSB + oC;
#pragma unroll
      for (int v = 0; v < 16; ++v) T[((v & 3) + 8 * (v >> 2)) * LC] = cd[v];
    };
    __syncthreads();   // (the producers' first planes)
    for (int i = 0; i < nsteps2; i += 2) {
      step(P0);
      __syncthreads();
      step(P1);
      __syncthreads();
    }
#undef BTR_X6G
    // ---- partials of this chunk: sparse^T X, G
    {
      float *out = a.pw + (size_t)chunk * N * K;
#pragma unroll
      for (int q = 0; q < KT; ++q) {
        const int col = (wk_ + q) * 32 + l31;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int row = wn * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
          if (row < N && col < K) out[(size_t)row * K + col] = acc[q][v];
        }
      }
      float *og = a.gp + (size_t)chunk * K * K;
      const int col = (mw & 1) * 32 + l31;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int row = (mw >> 1) * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
        if (row < K && col < K) og[(size_t)row * K + col] = gacc[v];
      }
    }
  }
  // ---- the producers' column sums: 16 row-threads per k column group, fixed order
  __syncthreads();
  float *red = Cs;   // [3][16 row threads][64 k]
  if (wave < 4) {
    *reinterpret_cast<float4 *>(&red[(0 * 16 + xr) * 64 + xc4]) = s1;
    *reinterpret_cast<float4 *>(&red[(1 * 16 + xr) * 64 + xc4]) = s2;
    *reinterpret_cast<float4 *>(&red[(2 * 16 + xr) * 64 + xc4]) = sxa;
  }
  __syncthreads();
  if (tid < 192) {
    const int which = tid >> 6, c = tid & 63;
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[(which * 16 + q) * 64 + c];
    if (c < K) {
      if (which < 2) a.spart[((size_t)chunk * 2 + which) * K + c] = t;
      else a.sxp[(size_t)chunk * K + c] = t;
    }
  }
}

// M = W^T diag(alpha) W and c = beta^T W of the Gram-form backward, from Wt = W^T [K][ldw].
// One workgroup per 4 x 16 block of M: thread (i, j, slice) adds every 4th n of its slice in
// float64 (alpha changes sign across channels), the four slices meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void gram_prep_kernel(int N, int K, const float *__restrict__ Wt,
                                                        int ldw, const float *__restrict__ alpha,
                                                        const float *__restrict__ beta,
                                                        float *__restrict__ M,
                                                        float *__restrict__ cvec) {
  __shared__ double red[2][4][64];
  const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int i = blockIdx.y * 4 + (e >> 4), j = blockIdx.x * 16 + (e & 15);
  double m = 0.0, c = 0.0;
  if (i < K && j < K) {
    const float *wi = Wt + (size_t)i * ldw, *wj = Wt + (size_t)j * ldw;
    const int n0 = (N + 3) / 4 * sl, n1 = min(N, n0 + (N + 3) / 4);
    for (int n = n0; n < n1; ++n) {
      m += (double)alpha[n] * (double)wi[n] * (double)wj[n];
      if (j == 0) c += (double)beta[n] * (double)wi[n];
    }
  }
  red[0][sl][e] = m;
  red[1][sl][e] = c;
  __syncthreads();
  if (sl == 0 && i < K && j < K) {
    M[(size_t)i * K + j] = (float)(red[0][0][e] + red[0][1][e] + red[0][2][e] + red[0][3][e]);
    if (j == 0) cvec[i] = (float)(red[1][0][e] + red[1][1][e] + red[1][2][e] + red[1][3][e]);
  }
}

// G64[i][j] = sum over chunks of gp[chunk][i][j]; sx64[k] likewise (float64, fixed order): 16
// elements x 16 chunk slices per workgroup
__global__ __launch_bounds__(256) void gram_reduce_kernel(int K, int chunks,
                                                          const float *__restrict__ gp,
                                                          const float *__restrict__ sxp,
                                                          double *__restrict__ G64,
                                                          double *__restrict__ sx64) {
  __shared__ double red[16][17];
  const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + e;
  const int total = K * K;
  double s = 0.0;
  if (i < total) {
#pragma unroll 4
    for (int c = sl; c < chunks; c += 16) s += (double)gp[(size_t)c * total + i];
  } else if (i < total + K) {
    const int k = i - total;
#pragma unroll 4
    for (int c = sl; c < chunks; c += 16) s += (double)sxp[(size_t)c * K + k];
  }
  red[sl][e] = s;
  __syncthreads();
  if (sl == 0 && i < total + K) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][e];
    if (i < total) G64[i] = t;
    else sx64[i - total] = t;
  }
}

// dW[n][k] = sum over chunks of pw[chunk][n][k]  +  alpha[n] sum_j W[n][j] G[j][k]  +  beta[n] sx[k]
// (w = W_l [N][K] row-major).  16 elements (lanes along k) x 16 slices per workgroup: a slice adds
// every 16th chunk and every 16th j, the slices meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void gram_finish_kernel(
    int N, int K, int chunks, const float *__restrict__ pw, const float *__restrict__ w,
    const float *__restrict__ alpha, const float *__restrict__ beta,
    const double *__restrict__ G64, const double *__restrict__ sx64, float *__restrict__ dw) {
  __shared__ double red[2][16][17];
  const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + e;
  double s = 0.0, d = 0.0;
  int n = 0, k = 0;
  if (i < N * K) {
    n = i / K;
    k = i - n * K;
#pragma unroll 4
    for (int c = sl; c < chunks; c += 16) s += (double)pw[(size_t)c * N * K + i];
    const float *wr = w + (size_t)n * K;
    for (int j = sl; j < K; j += 16) d += (double)wr[j] * G64[(size_t)j * K + k];
  }
  red[0][sl][e] = s;
  red[1][sl][e] = d;
  __syncthreads();
  if (sl == 0 && i < N * K) {
    double ts = 0.0, td = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      ts += red[0][q][e];
      td += red[1][q][e];
    }
    dw[i] = (float)(ts + (double)alpha[n] * td + (double)beta[n] * sx64[k]);
  }
}

// dw[i] = sum over chunks of pw[chunk][i], fixed order; EL elements x SL chunk slices per block
// (4 x 64 for the small weight matrices, whose launches are latency-bound; 16 x 16 keeps the
// reads of the large ones coalesced).
template <int EL, int SL>
__global__ __launch_bounds__(256) void reduce_chunks_kernel(int total, int chunks,
                                                            const float *__restrict__ pw,
                                                            float *__restrict__ dw) {
  static_assert(EL * SL == 256, "one block");
  __shared__ double red[SL][EL + 1];
  const int tx = threadIdx.x % EL, ty = threadIdx.x / EL;
  const int i = blockIdx.x * EL + tx;
  double s = 0.0;
  if (i < total)
#pragma unroll 8
    for (int c = ty; c < chunks; c += SL) s += (double)pw[(size_t)c * total + i];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && i < total) {
    s = 0.0;
    for (int y0 = 0; y0 < SL; y0 += 8) {  // fixed order
      double a = 0.0;
#pragma unroll
      for (int y = 0; y < 8; ++y) a += red[y0 + y][tx];
      s += a;
    }
    dw[i] = (float)s;
  }
}

// The same reduction for SEVERAL weight gradients in one launch: a layer call's TN GEMMs all
// run on the weight-gradient stream and nothing reads a dW before the call returns, so their
// split-K reductions are collected (reduce_batch) and issued once, behind the last GEMM -- one
// launch per call instead of one per layer (25 -> 9 per training step).  Same summation order
// per element as reduce_chunks_kernel<16, 16>.
constexpr int kMaxReduceSeg = 32;   // (a whole backbone backward: 12 + 4 layers, their first-layer halves)
struct ReduceArgs {
  const float *pw[kMaxReduceSeg];
  float *dw[kMaxReduceSeg];
  int total[kMaxReduceSeg], chunks[kMaxReduceSeg], first[kMaxReduceSeg + 1];
  // kpad != 0 (reduce_chunks_multi_kernel only): the partials are [.][kpad] rows whose first kout
  // columns are written as dense [.][kout] rows -- a first layer's weight gradient without the
  // zero columns its 4-aligned input width added (see reduce_unpad_next)
  int kpad[kMaxReduceSeg], kout[kMaxReduceSeg];
  // ... written with row pitch ldo at column offset coff (default: ldo = kout, coff = 0): a
  // gradient that is one column block of a wider parameter (the per-point first layer's dW_x /
  // dW_f halves of dW_0) goes straight into the parameter's layout
  int ldo[kMaxReduceSeg], coff[kMaxReduceSeg];
  int n;
};
__global__ __launch_bounds__(256) void reduce_chunks_multi_kernel(ReduceArgs a) {
  __shared__ double red[16][65];
  // the segment of this block: unrolled scan with STATIC indices (see prep_weights_kernel)
  const float *pw = a.pw[0];
  float *dw = a.dw[0];
  int total = a.total[0], chunks = a.chunks[0], first = 0, kpad = a.kpad[0], kout = a.kout[0];
  int ldo = a.ldo[0], coff = a.coff[0];
#pragma unroll
  for (int i = 1; i < kMaxReduceSeg; ++i)
    if (i < a.n && (int)blockIdx.x >= a.first[i]) {
      pw = a.pw[i]; dw = a.dw[i]; total = a.total[i]; chunks = a.chunks[i]; first = a.first[i];
      kpad = a.kpad[i]; kout = a.kout[i]; ldo = a.ldo[i]; coff = a.coff[i];
    }
  // 16 slices of the chunk axis x 16 threads of four consecutive elements: a slice reads 256
  // contiguous bytes per chunk (one element per thread read 64: SA2's 16 MB of partials took
  // 22 us); every element's chunks are added in the order they always were
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
  const int i = (((int)blockIdx.x - first) * 16 + tx) * 4;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (i < total) {   // (total % 4 == 0)
#pragma unroll 4
    for (int c = ty; c < chunks; c += 16) {
      const float4 v = *reinterpret_cast<const float4 *>(pw + (size_t)c * total + i);
      s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
    }
  }
  red[ty][tx * 4 + 0] = s0; red[ty][tx * 4 + 1] = s1;
  red[ty][tx * 4 + 2] = s2; red[ty][tx * 4 + 3] = s3;
  __syncthreads();
  if (threadIdx.x < 64 && ((int)blockIdx.x - first) * 64 + (int)threadIdx.x < total) {
    const int e = threadIdx.x, at = ((int)blockIdx.x - first) * 64 + e;
    double s = 0.0;
    for (int y0 = 0; y0 < 16; y0 += 8) {
      double acc = 0.0;
#pragma unroll
      for (int y = 0; y < 8; ++y) acc += red[y0 + y][e];
      s += acc;
    }
    if (kpad == 0) {
      dw[at] = (float)s;
    } else {
      const int row = at / kpad, col = at - row * kpad;
      if (col < kout) dw[(size_t)row * ldo + coff + col] = (float)s;
    }
  }
}

// Few chunks of a LARGE gradient (the decoder's 288 x 2048 FFN weights: 4 - 16 partials of
// 590 000 elements): one thread per 4 consecutive elements streams the partials with 16-byte
// loads (the 16-element blocks above read 64-byte pieces and leave 3/4 of their threads idle
// when there are only 4 chunks: 64 us for a reduction that moves 12 MB).
__global__ __launch_bounds__(256) void reduce_chunks_wide_kernel(int total4, int chunks,
                                                                 const float4 *__restrict__ pw,
                                                                 float4 *__restrict__ dw) {
  const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (i >= total4) return;
  double x = 0.0, y = 0.0, z = 0.0, w = 0.0;
#pragma unroll 4
  for (int c = 0; c < chunks; ++c) {
    const float4 v = pw[(size_t)c * total4 + i];
    x += (double)v.x; y += (double)v.y; z += (double)v.z; w += (double)v.w;
  }
  dw[i] = make_float4((float)x, (float)y, (float)z, (float)w);
}

// ... and of several such gradients in one launch (a decoder layer's seven, a point-wise chain's
// two or three): same per-element order as reduce_chunks_wide_kernel.
__global__ __launch_bounds__(256) void reduce_chunks_wide_multi_kernel(ReduceArgs a) {
  const float *pw = a.pw[0];
  float *dw = a.dw[0];
  int total4 = a.total[0], chunks = a.chunks[0], first = 0;
#pragma unroll
  for (int i = 1; i < kMaxReduceSeg; ++i)   // static indices (see prep_weights_kernel)
    if (i < a.n && (int)blockIdx.x >= a.first[i]) {
      pw = a.pw[i]; dw = a.dw[i]; total4 = a.total[i]; chunks = a.chunks[i]; first = a.first[i];
    }
  const int i = ((int)blockIdx.x - first) * 256 + (int)threadIdx.x;
  if (i >= total4) return;
  double x = 0.0, y = 0.0, z = 0.0, w = 0.0;
#pragma unroll 4
  for (int c = 0; c < chunks; ++c) {
    const float4 v = reinterpret_cast<const float4 *>(pw)[(size_t)c * total4 + i];
    x += (double)v.x; y += (double)v.y; z += (double)v.z; w += (double)v.w;
  }
  reinterpret_cast<float4 *>(dw)[i] = make_float4((float)x, (float)y, (float)z, (float)w);
}

struct ReduceBatch {
  ReduceArgs args;
  ReduceArgs wide;   // (total = number of float4, first = 256-thread blocks)
  bool on = false;
  int depth = 0;   // nested scopes: the outermost one launches (btr_backbone_backward around its layers)
  int next_kpad = 0, next_kout = 0, next_ldo = 0, next_coff = 0;   // reduce_unpad_next
};
inline ReduceBatch &reduce_batch() {
  static thread_local ReduceBatch b;
  return b;
}
// dw = sum over chunks of pw: now, or with the batch of the running layer call
inline void reduce_chunks_launch(int total, int chunks, const float *pw, float *dw,
                                 hipStream_t st) {
  ReduceBatch &b = reduce_batch();
  const int kpad = b.next_kpad, kout = b.next_kout;
  const int ldo = b.next_ldo ? b.next_ldo : kout, coff = b.next_coff;
  b.next_kpad = b.next_kout = b.next_ldo = b.next_coff = 0;
  if (kpad != 0 && !(b.on && b.args.n < kMaxReduceSeg)) {   // a launch of its own
    ReduceArgs a{};
    a.pw[0] = pw; a.dw[0] = dw; a.total[0] = total; a.chunks[0] = chunks;
    a.kpad[0] = kpad; a.kout[0] = kout; a.ldo[0] = ldo; a.coff[0] = coff;
    a.first[1] = cdiv(total, 64); a.n = 1;
    hipLaunchKernelGGL(reduce_chunks_multi_kernel, dim3(a.first[1]), dim3(256), 0, st, a);
    return;
  }
  if (kpad == 0 && chunks <= 16 && total >= 65536 && total % 4 == 0) {
    if (b.on && b.wide.n < kMaxReduceSeg) {
      ReduceArgs &a = b.wide;
      a.pw[a.n] = pw;
      a.dw[a.n] = dw;
      a.total[a.n] = total / 4;
      a.chunks[a.n] = chunks;
      a.first[a.n + 1] = a.first[a.n] + cdiv(total / 4, 256);
      ++a.n;
      return;
    }
    hipLaunchKernelGGL(reduce_chunks_wide_kernel, dim3(cdiv(total / 4, 256)), dim3(256), 0, st,
                       total / 4, chunks, (const float4 *)pw, (float4 *)dw);
    return;
  }
  if (b.on && b.args.n < kMaxReduceSeg) {
    ReduceArgs &a = b.args;
    a.pw[a.n] = pw;
    a.dw[a.n] = dw;
    a.total[a.n] = total;
    a.chunks[a.n] = chunks;
    a.kpad[a.n] = kpad;
    a.kout[a.n] = kout;
    a.ldo[a.n] = ldo;
    a.coff[a.n] = coff;
    a.first[a.n + 1] = a.first[a.n] + cdiv(total, 64);
    ++a.n;
    return;
  }
  if (total <= 1024 && chunks >= 64)
    hipLaunchKernelGGL((reduce_chunks_kernel<4, 64>), dim3(cdiv(total, 4)), dim3(256), 0, st,
                       total, chunks, pw, dw);
  else
    hipLaunchKernelGGL((reduce_chunks_kernel<16, 16>), dim3(cdiv(total, 16)), dim3(256), 0, st,
                       total, chunks, pw, dw);
}
// (internal.hpp) the NEXT split-K reduction issued on this host thread writes dense [.][kout] rows
// from its [.][kpad] partials
void reduce_unpad_next(int kpad, int kout, int ldo, int coff) {
  ReduceBatch &b = reduce_batch();
  b.next_kpad = kpad;
  b.next_kout = kout;
  b.next_ldo = ldo;
  b.next_coff = coff;
}
// (internal.hpp) collect the split-K reductions issued on this host thread until the flush
void reduce_batch_begin() {
  ReduceBatch &b = reduce_batch();
  if (b.depth++ > 0) return;   // inside an outer scope: keep collecting into it
  b.on = true;
  b.args.n = b.wide.n = 0;
  b.args.first[0] = b.wide.first[0] = 0;
}
void reduce_batch_flush(hipStream_t st) {
  ReduceBatch &b = reduce_batch();
  if (b.depth > 0 && --b.depth > 0) return;   // the outer scope's flush launches
  GemmTrace trace_(st);   // (the second half of the family's split-K products)
  b.on = false;
  if (b.args.n > 0)
    hipLaunchKernelGGL(reduce_chunks_multi_kernel, dim3(b.args.first[b.args.n]), dim3(256), 0, st,
                       b.args);
  if (b.wide.n > 0)
    hipLaunchKernelGGL(reduce_chunks_wide_multi_kernel, dim3(b.wide.first[b.wide.n]), dim3(256), 0,
                       st, b.wide);
  b.args.n = b.wide.n = 0;
}

// ------------------------------------------------------------- scatter of dX0 (layer-0 dgrad)
// dX0[r][c] -> c < 3: d xyz[b, idx, c] += v/radius, d new_xyz[b, m, c] -= v/radius;
// c >= 3: d features[b, idx, c-3] += v.  Instead of float atomics on (B,C,N) (what
// group_points_grad does) the neighbour lists are inverted once per call:
//   refs[b][off[n] .. off[n+1]) = the rows jk = m*S+s of batch b whose idx == n
// (integer atomics only), then one wave per point sums its rows -- each a contiguous run of
// C floats -- and writes the point's gradient channel-last.
__global__ __launch_bounds__(256) void csr_count_kernel(long long ms, int N,
                                                        const int *__restrict__ idx,
                                                        int *__restrict__ cnt) {
  const int bi = blockIdx.y;
  for (long long jk = (long long)blockIdx.x * 256 + threadIdx.x; jk < ms;
       jk += (long long)gridDim.x * 256)
    atomicAdd(cnt + (size_t)bi * (N + 1) + idx[(size_t)bi * ms + jk], 1);
}

// One block per batch element: exclusive scan of cnt[0..N) -> off[0..N], cursor = off.
__global__ __launch_bounds__(1024) void csr_scan_kernel(int N, int *__restrict__ cnt_off,
                                                        int *__restrict__ cursor) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int bi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int *c = cnt_off + (size_t)bi * (N + 1);
  int *cur = cursor + (size_t)bi * N;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < N; base += 1024) {
    const int i = base + tid;
    const int v = i < N ? c[i] : 0;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int pre = carry_s;
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    const int excl = pre + incl - v;
    if (i < N) {
      c[i] = excl;
      cur[i] = excl;
    }
    __syncthreads();
    if (tid == 1023) carry_s = excl + v;
    __syncthreads();
  }
  if (tid == 0) c[N] = carry_s;
}

__global__ __launch_bounds__(256) void csr_fill_kernel(long long ms, int N,
                                                       const int *__restrict__ idx,
                                                       int *__restrict__ cursor,
                                                       int *__restrict__ refs) {
  const int bi = blockIdx.y;
  for (long long jk = (long long)blockIdx.x * 256 + threadIdx.x; jk < ms;
       jk += (long long)gridDim.x * 256) {
    const int pos = atomicAdd(cursor + (size_t)bi * N + idx[(size_t)bi * ms + jk], 1);
    refs[(size_t)bi * ms + pos] = (int)jk;
  }
}

// Small scenes (N <= kCsrSmallN points): the whole inversion -- count, exclusive scan, fill --
// by ONE workgroup per batch element with the counters in LDS.  The four-launch global-atomic
// version spreads a scene's updates over every XCD, so its integer atomics are resolved at
// device scope (23 us per launch for 262 144 updates); here they are LDS atomics.
constexpr int kCsrSmallN = 8192;
__global__ __launch_bounds__(1024) void csr_small_kernel(long long ms, int N,
                                                         const int *__restrict__ idx,
                                                         int *__restrict__ off,
                                                         int *__restrict__ refs) {
  __shared__ int cnt[kCsrSmallN];
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int bi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  idx += (size_t)bi * ms;
  refs += (size_t)bi * ms;
  off += (size_t)bi * (N + 1);
  for (int i = tid; i < N; i += 1024) cnt[i] = 0;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (long long jk = tid; jk < ms; jk += 1024) atomicAdd(&cnt[idx[jk]], 1);
  __syncthreads();
  for (int base = 0; base < N; base += 1024) {  // exclusive scan, 1024 bins at a time
    const int i = base + tid;
    const int v = i < N ? cnt[i] : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int pre = carry_s;
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    const int excl = pre + incl - v;
    if (i < N) {
      cnt[i] = excl;  // becomes the fill cursor
      off[i] = excl;
    }
    __syncthreads();
    if (tid == 1023) carry_s = excl + v;
    __syncthreads();
  }
  if (tid == 0) off[N] = carry_s;
  for (long long jk = tid; jk < ms; jk += 1024) {
    const int pos = atomicAdd(&cnt[idx[jk]], 1);
    refs[pos] = (int)jk;
  }
}

// (host entry for other translation units: interpolate.hip inverts its 3-NN lists with it)
bool csr_small_supported(int n_bins) { return n_bins > 0 && n_bins <= kCsrSmallN; }
void csr_small_launch(int b, long long entries, int n_bins, const int *idx, int *off, int *refs,
                      hipStream_t st) {
  hipLaunchKernelGGL(csr_small_kernel, dim3(b), dim3(1024), 0, st, entries, n_bins, idx, off,
                     refs);
}

// Sum of the rows base[rf[beg .. end)][0 .. C) by one wave, C in {128, 256} and 16-byte aligned rows:
// a lane owns four channels (float4 loads: a row is 32 or 64 lanes), with C = 128 the two halves
// of the wave take alternate rows; four rows per lane in flight.  The result (all rows) is valid
// in the lanes < C / 4.  (The 4-byte-per-lane form -- one dependent index load and one or two
// 256-byte loads per row -- moved SA2's 58 MB at 1.2 TB/s.)
__device__ __forceinline__ float4 wave_sum_rows4(const float *__restrict__ base, int ldx, int C,
                                                 const int *__restrict__ rf, int beg, int end,
                                                 int lane) {
  const int lpr = C >> 2;                 // lanes per row: 32 or 64
  const int sub = lane / lpr, nsub = 64 / lpr;
  const float *col = base + (lane % lpr) * 4;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  int i = beg + sub;
  for (; i + 3 * nsub < end; i += 4 * nsub) {
    const int r0 = rf[i], r1 = rf[i + nsub], r2 = rf[i + 2 * nsub], r3 = rf[i + 3 * nsub];
    const float4 v0 = *reinterpret_cast<const float4 *>(col + (size_t)r0 * ldx);
    const float4 v1 = *reinterpret_cast<const float4 *>(col + (size_t)r1 * ldx);
    const float4 v2 = *reinterpret_cast<const float4 *>(col + (size_t)r2 * ldx);
    const float4 v3 = *reinterpret_cast<const float4 *>(col + (size_t)r3 * ldx);
    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
    a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
    a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
  }
  for (; i < end; i += nsub) {
    const float4 v0 = *reinterpret_cast<const float4 *>(col + (size_t)rf[i] * ldx);
    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
  }
  float4 a = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y),
                         (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
  if (nsub == 2) {
    a.x += __shfl_xor(a.x, 32);
    a.y += __shfl_xor(a.y, 32);
    a.z += __shfl_xor(a.z, 32);
    a.w += __shfl_xor(a.w, 32);
  }
  return a;
}
__device__ __forceinline__ bool wave_sum_rows4_ok(const float *p, int ldx, int xoff, int C) {
  return (C == 128 || C == 256) && ldx % 4 == 0 && xoff % 4 == 0 &&
         (reinterpret_cast<size_t>(p) & 15) == 0;
}

// One wave per point n: dfeat_cl[b][n][c] = sum over refs of dX0[ref][xoff + c];
// dxyz[b][n][0..3) = (sum of dX0[ref][0..3)) * inv_radius.
__global__ __launch_bounds__(256) void csr_reduce_kernel(
    int N, long long ms, int C, int ldx, int xoff, float inv_radius,
    const float *__restrict__ dX, const int *__restrict__ off, const int *__restrict__ refs,
    float *__restrict__ dfeat_cl, float *__restrict__ dxyz) {
  const int bi = blockIdx.y;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  const int beg = off[(size_t)bi * (N + 1) + n], end = off[(size_t)bi * (N + 1) + n + 1];
  const int *rf = refs + (size_t)bi * ms;
  const float *base = dX + (size_t)bi * ms * ldx;
  if (dfeat_cl && wave_sum_rows4_ok(base, ldx, xoff, C)) {
    const float4 a = wave_sum_rows4(base + xoff, ldx, C, rf, beg, end, lane);
    if (lane < (C >> 2))
      *reinterpret_cast<float4 *>(dfeat_cl + ((size_t)bi * N + n) * C + lane * 4) = a;
  } else if (dfeat_cl) {
    // up to 4 channels per lane (C <= 256), 4 neighbour rows in flight per iteration
    for (int c0 = 0; c0 < C; c0 += 256) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      int i = beg;
      for (; i + 4 <= end; i += 4) {
        const float *r0 = base + (size_t)rf[i] * ldx + xoff;
        const float *r1 = base + (size_t)rf[i + 1] * ldx + xoff;
        const float *r2 = base + (size_t)rf[i + 2] * ldx + xoff;
        const float *r3 = base + (size_t)rf[i + 3] * ldx + xoff;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = c0 + q * 64 + lane;
          if (c < C) acc[q] += (r0[c] + r1[c]) + (r2[c] + r3[c]);
        }
      }
      for (; i < end; ++i) {
        const float *r0 = base + (size_t)rf[i] * ldx + xoff;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = c0 + q * 64 + lane;
          if (c < C) acc[q] += r0[c];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = c0 + q * 64 + lane;
        if (c < C) dfeat_cl[((size_t)bi * N + n) * C + c] = acc[q];
      }
    }
  }
  if (dxyz && lane < 3) {
    float acc = 0.f;
    for (int i = beg; i < end; ++i) acc += base[(size_t)rf[i] * ldx + lane];
    dxyz[((size_t)bi * N + n) * 3 + lane] = acc * inv_radius;
  }
}

// dnew_xyz[b][m][c] = -(sum_s dX0[(b,m,s)][c]) * inv_radius  (rows of a centre are contiguous)
__global__ __launch_bounds__(256) void centre_grad_kernel(long long groups, int S, int ldx,
                                                          float inv_radius,
                                                          const float *__restrict__ dX,
                                                          float *__restrict__ dnew_xyz) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= groups * 3) return;
  const long long g = t / 3;
  const int c = (int)(t - g * 3);
  float acc = 0.f;
  for (int s2 = 0; s2 < S; ++s2) acc += dX[((size_t)g * S + s2) * ldx + c];
  dnew_xyz[t] = -acc * inv_radius;
}


// =================================================================== compact rows ("sac")
// The reference pads every ball-query row with copies of its first hit (ball_query_gpu.cu:39-43)
// and then runs the shared MLP over all nsample rows of every group: at the benchmark shape
// 37 % of SA1's and 69 % of SA2's rows are such copies.  A copy computes exactly what its
// original computes, so the layer is evaluated on the DISTINCT rows only:
//   * group g keeps len_g = ceil8(cnt_g) rows (cnt_g distinct neighbours; the len_g - cnt_g
//     alignment rows are further copies of the first hit), laid out back to back: row r
//     belongs to the 8-row block r >> 3, block b to group bgrp[b], group g starts at goff[g];
//   * the first row of a group stands for w = 1 + nsample - len_g copies of itself, every other
//     row for one: train-mode BatchNorm statistics weight the rows by w, the max-pool is
//     unaffected (first occurrence wins ties, as in F.max_pool2d), and in the backward each
//     compact row carries the gradient SUMMED over its copies, which only changes the
//     statistics term of the BatchNorm backward (it counts once per copy): see
//     bn_relu_bwd_apply_kernel and the PRO == 2 prologue.
// The row count is data dependent and lives on the device (dims[0]); every kernel takes it
// from there, grids and buffers are sized for the dense worst case, nothing synchronises.
// Same results as the dense evaluation up to float32 summation order.
__global__ __launch_bounds__(256) void sac_count_kernel(int groups, int S,
                                                        const int *__restrict__ idx,
                                                        int *__restrict__ len) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= groups) return;
  const int *row = idx + (size_t)g * S;
  const int first = row[0];
  int cnt = 1;
  for (int s2 = 1; s2 < S; ++s2) cnt += row[s2] != first ? 1 : 0;
  len[g] = (cnt + 7) & ~7;
}

// exclusive scan of len[0..groups) -> goff[0..groups]; dims[0] = rows, dims[1] = blocks
__global__ __launch_bounds__(1024) void sac_scan_kernel(int groups, const int *__restrict__ len,
                                                        int *__restrict__ goff,
                                                        int *__restrict__ dims) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < groups; base += 1024) {
    const int i = base + tid;
    const int v = i < groups ? len[i] : 0;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int pre = carry_s;
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    const int excl = pre + incl - v;
    if (i < groups) goff[i] = excl;
    __syncthreads();
    if (tid == 1023) carry_s = excl + v;
    __syncthreads();
  }
  if (tid == 0) {
    goff[groups] = carry_s;
    dims[0] = carry_s;
    dims[1] = carry_s >> 3;
  }
}

// per (group, slot): cidx[row] = neighbour index of the compact row; per block: group, weight
__global__ __launch_bounds__(256) void sac_fill_kernel(int groups, int S,
                                                       const int *__restrict__ idx,
                                                       const int *__restrict__ goff,
                                                       int *__restrict__ cidx,
                                                       int *__restrict__ bgrp,
                                                       float *__restrict__ bw) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)groups * S) return;
  const int g = (int)(t / S), slot = (int)(t - (long long)g * S);
  const int beg = goff[g], len = goff[g + 1] - beg;
  if (slot >= len) return;
  // a ball-query row is its distinct hits in ascending order, then copies of the first one:
  // slot < cnt is the slot-th hit, cnt <= slot < len a copy of the first (= what idx holds)
  cidx[beg + slot] = idx[(size_t)g * S + slot];
  if ((slot & 7) == 0) {
    bgrp[(beg + slot) >> 3] = g;
    bw[(beg + slot) >> 3] = slot == 0 ? (float)(1 + S - len) : 1.f;
  }
}

// X0 of the compact rows (as sa_gather_kernel, one group per bgrp entry)
__global__ __launch_bounds__(256) void sac_gather_kernel(
    int N, int M, int C, int ldx, int use_xyz, float inv_radius, const float *__restrict__ xyz,
    const float *__restrict__ new_xyz, const float *__restrict__ feats_cl,
    const int *__restrict__ cidx, const int *__restrict__ bgrp, const int *__restrict__ dims,
    float *__restrict__ X) {
  const int rows = dims[0];
  const int tpr = min(64, ldx);
  const int rows_per_block = 256 / tpr;
  const int lr = threadIdx.x / tpr, lc = threadIdx.x % tpr;
  if (lr >= rows_per_block) return;
  const int xoff = use_xyz ? 3 : 0;
  for (int r = blockIdx.x * rows_per_block + lr; r < rows; r += gridDim.x * rows_per_block) {
    const int g = bgrp[r >> 3];
    const int bi = g / M;
    const int ii = cidx[r];
    float *out = X + (size_t)r * ldx;
    const float *f = feats_cl ? feats_cl + ((size_t)bi * N + ii) * C : nullptr;
    for (int c = lc; c < ldx; c += tpr) {
      float v = 0.f;
      if (c < xoff) {
        v = (xyz[((size_t)bi * N + ii) * 3 + c] - new_xyz[(size_t)g * 3 + c]) * inv_radius;
      } else if (c - xoff < C) {
        v = f[c - xoff];
      }
      out[c] = v;
    }
  }
}

// Max-pool of the compact rows from the per-BLOCK extrema the last GEMM's epilogue emits
// (gemm_nt_kernel PS = 8): group g = blocks goff[g]/8 .. goff[g+1]/8.  out = relu(a*ext + b),
// first occurrence wins (blocks ascending, strict >); arg = row within the group.
__global__ __launch_bounds__(256) void sac_pool_kernel(
    int M, int C, long long groups, const float *__restrict__ gext,
    const unsigned char *__restrict__ aext, const int *__restrict__ goff,
    const float *__restrict__ scale, const float *__restrict__ shift, float *__restrict__ out,
    float *__restrict__ out_cl, unsigned char *__restrict__ arg,
    float *__restrict__ ywin = nullptr) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= groups * C) return;
  const long long g = t / C;
  const int c = (int)(t - g * C);
  const int b0 = goff[g] >> 3, b1 = goff[g + 1] >> 3;
  const float a = scale[c], b = shift[c];
  float best = -1.f, by = 0.f;
  int ba = 0;
  for (int blk = b0; blk < b1; ++blk) {
    const float e = gext[(size_t)blk * C + c];
    const float v = fmaxf(fmaf(a, e, b), 0.f);
    if (v > best) {
      best = v;
      by = e;
      ba = ((blk - b0) << 3) + aext[(size_t)blk * C + c];
    }
  }
  const long long bi = g / M;
  const int m = (int)(g - bi * M);
  out[((size_t)bi * C + c) * M + m] = best;
  if (out_cl) out_cl[t] = best;
  arg[t] = best > 0.f ? (unsigned char)ba : (unsigned char)0;
  if (ywin) ywin[t] = by;   // the pre-BN value behind `out` (only read where out > 0)
}

// The same two pools (goff != nullptr: per-block extrema of compact rows; nullptr: one extremum per
// group) on a 32-group x 32-channel tile per workgroup: the (B, C, M) output is written through an
// LDS transpose as 128-byte segments along m -- with a thread per (group, channel) the 4-byte
// stores of consecutive channels are M floats apart, one write transaction each (SA1: 2 M of them,
// 42 us for a kernel whose bytes take 18).  Needs M % 32 == 0 (a tile stays in one batch element).
__global__ __launch_bounds__(256) void sa_pool_tile_kernel(
    int M, int C, const float *__restrict__ gext, const unsigned char *__restrict__ aext,
    const int *__restrict__ goff, const float *__restrict__ scale,
    const float *__restrict__ shift, float *__restrict__ out, float *__restrict__ out_cl,
    unsigned char *__restrict__ arg, float *__restrict__ ywin = nullptr) {
  __shared__ float T[32 * 33];
  const int c = (int)blockIdx.y * 32 + (threadIdx.x & 31), jj = threadIdx.x >> 5;
  const long long g0 = (long long)blockIdx.x * 32;
  const bool live = c < C;
  const float a = live ? scale[c] : 0.f, b = live ? shift[c] : 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = jj + 8 * i;
    const long long g = g0 + j;
    float best = -1.f, by = 0.f;
    int ba = 0;
    if (live) {
      if (goff) {
        const int b0 = goff[g] >> 3, b1 = goff[g + 1] >> 3;
        for (int blk = b0; blk < b1; ++blk) {
          const float e = gext[(size_t)blk * C + c];
          const float v = fmaxf(fmaf(a, e, b), 0.f);
          if (v > best) {
            best = v;
            by = e;
            ba = ((blk - b0) << 3) + aext[(size_t)blk * C + c];
          }
        }
      } else {
        by = gext[(size_t)g * C + c];
        best = fmaxf(fmaf(a, by, b), 0.f);
        ba = aext[(size_t)g * C + c];
      }
      if (out_cl) out_cl[(size_t)g * C + c] = best;
      arg[(size_t)g * C + c] = best > 0.f ? (unsigned char)ba : (unsigned char)0;
      if (ywin) ywin[(size_t)g * C + c] = by;   // the pre-BN value behind `out`
    }
    T[(threadIdx.x & 31) * 33 + j] = best;
  }
  __syncthreads();
  const long long bi = g0 / M;
  const int m0 = (int)(g0 - bi * M);
  const int j = threadIdx.x & 31;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cl = jj + 8 * i, cc = (int)blockIdx.y * 32 + cl;
    if (cc < C) out[((size_t)bi * C + cc) * M + m0 + j] = T[cl * 33 + j];
  }
}

// The same pool with four channels per thread (C % 4 == 0): a workgroup takes 32 groups x 128
// channels, a thread one group's 16-byte channel quad -- 512-byte rows of gext and 128-byte rows
// of aext per 32 lanes instead of 128-byte / 32-byte ones, and a quarter of the loop iterations
// (SA1: 88 000 blocks x 128 channels, 59 us in the one-channel form for 56 MB).  Same selection
// rule (blocks ascending, strict >), so the results are identical.
__global__ __launch_bounds__(256) void sa_pool_tile4_kernel(
    int M, int C, const float *__restrict__ gext, const unsigned char *__restrict__ aext,
    const int *__restrict__ goff, const float *__restrict__ scale,
    const float *__restrict__ shift, float *__restrict__ out, float *__restrict__ out_cl,
    unsigned char *__restrict__ arg, float *__restrict__ ywin) {
  __shared__ float T[128 * 33];
  const int q = threadIdx.x & 31, gi = threadIdx.x >> 5;
  const int c = (int)blockIdx.y * 128 + q * 4;
  const long long g0 = (long long)blockIdx.x * 32;
  const bool live = c < C;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (live) {
    a = *reinterpret_cast<const float4 *>(scale + c);
    b = *reinterpret_cast<const float4 *>(shift + c);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = gi + 8 * i;
    const long long g = g0 + j;
    float4 best = make_float4(-1.f, -1.f, -1.f, -1.f), by = make_float4(0.f, 0.f, 0.f, 0.f);
    int ba[4] = {0, 0, 0, 0};
    if (live) {
      const int b0 = goff ? goff[g] >> 3 : (int)g, b1 = goff ? goff[g + 1] >> 3 : (int)g + 1;
      for (int blk = b0; blk < b1; ++blk) {
        const float4 e = *reinterpret_cast<const float4 *>(gext + (size_t)blk * C + c);
        const uchar4 r = *reinterpret_cast<const uchar4 *>(aext + (size_t)blk * C + c);
        const int off = goff ? (blk - b0) << 3 : 0;
        const float vx = fmaxf(fmaf(a.x, e.x, b.x), 0.f), vy = fmaxf(fmaf(a.y, e.y, b.y), 0.f);
        const float vz = fmaxf(fmaf(a.z, e.z, b.z), 0.f), vw = fmaxf(fmaf(a.w, e.w, b.w), 0.f);
        if (vx > best.x) { best.x = vx; by.x = e.x; ba[0] = off + r.x; }
        if (vy > best.y) { best.y = vy; by.y = e.y; ba[1] = off + r.y; }
        if (vz > best.z) { best.z = vz; by.z = e.z; ba[2] = off + r.z; }
        if (vw > best.w) { best.w = vw; by.w = e.w; ba[3] = off + r.w; }
      }
      const size_t o = (size_t)g * C + c;
      if (out_cl) *reinterpret_cast<float4 *>(out_cl + o) = best;
      *reinterpret_cast<uchar4 *>(arg + o) =
          make_uchar4(best.x > 0.f ? (unsigned char)ba[0] : 0, best.y > 0.f ? (unsigned char)ba[1] : 0,
                      best.z > 0.f ? (unsigned char)ba[2] : 0, best.w > 0.f ? (unsigned char)ba[3] : 0);
      if (ywin) *reinterpret_cast<float4 *>(ywin + o) = by;
    }
    T[(q * 4 + 0) * 33 + j] = best.x;
    T[(q * 4 + 1) * 33 + j] = best.y;
    T[(q * 4 + 2) * 33 + j] = best.z;
    T[(q * 4 + 3) * 33 + j] = best.w;
  }
  __syncthreads();
  const long long bi = g0 / M;
  const int m0 = (int)(g0 - bi * M);
  const int j = threadIdx.x & 31;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int cl = gi + 8 * i, cc = (int)blockIdx.y * 128 + cl;
    if (cc < C) out[((size_t)bi * C + cc) * M + m0 + j] = T[cl * 33 + j];
  }
}

// Inverted neighbour lists of the compact rows of ONE batch element per workgroup (as
// csr_small_kernel; rows goff[b*M] .. goff[(b+1)*M) of batch element b, N <= kCsrSmallN bins).
__global__ __launch_bounds__(1024) void sac_csr_kernel(int M, int N, const int *__restrict__ cidx,
                                                       const int *__restrict__ goff,
                                                       int *__restrict__ off,
                                                       int *__restrict__ refs) {
  __shared__ int cnt[kCsrSmallN];
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int bi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = goff[(size_t)bi * M], r1 = goff[(size_t)(bi + 1) * M];
  off += (size_t)bi * (N + 1);
  for (int i = tid; i < N; i += 1024) cnt[i] = 0;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int r = r0 + tid; r < r1; r += 1024) atomicAdd(&cnt[cidx[r]], 1);
  __syncthreads();
  for (int base = 0; base < N; base += 1024) {
    const int i = base + tid;
    const int v = i < N ? cnt[i] : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int pre = carry_s;
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    const int excl = pre + incl - v;
    if (i < N) {
      cnt[i] = excl;
      off[i] = r0 + excl;   // refs is indexed by global compact row position
    }
    __syncthreads();
    if (tid == 1023) carry_s = excl + v;
    __syncthreads();
  }
  if (tid == 0) off[N] = r0 + carry_s;
  for (int r = r0 + tid; r < r1; r += 1024) {
    const int pos = atomicAdd(&cnt[cidx[r]], 1);
    refs[r0 + pos] = r;      // global compact row
  }
}

// One wave per point: dfeat_cl[b][n][c] = sum over refs of dX0[row][xoff + c] (rows global)
__global__ __launch_bounds__(256) void sac_reduce_kernel(int N, int C, int ldx, int xoff,
                                                         const float *__restrict__ dX,
                                                         const int *__restrict__ off,
                                                         const int *__restrict__ refs,
                                                         float *__restrict__ dfeat_cl) {
  const int bi = blockIdx.y;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  const int beg = off[(size_t)bi * (N + 1) + n], end = off[(size_t)bi * (N + 1) + n + 1];
  if (wave_sum_rows4_ok(dX, ldx, xoff, C)) {
    const float4 a = wave_sum_rows4(dX + xoff, ldx, C, refs, beg, end, lane);
    if (lane < (C >> 2))
      *reinterpret_cast<float4 *>(dfeat_cl + ((size_t)bi * N + n) * C + lane * 4) = a;
    return;
  }
  for (int c0 = 0; c0 < C; c0 += 256) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = beg; i < end; ++i) {
      const float *r0 = dX + (size_t)refs[i] * ldx + xoff;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = c0 + q * 64 + lane;
        if (c < C) acc[q] += r0[c];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + q * 64 + lane;
      if (c < C) dfeat_cl[((size_t)bi * N + n) * C + c] = acc[q];
    }
  }
}


// ======================================================== point-wise MLP chains ("pm")
// The 1x1-convolution + BatchNorm + ReLU chains around the set-abstraction stack (feature
// propagation MLPs, vote generator, proposal head: 2 048 - 16 384 rows) run on the same GEMM /
// statistics / BN-backward kernels, with 64-row tiles so that these small problems still fill
// the chip (btr_pm_gemm_nt).  pm_out_kernel turns the last pre-BN output into what the next torch
// op reads: out[b][c][n] = f(a[c] * Y[b*N + n][c] + s[c]) in (B, C, N) and channel-last.
__global__ __launch_bounds__(256) void pm_out_kernel(int N, int C, int ldy,
                                                     const float *__restrict__ Y,
                                                     const float *__restrict__ scale,
                                                     const float *__restrict__ shift, int relu,
                                                     float *__restrict__ out_bcn,
                                                     float *__restrict__ out_cl,
                                                     const float *__restrict__ add = nullptr,
                                                     long long add_bs = 0) {
  __shared__ float tile[64][65];
  const int bi = blockIdx.z, n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
#pragma unroll
  for (int i = 0; i < 16; ++i) {   // read rows n0 + ty + 4i, channel c0 + tx (coalesced in c)
    const int n = n0 + ty + 4 * i, c = c0 + tx;
    float v = 0.f;
    if (n < N && c < C) {
      v = Y[((size_t)bi * N + n) * ldy + c];
      if (scale) v = fmaf(scale[c], v, shift[c]);
      if (relu) v = fmaxf(v, 0.f);
      if (out_cl) out_cl[((size_t)bi * N + n) * C + c] = v;
    }
    tile[ty + 4 * i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {   // write channel c0 + ty + 4i, point n0 + tx (coalesced in n)
    const int c = c0 + ty + 4 * i, n = n0 + tx;
    if (n < N && c < C) {
      float v = tile[tx][ty + 4 * i];
      if (add) v += add[(size_t)bi * add_bs + (size_t)c * N + n];
      out_bcn[((size_t)bi * C + c) * N + n] = v;
    }
  }
}

// rows[b*N + n][c] = x[b][c][n] (c < C), zero for C <= c < ldr: (B, C, N) -> channel-last rows
// zero / nzero: a run of floats the first workgroup clears on the way (the chain backward's bias
// gradients -- was a memset of its own in front of this kernel)
__global__ __launch_bounds__(256) void pm_rows_kernel(int N, int C, int ldr,
                                                      const float *__restrict__ x,
                                                      float *__restrict__ rows,
                                                      float *__restrict__ zero, int nzero,
                                                      float *__restrict__ colpart) {
  __shared__ float tile[64][65];
  __shared__ float red[4][64];
  if (zero && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
    for (int i = threadIdx.x; i < nzero; i += 256) zero[i] = 0.f;
  const int bi = blockIdx.z, n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, n = n0 + tx;
    tile[ty + 4 * i][tx] = (c < C && n < N) ? x[((size_t)bi * C + c) * N + n] : 0.f;
  }
  __syncthreads();
  float cs = 0.f;   // colpart: this tile's column sums (rows n >= N and columns c >= C hold zeros)
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int n = n0 + ty + 4 * i, c = c0 + tx;
    const float v = tile[tx][ty + 4 * i];
    if (n < N && c < ldr) rows[((size_t)bi * N + n) * ldr + c] = v;
    cs += v;
  }
  if (colpart) {   // colpart[(b * gridDim.x + tile)][ldr]: reduced by colsum_final_kernel
    red[ty][tx] = cs;
    __syncthreads();
    if (ty == 0 && c0 + tx < ldr)
      colpart[((size_t)bi * gridDim.x + blockIdx.x) * ldr + c0 + tx] =
          (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
  }
}

// (internal.hpp) btr_sa_bn_finalize with the skipped convolution bias added to the running mean
int bn_finalize_bias(int n, int nblk, double count, float eps, float momentum, const float *part,
                     const float *gamma, const float *beta, float *scale, float *shift,
                     float *mean, float *invstd, float *running_mean, float *running_var,
                     const float *rbias, int nbias, hipStream_t stream) {
  if (n <= 0) return BTR_OK;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(n, kRedCh)), dim3(256), 0, stream, n, nblk,
                     count, eps, momentum, part, gamma, beta, scale, shift, mean, invstd,
                     running_mean, running_var, rbias, nbias);
  return check_launch("sa_bn_finalize");
}

// (internal.hpp) btr_pm_rows that also clears `nzero` floats at `zero` and, with colpart
// [b * cdiv(n, 64)][ldr], leaves the column sums of every 64-row tile there
int pm_rows_zero(int b, int n, int c, int ldr, const float *x, float *rows, float *zero,
                 int nzero, float *colpart, hipStream_t stream) {
  if (b <= 0 || n <= 0 || c <= 0) return BTR_OK;
  BTR_REQUIRE(x && rows && ldr >= c, "pm_rows: bad arguments");
  hipLaunchKernelGGL(pm_rows_kernel, dim3(cdiv(n, 64), cdiv(ldr, 64), b), dim3(256), 0, stream, n,
                     c, ldr, x, rows, zero, nzero, colpart);
  return check_launch("pm_rows");
}

// rows[b*N + n][c] = a0[b*N + n][c] (+ a1[...]) for c < C, zero for C <= c < ldr: pm_rows_kernel for
// a gradient that already is channel-last rows (leading dimension C).  Same tiles, same zero /
// colpart duties, the column sums added in the same order, so a chain backward fed this way
// returns what it returns for the (B, C, N) form of the same values.
__global__ __launch_bounds__(256) void pm_rows_in_kernel(int N, int C, int ldr,
                                                         const float *__restrict__ a0,
                                                         const float *__restrict__ a1,
                                                         float *__restrict__ rows,
                                                         float *__restrict__ zero, int nzero,
                                                         float *__restrict__ colpart) {
  __shared__ float red[4][64];
  if (zero && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
    for (int i = threadIdx.x; i < nzero; i += 256) zero[i] = 0.f;
  const int bi = blockIdx.z, n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = c0 + tx;
  float cs = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int n = n0 + ty + 4 * i;
    float v = 0.f;
    if (n < N && c < C) {
      const size_t at = ((size_t)bi * N + n) * C + c;
      v = a0[at];
      if (a1) v += a1[at];
    }
    if (n < N && c < ldr) rows[((size_t)bi * N + n) * ldr + c] = v;
    cs += v;
  }
  if (colpart) {
    red[ty][tx] = cs;
    __syncthreads();
    if (ty == 0 && c < ldr)
      colpart[((size_t)bi * gridDim.x + blockIdx.x) * ldr + c] =
          (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
  }
}
int pm_rows_in(int b, int n, int c, int ldr, const float *a0, const float *a1, float *rows,
               float *zero, int nzero, float *colpart, hipStream_t stream) {
  if (b <= 0 || n <= 0 || c <= 0) return BTR_OK;
  BTR_REQUIRE(a0 && rows && ldr >= c, "pm_rows_in: bad arguments");
  hipLaunchKernelGGL(pm_rows_in_kernel, dim3(cdiv(n, 64), cdiv(ldr, 64), b), dim3(256), 0, stream,
                     n, c, ldr, a0, a1, rows, zero, nzero, colpart);
  return check_launch("pm_rows_in");
}

// (internal.hpp) pm_out with an operand added to the (B, C, N) output
int pm_out_add(int b, int n, int c, int ldy, const float *y, const float *scale,
               const float *shift, int relu, float *out_bcn, float *out_cl, const float *add,
               long long add_bstride, hipStream_t stream) {
  if (b <= 0 || n <= 0 || c <= 0) return BTR_OK;
  BTR_REQUIRE(y && out_bcn && (scale == nullptr) == (shift == nullptr), "pm_out: bad arguments");
  hipLaunchKernelGGL(pm_out_kernel, dim3(cdiv(n, 64), cdiv(c, 64), b), dim3(256), 0, stream, n, c,
                     ldy, y, scale, shift, relu, out_bcn, out_cl, add, add_bstride);
  return check_launch("pm_out");
}

// The compact description bound on this host thread (btr_sac_bind); read by the launchers.
struct HostCompact {
  Compact dev;
  double count = 0.0;  // number of rows of the DENSE evaluation (BatchNorm's N)
  bool on = false;
};
inline HostCompact &host_compact() {
  static thread_local HostCompact hc;
  return hc;
}
inline Compact cur_compact() { return host_compact().on ? host_compact().dev : Compact{}; }
const int *trace_compact_dims() { return host_compact().on ? host_compact().dev.dims : nullptr; }


// the BatchNorm finalisation armed for the next statistics GEMM of this host thread (internal.hpp)
struct HostBnFin {
  BnFin fin{};
  bool on = false;
};
inline HostBnFin &host_bnfin() {
  static thread_local HostBnFin h;
  return h;
}
inline BnFin take_bnfin() {
  HostBnFin &h = host_bnfin();
  if (!h.on) return BnFin{};
  h.on = false;
  return h.fin;
}

// BTR_GEMM=f32: the f32-input MFMA kernels (v_mfma_f32_32x32x2_f32) instead of bf16x6
inline bool gemm_x6() {
  static const bool on = !(getenv("BTR_GEMM") && getenv("BTR_GEMM")[0] == 'f');
  return on;
}
inline bool tn_x6() { return gemm_x6(); }        // the plain weight-gradient GEMM
inline bool tn_pool_x6() { return gemm_x6(); }   // the pooled-gradient / first-layer-recompute ones

template <int W, bool P, bool GP, bool XR>
inline void launch_tn(bool x6, dim3 grid, hipStream_t st, const float *g, int ldg, const float *x,
                      int ldx, int rows, int n, int k, const float *pa, const float *pb, int rpc,
                      float *pw, const unsigned char *garg, const float *gdcl,
                      const float *galpha, const float *gbeta, int ssh, const float *xw0) {
  if (x6)
    hipLaunchKernelGGL((gemm_tn_x6_kernel<W, P, GP, XR>), grid, dim3(256), 0, st, g, ldg, x, ldx,
                       rows, n, k, pa, pb, rpc, pw, garg, gdcl, galpha, gbeta, ssh, xw0,
                       cur_compact());
  else
    hipLaunchKernelGGL((gemm_tn_kernel<W, P, GP, XR>), grid, dim3(256), 0, st, g, ldg, x, ldx,
                       rows, n, k, pa, pb, rpc, pw, garg, gdcl, galpha, gbeta, ssh, xw0,
                       cur_compact());
}

// (internal.hpp) may a statistics GEMM over `rows` rows finalise its BatchNorm itself?
bool bnfin_rows_ok(long long rows) {
  static const bool off = getenv("BTR_BN_TICKET") && getenv("BTR_BN_TICKET")[0] == '0';
  static const long long max_rows =
      getenv("BTR_BN_TICKET_MAX_ROWS") ? atoll(getenv("BTR_BN_TICKET_MAX_ROWS")) : 2048;
  return !off && gemm_x6() && rows <= max_rows;
}
bool bnfin_arm(const BnFin &fin, long long rows) {
  if (!fin.ticket || !bnfin_rows_ok(rows)) return false;
  HostBnFin &h = host_bnfin();
  h.fin = fin;
  h.fin.fence = 0;   // (1: full fences around the ticket; same results, r05 A/B: +9 - 12 us per GEMM)
  h.on = true;
  return true;
}

}  // namespace btr

// sa_fwd_stream_kernel for the NT entry points below: true when it took the launch.
// BTR_FWD_STREAM=0: never.  Rows below kStreamMinRows stay on gemm_nt_kernel (its BatchNorm
// ticket path covers the tiny layers, and a streaming grid of few steps gains nothing).
namespace btr {
constexpr int kStreamMinRows = 16384;
static bool stream_ok(int rows, int n, int k, int lda_ok, bool fin_armed) {
  const char *e = getenv("BTR_FWD_STREAM");   // (read per call: tests and A/B runs toggle it)
  const bool off = e && e[0] == '0';
  return !off && gemm_x6() && !fin_armed && lda_ok && rows >= kStreamMinRows && n % 4 == 0 &&
         k % 4 == 0 && n <= 256 && k <= 128 && !host_compact().dev.kz;
}
template <int BR, int KMAX, int PRO, bool STATS, int PS>
static void launch_stream(int gx, hipStream_t st, StreamArgs &a) {
  a.rows_per_chunk = cdiv(cdiv(a.R, gx), BR) * BR;
  hipLaunchKernelGGL((sa_fwd_stream_kernel<BR, KMAX, PRO, STATS, PS>), dim3(gx << a.hs), dim3(256),
                     0, st, a, cur_compact());
}
// pro: 0 / 1 / 3;  ps: 0 / 8.  The instantiated combinations are the layers' (see the kernel).
static bool try_stream(int rows, int n, int k, const float *A, int lda, const float *W, int ldw,
                       float *C, int ldc, const float *pa, const float *pb, const float *w0,
                       float *part, int ps, const float *gamma, float *gext, unsigned char *aext,
                       int pro, hipStream_t st) {
  StreamArgs a{};
  a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.R = rows; a.N = n; a.K = k;
  a.pa = pa; a.pb = pb; a.w0 = w0; a.part = part; a.gsign = gamma; a.gext = gext; a.aext = aext;
  const int gx = btr_sa_gemm_grid(rows);
  const bool stats = part != nullptr;
  // n > 128 (the 256-wide pooled layers): two 128-column slabs per row chunk -- with the
  // 32-row steps only (four column tiles per workgroup)
  a.hs = n > 128 ? 1 : 0;
  if (ps == 8) {
    if (pro == 1 && stats && n > 64 && k <= 64) { launch_stream<32, 64, 1, true, 8>(gx, st, a); return true; }
    if (pro == 1 && stats && n > 64) { launch_stream<32, 128, 1, true, 8>(gx, st, a); return true; }
    return false;
  }
  if (ps == 16) {
    if (pro == 1 && stats && n > 64 && k > 64) { launch_stream<32, 128, 1, true, 16>(gx, st, a); return true; }
    return false;
  }
  if (ps != 0) return false;
  if (pro == 3) {
    if (stats && n <= 64 && k <= 64) { launch_stream<64, 64, 3, true, 0>(gx, st, a); return true; }
    return false;
  }
  if (pro == 1 && stats) {
    if (n <= 64 && k <= 64) { launch_stream<64, 64, 1, true, 0>(gx, st, a); return true; }
    if (n > 64 && k <= 64) { launch_stream<32, 64, 1, true, 0>(gx, st, a); return true; }
    if (n > 64) { launch_stream<32, 128, 1, true, 0>(gx, st, a); return true; }
    return false;
  }
  if (pro == 0 && !stats) {   // input gradients
    if (n <= 64 && k <= 64) { launch_stream<64, 64, 0, false, 0>(gx, st, a); return true; }
    if (n > 64 && k <= 64) { launch_stream<32, 64, 0, false, 0>(gx, st, a); return true; }
    if (n > 64) { launch_stream<32, 128, 0, false, 0>(gx, st, a); return true; }
    return false;
  }
  return false;
}
}  // namespace btr

using namespace btr;

extern "C" {

// Rows of X0: (b*M+m)*S+s; ldx >= use_xyz*3 + C, multiple of 4.  feats_cl may be NULL (C=0).
int btr_sa_gather(int b, int n, int m, int s, int c, int ldx, int use_xyz, float radius_div,
                  const float *xyz, const float *new_xyz, const float *feats_cl, const int *idx,
                  float *x0, btr_stream_t stream) {
  if (b <= 0 || m <= 0 || s <= 0) return BTR_OK;
  BTR_REQUIRE(xyz && new_xyz && idx && x0 && ldx >= (use_xyz ? 3 : 0) + c && ldx % 4 == 0,
              "sa_gather: bad arguments (ldx=%d, c=%d)", ldx, c);
  BTR_REQUIRE(c == 0 || feats_cl, "sa_gather: features missing");
  const int tpr = std::min(64, ldx);
  const int rpb = 256 / tpr;
  const long long ms = (long long)m * s;
  const int gx = (int)std::min<long long>(cdiv(ms, rpb), 4096);
  const float inv = radius_div != 0.f ? 1.0f / radius_div : 1.0f;
  hipLaunchKernelGGL(sa_gather_kernel, dim3(gx, b), dim3(256), 0, as_stream(stream), n, m, s, c,
                     ldx, use_xyz, inv, xyz, new_xyz, feats_cl, idx, x0);
  return check_launch("sa_gather");
}

static int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// Number of workgroups (= rows of the `part` buffer) btr_sa_gemm_nt uses along rows.
// One round of resident workgroups per launch, two row chunks per CU for the NT kernels (round 4,
// same box: two / four rounds of smaller chunks helped the forward beside the pyramid by 34 us and
// cost the backward more in partial sums, 4.58 -> 4.65 / 4.83 ms; three / four chunks per CU 4.11 /
// 4.04 vs 3.98 ms).
static constexpr int grid_rounds() { return 1; }
static constexpr int gemm_wgs_per_cu() { return 2; }
int btr_sa_gemm_grid(int rows) {
  return std::max(1, std::min(cdiv(rows, kBM),
                              gemm_wgs_per_cu() * grid_cus() * grid_rounds()));
}

// Whether the last layer's GEMM can emit the per-group extrema itself (pooling epilogue):
// 128-column tiles (n > 64) and whole groups of 16 / 32 / 64 rows per wave.
int btr_sa_gemm_nt_poolfwd_supported(int rows, int n, int s) {
  return n > 64 && (s == 8 || s == 16 || s == 32 || s == 64) && rows > 0 && rows % s == 0;
}

// C[rows][n] = f(A)[rows][k] . W[n][k]^T;  pa/pb != NULL: f = relu(pa*y+pb) per k;
// part != NULL: per-workgroup column sums / sums of squares -> part[grid][2][n].
int btr_sa_gemm_nt(int rows, int n, int k, const float *a, int lda, const float *w, int ldw,
                   float *c, int ldc, const float *pa, const float *pb, float *part,
                   btr_stream_t stream) {
  GemmTrace trace_((hipStream_t)stream);
  if (rows <= 0 || n <= 0) return BTR_OK;
  trace_.work(rows, 2.0 * n * k, 4.0 * (k + (c ? n : 0)), 4.0 * n * k, true);
  BTR_REQUIRE(a && w && (c || part) && k > 0 && k % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0,
              "sa_gemm_nt: k=%d lda=%d ldw=%d must be multiples of 4", k, lda, ldw);
  BTR_REQUIRE((pa == nullptr) == (pb == nullptr), "sa_gemm_nt: pa/pb must come together");
  BTR_REQUIRE(pa == nullptr || k <= kMaxK, "sa_gemm_nt: k=%d > %d with a prologue", k, kMaxK);
  const int gx = btr_sa_gemm_grid(rows);
  hipStream_t s = as_stream(stream);
  const bool pro = pa != nullptr, st = part != nullptr;
  const BnFin fin = take_bnfin();   // (armed by the caller: finalisation inside this launch)
  if (c && stream_ok(rows, n, k, lda % 4 == 0, fin.ticket != nullptr) &&
      try_stream(rows, n, k, a, lda, w, ldw, c, ldc, pa, pb, nullptr, part, 0, nullptr, nullptr,
                 nullptr, pro ? 1 : 0, s))
    return check_launch("sa_gemm_nt(stream)");
#define BTR_GEMM_MM(BN, P, S, MM)                                                            \
  hipLaunchKernelGGL((gemm_nt_kernel<BN, P, S, 0, kBM, false, MM>), dim3(gx, cdiv(n, BN)),    \
                     dim3(256), 0, s, a, lda, w, ldw, c, ldc, rows, n, k, pa, pb, part,       \
                     (const unsigned char *)nullptr, (const float *)nullptr, 0,                 \
                     (const float *)nullptr, (float *)nullptr, (unsigned char *)nullptr,        \
                     cur_compact(), fin)
#define BTR_GEMM(BN, P, S)            \
  do {                                \
    if (gemm_x6()) BTR_GEMM_MM(BN, P, S, 1); \
    else BTR_GEMM_MM(BN, P, S, 0);    \
  } while (0)
  if (n <= 64) {
    if (pro) { if (st) BTR_GEMM(64, 1, true); else BTR_GEMM(64, 1, false); }
    else     { if (st) BTR_GEMM(64, 0, true); else BTR_GEMM(64, 0, false); }
  } else {
    if (pro) { if (st) BTR_GEMM(128, 1, true); else BTR_GEMM(128, 1, false); }
    else     { if (st) BTR_GEMM(128, 0, true); else BTR_GEMM(128, 0, false); }
  }
#undef BTR_GEMM
#undef BTR_GEMM_MM
  return check_launch("sa_gemm_nt");
}

int btr_sa_bn_finalize(int n, int nblk, double count, float eps, float momentum,
                       const float *part, const float *gamma, const float *beta, float *scale,
                       float *shift, float *mean, float *invstd, float *running_mean,
                       float *running_var, btr_stream_t stream) {
  return bn_finalize_bias(n, nblk, count, eps, momentum, part, gamma, beta, scale, shift, mean,
                          invstd, running_mean, running_var, nullptr, 0, as_stream(stream));
}

// c == NULL is accepted by btr_sa_gemm_nt_poolfwd exactly when this returns 1 (s: 8 for compact
// rows' block extrema, else the group size; no in-kernel BatchNorm finalisation armed)
int btr_sa_gemm_nt_poolfwd_nostore_supported(int rows, int n, int k, int s) {
  if (!(s == 8 || s == 16) || !stream_ok(rows, n, k, true, false)) return 0;
  if (n <= 64) return 0;            // (the instantiated streaming variants: try_stream)
  return s == 8 ? 1 : (k > 64 ? 1 : 0);
}

int btr_sa_gemm_nt_poolfwd(int rows, int n, int k, const float *a, int lda, const float *w,
                           int ldw, float *c, int ldc, const float *pa, const float *pb,
                           float *part, int s, const float *gamma, float *gext,
                           unsigned char *aext, btr_stream_t stream) {
  GemmTrace trace_((hipStream_t)stream);
  if (rows <= 0 || n <= 0) return BTR_OK;
  // (+ the extrema: a value and a position byte per s-row block and column)
  trace_.work(rows, 2.0 * n * k, 4.0 * (k + (c ? n : 0)) + 5.0 * n / (s > 0 ? s : 1),
              4.0 * n * k, true);
  BTR_REQUIRE(a && w && part && pa && pb && gamma && gext && aext && k > 0 &&
                  k % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0,
              "sa_gemm_nt_poolfwd: null pointer or k=%d lda=%d ldw=%d not multiples of 4", k,
              lda, ldw);
  BTR_REQUIRE(btr_sa_gemm_nt_poolfwd_supported(rows, n, s) && k <= kMaxK,
              "sa_gemm_nt_poolfwd: n=%d / nsample=%d / k=%d not supported", n, s, k);
  const int gx = btr_sa_gemm_grid(rows);
  hipStream_t st = as_stream(stream);
  const BnFin fin = take_bnfin();
  if ((s == 8 || s == 16) && stream_ok(rows, n, k, lda % 4 == 0, fin.ticket != nullptr) &&
      try_stream(rows, n, k, a, lda, w, ldw, c, ldc, pa, pb, nullptr, part, s, gamma, gext, aext, 1,
                 st))
    return check_launch("sa_gemm_nt_poolfwd(stream)");
  // c == NULL (statistics + extrema only, the Gram-form backward needs no Y_l): the streaming
  // kernel's option -- btr_sa_gemm_nt_poolfwd_nostore_supported() says when it runs
  BTR_REQUIRE(c, "sa_gemm_nt_poolfwd: no output matrix outside the streaming kernel's shapes");
#define BTR_GEMM_MM(PS, MM)                                                                   \
  hipLaunchKernelGGL((gemm_nt_kernel<128, 1, true, PS, kBM, false, MM>), dim3(gx, cdiv(n, 128)), \
                     dim3(256), 0, st, a, lda, w, ldw, c, ldc, rows, n, k, pa, pb, part,        \
                     (const unsigned char *)nullptr, (const float *)nullptr, 0, gamma, gext,    \
                     aext, cur_compact(), fin)
#define BTR_GEMM(PS)                  \
  do {                                \
    if (gemm_x6()) BTR_GEMM_MM(PS, 1); \
    else BTR_GEMM_MM(PS, 0);          \
  } while (0)
  if (s == 8) BTR_GEMM(8);
  else if (s == 16) BTR_GEMM(16);
  else if (s == 32) BTR_GEMM(32);
  else BTR_GEMM(64);
#undef BTR_GEMM
#undef BTR_GEMM_MM
  return check_launch("sa_gemm_nt_poolfwd");
}

// sa_pool_tile_kernel takes the pools whose groups tile by 32 (BTR_POOL_TILE=0: the
// thread-per-element kernels; read per call, the tests compare the two)
// ... four channels per thread (c % 4 == 0; BTR_POOL_TILE4=0: the one-channel tile kernel)
static bool pool_tile4(int c) {
  const char *e = getenv("BTR_POOL_TILE4");
  return c % 4 == 0 && !(e && e[0] == '0');
}
static bool pool_tiled(int m) {
  const char *e = getenv("BTR_POOL_TILE");
  return m % 32 == 0 && !(e && e[0] == '0');
}

// ywin (b, m, c) or NULL: the pre-BN value of the arg-max row -- what the pooled BatchNorm's
// backward needs of Y_l when the forward did not store it (btr_sa_pool_bwd_coef with ldy == 0)
int btr_sa_pool_fin_y(int b, int m, int c, const float *gext, const unsigned char *aext,
                      const float *scale, const float *shift, float *out, float *out_cl,
                      unsigned char *arg, float *ywin, btr_stream_t stream) {
  const long long groups = (long long)b * m;
  if (groups <= 0 || c <= 0) return BTR_OK;
  BTR_REQUIRE(gext && aext && scale && shift && out && arg, "sa_pool_fin: null pointer");
  if (pool_tiled(m) && pool_tile4(c))
    hipLaunchKernelGGL(sa_pool_tile4_kernel, dim3((unsigned)(groups / 32), cdiv(c, 128)), dim3(256),
                       0, as_stream(stream), m, c, gext, aext, (const int *)nullptr, scale, shift,
                       out, out_cl, arg, ywin);
  else if (pool_tiled(m))
    hipLaunchKernelGGL(sa_pool_tile_kernel, dim3((unsigned)(groups / 32), cdiv(c, 32)), dim3(256),
                       0, as_stream(stream), m, c, gext, aext, (const int *)nullptr, scale, shift,
                       out, out_cl, arg, ywin);
  else
    hipLaunchKernelGGL(sa_pool_fin_kernel, dim3(cdiv(groups * c, 256)), dim3(256), 0,
                       as_stream(stream), m, c, gext, aext, scale, shift, out, out_cl, arg,
                       groups, ywin);
  return check_launch("sa_pool_fin");
}
int btr_sa_pool_fin(int b, int m, int c, const float *gext, const unsigned char *aext,
                    const float *scale, const float *shift, float *out, float *out_cl,
                    unsigned char *arg, btr_stream_t stream) {
  return btr_sa_pool_fin_y(b, m, c, gext, aext, scale, shift, out, out_cl, arg, nullptr, stream);
}

int btr_sa_pool(int b, int m, int s, int c, int ldy, const float *y, const float *scale,
                const float *shift, float *out, float *out_cl, unsigned char *arg,
                btr_stream_t stream) {
  const long long groups = (long long)b * m;
  if (groups <= 0 || c <= 0) return BTR_OK;
  BTR_REQUIRE(s <= 255, "sa_pool: nsample %d > 255", s);
  if (c % 4 == 0 && ldy % 4 == 0)
    hipLaunchKernelGGL(sa_pool4_kernel, dim3(cdiv(groups * (c / 4), 256)), dim3(256), 0,
                       as_stream(stream), m, s, c, ldy, y, scale, shift, out, out_cl, arg,
                       groups);
  else
    hipLaunchKernelGGL(sa_pool_kernel, dim3(cdiv(groups * c, 256)), dim3(256), 0,
                       as_stream(stream), m, s, c, ldy, y, scale, shift, out, out_cl, arg,
                       groups);
  return check_launch("sa_pool");
}

int btr_sa_pool_bwd(int b, int m, int s, int c, int ldy, float *y, const float *dout,
                    const float *out, const unsigned char *arg, const float *mean,
                    const float *invstd, const float *scale, float *part /*[1024][2][c]*/,
                    float *m1, float *m2, float *dgamma, float *dbeta, btr_stream_t stream) {
  const long long groups = (long long)b * m;
  if (groups <= 0 || c <= 0) return BTR_OK;
  hipStream_t st = as_stream(stream);
  const int nblk = (int)std::min<long long>(groups, 1024);
  hipLaunchKernelGGL(sa_pool_bwd_stats_kernel, dim3(nblk), dim3(256), 0, st, m, s, c, ldy, y,
                     dout, out, arg, mean, invstd, groups, part);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(c, kRedCh)), dim3(256), 0, st, c, nblk,
                     (double)groups * s, part, m1, m2, dgamma, dbeta);
  hipLaunchKernelGGL(sa_pool_bwd_apply_kernel, dim3(cdiv(groups * c, 256)), dim3(256), 0, st, m,
                     s, c, ldy, y, dout, out, arg, mean, invstd, scale, m1, m2, groups);
  return check_launch("sa_pool_bwd");
}

// Same statistics as btr_sa_pool_bwd, but instead of writing the dense dY over y it returns
// the coefficients (dcl [b*m][c], alpha [c], beta [c]) from which btr_sa_gemm_nt_pool /
// btr_sa_gemm_tn_pool form dY inside their operand staging.  y is left untouched.
int btr_sa_pool_bwd_coef(int b, int m, int s, int c, int ldy, const float *y, const float *dout,
                         const float *out, const unsigned char *arg, const float *mean,
                         const float *invstd, const float *scale, const float *shift,
                         float *part /*[1024][2][c]*/, float *m1, float *m2, float *dgamma,
                         float *dbeta, float *dcl, float *alpha, float *beta,
                         btr_stream_t stream) {
  const long long groups = (long long)b * m;
  if (groups <= 0 || c <= 0) return BTR_OK;
  hipStream_t st = as_stream(stream);
  const long long tiles = (long long)b * cdiv(m, 64);
  // ldy == 0: y holds only the arg-max rows' values (b, m, c) -- the tile kernel's operand
  BTR_REQUIRE(ldy != 0 || (shift != nullptr && tiles <= 1024 && c % 4 == 0 && b < 65536),
              "sa_pool_bwd_coef: arg-max values instead of Y only for c %% 4 == 0, <= 1024 tiles");
  if (shift != nullptr && tiles <= 1024 && c % 4 == 0 && b < 65536) {
    hipLaunchKernelGGL(sa_pool_bwd_tile_kernel, dim3(cdiv(m, 64), cdiv(c, 16), b), dim3(256), 0,
                       st, m, s, c, ldy, y, dout, out, arg, mean, invstd, scale, shift, part, dcl,
                       cur_compact().goff);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(c, kRedCh)), dim3(256), 0, st, c,
                       (int)tiles, (double)groups * s, part, m1, m2, dgamma, dbeta, scale, mean,
                       invstd, alpha, beta);
    return check_launch("sa_pool_bwd_coef(tile)");
  }
  const int nblk = (int)std::min<long long>(groups, 1024);
  hipLaunchKernelGGL(sa_pool_bwd_stats_kernel, dim3(nblk), dim3(256), 0, st, m, s, c, ldy, y,
                     dout, out, arg, mean, invstd, groups, part);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(c, kRedCh)), dim3(256), 0, st, c, nblk,
                     (double)groups * s, part, m1, m2, dgamma, dbeta);
  hipLaunchKernelGGL(sa_pool_coef_kernel, dim3(cdiv(groups * c, 256)), dim3(256), 0, st, m, c,
                     groups, dout, out, scale, mean, invstd, m1, m2, dcl, alpha, beta);
  return check_launch("sa_pool_bwd_coef");
}

// dX[rows][n] = dY . Wt^T with dY formed from (y, arg, dcl, alpha, beta) on the fly; k = the
// pooled layer's channel count (row length of arg / dcl), s = samples per group.
int btr_sa_gemm_nt_pool(int rows, int n, int k, const float *y, int ldy, const float *w, int ldw,
                        float *c, int ldc, int s, const unsigned char *arg, const float *dcl,
                        const float *alpha, const float *beta, btr_stream_t stream) {
  GemmTrace trace_((hipStream_t)stream);
  if (rows <= 0 || n <= 0) return BTR_OK;
  trace_.work(rows, 2.0 * n * k, 4.0 * (k + n), 4.0 * n * k, true);
  BTR_REQUIRE(y && w && c && arg && dcl && alpha && beta && s > 0 && k > 0 && k % 4 == 0 &&
                  ldy % 4 == 0 && ldw % 4 == 0,
              "sa_gemm_nt_pool: bad arguments (k=%d ldy=%d ldw=%d s=%d)", k, ldy, ldw, s);
  BTR_REQUIRE(host_compact().on || s == 16 || s == 32 || s == 64 || s == 128,
              "sa_gemm_nt_pool: nsample %d must be 16, 32, 64 or 128", s);
  BTR_REQUIRE(k <= kMaxK, "sa_gemm_nt_pool: k=%d > %d", k, kMaxK);
  const int ssh = ilog2(s);
  const int gx = btr_sa_gemm_grid(rows);
  hipStream_t st = as_stream(stream);
#define BTR_NTP(BN, MM)                                                                       \
  hipLaunchKernelGGL((gemm_nt_kernel<BN, 2, false, 0, kBM, false, MM>), dim3(gx, cdiv(n, BN)), \
                     dim3(256), 0, st, y, ldy, w, ldw, c, ldc, rows, n, k, alpha, beta,        \
                     (float *)nullptr, arg, dcl, ssh, (const float *)nullptr, (float *)nullptr, \
                     (unsigned char *)nullptr, cur_compact())
  if (n <= 64) {
    if (gemm_x6()) BTR_NTP(64, 1); else BTR_NTP(64, 0);
  } else {
    if (gemm_x6()) BTR_NTP(128, 1); else BTR_NTP(128, 0);
  }
#undef BTR_NTP
  return check_launch("sa_gemm_nt_pool");
}

// Number of row chunks btr_sa_gemm_tn uses; the caller provides pw[chunks][n][k] floats.
static int tn_tile_n(int n) { return n >= 128 ? 128 : 64; }

int btr_sa_gemm_tn_chunks(int rows, int n, int k) {
  const int tiles = cdiv(n, tn_tile_n(n)) * cdiv(k, 64);
  // two workgroups per CU in flight (four until the weight gradients moved to the caller's
  // stream: half the partials to write and reduce, same box 4.10 -> 4.02 ms per step; three 4.00,
  // one 4.54 -- the fused backward's chunk count is capped by this one)
  constexpr int per_cu = 2;
  const int flight = per_cu * grid_cus() * grid_rounds();
  int chunks = std::max(1, std::min(flight / tiles, flight));
  // a workgroup (alone on its CU in these launches: one wave per SIMD, nothing to overlap with)
  // spends ~1.3 us per 32-row step: few-row GEMMs (the 1024-row decoder / head layers: 8 steps
  // per workgroup, 14 us) get chunks of down to 64 rows as long as the partials stay small
  // (<= 16 MB to write and reduce)
  constexpr int min_rows = 64;
  const long long by_mem = std::max<long long>(4, (4ll << 20) / ((long long)n * k));
  const long long by_rows = std::max(1, rows / std::max(min_rows, 32));
  const long long cap = std::min<long long>(by_rows, std::max<long long>(by_mem, rows / 256));
  chunks = (int)std::min<long long>(chunks, std::max<long long>(1, cap));
  return chunks;
}

// ---- first-layer recompute (input rows of <= 4 columns, x0 with leading dimension 4; w0 is
// the first layer's weight [k][4]).  See rc_y4.
// C[rows][n] = relu(pa*(x0.w0^T)+pb) . W^T  (the second layer's forward; k = first layer width)
int btr_sa_gemm_nt_rc(int rows, int n, int k, const float *x0, const float *w0, const float *w,
                      int ldw, float *c, int ldc, const float *pa, const float *pb, float *part,
                      btr_stream_t stream) {
  GemmTrace trace_((hipStream_t)stream);
  if (rows <= 0 || n <= 0) return BTR_OK;
  // (the first layer rebuilt from the 4-column input row: 2 * 4 * k flops per row more)
  trace_.work(rows, 2.0 * n * k + 8.0 * k, 16.0 + 4.0 * n, 4.0 * n * k, true);
  BTR_REQUIRE(x0 && w0 && w && c && pa && pb && k > 0 && k % 4 == 0 && ldw % 4 == 0 &&
                  k <= kMaxK,
              "sa_gemm_nt_rc: null pointer or k=%d ldw=%d not multiples of 4", k, ldw);
  const int gx = btr_sa_gemm_grid(rows);
  hipStream_t st = as_stream(stream);
  const BnFin fin = take_bnfin();
  if (stream_ok(rows, n, k, true, fin.ticket != nullptr) &&
      try_stream(rows, n, k, x0, 4, w, ldw, c, ldc, pa, pb, w0, part, 0, nullptr, nullptr, nullptr,
                 3, st))
    return check_launch("sa_gemm_nt_rc(stream)");
#define BTR_NTRC_MM(BN, S, MM)                                                                \
  hipLaunchKernelGGL((gemm_nt_kernel<BN, 3, S, 0, kBM, false, MM>), dim3(gx, cdiv(n, BN)),     \
                     dim3(256), 0, st, x0, 4, w, ldw, c, ldc, rows, n, k, pa, pb, part,        \
                     (const unsigned char *)nullptr, w0, 0, (const float *)nullptr,           \
                     (float *)nullptr, (unsigned char *)nullptr, cur_compact(), fin)
#define BTR_NTRC(BN, S)                \
  do {                                 \
    if (gemm_x6()) BTR_NTRC_MM(BN, S, 1); \
    else BTR_NTRC_MM(BN, S, 0);        \
  } while (0)
  if (n <= 64) { if (part) BTR_NTRC(64, true); else BTR_NTRC(64, false); }
  else         { if (part) BTR_NTRC(128, true); else BTR_NTRC(128, false); }
#undef BTR_NTRC
#undef BTR_NTRC_MM
  return check_launch("sa_gemm_nt_rc");
}

// dW[n][k] = sum_r G[r][n] * relu(pa*(x0.w0^T)+pb)[r][k]  (the second layer's weight gradient)
int btr_sa_gemm_tn_rc(int rows, int n, int k, const float *g, int ldg, const float *x0,
                      const float *w0, const float *pa, const float *pb, float *pw, float *dw,
                      btr_stream_t stream) {
  GemmTrace trace_((hipStream_t)stream);
  if (n <= 0 || k <= 0) return BTR_OK;
  trace_.work(rows, 2.0 * n * k + 8.0 * k, 4.0 * n + 16.0, 4.0 * n * k, true);
  BTR_REQUIRE(g && x0 && w0 && pa && pb && pw && dw && ldg % 4 == 0 && n % 4 == 0 && k % 4 == 0,
              "sa_gemm_tn_rc: sizes must be multiples of 4 (n=%d k=%d)", n, k);
  hipStream_t st = as_stream(stream);
  const int chunks = btr_sa_gemm_tn_chunks(rows, n, k);
  const int rpc = cdiv(cdiv(rows, chunks), 32) * 32;
  const int tn = tn_tile_n(n);
  const dim3 grid(cdiv(n, tn), cdiv(k, 64), chunks);
  if (tn == 128)
    launch_tn<4, true, false, true>(tn_pool_x6(), grid, st, g, ldg, x0, 4, rows, n, k, pa, pb, rpc,
                                    pw, nullptr, nullptr, nullptr, nullptr, 0, w0);
  else
    launch_tn<2, true, false, true>(tn_pool_x6(), grid, st, g, ldg, x0, 4, rows, n, k, pa, pb, rpc,
                                    pw, nullptr, nullptr, nullptr, nullptr, 0, w0);
  reduce_chunks_launch(n * k, chunks, pw, dw, st);
  return check_launch("sa_gemm_tn_rc");
}

// Workgroups (= rows of pw [blocks][c][4]) btr_sa_bn_relu_bwd_rc uses for the weight gradient.
int btr_sa_rc_wgrad_blocks(long long rows, int c) {
  return (int)std::max<long long>(1, std::min<long long>(cdiv(rows, 256 / (c / 4)), 512));
}

// First layer's BN+ReLU backward with y0 recomputed, fused with its weight gradient: g
// [rows][c] (gradient w.r.t. the post-ReLU activation) is only read; outputs dgamma, dbeta
// (and m1, m2) and dw0 [c][4] = dY0^T . x0.  part: [512][2][c], pw: [blocks][c][4] floats.
int btr_sa_bn_relu_bwd_rc(long long rows, int c, int ldg, const float *g, const float *x0,
                          const float *w0, const float *scale, const float *shift,
                          const float *mean, const float *invstd, float *part, float *m1,
                          float *m2, float *dgamma, float *dbeta, float *pw, float *dw0,
                          btr_stream_t stream) {
  if (rows <= 0 || c <= 0) return BTR_OK;
  BTR_REQUIRE(c <= kBnBwdMaxC && c % 4 == 0 && ldg % 4 == 0,
              "sa_bn_relu_bwd_rc: %d channels must be a multiple of 4 and <= %d", c, kBnBwdMaxC);
  hipStream_t st = as_stream(stream);
  const int nblk = btr_sa_rc_wgrad_blocks(rows, c);
  const double count = host_compact().on ? host_compact().count : (double)rows;
  hipLaunchKernelGGL(bn_relu_bwd_stats_rc_kernel, dim3(nblk), dim3(256), 0, st, rows, c, ldg, g,
                     x0, w0, scale, shift, mean, invstd, part, cur_compact());
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(c, kRedCh)), dim3(256), 0, st, c, nblk,
                     count, part, m1, m2, dgamma, dbeta);
  hipLaunchKernelGGL(bn_relu_bwd_wgrad0_rc_kernel, dim3(nblk), dim3(256), 0, st, rows, c, ldg, g,
                     x0, w0, scale, shift, mean, invstd, m1, m2, pw, cur_compact());
  hipLaunchKernelGGL((reduce_chunks_kernel<16, 16>), dim3(cdiv(c * 4, 16)), dim3(256), 0, st,
                     c * 4, nblk, pw, dw0);
  return check_launch("sa_bn_relu_bwd_rc");
}

// ---- single-launch inference set-abstraction layer (sa_eval_fused_kernel)
int btr_sa_eval_fused_supported(int s, int c, int use_xyz, int c1, int c2, int c3) {
  static const bool off = getenv("BTR_EVAL_FUSED") && getenv("BTR_EVAL_FUSED")[0] == '0';
  const int k0 = (use_xyz ? 3 : 0) + c;
  return !off && gemm_x6() && (s == 16 || s == 32 || s == 64) && k0 >= 1 && k0 <= 4 && c1 >= 4 &&
         c1 <= 64 && c2 >= 4 && c2 <= 64 && c3 >= 4 && c3 <= 128 && c1 % 4 == 0 && c2 % 4 == 0;
}

int btr_sa_eval_fused(int b, int n, int m, int s, int c, int use_xyz, float radius_div,
                      const float *xyz, const float *new_xyz, const float *feats_cl,
                      const int *idx, int c1, int c2, int c3, const float *w0, const float *w1,
                      int ld1, const float *w2, int ld2, const float *a0, const float *b0,
                      const float *a1, const float *b1, const float *a2, const float *b2,
                      float *out, float *out_cl, btr_stream_t stream) {
  if (b <= 0 || m <= 0) return BTR_OK;
  BTR_REQUIRE(btr_sa_eval_fused_supported(s, c, use_xyz, c1, c2, c3),
              "sa_eval_fused: nsample %d, %d + %d input columns, widths %d %d %d", s,
              use_xyz ? 3 : 0, c, c1, c2, c3);
  BTR_REQUIRE(xyz && new_xyz && idx && w0 && w1 && w2 && a0 && b0 && a1 && b1 && a2 && b2 && out &&
                  out_cl && (c == 0 || feats_cl) && ld1 % 4 == 0 && ld2 % 4 == 0 && ld1 >= c1 &&
                  ld2 >= c2,
              "sa_eval_fused: null pointer or bad leading dimension");
  EvalArgs a{};
  a.xyz = xyz; a.new_xyz = new_xyz; a.feat_cl = feats_cl; a.idx = idx;
  a.w0 = w0; a.w1 = w1; a.w2 = w2; a.ld1 = ld1; a.ld2 = ld2;
  a.a0 = a0; a.b0 = b0; a.a1 = a1; a.b1 = b1; a.a2 = a2; a.b2 = b2;
  a.out = out; a.out_cl = out_cl;
  a.B = b; a.N = n; a.M = m; a.S = s; a.C = c; a.use_xyz = use_xyz; a.c1 = c1; a.c2 = c2; a.c3 = c3;
  a.inv_radius = radius_div != 0.f ? 1.0f / radius_div : 1.0f;
  const long long steps = ((long long)b * m * s + 63) / 64;
  const int wgs = (int)std::min<long long>(steps, 1024);
  a.steps_per_wg = (int)((steps + wgs - 1) / wgs);
  hipLaunchKernelGGL(sa_eval_fused_kernel, dim3(cdiv((int)steps, a.steps_per_wg)), dim3(256), 0,
                     as_stream(stream), a);
  return check_launch("sa_eval_fused");
}

// btr_sa_bn_relu_bwd_rc minus its statistics pass: the sums come from btr_sa_bwd_fused.
int btr_sa_bn_relu_bwd_rc_apply(long long rows, int c, int ldg, const float *g, const float *x0,
                                const float *w0, const float *scale, const float *shift,
                                const float *mean, const float *invstd, const float *m1,
                                const float *m2, float *pw, float *dw0, btr_stream_t stream) {
  if (rows <= 0 || c <= 0) return BTR_OK;
  BTR_REQUIRE(c <= kBnBwdMaxC && c % 4 == 0 && ldg % 4 == 0,
              "sa_bn_relu_bwd_rc_apply: %d channels must be a multiple of 4 and <= %d", c,
              kBnBwdMaxC);
  hipStream_t st = as_stream(stream);
  const int nblk = btr_sa_rc_wgrad_blocks(rows, c);
  hipLaunchKernelGGL(bn_relu_bwd_wgrad0_rc_kernel, dim3(nblk), dim3(256), 0, st, rows, c, ldg, g,
                     x0, w0, scale, shift, mean, invstd, m1, m2, pw, cur_compact());
  hipLaunchKernelGGL((reduce_chunks_kernel<16, 16>), dim3(cdiv(c * 4, 16)), dim3(256), 0, st,
                     c * 4, nblk, pw, dw0);
  return check_launch("sa_bn_relu_bwd_rc_apply");
}

// In place: g (gradient w.r.t. the post-ReLU activation of a hidden layer) -> gradient w.r.t.
// the layer's pre-BN output y; also dgamma/dbeta.  part: [1024][2][c] floats.
int btr_sa_bn_relu_bwd(long long rows, int c, int ld, float *g, const float *y,
                       const float *scale, const float *shift, const float *mean,
                       const float *invstd, float *part, float *m1, float *m2, float *dgamma,
                       float *dbeta, btr_stream_t stream) {
  const int rc = btr_sa_bn_relu_bwd_sums(rows, c, ld, g, y, scale, shift, mean, invstd, part, m1,
                                         m2, dgamma, dbeta, stream);
  if (rc != BTR_OK) return rc;
  return btr_sa_bn_relu_bwd_apply(rows, c, ld, g, y, scale, shift, mean, invstd, m1, m2, stream);
}

// The two halves of btr_sa_bn_relu_bwd: the statistics pass + finalisation (g is only read), and
// the in-place apply pass.  A caller that hands (g, y, m1, m2) to btr_sa_bwd_fused -- which
// applies BatchNorm's backward while it stages its operand -- only needs the first.
int btr_sa_bn_relu_bwd_sums(long long rows, int c, int ld, const float *g, const float *y,
                            const float *scale, const float *shift, const float *mean,
                            const float *invstd, float *part, float *m1, float *m2,
                            float *dgamma, float *dbeta, btr_stream_t stream) {
  if (rows <= 0 || c <= 0) return BTR_OK;
  BTR_REQUIRE(c <= kBnBwdMaxC && c % 4 == 0 && ld % 4 == 0,
              "sa_bn_relu_bwd: %d channels must be a multiple of 4 and <= %d", c, kBnBwdMaxC);
  hipStream_t st = as_stream(stream);
  const int nblk = (int)std::min<long long>(cdiv(rows, 256 / (c / 4)), 1024);
  const double count = host_compact().on ? host_compact().count : (double)rows;
  hipLaunchKernelGGL(bn_relu_bwd_stats_kernel, dim3(nblk), dim3(256), 0, st, rows, c, ld, g, y,
                     scale, shift, mean, invstd, part, cur_compact());
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(c, kRedCh)), dim3(256), 0, st, c, nblk,
                     count, part, m1, m2, dgamma, dbeta);
  return check_launch("sa_bn_relu_bwd_sums");
}

int btr_sa_bn_relu_bwd_apply(long long rows, int c, int ld, float *g, const float *y,
                             const float *scale, const float *shift, const float *mean,
                             const float *invstd, const float *m1, const float *m2,
                             btr_stream_t stream) {
  if (rows <= 0 || c <= 0) return BTR_OK;
  BTR_REQUIRE(c <= kBnBwdMaxC && c % 4 == 0 && ld % 4 == 0,
              "sa_bn_relu_bwd: %d channels must be a multiple of 4 and <= %d", c, kBnBwdMaxC);
  hipStream_t st = as_stream(stream);
  const int gx = (int)std::min<long long>(cdiv(rows * (c / 4), 256), 256 * 16);
  hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3(gx), dim3(256), 0, st, rows, c, ld, g, y,
                     scale, shift, mean, invstd, m1, m2, cur_compact());
  return check_launch("sa_bn_relu_bwd_apply");
}

// ---- a hidden layer's whole backward as ONE pass (sa_bwd_fused_kernel) ------------------------
// dY_l formed while staging (pooled: arg != NULL, g = the pooled layer's pre-BN output; else
// BatchNorm_l's backward from g = dZ_l, yl = Y_l and the finalised sums of layer l), then
//   dw [n][k]    = dY_l^T . X_{l-1}          (pw: [btr_sa_bwd_fused_chunks][n][k] partials)
//   dz [rows][k] = dY_l . W_l                (wt = W_l^T [k][ldw])
//   m1, m2, dgamma, dbeta of BatchNorm_{l-1} from (dz, Y_{l-1})   (spart: [chunks][2][k])
// X_{l-1} = relu(pa * y + pb) with y = x [rows][ldx], or y rebuilt from the 4-column input rows
// x [rows][4] and w0 [k][4] when w0 != NULL (first-layer recompute).
int btr_sa_bwd_fused_supported(int rows, int n, int k) {
  static const bool off = getenv("BTR_BWD_FUSED") && getenv("BTR_BWD_FUSED")[0] == '0';
  // (n > 128: the one-workgroup-per-CU variant)
  return !off && gemm_x6() && rows > 0 && n >= 32 && n <= 256 && n % 4 == 0 && k >= 4 &&
         k <= 512 && k % 4 == 0;
}

// (n > 128 as two launches of the 128-column variant over column slabs instead of one launch of
// the 256-column variant -- 490 registers, one workgroup of four waves per CU: SA2's pooled layer
// moves its 233 MB at 1.2 TB/s -- was measured in round 4: one launch 4.00 ms per step, two
// launches 4.05 (the second slab re-reads X and dZ; with float atomics for the add 4.31).  Removed.)
static int fused_pass_chunks(int rows, int n, int k) {
  // one round of resident workgroups: the CUs x (2 workgroups per CU; the 256-column variant: one)
  // over the 64-wide k blocks -- more chunks only add partials to write and reduce and a second,
  // half-empty round
  const int np = n;
  const int resident = grid_cus() * (np > 128 ? 1 : 2) * grid_rounds();
  const int kblocks = (k + 63) / 64;
  const int want = std::max(32, resident / kblocks);
  return std::max(1, std::min(std::min(want, kFusedMaxChunks * grid_rounds()),
                              btr_sa_gemm_tn_chunks(rows, np, k)));
}
// Rows of `spart` ([.][2][k]) and chunk count `pw` ([.][n][k]) is sized for.
int btr_sa_bwd_fused_chunks(int rows, int n, int k) { return fused_pass_chunks(rows, n, k); }

int btr_sa_bwd_fused(int rows, int n, int k, const float *g, int ldg, const float *yl,
                     const float *sc, const float *sh, const float *mu, const float *is,
                     const float *m1l, const float *m2l, int s, const unsigned char *arg,
                     const float *dcl, const float *alpha, const float *beta, const float *x,
                     int ldx, const float *w0, const float *pa, const float *pb,
                     const float *mu_p, const float *is_p, const float *wt, int ldw, float *dz,
                     int ldz, float *pw, float *dw, float *spart, float *m1, float *m2,
                     float *dgamma, float *dbeta, btr_stream_t stream) {
  GemmTrace trace_((hipStream_t)stream);
  BTR_REQUIRE(btr_sa_bwd_fused_supported(rows, n, k), "sa_bwd_fused: shape %d x %d x %d", rows, n,
              k);
  const bool pooled = arg != nullptr;
  // two products per row (weight gradient, input gradient); read: Y_l (pooled) or dZ_l and Y_l,
  // X_{l-1} (or its 4-column source); written: dZ_{l-1}
  trace_.work(rows, 4.0 * n * k, 4.0 * ((pooled ? n : 2 * n) + (w0 ? 4 : k) + k), 8.0 * n * k,
              true);
  BTR_REQUIRE(g && x && pa && pb && mu_p && is_p && wt && dz && pw && dw && spart && m1 && m2 &&
                  dgamma && dbeta && ldg % 4 == 0 && ldw % 4 == 0 && ldz % 4 == 0 &&
                  (w0 || ldx % 4 == 0),
              "sa_bwd_fused: null pointer or unaligned leading dimension");
  BTR_REQUIRE(pooled ? (dcl && alpha && beta && s > 0) : (yl && sc && sh && mu && is && m1l && m2l),
              "sa_bwd_fused: incomplete %s operands", pooled ? "pooled-gradient" : "BatchNorm");
  BTR_REQUIRE(!pooled || host_compact().on || s == 16 || s == 32 || s == 64 || s == 128,
              "sa_bwd_fused: nsample %d must be 16, 32, 64 or 128", s);
  BTR_REQUIRE(n <= 128 || !w0, "sa_bwd_fused: first-layer recompute behind %d > 128 columns", n);
  hipStream_t st = as_stream(stream);
  const int chunks = fused_pass_chunks(rows, n, k);
  FusedArgs a{};
  a.G = g; a.Yl = yl; a.ldg = ldg; a.X = x; a.ldx = ldx; a.R = rows; a.N = n; a.K = k;
  a.rows_per_chunk = cdiv(cdiv(rows, chunks), 32) * 32;
  a.pa = pa; a.pb = pb; a.mu_p = mu_p; a.is_p = is_p; a.xw0 = w0; a.Wt = wt; a.ldw = ldw;
  a.Z = dz; a.ldz = ldz; a.pw = pw; a.spart = spart;
  a.garg = arg; a.gdcl = dcl; a.galpha = alpha; a.gbeta = beta; a.SSH = pooled ? ilog2(s) : 0;
  a.ldt = n;
  a.sc = sc; a.sh = sh; a.mu = mu; a.is = is; a.m1 = m1l; a.m2 = m2l;
  const dim3 grid(1, cdiv(k, 64), chunks);
#define BTR_FUSED(W, GM, XR) \
  hipLaunchKernelGGL((sa_bwd_fused_kernel<W, GM, XR>), grid, dim3(256), 0, st, a, cur_compact())
  if (n > 128) {   // (no first-layer recompute behind a 256-wide layer: w0 is refused above)
    if (pooled) BTR_FUSED(8, 1, false); else BTR_FUSED(8, 2, false);
  } else if (n > 64) {
    if (pooled) { if (w0) BTR_FUSED(4, 1, true); else BTR_FUSED(4, 1, false); }
    else        { if (w0) BTR_FUSED(4, 2, true); else BTR_FUSED(4, 2, false); }
  } else {
    if (pooled) { if (w0) BTR_FUSED(2, 1, true); else BTR_FUSED(2, 1, false); }
    else        { if (w0) BTR_FUSED(2, 2, true); else BTR_FUSED(2, 2, false); }
  }
#undef BTR_FUSED
  const double count = host_compact().on ? host_compact().count : (double)rows;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(k, kRedCh)), dim3(256), 0, st, k, chunks,
                     count, spart, m1, m2, dgamma, dbeta);
  reduce_chunks_launch(n * k, chunks, pw, dw, st);
  return check_launch("sa_bwd_fused");
}

// ---- the pooled layer's backward in Gram form (sa_bwd_gram_kernel): Y_l is not an operand ----
//   dz [rows][k] = w_r (x_r M + c) + sparse_r W_l,  dw [n][k] = diag(alpha) W_l G + beta (x) sx +
//   sparse^T X,  m1 / m2 / dgamma / dbeta of BatchNorm_{l-1};  X = relu(pa * x + pb), x [rows][ldx].
// w = W_l [n][k] (row-major, leading dimension k), wt = W_l^T [k][ldw]; arg / dcl / alpha / beta:
// what btr_sa_pool_bwd_coef left.  pw: [btr_sa_bwd_gram_chunks()][n][k] floats, gscratch:
// btr_sa_bwd_gram_scratch_floats() floats, spart: [chunks][2][k].
// the shapes of the producer / consumer kernel (sa_bwd_gram_ws_kernel)
static bool gram_ws(int n, int k) { return n <= 128 && k <= 64; }
static int gram_chunks(int rows, int n, int k);
int btr_sa_bwd_gram_supported(int rows, int n, int k) {
  const char *e = getenv("BTR_POOL_GRAM");   // (read per call: the tests toggle it)
  const bool off = e && e[0] == '0';
  // the shapes of the producer / consumer kernel (n <= 128, k <= 64: SA1's pooled layer)
  if (off || !btr_sa_bwd_fused_supported(rows, n, k) || !gram_ws(n, k)) return 0;
  // the kernel keeps a chunk's block -> group table in LDS: beyond kFusedMaxChunks chunks of
  // 8 * kGramWsMaxBlocks rows the plan must keep the Y_l-reading form (the same arithmetic as the
  // BTR_REQUIRE in btr_sa_bwd_gram)
  const int chunks = gram_chunks(rows, n, k);
  if (cdiv(cdiv(rows, chunks), 32) * 32 > 8 * kGramWsMaxBlocks) return 0;
  return 1;
}
static int gram_chunks(int rows, int n, int k) {
  const int resident = grid_cus() * grid_rounds();   // one 512-thread workgroup per CU
  const int kblocks = (k + 63) / 64;
  int want = std::max(32, resident / kblocks);
  // (a chunk's block -> group table lives in LDS)
  want = std::max(want, cdiv(rows, 8 * kGramWsMaxBlocks - 64));
  return std::max(1, std::min(std::min(want, kFusedMaxChunks * grid_rounds()), cdiv(rows, 32)));
}
int btr_sa_bwd_gram_chunks(int rows, int n, int k) { return gram_chunks(rows, n, k); }
size_t btr_sa_bwd_gram_scratch_floats(int rows, int n, int k) {
  const size_t kk = (size_t)k * k, ch = (size_t)gram_chunks(rows, n, k);
  // M, c | G partials, sx partials | G, sx in float64
  return kk + k + ch * (kk + k) + 2 * (kk + k) + 8;
}

int btr_sa_bwd_gram(int rows, int n, int k, const float *x, int ldx, const float *pa,
                    const float *pb, const float *mu_p, const float *is_p, const float *w,
                    const float *wt, int ldw, int s, const unsigned char *arg, const float *dcl,
                    const float *alpha, const float *beta, float *dz, int ldz, float *pw,
                    float *dw, float *gscratch, float *spart, float *m1, float *m2,
                    float *dgamma, float *dbeta, btr_stream_t stream) {
  GemmTrace trace_((hipStream_t)stream);
  BTR_REQUIRE(btr_sa_bwd_gram_supported(rows, n, k), "sa_bwd_gram: shape %d x %d x %d", rows, n, k);
  // per row: x M (k x k) and the Gram update x x^T (k x k); X read, dZ written
  trace_.work(rows, 4.0 * k * k, 8.0 * k, 8.0 * n * k, true);
  BTR_REQUIRE(x && pa && pb && mu_p && is_p && w && wt && arg && dcl && alpha && beta && dz &&
                  pw && dw && gscratch && spart && m1 && m2 && dgamma && dbeta &&
                  ldx % 4 == 0 && ldw % 4 == 0 && ldz % 4 == 0 && s > 0,
              "sa_bwd_gram: null pointer or unaligned leading dimension");
  BTR_REQUIRE(host_compact().on || s == 16 || s == 32 || s == 64 || s == 128,
              "sa_bwd_gram: nsample %d must be 16, 32, 64 or 128", s);
  hipStream_t st = as_stream(stream);
  const int chunks = gram_chunks(rows, n, k);
  BTR_REQUIRE(cdiv(cdiv(rows, chunks), 32) * 32 <= 8 * kGramWsMaxBlocks,
              "sa_bwd_gram: %d rows in %d chunks exceed the block table", rows, chunks);
  const size_t kk = (size_t)k * k;
  float *M = gscratch, *cvec = M + kk, *gp = cvec + k, *sxp = gp + (size_t)chunks * kk;
  size_t off = (size_t)(sxp + (size_t)chunks * k - gscratch);
  off = (off + 1) & ~(size_t)1;   // float64 from here on (gscratch itself is 256-byte aligned)
  double *G64 = reinterpret_cast<double *>(gscratch + off), *sx64 = G64 + kk;
  hipLaunchKernelGGL(gram_prep_kernel, dim3(cdiv(k, 16), cdiv(k, 4)), dim3(256), 0, st, n, k, wt,
                     ldw, alpha, beta, M, cvec);
  GramArgs a{};
  a.X = x; a.ldx = ldx; a.R = rows; a.N = n; a.K = k;
  a.rows_per_chunk = cdiv(cdiv(rows, chunks), 32) * 32;
  a.pa = pa; a.pb = pb; a.mu_p = mu_p; a.is_p = is_p; a.Wt = wt; a.ldw = ldw; a.M = M;
  a.cvec = cvec; a.Z = dz; a.ldz = ldz; a.pw = pw; a.gp = gp; a.sxp = sxp; a.spart = spart;
  a.garg = arg; a.gdcl = dcl; a.SSH = ilog2(s); a.ldt = n;
  const dim3 grid(1, cdiv(k, 64), chunks);
  if (n > 64)
    hipLaunchKernelGGL((sa_bwd_gram_ws_kernel<4>), grid, dim3(512), 0, st, a, cur_compact());
  else
    hipLaunchKernelGGL((sa_bwd_gram_ws_kernel<2>), grid, dim3(512), 0, st, a, cur_compact());
  const double count = host_compact().on ? host_compact().count : (double)rows;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(k, kRedCh)), dim3(256), 0, st, k, chunks,
                     count, spart, m1, m2, dgamma, dbeta);
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(cdiv((int)kk + k, 16)), dim3(256), 0, st, k, chunks,
                     gp, sxp, G64, sx64);
  hipLaunchKernelGGL(gram_finish_kernel, dim3(cdiv(n * k, 16)), dim3(256), 0, st, n, k, chunks,
                     pw, w, alpha, beta, G64, sx64, dw);
  return check_launch("sa_bwd_gram");
}

// dW[n][k] = sum_r G[r][n] * f(X[r][k]).
int btr_sa_gemm_tn(int rows, int n, int k, const float *g, int ldg, const float *x, int ldx,
                   const float *pa, const float *pb, float *pw, float *dw,
                   btr_stream_t stream) {
  GemmTrace trace_((hipStream_t)stream);
  if (n <= 0 || k <= 0) return BTR_OK;
  trace_.work(rows, 2.0 * n * k, 4.0 * (n + k), 4.0 * n * k, true);
  BTR_REQUIRE(g && x && pw && dw && ldg % 4 == 0 && ldx % 4 == 0 && n % 4 == 0 && k % 4 == 0,
              "sa_gemm_tn: sizes must be multiples of 4 (n=%d k=%d)", n, k);
  hipStream_t st = as_stream(stream);
  const int chunks = btr_sa_gemm_tn_chunks(rows, n, k);
  const int rpc = cdiv(cdiv(rows, chunks), 32) * 32;
  const int tn = tn_tile_n(n);
  const dim3 grid(cdiv(n, tn), cdiv(k, 64), chunks);
#define BTR_TN(W, P)                                                                       \
  launch_tn<W, P, false, false>(tn_x6(), grid, st, g, ldg, x, ldx, rows, n, k, pa, pb, rpc, pw, \
                                nullptr, nullptr, nullptr, nullptr, 0, nullptr)
  if (tn == 128) {
    if (pa) BTR_TN(4, true); else BTR_TN(4, false);
  } else {
    if (pa) BTR_TN(2, true); else BTR_TN(2, false);
  }
#undef BTR_TN
  reduce_chunks_launch(n * k, chunks, pw, dw, st);
  return check_launch("sa_gemm_tn");
}

// dW[n][k] = sum_r dY[r][n] * f(X[r][k]) with dY formed from (y, arg, dcl, alpha, beta) on the
// fly (n = the pooled layer's channel count); otherwise as btr_sa_gemm_tn.
int btr_sa_gemm_tn_pool(int rows, int n, int k, const float *y, int ldy, int s,
                        const unsigned char *arg, const float *dcl, const float *alpha,
                        const float *beta, const float *x, int ldx, const float *pa,
                        const float *pb, float *pw, float *dw, btr_stream_t stream) {
  GemmTrace trace_((hipStream_t)stream);
  if (n <= 0 || k <= 0) return BTR_OK;
  trace_.work(rows, 2.0 * n * k, 4.0 * (n + k), 4.0 * n * k, true);
  BTR_REQUIRE(y && x && pw && dw && arg && dcl && alpha && beta && s > 0 && ldy % 4 == 0 &&
                  ldx % 4 == 0 && n % 4 == 0 && k % 4 == 0,
              "sa_gemm_tn_pool: sizes must be multiples of 4 (n=%d k=%d)", n, k);
  BTR_REQUIRE(host_compact().on || s == 16 || s == 32 || s == 64 || s == 128,
              "sa_gemm_tn_pool: nsample %d must be 16, 32, 64 or 128", s);
  const int ssh = ilog2(s);
  hipStream_t st = as_stream(stream);
  const int chunks = btr_sa_gemm_tn_chunks(rows, n, k);
  const int rpc = cdiv(cdiv(rows, chunks), 32) * 32;
  const int tn = tn_tile_n(n);
  const dim3 grid(cdiv(n, tn), cdiv(k, 64), chunks);
#define BTR_TNP(W, P)                                                                      \
  launch_tn<W, P, true, false>(tn_pool_x6(), grid, st, y, ldy, x, ldx, rows, n, k, pa, pb, rpc, \
                               pw, arg, dcl, alpha, beta, ssh, nullptr)
  if (tn == 128) {
    if (pa) BTR_TNP(4, true); else BTR_TNP(4, false);
  } else {
    if (pa) BTR_TNP(2, true); else BTR_TNP(2, false);
  }
#undef BTR_TNP
  reduce_chunks_launch(n * k, chunks, pw, dw, st);
  return check_launch("sa_gemm_tn_pool");
}

size_t btr_sa_scatter_workspace_bytes(int b, int n, int m, int s) {
  if (b <= 0) return 0;
  return sizeof(int) * ((size_t)b * (n + 1) + (size_t)b * n + (size_t)b * m * s);
}

// Every output is fully written (no zero-init needed); any of them may be NULL.
// dfeat_cl is CHANNEL-LAST (b,n,c); dxyz (b,n,3); dnew_xyz (b,m,3).
int btr_sa_scatter(int b, int n, int m, int s, int c, int ldx, int use_xyz, float radius_div,
                   const float *dx0, const int *idx, float *dfeat_cl, float *dxyz,
                   float *dnew_xyz, void *workspace, size_t workspace_bytes,
                   btr_stream_t stream) {
  return sa_scatter_ex(b, n, m, s, c, ldx, use_xyz, radius_div, dx0, idx, dfeat_cl, dxyz,
                       dnew_xyz, workspace, workspace_bytes, kScatterBoth, as_stream(stream));
}

}  // extern "C"

int btr::sa_scatter_ex(int b, int n, int m, int s, int c, int ldx, int use_xyz, float radius_div,
                       const float *dx0, const int *idx, float *dfeat_cl, float *dxyz,
                       float *dnew_xyz, void *workspace, size_t workspace_bytes, int mode,
                       hipStream_t st) {
  if (b <= 0 || m <= 0 || s <= 0 || n <= 0) return BTR_OK;
  const long long ms = (long long)m * s;
  const float inv = radius_div != 0.f ? 1.0f / radius_div : 1.0f;
  const int xoff = use_xyz ? 3 : 0;
  if (!use_xyz) dxyz = dnew_xyz = nullptr;
  if (c <= 0) dfeat_cl = nullptr;
  if (dfeat_cl || dxyz || mode == kScatterBuild) {
    BTR_REQUIRE(workspace && workspace_bytes >= btr_sa_scatter_workspace_bytes(b, n, m, s),
                "sa_scatter: workspace too small");
    int *cnt_off = (int *)workspace;
    int *cursor = cnt_off + (size_t)b * (n + 1);
    int *refs = cursor + (size_t)b * n;
    if (mode != kScatterReduce) {
      if (n <= kCsrSmallN) {
        hipLaunchKernelGGL(csr_small_kernel, dim3(b), dim3(1024), 0, st, ms, n, idx, cnt_off, refs);
      } else {
        hipError_t e = hipMemsetAsync(cnt_off, 0, sizeof(int) * (size_t)b * (n + 1), st);
        if (e != hipSuccess) return fail((int)e, "sa_scatter memset: %s", hipGetErrorString(e));
        const int gx = (int)std::min<long long>(cdiv(ms, 256), 1024);
        hipLaunchKernelGGL(csr_count_kernel, dim3(gx, b), dim3(256), 0, st, ms, n, idx, cnt_off);
        hipLaunchKernelGGL(csr_scan_kernel, dim3(b), dim3(1024), 0, st, n, cnt_off, cursor);
        hipLaunchKernelGGL(csr_fill_kernel, dim3(gx, b), dim3(256), 0, st, ms, n, idx, cursor,
                           refs);
      }
    }
    if (mode != kScatterBuild)
      hipLaunchKernelGGL(csr_reduce_kernel, dim3(cdiv(n, 4), b), dim3(256), 0, st, n, ms, c, ldx,
                         xoff, inv, dx0, cnt_off, refs, dfeat_cl, dxyz);
  }
  if (dnew_xyz && mode != kScatterBuild) {
    const long long groups = (long long)b * m;
    hipLaunchKernelGGL(centre_grad_kernel, dim3(cdiv(groups * 3, 256)), dim3(256), 0, st, groups,
                       s, ldx, inv, dx0, dnew_xyz);
  }
  return check_launch("sa_scatter");
}

extern "C" {

// ------------------------------------------------------------------ compact rows (see above)
void btr_sac_bind(const btr_compact_t *cm) {
  HostCompact &hc = host_compact();
  hc.on = cm != nullptr;
  if (cm) {
    hc.dev.dims = cm->dims;
    hc.dev.bw = cm->bw;
    hc.dev.bgrp = cm->bgrp;
    hc.dev.goff = cm->goff;
    hc.count = cm->dense_rows;
  }
}

int btr_sac_plan(int groups, int s, const int *idx, int *len_tmp, int *goff, int *dims,
                 int *cidx, int *bgrp, float *bw, btr_stream_t stream) {
  if (groups <= 0 || s <= 0) return BTR_OK;
  BTR_REQUIRE(idx && len_tmp && goff && dims && cidx && bgrp && bw && s % 8 == 0,
              "sac_plan: null pointer or nsample %d not a multiple of 8", s);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(sac_count_kernel, dim3(cdiv(groups, 256)), dim3(256), 0, st, groups, s, idx,
                     len_tmp);
  hipLaunchKernelGGL(sac_scan_kernel, dim3(1), dim3(1024), 0, st, groups, len_tmp, goff, dims);
  hipLaunchKernelGGL(sac_fill_kernel, dim3(cdiv((long long)groups * s, 256)), dim3(256), 0, st,
                     groups, s, idx, goff, cidx, bgrp, bw);
  return check_launch("sac_plan");
}

int btr_sac_gather(int b, int n, int m, int max_rows, int c, int ldx, int use_xyz,
                   float radius_div, const float *xyz, const float *new_xyz,
                   const float *feats_cl, const int *cidx, const int *bgrp, const int *dims,
                   float *x0, btr_stream_t stream) {
  if (b <= 0 || m <= 0 || max_rows <= 0) return BTR_OK;
  BTR_REQUIRE(xyz && new_xyz && cidx && bgrp && dims && x0 &&
                  ldx >= (use_xyz ? 3 : 0) + c && ldx % 4 == 0,
              "sac_gather: bad arguments (ldx=%d, c=%d)", ldx, c);
  BTR_REQUIRE(c == 0 || feats_cl, "sac_gather: features missing");
  const int tpr = std::min(64, ldx);
  const int rpb = 256 / tpr;
  const int gx = (int)std::min<long long>(cdiv(max_rows, rpb), 8192);
  const float inv = radius_div != 0.f ? 1.0f / radius_div : 1.0f;
  hipLaunchKernelGGL(sac_gather_kernel, dim3(gx), dim3(256), 0, as_stream(stream), n, m, c, ldx,
                     use_xyz, inv, xyz, new_xyz, feats_cl, cidx, bgrp, dims, x0);
  return check_launch("sac_gather");
}

int btr_sac_pool_y(int b, int m, int c, const float *gext, const unsigned char *aext,
                   const int *goff, const float *scale, const float *shift, float *out,
                   float *out_cl, unsigned char *arg, float *ywin, btr_stream_t stream) {
  const long long groups = (long long)b * m;
  if (groups <= 0 || c <= 0) return BTR_OK;
  BTR_REQUIRE(gext && aext && goff && scale && shift && out && arg, "sac_pool: null pointer");
  if (pool_tiled(m) && pool_tile4(c))
    hipLaunchKernelGGL(sa_pool_tile4_kernel, dim3((unsigned)(groups / 32), cdiv(c, 128)), dim3(256),
                       0, as_stream(stream), m, c, gext, aext, goff, scale, shift, out, out_cl,
                       arg, ywin);
  else if (pool_tiled(m))
    hipLaunchKernelGGL(sa_pool_tile_kernel, dim3((unsigned)(groups / 32), cdiv(c, 32)), dim3(256),
                       0, as_stream(stream), m, c, gext, aext, goff, scale, shift, out, out_cl,
                       arg, ywin);
  else
    hipLaunchKernelGGL(sac_pool_kernel, dim3(cdiv(groups * c, 256)), dim3(256), 0,
                       as_stream(stream), m, c, groups, gext, aext, goff, scale, shift, out,
                       out_cl, arg, ywin);
  return check_launch("sac_pool");
}
int btr_sac_pool(int b, int m, int c, const float *gext, const unsigned char *aext,
                 const int *goff, const float *scale, const float *shift, float *out,
                 float *out_cl, unsigned char *arg, btr_stream_t stream) {
  return btr_sac_pool_y(b, m, c, gext, aext, goff, scale, shift, out, out_cl, arg, nullptr, stream);
}

size_t btr_sac_scatter_workspace_bytes(int b, int n, int max_rows) {
  if (b <= 0) return 0;
  return sizeof(int) * ((size_t)b * (n + 1) + (size_t)max_rows);
}

// dfeat_cl[b][n][c] = sum over the compact rows that reference point n of dx0[row][xoff + c]
// (n <= 8192 points per batch element: the layers behind SA1).
int btr_sac_scatter(int b, int n, int m, int c, int ldx, int use_xyz, const float *dx0,
                    const int *cidx, const int *goff, float *dfeat_cl, void *workspace,
                    size_t workspace_bytes, int max_rows, btr_stream_t stream) {
  return sac_scatter_ex(b, n, m, c, ldx, use_xyz, dx0, cidx, goff, dfeat_cl, workspace,
                        workspace_bytes, max_rows, kScatterBoth, as_stream(stream));
}

}  // extern "C"

int btr::sac_scatter_ex(int b, int n, int m, int c, int ldx, int use_xyz, const float *dx0,
                        const int *cidx, const int *goff, float *dfeat_cl, void *workspace,
                        size_t workspace_bytes, int max_rows, int mode, hipStream_t st) {
  if (b <= 0 || m <= 0 || n <= 0 || c <= 0) return BTR_OK;
  BTR_REQUIRE(cidx && goff && workspace && (mode == kScatterBuild || (dx0 && dfeat_cl)) &&
                  workspace_bytes >= btr_sac_scatter_workspace_bytes(b, n, max_rows),
              "sac_scatter: null pointer or workspace too small");
  BTR_REQUIRE(n <= kCsrSmallN, "sac_scatter: %d points per batch element > %d", n, kCsrSmallN);
  int *off = (int *)workspace;
  int *refs = off + (size_t)b * (n + 1);
  if (mode != kScatterReduce)
    hipLaunchKernelGGL(sac_csr_kernel, dim3(b), dim3(1024), 0, st, m, n, cidx, goff, off, refs);
  if (mode != kScatterBuild)
    hipLaunchKernelGGL(sac_reduce_kernel, dim3(cdiv(n, 4), b), dim3(256), 0, st, n, c, ldx,
                       use_xyz ? 3 : 0, dx0, off, refs, dfeat_cl);
  return check_launch("sac_scatter");
}

extern "C" {


}  // extern "C"
namespace btr {

// ---- per-point first layer ("PPFL") --------------------------------------------------------------
// The first convolution of a set-abstraction MLP acts on rows [xyz_j - c_i, f_j]: its feature part
// W_f f_j depends on the NEIGHBOUR POINT j only, not on the centre i it is grouped under -- and a
// level has 7 - 16 times as many (centre, neighbour) rows as points (SA2: 114 000 compact rows
// over 16 384 points).  So
//   P[j]  = W_f f_j                       one small GEMM over the points (btr_pm_gemm_nt)
//   Y0[r] = P[j(r)] + W_x rel(r)          this kernel: a 512-byte gather + three FMAs per channel
// replaces gathering the [3 + C]-wide rows (X0 written and read: 2 x 61 MB at SA2) and the
// rows x n x (3 + C) product.  The kernel also leaves rel(r) = (xyz_j - c_i) / radius as 16-byte
// rows (the weight gradient of W_x needs them) and the BatchNorm statistics partials of Y0 -- one
// `part` row per workgroup, rows weighted as in gemm_nt_kernel (compact rows' first row stands
// for 1 + S - len copies).  Same values as the row-wise product up to the order of the k sum.
// Backward (csrc/sa_layer.hip): dY0's rows are summed per point FIRST (the scatter the feature
// gradient needed anyway), then dF = S W_f and dW_f = S^T F are products over the points.
struct PpflArgs {
  const float *xyz, *new_xyz;   // (B, N, 3), (B, M, 3)
  const int *idx;               // dense rows: (B, M, S) neighbour index;  compact: cidx[row]
  const float *P;               // (B * N, nl)
  const float *w0x;             // [nl][4]: the xyz columns of W0 (4th zero)
  float *y0;                    // out (rows, nl)
  float *relx;                  // out (rows, 4)
  float *part;                  // out [gridDim.x][2][nl]
  int N, M, S, nl, rows, rows_per_wg;
  float inv_radius;
};
__global__ __launch_bounds__(256) void ppfl_gather_add_kernel(PpflArgs a, Compact cm) {
  __shared__ double red[2][8][132];
  const int q = threadIdx.x & 31, rr = threadIdx.x >> 5;   // 32 channel quads x 8 rows per pass
  int R = a.rows, rows_per_wg = a.rows_per_wg;
  if (cm.dims) {
    R = cm.dims[0];
    rows_per_wg = ((R + (int)gridDim.x - 1) / (int)gridDim.x + 7) / 8 * 8;
  }
  const int rbeg = blockIdx.x * rows_per_wg, rend = min(R, rbeg + rows_per_wg);
  double d1[4] = {0, 0, 0, 0}, d2[4] = {0, 0, 0, 0};
  for (int c0 = 0; c0 < a.nl; c0 += 128) {   // (nl <= 128 in every layer so far: one pass)
    const int c = c0 + q * 4;
    const bool live = c < a.nl;
    float4 wx = make_float4(0.f, 0.f, 0.f, 0.f), wy = wx, wz = wx;
    if (live) {   // W0x rows of this thread's four channels: [n][4] -> three per-axis quads
      const float4 r0 = *reinterpret_cast<const float4 *>(a.w0x + (size_t)(c + 0) * 4);
      const float4 r1 = *reinterpret_cast<const float4 *>(a.w0x + (size_t)(c + 1) * 4);
      const float4 r2 = *reinterpret_cast<const float4 *>(a.w0x + (size_t)(c + 2) * 4);
      const float4 r3 = *reinterpret_cast<const float4 *>(a.w0x + (size_t)(c + 3) * 4);
      wx = make_float4(r0.x, r1.x, r2.x, r3.x);
      wy = make_float4(r0.y, r1.y, r2.y, r3.y);
      wz = make_float4(r0.z, r1.z, r2.z, r3.z);
    }
    if (c0 > 0)
      for (int e = 0; e < 4; ++e) d1[e] = d2[e] = 0.0;
    // four rows of this thread's slot per trip: their index -> coordinate / P loads are in flight
    // together (one row at a time is a chain of two dependent memory round trips per 8 rows)
    for (int r4 = rbeg + rr; r4 < rend; r4 += 32) {
      int g[4], j[4];
      float w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r4 + 8 * u;
        g[u] = j[u] = 0;
        w[u] = 1.f;
        if (r < rend) {
          j[u] = a.idx[r];
          if (cm.bgrp) {
            g[u] = cm.bgrp[r >> 3];
            if ((r & 7) == 0) w[u] = cm.bw[r >> 3];
          } else {
            g[u] = r / a.S;
          }
        }
      }
      float rx[4], ry[4], rz[4];
      float4 p[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int bi = g[u] / a.M;
        const float *pj = a.xyz + ((size_t)bi * a.N + j[u]) * 3;
        const float *ci = a.new_xyz + (size_t)g[u] * 3;
        rx[u] = (pj[0] - ci[0]) * a.inv_radius;
        ry[u] = (pj[1] - ci[1]) * a.inv_radius;
        rz[u] = (pj[2] - ci[2]) * a.inv_radius;
        p[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) p[u] = *reinterpret_cast<const float4 *>(a.P + ((size_t)bi * a.N + j[u]) * a.nl + c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r4 + 8 * u;
        if (r >= rend) continue;
        if (c0 == 0 && q == 0)
          *reinterpret_cast<float4 *>(a.relx + (size_t)r * 4) = make_float4(rx[u], ry[u], rz[u], 0.f);
        if (live) {
          // the row-wise product's order: xyz columns first, then the feature sum
          float4 y;
          y.x = fmaf(rz[u], wz.x, fmaf(ry[u], wy.x, rx[u] * wx.x)) + p[u].x;
          y.y = fmaf(rz[u], wz.y, fmaf(ry[u], wy.y, rx[u] * wx.y)) + p[u].y;
          y.z = fmaf(rz[u], wz.z, fmaf(ry[u], wy.z, rx[u] * wx.z)) + p[u].z;
          y.w = fmaf(rz[u], wz.w, fmaf(ry[u], wy.w, rx[u] * wx.w)) + p[u].w;
          *reinterpret_cast<float4 *>(a.y0 + (size_t)r * a.nl + c) = y;
          const float ww = w[u];
          d1[0] += (double)(ww * y.x); d2[0] += (double)(ww * y.x * y.x);
          d1[1] += (double)(ww * y.y); d2[1] += (double)(ww * y.y * y.y);
          d1[2] += (double)(ww * y.z); d2[2] += (double)(ww * y.z * y.z);
          d1[3] += (double)(ww * y.w); d2[3] += (double)(ww * y.w * y.w);
        }
      }
    }
    // the eight row slots of a channel: fixed order
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[0][rr][q * 4 + e] = d1[e];
      red[1][rr][q * 4 + e] = d2[e];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * 128; t += 256) {
      const int which = t >> 7, cc = t & 127;
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) s += red[which][k][cc];
      if (c0 + cc < a.nl) a.part[((size_t)blockIdx.x * 2 + which) * a.nl + c0 + cc] = (float)s;
    }
    __syncthreads();
  }
}


// (internal.hpp) Y0 / relx / part of a PPFL first layer; `grid` = rows of `part` = workgroups
int ppfl_forward(int b, int n, int m, int s, int nl, int rows, float inv_radius, const float *xyz,
                 const float *new_xyz, const int *idx, const float *P, const float *w0x, float *y0,
                 float *relx, float *part, int grid, hipStream_t st) {
  GemmTrace trace_((hipStream_t)st);
  // (no matrix product of its own: the per-point product is a btr_pm_gemm_nt call; per row the
  // coordinate part's 3 multiply-adds per channel; the gathered product row read, Y0 written)
  trace_.work(rows, 6.0 * nl, 8.0 * nl + 16.0, 0.0, true);
  BTR_REQUIRE(xyz && new_xyz && idx && P && w0x && y0 && relx && part && nl % 4 == 0 && grid > 0,
              "ppfl_forward: bad arguments");
  PpflArgs a{};
  a.xyz = xyz; a.new_xyz = new_xyz; a.idx = idx; a.P = P; a.w0x = w0x; a.y0 = y0; a.relx = relx;
  a.part = part; a.N = n; a.M = m; a.S = s; a.nl = nl; a.rows = rows;
  a.rows_per_wg = cdiv(cdiv(rows, grid), 8) * 8;
  a.inv_radius = inv_radius;
  hipLaunchKernelGGL(ppfl_gather_add_kernel, dim3(grid), dim3(256), 0, st, a, cur_compact());
  return check_launch("ppfl_forward");
}
}  // namespace btr
extern "C" {

// ------------------------------------------------------------- point-wise MLP chains (pm)
// gemm_nt_kernel's row-tile workgroups (64-row tiles, grid-stride)
static int pm_tiles64_grid(int rows) {
  return std::max(1, std::min(cdiv(rows, 64),
                              gemm_wgs_per_cu() * grid_cus() * grid_rounds()));
}
// the small-M kernel (gemm_nt_sm_kernel): few rows, k in at most two staged chunks;
// BTR_PM_SM=0: never (read per call: the tests compare the two kernels)
// Measured (tools/gemm_sm_ab.py, alone on the chip, BatchNorm prologue + statistics; small-M /
// 64-row tiles): 1 024 x 288 x 288 8.2 / 15.3 us, 2 048 x 128 x 128 7.2 / 8.7, 4 096 x 256 x 256
// 12.4 / 14.5, 4 096 x 256 x 512 21.1 / 24.3 -- and 8 192 x 256 x 256 23.0 / 19.4, 16 384 x 256 x
// 256 40.4 / 21.8: with four and more tiles per CU each tile's own load -> split -> MFMA latency
// chain and its private copy of the weight fragments cost more than gemm_nt_kernel's 256
// fully parallel workgroups.  Inside the FSB step (beside the next batch's sampling kernel) the
// 4 096-row layers measured no gain either (3.98 vs 3.96 ms per step, three runs each): the default
// is <= 2 048 rows -- GroupFree3D's 1 024-row layers, the proposal head (BTR_PM_SM_ROWS overrides).
static bool pm_sm_rows(int rows) {
  const char *e = getenv("BTR_PM_SM");
  const char *mr = getenv("BTR_PM_SM_ROWS");   // (read per call, like BTR_PM_SM)
  const int max_rows = mr ? atoi(mr) : 2048;
  return rows > 0 && rows <= max_rows && gemm_x6() && !(e && e[0] == '0');
}
// Rows of `part` a statistics GEMM of the chains writes (and bn_finalize reads): one per 32-row
// tile where the small-M kernel may run, else one per workgroup of gemm_nt_kernel -- which writes
// zero rows for workgroups without a tile, so both kernels serve either count.
int btr_pm_gemm_grid(int rows) {
  return pm_sm_rows(rows) ? cdiv(rows, 32) : pm_tiles64_grid(rows);
}
static int ceil16i(int v) { return (v + 15) / 16 * 16; }
size_t btr_pm_weight_planes_bytes(int n, int k) { return (size_t)3 * n * ceil16i(k) * 2; }
int btr_pm_weight_planes(int n, int k, const float *w, int ldw, void *planes,
                         btr_stream_t stream) {
  BTR_REQUIRE(w && planes && n > 0 && k > 0, "pm_weight_planes: bad arguments");
  const int kp = ceil16i(k);
  hipLaunchKernelGGL(pm_weight_planes_kernel, dim3(cdiv((long long)n * kp, 256)), dim3(256), 0,
                     as_stream(stream), n, k, kp, w, ldw, (__bf16 *)planes);
  return check_launch("pm_weight_planes");
}
// Live timing of the GEMM family: between _begin and _end every GEMM-family entry point called on
// this host thread is bracketed by a HIP event pair on its stream.  _end synchronises the device,
// returns the summed milliseconds and the number of pairs, and closes the trace.
void btr_gemm_trace_begin(void) {
  GemmTraceState &st = gemm_trace_state();
  std::lock_guard<std::mutex> lock(st.mu);
  for (hipEvent_t e : st.ev) (void)hipEventDestroy(e);
  st.ev.clear();
  st.work.clear();
  st.nslot = 0;
  if (!st.pinned &&
      hipHostMalloc((void **)&st.pinned, kGemmTraceSlots * sizeof(int), hipHostMallocDefault) !=
          hipSuccess)
    st.pinned = nullptr;   // (rows then fall back to the host's dense bound)
  st.on.store(true);
}
int btr_gemm_trace_end(double *total_ms, int *pairs) {
  GemmTraceState &st = gemm_trace_state();
  st.on.store(false);
  (void)hipDeviceSynchronize();
  std::lock_guard<std::mutex> lock(st.mu);
  double ms = 0.0;
  for (size_t i = 0; i + 1 < st.ev.size(); i += 2) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, st.ev[i], st.ev[i + 1]) == hipSuccess) ms += t;
  }
  if (total_ms) *total_ms = ms;
  if (pairs) *pairs = (int)(st.ev.size() / 2);
  for (hipEvent_t e : st.ev) (void)hipEventDestroy(e);
  st.ev.clear();
  st.flops = st.bytes = st.dense_flops = 0.0;
  for (const GemmWork &w : st.work) {
    const double rows = (w.slot >= 0 && st.pinned) ? (double)st.pinned[w.slot] : (double)w.rows;
    st.flops += rows * w.flops_per_row;
    st.bytes += rows * w.bytes_per_row + w.bytes_fixed;
    st.dense_flops += (double)w.rows * w.flops_per_row;
  }
  st.work.clear();
  return BTR_OK;
}
// Work of the launches the last closed trace bracketed: flops (2 per multiply-add of the matrix
// products as executed), algorithmic operand bytes (every operand and result of a launch once),
// and the flops at the host's row counts (the dense bound where rows are compact).
int btr_gemm_trace_work(double *flops, double *bytes, double *dense_flops) {
  GemmTraceState &st = gemm_trace_state();
  std::lock_guard<std::mutex> lock(st.mu);
  if (flops) *flops = st.flops;
  if (bytes) *bytes = st.bytes;
  if (dense_flops) *dense_flops = st.dense_flops;
  return BTR_OK;
}
int btr_pm_gemm_nt_sm_supported(int rows, int n, int k) {
  return pm_sm_rows(rows) && n > 0 && k > 0 && k % 4 == 0 && k <= 2 * kSmKC && k <= kMaxK;
}
// C = f(A) . W^T with W given as its three bf16 planes (btr_pm_weight_planes); otherwise as
// btr_pm_gemm_nt (part: [btr_pm_gemm_grid(rows)][2][n])
int btr_pm_gemm_nt_sm(int rows, int n, int k, const float *a, int lda, const void *planes,
                      float *c, int ldc, const float *pa, const float *pb, float *part,
                      const float *bias, btr_stream_t stream) {
  GemmTrace trace_((hipStream_t)stream);
  if (rows <= 0 || n <= 0) return BTR_OK;
  trace_.work(rows, 2.0 * n * k, 4.0 * (n + k), 6.0 * n * k, false);
  BTR_REQUIRE(btr_pm_gemm_nt_sm_supported(rows, n, k), "pm_gemm_nt_sm: shape %d x %d x %d", rows,
              n, k);
  BTR_REQUIRE(a && planes && c && lda % 4 == 0 && ldc % 4 == 0,
              "pm_gemm_nt_sm: null pointer or unaligned leading dimension");
  BTR_REQUIRE((pa == nullptr) == (pb == nullptr), "pm_gemm_nt_sm: pa/pb must come together");
  BTR_REQUIRE(!(bias && part), "pm_gemm_nt_sm: bias and statistics are exclusive");
  hipStream_t s = as_stream(stream);
  const BnFin fin = take_bnfin();
  SmArgs sa{};
  sa.A = a; sa.lda = lda; sa.Wp = (const __bf16 *)planes; sa.kp = ceil16i(k); sa.C = c; sa.ldc = ldc;
  sa.ps = (long long)n * sa.kp;
  sa.R = rows; sa.N = n; sa.K = k; sa.pa = pa; sa.pb = pb; sa.part = part; sa.bias = bias;
  // (one 32-row tile per workgroup.  A persistent form -- ~2 workgroups per CU walking several row
  // tiles with the weight fragments kept in registers and the next tile's rows prefetched -- was
  // measured slower: 8 192 x 256 x 256 27.4 vs 23.0 us, 4 096 x 256 x 256 20.0 vs 12.4 us)
  const dim3 grid(cdiv(rows, 32), cdiv(n, 64));
#define BTR_SM(P, S, B) \
  hipLaunchKernelGGL((gemm_nt_sm_kernel<P, S, B>), grid, dim3(256), 0, s, sa, fin)
  if (pa) {
    if (part) BTR_SM(1, true, false);
    else if (bias) BTR_SM(1, false, true);
    else BTR_SM(1, false, false);
  } else {
    if (part) BTR_SM(0, true, false);
    else if (bias) BTR_SM(0, false, true);
    else BTR_SM(0, false, false);
  }
#undef BTR_SM
  return check_launch("pm_gemm_nt_sm");
}

}  // extern "C"
namespace btr {
// (internal.hpp) C = A . W^T (+ bias) on the small-M kernel with W given as the planes of a
// sub-block of a wider matrix: row pitch kp, plane stride ps (elements); any k % 4 == 0
int pm_gemm_nt_planes(int rows, int n, int k, const float *a, int lda, const void *planes, int kp,
                      long long ps, float *c, int ldc, const float *bias, hipStream_t s) {
  GemmTrace trace_((hipStream_t)s);
  if (rows <= 0 || n <= 0) return BTR_OK;
  trace_.work(rows, 2.0 * n * k, 4.0 * (n + k), 6.0 * n * k, false);
  BTR_REQUIRE(a && planes && c && k > 0 && k % 4 == 0 && lda % 4 == 0 && ldc % 4 == 0 &&
                  kp % 8 == 0 && ps % 8 == 0,
              "pm_gemm_nt_planes: bad arguments (%d x %d x %d)", rows, n, k);
  SmArgs sa{};
  sa.A = a; sa.lda = lda; sa.Wp = (const __bf16 *)planes; sa.kp = kp; sa.ps = ps; sa.C = c;
  sa.ldc = ldc; sa.R = rows; sa.N = n; sa.K = k; sa.bias = bias;
  const dim3 grid(cdiv(rows, 32), cdiv(n, 64));
  const BnFin fin{};
  if (bias)
    hipLaunchKernelGGL((gemm_nt_sm_kernel<0, false, true>), grid, dim3(256), 0, s, sa, fin);
  else
    hipLaunchKernelGGL((gemm_nt_sm_kernel<0, false, false>), grid, dim3(256), 0, s, sa, fin);
  return check_launch("pm_gemm_nt_planes");
}
}  // namespace btr
extern "C" {

// As btr_sa_gemm_nt on 64-row tiles (n > 64) with an optional bias row added to C (layers
// without BatchNorm); part: [btr_pm_gemm_grid(rows)][2][n].
int btr_pm_gemm_nt(int rows, int n, int k, const float *a, int lda, const float *w, int ldw,
                   float *c, int ldc, const float *pa, const float *pb, float *part,
                   const float *bias, btr_stream_t stream) {
  GemmTrace trace_((hipStream_t)stream);
  if (rows <= 0 || n <= 0) return BTR_OK;
  trace_.work(rows, 2.0 * n * k, 4.0 * (n + k), 4.0 * n * k, false);
  BTR_REQUIRE(a && w && c && k > 0 && k % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0,
              "pm_gemm_nt: k=%d lda=%d ldw=%d must be multiples of 4", k, lda, ldw);
  BTR_REQUIRE((pa == nullptr) == (pb == nullptr), "pm_gemm_nt: pa/pb must come together");
  BTR_REQUIRE(pa == nullptr || k <= kMaxK, "pm_gemm_nt: k=%d > %d with a prologue", k, kMaxK);
  BTR_REQUIRE(!(bias && part), "pm_gemm_nt: bias and statistics are exclusive");
  hipStream_t s = as_stream(stream);
  const int gx = btr_pm_gemm_grid(rows);
  const BnFin fin = take_bnfin();
#define BTR_PM_MM(BN, P, S, BIAS, MM)                                                          \
  hipLaunchKernelGGL((gemm_nt_kernel<BN, P, S, 0, 64, BIAS, MM>), dim3(gx, cdiv(n, BN)),         \
                     dim3(256), 0, s, a, lda, w, ldw, c, ldc, rows, n, k, pa, pb, part,         \
                     (const unsigned char *)nullptr, (const float *)nullptr, 0, bias,          \
                     (float *)nullptr, (unsigned char *)nullptr, Compact{}, fin)
#define BTR_PM(BN, P, S, BIAS)              \
  do {                                      \
    if (gemm_x6()) BTR_PM_MM(BN, P, S, BIAS, 1); \
    else BTR_PM_MM(BN, P, S, BIAS, 0);      \
  } while (0)
  if (pa) {
    if (part) BTR_PM(128, 1, true, false);
    else if (bias) BTR_PM(128, 1, false, true);
    else BTR_PM(128, 1, false, false);
  } else {
    if (part) BTR_PM(128, 0, true, false);
    else if (bias) BTR_PM(128, 0, false, true);
    else BTR_PM(128, 0, false, false);
  }
#undef BTR_PM
#undef BTR_PM_MM
  return check_launch("pm_gemm_nt");
}

}  // extern "C"

// Skinny products with a long reduction (the decoder's 1024 x 288 x 2048 FFN GEMMs: 48
// workgroups on 256 CUs, each looping over all of k): `slices` workgroups share one C tile's
// reduction and write their partial products to parts + z * part_stride ((rows, n), leading
// dimension n); the consumer adds the planes.  pm_splitk_slices: 1 = not worth it.
int btr::pm_splitk_slices(int rows, int n, int k) {
  const int wg = pm_tiles64_grid(rows) * cdiv(n, 128);
  if (wg >= 128 || k < 1024) return 1;
  return std::max(1, std::min(std::min(8, 256 / wg), k / 256));
}
int btr::pm_gemm_nt_splitk(int rows, int n, int k, const float *a, int lda, const float *w,
                           int ldw, float *parts, long long part_stride, int slices,
                           hipStream_t s) {
  if (rows <= 0 || n <= 0) return BTR_OK;
  BTR_REQUIRE(a && w && parts && k > 0 && k % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 &&
                  slices >= 1 && slices <= 64,
              "pm_gemm_nt_splitk: bad arguments (k=%d lda=%d ldw=%d slices=%d)", k, lda, ldw,
              slices);
  Compact cm{};
  cm.kz = cdiv(cdiv(k, slices), kBK) * kBK;
  cm.czs = part_stride;
  const dim3 grid(pm_tiles64_grid(rows), cdiv(n, 128), cdiv(k, cm.kz));
  BTR_REQUIRE((int)grid.z == slices, "pm_gemm_nt_splitk: %d slices of %d do not tile k=%d",
              slices, cm.kz, k);
#define BTR_SK(MM)                                                                               \
  hipLaunchKernelGGL((gemm_nt_kernel<128, 0, false, 0, 64, false, MM>), grid, dim3(256), 0, s, a, \
                     lda, w, ldw, parts, n, rows, n, k, (const float *)nullptr,                  \
                     (const float *)nullptr, (float *)nullptr, (const unsigned char *)nullptr,   \
                     (const float *)nullptr, 0, (const float *)nullptr, (float *)nullptr,        \
                     (unsigned char *)nullptr, cm)
  if (gemm_x6()) BTR_SK(1);
  else BTR_SK(0);
#undef BTR_SK
  return check_launch("pm_gemm_nt_splitk");
}

extern "C" {

// out_bcn (B, C, N) [and out_cl (B*N, C)] = f(scale * y + shift); scale/shift NULL: identity
int btr_pm_out(int b, int n, int c, int ldy, const float *y, const float *scale,
               const float *shift, int relu, float *out_bcn, float *out_cl,
               btr_stream_t stream) {
  return pm_out_add(b, n, c, ldy, y, scale, shift, relu, out_bcn, out_cl, nullptr, 0,
                    as_stream(stream));
}

// rows (B*N, ldr) = x (B, C, N) transposed, columns C .. ldr zero
int btr_pm_rows(int b, int n, int c, int ldr, const float *x, float *rows, btr_stream_t stream) {
  return pm_rows_zero(b, n, c, ldr, x, rows, nullptr, 0, nullptr, as_stream(stream));
}

}  // extern "C"