// sa_layer.hip -- whole-layer entry points: one set-abstraction layer (grouping + shared MLP +
// max-pool) or one point-wise MLP chain per call, forward and backward.
//
// Nothing here computes: these functions SEQUENCE the btr_sa_* / btr_sac_* / btr_pm_* launches
// of sa_mlp.hip, exactly as a caller of those entry points would (the Python orchestration in
// pointnet2/fused_sa.py / fused_mlp.py is kept as the readable statement of the same sequence
// and as the test oracle for this file: both must give bit-identical results).  Why it exists:
// the VoteNet step is ~400 launches; issued one binding call at a time (argument marshalling,
// one allocator call per intermediate) the host needs ~6.5 ms per step, more than the GPU needs
// to run them.  Here a layer costs the host one call, and its intermediates live at fixed
// offsets of three caller-provided buffers (saved for backward / scratch / flat gradients).
//
// reference: PointnetSAModuleVotes.forward (pointnet2/pointnet2_modules.py:243-267), SharedMLP
// (pointnet2/pytorch_utils.py:11-36), the Conv1d+BatchNorm1d+ReLU chains of
// models/voting_module.py:37-56 and models/proposal_module.py:75-113, and their autograd
// backward.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "internal.hpp"

namespace btr {
namespace {

constexpr int kMaxL = BTR_MAX_LAYERS;
constexpr size_t kAlign = 256;

inline int ceil4(int v) { return (v + 3) / 4 * 4; }
inline int ceil16(int v) { return (v + 15) / 16 * 16; }
// point-wise chains: the saved blocks of W2 (np x kin) and W2^T (kin x np) are followed by the
// matrix's three bf16 planes (see PrepArgsT::wp2); floats a block takes, and where its planes start
inline size_t pm_w_floats(int np, int kin, bool transposed) {
  const size_t planes = transposed ? (size_t)3 * kin * ceil16(np) : (size_t)3 * np * ceil16(kin);
  return (size_t)np * kin + (planes + 1) / 2;
}
inline __bf16 *pm_planes(float *w, int np, int kin) {
  return reinterpret_cast<__bf16 *>(w + (size_t)np * kin);
}
inline size_t up(size_t v) { return (v + kAlign - 1) / kAlign * kAlign; }

// Bump allocator over a byte range the caller owns (offsets only: usable for planning too).
struct Bump {
  size_t off = 0;
  size_t take(size_t bytes) {
    const size_t at = off;
    off = up(off + bytes);
    return at;
  }
  size_t floats(size_t n) { return take(n * sizeof(float)); }
};

// ---------------------------------------------------------------- weight preparation kernel
// One launch per layer stack: W2[l] = W[l] zero-padded to (np[l], kin[l]) and Wt[l] = W2[l]^T
// (the operands of the forward / input-gradient GEMMs), and num_batches_tracked += 1.
// the three bf16 pieces of an f32 (round to nearest each: csrc/sa_mlp.hip split4 / split1)
struct f32x2p {
  __bf16 h, m, l;
};
__device__ __forceinline__ f32x2p split3(float v) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  auto widen = [](b2 b) { return __uint_as_float(__builtin_bit_cast(unsigned, b) << 16); };
  const b2 bh = __builtin_convertvector(f2{v, 0.f}, b2);
  const float r1 = v - widen(bh);
  const b2 bm = __builtin_convertvector(f2{r1, 0.f}, b2);
  const float r2 = r1 - widen(bm);
  const b2 bl = __builtin_convertvector(f2{r2, 0.f}, b2);
  return f32x2p{bh.x, bm.x, bl.x};
}

template <int ML>
struct PrepArgsT {
  const float *w[ML];
  float *w2[ML];
  float *wt[ML];
  int n[ML];      // rows of W
  int np[ML];     // rows of W2 (zero rows beyond n)
  int kraw[ML];   // columns of W
  int kin[ML];    // columns of W2 (zero columns beyond kraw)
  int first[ML + 1];  // first block of layer l
  long long *nbt[ML];
  // point-wise chains: the bf16x6 planes of W2 ([3][np][ceil16(kin)]) and of W2^T
  // ([3][kin][ceil16(np)]) for the small-M GEMM (csrc/sa_mlp.hip gemm_nt_sm_kernel); NULL: none
  __bf16 *wp2[ML], *wpt[ML];
  // per-point first layer (c > 0 feature channels): W2 = [W_f (n, c) | W_x (n, 4)], Wt = W_f^T
  int ppfl_c[ML];
  int layers;
  unsigned *tickets;   // BatchNorm-finalisation tickets of this call (see BnFin): cleared here
  int ntickets;
  // bias row of a bare last layer whose width is not a multiple of 4, zero-padded to np
  const float *pbias_src;
  float *pbias_dst;
  int pbias_n, pbias_np;
};
typedef PrepArgsT<kMaxL> PrepArgs;
// several layer stacks in one launch (the whole-backbone call: 4 levels x 3 + 2 modules x 2 layers)
constexpr int kPrepBatchL = 24;
typedef PrepArgsT<kPrepBatchL> PrepArgsBatch;

template <int ML>
__global__ __launch_bounds__(256) void prep_weights_kernel(PrepArgsT<ML> a) {
  // the layer of this block: a fully unrolled scan with STATIC indices (a run-time index into
  // the by-value argument arrays would make the compiler spill the whole struct to scratch)
  const float *w = a.w[0];
  float *w2 = a.w2[0], *wt = a.wt[0];
  __bf16 *wp2 = a.wp2[0], *wpt = a.wpt[0];
  int pc = a.ppfl_c[0];
  long long *nbt = a.nbt[0];
  int n = a.n[0], np = a.np[0], kraw = a.kraw[0], kin = a.kin[0], first = 0;
#pragma unroll
  for (int i = 1; i < ML; ++i)
    if (i < a.layers && (int)blockIdx.x >= a.first[i]) {
      w = a.w[i]; w2 = a.w2[i]; wt = a.wt[i]; nbt = a.nbt[i]; wp2 = a.wp2[i]; wpt = a.wpt[i];
      pc = a.ppfl_c[i];
      n = a.n[i]; np = a.np[i]; kraw = a.kraw[i]; kin = a.kin[i]; first = a.first[i];
    }
  // (the element grid covers the 16-padded matrix: the planes' zero columns are written too)
  const int np16 = (np + 15) / 16 * 16, kp16 = (kin + 15) / 16 * 16;
  const int total = np16 * kp16;
  const int e = ((int)blockIdx.x - first) * 256 + (int)threadIdx.x;
  if (e < total) {
    const int r = e / kp16, c = e - r * kp16;
    float v = (r < n && c < kraw) ? w[(size_t)r * kraw + c] : 0.f;
    if (pc > 0) {   // (kin = pc + 4, kraw = 3 + pc: columns [xyz | features])
      if (r < np && c < pc) {
        v = w[(size_t)r * kraw + 3 + c];
        w2[(size_t)r * pc + c] = v;
        wt[(size_t)c * np + r] = v;
      } else if (r < np && c < kin) {
        const int cc = c - pc;
        const float wx = cc < 3 ? w[(size_t)r * kraw + cc] : 0.f;
        w2[(size_t)np * pc + (size_t)r * 4 + cc] = wx;
        wt[(size_t)c * np + r] = wx;   // W_x^T: rows pc .. pc + 3 behind W_f^T (coordinate gradient)
      }
    } else if (r < np && c < kin) {
      w2[(size_t)r * kin + c] = v;
      wt[(size_t)c * np + r] = v;
    }
    if (wp2) {
      const f32x2p f = split3(v);
      if (r < np) {
        wp2[((size_t)0 * np + r) * kp16 + c] = f.h;
        wp2[((size_t)1 * np + r) * kp16 + c] = f.m;
        wp2[((size_t)2 * np + r) * kp16 + c] = f.l;
      }
      if (c < kin) {
        wpt[((size_t)0 * kin + c) * np16 + r] = f.h;
        wpt[((size_t)1 * kin + c) * np16 + r] = f.m;
        wpt[((size_t)2 * kin + c) * np16 + r] = f.l;
      }
    }
  }
  if ((int)blockIdx.x == first && threadIdx.x == 0 && nbt) *nbt += 1;
  if (blockIdx.x == 0 && a.pbias_dst)
    for (int c = threadIdx.x; c < a.pbias_np; c += 256)
      a.pbias_dst[c] = c < a.pbias_n ? a.pbias_src[c] : 0.f;
  if (blockIdx.x == 0 && a.tickets)
    for (int c = threadIdx.x; c < a.ntickets; c += 256) a.tickets[c] = 0u;
}

// ---- one weight-preparation launch for several layer calls (btr_backbone_forward) -------------
// prep_batch_begin(): the layer calls that follow on this host thread only ADD their layers to
// the batch and return; prep_batch_launch(): one prep_weights_kernel for all of them, after which
// the same calls, issued again, skip their own launch; prep_batch_end(): back to normal.  A call
// that clears BatchNorm tickets or pads a bias row in its prep (few-row layers, bare last layers)
// does not take part: it reports `false` from prep_batch_take() in both passes and preps itself.
struct PrepBatch {
  PrepArgsBatch args;
  int blocks = 0;
  int mode = 0;   // 0 off, 1 collecting, 2 launched
};
inline PrepBatch &prep_batch() {
  static thread_local PrepBatch b;
  return b;
}
}  // namespace
void prep_batch_begin() {
  PrepBatch &b = prep_batch();
  b.args = PrepArgsBatch{};
  b.blocks = 0;
  b.mode = 1;
}
void prep_batch_launch(hipStream_t st) {
  PrepBatch &b = prep_batch();
  if (b.mode == 1 && b.args.layers > 0) {
    b.args.first[b.args.layers] = b.blocks;
    hipLaunchKernelGGL(prep_weights_kernel<kPrepBatchL>, dim3(b.blocks), dim3(256), 0, st, b.args);
  }
  b.mode = 2;
}
void prep_batch_end() { prep_batch().mode = 0; }
namespace {
bool prep_batch_collecting() { return prep_batch().mode == 1; }
// Collecting: adds the stack's layers (true), or refuses when the batch is full (false: the call
// then preps itself in the second pass).  Launched: true when the stack was added before.
template <int ML>
inline bool prep_batch_take(const PrepArgsT<ML> &pa, bool eligible) {
  PrepBatch &b = prep_batch();
  if (b.mode == 0 || !eligible) return false;
  if (b.mode == 2) {   // was this stack collected?  (its first W2 pointer identifies it)
    for (int i = 0; i < b.args.layers; ++i)
      if (b.args.w2[i] == pa.w2[0]) return true;
    return false;
  }
  if (b.args.layers + pa.layers > kPrepBatchL) return false;
  for (int l = 0; l < pa.layers; ++l) {
    const int i = b.args.layers + l;
    b.args.w[i] = pa.w[l]; b.args.w2[i] = pa.w2[l]; b.args.wt[i] = pa.wt[l];
    b.args.n[i] = pa.n[l]; b.args.np[i] = pa.np[l]; b.args.kraw[i] = pa.kraw[l];
    b.args.kin[i] = pa.kin[l]; b.args.nbt[i] = pa.nbt[l];
    b.args.wp2[i] = pa.wp2[l]; b.args.wpt[i] = pa.wpt[l]; b.args.ppfl_c[i] = pa.ppfl_c[l];
    b.args.first[i] = b.blocks + pa.first[l];
  }
  b.blocks += pa.first[pa.layers];
  b.args.layers += pa.layers;
  return true;
}

// out[c] = sum_r g[r][c]  (bias gradient of a bare last layer), two deterministic stages:
// part[tile][c] = sum over a 64-row tile (written by pm_rows_kernel on its way), then the tiles.
// (16 columns x 16 tile slices per workgroup, the slices added in LDS in a fixed order: as one
// thread per column -- a single workgroup walking all 128 tiles -- the voting module's call took
// 30 us)
__global__ __launch_bounds__(256) void colsum_final_kernel(int chunks, int c,
                                                           const float *__restrict__ part,
                                                           float *__restrict__ out) {
  __shared__ float red[16][17];
  const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int col = (int)blockIdx.x * 16 + e;
  float acc = 0.f;
  if (col < c)
#pragma unroll 4
    for (int k = sl; k < chunks; k += 16) acc += part[(size_t)k * c + col];
  red[sl][e] = acc;
  __syncthreads();
  if (sl == 0 && col < c) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][e];
    out[col] = t;
  }
}


// ------------------------------------------------------------------- vote assembly
// votes from the vote generator's last layer (models/voting_module.py:57-64, vote_factor 1):
//   vote_xyz[b][i][0..2]  = seed_xyz[b][i] + net[b][i][0..2]
//   vote_feat[b][i][c]    = seed_feat[b][i][c] + net[b][i][3 + c]
// on channel-last rows; the features are written channel-last and as (B, C, N) in one pass
// (the reference: transpose, two adds, two contiguous copies, a transpose back).
__global__ __launch_bounds__(256) void vote_assemble_kernel(
    int N, int C, int ldn, const float *__restrict__ net_cl, const float *__restrict__ seed_xyz,
    const float *__restrict__ seed_cl, float *__restrict__ vote_xyz,
    float *__restrict__ feat_bcn, float *__restrict__ feat_cl) {
  __shared__ float tile[64][65];
  const int bi = blockIdx.z, n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int n = n0 + ty + 4 * i, c = c0 + tx;
    float v = 0.f;
    if (n < N && c < C) {
      const size_t r = (size_t)bi * N + n;
      v = seed_cl[r * C + c] + net_cl[r * ldn + 3 + c];
      feat_cl[r * C + c] = v;
    }
    tile[ty + 4 * i][tx] = v;
  }
  if (blockIdx.y == 0 && threadIdx.x < 192) {   // the three coordinates of this tile's 64 points
    const int n = n0 + (int)threadIdx.x / 3, k = (int)threadIdx.x % 3;
    if (n < N) {
      const size_t r = (size_t)bi * N + n;
      vote_xyz[r * 3 + k] = seed_xyz[r * 3 + k] + net_cl[r * ldn + k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, n = n0 + tx;
    if (n < N && c < C) feat_bcn[((size_t)bi * C + c) * N + n] = tile[tx][ty + 4 * i];
  }
}

// gradient w.r.t. the generator's output (B, 3 + C, N): rows 0..2 = d vote_xyz transposed,
// rows 3.. = d vote_feat (the seed features' gradient is d vote_feat itself)
__global__ __launch_bounds__(256) void vote_assemble_bwd_kernel(
    int N, int C, const float *__restrict__ dxyz, const float *__restrict__ dfeat,
    float *__restrict__ dnet) {
  const int bi = blockIdx.z, ch = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float v = ch < 3 ? dxyz[((size_t)bi * N + n) * 3 + ch]
                         : dfeat[((size_t)bi * C + (ch - 3)) * N + n];
  dnet[((size_t)bi * (C + 3) + ch) * N + n] = v;
}

// The same with the L2 normalisation VoteNet applies to the vote features right after
// (models/votenet.py:98-99: features / ||features||_2 over the channels) folded in: a block owns
// 32 points and ALL channels (CJ * 8 of them: 8 threads per point, CJ values per thread in
// registers), so the norm is a reduction inside the block.  nrm (b*n): saved for the backward.
template <int CJ>
__global__ __launch_bounds__(256) void vote_assemble_norm_kernel(
    int N, int C, int ldn, const float *__restrict__ net_cl, const float *__restrict__ seed_xyz,
    const float *__restrict__ seed_cl, float *__restrict__ vote_xyz,
    float *__restrict__ feat_bcn, float *__restrict__ feat_cl, float *__restrict__ nrm) {
  __shared__ float tile[CJ * 8][33];
  const int bi = blockIdx.y, n0 = blockIdx.x * 32;
  const int row = threadIdx.x >> 3, sub = threadIdx.x & 7;
  const int n = n0 + row;
  const size_t r = (size_t)bi * N + min(n, N - 1);
  float v[CJ];
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < CJ; ++j) {
    const int c = sub + 8 * j;
    v[j] = c < C ? seed_cl[r * C + c] + net_cl[r * ldn + 3 + c] : 0.f;
    ss = fmaf(v[j], v[j], ss);
  }
  ss += __shfl_xor(ss, 1);
  ss += __shfl_xor(ss, 2);
  ss += __shfl_xor(ss, 4);
  const float nr = sqrtf(ss);
#pragma unroll
  for (int j = 0; j < CJ; ++j) {
    const int c = sub + 8 * j;
    const float y = v[j] / nr;
    if (c < C && n < N) feat_cl[r * C + c] = y;
    tile[c][row] = y;
  }
  if (n < N && sub == 0) nrm[r] = nr;
  if (n < N && sub < 3) vote_xyz[r * 3 + sub] = seed_xyz[r * 3 + sub] + net_cl[r * ldn + sub];
  __syncthreads();
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 points x 8 channel rows
  for (int c = ty; c < C; c += 8)
    if (n0 + tx < N) feat_bcn[((size_t)bi * C + c) * N + n0 + tx] = tile[c][tx];
}

// backward of the normalised assembly: y = v / ||v||,  dv = (dy - y (y . dy)) / ||v||;
// dnet rows 3.. and the seed features' gradient both receive dv, rows 0..2 = d vote_xyz^T
template <int CJ>
__global__ __launch_bounds__(256) void vote_assemble_norm_bwd_kernel(
    int N, int C, const float *__restrict__ dxyz, const float *__restrict__ dfeat,
    const float *__restrict__ y_cl, const float *__restrict__ nrm, float *__restrict__ dnet,
    float *__restrict__ dseed) {
  __shared__ float tile[CJ * 8][33];
  const int bi = blockIdx.y, n0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int c = ty; c < CJ * 8; c += 8)
    tile[c][tx] = (c < C && n0 + tx < N) ? dfeat[((size_t)bi * C + c) * N + n0 + tx] : 0.f;
  __syncthreads();
  const int row = threadIdx.x >> 3, sub = threadIdx.x & 7;
  const int n = n0 + row;
  const size_t r = (size_t)bi * N + min(n, N - 1);
  float y[CJ], dy[CJ];
  float dot = 0.f;
#pragma unroll
  for (int j = 0; j < CJ; ++j) {
    const int c = sub + 8 * j;
    y[j] = c < C ? y_cl[r * C + c] : 0.f;
    dy[j] = tile[c][row];
    dot = fmaf(y[j], dy[j], dot);
  }
  dot += __shfl_xor(dot, 1);
  dot += __shfl_xor(dot, 2);
  dot += __shfl_xor(dot, 4);
  const float nr = nrm[r];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < CJ; ++j) tile[sub + 8 * j][row] = (dy[j] - y[j] * dot) / nr;
  __syncthreads();
  for (int c = ty; c < C; c += 8)
    if (n0 + tx < N) {
      const float g = tile[c][tx];
      dnet[((size_t)bi * (C + 3) + 3 + c) * N + n0 + tx] = g;
      dseed[((size_t)bi * C + c) * N + n0 + tx] = g;
    }
  if (threadIdx.x < 96) {
    const int k = threadIdx.x >> 5, i = n0 + (threadIdx.x & 31);
    if (i < N) dnet[((size_t)bi * (C + 3) + k) * N + i] = dxyz[((size_t)bi * N + i) * 3 + k];
  }
}

// ------------------------------------------------------------- weight gradients on a side stream
// In a layer's backward the weight-gradient GEMM of layer l (dW_l = dY_l^T X_l) and the chain that
// continues to layer l - 1 (dX = dY_l W_l, BatchNorm backward, ...) only share their input dY_l.
// On the small layers (SA3 / SA4 / vote aggregation / the MLP chains: 2 000 - 65 000 rows) neither
// fills the chip, so the wgrad launches go to a second stream: fork after dY_l is final, join at
// the end of the call (the caller sees one stream).  Not while a HIP graph is being captured.
// OFF by default since round 4 (BTR_WGRAD_STREAM=1 turns it on): with a hidden layer's whole
// backward in one launch (sa_bwd_fused_kernel) few weight gradients are left to overlap, and a
// launch of equal row chunks that shares CUs with another stream's waves ends with its slowest
// workgroup -- same box, 20 steps: 4.39 -> 4.16 ms per VoteNet step, Back-to-Reality 10.43 -> 9.77.
// calls with fewer rows keep their weight gradients in line: point-wise chains under 8 192 rows
// (the fork / join calls cost the host more than two overlapped 10 us kernels return)
inline long long side_min_rows(bool sa) { return sa ? 0 : 8192; }
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t ready[kMaxL + 1] = {}, done[kMaxL + 1] = {};
  int device = -1;
};
SideStream *side_stream() {
  static thread_local SideStream ctx[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  SideStream &c = ctx[dev];
  if (!c.s) {
    if (hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking) != hipSuccess) {
      c.s = nullptr;
      return nullptr;
    }
    for (int i = 0; i <= kMaxL; ++i) {
      (void)hipEventCreateWithFlags(&c.ready[i], hipEventDisableTiming);
      (void)hipEventCreateWithFlags(&c.done[i], hipEventDisableTiming);
    }
    c.device = dev;
  }
  return &c;
}
// the side stream to use for this call, or NULL (disabled / capturing / creation failed)
SideStream *wgrad_side(hipStream_t main) {
  static const bool on = getenv("BTR_WGRAD_STREAM") && getenv("BTR_WGRAD_STREAM")[0] == '1';
  if (!on) return nullptr;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  // (inside a captured GroupFree3D step the fork / join nodes cost more than the overlap buys:
  // 15.3 vs 14.3 ms per replay)
  if (hipStreamIsCapturing(main, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return nullptr;
  }
  return side_stream();
}

struct Unbind {
  ~Unbind() { btr_sac_bind(nullptr); }
};
// collects the split-K reductions of the TN GEMMs issued in its scope (see reduce_batch_begin);
// flushes on every exit path
struct ReduceBatchScope {
  hipStream_t st;
  bool open = true;
  explicit ReduceBatchScope(hipStream_t s) : st(s) { reduce_batch_begin(); }
  void flush() {
    if (open) reduce_batch_flush(st);
    open = false;
  }
  ~ReduceBatchScope() { flush(); }
};

#define BTR_TRY(call)            \
  do {                           \
    const int rc_ = (call);      \
    if (rc_ != BTR_OK) return rc_; \
  } while (0)

inline float *at_f(void *base, size_t off) { return (float *)((char *)base + off); }
inline int *at_i(void *base, size_t off) { return (int *)((char *)base + off); }
inline unsigned char *at_b(void *base, size_t off) { return (unsigned char *)base + off; }

inline bool pool_grad_ok(int s) { return s == 16 || s == 32 || s == 64 || s == 128; }

// chunks the split-K partials [.][n][k] of a layer's weight gradient are sized for: whichever of
// the plain TN GEMM and the fused backward (csrc/sa_mlp.hip) splits the rows finer
inline int wgrad_partial_chunks(int rows, int n, int k) {
  int c = btr_sa_gemm_tn_chunks(rows, n, k);
  if (btr_sa_bwd_fused_supported(rows, n, k)) c = std::max(c, btr_sa_bwd_fused_chunks(rows, n, k));
  if (btr_sa_bwd_gram_supported(rows, n, k)) c = std::max(c, btr_sa_bwd_gram_chunks(rows, n, k));
  return c;
}

// scratch layouts (recomputed identically by plan / forward / backward)
struct SaFwdScratch {
  size_t part, len_tmp, extg, exta, tickets, ppfl_p, bytes;
};
struct SaBwdScratch {
  size_t part, m1, m2, dcl, alpha, beta, pw[kMaxL], pw0, g[2], scat, dfeat_cl, gram, bytes;
  size_t scat_bytes;
  size_t ppfl_s, ppfl_pw;   // per-point first layer: see sa_layer_backward_add
};

// Per-point first layer (csrc/sa_mlp.hip ppfl_gather_add_kernel): W_f f_j once per point, the rows
// gather it.  A function of the description and the plan only (the plan's buffer sizes, the
// forward and the backward must agree): feature input of >= 32 channels (the levels behind SA1).
inline bool sa_ppfl(const btr_sa_layer_t &d, const btr_sa_plan_t &p) {
  return (d.options & BTR_SA_OPT_PPFL) && !p.recompute && d.layers >= 2 && d.use_xyz &&
         d.c >= 32 && d.c % 4 == 0 && p.k0p == d.c + 4 &&
         // (not with coordinate gradients: the vote aggregation in this form -- d rel = dY_0 W_x as
         // one more 4-column product, the branch below -- is parity-green and NEUTRAL, 95 against
         // 96 us for its first layer's backward, 29 against 33 forward: five more launches at the
         // 7 - 10 us floor of 32 768-row kernels, profiles/r05_g_*; BTR_SA_OPT_PPFL_XYZ enables it)
         (!(d.need_dxyz || d.need_dnew_xyz) ||
          (!p.compact && (d.options & BTR_SA_OPT_PPFL_XYZ))) &&
         d.width[0] % 4 == 0 && d.width[0] <= 128 &&
         (!p.compact || d.n <= 8192);
}

// May the pooled (last) layer run without its stored output?  (see btr_sa_bwd_gram)
inline bool sa_gram_ok(const btr_sa_layer_t &d, const btr_sa_plan_t &p) {
  const int L = d.layers;
  if (!(d.options & BTR_SA_OPT_POOL_GRAM) || !p.pool_grad || !p.pool_epilogue || L < 2) return false;
  const int n = d.width[L - 1], k = p.kin[L - 1];
  return btr_sa_bwd_gram_supported(p.rows, n, k) != 0 &&
         btr_sa_gemm_nt_poolfwd_nostore_supported(p.rows, n, k, p.compact ? 8 : d.s) != 0 &&
         // (the pooled BatchNorm's backward through its tile kernel: btr_sa_pool_bwd_coef, ldy == 0)
         (long long)d.b * cdiv(d.m, 64) <= 1024 && n % 4 == 0 && d.b < 65536;
}

SaFwdScratch sa_fwd_scratch(const btr_sa_layer_t &d, const btr_sa_plan_t &p) {
  SaFwdScratch s{};
  Bump b;
  int maxn = 0;
  for (int l = 0; l < d.layers; ++l) maxn = std::max(maxn, d.width[l]);
  s.part = b.floats((size_t)btr_sa_gemm_grid(p.rows) * 2 * maxn);
  s.len_tmp = b.take(sizeof(int) * (size_t)d.b * d.m);
  const int cl = d.width[d.layers - 1];
  const int ps = p.compact ? 8 : d.s;
  const size_t ext = p.pool_epilogue ? (size_t)(p.rows / ps) * cl : 0;
  s.extg = b.floats(ext);
  s.exta = b.take(ext);
  s.tickets = b.take(sizeof(unsigned) * kBnTickets * kMaxL);
  s.ppfl_p = b.floats(sa_ppfl(d, p) ? (size_t)d.b * d.n * d.width[0] : 0);
  s.bytes = b.off;
  return s;
}

SaBwdScratch sa_bwd_scratch(const btr_sa_layer_t &d, const btr_sa_plan_t &p) {
  SaBwdScratch s{};
  Bump b;
  int maxc = 0, maxk = 0;
  for (int l = 0; l < d.layers; ++l) {
    maxc = std::max(maxc, d.width[l]);
    maxk = std::max(maxk, p.kin[l]);
  }
  const size_t pw0 = p.recompute ? (size_t)btr_sa_rc_wgrad_blocks(p.rows, d.width[0]) * d.width[0] * 4 : 0;
  const int cl = d.width[d.layers - 1];
  // [rows of partial sums][2][maxc]: the BatchNorm-backward statistics passes write <= 1024 rows,
  // btr_sa_bwd_fused writes one row per chunk
  int part_rows = 1024;
  for (int l = 1; l < d.layers; ++l)
    if (btr_sa_bwd_fused_supported(p.rows, d.width[l], p.kin[l]))
      part_rows = std::max(part_rows, btr_sa_bwd_fused_chunks(p.rows, d.width[l], p.kin[l]));
  if (p.pool_grad == 2)
    part_rows = std::max(part_rows, btr_sa_bwd_gram_chunks(p.rows, d.width[d.layers - 1],
                                                           p.kin[d.layers - 1]));
  s.part = b.floats((size_t)part_rows * 2 * maxc);
  s.m1 = b.floats(maxc);
  s.m2 = b.floats(maxc);
  s.dcl = b.floats((size_t)d.b * d.m * cl);
  s.alpha = b.floats(cl);
  s.beta = b.floats(cl);
  // split-K partials: one region per layer (their reductions are issued together at the end)
  for (int l = 0; l < d.layers; ++l)
    s.pw[l] = b.floats((size_t)wgrad_partial_chunks(p.rows, d.width[l], p.kin[l]) * d.width[l] *
                       p.kin[l]);
  s.pw0 = b.floats(pw0);   // (first-layer recompute: its partials, written on the main stream)
  s.gram = b.floats(p.pool_grad == 2 ? btr_sa_bwd_gram_scratch_floats(
                                           p.rows, d.width[d.layers - 1], p.kin[d.layers - 1])
                                     : 0);
  s.g[0] = b.floats((size_t)p.rows * maxk);
  s.g[1] = b.floats((size_t)p.rows * maxk);
  s.scat_bytes = p.compact ? btr_sac_scatter_workspace_bytes(d.b, d.n, p.rows)
                           : btr_sa_scatter_workspace_bytes(d.b, d.n, d.m, d.s);
  s.scat = b.take(s.scat_bytes);
  s.dfeat_cl = b.floats((size_t)d.b * d.n * std::max(d.c, 1));
  if (sa_ppfl(d, p)) {
    const int nl = d.width[0];
    s.ppfl_s = b.floats((size_t)d.b * d.n * nl);
    s.ppfl_pw = b.floats((size_t)btr_sa_gemm_tn_chunks(d.b * d.n, nl, d.c) * nl * d.c);
  }
  s.bytes = b.off;
  return s;
}

}  // namespace
}  // namespace btr

using namespace btr;

extern "C" {

// 1 when the layer computes its first layer per point (BTR_SA_OPT_PPFL and a covered shape): what
// the forward / backward of this (description, plan) pair do -- for tests and logs
int btr_sa_layer_ppfl(const btr_sa_layer_t *dp, const btr_sa_plan_t *p) {
  return dp && p && sa_ppfl(*dp, *p) ? 1 : 0;
}

int btr_sa_layer_plan(const btr_sa_layer_t *dp, btr_sa_plan_t *p) {
  BTR_REQUIRE(dp && p, "sa_layer_plan: null pointer");
  const btr_sa_layer_t &d = *dp;
  BTR_REQUIRE(d.layers >= 1 && d.layers <= kMaxL, "sa_layer_plan: %d layers", d.layers);
  BTR_REQUIRE(d.b > 0 && d.n > 0 && d.m > 0 && d.s > 0 && d.s <= 255 && d.c >= 0,
              "sa_layer_plan: bad sizes (b=%d n=%d m=%d s=%d c=%d)", d.b, d.n, d.m, d.s, d.c);
  BTR_REQUIRE((long long)d.b * d.m * d.s < (1ll << 31), "sa_layer_plan: too many rows");
  std::memset(p, 0, sizeof(*p));
  const int L = d.layers;
  p->rows = d.b * d.m * d.s;
  p->k0 = (d.use_xyz ? 3 : 0) + d.c;
  BTR_REQUIRE(p->k0 > 0, "sa_layer_plan: neither coordinates nor features");
  p->k0p = ceil4(p->k0);
  for (int l = 0; l < L; ++l) {
    BTR_REQUIRE(d.width[l] > 0 && d.width[l] % 4 == 0, "sa_layer_plan: width %d", d.width[l]);
    p->kin[l] = l == 0 ? p->k0p : d.width[l - 1];
  }
  const bool any_in = d.need_dxyz || d.need_dnew_xyz || d.need_dfeat;
  p->pool_grad = (d.options & BTR_SA_OPT_POOL_GRAD) && pool_grad_ok(d.s);
  const bool epi_allowed = (d.options & BTR_SA_OPT_POOL_EPILOGUE) != 0;
  p->compact = (d.options & BTR_SA_OPT_COMPACT) && d.s >= 32 && d.s % 8 == 0 && L >= 2 &&
               d.width[L - 1] > 64 && epi_allowed && p->pool_grad && !d.need_dxyz &&
               !d.need_dnew_xyz && (!d.need_dfeat || d.n <= 8192) &&
               (long long)d.b * ((d.m + 63) / 64) <= 1024;
  // (inference: the first layer's statistics-only launch would be for nothing)
  p->recompute = (d.options & BTR_SA_OPT_RECOMPUTE) && p->k0p == 4 && L >= 3 && !any_in &&
                 !(d.options & BTR_SA_OPT_EVAL);
  p->pool_epilogue = epi_allowed && L >= 2 &&
                     btr_sa_gemm_nt_poolfwd_supported(p->rows, d.width[L - 1],
                                                      p->compact ? 8 : d.s);
  BTR_REQUIRE(!p->compact || p->pool_epilogue, "sa_layer_plan: compact rows without epilogue");
  // pool_grad == 2: the pooled layer in Gram form (btr_sa_bwd_gram) -- its pre-BN output is
  // neither stored nor read; y[L-1] then holds the arg-max rows' values only (b, m, width)
  if (sa_gram_ok(d, *p)) p->pool_grad = 2;
  Bump sv;
  // (per-point first layer: the rows' relative coordinates [rows][4] + a copy of the features)
  p->x0 = sv.floats(sa_ppfl(d, *p) ? (size_t)p->rows * 4 + (size_t)d.b * d.n * d.c
                                   : (size_t)p->rows * p->k0p);
  for (int l = 0; l < L; ++l) {
    p->y[l] = sv.floats((p->recompute && l == 0) ? 0
                        : (p->pool_grad == 2 && l == L - 1) ? (size_t)d.b * d.m * d.width[l]
                                                            : (size_t)p->rows * d.width[l]);
    p->w2[l] = sv.floats((size_t)d.width[l] * p->kin[l]);
    p->wt[l] = sv.floats((size_t)d.width[l] * p->kin[l]);
    p->stats[l] = sv.floats((size_t)4 * d.width[l]);
  }
  p->arg = sv.take((size_t)d.b * d.m * d.width[L - 1]);
  if (p->compact) {
    p->goff = sv.take(sizeof(int) * ((size_t)d.b * d.m + 1));
    p->dims = sv.take(sizeof(int) * 2);
    p->cidx = sv.take(sizeof(int) * (size_t)p->rows);
    p->bgrp = sv.take(sizeof(int) * (size_t)(p->rows / 8));
    p->bw = sv.take(sizeof(float) * (size_t)(p->rows / 8));
  }
  p->saved_bytes = sv.off;
  p->fwd_scratch_bytes = sa_fwd_scratch(d, *p).bytes;
  p->bwd_scratch_bytes = sa_bwd_scratch(d, *p).bytes;
  size_t g = 0;
  for (int l = 0; l < L; ++l) {
    p->dw[l] = g;
    g += (size_t)d.width[l] * p->kin[l];
    p->dgamma[l] = g;
    g += d.width[l];
    p->dbeta[l] = g;
    g += d.width[l];
  }
  p->grads_floats = g;
  return BTR_OK;
}

int btr_sa_layer_forward(const btr_sa_layer_t *dp, const btr_sa_plan_t *pp, const float *xyz,
                         const float *new_xyz, const float *feats_cl, const int *idx, float *out,
                         float *out_cl, void *saved, void *scratch, btr_stream_t stream) {
  return sa_layer_forward_geom(dp, pp, xyz, new_xyz, feats_cl, idx, out, out_cl, saved, scratch,
                               nullptr, stream);
}

}  // extern "C"

// (the compact-row plan of a layer: in its `saved` block unless the caller prepared it)
namespace {
struct CompactPtrs {
  int *goff, *dims, *cidx, *bgrp;
  float *bw;
};
inline CompactPtrs compact_ptrs(const btr_sa_plan_t &p, void *saved, const btr::SaGeom *g) {
  if (g && g->goff) return CompactPtrs{g->goff, g->dims, g->cidx, g->bgrp, g->bw};
  return CompactPtrs{at_i(saved, p.goff), at_i(saved, p.dims), at_i(saved, p.cidx),
                     at_i(saved, p.bgrp), at_f(saved, p.bw)};
}
}  // namespace

int btr::sa_layer_forward_geom(const btr_sa_layer_t *dp, const btr_sa_plan_t *pp,
                               const float *xyz, const float *new_xyz, const float *feats_cl,
                               const int *idx, float *out, float *out_cl, void *saved,
                               void *scratch, const SaGeom *geom, btr_stream_t stream) {
  BTR_REQUIRE(dp && pp && xyz && new_xyz && idx && out && saved && scratch,
              "sa_layer_forward: null pointer");
  const btr_sa_layer_t &d = *dp;
  const btr_sa_plan_t &p = *pp;
  const int L = d.layers, R = p.rows;
  const SaFwdScratch sc = sa_fwd_scratch(d, p);
  btr_sac_bind(nullptr);
  Unbind unbind;

  PrepArgs pa{};
  pa.layers = L;
  int blocks = 0;
  for (int l = 0; l < L; ++l) {
    BTR_REQUIRE(d.w[l] && d.gamma[l] && d.beta[l], "sa_layer_forward: layer %d parameters", l);
    pa.w[l] = d.w[l];
    pa.w2[l] = at_f(saved, p.w2[l]);
    pa.wt[l] = at_f(saved, p.wt[l]);
    pa.n[l] = pa.np[l] = d.width[l];
    pa.kraw[l] = l == 0 ? p.k0 : d.width[l - 1];
    pa.kin[l] = p.kin[l];
    pa.first[l] = blocks;
    blocks += cdiv((long long)ceil16(d.width[l]) * ceil16(p.kin[l]), 256);
    pa.nbt[l] = d.running_mean[l] ? d.num_batches_tracked[l] : nullptr;
  }
  // Inference (BTR_SA_OPT_EVAL): gamma[l] / beta[l] are the affine map of BatchNorm on its running
  // statistics, derived by the caller -- no finaliser runs, nothing is tracked; the statistics
  // epilogues still write their partial sums into the scratch (nobody reads them)
  const bool eval = (d.options & BTR_SA_OPT_EVAL) != 0;
  if (eval)
    for (int l = 0; l < L; ++l) {
      BTR_REQUIRE(!d.running_mean[l] && !d.running_var[l],
                  "sa_layer_forward: inference mode tracks no statistics");
      pa.nbt[l] = nullptr;
    }
  const bool ppfl = sa_ppfl(d, p);
  if (ppfl) pa.ppfl_c[0] = d.c;
  pa.first[L] = blocks;
  unsigned *tickets = reinterpret_cast<unsigned *>(at_b(scratch, sc.tickets));
  pa.tickets = tickets;
  pa.ntickets = kBnTickets * kMaxL;
  // (a whole-backbone call preps every stack with one launch: see prep_batch_begin)
  const bool batched = prep_batch_take(pa, !bnfin_rows_ok(R));
  if (prep_batch_collecting()) return BTR_OK;
  if (!batched)
    hipLaunchKernelGGL(prep_weights_kernel<kMaxL>, dim3(blocks), dim3(256), 0, as_stream(stream),
                       pa);

  float *x0 = at_f(saved, p.x0);
  btr_compact_t cm{};
  const CompactPtrs cp = compact_ptrs(p, saved, geom);
  if (p.compact) {
    cm.dims = cp.dims;
    cm.bw = cp.bw;
    cm.bgrp = cp.bgrp;
    cm.goff = cp.goff;
    cm.dense_rows = (double)R;
    if (!(geom && geom->goff))   // (else: planned by btr_backbone_sampling)
      BTR_TRY(btr_sac_plan(d.b * d.m, d.s, idx, at_i(scratch, sc.len_tmp), cp.goff, cp.dims,
                           cp.cidx, cp.bgrp, cp.bw, stream));
    if (!ppfl)
      BTR_TRY(btr_sac_gather(d.b, d.n, d.m, R, d.c, p.k0p, d.use_xyz, d.radius_div, xyz, new_xyz,
                             feats_cl, cp.cidx, cp.bgrp, cp.dims, x0, stream));
    btr_sac_bind(&cm);
  } else if (!ppfl) {
    BTR_TRY(btr_sa_gather(d.b, d.n, d.m, d.s, d.c, p.k0p, d.use_xyz, d.radius_div, xyz, new_xyz,
                          feats_cl, idx, x0, stream));
  }
  const int grid = btr_sa_gemm_grid(R);
  float *part = at_f(scratch, sc.part);
  float *extg = at_f(scratch, sc.extg);
  unsigned char *exta = at_b(scratch, sc.exta);
  const float *A = x0;
  int lda = p.k0p;
  const float *pscale = nullptr, *pshift = nullptr;
  for (int l = 0; l < L; ++l) {
    const int nl = d.width[l], k = p.kin[l];
    const float *w2 = at_f(saved, p.w2[l]);
    float *y = at_f(saved, p.y[l]);
    float *st = at_f(saved, p.stats[l]);
    // BatchNorm finalisation by the statistics GEMM's last workgroup (false: a launch of its own)
    const bool fin_fused = eval || (!(ppfl && l == 0) && bnfin_arm(BnFin{
        tickets + kBnTickets * l, d.gamma[l], d.beta[l], st, st + nl, st + 2 * nl, st + 3 * nl,
        d.running_mean[l], d.running_var[l], nullptr, 0, (double)R, d.eps[l], d.momentum[l]}, R));
    if (ppfl && l == 0) {
      // ---- per-point first layer: P = F W_f^T over the POINTS, the rows gather it (+ W_x rel);
      // x0 = [the rows' relative coordinates (R, 4) | a copy of the features for the backward]
      BTR_REQUIRE(feats_cl, "sa_layer_forward: features missing");
      const float *w0f = w2, *w0x = w2 + (size_t)nl * d.c;
      float *fcopy = x0 + (size_t)R * 4;
      const size_t fbytes = sizeof(float) * (size_t)d.b * d.n * d.c;
      if (!eval) {   // (the backward's copy of the features)
        const hipError_t ce = hipMemcpyAsync(fcopy, feats_cl, fbytes, hipMemcpyDeviceToDevice,
                                             as_stream(stream));
        if (ce != hipSuccess) return fail((int)ce, "sa_layer_forward: %s", hipGetErrorString(ce));
      }
      float *P = at_f(scratch, sc.ppfl_p);
      btr_sac_bind(nullptr);   // (the product over the points is no compact-row GEMM)
      BTR_TRY(btr_pm_gemm_nt(d.b * d.n, nl, d.c, feats_cl, d.c, w0f, d.c, P, nl, nullptr, nullptr,
                             nullptr, nullptr, stream));
      if (p.compact) btr_sac_bind(&cm);
      BTR_TRY(ppfl_forward(d.b, d.n, d.m, d.s, nl, R,
                           d.radius_div != 0.f ? 1.0f / d.radius_div : 1.0f, xyz, new_xyz,
                           p.compact ? cp.cidx : idx, P, w0x, y, x0, part, grid, as_stream(stream)));
    } else if (p.recompute && l == 0) {  // statistics only
      BTR_TRY(btr_sa_gemm_nt(R, nl, k, A, lda, w2, k, nullptr, nl, nullptr, nullptr, part, stream));
    } else if (p.recompute && l == 1) {
      BTR_TRY(btr_sa_gemm_nt_rc(R, nl, k, x0, at_f(saved, p.w2[0]), w2, k, y, nl, pscale, pshift,
                                part, stream));
    } else if (l == L - 1 && p.pool_epilogue) {
      // (Gram-form backward: no Y_l -- y then is the (b, m, nl) block of arg-max values below)
      BTR_TRY(btr_sa_gemm_nt_poolfwd(R, nl, k, A, lda, w2, k, p.pool_grad == 2 ? nullptr : y, nl,
                                     pscale, pshift, part, p.compact ? 8 : d.s, d.gamma[l], extg,
                                     exta, stream));
    } else {
      BTR_TRY(btr_sa_gemm_nt(R, nl, k, A, lda, w2, k, y, nl, pscale, pshift, part, stream));
    }
    if (!fin_fused)
      BTR_TRY(btr_sa_bn_finalize(nl, grid, (double)R, d.eps[l], d.momentum[l], part, d.gamma[l],
                                 d.beta[l], st, st + nl, st + 2 * nl, st + 3 * nl,
                                 d.running_mean[l], d.running_var[l], stream));
    A = y;
    lda = nl;
    pscale = eval ? d.gamma[l] : st;
    pshift = eval ? d.beta[l] : st + nl;
  }
  btr_sac_bind(nullptr);
  const int cl = d.width[L - 1];
  unsigned char *arg = at_b(saved, p.arg);
  float *ywin = p.pool_grad == 2 ? at_f(saved, p.y[L - 1]) : nullptr;
  if (p.compact)
    BTR_TRY(btr_sac_pool_y(d.b, d.m, cl, extg, exta, cp.goff, pscale, pshift, out,
                           out_cl, arg, ywin, stream));
  else if (p.pool_epilogue)
    BTR_TRY(btr_sa_pool_fin_y(d.b, d.m, cl, extg, exta, pscale, pshift, out, out_cl, arg, ywin,
                              stream));
  else
    BTR_TRY(btr_sa_pool(d.b, d.m, d.s, cl, cl, A, pscale, pshift, out, out_cl, arg, stream));
  return check_launch("sa_layer_forward");
}

extern "C" {

int btr_sa_layer_backward(const btr_sa_layer_t *dp, const btr_sa_plan_t *pp, const int *idx,
                          const float *out, const float *dout, void *saved, float *grads,
                          float *dfeat, float *dxyz, float *dnew_xyz, void *scratch,
                          btr_stream_t stream) {
  return sa_layer_backward_add(dp, pp, idx, out, dout, saved, grads, dfeat, dxyz, dnew_xyz,
                               scratch, nullptr, 0, nullptr, stream);
}

}  // extern "C"

int btr::sa_layer_backward_add(const btr_sa_layer_t *dp, const btr_sa_plan_t *pp, const int *idx,
                               const float *out, const float *dout, void *saved, float *grads,
                               float *dfeat, float *dxyz, float *dnew_xyz, void *scratch,
                               const float *dfeat_add, long long dfeat_add_bstride,
                               const SaGeom *geom, btr_stream_t stream) {
  BTR_REQUIRE(dp && pp && idx && out && dout && saved && grads && scratch,
              "sa_layer_backward: null pointer");
  BTR_REQUIRE(!(dp->options & BTR_SA_OPT_EVAL), "sa_layer_backward: an inference-mode layer");
  const btr_sa_layer_t &d = *dp;
  const btr_sa_plan_t &p = *pp;
  const int L = d.layers, R = p.rows;
  const SaBwdScratch sc = sa_bwd_scratch(d, p);
  btr_sac_bind(nullptr);
  Unbind unbind;
  btr_compact_t cm{};
  const CompactPtrs cp = compact_ptrs(p, saved, geom);
  if (p.compact) {
    cm.dims = cp.dims;
    cm.bw = cp.bw;
    cm.bgrp = cp.bgrp;
    cm.goff = cp.goff;
    cm.dense_rows = (double)R;
    btr_sac_bind(&cm);
  }
  float *part = at_f(scratch, sc.part), *m1 = at_f(scratch, sc.m1), *m2 = at_f(scratch, sc.m2);
  float *dcl = at_f(scratch, sc.dcl), *alpha = at_f(scratch, sc.alpha);
  float *beta = at_f(scratch, sc.beta);
  float *x0 = at_f(saved, p.x0);
  const unsigned char *arg = at_b(saved, p.arg);
  auto stat = [&](int l, int which) { return at_f(saved, p.stats[l]) + which * d.width[l]; };

  const int cl = d.width[L - 1];
  float *ylast = at_f(saved, p.y[L - 1]);
  const bool gram = p.pool_grad == 2;   // ylast = the arg-max rows' values (b, m, cl): ldy = 0
  const bool ppfl = sa_ppfl(d, p);
  if (p.pool_grad)
    BTR_TRY(btr_sa_pool_bwd_coef(d.b, d.m, d.s, cl, gram ? 0 : cl, ylast, dout, out, arg, stat(L - 1, 2),
                                 stat(L - 1, 3), stat(L - 1, 0), stat(L - 1, 1), part, m1, m2,
                                 grads + p.dgamma[L - 1], grads + p.dbeta[L - 1], dcl, alpha, beta,
                                 stream));
  else
    BTR_TRY(btr_sa_pool_bwd(d.b, d.m, d.s, cl, cl, ylast, dout, out, arg, stat(L - 1, 2),
                            stat(L - 1, 3), stat(L - 1, 0), part, m1, m2,
                            grads + p.dgamma[L - 1], grads + p.dbeta[L - 1], stream));
  const bool any_in = d.need_dxyz || d.need_dnew_xyz || d.need_dfeat;
  float *dy = ylast;
  int flip = 0;
  hipStream_t hmain = as_stream(stream);
  SideStream *side = R >= side_min_rows(true) ? wgrad_side(hmain) : nullptr;
  int last_done = -1;
  // the layers' split-K reductions: ONE launch behind the last TN GEMM, on the stream they ran on
  ReduceBatchScope batch(side ? side->s : hmain);
  // `lazy`: dy holds dZ_l, the gradient w.r.t. layer l's ACTIVATION, with BatchNorm_l's backward
  // sums finalised in (m1, m2) -- not yet dY_l.  btr_sa_bwd_fused applies them while it stages
  // its operand; any other consumer gets them applied in place first.
  bool lazy = false, fused_any = false;
  // (the first-layer recompute rides in the fused call of layer 1 only behind <= 128 columns:
  // a wider layer 1 takes the separate calls)
  auto fusable = [&](int l) {
    return l >= 1 && btr_sa_bwd_fused_supported(R, d.width[l], p.kin[l]) != 0 &&
           !(p.recompute && l == 1 && d.width[l] > 128);
  };
  for (int l = L - 1; l >= 0; --l) {
    const int nl = d.width[l], k = p.kin[l];
    const float *xsrc = l == 0 ? x0 : at_f(saved, p.y[l - 1]);
    const int ldx = l == 0 ? p.k0p : d.width[l - 1];
    const float *pa = l == 0 ? nullptr : stat(l - 1, 0);
    const float *pb = l == 0 ? nullptr : stat(l - 1, 1);
    const bool pooled = p.pool_grad && l == L - 1;
    if (p.recompute && l == 0) {
      // the recomputed first layer behind a fused call: its sums exist, one pass left
      if (lazy)
        BTR_TRY(btr_sa_bn_relu_bwd_rc_apply(R, d.width[0], d.width[0], dy, x0,
                                            at_f(saved, p.w2[0]), stat(0, 0), stat(0, 1),
                                            stat(0, 2), stat(0, 3), m1, m2,
                                            at_f(scratch, sc.pw0), grads + p.dw[0], stream));
      break;  // (otherwise finished by btr_sa_bn_relu_bwd_rc below)
    }
    if (pooled && gram) {
      // ---- the pooled layer in Gram form (csrc/sa_mlp.hip sa_bwd_gram_kernel): dW_l, dZ_{l-1}
      // and BatchNorm_{l-1}'s sums from X_{l-1} alone
      float *g = at_f(scratch, sc.g[flip]);
      flip ^= 1;
      BTR_TRY(btr_sa_bwd_gram(R, nl, k, xsrc, ldx, pa, pb, stat(l - 1, 2), stat(l - 1, 3),
                              at_f(saved, p.w2[l]), at_f(saved, p.wt[l]), nl, d.s, arg, dcl,
                              alpha, beta, g, k, at_f(scratch, sc.pw[l]), grads + p.dw[l],
                              at_f(scratch, sc.gram), part, m1, m2, grads + p.dgamma[l - 1],
                              grads + p.dbeta[l - 1], stream));
      dy = g;
      lazy = true;   // (no split-K reduction left for the join below: the call finished dW_l)
      continue;
    }
    if (fusable(l) && (pooled || lazy)) {
      // ---- the whole backward of layer l in one pass (csrc/sa_mlp.hip sa_bwd_fused_kernel):
      // dW_l, dZ_{l-1} and BatchNorm_{l-1}'s sums; dY_l is formed while its rows are staged
      float *g = at_f(scratch, sc.g[flip]);
      flip ^= 1;
      // g is the buffer dY_{l+1} lived in: a side-stream weight gradient must be through with it
      if (side && last_done == l + 1) (void)hipStreamWaitEvent(hmain, side->done[l + 1], 0);
      const bool rc1 = p.recompute && l == 1;
      BTR_TRY(btr_sa_bwd_fused(
          R, nl, k, pooled ? ylast : dy, nl, pooled ? nullptr : at_f(saved, p.y[l]), stat(l, 0),
          stat(l, 1), stat(l, 2), stat(l, 3), m1, m2, d.s, pooled ? arg : nullptr,
          pooled ? dcl : nullptr, pooled ? alpha : nullptr, pooled ? beta : nullptr,
          rc1 ? x0 : xsrc, rc1 ? 4 : ldx, rc1 ? at_f(saved, p.w2[0]) : nullptr, pa, pb,
          stat(l - 1, 2), stat(l - 1, 3), at_f(saved, p.wt[l]), nl, g, k,
          at_f(scratch, sc.pw[l]), grads + p.dw[l], part, m1, m2, grads + p.dgamma[l - 1],
          grads + p.dbeta[l - 1], stream));
      dy = g;
      lazy = fused_any = true;
      continue;
    }
    if (lazy) {   // a consumer that wants dY_l itself
      BTR_TRY(btr_sa_bn_relu_bwd_apply(R, nl, nl, dy, at_f(saved, p.y[l]), stat(l, 0), stat(l, 1),
                                       stat(l, 2), stat(l, 3), m1, m2, stream));
      lazy = false;
    }
    if (ppfl && l == 0) {
      // ---- per-point first layer (csrc/sa_mlp.hip ppfl_gather_add_kernel): dY_0's rows are summed
      // per POINT first -- the scatter the feature gradient needs anyway -- and the feature parts
      // of both gradients become products over the points:
      //   S[j] = sum of dY_0[r] over the rows r of point j          (n points x nl)
      //   dW_f = S^T F,  dF = S W_f;   dW_x = dY_0^T rel  (k = 4, over the rows)
      const int n0 = d.width[0], np = d.b * d.n;
      float *relx = x0, *fcopy = x0 + (size_t)R * 4;
      float *S = at_f(scratch, sc.ppfl_s);
      btr_stream_t ws = stream;
      const int prev_done = last_done;
      if (side) {
        (void)hipEventRecord(side->ready[0], hmain);
        (void)hipStreamWaitEvent(side->s, side->ready[0], 0);
        ws = (btr_stream_t)side->s;
      }
      // (both halves of dW_0 (n0, 3 + c) = [dW_x | dW_f] leave their split-K reductions in the
      // parameter's layout: was a launch of its own behind them, ppfl_assemble_kernel)
      reduce_unpad_next(4, 3, 3 + d.c, 0);
      BTR_TRY(btr_sa_gemm_tn(R, n0, 4, dy, n0, relx, 4, nullptr, nullptr, at_f(scratch, sc.pw[0]),
                             grads + p.dw[0], ws));
      reduce_unpad_next(0, 0);
      if (side) {
        (void)hipEventRecord(side->done[0], side->s);
        last_done = 0;
      }
      const bool pre = geom && geom->scatter_ws;
      void *ws2 = pre ? geom->scatter_ws : (char *)scratch + sc.scat;
      const int mode = pre ? kScatterReduce : kScatterBoth;
      if (p.compact)
        BTR_TRY(sac_scatter_ex(d.b, d.n, d.m, n0, n0, 0, dy, cp.cidx, cp.goff, S, ws2,
                               sc.scat_bytes, R, mode, hmain));
      else
        BTR_TRY(sa_scatter_ex(d.b, d.n, d.m, d.s, n0, n0, 0, d.radius_div, dy, idx, S, nullptr,
                              nullptr, ws2, sc.scat_bytes, mode, hmain));
      btr_sac_bind(nullptr);   // (products over the points: no compact rows)
      reduce_unpad_next(d.c, d.c, 3 + d.c, 3);
      BTR_TRY(btr_sa_gemm_tn(np, n0, d.c, S, n0, fcopy, d.c, nullptr, nullptr,
                             at_f(scratch, sc.ppfl_pw), grads + p.dw[0], stream));
      reduce_unpad_next(0, 0);
      if (d.need_dfeat && dfeat) {
        float *dfeat_cl = at_f(scratch, sc.dfeat_cl);
        BTR_TRY(btr_pm_gemm_nt(np, d.c, n0, S, n0, at_f(saved, p.wt[0]), n0, dfeat_cl, d.c,
                               nullptr, nullptr, nullptr, nullptr, stream));
        BTR_TRY(pm_out_add(d.b, d.n, d.c, d.c, dfeat_cl, nullptr, nullptr, 0, dfeat, nullptr,
                           dfeat_add, dfeat_add_bstride, hmain));
      }
      if (d.need_dxyz || d.need_dnew_xyz) {
        // d rel (rows x 4) = dY_0 W_x, then the coordinate part of the row-wise scatter alone (the
        // inverted lists are there: built with the sampling or by the scatter of S above)
        float *drel = at_f(scratch, sc.g[flip]);
        // (the other gradient buffer: a side-stream weight gradient may still be reading it)
        if (side && prev_done >= 1) (void)hipStreamWaitEvent(hmain, side->done[prev_done], 0);
        BTR_TRY(btr_sa_gemm_nt(R, 4, n0, dy, n0, at_f(saved, p.wt[0]) + (size_t)d.c * n0, n0, drel,
                               4, nullptr, nullptr, nullptr, stream));
        BTR_TRY(sa_scatter_ex(d.b, d.n, d.m, d.s, 0, 4, 1, d.radius_div, drel, idx, nullptr,
                              d.need_dxyz ? dxyz : nullptr, d.need_dnew_xyz ? dnew_xyz : nullptr,
                              ws2, sc.scat_bytes, kScatterReduce, hmain));
      }
      if (p.compact) btr_sac_bind(&cm);
      fused_any = true;   // (partials written on the main stream: see the join)
      break;
    }
    float *dw = grads + p.dw[l];
    float *pw = at_f(scratch, sc.pw[l]);
    // weight gradient of layer l: dY_l is final here -> fork
    btr_stream_t ws = stream;
    if (side) {
      (void)hipEventRecord(side->ready[l], hmain);
      (void)hipStreamWaitEvent(side->s, side->ready[l], 0);
      ws = (btr_stream_t)side->s;
    }
    // the first layer's gradient leaves without the zero columns of its 4-aligned input width:
    // dense [nl][3 + c] rows, what the parameter's gradient is (a strided view would be copied
    // by autograd: one launch per level at the very end of the backward)
    const int k_real = (d.use_xyz ? 3 : 0) + d.c;
    if (l == 0 && !p.recompute && k_real != k) reduce_unpad_next(k, k_real);
    if (p.recompute && l == 1)
      BTR_TRY(btr_sa_gemm_tn_rc(R, nl, k, dy, nl, x0, at_f(saved, p.w2[0]), pa, pb, pw, dw, ws));
    else if (pooled)
      BTR_TRY(btr_sa_gemm_tn_pool(R, nl, k, dy, nl, d.s, arg, dcl, alpha, beta, xsrc, ldx, pa, pb,
                                  pw, dw, ws));
    else
      BTR_TRY(btr_sa_gemm_tn(R, nl, k, dy, nl, xsrc, ldx, pa, pb, pw, dw, ws));
    reduce_unpad_next(0, 0);
    if (side) {
      (void)hipEventRecord(side->done[l], side->s);
      last_done = l;
    }
    if (l > 0 || any_in) {
      const float *wt = at_f(saved, p.wt[l]);
      float *g = at_f(scratch, sc.g[flip]);
      flip ^= 1;
      // g is the buffer dY_{l+1} lived in: its weight gradient must be through with it
      if (side && l + 1 <= L - 2) (void)hipStreamWaitEvent(hmain, side->done[l + 1], 0);
      // the first layer's input gradient when nobody asks for the coordinate part (a backbone
      // level: xyz carries no gradient): only the feature columns 3 .. 3 + c of dX0 are computed,
      // as a dense (rows, c) tile -- with k0p = 3 + c (132, 260) the xyz columns cost a whole
      // extra column block whose 124 other columns are padding
      const bool feat_only = l == 0 && !pooled && d.use_xyz && d.c > 0 &&
                             d.c % 4 == 0 && !d.need_dxyz && !d.need_dnew_xyz;
      const int gld = feat_only ? d.c : k;
      if (pooled)
        BTR_TRY(btr_sa_gemm_nt_pool(R, k, nl, dy, nl, wt, nl, g, k, d.s, arg, dcl, alpha, beta,
                                    stream));
      else if (feat_only)
        BTR_TRY(btr_sa_gemm_nt(R, d.c, nl, dy, nl, wt + (size_t)3 * nl, nl, g, d.c, nullptr,
                               nullptr, nullptr, stream));
      else
        BTR_TRY(btr_sa_gemm_nt(R, k, nl, dy, nl, wt, nl, g, k, nullptr, nullptr, nullptr,
                               stream));
      if (l > 0) {
        float *dg = grads + p.dgamma[l - 1], *db = grads + p.dbeta[l - 1];
        if (p.recompute && l == 1)
          BTR_TRY(btr_sa_bn_relu_bwd_rc(R, k, k, g, x0, at_f(saved, p.w2[0]), stat(0, 0),
                                        stat(0, 1), stat(0, 2), stat(0, 3), part, m1, m2, dg, db,
                                        at_f(scratch, sc.pw0), grads + p.dw[0], stream));
        else if (fusable(l - 1)) {   // the next layer applies the sums while it stages g
          BTR_TRY(btr_sa_bn_relu_bwd_sums(R, k, k, g, at_f(saved, p.y[l - 1]), stat(l - 1, 0),
                                          stat(l - 1, 1), stat(l - 1, 2), stat(l - 1, 3), part,
                                          m1, m2, dg, db, stream));
          lazy = true;
        } else
          BTR_TRY(btr_sa_bn_relu_bwd(R, k, k, g, at_f(saved, p.y[l - 1]), stat(l - 1, 0),
                                     stat(l - 1, 1), stat(l - 1, 2), stat(l - 1, 3), part, m1, m2,
                                     dg, db, stream));
        dy = g;
      } else {
        float *dfeat_cl = (d.need_dfeat && d.c > 0 && dfeat) ? at_f(scratch, sc.dfeat_cl) : nullptr;
        // the inverted neighbour lists: prepared with the sampling, or built here
        const bool pre = geom && geom->scatter_ws;
        void *ws2 = pre ? geom->scatter_ws : (char *)scratch + sc.scat;
        const int mode = pre ? kScatterReduce : kScatterBoth;
        if (p.compact) {
          if (dfeat_cl)
            BTR_TRY(sac_scatter_ex(d.b, d.n, d.m, d.c, gld, feat_only ? 0 : d.use_xyz, g, cp.cidx,
                                   cp.goff, dfeat_cl, ws2, sc.scat_bytes, R, mode, hmain));
        } else {
          BTR_TRY(sa_scatter_ex(d.b, d.n, d.m, d.s, d.c, gld, feat_only ? 0 : d.use_xyz,
                                d.radius_div, g, idx, dfeat_cl, d.need_dxyz ? dxyz : nullptr,
                                d.need_dnew_xyz ? dnew_xyz : nullptr, ws2, sc.scat_bytes, mode,
                                hmain));
        }
        if (dfeat_cl)
          BTR_TRY(pm_out_add(d.b, d.n, d.c, d.c, dfeat_cl, nullptr, nullptr, 0, dfeat, nullptr,
                             dfeat_add, dfeat_add_bstride, hmain));
      }
    }
  }
  if (side && fused_any) {   // the fused calls' partials were written on the main stream
    (void)hipEventRecord(side->ready[kMaxL], hmain);
    (void)hipStreamWaitEvent(side->s, side->ready[kMaxL], 0);
  }
  batch.flush();
  if (side && (last_done >= 0 || fused_any)) {   // join
    (void)hipEventRecord(side->done[kMaxL], side->s);
    (void)hipStreamWaitEvent(hmain, side->done[kMaxL], 0);
  }
  return check_launch("sa_layer_backward");
}

extern "C" {

// ================================================================== point-wise MLP chains
namespace {
struct PmBwdScratch {
  size_t g[2], part, m1, m2, pw[kMaxL], colsum, bytes;
};
PmBwdScratch pm_bwd_scratch(const btr_pm_chain_t &d, const btr_pm_plan_t &p) {
  PmBwdScratch s{};
  Bump b;
  int maxc = 0;
  for (int l = 0; l < d.layers; ++l) maxc = std::max(maxc, std::max(p.np[l], p.kin[l]));
  s.g[0] = b.floats((size_t)p.rows * maxc);
  s.g[1] = b.floats((size_t)p.rows * maxc);
  int part_rows = 1024;   // (as in sa_bwd_scratch: one row per chunk of a fused call)
  for (int l = 1; l < d.layers; ++l)
    if (btr_sa_bwd_fused_supported(p.rows, p.np[l], p.kin[l]))
      part_rows = std::max(part_rows, btr_sa_bwd_fused_chunks(p.rows, p.np[l], p.kin[l]));
  s.part = b.floats((size_t)part_rows * 2 * maxc);
  s.m1 = b.floats(maxc);
  s.m2 = b.floats(maxc);
  for (int l = 0; l < d.layers; ++l)   // split-K partials, one region per layer
    s.pw[l] = b.floats((size_t)wgrad_partial_chunks(p.rows, p.np[l], p.kin[l]) * p.np[l] *
                       p.kin[l]);
  s.colsum = b.floats((size_t)d.b * cdiv(d.n, 64) * p.np[d.layers - 1]);   // per 64-row tile
  s.bytes = b.off;
  return s;
}
}  // namespace

// C = f(A) . W^T for a chain layer whose saved weight block holds W [n][ldw = k] followed by its
// bf16 planes (pm_w_floats): the small-M kernel where it applies, else gemm_nt_kernel
static int pm_gemm_nt_auto(int rows, int n, int k, const float *a, int lda, const float *w, int ldw,
                           float *c, int ldc, const float *pa, const float *pb, float *part,
                           const float *bias, btr_stream_t stream) {
  if (ldw == k && btr_pm_gemm_nt_sm_supported(rows, n, k))
    return btr_pm_gemm_nt_sm(rows, n, k, a, lda, w + (size_t)n * k, c, ldc, pa, pb, part, bias,
                             stream);
  return btr_pm_gemm_nt(rows, n, k, a, lda, w, ldw, c, ldc, pa, pb, part, bias, stream);
}

int btr_pm_chain_plan(const btr_pm_chain_t *dp, btr_pm_plan_t *p) {
  BTR_REQUIRE(dp && p, "pm_chain_plan: null pointer");
  const btr_pm_chain_t &d = *dp;
  BTR_REQUIRE(d.layers >= 1 && d.layers <= kMaxL, "pm_chain_plan: %d layers", d.layers);
  BTR_REQUIRE(d.b > 0 && d.n > 0 && d.c > 0, "pm_chain_plan: bad sizes");
  BTR_REQUIRE((long long)d.b * d.n < (1ll << 31), "pm_chain_plan: too many rows");
  std::memset(p, 0, sizeof(*p));
  const int L = d.layers;
  p->rows = d.b * d.n;
  int maxn = 0;
  for (int l = 0; l < L; ++l) {
    BTR_REQUIRE(d.width[l] > 0, "pm_chain_plan: width");
    p->np[l] = ceil4(d.width[l]);
    p->kin[l] = l == 0 ? ceil4(d.c) : p->np[l - 1];   // input rows zero-padded to 4 columns
    BTR_REQUIRE(!d.has_bn[l] || (d.width[l] % 4 == 0 && d.width[l] <= 512),
                "pm_chain_plan: BatchNorm layer of width %d", d.width[l]);
    BTR_REQUIRE(d.has_bn[l] || l == L - 1, "pm_chain_plan: only the last layer may lack BN");
    maxn = std::max(maxn, p->np[l]);
  }
  Bump sv;
  p->x0 = sv.floats((size_t)p->rows * p->kin[0]);
  for (int l = 0; l < L; ++l) {
    p->y[l] = sv.floats((size_t)p->rows * p->np[l]);
    p->w2[l] = sv.floats(pm_w_floats(p->np[l], p->kin[l], false));   // + its bf16 planes
    p->wt[l] = sv.floats(pm_w_floats(p->np[l], p->kin[l], true));
    p->stats[l] = sv.floats((size_t)4 * p->np[l]);
  }
  p->saved_bytes = sv.off;
  p->fwd_scratch_bytes = up(sizeof(float) * ((size_t)btr_pm_gemm_grid(p->rows) * 2 * maxn + maxn) +
                            sizeof(unsigned) * kBnTickets * kMaxL);   // part, bias_pad, tickets
  p->bwd_scratch_bytes = pm_bwd_scratch(d, *p).bytes;
  size_t g = 0;
  for (int l = 0; l < L; ++l) {
    p->dw[l] = g;
    g += (size_t)p->np[l] * p->kin[l];
    p->dgamma[l] = g;
    g += p->np[l];
    p->dbeta[l] = g;
    g += p->np[l];
  }
  // bias gradients last: the ones in front of a BatchNorm are exact zeros (one memset)
  for (int l = 0; l < L; ++l) {
    p->dbias[l] = g;
    g += p->np[l];
  }
  p->grads_floats = g;
  return BTR_OK;
}

int btr_pm_chain_forward(const btr_pm_chain_t *dp, const btr_pm_plan_t *pp, const float *x_bcn,
                         const float *x_cl, float *out, float *out_cl, void *saved,
                         void *scratch, btr_stream_t stream) {
  BTR_REQUIRE(dp && pp && (x_bcn || x_cl) && (out || out_cl) && saved && scratch,
              "pm_chain_forward: null pointer");
  const btr_pm_chain_t &d = *dp;
  const btr_pm_plan_t &p = *pp;
  const int L = d.layers, rows = p.rows;
  hipStream_t hs = as_stream(stream);

  PrepArgs pa{};
  pa.layers = L;
  int blocks = 0;
  for (int l = 0; l < L; ++l) {
    BTR_REQUIRE(d.w[l] && (!d.has_bn[l] || (d.gamma[l] && d.beta[l])),
                "pm_chain_forward: layer %d parameters", l);
    pa.w[l] = d.w[l];
    pa.w2[l] = at_f(saved, p.w2[l]);
    pa.wt[l] = at_f(saved, p.wt[l]);
    pa.n[l] = d.width[l];
    pa.np[l] = p.np[l];
    pa.kraw[l] = l == 0 ? d.c : d.width[l - 1];
    pa.kin[l] = p.kin[l];
    pa.first[l] = blocks;
    blocks += cdiv((long long)ceil16(p.np[l]) * ceil16(p.kin[l]), 256);
    const bool track = d.has_bn[l] && d.running_mean[l];
    pa.nbt[l] = track ? d.num_batches_tracked[l] : nullptr;
    pa.wp2[l] = pm_planes(at_f(saved, p.w2[l]), p.np[l], p.kin[l]);
    pa.wpt[l] = pm_planes(at_f(saved, p.wt[l]), p.np[l], p.kin[l]);
  }
  pa.first[L] = blocks;
  const int grid = btr_pm_gemm_grid(rows);
  int maxn = 0;
  for (int l = 0; l < L; ++l) maxn = std::max(maxn, p.np[l]);
  float *part = (float *)scratch;
  float *bias_pad = part + (size_t)grid * 2 * maxn;
  unsigned *tickets = reinterpret_cast<unsigned *>(bias_pad + maxn);
  pa.tickets = tickets;
  pa.ntickets = kBnTickets * kMaxL;
  if (!d.has_bn[L - 1] && d.bias[L - 1] && p.np[L - 1] != d.width[L - 1]) {
    pa.pbias_src = d.bias[L - 1];   // padded by the same launch (was a memset + a copy)
    pa.pbias_dst = bias_pad;
    pa.pbias_n = d.width[L - 1];
    pa.pbias_np = p.np[L - 1];
  }
  const bool batched = prep_batch_take(pa, !bnfin_rows_ok(rows) && !pa.pbias_dst);
  if (prep_batch_collecting()) return BTR_OK;
  if (!batched) hipLaunchKernelGGL(prep_weights_kernel<kMaxL>, dim3(blocks), dim3(256), 0, hs, pa);

  const int k0 = p.kin[0];
  BTR_REQUIRE(!x_cl || k0 == d.c, "pm_chain_forward: x_cl needs a channel count that is a multiple of 4");
  const float *A = x_cl;
  if (!A) {
    float *x0 = at_f(saved, p.x0);
    BTR_TRY(btr_pm_rows(d.b, d.n, d.c, k0, x_bcn, x0, stream));
    A = x0;
  }  // else: the caller keeps x_cl alive and hands it to the backward again
  int lda = k0;
  const float *pscale = nullptr, *pshift = nullptr;
  // a caller that only wants the channel-last rows of a chain whose last layer is a bare
  // convolution (out == NULL: the decoder stack's position embeddings) gets them from the last
  // GEMM itself -- no layout launch behind it
  const bool direct = !out && out_cl && !d.has_bn[L - 1] && p.np[L - 1] == d.width[L - 1];
  BTR_REQUIRE(out || direct, "pm_chain_forward: out == NULL needs a bare last layer of a width "
                              "that is a multiple of 4");
  for (int l = 0; l < L; ++l) {
    const int np = p.np[l], k = p.kin[l];
    const float *w2 = at_f(saved, p.w2[l]);
    float *y = (direct && l == L - 1) ? out_cl : at_f(saved, p.y[l]);
    if (d.has_bn[l]) {
      float *st = at_f(saved, p.stats[l]);
      // (a convolution bias in front of the BatchNorm is skipped: it only moves the running mean)
      const float *rbias = d.running_mean[l] ? d.bias[l] : nullptr;
      const bool fin_fused = bnfin_arm(BnFin{
          tickets + kBnTickets * l, d.gamma[l], d.beta[l], st, st + np, st + 2 * np, st + 3 * np,
          d.running_mean[l], d.running_var[l], rbias, d.width[l], (double)rows, d.eps[l],
          d.momentum[l]}, rows);
      BTR_TRY(pm_gemm_nt_auto(rows, np, k, A, lda, w2, k, y, np, pscale, pshift, part, nullptr,
                             stream));
      if (!fin_fused)
        BTR_TRY(bn_finalize_bias(np, grid, (double)rows, d.eps[l], d.momentum[l], part,
                                 d.gamma[l], d.beta[l], st, st + np, st + 2 * np, st + 3 * np,
                                 d.running_mean[l], d.running_var[l], rbias, d.width[l], hs));
      pscale = st;
      pshift = st + np;
    } else {
      const float *bp = d.bias[l];
      if (bp && np != d.width[l]) bp = bias_pad;   // (padded by prep_weights_kernel)
      BTR_TRY(pm_gemm_nt_auto(rows, np, k, A, lda, w2, k, y, np, pscale, pshift, nullptr, bp,
                             stream));
      pscale = pshift = nullptr;
    }
    A = y;
    lda = np;
  }
  if (!direct)
    BTR_TRY(btr_pm_out(d.b, d.n, d.width[L - 1], p.np[L - 1], A, pscale, pshift, pscale ? 1 : 0,
                       out, out_cl, stream));
  return check_launch("pm_chain_forward");
}

int btr_pm_chain_backward(const btr_pm_chain_t *dp, const btr_pm_plan_t *pp, const float *x_cl,
                          const float *dout, void *saved, float *grads, float *dx,
                          void *scratch, btr_stream_t stream) {
  return pm_chain_backward_rows(dp, pp, x_cl, dout, nullptr, nullptr, saved, grads, dx, nullptr,
                                scratch, stream);
}

}  // extern "C"

namespace btr {
// (internal.hpp) the chain backward with row-form hand-overs on either side
int pm_chain_backward_rows(const btr_pm_chain_t *dp, const btr_pm_plan_t *pp, const float *x_cl,
                           const float *dout, const float *a0, const float *a1, void *saved,
                           float *grads, float *dx, float *dx_rows, void *scratch,
                           btr_stream_t stream) {
  BTR_REQUIRE(dp && pp && (dout || a0) && saved && grads && scratch,
              "pm_chain_backward: null pointer");
  const btr_pm_chain_t &d = *dp;
  const btr_pm_plan_t &p = *pp;
  const int L = d.layers, rows = p.rows;
  hipStream_t hs = as_stream(stream);
  const PmBwdScratch sc = pm_bwd_scratch(d, p);
  float *part = at_f(scratch, sc.part), *m1 = at_f(scratch, sc.m1), *m2 = at_f(scratch, sc.m2);
  float *colsum = at_f(scratch, sc.colsum);
  auto stat = [&](int l, int which) { return at_f(saved, p.stats[l]) + which * p.np[l]; };
  const int nl = d.width[L - 1], npl = p.np[L - 1];
  int flip = 0;
  float *g = at_f(scratch, sc.g[flip]);
  flip ^= 1;
  // (+ the bias gradients: zero in front of a BatchNorm, and the padding of the others.)  The
  // bias gradient of a bare last layer = column sums of dout: per-tile partials from the same
  // launch, then one small reduction
  const bool bare_bias = !d.has_bn[L - 1] && d.bias[L - 1];
  if (dout)
    BTR_TRY(pm_rows_zero(d.b, d.n, nl, npl, dout, g, grads + p.dbias[0],
                         (int)(p.grads_floats - p.dbias[0]), bare_bias ? colsum : nullptr, hs));
  else
    BTR_TRY(pm_rows_in(d.b, d.n, nl, npl, a0, a1, g, grads + p.dbias[0],
                       (int)(p.grads_floats - p.dbias[0]), bare_bias ? colsum : nullptr, hs));
  // `lazy` / `fusable`: as in btr_sa_layer_backward -- a hidden layer behind a BatchNorm runs its
  // whole backward (dW_l, dZ_{l-1}, BatchNorm_{l-1}'s sums) as one btr_sa_bwd_fused call, which
  // applies BatchNorm_l's backward from the finalised sums (m1, m2) while it stages dZ_l
  const char *cf_env = getenv("BTR_CHAIN_FUSED");   // (read per call: the tests toggle it)
  const bool chain_fused_off = cf_env && cf_env[0] == '0';
  auto fusable = [&](int l) {
    return !chain_fused_off && l >= 1 && d.has_bn[l] &&
           btr_sa_bwd_fused_supported(rows, p.np[l], p.kin[l]) != 0;
  };
  bool lazy = false, fused_any = false;
  if (d.has_bn[L - 1]) {
    if (fusable(L - 1)) {
      BTR_TRY(btr_sa_bn_relu_bwd_sums(rows, npl, npl, g, at_f(saved, p.y[L - 1]), stat(L - 1, 0),
                                      stat(L - 1, 1), stat(L - 1, 2), stat(L - 1, 3), part, m1, m2,
                                      grads + p.dgamma[L - 1], grads + p.dbeta[L - 1], stream));
      lazy = true;
    } else
      BTR_TRY(btr_sa_bn_relu_bwd(rows, npl, npl, g, at_f(saved, p.y[L - 1]), stat(L - 1, 0),
                                 stat(L - 1, 1), stat(L - 1, 2), stat(L - 1, 3), part, m1, m2,
                                 grads + p.dgamma[L - 1], grads + p.dbeta[L - 1], stream));
  } else if (bare_bias) {
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(npl, 16)), dim3(256), 0, hs,
                       d.b * cdiv(d.n, 64), npl, colsum, grads + p.dbias[L - 1]);
  }
  float *dy = g;
  // (few-row chains -- GroupFree3D's 1024-row heads and position embeddings -- stay on one
  // stream: the fork / join calls cost the host more than the overlap of two 10 us kernels
  // returns, 12.4 -> 12.0 ms per GroupFree3D step)
  SideStream *side = rows >= side_min_rows(false) ? wgrad_side(hs) : nullptr;
  int last_done = -1;
  ReduceBatchScope batch(side ? side->s : hs);   // (see btr_sa_layer_backward)
  for (int l = L - 1; l >= 0; --l) {
    const int np = p.np[l], k = p.kin[l];
    const float *xsrc = l == 0 ? (x_cl ? x_cl : at_f(saved, p.x0)) : at_f(saved, p.y[l - 1]);
    const int ldx = l == 0 ? p.kin[0] : p.np[l - 1];
    const float *pa = l == 0 ? nullptr : stat(l - 1, 0);
    const float *pb = l == 0 ? nullptr : stat(l - 1, 1);
    if (lazy && fusable(l)) {
      float *gn = at_f(scratch, sc.g[flip]);
      flip ^= 1;
      // gn is the buffer dY_{l+1} lived in: its weight gradient must be through with it
      if (side && l + 1 <= L - 1 && last_done == l + 1)
        (void)hipStreamWaitEvent(hs, side->done[l + 1], 0);
      BTR_TRY(btr_sa_bwd_fused(rows, np, k, dy, np, at_f(saved, p.y[l]), stat(l, 0), stat(l, 1),
                               stat(l, 2), stat(l, 3), m1, m2, 0, nullptr, nullptr, nullptr,
                               nullptr, xsrc, ldx, nullptr, pa, pb, stat(l - 1, 2), stat(l - 1, 3),
                               at_f(saved, p.wt[l]), np, gn, k, at_f(scratch, sc.pw[l]),
                               grads + p.dw[l], part, m1, m2, grads + p.dgamma[l - 1],
                               grads + p.dbeta[l - 1], stream));
      dy = gn;
      fused_any = true;   // (and still lazy: dy = dZ_{l-1} with BatchNorm_{l-1}'s sums in m1, m2)
      continue;
    }
    if (lazy) {   // a consumer that wants dY_l itself
      BTR_TRY(btr_sa_bn_relu_bwd_apply(rows, np, np, dy, at_f(saved, p.y[l]), stat(l, 0),
                                       stat(l, 1), stat(l, 2), stat(l, 3), m1, m2, stream));
      lazy = false;
    }
    btr_stream_t ws = stream;   // weight gradient on the side stream (see SideStream)
    if (side) {
      (void)hipEventRecord(side->ready[l], hs);
      (void)hipStreamWaitEvent(side->s, side->ready[l], 0);
      ws = (btr_stream_t)side->s;
    }
    BTR_TRY(btr_sa_gemm_tn(rows, np, k, dy, np, xsrc, ldx, pa, pb, at_f(scratch, sc.pw[l]),
                           grads + p.dw[l], ws));
    if (side) {
      (void)hipEventRecord(side->done[l], side->s);
      last_done = l;
    }
    if (l > 0 || d.need_dx) {
      float *gn = at_f(scratch, sc.g[flip]);
      flip ^= 1;
      if (l == 0 && dx_rows) gn = dx_rows;   // the caller's rows (leading dimension kin[0])
      // gn is the buffer dY_{l+1} lived in: its weight gradient must be through with it
      if (side && l + 1 <= L - 1 && last_done == l + 1)
        (void)hipStreamWaitEvent(hs, side->done[l + 1], 0);
      BTR_TRY(pm_gemm_nt_auto(rows, k, np, dy, np, at_f(saved, p.wt[l]), np, gn, k, nullptr,
                             nullptr, nullptr, nullptr, stream));
      if (l > 0) {
        if (fusable(l - 1)) {   // the next layer applies the sums while it stages gn
          BTR_TRY(btr_sa_bn_relu_bwd_sums(rows, k, k, gn, at_f(saved, p.y[l - 1]),
                                          stat(l - 1, 0), stat(l - 1, 1), stat(l - 1, 2),
                                          stat(l - 1, 3), part, m1, m2, grads + p.dgamma[l - 1],
                                          grads + p.dbeta[l - 1], stream));
          lazy = true;
        } else
          BTR_TRY(btr_sa_bn_relu_bwd(rows, k, k, gn, at_f(saved, p.y[l - 1]), stat(l - 1, 0),
                                     stat(l - 1, 1), stat(l - 1, 2), stat(l - 1, 3), part, m1, m2,
                                     grads + p.dgamma[l - 1], grads + p.dbeta[l - 1], stream));
        dy = gn;
      } else {
        BTR_REQUIRE(dx || dx_rows, "pm_chain_backward: dx missing");
        if (dx)
          BTR_TRY(btr_pm_out(d.b, d.n, d.c, p.kin[0], gn, nullptr, nullptr, 0, dx, nullptr,
                             stream));
      }
    }
  }
  if (side && fused_any) {   // the fused calls' partials were written on the main stream
    (void)hipEventRecord(side->ready[kMaxL], hs);
    (void)hipStreamWaitEvent(side->s, side->ready[kMaxL], 0);
  }
  batch.flush();
  if (side && (last_done >= 0 || fused_any)) {   // join
    (void)hipEventRecord(side->done[kMaxL], side->s);
    (void)hipStreamWaitEvent(hs, side->done[kMaxL], 0);
  }
  return check_launch("pm_chain_backward");
}
}  // namespace btr

extern "C" {

int btr_vote_assemble(int b, int n, int c, const float *net_cl, int ld_net,
                      const float *seed_xyz, const float *seed_cl, float *vote_xyz,
                      float *vote_feat_bcn, float *vote_feat_cl, float *nrm,
                      btr_stream_t stream) {
  if (b <= 0 || n <= 0 || c <= 0) return BTR_OK;
  BTR_REQUIRE(net_cl && seed_xyz && seed_cl && vote_xyz && vote_feat_bcn && vote_feat_cl &&
                  ld_net >= c + 3 && b < 65536,
              "vote_assemble: bad arguments");
  if (nrm) {   // with the L2 normalisation of the features
    BTR_REQUIRE(c <= 256, "vote_assemble: normalised form covers <= 256 channels, got %d", c);
    const dim3 grid(cdiv(n, 32), b);
#define BTR_VA(CJ)                                                                          \
  hipLaunchKernelGGL((vote_assemble_norm_kernel<CJ>), grid, dim3(256), 0, as_stream(stream), \
                     n, c, ld_net, net_cl, seed_xyz, seed_cl, vote_xyz, vote_feat_bcn,      \
                     vote_feat_cl, nrm)
    if (c <= 64) BTR_VA(8);
    else if (c <= 128) BTR_VA(16);
    else BTR_VA(32);
#undef BTR_VA
    return check_launch("vote_assemble(norm)");
  }
  hipLaunchKernelGGL(vote_assemble_kernel, dim3(cdiv(n, 64), cdiv(c, 64), b), dim3(256), 0,
                     as_stream(stream), n, c, ld_net, net_cl, seed_xyz, seed_cl, vote_xyz,
                     vote_feat_bcn, vote_feat_cl);
  return check_launch("vote_assemble");
}

int btr_vote_assemble_bwd(int b, int n, int c, const float *dvote_xyz, const float *dvote_feat_bcn,
                          const float *vote_feat_cl, const float *nrm, float *dnet_bcn,
                          float *dseed_bcn, btr_stream_t stream) {
  if (b <= 0 || n <= 0 || c <= 0) return BTR_OK;
  BTR_REQUIRE(dvote_xyz && dvote_feat_bcn && dnet_bcn && b < 65536 && c + 3 < 65536,
              "vote_assemble_bwd: bad arguments");
  if (nrm) {
    BTR_REQUIRE(vote_feat_cl && dseed_bcn && c <= 256, "vote_assemble_bwd: normalised form");
    const dim3 grid(cdiv(n, 32), b);
#define BTR_VA(CJ)                                                                              \
  hipLaunchKernelGGL((vote_assemble_norm_bwd_kernel<CJ>), grid, dim3(256), 0, as_stream(stream), \
                     n, c, dvote_xyz, dvote_feat_bcn, vote_feat_cl, nrm, dnet_bcn, dseed_bcn)
    if (c <= 64) BTR_VA(8);
    else if (c <= 128) BTR_VA(16);
    else BTR_VA(32);
#undef BTR_VA
    return check_launch("vote_assemble_bwd(norm)");
  }
  hipLaunchKernelGGL(vote_assemble_bwd_kernel, dim3(cdiv(n, 256), c + 3, b), dim3(256), 0,
                     as_stream(stream), n, c, dvote_xyz, dvote_feat_bcn, dnet_bcn);
  return check_launch("vote_assemble_bwd");
}

}  // extern "C"
