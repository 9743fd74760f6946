// decoder.hip -- one GroupFree3D decoder layer per call (forward / backward).
//
// reference: detection/GroupFree3D/models/transformer.py:11-76 (TransformerDecoderLayer:
// post-norm self-attention over the query points, cross-attention onto the seed points, FFN;
// the position embeddings are added to queries, keys AND values) with the vendored
// models/multi_head_attention.py for the two attention modules.
//
// Everything works on batch-major channel-last rows: query rows (B*Pq, E), key rows (B*Pk, E).
//     qp0 = x0 + qpos                       qkv = qp0 Win^T + bin            (B*Pq, 3E)
//     a1  = attention(qkv)                  o1  = a1 Wo^T + bo
//     x1  = LN1(x0 + drop(o1))              qp1 = x1 + qpos
//     q2  = qp1 Wq^T + bq                   kp  = key + kpos;  kv = kp Wkv^T + bkv  (B*Pk, 2E)
//     a2  = attention(q2, kv)               o2  = a2 Wo2^T + bo2
//     x2  = LN2(x1 + drop(o2))              h   = drop(relu(x2 W1^T + b1))   (B*Pq, F)
//     f   = h W2^T + b2                     x3  = LN3(x2 + drop(f))
// GEMMs: the NT / TN kernels of sa_mlp.hip (btr_pm_gemm_nt with its bias epilogue, btr_sa_gemm_tn
// for the weight gradients); attention core: attention.hip on the projection outputs in place.
// Written here: residual + dropout + LayerNorm (forward / backward) as one kernel each, the
// ReLU + dropout of the FFN, the bias / LayerNorm parameter gradients (one partial-sum launch and
// one final launch for all of them), weight transposes for the input-gradient GEMMs (one
// launch), and the layout kernels at the (B, E, P) boundary of the module.
#include <algorithm>
#include <cstring>

#include "internal.hpp"

namespace btr {
namespace {

constexpr size_t kAlign = 256;
inline size_t up(size_t v) { return (v + kAlign - 1) / kAlign * kAlign; }
struct Bump {
  size_t off = 0;
  size_t take(size_t bytes) {
    const size_t at = off;
    off = up(off + bytes);
    return at;
  }
  size_t floats(size_t n) { return take(n * sizeof(float)); }
};
inline float *at_f(void *base, size_t off) { return (float *)((char *)base + off); }

#define BTR_TRY(call)              \
  do {                             \
    const int rc_ = (call);        \
    if (rc_ != BTR_OK) return rc_; \
  } while (0)

// ------------------------------------------------------------------------------- dropout masks
// keep(element) = hash(seed, *step, which dropout of the layer, element index) >= threshold:
// the backward regenerates the mask of the forward from the same (seed, step); a replayed HIP
// graph draws fresh masks because *step lives in device memory.
struct Drop {
  unsigned thr;         // 0: no dropout
  float keep_inv;
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
__device__ __forceinline__ unsigned long long drop_key(const Drop &d) {
  const unsigned long long s = d.step ? (unsigned long long)*d.step : 0ull;
  return d.seed ^ (s * 0x9E3779B97F4A7C15ull);
}
__device__ __forceinline__ float drop1(unsigned long long key, unsigned long long e,
                                       const Drop &d, float v) {
  return mix32(key + e * 0xD6E8FEB86659FD93ull) >= d.thr ? v * d.keep_inv : 0.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// ------------------------------------------------------------------------------ small kernels
// out = a + b (n4 float4 elements)
__global__ __launch_bounds__(256) void add2_kernel(long long n4, const float4 *__restrict__ a,
                                                   const float4 *__restrict__ b,
                                                   float4 *__restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 x = a[i], y = b[i];
  out[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}

// h = dropout(relu(h)) in place (the FFN's hidden rows; the bias is already in)
__global__ __launch_bounds__(256) void relu_drop_kernel(long long n4, float4 *__restrict__ h,
                                                        Drop d) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 v = h[i];
  v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  if (d.thr) {
    const unsigned long long key = drop_key(d), e = (unsigned long long)i * 4;
    v.x = drop1(key, e, d, v.x); v.y = drop1(key, e + 1, d, v.y);
    v.z = drop1(key, e + 2, d, v.z); v.w = drop1(key, e + 3, d, v.w);
  }
  h[i] = v;
}
// dh (in place) = dL/d(pre-activation): h > 0 exactly where the unit was positive AND kept
__global__ __launch_bounds__(256) void relu_drop_bwd_kernel(long long n4, float4 *__restrict__ dh,
                                                            const float4 *__restrict__ h,
                                                            float keep_inv) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 g = dh[i];
  const float4 v = h[i];
  g.x = v.x > 0.f ? g.x * keep_inv : 0.f; g.y = v.y > 0.f ? g.y * keep_inv : 0.f;
  g.z = v.z > 0.f ? g.z * keep_inv : 0.f; g.w = v.w > 0.f ? g.w * keep_inv : 0.f;
  dh[i] = g;
}

// out (B, C, P) = (a + b + c) with a, b, c rows (B*P, C); b, c optional.  32 x 32 tiles.
__global__ __launch_bounds__(256) void rows_to_bcp_kernel(int P, int C,
                                                          const float *__restrict__ a,
                                                          const float *__restrict__ b,
                                                          const float *__restrict__ c,
                                                          float *__restrict__ out) {
  __shared__ float t[32][33];
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32, bb = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, ch = c0 + tx;
    float v = 0.f;
    if (p < P && ch < C) {
      const size_t at = ((size_t)bb * P + p) * C + ch;
      v = a[at];
      if (b) v += b[at];
      if (c) v += c[at];
    }
    t[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int ch = c0 + r, p = p0 + tx;
    if (p < P && ch < C) out[((size_t)bb * C + ch) * P + p] = t[tx][r];
  }
}

// ---------------------------------------------------------------- residual + dropout + LayerNorm
// One wave per row (E <= 256 * NV columns, 4 per lane and step):
//   v = res + dropout(o);  xhat = (v - mean) * rstd;  y = xhat * gamma + beta;  ypos = y + pos
struct LnFwd {
  int rows, e;
  const float *res, *o;
  int on;               // o = obias + sum of `on` planes, `ostride` floats apart (split-K GEMM)
  long long ostride;
  const float *obias;   // optional
  Drop drop;
  const float *gamma, *beta;
  float eps;
  float *y, *xhat, *rstd;
  const float *pos;   // optional
  float *ypos;
};
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnFwd a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const int q = a.e >> 2;
  const size_t base = (size_t)row * q;
  const float4 *res4 = (const float4 *)a.res + base, *o4 = (const float4 *)a.o + base;
  const unsigned long long key = a.drop.thr ? drop_key(a.drop) : 0ull;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + 64 * i;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < q) {
      const float4 r = res4[j];
      float4 o = o4[j];
      for (int z = 1; z < a.on; ++z) {
        const float4 t = ((const float4 *)(a.o + (size_t)z * a.ostride))[base + j];
        o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
      }
      if (a.obias) {
        const float4 t = ((const float4 *)a.obias)[j];
        o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
      }
      if (a.drop.thr) {
        const unsigned long long e = ((unsigned long long)row * q + j) * 4;
        o.x = drop1(key, e, a.drop, o.x); o.y = drop1(key, e + 1, a.drop, o.y);
        o.z = drop1(key, e + 2, a.drop, o.z); o.w = drop1(key, e + 3, a.drop, o.w);
      }
      v[i] = make_float4(r.x + o.x, r.y + o.y, r.z + o.z, r.w + o.w);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(s) / (float)a.e;
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (lane + 64 * i < q) {
      const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
      s2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  const float rstd = 1.f / sqrtf(wave_sum(s2) / (float)a.e + a.eps);
  if (lane == 0) a.rstd[row] = rstd;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + 64 * i;
    if (j < q) {
      const float4 g = ((const float4 *)a.gamma)[j], bt = ((const float4 *)a.beta)[j];
      const float4 xh = make_float4((v[i].x - mean) * rstd, (v[i].y - mean) * rstd,
                                    (v[i].z - mean) * rstd, (v[i].w - mean) * rstd);
      const float4 y = make_float4(xh.x * g.x + bt.x, xh.y * g.y + bt.y, xh.z * g.z + bt.z,
                                   xh.w * g.w + bt.w);
      ((float4 *)a.xhat)[base + j] = xh;
      ((float4 *)a.y)[base + j] = y;
      if (a.ypos) {
        const float4 p = ((const float4 *)a.pos)[base + j];
        ((float4 *)a.ypos)[base + j] = make_float4(y.x + p.x, y.y + p.y, y.z + p.z, y.w + p.w);
      }
    }
  }
}

// backward: dy = g0 + g1 + g2 (the gradients that reach the LayerNorm output: g1, g2 optional)
//   gy = dy * gamma;  dv = rstd * (gy - mean(gy) - xhat * mean(gy * xhat))
//   dres = dv (residual branch);  dout = dropout mask * dv (the branch that was normalised in)
//   part[block][0..E) = sum_rows dy * xhat (dgamma), part[block][E..2E) = sum_rows dy (dbeta)
struct LnBwd {
  int rows, e;
  const float *g0, *g1, *g2;
  int g1n;              // g1 = sum of `g1n` planes, `g1stride` floats apart (split-K GEMM)
  long long g1stride;
  const float *xhat, *rstd, *gamma;
  Drop drop;
  float *dres, *dout;
  float *part;
};
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwd a) {
  extern __shared__ float red[];   // [4][2 * e]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int q = a.e >> 2;
  const unsigned long long key = a.drop.thr ? drop_key(a.drop) : 0ull;
  const float inv_e = 1.f / (float)a.e;
  float4 ag[NV], ab[NV], gm[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ag[i] = ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int j = lane + 64 * i;
    gm[i] = j < q ? ((const float4 *)a.gamma)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int row = blockIdx.x * 4 + w; row < a.rows; row += gridDim.x * 4) {
    const size_t base = (size_t)row * q;
    float4 dy[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = lane + 64 * i;
      dy[i] = xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < q) {
        dy[i] = ((const float4 *)a.g0)[base + j];
        for (int z = 0; z < a.g1n; ++z) {
          const float4 t = ((const float4 *)(a.g1 + (size_t)z * a.g1stride))[base + j];
          dy[i].x += t.x; dy[i].y += t.y; dy[i].z += t.z; dy[i].w += t.w;
        }
        if (a.g2) {
          const float4 t = ((const float4 *)a.g2)[base + j];
          dy[i].x += t.x; dy[i].y += t.y; dy[i].z += t.z; dy[i].w += t.w;
        }
        xh[i] = ((const float4 *)a.xhat)[base + j];
        const float4 gy = make_float4(dy[i].x * gm[i].x, dy[i].y * gm[i].y, dy[i].z * gm[i].z,
                                      dy[i].w * gm[i].w);
        s1 += (gy.x + gy.y) + (gy.z + gy.w);
        s2 += (gy.x * xh[i].x + gy.y * xh[i].y) + (gy.z * xh[i].z + gy.w * xh[i].w);
        ag[i].x += dy[i].x * xh[i].x; ag[i].y += dy[i].y * xh[i].y;
        ag[i].z += dy[i].z * xh[i].z; ag[i].w += dy[i].w * xh[i].w;
        ab[i].x += dy[i].x; ab[i].y += dy[i].y; ab[i].z += dy[i].z; ab[i].w += dy[i].w;
      }
    }
    const float m1 = wave_sum(s1) * inv_e, m2 = wave_sum(s2) * inv_e;
    const float rstd = a.rstd[row];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = lane + 64 * i;
      if (j < q) {
        float4 dv;
        dv.x = rstd * (dy[i].x * gm[i].x - m1 - xh[i].x * m2);
        dv.y = rstd * (dy[i].y * gm[i].y - m1 - xh[i].y * m2);
        dv.z = rstd * (dy[i].z * gm[i].z - m1 - xh[i].z * m2);
        dv.w = rstd * (dy[i].w * gm[i].w - m1 - xh[i].w * m2);
        ((float4 *)a.dres)[base + j] = dv;
        if (a.drop.thr) {
          const unsigned long long e = (base + j) * 4;
          dv.x = drop1(key, e, a.drop, dv.x); dv.y = drop1(key, e + 1, a.drop, dv.y);
          dv.z = drop1(key, e + 2, a.drop, dv.z); dv.w = drop1(key, e + 3, a.drop, dv.w);
        }
        ((float4 *)a.dout)[base + j] = dv;
      }
    }
  }
  float *mine = red + (size_t)w * 2 * a.e;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + 64 * i;
    if (j < q) {
      ((float4 *)mine)[j] = ag[i];
      ((float4 *)(mine + a.e))[j] = ab[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * a.e; c += 256)
    a.part[(size_t)blockIdx.x * 2 * a.e + c] =
        (red[c] + red[2 * a.e + c]) + (red[4 * a.e + c] + red[6 * a.e + c]);
}

// ------------------------------------------------------------------- batched small reductions
constexpr int kMaxSeg = 12;
constexpr int kCsRows = 128;   // rows per partial sum
// stage 1: part[chunk][c] = sum of the chunk's rows of g (rows, c) with leading dimension ld
struct ColsumArgs {
  const float *g[kMaxSeg];
  float *part[kMaxSeg];
  int rows[kMaxSeg], c[kMaxSeg], ld[kMaxSeg];
  int first[kMaxSeg + 1];
  int count;
};
__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumArgs a) {
  __shared__ float red[4][64];
  // the segment of this block: unrolled scan with STATIC indices (a run-time index into the
  // by-value argument arrays would spill the struct to scratch)
  const float *g = a.g[0];
  float *part = a.part[0];
  int rows = a.rows[0], c = a.c[0], ld = a.ld[0], first = 0;
#pragma unroll
  for (int i = 1; i < kMaxSeg; ++i)
    if (i < a.count && (int)blockIdx.x >= a.first[i]) {
      g = a.g[i]; part = a.part[i]; rows = a.rows[i]; c = a.c[i]; ld = a.ld[i];
      first = a.first[i];
    }
  const int tiles = (c + 63) / 64;
  const int t = (int)blockIdx.x - first, chunk = t / tiles, tile = t - chunk * tiles;
  const int col = tile * 64 + (int)(threadIdx.x & 63), sub = (int)(threadIdx.x >> 6);
  const int r0 = chunk * kCsRows, r1 = min(rows, r0 + kCsRows);
  float acc = 0.f;
  if (col < c)
    for (int r = r0 + sub; r < r1; r += 4) acc += g[(size_t)r * ld + col];
  red[sub][threadIdx.x & 63] = acc;
  __syncthreads();
  if (sub == 0 && col < c)
    part[(size_t)chunk * c + col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) +
                                    (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// stage 2: out[c] = sum over chunks of part[chunk][c]
struct FinalArgs {
  const float *part[kMaxSeg];
  float *out[kMaxSeg];
  int chunks[kMaxSeg], c[kMaxSeg];
  int first[kMaxSeg + 1];
  int count;
};
__global__ __launch_bounds__(256) void colsum_final_multi_kernel(FinalArgs a) {
  const float *part = a.part[0];
  float *out = a.out[0];
  int chunks = a.chunks[0], c = a.c[0], first = 0;
#pragma unroll
  for (int i = 1; i < kMaxSeg; ++i)
    if (i < a.count && (int)blockIdx.x >= a.first[i]) {
      part = a.part[i]; out = a.out[i]; chunks = a.chunks[i]; c = a.c[i]; first = a.first[i];
    }
  // 64 columns per block, 4 partial sums per column in flight
  __shared__ float red[4][64];
  const int col = ((int)blockIdx.x - first) * 64 + (int)(threadIdx.x & 63);
  const int sub = (int)(threadIdx.x >> 6);
  float acc = 0.f;
  if (col < c)
    for (int k = sub; k < chunks; k += 4) acc += part[(size_t)k * c + col];
  red[sub][threadIdx.x & 63] = acc;
  __syncthreads();
  if (sub == 0 && col < c)
    out[col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) +
               (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// Up to 6 weight matrices in one launch (32 x 32 tiles), per matrix src (rows, cols):
//   tr = 1: dst (cols, rows) = src^T in f32 (dst may be NULL) and, when planes != NULL, the three
//           bf16 pieces of src^T as planes [3][cols][kp], kp = ceil16(rows)
//   tr = 0: planes [3][rows][kp] of src itself, kp = ceil16(cols)
// (the small-M GEMM kernel of sa_mlp.hip loads its weight fragments straight from such planes; the
// columns between the width and kp hold zeros)
constexpr int kMaxTr = 6;
struct TransposeArgs {
  const float *src[kMaxTr];
  float *dst[kMaxTr];
  __bf16 *planes[kMaxTr];
  int rows[kMaxTr], cols[kMaxTr], tr[kMaxTr];
  int first[kMaxTr + 1];
  int count;
};
__device__ __forceinline__ void split3(float v, __bf16 &h, __bf16 &m, __bf16 &l) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  auto widen = [](b2 b) { return __uint_as_float(__builtin_bit_cast(unsigned, b) << 16); };
  const b2 bh = __builtin_convertvector(f2{v, 0.f}, b2);
  const float r1 = v - widen(bh);
  const b2 bm = __builtin_convertvector(f2{r1, 0.f}, b2);
  const float r2 = r1 - widen(bm);
  const b2 bl = __builtin_convertvector(f2{r2, 0.f}, b2);
  h = bh.x; m = bm.x; l = bl.x;
}
inline int ceil16(int v) { return (v + 15) / 16 * 16; }
__global__ __launch_bounds__(256) void transpose_multi_kernel(TransposeArgs a) {
  __shared__ float t[32][33];
  const float *src = a.src[0];
  float *dst = a.dst[0];
  __bf16 *planes = a.planes[0];
  int rows = a.rows[0], cols = a.cols[0], first = 0, tr = a.tr[0];
#pragma unroll
  for (int i = 1; i < kMaxTr; ++i)
    if (i < a.count && (int)blockIdx.x >= a.first[i]) {
      src = a.src[i]; dst = a.dst[i]; planes = a.planes[i]; rows = a.rows[i]; cols = a.cols[i];
      first = a.first[i]; tr = a.tr[i];
    }
  const int tc = (cols + 31) / 32;
  const int tl = (int)blockIdx.x - first, trow = tl / tc;
  const int r0 = trow * 32, c0 = (tl - trow * tc) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8)
    t[r][tx] = (r0 + r < rows && c0 + tx < cols) ? src[(size_t)(r0 + r) * cols + c0 + tx] : 0.f;
  __syncthreads();
  if (tr) {
    const int kp = (rows + 15) / 16 * 16;
    for (int r = ty; r < 32; r += 8) {
      if (c0 + r >= cols) continue;
      const float v = t[tx][r];
      if (dst && r0 + tx < rows) dst[(size_t)(c0 + r) * rows + r0 + tx] = v;
      if (planes && r0 + tx < kp) {
        __bf16 h, m, l;
        split3(v, h, m, l);
        const size_t at = (size_t)(c0 + r) * kp + r0 + tx, ps = (size_t)cols * kp;
        planes[at] = h; planes[ps + at] = m; planes[2 * ps + at] = l;
      }
    }
  } else if (planes) {
    const int kp = (cols + 15) / 16 * 16;
    for (int r = ty; r < 32; r += 8) {
      if (r0 + r >= rows || c0 + tx >= kp) continue;
      __bf16 h, m, l;
      split3(t[r][tx], h, m, l);
      const size_t at = (size_t)(r0 + r) * kp + c0 + tx, ps = (size_t)rows * kp;
      planes[at] = h; planes[ps + at] = m; planes[2 * ps + at] = l;
    }
  }
}

// ------------------------------------------------------------------------------ host helpers
Drop make_drop(const btr_decoder_layer_t &d, int which) {
  Drop r{};
  const float p = d.dropout;
  if (p > 0.f) {
    r.thr = (unsigned)((double)p * 4294967296.0);
    if (r.thr == 0u) r.thr = 1u;
    r.keep_inv = 1.f / (1.f - p);
  } else {
    r.thr = 0u;
    r.keep_inv = 1.f;
  }
  r.seed = d.seed + 0xA24BAED4963EE407ull * (unsigned long long)(which + 1);
  r.step = d.step;
  return r;
}
// the seed of attention call `which` (0 self, 1 cross), apart from the four dropouts above
unsigned long long attn_seed(const btr_decoder_layer_t &d, int which) {
  return d.seed + 0x9FB21C651E98DF25ull * (unsigned long long)(which + 5);
}

int ln_forward(hipStream_t s, LnFwd a) {
  const dim3 grid(cdiv(a.rows, 4));
  if (a.e <= 256) hipLaunchKernelGGL(ln_fwd_kernel<1>, grid, dim3(256), 0, s, a);
  else if (a.e <= 512) hipLaunchKernelGGL(ln_fwd_kernel<2>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(ln_fwd_kernel<4>, grid, dim3(256), 0, s, a);
  return check_launch("decoder layer_norm");
}
int ln_blocks(int rows) { return std::max(1, std::min(256, cdiv(rows, 16))); }
int ln_backward(hipStream_t s, LnBwd a) {
  const dim3 grid(ln_blocks(a.rows));
  const size_t lds = sizeof(float) * 8 * a.e;
  if (a.e <= 256) hipLaunchKernelGGL(ln_bwd_kernel<1>, grid, dim3(256), lds, s, a);
  else if (a.e <= 512) hipLaunchKernelGGL(ln_bwd_kernel<2>, grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL(ln_bwd_kernel<4>, grid, dim3(256), lds, s, a);
  return check_launch("decoder layer_norm backward");
}
int add2(hipStream_t s, long long n, const float *a, const float *b, float *out) {
  hipLaunchKernelGGL(add2_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, s, n / 4,
                     (const float4 *)a, (const float4 *)b, (float4 *)out);
  return check_launch("decoder add");
}
int rows_to_bcp(hipStream_t s, int b, int p, int c, const float *a0, const float *a1,
                const float *a2, float *out) {
  hipLaunchKernelGGL(rows_to_bcp_kernel, dim3(cdiv(p, 32), cdiv(c, 32), b), dim3(256), 0, s, p, c,
                     a0, a1, a2, out);
  return check_launch("decoder rows_to_bcp");
}

// BTR_DECODER_SM=0: the projections / input-gradient GEMMs on gemm_nt_kernel's 64 x 128 tiles as
// before; default: the small-M kernel on weight planes made once per call (1 024-row query GEMMs
// of 288 - 864 columns: 13 - 35 us -> 6 - 12 us, profiles/r05_gf_*).  Needs e % 16 == 0 (sub-blocks
// of the packed in_proj weights are addressed inside one set of planes).
bool decoder_sm(const btr_decoder_layer_t &d) {
  const char *e = getenv("BTR_DECODER_SM");
  return !(e && e[0] == '0') && d.e % 16 == 0 && d.ff % 16 == 0;
}
// ... and only where it measured faster (same box, r05_gf_eager2 against r05_gf_eager): few rows and
// a moderate output width.  1 024 x 288 x 288: 13.4 -> 8.6 us, 1 024 x 864 x 288 18 -> 13.6,
// 1 024 x 288 x 864 34.4 -> 20.8; but 4 096 x 576 x 288 (keys) 21 -> 28 and 1 024 x 2 048 x 288
// (feed-forward) 22 -> 24: every 32-row tile re-reads its 64 weight rows from L2.
bool sm_shape(int rows, int n_out) { return rows <= 2048 && n_out <= 1024; }
// forward planes: in_proj (3e, e), out_proj (e, e) of both attentions, linear1 (ff, e)
struct FwdPlanes {
  size_t sa_in, sa_out, ca_in, ca_out, l1, bytes;
};
FwdPlanes fwd_planes(const btr_decoder_layer_t &d, size_t base) {
  FwdPlanes f{};
  Bump b;
  b.off = base;
  const size_t e = d.e, ff = d.ff;
  f.sa_in = b.take(3 * 3 * e * e * 2); f.sa_out = b.take(3 * e * e * 2);
  f.ca_in = b.take(3 * 3 * e * e * 2); f.ca_out = b.take(3 * e * e * 2);
  f.l1 = b.take(3 * ff * e * 2);
  f.bytes = b.off - base;
  return f;
}
struct BwdScratch {
  size_t dx3, dres3, df, dh, dx2f, dres2, do2, da2, dq2, dkv, dqp1, dkp, dres1, do1, da1, dqkv,
      dqp0, dsum, dsum2;
  size_t t_sa_in, t_sa_out, t_ca_in, t_ca_out, t_l1, t_l2;
  size_t p_sa_in, p_sa_out, p_ca_in, p_ca_out, p_l2;   // bf16 planes of the transposes (decoder_sm)
  size_t pw[8];
  size_t cs[8];      // column-sum partials of the 8 bias gradients
  size_t lnp[3];     // LayerNorm gamma / beta partials
  size_t bytes;
};
// the 8 linear maps of the layer: (output width n, input width k, rows)
struct Lin {
  int n, k, rows;
};
void linears(const btr_decoder_layer_t &d, Lin *l) {
  const int rq = d.b * d.pq, rk = d.b * d.pk, e = d.e;
  l[0] = {3 * e, e, rq};   // self-attention in_proj
  l[1] = {e, e, rq};       // self-attention out_proj
  l[2] = {e, e, rq};       // cross-attention in_proj, q rows
  l[3] = {2 * e, e, rk};   // cross-attention in_proj, k / v rows
  l[4] = {e, e, rq};       // cross-attention out_proj
  l[5] = {d.ff, e, rq};    // linear1
  l[6] = {e, d.ff, rq};    // linear2
}
BwdScratch bwd_scratch(const btr_decoder_layer_t &d) {
  BwdScratch s{};
  Bump b;
  const size_t rq = (size_t)d.b * d.pq, rk = (size_t)d.b * d.pk, e = d.e, f = d.ff;
  s.dx3 = b.floats(rq * e); s.dres3 = b.floats(rq * e); s.df = b.floats(rq * e);
  s.dh = b.floats(rq * f);
  s.dx2f = b.floats(rq * e * (size_t)pm_splitk_slices((int)rq, d.e, d.ff));
  s.dres2 = b.floats(rq * e);
  s.do2 = b.floats(rq * e); s.da2 = b.floats(rq * e); s.dq2 = b.floats(rq * e);
  s.dkv = b.floats(rk * 2 * e); s.dqp1 = b.floats(rq * e); s.dkp = b.floats(rk * e);
  s.dres1 = b.floats(rq * e); s.do1 = b.floats(rq * e); s.da1 = b.floats(rq * e);
  s.dqkv = b.floats(rq * 3 * e); s.dqp0 = b.floats(rq * e);
  s.dsum = b.floats((size_t)d.b * d.heads * d.pq);
  s.dsum2 = b.floats((size_t)d.b * d.heads * d.pq);
  s.t_sa_in = b.floats(3 * e * e); s.t_sa_out = b.floats(e * e);
  s.t_ca_in = b.floats(3 * e * e); s.t_ca_out = b.floats(e * e);
  s.t_l1 = b.floats(f * e); s.t_l2 = b.floats(f * e);
  if (decoder_sm(d)) {
    s.p_sa_in = b.take(3 * 3 * e * e * 2); s.p_sa_out = b.take(3 * e * e * 2);
    s.p_ca_in = b.take(3 * 3 * e * e * 2); s.p_ca_out = b.take(3 * e * e * 2);
    s.p_l2 = b.take(3 * f * e * 2);
  }
  Lin l[7];
  linears(d, l);
  for (int i = 0; i < 7; ++i) {
    s.pw[i] = b.floats((size_t)btr_sa_gemm_tn_chunks(l[i].rows, l[i].n, l[i].k) * l[i].n * l[i].k);
    s.cs[i] = b.floats((size_t)cdiv(l[i].rows, kCsRows) * l[i].n);
  }
  for (int i = 0; i < 3; ++i) s.lnp[i] = b.floats((size_t)ln_blocks((int)rq) * 2 * e);
  s.bytes = b.off;
  return s;
}

}  // namespace
}  // namespace btr

using namespace btr;

extern "C" {

int btr_decoder_layer_plan(const btr_decoder_layer_t *dp, btr_decoder_plan_t *p) {
  BTR_REQUIRE(dp && p, "decoder_layer_plan: null pointer");
  const btr_decoder_layer_t &d = *dp;
  BTR_REQUIRE(d.b > 0 && d.pq > 0 && d.pk > 0 && d.e > 0 && d.heads > 0 && d.ff > 0,
              "decoder_layer_plan: bad sizes");
  BTR_REQUIRE(d.e % d.heads == 0 && btr_attention_supported(d.e / d.heads),
              "decoder_layer_plan: width %d with %d heads (head width 1..64)", d.e, d.heads);
  BTR_REQUIRE(d.e % 4 == 0 && d.ff % 4 == 0 && d.e <= 1024,
              "decoder_layer_plan: e=%d (<= 1024) and ff=%d must be multiples of 4", d.e, d.ff);
  BTR_REQUIRE((long long)d.b * std::max(d.pq, d.pk) < (1ll << 24) &&
                  (long long)d.b * d.heads <= 65535,
              "decoder_layer_plan: too many rows / heads");
  BTR_REQUIRE(d.dropout >= 0.f && d.dropout < 1.f, "decoder_layer_plan: dropout %g", d.dropout);
  std::memset(p, 0, sizeof(*p));
  const size_t rq = (size_t)d.b * d.pq, rk = (size_t)d.b * d.pk, e = d.e, f = d.ff;
  p->rq = (int)rq;
  p->rk = (int)rk;
  Bump sv;
  p->qp0 = sv.floats(rq * e); p->qkv = sv.floats(rq * 3 * e); p->a1 = sv.floats(rq * e);
  p->lse1 = sv.floats((size_t)d.b * d.heads * d.pq);
  p->xh1 = sv.floats(rq * e); p->rs1 = sv.floats(rq); p->x1 = sv.floats(rq * e);
  p->qp1 = sv.floats(rq * e); p->q2 = sv.floats(rq * e);
  p->kp = sv.floats(rk * e); p->kv = sv.floats(rk * 2 * e); p->a2 = sv.floats(rq * e);
  p->lse2 = sv.floats((size_t)d.b * d.heads * d.pq);
  p->xh2 = sv.floats(rq * e); p->rs2 = sv.floats(rq); p->x2 = sv.floats(rq * e);
  p->h = sv.floats(rq * f); p->xh3 = sv.floats(rq * e); p->rs3 = sv.floats(rq);
  p->saved_bytes = sv.off;
  Bump fs;   // forward scratch: the branch output that is normalised in (split-K: its planes)
  fs.floats(rq * e * (size_t)pm_splitk_slices((int)rq, d.e, d.ff));
  if (decoder_sm(d)) fs.off += fwd_planes(d, fs.off).bytes;
  p->fwd_scratch_bytes = fs.off;
  p->bwd_scratch_bytes = bwd_scratch(d).bytes;
  size_t g = 0;   // flat gradients, floats
  auto take = [&](size_t n) { const size_t at = g; g += n; return at; };
  p->g_sa_in_w = take(3 * e * e); p->g_sa_in_b = take(3 * e);
  p->g_sa_out_w = take(e * e); p->g_sa_out_b = take(e);
  p->g_ca_in_w = take(3 * e * e); p->g_ca_in_b = take(3 * e);
  p->g_ca_out_w = take(e * e); p->g_ca_out_b = take(e);
  p->g_lin1_w = take(f * e); p->g_lin1_b = take(f);
  p->g_lin2_w = take(e * f); p->g_lin2_b = take(e);
  for (int i = 0; i < 3; ++i) p->g_ln[i] = take(2 * e);
  p->grads_floats = g;
  return BTR_OK;
}

int btr_decoder_layer_forward(const btr_decoder_layer_t *dp, const btr_decoder_plan_t *pp,
                              const float *x_cl, const float *key_cl, const float *qpos_cl,
                              const float *kpos_cl, float *out_bcp, float *out_cl, void *saved,
                              void *scratch, btr_stream_t stream) {
  return decoder_layer_forward_ex(dp, pp, x_cl, key_cl, qpos_cl, kpos_cl, out_bcp, out_cl, saved,
                                  scratch, 0, kDecoderFwdSelf | kDecoderFwdRest, stream);
}

int btr_decoder_layer_backward(const btr_decoder_layer_t *dp, const btr_decoder_plan_t *pp,
                               const float *x_cl, const float *key_cl, const float *qpos_cl,
                               const float *kpos_cl, const float *dout_bcp, void *saved,
                               float *grads, float *dx_bcp, float *dkey_bcp, float *dqpos_bcp,
                               void *scratch, btr_stream_t stream) {
  return decoder_layer_backward_rows(dp, pp, x_cl, key_cl, qpos_cl, kpos_cl, dout_bcp, nullptr,
                                     nullptr, nullptr, saved, grads, dx_bcp, dkey_bcp, dqpos_bcp,
                                     nullptr, scratch, stream);
}

}  // extern "C"

namespace btr {
// (internal.hpp) the key / value rows of the cross-attention on their own: kp = key + kpos,
// kv = kp Wkv^T + bkv (saved: kp, kv) -- they depend on nothing the layer computes, so a caller
// may issue them ahead on another stream.  Only where the layer itself would use the plain NT
// GEMM for them (decoder_kv_separable): the small-M kernel reads weight planes that the layer's
// first launch prepares in its scratch, and the two kernels add in a different order.
bool decoder_kv_separable(const btr_decoder_layer_t *dp, const btr_decoder_plan_t *pp) {
  return !(decoder_sm(*dp) && sm_shape(pp->rk, 2 * dp->e));
}
int decoder_layer_kv(const btr_decoder_layer_t *dp, const btr_decoder_plan_t *pp,
                     const float *key_cl, const float *kpos_cl, void *saved,
                     btr_stream_t stream) {
  BTR_REQUIRE(dp && pp && key_cl && saved, "decoder_layer_kv: null pointer");
  const btr_decoder_layer_t &d = *dp;
  const btr_decoder_plan_t &p = *pp;
  BTR_REQUIRE(decoder_kv_separable(dp, pp), "decoder_layer_kv: not separable at these sizes");
  hipStream_t hs = as_stream(stream);
  const int rk = p.rk, e = d.e;
  float *kp = at_f(saved, p.kp), *kv = at_f(saved, p.kv);
  const float *ksrc = key_cl;
  if (kpos_cl) {
    BTR_TRY(add2(hs, (long long)rk * e, key_cl, kpos_cl, kp));
    ksrc = kp;
  }
  BTR_TRY(btr_pm_gemm_nt(rk, 2 * e, e, ksrc, e, d.ca_in_w + (size_t)e * e, e, kv, 2 * e, nullptr,
                         nullptr, nullptr, d.ca_in_b + e, stream));
  return check_launch("decoder_layer_kv");
}

// (internal.hpp) btr_decoder_layer_forward; kv_ready != 0: the cross-attention's key / value rows
// (saved: kp, kv) are there already -- decoder_layer_kv ran, on whatever stream, and the caller
// ordered it before this call.  parts: kDecoderFwdSelf = the weight planes, the self-attention
// with its LayerNorm and the cross-attention's query rows (nothing of it reads a key);
// kDecoderFwdRest = everything from the key / value rows on; both = the whole layer.  A caller
// that waits for the keys between the two calls hides that wait behind the first
int decoder_layer_forward_ex(const btr_decoder_layer_t *dp, const btr_decoder_plan_t *pp,
                             const float *x_cl, const float *key_cl, const float *qpos_cl,
                             const float *kpos_cl, float *out_bcp, float *out_cl, void *saved,
                             void *scratch, int kv_ready, int parts, btr_stream_t stream) {
  BTR_REQUIRE(dp && pp && x_cl && key_cl && out_cl && saved && scratch,
              "decoder_layer_forward: null pointer");
  const bool part_self = (parts & kDecoderFwdSelf) != 0, part_rest = (parts & kDecoderFwdRest) != 0;
  const btr_decoder_layer_t &d = *dp;
  const btr_decoder_plan_t &p = *pp;
  hipStream_t hs = as_stream(stream);
  const int rq = p.rq, rk = p.rk, e = d.e, f = d.ff, hd = e / d.heads;
  const float scale = 1.f / sqrtf((float)hd);
  float *o = (float *)scratch;   // branch output in front of each LayerNorm
  float *qp0 = at_f(saved, p.qp0), *qkv = at_f(saved, p.qkv), *a1 = at_f(saved, p.a1);
  float *x1 = at_f(saved, p.x1), *qp1 = at_f(saved, p.qp1), *q2 = at_f(saved, p.q2);
  float *kp = at_f(saved, p.kp), *kv = at_f(saved, p.kv), *a2 = at_f(saved, p.a2);
  float *x2 = at_f(saved, p.x2), *h = at_f(saved, p.h);

  // ---- the weights as bf16 planes (one launch), for the small-M GEMM kernel
  const bool sm = decoder_sm(d);
  FwdPlanes fp{};
  if (sm) {
    Bump fs;
    fs.floats((size_t)rq * e * (size_t)pm_splitk_slices(rq, d.e, d.ff));
    fp = fwd_planes(d, fs.off);
  }
  if (sm && part_self) {
    TransposeArgs t{};
    const float *src[5] = {d.sa_in_w, d.sa_out_w, d.ca_in_w, d.ca_out_w, d.lin1_w};
    const size_t off[5] = {fp.sa_in, fp.sa_out, fp.ca_in, fp.ca_out, fp.l1};
    const int rr[5] = {3 * e, e, 3 * e, e, f};
    int blocks = 0;
    for (int i = 0; i < 5; ++i) {
      t.src[i] = src[i]; t.dst[i] = nullptr; t.planes[i] = (__bf16 *)((char *)scratch + off[i]);
      t.rows[i] = rr[i]; t.cols[i] = e; t.tr[i] = 0;
      t.first[i] = blocks;
      blocks += cdiv(rr[i], 32) * cdiv(e, 32);
    }
    t.first[5] = blocks;
    t.count = 5;
    hipLaunchKernelGGL(transpose_multi_kernel, dim3(blocks), dim3(256), 0, hs, t);
  }
  // C (rows, n) = A (rows, k) . W^T + bias, W (n, k) rows `row0`.. of a weight whose planes start
  // at `poff` (pitch e: every weight here has k = e columns)
  auto proj = [&](int rows, int n, const float *a, const float *w, size_t poff, int row0,
                  int nall, float *c, const float *bias) {
    if (sm && sm_shape(rows, n))
      return pm_gemm_nt_planes(rows, n, e, a, e,
                               (const __bf16 *)((const char *)scratch + poff) + (size_t)row0 * e,
                               e, (long long)nall * e, c, n, bias, hs);
    return btr_pm_gemm_nt(rows, n, e, a, e, w + (size_t)row0 * e, e, c, n, nullptr, nullptr,
                          nullptr, bias, stream);
  };
  if (part_self) {
    // ---- self-attention
    const float *qsrc = x_cl;
    if (qpos_cl) {
      BTR_TRY(add2(hs, (long long)rq * e, x_cl, qpos_cl, qp0));
      qsrc = qp0;
    }
    BTR_TRY(proj(rq, 3 * e, qsrc, d.sa_in_w, fp.sa_in, 0, 3 * e, qkv, d.sa_in_b));
    BTR_TRY(attention_fwd_strided(d.pq, d.pq, d.b, d.heads, hd, qkv, 3 * e,
                                  (long long)d.pq * 3 * e, qkv + e, qkv + 2 * e, 3 * e,
                                  (long long)d.pq * 3 * e, a1, e, (long long)d.pq * e,
                                  at_f(saved, p.lse1), scale, d.dropout, attn_seed(d, 0), d.step,
                                  stream));
    BTR_TRY(proj(rq, e, a1, d.sa_out_w, fp.sa_out, 0, e, o, d.sa_out_b));
    {
      LnFwd a{rq, e, x_cl, o, 1, 0, nullptr, make_drop(d, 0), d.ln_w[0], d.ln_b[0], d.ln_eps[0],
              x1, at_f(saved, p.xh1), at_f(saved, p.rs1), qpos_cl, qpos_cl ? qp1 : nullptr};
      BTR_TRY(ln_forward(hs, a));
    }
    // ---- cross-attention: the query rows
    const float *q2src = qpos_cl ? qp1 : x1;
    BTR_TRY(proj(rq, e, q2src, d.ca_in_w, fp.ca_in, 0, 3 * e, q2, d.ca_in_b));
  }
  if (!part_rest) return check_launch("decoder_layer_forward");
  if (!kv_ready) {
    const float *ksrc = key_cl;
    if (kpos_cl) {
      BTR_TRY(add2(hs, (long long)rk * e, key_cl, kpos_cl, kp));
      ksrc = kp;
    }
    BTR_TRY(proj(rk, 2 * e, ksrc, d.ca_in_w, fp.ca_in, e, 3 * e, kv, d.ca_in_b + e));
  }
  BTR_TRY(attention_fwd_strided(d.pq, d.pk, d.b, d.heads, hd, q2, e, (long long)d.pq * e, kv,
                                kv + e, 2 * e, (long long)d.pk * 2 * e, a2, e,
                                (long long)d.pq * e, at_f(saved, p.lse2), scale, d.dropout,
                                attn_seed(d, 1), d.step, stream));
  BTR_TRY(proj(rq, e, a2, d.ca_out_w, fp.ca_out, 0, e, o, d.ca_out_b));
  {
    LnFwd a{rq, e, x1, o, 1, 0, nullptr, make_drop(d, 1), d.ln_w[1], d.ln_b[1], d.ln_eps[1], x2,
            at_f(saved, p.xh2), at_f(saved, p.rs2), nullptr, nullptr};
    BTR_TRY(ln_forward(hs, a));
  }
  // ---- feed-forward
  BTR_TRY(proj(rq, f, x2, d.lin1_w, fp.l1, 0, f, h, d.lin1_b));
  hipLaunchKernelGGL(relu_drop_kernel, dim3(cdiv((long long)rq * f / 4, 256)), dim3(256), 0, hs,
                     (long long)rq * f / 4, (float4 *)h, make_drop(d, 2));
  const int sk = pm_splitk_slices(rq, e, f);   // linear2: 288 columns, reduction over ff
  if (sk > 1)
    BTR_TRY(pm_gemm_nt_splitk(rq, e, f, h, f, d.lin2_w, f, o, (long long)rq * e, sk, hs));
  else
    BTR_TRY(btr_pm_gemm_nt(rq, e, f, h, f, d.lin2_w, f, o, e, nullptr, nullptr, nullptr,
                           d.lin2_b, stream));
  {
    LnFwd a{rq, e, x2, o, sk, (long long)rq * e, sk > 1 ? d.lin2_b : nullptr, make_drop(d, 3),
            d.ln_w[2], d.ln_b[2], d.ln_eps[2], out_cl,
            at_f(saved, p.xh3), at_f(saved, p.rs3), nullptr, nullptr};
    BTR_TRY(ln_forward(hs, a));
  }
  if (out_bcp) BTR_TRY(rows_to_bcp(hs, d.b, d.pq, e, out_cl, nullptr, nullptr, out_bcp));
  return check_launch("decoder_layer_forward");
}

// (internal.hpp) btr_decoder_layer_backward with the gradients handed over as channel-last rows:
// the output gradient is dout_bcp (transposed here) or the row operands (g0 + g1) + g2 (g1, g2
// optional); `out` (optional) names caller-owned row buffers for d res1 / d qp0 / d qp1 / d kp --
// dx = res1 + qp0, dqpos = qp0 + qp1, dkey = kp -- instead of the layer's scratch, so a caller that
// consumes rows (csrc/gf_stack.hip) passes dx_bcp / dkey_bcp / dqpos_bcp = NULL and no layout
// kernel runs.
int decoder_layer_backward_rows(const btr_decoder_layer_t *dp, const btr_decoder_plan_t *pp,
                                const float *x_cl, const float *key_cl, const float *qpos_cl,
                                const float *kpos_cl, const float *dout_bcp, const float *g0,
                                const float *g1, const float *g2, void *saved, float *grads,
                                float *dx_bcp, float *dkey_bcp, float *dqpos_bcp,
                                const DecoderRowsOut *out, void *scratch, btr_stream_t stream) {
  return decoder_layer_backward_parts(dp, pp, x_cl, key_cl, qpos_cl, kpos_cl, dout_bcp, g0, g1, g2,
                                      saved, grads, dx_bcp, dkey_bcp, dqpos_bcp, out, scratch,
                                      kDecoderBwdChain | kDecoderBwdRest | kDecoderBwdKey, stream);
}

// The layer's backward in two parts that may run on two streams (csrc/gf_stack.hip):
//   kDecoderBwdChain: what the gradient of the layer's INPUT waits for -- the LayerNorm backwards,
//       the input-gradient GEMMs, the attention backwards (writes every dY of the seven linear
//       maps into the scratch on its way);
//   kDecoderBwdKey: the cross-attention's d k / d v (88 us at 256 queries x 1 024 keys: the
//       layer's input gradient needs d q only) and the key / value rows' input gradient (d kp:
//       only the keys and their position embedding wait for it);
//   kDecoderBwdRest: what only the parameters wait for -- the seven weight gradients with their
//       reduction, the bias and LayerNorm parameter gradients.  Reads the d k / d v rows: after
//       kDecoderBwdKey.
// Key and Rest read what the chain left in the SAME scratch: issue them after the chain (same
// stream, or others ordered behind it), Rest behind Key, and keep the scratch until they are
// through.
// All three = the one-call backward: the same launches with the same operands, the later parts'
// after the chain's (they were interleaved; no result depends on that).
int decoder_layer_backward_parts(const btr_decoder_layer_t *dp, const btr_decoder_plan_t *pp,
                                 const float *x_cl, const float *key_cl, const float *qpos_cl,
                                 const float *kpos_cl, const float *dout_bcp, const float *g0,
                                 const float *g1, const float *g2, void *saved, float *grads,
                                 float *dx_bcp, float *dkey_bcp, float *dqpos_bcp,
                                 const DecoderRowsOut *out, void *scratch, int parts,
                                 btr_stream_t stream) {
  BTR_REQUIRE(dp && pp && x_cl && key_cl && saved && grads && scratch,
              "decoder_layer_backward: null pointer");
  BTR_REQUIRE(!(parts & kDecoderBwdChain) || dout_bcp || g0,
              "decoder_layer_backward: no output gradient");
  const btr_decoder_layer_t &d = *dp;
  const btr_decoder_plan_t &p = *pp;
  hipStream_t hs = as_stream(stream);
  const int rq = p.rq, rk = p.rk, e = d.e, f = d.ff, hd = e / d.heads;
  const float scale = 1.f / sqrtf((float)hd);
  const BwdScratch sc = bwd_scratch(d);
  auto S = [&](size_t off) { return at_f(scratch, off); };
  const float *qp0 = qpos_cl ? at_f(saved, p.qp0) : x_cl;
  const float *qkv = at_f(saved, p.qkv), *a1 = at_f(saved, p.a1), *x1 = at_f(saved, p.x1);
  const float *qp1 = qpos_cl ? at_f(saved, p.qp1) : x1, *q2 = at_f(saved, p.q2);
  const float *kp = kpos_cl ? at_f(saved, p.kp) : key_cl, *kv = at_f(saved, p.kv);
  const float *a2 = at_f(saved, p.a2), *x2 = at_f(saved, p.x2), *h = at_f(saved, p.h);
  Lin lin[7];
  linears(d, lin);
  const bool sm = decoder_sm(d);
  // dX (rows, k) = dY (rows, n) . W (n, k), as an NT GEMM on wt = W^T (k, n) (leading dim ldw)
  // (planes of wt: `poff` = start of the transposed weight's planes, col0 = first column of the
  // block, nall = its rows: the pitch is ceil16(ldw), the plane stride nall * pitch)
  auto dgrad = [&](int rows, int n, int k, const float *g, const float *wt, int ldw, float *dx,
                   size_t poff, int col0, int nall) {
    if (sm && sm_shape(rows, k) && n % 16 == 0 && col0 % 8 == 0)
      return pm_gemm_nt_planes(rows, k, n, g, n,
                               (const __bf16 *)((const char *)scratch + poff) + col0, ceil16(ldw),
                               (long long)nall * ceil16(ldw), dx, k, nullptr, hs);
    return btr_pm_gemm_nt(rows, k, n, g, n, wt, ldw, dx, k, nullptr, nullptr, nullptr, nullptr,
                          stream);
  };
  // the four row tensors the module's input gradients are made of: the layer's scratch, or the
  // caller's buffers
  float *r_dres1 = out && out->dres1 ? out->dres1 : S(sc.dres1);
  float *r_dqp0 = out && out->dqp0 ? out->dqp0 : S(sc.dqp0);
  float *r_dqp1 = out && out->dqp1 ? out->dqp1 : S(sc.dqp1);
  float *r_dkp = out && out->dkp ? out->dkp : S(sc.dkp);

  if (parts & kDecoderBwdChain) {
    {  // transposed weights: the input-gradient GEMMs are NT GEMMs on W^T
      TransposeArgs t{};
      const float *src[kMaxTr] = {d.sa_in_w, d.sa_out_w, d.ca_in_w, d.ca_out_w, d.lin1_w, d.lin2_w};
      float *dst[kMaxTr] = {S(sc.t_sa_in), S(sc.t_sa_out), S(sc.t_ca_in),
                            S(sc.t_ca_out), S(sc.t_l1), S(sc.t_l2)};
      const int rr[kMaxTr] = {3 * e, e, 3 * e, e, f, e}, cc[kMaxTr] = {e, e, e, e, e, f};
      const size_t pl[kMaxTr] = {sc.p_sa_in, sc.p_sa_out, sc.p_ca_in, sc.p_ca_out, 0, sc.p_l2};
      int blocks = 0;
      for (int i = 0; i < kMaxTr; ++i) {
        t.src[i] = src[i]; t.dst[i] = dst[i]; t.rows[i] = rr[i]; t.cols[i] = cc[i];
        t.tr[i] = 1;
        t.planes[i] = (sm && i != 4) ? (__bf16 *)((char *)scratch + pl[i]) : nullptr;
        t.first[i] = blocks;
        blocks += cdiv(rr[i], 32) * cdiv(cc[i], 32);
      }
      t.first[kMaxTr] = blocks;
      t.count = kMaxTr;
      hipLaunchKernelGGL(transpose_multi_kernel, dim3(blocks), dim3(256), 0, hs, t);
    }
    // ---- LayerNorm 3, feed-forward
    if (dout_bcp) {
      BTR_TRY(btr_pm_rows(d.b, d.pq, e, e, dout_bcp, S(sc.dx3), stream));
      g0 = S(sc.dx3);
      g1 = g2 = nullptr;
    }
    {
      LnBwd a{rq, e, g0, g1, g2, g1 ? 1 : 0, 0, at_f(saved, p.xh3), at_f(saved, p.rs3),
              d.ln_w[2], make_drop(d, 3), S(sc.dres3), S(sc.df), S(sc.lnp[2])};
      BTR_TRY(ln_backward(hs, a));
    }
    BTR_TRY(dgrad(rq, e, f, S(sc.df), S(sc.t_l2), e, S(sc.dh), sc.p_l2, 0, f));
    hipLaunchKernelGGL(relu_drop_bwd_kernel, dim3(cdiv((long long)rq * f / 4, 256)), dim3(256), 0,
                       hs, (long long)rq * f / 4, (float4 *)S(sc.dh), (const float4 *)h,
                       make_drop(d, 2).keep_inv);
    const int sk = pm_splitk_slices(rq, e, f);   // dX2 = dH W1: 288 columns, reduction over ff
    if (sk > 1)
      BTR_TRY(pm_gemm_nt_splitk(rq, e, f, S(sc.dh), f, S(sc.t_l1), f, S(sc.dx2f),
                                (long long)rq * e, sk, hs));
    else
      BTR_TRY(btr_pm_gemm_nt(rq, e, f, S(sc.dh), f, S(sc.t_l1), f, S(sc.dx2f), e, nullptr, nullptr,
                             nullptr, nullptr, stream));
    // ---- LayerNorm 2, cross-attention
    {
      LnBwd a{rq, e, S(sc.dres3), S(sc.dx2f), nullptr, sk, (long long)rq * e, at_f(saved, p.xh2), at_f(saved, p.rs2),
              d.ln_w[1], make_drop(d, 1), S(sc.dres2), S(sc.do2), S(sc.lnp[1])};
      BTR_TRY(ln_backward(hs, a));
    }
    BTR_TRY(dgrad(rq, e, e, S(sc.do2), S(sc.t_ca_out), e, S(sc.da2), sc.p_ca_out, 0, e));
    // (cross-attention: d q2 only -- its d k / d v go to the keys and to parameters, nothing the
    // layer's input gradient waits for: kDecoderBwdKey.  The row sums D live in their own
    // buffer, the self-attention below must not overwrite them before that part has run)
    BTR_TRY(attention_bwd_strided_parts(
        d.pq, d.pk, d.b, d.heads, hd, q2, e, (long long)d.pq * e, kv, kv + e, 2 * e,
        (long long)d.pk * 2 * e, a2, S(sc.da2), e, (long long)d.pq * e, at_f(saved, p.lse2),
        S(sc.dsum2), S(sc.dq2), e, (long long)d.pq * e, S(sc.dkv), S(sc.dkv) + e, 2 * e,
        (long long)d.pk * 2 * e, scale, d.dropout, attn_seed(d, 1), d.step, kAttnBwdQ, stream));
    BTR_TRY(dgrad(rq, e, e, S(sc.dq2), S(sc.t_ca_in), 3 * e, r_dqp1, sc.p_ca_in, 0, e));
    // ---- LayerNorm 1, self-attention
    {
      LnBwd a{rq, e, S(sc.dres2), r_dqp1, nullptr, 1, 0, at_f(saved, p.xh1), at_f(saved, p.rs1),
              d.ln_w[0], make_drop(d, 0), r_dres1, S(sc.do1), S(sc.lnp[0])};
      BTR_TRY(ln_backward(hs, a));
    }
    BTR_TRY(dgrad(rq, e, e, S(sc.do1), S(sc.t_sa_out), e, S(sc.da1), sc.p_sa_out, 0, e));
    BTR_TRY(attention_bwd_strided(
        d.pq, d.pq, d.b, d.heads, hd, qkv, 3 * e, (long long)d.pq * 3 * e, qkv + e, qkv + 2 * e,
        3 * e, (long long)d.pq * 3 * e, a1, S(sc.da1), e, (long long)d.pq * e, at_f(saved, p.lse1),
        S(sc.dsum), S(sc.dqkv), 3 * e, (long long)d.pq * 3 * e, S(sc.dqkv) + e,
        S(sc.dqkv) + 2 * e, 3 * e, (long long)d.pq * 3 * e, scale, d.dropout, attn_seed(d, 0),
        d.step, stream));
    BTR_TRY(dgrad(rq, 3 * e, e, S(sc.dqkv), S(sc.t_sa_in), 3 * e, r_dqp0, sc.p_sa_in, 0, e));
    // ---- gradients of the module inputs, (B, E, P)
    if (dx_bcp) BTR_TRY(rows_to_bcp(hs, d.b, d.pq, e, r_dres1, r_dqp0, nullptr, dx_bcp));
    if (dqpos_bcp && qpos_cl)
      BTR_TRY(rows_to_bcp(hs, d.b, d.pq, e, r_dqp0, r_dqp1, nullptr, dqpos_bcp));
  }

  if (parts & kDecoderBwdKey) {
    // the cross-attention's d k / d v (from the D the chain's d q kernel left), then the key
    // rows' input gradient
    BTR_TRY(attention_bwd_strided_parts(
        d.pq, d.pk, d.b, d.heads, hd, q2, e, (long long)d.pq * e, kv, kv + e, 2 * e,
        (long long)d.pk * 2 * e, a2, S(sc.da2), e, (long long)d.pq * e, at_f(saved, p.lse2),
        S(sc.dsum2), S(sc.dq2), e, (long long)d.pq * e, S(sc.dkv), S(sc.dkv) + e, 2 * e,
        (long long)d.pk * 2 * e, scale, d.dropout, attn_seed(d, 1), d.step, kAttnBwdKV, stream));
    if (dkey_bcp || (out && out->dkp))
      BTR_TRY(dgrad(rk, 2 * e, e, S(sc.dkv), S(sc.t_ca_in) + e, 3 * e, r_dkp, sc.p_ca_in, e, e));
    if (dkey_bcp) BTR_TRY(rows_to_bcp(hs, d.b, d.pk, e, r_dkp, nullptr, nullptr, dkey_bcp));
  }
  if (parts & kDecoderBwdRest) {
    {
      reduce_batch_begin();
      struct Flush {
        hipStream_t s;
        bool open = true;
        ~Flush() { if (open) reduce_batch_flush(s); }
      } flush{hs};
      auto wgrad = [&](int i, const float *g, const float *x, float *dw) {
        return btr_sa_gemm_tn(lin[i].rows, lin[i].n, lin[i].k, g, lin[i].n, x, lin[i].k, nullptr,
                              nullptr, S(sc.pw[i]), dw, stream);
      };
      BTR_TRY(wgrad(6, S(sc.df), h, grads + p.g_lin2_w));
      BTR_TRY(wgrad(5, S(sc.dh), x2, grads + p.g_lin1_w));
      BTR_TRY(wgrad(4, S(sc.do2), a2, grads + p.g_ca_out_w));
      BTR_TRY(wgrad(2, S(sc.dq2), qp1, grads + p.g_ca_in_w));
      BTR_TRY(wgrad(3, S(sc.dkv), kp, grads + p.g_ca_in_w + (size_t)e * e));
      BTR_TRY(wgrad(1, S(sc.do1), a1, grads + p.g_sa_out_w));
      BTR_TRY(wgrad(0, S(sc.dqkv), qp0, grads + p.g_sa_in_w));
      reduce_batch_flush(hs);
      flush.open = false;
    }
    {  // bias gradients (column sums of the 7 dY) and the LayerNorm parameter gradients
      ColsumArgs c{};
      const float *g[7] = {S(sc.dqkv), S(sc.do1), S(sc.dq2), S(sc.dkv), S(sc.do2), S(sc.dh),
                           S(sc.df)};
      int blocks = 0;
      for (int i = 0; i < 7; ++i) {
        c.g[i] = g[i]; c.part[i] = S(sc.cs[i]); c.rows[i] = lin[i].rows; c.c[i] = lin[i].n;
        c.ld[i] = lin[i].n;
        c.first[i] = blocks;
        blocks += cdiv(lin[i].n, 64) * cdiv(lin[i].rows, kCsRows);
      }
      c.first[7] = blocks;
      c.count = 7;
      hipLaunchKernelGGL(colsum_multi_kernel, dim3(blocks), dim3(256), 0, hs, c);
      FinalArgs fa{};
      float *outp[7] = {grads + p.g_sa_in_b, grads + p.g_sa_out_b, grads + p.g_ca_in_b,
                        grads + p.g_ca_in_b + e, grads + p.g_ca_out_b, grads + p.g_lin1_b,
                        grads + p.g_lin2_b};
      blocks = 0;
      for (int i = 0; i < 7; ++i) {
        fa.part[i] = S(sc.cs[i]); fa.out[i] = outp[i]; fa.chunks[i] = cdiv(lin[i].rows, kCsRows);
        fa.c[i] = lin[i].n;
        fa.first[i] = blocks;
        blocks += cdiv(lin[i].n, 64);
      }
      for (int i = 0; i < 3; ++i) {
        fa.part[7 + i] = S(sc.lnp[i]); fa.out[7 + i] = grads + p.g_ln[i];
        fa.chunks[7 + i] = ln_blocks(rq); fa.c[7 + i] = 2 * e;
        fa.first[7 + i] = blocks;
        blocks += cdiv(2 * e, 64);
      }
      fa.first[10] = blocks;
      fa.count = 10;
      hipLaunchKernelGGL(colsum_final_multi_kernel, dim3(blocks), dim3(256), 0, hs, fa);
    }
  }
  return check_launch("decoder_layer_backward");
}

}  // namespace btr
