// interpolate.hip -- three_nn, three_interpolate(+grad) for gfx950.
// Replaces src/interpolate_gpu.cu:14-159 of the reference.  Built with -ffp-contract=off:
// the 3-NN distances and the blend `p1*w1 + p2*w2 + p3*w3` round exactly as written there.
#include <algorithm>

#include <cstdlib>

#include "internal.hpp"

namespace btr {

// Lanes = queries; the known points are wave-uniform (scalar-cache broadcast).  The reference
// keeps best1..3 as doubles initialised to 1e40 and compares the promoted f32 distance
// (interpolate_gpu.cu:32,39-54); with finite or infinite f32 distances that is the same
// ordering as f32 comparisons against +inf, and (float)1e40 == +inf is what it stores when
// fewer than three known points exist.
// the inverse-distance blend weights of the feature-propagation module
// (pointnet2_modules.py:493-496: 1 / (sqrt(d2) + 1e-8), normalised over the three neighbours)
__device__ __forceinline__ void nn_weights(float d1, float d2, float d3, float *w) {
  const float r1 = 1.0f / (sqrtf(d1) + 1e-8f), r2 = 1.0f / (sqrtf(d2) + 1e-8f),
              r3 = 1.0f / (sqrtf(d3) + 1e-8f);
  const float norm = (r1 + r2) + r3;
  w[0] = r1 / norm;
  w[1] = r2 / norm;
  w[2] = r3 / norm;
}

__global__ __launch_bounds__(64) void three_nn_kernel(int n, int m,
                                                      const float *__restrict__ unknown,
                                                      const float *__restrict__ known,
                                                      float *__restrict__ dist2,
                                                      int *__restrict__ idx,
                                                      float *__restrict__ weight = nullptr) {
  const int bi = blockIdx.y;
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= n) return;
  known += (size_t)bi * m * 3;
  const float *u = unknown + ((size_t)bi * n + j) * 3;
  const float ux = u[0], uy = u[1], uz = u[2];
  const float inf = __builtin_huge_valf();
  float best1 = inf, best2 = inf, best3 = inf;
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int k = 0; k < m; ++k) {
    const float x = known[k * 3 + 0], y = known[k * 3 + 1], z = known[k * 3 + 2];
    const float d = sq3(ux - x, uy - y, uz - z);
    // branch-free form of the reference's if / else-if / else-if insertion (:39-54):
    // strict `<` against the current 1st/2nd/3rd keeps the earliest index on ties
    const bool c1 = d < best1, c2 = d < best2, c3 = d < best3;
    best3 = c2 ? best2 : (c3 ? d : best3);
    besti3 = c2 ? besti2 : (c3 ? k : besti3);
    best2 = c1 ? best1 : (c2 ? d : best2);
    besti2 = c1 ? besti1 : (c2 ? k : besti2);
    best1 = c1 ? d : best1;
    besti1 = c1 ? k : besti1;
  }
  float *d2 = dist2 + ((size_t)bi * n + j) * 3;
  int *id = idx + ((size_t)bi * n + j) * 3;
  d2[0] = best1; d2[1] = best2; d2[2] = best3;
  id[0] = besti1; id[1] = besti2; id[2] = besti3;
  if (weight) nn_weights(best1, best2, best3, weight + ((size_t)bi * n + j) * 3);
}

// Same result with 8 lanes per query: each lane scans a contiguous slice of the known points
// (ascending indices, strict `<`), then the sorted triples are merged pairwise under the order
// the sequential scan realises -- smaller distance first, smaller index first on equal
// distances.  The one-lane-per-query kernel leaves the chip nearly empty at the FP-module sizes
// (1024 queries x 512 known points per scene: 49 us -> ~15 us).
struct Top3 {
  float d1, d2, d3;
  int i1, i2, i3;
};
__device__ __forceinline__ bool nn_less(float d, int k, float bd, int bk) {
  return d < bd || (d == bd && k < bk);
}
__device__ __forceinline__ void nn_insert(Top3 &t, float d, int k) {
  const bool c1 = nn_less(d, k, t.d1, t.i1), c2 = nn_less(d, k, t.d2, t.i2),
             c3 = nn_less(d, k, t.d3, t.i3);
  t.d3 = c2 ? t.d2 : (c3 ? d : t.d3);
  t.i3 = c2 ? t.i2 : (c3 ? k : t.i3);
  t.d2 = c1 ? t.d1 : (c2 ? d : t.d2);
  t.i2 = c1 ? t.i1 : (c2 ? k : t.i2);
  t.d1 = c1 ? d : t.d1;
  t.i1 = c1 ? k : t.i1;
}

__global__ __launch_bounds__(64) void three_nn_split_kernel(int n, int m,
                                                            const float *__restrict__ unknown,
                                                            const float *__restrict__ known,
                                                            float *__restrict__ dist2,
                                                            int *__restrict__ idx,
                                                            float *__restrict__ weight = nullptr) {
  const int bi = blockIdx.y;
  const int part = threadIdx.x & 7;
  const int j = blockIdx.x * 8 + (threadIdx.x >> 3);
  const int jj = min(j, n - 1);
  known += (size_t)bi * m * 3;
  const float *u = unknown + ((size_t)bi * n + jj) * 3;
  const float ux = u[0], uy = u[1], uz = u[2];
  const float inf = __builtin_huge_valf();
  Top3 t{inf, inf, inf, 0, 0, 0};
  const int per = (m + 7) / 8;
  const int k1 = min(m, (part + 1) * per);
  for (int k = part * per; k < k1; ++k) {
    const float x = known[k * 3 + 0], y = known[k * 3 + 1], z = known[k * 3 + 2];
    const float d = sq3(ux - x, uy - y, uz - z);
    const bool c1 = d < t.d1, c2 = d < t.d2, c3 = d < t.d3;  // indices ascend inside a slice
    t.d3 = c2 ? t.d2 : (c3 ? d : t.d3);
    t.i3 = c2 ? t.i2 : (c3 ? k : t.i3);
    t.d2 = c1 ? t.d1 : (c2 ? d : t.d2);
    t.i2 = c1 ? t.i1 : (c2 ? k : t.i2);
    t.d1 = c1 ? d : t.d1;
    t.i1 = c1 ? k : t.i1;
  }
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) {
    const float od1 = __shfl_xor(t.d1, off), od2 = __shfl_xor(t.d2, off),
                od3 = __shfl_xor(t.d3, off);
    const int oi1 = __shfl_xor(t.i1, off), oi2 = __shfl_xor(t.i2, off),
              oi3 = __shfl_xor(t.i3, off);
    // an empty slot is (inf, 0): never inserted (inf < inf is false, and 0 < 0 is false)
    if (od1 < inf) nn_insert(t, od1, oi1);
    if (od2 < inf) nn_insert(t, od2, oi2);
    if (od3 < inf) nn_insert(t, od3, oi3);
  }
  if (part == 0 && j < n) {
    float *d2 = dist2 + ((size_t)bi * n + j) * 3;
    int *id = idx + ((size_t)bi * n + j) * 3;
    d2[0] = t.d1; d2[1] = t.d2; d2[2] = t.d3;
    id[0] = t.i1; id[1] = t.i2; id[2] = t.i3;
    if (weight) nn_weights(t.d1, t.d2, t.d3, weight + ((size_t)bi * n + j) * 3);
  }
}

constexpr int kInterpCh = 8;  // channels per thread

__global__ __launch_bounds__(256) void three_interpolate_kernel(
    int c, int m, int n, const float *__restrict__ points, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ out) {
  const int bi = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int l0 = blockIdx.y * kInterpCh;
  const float *w = weight + ((size_t)bi * n + j) * 3;
  const int *id = idx + ((size_t)bi * n + j) * 3;
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  const int i1 = id[0], i2 = id[1], i3 = id[2];
  const int lc = min(kInterpCh, c - l0);
  for (int l = 0; l < lc; ++l) {
    const float *p = points + ((size_t)bi * c + l0 + l) * m;
    out[((size_t)bi * c + l0 + l) * n + j] = dot3(p[i1], w1, p[i2], w2, p[i3], w3);
  }
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(
    int c, int n, int m, const float *__restrict__ grad_out, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points) {
  const int bi = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int l0 = blockIdx.y * kInterpCh;
  const float *w = weight + ((size_t)bi * n + j) * 3;
  const int *id = idx + ((size_t)bi * n + j) * 3;
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  const int i1 = id[0], i2 = id[1], i3 = id[2];
  const int lc = min(kInterpCh, c - l0);
  for (int l = 0; l < lc; ++l) {
    const float g = grad_out[((size_t)bi * c + l0 + l) * n + j];
    float *gp = grad_points + ((size_t)bi * c + l0 + l) * m;
    atomicAdd(gp + i1, g * w1);
    atomicAdd(gp + i2, g * w2);
    atomicAdd(gp + i3, g * w3);
  }
}

// ---- three_interpolate_grad without float atomics: invert idx (b, n, 3) into per-known-point
// lists (integer atomics only), then every (b, channel, known point) sums its own list.
// 8 x 256 x 1024 -> 512: 115 us (3 f32 atomics per element, heavy contention: every known
// point receives ~6 contributions per channel) -> ~25 us.
__global__ __launch_bounds__(256) void ti_count_kernel(int n3, int m, const int *__restrict__ idx,
                                                       int *__restrict__ cnt) {
  const int bi = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < n3) atomicAdd(&cnt[(size_t)bi * (m + 1) + idx[(size_t)bi * n3 + e]], 1);
}

// exclusive scan of cnt[b][0..m) in place (off[b][m] = total); cursor[b][j] = off[b][j]
__global__ __launch_bounds__(256) void ti_scan_kernel(int m, int *__restrict__ off,
                                                      int *__restrict__ cursor) {
  __shared__ int wsum[4];
  __shared__ int carry_s;
  const int bi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int *o = off + (size_t)bi * (m + 1);
  int *cur = cursor + (size_t)bi * m;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < m; base += 256) {
    const int j = base + tid;
    const int v = j < m ? o[j] : 0;
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
    if (j < m) {
      o[j] = excl;
      cur[j] = excl;
    }
    __syncthreads();
    if (tid == 255) carry_s = pre + incl;
    __syncthreads();
  }
  if (tid == 0) o[m] = carry_s;
}

__global__ __launch_bounds__(256) void ti_fill_kernel(int n3, int m, const int *__restrict__ idx,
                                                      int *__restrict__ cursor,
                                                      int *__restrict__ refs) {
  const int bi = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n3) return;
  const int pos = atomicAdd(&cursor[(size_t)bi * m + idx[(size_t)bi * n3 + e]], 1);
  refs[(size_t)bi * n3 + pos] = e;
}

constexpr int kTiCpt = 8;  // channels per thread in the list reduction
__global__ __launch_bounds__(256) void ti_reduce_kernel(
    int c, int n, int m, const float *__restrict__ grad_out, const float *__restrict__ weight,
    const int *__restrict__ off, const int *__restrict__ refs, float *__restrict__ grad_points,
    long long go_bs) {
  const int bi = blockIdx.z;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int c0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * kTiCpt;
  if (j >= m || c0 >= c) return;
  const int beg = off[(size_t)bi * (m + 1) + j], end = off[(size_t)bi * (m + 1) + j + 1];
  const int *r = refs + (size_t)bi * n * 3;
  const float *w = weight + (size_t)bi * n * 3;
  float acc[kTiCpt];
#pragma unroll
  for (int l = 0; l < kTiCpt; ++l) acc[l] = 0.f;
  for (int e = beg; e < end; ++e) {
    const int ref = r[e];
    const float wt = w[ref];
    const float *g = grad_out + (size_t)bi * go_bs + (size_t)c0 * n + ref / 3;
#pragma unroll
    for (int l = 0; l < kTiCpt; ++l)
      if (c0 + l < c) acc[l] += g[(size_t)l * n] * wt;
  }
#pragma unroll
  for (int l = 0; l < kTiCpt; ++l)
    if (c0 + l < c) grad_points[((size_t)bi * c + c0 + l) * m + j] = acc[l];
}

// sa_mlp.hip: count + scan + fill of an inverted index list by one workgroup per batch element
bool csr_small_supported(int n_bins);
void csr_small_launch(int b, long long entries, int n_bins, const int *idx, int *off, int *refs,
                      hipStream_t st);

size_t ti_grad_workspace_bytes(int b, int n, int m) {
  if (b <= 0 || n <= 0 || m <= 0) return 0;
  return sizeof(int) * ((size_t)b * (m + 1) + (size_t)b * m + (size_t)b * n * 3);
}

// grad_points (b, c, m) = scatter-add of grad_out (b, c, n) [batch stride go_bstride floats]
// through the 3-NN lists, without float atomics (see ti_reduce_kernel)
int ti_grad_lists(int b, int c, int n, int m, const float *grad_out, long long go_bstride,
                  const int *idx, const float *weight, float *grad_points, void *workspace,
                  size_t workspace_bytes, hipStream_t st, int mode) {
  BTR_REQUIRE(idx && workspace && (mode == kScatterBuild || (grad_out && weight && grad_points)) &&
                  workspace_bytes >= ti_grad_workspace_bytes(b, n, m),
              "three_interpolate_grad: null pointer or workspace too small");
  BTR_REQUIRE(b < 65536 && (long long)n * 3 < 0x7fffffffLL, "three_interpolate_grad: too large");
  const int n3 = n * 3;
  const size_t off_b = sizeof(int) * (size_t)b * (m + 1), cur_b = sizeof(int) * (size_t)b * m;
  char *ws = (char *)workspace;
  int *off = (int *)ws, *cursor = (int *)(ws + off_b), *refs = (int *)(ws + off_b + cur_b);
  if (mode != kScatterReduce) {
    if (csr_small_supported(m)) {
      csr_small_launch(b, n3, m, idx, off, refs, st);
    } else {
      (void)hipMemsetAsync(off, 0, off_b, st);
      hipLaunchKernelGGL(ti_count_kernel, dim3(cdiv(n3, 256), b), dim3(256), 0, st, n3, m, idx,
                         off);
      hipLaunchKernelGGL(ti_scan_kernel, dim3(b), dim3(256), 0, st, m, off, cursor);
      hipLaunchKernelGGL(ti_fill_kernel, dim3(cdiv(n3, 256), b), dim3(256), 0, st, n3, m, idx,
                         cursor, refs);
    }
  }
  if (mode != kScatterBuild)
    hipLaunchKernelGGL(ti_reduce_kernel, dim3(cdiv(m, 64), cdiv(c, 4 * kTiCpt), b), dim3(256), 0,
                       st, c, n, m, grad_out, weight, off, refs, grad_points, go_bstride);
  return check_launch("three_interpolate_grad(lists)");
}

}  // namespace btr

using namespace btr;

extern "C" {

static int three_nn_launch(int b, int n, int m, const float *unknown, const float *known,
                           float *dist2, int *idx, float *weight, btr_stream_t stream) {
  if (b <= 0 || n <= 0) return BTR_OK;
  BTR_REQUIRE(unknown && dist2 && idx && (m <= 0 || known), "three_nn: null pointer");
  BTR_REQUIRE(b < 65536, "three_nn: batch too large");
  if (m >= 64)
    hipLaunchKernelGGL(three_nn_split_kernel, dim3(cdiv(n, 8), b), dim3(64), 0,
                       as_stream(stream), n, m, unknown, known, dist2, idx, weight);
  else
    hipLaunchKernelGGL(three_nn_kernel, dim3(cdiv(n, 64), b), dim3(64), 0, as_stream(stream), n,
                       std::max(m, 0), unknown, known, dist2, idx, weight);
  return check_launch("three_nn");
}

int btr_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                 int *idx, btr_stream_t stream) {
  return three_nn_launch(b, n, m, unknown, known, dist2, idx, nullptr, stream);
}

// three_nn + the feature-propagation module's blend weights in the same launch:
// weight (b, n, 3) = normalised 1 / (sqrt(dist2) + 1e-8)  (pointnet2_modules.py:492-496)
int btr_three_nn_weights(int b, int n, int m, const float *unknown, const float *known,
                         float *dist2, int *idx, float *weight, btr_stream_t stream) {
  BTR_REQUIRE(weight || b <= 0 || n <= 0, "three_nn_weights: null pointer");
  return three_nn_launch(b, n, m, unknown, known, dist2, idx, weight, stream);
}

int btr_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                          const float *weight, float *out, btr_stream_t stream) {
  if (b <= 0 || c <= 0 || n <= 0) return BTR_OK;
  BTR_REQUIRE(points && idx && weight && out && m > 0, "three_interpolate: null pointer or m=%d",
              m);
  BTR_REQUIRE(b < 65536, "three_interpolate: batch too large");
  hipLaunchKernelGGL(three_interpolate_kernel, dim3(cdiv(n, 256), cdiv(c, kInterpCh), b),
                     dim3(256), 0, as_stream(stream), c, m, n, points, idx, weight, out);
  return check_launch("three_interpolate");
}

int btr_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                               const int *idx, const float *weight, float *grad_points,
                               btr_stream_t stream) {
  const long long nout = (long long)b * c * m;
  if (nout <= 0) return BTR_OK;
  BTR_REQUIRE(grad_points, "three_interpolate_grad: null output");
  // default: inverted lists (no float atomics); BTR_TI_GRAD=atomic keeps the reference's form
  static const bool atomic_path = getenv("BTR_TI_GRAD") && getenv("BTR_TI_GRAD")[0] == 'a';
  if (!atomic_path && n > 0 && (long long)n * 3 < 0x7fffffffLL) {
    hipStream_t st = as_stream(stream);
    const size_t wsb = ti_grad_workspace_bytes(b, n, m);
    char *ws = nullptr;
    hipError_t e = hipMallocAsync((void **)&ws, wsb, st);
    if (e != hipSuccess)
      return fail((int)e, "three_interpolate_grad workspace: %s", hipGetErrorString(e));
    const int rc = ti_grad_lists(b, c, n, m, grad_out, (long long)c * n, idx, weight, grad_points,
                                 ws, wsb, st);
    (void)hipFreeAsync(ws, st);
    return rc;
  }
  hipError_t e = hipMemsetAsync(grad_points, 0, sizeof(float) * nout, as_stream(stream));
  if (e != hipSuccess)
    return fail((int)e, "three_interpolate_grad memset: %s", hipGetErrorString(e));
  if (n <= 0) return BTR_OK;
  BTR_REQUIRE(grad_out && idx && weight, "three_interpolate_grad: null pointer");
  BTR_REQUIRE(b < 65536, "three_interpolate_grad: batch too large");
  hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(cdiv(n, 256), cdiv(c, kInterpCh), b),
                     dim3(256), 0, as_stream(stream), c, n, m, grad_out, idx, weight,
                     grad_points);
  return check_launch("three_interpolate_grad");
}

}  // extern "C"
