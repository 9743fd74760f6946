// ball_query_grid.hip -- ball query with a uniform-grid cull for large scenes on gfx950.
//
// Same result as the reference kernel (ball_query_gpu.cu:14-49: for every centre the FIRST
// nsample point indices, in ascending index order, with d2 < radius^2; short rows padded with
// the first hit, empty rows zero), but instead of testing all n points per centre:
//   1. the scene is counting-sorted into cells of edge >= radius (x fastest), so the 27 cells
//      around a centre are 9 contiguous runs of the sorted array and contain every point that
//      can pass the test;
//   2. one wave per centre tests the candidates of those runs (the exact f32 expression of
//      the reference, so the hit set is bit-identical) and sets bit `index` in a per-wave LDS
//      bitmap over [0, n);
//   3. the bitmap is scanned in index order: per-lane popcounts, a wave prefix sum, and the
//      lanes holding the first nsample set bits write them out -- "first nsample in index
//      order" without sorting the hits, independent of the order of points inside a cell.
// Distance tests drop from n per centre to a few hundred (8x40000 points, 2048 centres,
// r = 0.2: 6.55e8 -> ~3e6).
#include <algorithm>
#include <cmath>

#include "common.hpp"

namespace btr {

struct GridMeta {       // per scene, written by bq_grid_build_kernel
  float mnx, mny, mnz;  // grid origin
  float inv_cs;         // 1 / cell size
  int gx, gy, gz;       // cells per axis
  int ncell;
};

constexpr int kMaxCells = 1 << 18;

__device__ __forceinline__ int cell_coord(float v, float mn, float inv_cs, int g) {
  const int c = (int)floorf((v - mn) * inv_cs);
  return min(max(c, 0), g - 1);
}

// ------------------------------------------------------------------------- the grid build
// One workgroup of 1024 threads per scene does the whole build -- bounding box, cell histogram,
// exclusive scan, scatter -- with the histogram / cursors in LDS (the points stay in L2
// between the three passes): one launch of ~35 us instead of memset + four launches of ~95 us
// whose count / fill passes hammer global atomics.  The grid must fit the LDS table
// (kLdsCells cells = 144 KB): the cell edge grows past 1.001 r until it does (a 12 x 12 x 3 m
// scene at r = 0.2 ends up with 0.25 m cells), which only changes how many candidates a
// centre tests, never the result.  (The memset + four-launch build with global atomics it
// replaced was removed in round 6.)
constexpr int kLdsCells = 36864;

// Visits every point of a scene from a 1024-thread workgroup, four consecutive points (three
// 16-byte loads) per thread and trip: f(k, x, y, z).
template <typename F>
__device__ __forceinline__ void for_each_point4(const float *__restrict__ xyz, int n, int tid,
                                                F f) {
  const int n4 = n >> 2;
  const float4 *v = reinterpret_cast<const float4 *>(xyz);
#pragma unroll 2
  for (int q = tid; q < n4; q += 1024) {
    const float4 a = v[q * 3 + 0], b = v[q * 3 + 1], c = v[q * 3 + 2];
    f(q * 4 + 0, a.x, a.y, a.z);
    f(q * 4 + 1, a.w, b.x, b.y);
    f(q * 4 + 2, b.z, b.w, c.x);
    f(q * 4 + 3, c.y, c.z, c.w);
  }
  const int k = n4 * 4 + tid;
  if (k < n) f(k, xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2]);
}

__global__ __launch_bounds__(1024) void bq_grid_build_kernel(int n, float radius,
                                                             const float *__restrict__ xyz,
                                                             GridMeta *__restrict__ meta,
                                                             int *__restrict__ cell_off,
                                                             float4 *__restrict__ sorted) {
  extern __shared__ __attribute__((aligned(16))) int hist[];  // [kLdsCells]
  __shared__ float red[6][16];
  __shared__ int wsum[16];
  __shared__ GridMeta gm;
  const int bi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  xyz += (size_t)bi * n * 3;
  int *off = cell_off + (size_t)bi * (kMaxCells + 1);
  float4 *out = sorted + (size_t)bi * n;

  // ---- A: bounding box -> grid geometry (as bq_grid_bbox_kernel, LDS capacity as the limit)
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for_each_point4(xyz, n, tid, [&](int, float x, float y, float z) {
    mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
    mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
    mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
  });
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
    }
    if (lane == 0) {
      red[a][wave] = mn[a];
      red[3 + a][wave] = mx[a];
    }
  }
  __syncthreads();
  if (tid == 0) {
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
      lo[a] = red[a][0];
      hi[a] = red[3 + a][0];
      for (int w = 1; w < 16; ++w) {
        lo[a] = fminf(lo[a], red[a][w]);
        hi[a] = fmaxf(hi[a], red[3 + a][w]);
      }
    }
    float cs = radius * 1.001f;
    if (!(cs > 0.f)) cs = 1.f;
    int g[3];
    for (int it = 0; it < 96; ++it) {
      double cells = 1.0;
      for (int a = 0; a < 3; ++a) {
        const float e = fmaxf(hi[a] - lo[a], 0.f);
        const double q = floor((double)e / (double)cs) + 1.0;
        g[a] = q > 1.0e6 ? 1000000 : (int)q;
        cells *= (double)g[a];
      }
      if (cells <= (double)kLdsCells) break;
      cs *= 1.26f;
    }
    GridMeta m;
    m.mnx = lo[0]; m.mny = lo[1]; m.mnz = lo[2];
    m.inv_cs = 1.0f / cs;
    m.gx = g[0]; m.gy = g[1]; m.gz = g[2];
    m.ncell = g[0] * g[1] * g[2];
    gm = m;
    meta[bi] = m;
  }
  __syncthreads();
  const GridMeta m = gm;
  const int ncell = m.ncell;

  // ---- B: histogram in LDS
  for (int c = tid; c < ncell; c += 1024) hist[c] = 0;
  __syncthreads();
  for_each_point4(xyz, n, tid, [&](int, float x, float y, float z) {
    const int cx = cell_coord(x, m.mnx, m.inv_cs, m.gx);
    const int cy = cell_coord(y, m.mny, m.inv_cs, m.gy);
    const int cz = cell_coord(z, m.mnz, m.inv_cs, m.gz);
    atomicAdd(&hist[(cz * m.gy + cy) * m.gx + cx], 1);
  });
  __syncthreads();

  // ---- C: exclusive scan in place (a contiguous chunk per thread, then the chunk sums)
  const int chunk = (ncell + 1023) / 1024;
  const int c0 = tid * chunk, c1 = min(c0 + chunk, ncell);
  int sum = 0;
  for (int c = c0; c < c1; ++c) sum += hist[c];
  int incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int pre = incl - sum;
  for (int w = 0; w < wave; ++w) pre += wsum[w];
  for (int c = c0; c < c1; ++c) {
    const int v = hist[c];
    hist[c] = pre;
    pre += v;
  }
  __syncthreads();
  for (int c = tid; c < ncell; c += 1024) off[c] = hist[c];
  if (tid == 0) off[ncell] = n;
  __syncthreads();  // the offsets are copied out before the scatter turns them into cursors

  // ---- D: scatter (hist is now the per-cell cursor)
  for_each_point4(xyz, n, tid, [&](int k, float x, float y, float z) {
    const int cx = cell_coord(x, m.mnx, m.inv_cs, m.gx);
    const int cy = cell_coord(y, m.mny, m.inv_cs, m.gy);
    const int cz = cell_coord(z, m.mnz, m.inv_cs, m.gz);
    const int pos = atomicAdd(&hist[(cz * m.gy + cy) * m.gx + cx], 1);
    out[pos] = make_float4(x, y, z, __int_as_float(k));
  });
}

// One wave per centre (4 per workgroup, grid-stride).  Dynamic LDS: 4 bitmaps of `words`
// 32-bit words (zero on entry and restored to zero after every centre).
__global__ __launch_bounds__(256) void bq_grid_query_kernel(
    int n, int m, int nsample, int words, float radius2, const float *__restrict__ new_xyz,
    const GridMeta *__restrict__ meta, const int *__restrict__ cell_off,
    const float4 *__restrict__ sorted, int *__restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) unsigned bitmaps[];
  const int bi = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned *bm = bitmaps + (size_t)wave * words;
  for (int w = lane; w < words; w += 64) bm[w] = 0u;

  const GridMeta g = meta[bi];
  const int *off = cell_off + (size_t)bi * (kMaxCells + 1);
  const float4 *pts = sorted + (size_t)bi * n;
  const int wpl = (words + 63) / 64;  // bitmap words per lane (contiguous range per lane)

  for (int j = blockIdx.x * 4 + wave; j < m; j += gridDim.x * 4) {
    const float *c = new_xyz + ((size_t)bi * m + j) * 3;
    const float new_x = c[0], new_y = c[1], new_z = c[2];
    // the centre's cell, NOT clamped: a centre outside the grid only sees the cells in range
    const int cx = (int)floorf((new_x - g.mnx) * g.inv_cs);
    const int cy = (int)floorf((new_y - g.mny) * g.inv_cs);
    const int cz = (int)floorf((new_z - g.mnz) * g.inv_cs);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.gx - 1);
    // 9 runs (dy, dz) of contiguous cells x0..x1; lane r < 9 owns run r
    int rbeg = 0, rlen = 0;
    if (lane < 9 && x0 <= x1) {
      const int yy = cy + (lane % 3) - 1, zz = cz + (lane / 3) - 1;
      if (yy >= 0 && yy < g.gy && zz >= 0 && zz < g.gz) {
        const int row = (zz * g.gy + yy) * g.gx;
        rbeg = off[row + x0];
        rlen = off[row + x1 + 1] - rbeg;
      }
    }
    // candidates of all runs, 64 at a time
    for (int r = 0; r < 9; ++r) {
      const int b = __builtin_amdgcn_readlane(rbeg, r);
      const int len = __builtin_amdgcn_readlane(rlen, r);
      for (int t = lane; t < len; t += 64) {
        const float4 p = pts[b + t];
        const float d2 = sq3(new_x - p.x, new_y - p.y, new_z - p.z);
        if (d2 < radius2) {
          const unsigned k = (unsigned)__float_as_int(p.w);
          atomicOr(&bm[k >> 5], 1u << (k & 31));
        }
      }
    }
    // first nsample set bits in index order
    int cnt = 0;
    const int w0 = lane * wpl;
    for (int i = 0; i < wpl; ++i)
      if (w0 + i < words) cnt += __builtin_popcount(bm[w0 + i]);
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    const int total = __builtin_amdgcn_readlane(incl, 63);
    int pos = incl - cnt;
    int *row = idx + ((size_t)bi * m + j) * nsample;
    int first = 0x7fffffff;
    for (int i = 0; i < wpl; ++i) {
      if (w0 + i >= words) break;
      unsigned v = bm[w0 + i];
      if (v) bm[w0 + i] = 0u;  // restore the bitmap for the next centre
      while (v) {
        const int k = ((w0 + i) << 5) + __builtin_ctz(v);
        v &= v - 1;
        first = min(first, k);
        if (pos < nsample) row[pos] = k;
        ++pos;
      }
    }
    // padding: the smallest hit (= the first in index order) fills the rest; no hit -> zeros
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) first = min(first, __shfl_xor(first, o));
    const int fill = total == 0 ? 0 : first;
    for (int l = min(total, nsample) + lane; l < nsample; l += 64) row[l] = fill;
  }
}

struct GridPlan {
  size_t meta_b, cell_b, sorted_b;
  int words;
};

static GridPlan grid_plan(int b, int n) {
  GridPlan p;
  p.meta_b = (sizeof(GridMeta) * b + 255) / 256 * 256;
  p.cell_b = sizeof(int) * (size_t)b * (kMaxCells + 1);
  p.sorted_b = sizeof(float4) * (size_t)b * n;
  p.words = (n + 31) / 32;
  return p;
}

bool bq_grid_supported(int n, int m, int nsample) {
  // bitmap of 4 waves must fit LDS with room to spare; small problems stay brute force
  return n >= 8192 && (size_t)((n + 31) / 32) * 4 * 4 <= 96 * 1024 && m > 0 && nsample > 0;
}

size_t bq_grid_workspace_bytes(int b, int n) {
  const GridPlan p = grid_plan(b, n);
  return p.meta_b + p.cell_b + p.sorted_b;
}

int bq_grid_launch(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                   const float *xyz, int *idx, void *ws, size_t ws_bytes, hipStream_t s) {
  const GridPlan p = grid_plan(b, n);
  BTR_REQUIRE(ws && ws_bytes >= p.meta_b + p.cell_b + p.sorted_b,
              "ball_query: workspace too small for the grid path");
  char *base = (char *)ws;
  GridMeta *meta = (GridMeta *)base;
  int *cell_off = (int *)(base + p.meta_b);
  float4 *sorted = (float4 *)(base + p.meta_b + p.cell_b);
  hipError_t e = hipSuccess;
  {
    static bool attr_set = false;
    if (!attr_set) {
      e = hipFuncSetAttribute((const void *)bq_grid_build_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(int) * kLdsCells));
      if (e != hipSuccess)
        return fail((int)e, "ball_query(grid) build attr: %s", hipGetErrorString(e));
      attr_set = true;
    }
    hipLaunchKernelGGL(bq_grid_build_kernel, dim3(b), dim3(1024), sizeof(int) * kLdsCells, s, n,
                       radius, xyz, meta, cell_off, sorted);
  }
  const size_t lds = sizeof(unsigned) * (size_t)p.words * 4;
  static size_t lds_set = 0;
  if (lds > lds_set && lds > 48 * 1024) {
    e = hipFuncSetAttribute((const void *)bq_grid_query_kernel,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail((int)e, "ball_query(grid) attr: %s", hipGetErrorString(e));
    lds_set = lds;
  }
  const float radius2 = radius * radius;  // ball_query_gpu.cu:27
  const int gq = std::min(cdiv(m, 4), 2048);
  hipLaunchKernelGGL(bq_grid_query_kernel, dim3(gq, b), dim3(256), lds, s, n, m, nsample,
                     p.words, radius2, new_xyz, meta, cell_off, sorted, idx);
  return check_launch("ball_query(grid)");
}

}  // namespace btr
