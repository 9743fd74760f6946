// eval_boxes.hip -- the box arithmetic of the evaluation path on gfx950:
//   * greedy NMS over axis-aligned 2-D / 3-D boxes      (utils/nms.py:42-156)
//   * "does the box contain at least `cap` points"       (models/ap_helper.py:21-30,116-127)
//   * IoU of two upright oriented 3-D boxes (8 corners)  (utils/box_util.py:13-128)
// The reference does all three per box in numpy / scipy on the host, in float64; the kernels
// keep float64 and the reference's operation order (the build uses -ffp-contract=off), so the
// NMS decisions `o > threshold` are taken on the same numbers.
#include <cstdint>

#include "common.hpp"

namespace btr {

// ----------------------------------------------------------------------------------- NMS
// One workgroup per scene.  Boxes are ranked by descending score (ties: the higher index first,
// what a stable ascending argsort read from the back gives; numpy's default sort leaves the
// order of equal scores unspecified), the K x K "a suppresses b" bits are built in LDS in
// ranked order, and one wave walks the ranks: a box that is still alive is picked and ORs its
// row into the removed set -- the same picks as the reference's `while I.size: pick I[-1];
// delete overlaps` loop.
template <int DIM>
__device__ __forceinline__ double nms_overlap(const double *__restrict__ a,
                                              const double *__restrict__ b, int old_type) {
  // a = the picked (higher-score) box i, b = the candidate j; [min..., max...]
  double inter = 1.0, area_a = 1.0, area_b = 1.0;
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    const double lo = fmax(a[d], b[d]);
    const double hi = fmin(a[DIM + d], b[DIM + d]);
    const double e = fmax(0.0, hi - lo);
    // (l*w)*h, (x2-x1)*(y2-y1)*(z2-z1): left to right like the numpy expressions
    inter = d == 0 ? e : inter * e;
    area_a = d == 0 ? (a[DIM] - a[0]) : area_a * (a[DIM + d] - a[d]);
    area_b = d == 0 ? (b[DIM] - b[0]) : area_b * (b[DIM + d] - b[d]);
  }
  return old_type ? inter / area_b : inter / (area_a + area_b - inter);
}

template <int DIM>
__global__ __launch_bounds__(256) void nms_kernel(int k, int words,
                                                  const double *__restrict__ boxes,
                                                  const double *__restrict__ score,
                                                  const int *__restrict__ cls,
                                                  const uint8_t *__restrict__ valid, double thr,
                                                  int old_type, uint8_t *__restrict__ pick) {
  extern __shared__ __attribute__((aligned(16))) unsigned nms_lds[];
  __shared__ int nvalid_s;
  int *order = (int *)nms_lds;                       // [k]
  unsigned *sup = nms_lds + ((k + 3) & ~3);          // [k][words]
  const int bi = blockIdx.x, tid = threadIdx.x;
  boxes += (size_t)bi * k * 2 * DIM;
  score += (size_t)bi * k;
  if (cls) cls += (size_t)bi * k;
  if (valid) valid += (size_t)bi * k;
  pick += (size_t)bi * k;
  if (tid == 0) nvalid_s = 0;
  __syncthreads();
  for (int i = tid; i < k; i += 256) {
    pick[i] = 0;
    if (valid && !valid[i]) continue;
    atomicAdd(&nvalid_s, 1);
    const double si = score[i];
    int r = 0;
    for (int j = 0; j < k; ++j) {
      if (valid && !valid[j]) continue;
      const double sj = score[j];
      r += (sj > si || (sj == si && j > i)) ? 1 : 0;
    }
    order[r] = i;
  }
  __syncthreads();
  const int nv = nvalid_s;
  for (int a = tid; a < nv; a += 256) {
    const int ia = order[a];
    double ba[2 * DIM];
#pragma unroll
    for (int d = 0; d < 2 * DIM; ++d) ba[d] = boxes[(size_t)ia * 2 * DIM + d];
    const int ca = cls ? cls[ia] : 0;
    for (int w = 0; w < words; ++w) {
      unsigned bits = 0u;
      for (int t = 0; t < 32; ++t) {
        const int b = w * 32 + t;
        if (b <= a || b >= nv) continue;
        const int ib = order[b];
        double o = nms_overlap<DIM>(ba, boxes + (size_t)ib * 2 * DIM, old_type);
        if (cls) o = o * (ca == cls[ib] ? 1.0 : 0.0);  // nms.py:142 `o * (cls1==cls2)`
        if (o > thr) bits |= 1u << t;
      }
      sup[(size_t)a * words + w] = bits;
    }
  }
  __syncthreads();
  if (tid < 64) {  // words <= 32: lane l keeps word l of the removed set
    unsigned rem = 0u;
    for (int a = 0; a < nv; ++a) {
      const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)rem, a >> 5);
      if ((w >> (a & 31)) & 1u) continue;
      if (tid == 0) pick[order[a]] = 1;
      if (tid < words) rem |= sup[(size_t)a * words + tid];
    }
  }
}

// ------------------------------------------------------------------- points inside a box
// One wave per box; lanes stride over the scene's points and stop once `cap` are inside.
// The point is taken to upright-camera coordinates (x, -z, y) (ap_helper.py:32-40), moved to
// the box frame (rotation about the camera y axis, box_util.py:183-190) and compared with the
// half extents (l, h, w)/2 of get_3d_box (box_util.py:211-227); the boundary is inside, like
// scipy's find_simplex(p) >= 0.
__global__ __launch_bounds__(256) void points_in_boxes_kernel(
    int n, int k, int pstride, int cap, const float *__restrict__ pts,
    const double *__restrict__ center, const double *__restrict__ size,
    const double *__restrict__ angle, int *__restrict__ count) {
  const int bi = blockIdx.y, lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= k) return;
  pts += (size_t)bi * n * pstride;
  const size_t bj = (size_t)bi * k + j;
  const double cx = center[bj * 3 + 0], cy = center[bj * 3 + 1], cz = center[bj * 3 + 2];
  const double hl = fabs(size[bj * 3 + 0]) * 0.5, hw = fabs(size[bj * 3 + 1]) * 0.5,
               hh = fabs(size[bj * 3 + 2]) * 0.5;
  const double c = cos(angle[bj]), s = sin(angle[bj]);
  int total = 0;
  for (int base = 0; base < n && total < cap; base += 64) {
    const int i = base + lane;
    bool in = false;
    if (i < n) {
      const double vx = (double)pts[(size_t)i * pstride + 0] - cx;
      const double vy = -(double)pts[(size_t)i * pstride + 2] - cy;
      const double vz = (double)pts[(size_t)i * pstride + 1] - cz;
      const double lx = c * vx - s * vz, lz = s * vx + c * vz;
      in = fabs(lx) <= hl && fabs(vy) <= hh && fabs(lz) <= hw;
    }
    total += __builtin_popcountll(__ballot(in));
  }
  if (lane == 0) count[bj] = total < cap ? total : cap;
}

// --------------------------------------------------------------------- oriented-box IoU
// box_util.py:98-128: bird's-eye rectangles (x, z) of corners 3,2,1,0, rectangle 1 clipped by
// rectangle 2 with Sutherland-Hodgman (:13-62, strict `inside`, the same line-intersection
// formula), overlap height from corners 0 and 4, volumes from three edge lengths.  The area of
// the clipped polygon is the shoelace sum (the reference takes scipy's ConvexHull volume of the
// same vertices; equal for the convex polygons the clipping yields).
struct P2 {
  double x, y;
};

__device__ __forceinline__ bool sh_inside(P2 p, P2 c1, P2 c2) {
  return (c2.x - c1.x) * (p.y - c1.y) > (c2.y - c1.y) * (p.x - c1.x);
}

__device__ __forceinline__ P2 sh_cross(P2 s, P2 e, P2 c1, P2 c2) {
  const double dcx = c1.x - c2.x, dcy = c1.y - c2.y;
  const double dpx = s.x - e.x, dpy = s.y - e.y;
  const double n1 = c1.x * c2.y - c1.y * c2.x;
  const double n2 = s.x * e.y - s.y * e.x;
  const double n3 = 1.0 / (dcx * dpy - dcy * dpx);
  return P2{(n1 * dpx - n2 * dcx) * n3, (n1 * dpy - n2 * dcy) * n3};
}

__device__ double shoelace(const P2 *p, int n) {
  // 0.5 * |sum x_i * y_{i-1} - y_i * x_{i-1}|   (box_util.py:64-66)
  double a = 0.0, b = 0.0;
  for (int i = 0; i < n; ++i) {
    const P2 q = p[(i + n - 1) % n];
    a += p[i].x * q.y;
    b += p[i].y * q.x;
  }
  return 0.5 * fabs(a - b);
}

__device__ double box3d_iou_corners(const double *__restrict__ c1,
                                    const double *__restrict__ c2) {
  P2 r1[4], r2[4];
  for (int i = 0; i < 4; ++i) {
    r1[i] = P2{c1[(3 - i) * 3 + 0], c1[(3 - i) * 3 + 2]};
    r2[i] = P2{c2[(3 - i) * 3 + 0], c2[(3 - i) * 3 + 2]};
  }
  P2 out[16], in[16];
  int no = 4;
  for (int i = 0; i < 4; ++i) out[i] = r1[i];
  P2 cp1 = r2[3];
  for (int e = 0; e < 4 && no > 0; ++e) {
    const P2 cp2 = r2[e];
    const int ni = no;
    for (int i = 0; i < ni; ++i) in[i] = out[i];
    no = 0;
    P2 s = in[ni - 1];
    for (int i = 0; i < ni; ++i) {
      const P2 q = in[i];
      if (sh_inside(q, cp1, cp2)) {
        if (!sh_inside(s, cp1, cp2)) out[no++] = sh_cross(s, q, cp1, cp2);
        out[no++] = q;
      } else if (sh_inside(s, cp1, cp2)) {
        out[no++] = sh_cross(s, q, cp1, cp2);
      }
      s = q;
    }
    cp1 = cp2;
  }
  const double inter_area = no > 0 ? shoelace(out, no) : 0.0;
  const double ymax = fmin(c1[1], c2[1]);
  const double ymin = fmax(c1[4 * 3 + 1], c2[4 * 3 + 1]);
  const double inter_vol = inter_area * fmax(0.0, ymax - ymin);
  auto edge = [](const double *c, int i, int j) {
    const double dx = c[i * 3] - c[j * 3], dy = c[i * 3 + 1] - c[j * 3 + 1],
                 dz = c[i * 3 + 2] - c[j * 3 + 2];
    return sqrt(dx * dx + dy * dy + dz * dz);
  };
  const double vol1 = edge(c1, 0, 1) * edge(c1, 1, 2) * edge(c1, 0, 4);
  const double vol2 = edge(c2, 0, 1) * edge(c2, 1, 2) * edge(c2, 0, 4);
  return inter_vol / (vol1 + vol2 - inter_vol);
}

// iou[s][p][g] for s < nscene: corners1 (S, P, 8, 3), corners2 (S, G, 8, 3)
__global__ __launch_bounds__(256) void box3d_iou_kernel(int p, int g,
                                                        const double *__restrict__ corners1,
                                                        const double *__restrict__ corners2,
                                                        double *__restrict__ iou) {
  const int si = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= p * g) return;
  const int pi = t / g, gi = t % g;
  iou[((size_t)si * p + pi) * g + gi] = box3d_iou_corners(
      corners1 + ((size_t)si * p + pi) * 24, corners2 + ((size_t)si * g + gi) * 24);
}

}  // namespace btr

using namespace btr;

extern "C" {

int btr_nms_boxes(int b, int k, int dim, const double *boxes, const double *score,
                  const int *cls, const unsigned char *valid, double threshold, int old_type,
                  unsigned char *pick, btr_stream_t stream) {
  if (b <= 0 || k <= 0) return BTR_OK;
  BTR_REQUIRE(boxes && score && pick, "nms_boxes: null pointer");
  BTR_REQUIRE(dim == 2 || dim == 3, "nms_boxes: dim %d is not 2 or 3", dim);
  BTR_REQUIRE(k <= 1024, "nms_boxes: %d boxes per scene (at most 1024)", k);
  const int words = (k + 31) / 32;
  const size_t lds = sizeof(unsigned) * (((size_t)k + 3) / 4 * 4 + (size_t)k * words);
  hipStream_t s = as_stream(stream);
  const void *fn = dim == 3 ? (const void *)nms_kernel<3> : (const void *)nms_kernel<2>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail((int)e, "nms_boxes attr: %s", hipGetErrorString(e));
  }
  if (dim == 3)
    hipLaunchKernelGGL(nms_kernel<3>, dim3(b), dim3(256), lds, s, k, words, boxes, score, cls,
                       valid, threshold, old_type, pick);
  else
    hipLaunchKernelGGL(nms_kernel<2>, dim3(b), dim3(256), lds, s, k, words, boxes, score, cls,
                       valid, threshold, old_type, pick);
  return check_launch("nms_boxes");
}

int btr_points_in_boxes(int b, int n, int k, int point_stride, int cap, const float *points,
                        const double *center, const double *size, const double *angle,
                        int *count, btr_stream_t stream) {
  if (b <= 0 || k <= 0) return BTR_OK;
  BTR_REQUIRE(points && center && size && angle && count, "points_in_boxes: null pointer");
  BTR_REQUIRE(n >= 0 && point_stride >= 3 && cap >= 1,
              "points_in_boxes: n=%d stride=%d cap=%d", n, point_stride, cap);
  hipLaunchKernelGGL(points_in_boxes_kernel, dim3(cdiv(k, 4), b), dim3(256), 0,
                     as_stream(stream), n, k, point_stride, cap, points, center, size, angle,
                     count);
  return check_launch("points_in_boxes");
}

int btr_box3d_iou(int nscene, int p, int g, const double *corners1, const double *corners2,
                  double *iou, btr_stream_t stream) {
  if (nscene <= 0 || p <= 0 || g <= 0) return BTR_OK;
  BTR_REQUIRE(corners1 && corners2 && iou, "box3d_iou: null pointer");
  BTR_REQUIRE(nscene <= 65535, "box3d_iou: %d scenes per call (at most 65535)", nscene);
  hipLaunchKernelGGL(box3d_iou_kernel, dim3(cdiv((long long)p * g, 256), nscene), dim3(256), 0,
                     as_stream(stream), p, g, corners1, corners2, iou);
  return check_launch("box3d_iou");
}

}  // extern "C"
