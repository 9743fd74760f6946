// interpolate.hip -- three_nn, three_interpolate(+grad) for gfx950.
// Replaces src/interpolate_gpu.cu:14-159 of the reference.  Built with -ffp-contract=off:
// the 3-NN distances and the blend `p1*w1 + p2*w2 + p3*w3` round exactly as written there.
#include <algorithm>

#include "common.hpp"

namespace btr {

// Lanes = queries; the known points are wave-uniform (scalar-cache broadcast).  The reference
// keeps best1..3 as doubles initialised to 1e40 and compares the promoted f32 distance
// (interpolate_gpu.cu:32,39-54); with finite or infinite f32 distances that is the same
// ordering as f32 comparisons against +inf, and (float)1e40 == +inf is what it stores when
// fewer than three known points exist.
__global__ __launch_bounds__(64) void three_nn_kernel(int n, int m,
                                                      const float *__restrict__ unknown,
                                                      const float *__restrict__ known,
                                                      float *__restrict__ dist2,
                                                      int *__restrict__ idx) {
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
    const float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
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
    out[((size_t)bi * c + l0 + l) * n + j] = p[i1] * w1 + p[i2] * w2 + p[i3] * w3;
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

}  // namespace btr

using namespace btr;

extern "C" {

int btr_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                 int *idx, btr_stream_t stream) {
  if (b <= 0 || n <= 0) return BTR_OK;
  BTR_REQUIRE(unknown && dist2 && idx && (m <= 0 || known), "three_nn: null pointer");
  BTR_REQUIRE(b < 65536, "three_nn: batch too large");
  hipLaunchKernelGGL(three_nn_kernel, dim3(cdiv(n, 64), b), dim3(64), 0, as_stream(stream), n,
                     std::max(m, 0), unknown, known, dist2, idx);
  return check_launch("three_nn");
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
