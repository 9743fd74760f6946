// group_points.hip -- grouped gather (B,C,N)[idx (B,M,S)] -> (B,C,M,S) and its scatter-add
// backward for gfx950.  Replaces src/group_points_gpu.cu:13-80 of the reference.
//
// Lanes run along the flattened (j,k) axis: the idx read and the output write are coalesced
// and each thread reuses its index for CH channels; the gather itself is served by L2.
#include <algorithm>

#include "common.hpp"

namespace btr {

constexpr int kGroupCh = 8;  // channels per thread

__global__ __launch_bounds__(256) void group_points_kernel(int c, int n, int ms,
                                                           const float *__restrict__ points,
                                                           const int *__restrict__ idx,
                                                           float *__restrict__ out) {
  const int bi = blockIdx.z;
  const int jk = blockIdx.x * 256 + threadIdx.x;
  if (jk >= ms) return;
  const int l0 = blockIdx.y * kGroupCh;
  const int ii = idx[(size_t)bi * ms + jk];
  const float *p = points + ((size_t)bi * c + l0) * n + ii;
  float *o = out + ((size_t)bi * c + l0) * ms + jk;
  const int lc = min(kGroupCh, c - l0);
  float v[kGroupCh];
#pragma unroll
  for (int l = 0; l < kGroupCh; ++l) v[l] = l < lc ? p[(size_t)l * n] : 0.f;
#pragma unroll
  for (int l = 0; l < kGroupCh; ++l)
    if (l < lc) o[(size_t)l * ms] = v[l];
}

__global__ __launch_bounds__(256) void group_points_grad_kernel(
    int c, int n, int ms, const float *__restrict__ grad_out, const int *__restrict__ idx,
    float *__restrict__ grad_points) {
  const int bi = blockIdx.z;
  const int jk = blockIdx.x * 256 + threadIdx.x;
  if (jk >= ms) return;
  const int l0 = blockIdx.y * kGroupCh;
  const int ii = idx[(size_t)bi * ms + jk];
  float *gp = grad_points + ((size_t)bi * c + l0) * n + ii;
  const float *go = grad_out + ((size_t)bi * c + l0) * ms + jk;
  const int lc = min(kGroupCh, c - l0);
  float v[kGroupCh];
#pragma unroll
  for (int l = 0; l < kGroupCh; ++l) v[l] = l < lc ? go[(size_t)l * ms] : 0.f;
#pragma unroll
  for (int l = 0; l < kGroupCh; ++l)
    if (l < lc) atomicAdd(gp + (size_t)l * n, v[l]);
}

}  // namespace btr

using namespace btr;

extern "C" {

int btr_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                     const int *idx, float *out, btr_stream_t stream) {
  const long long ms = (long long)npoints * nsample;
  if (b <= 0 || c <= 0 || ms <= 0) return BTR_OK;
  BTR_REQUIRE(points && idx && out && n > 0, "group_points: null pointer or n=%d", n);
  BTR_REQUIRE(ms < 0x7fffffffLL && b < 65536, "group_points: shape too large");
  hipLaunchKernelGGL(group_points_kernel, dim3(cdiv(ms, 256), cdiv(c, kGroupCh), b), dim3(256),
                     0, as_stream(stream), c, n, (int)ms, points, idx, out);
  return check_launch("group_points");
}

int btr_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                          const int *idx, float *grad_points, btr_stream_t stream) {
  const long long nout = (long long)b * c * n;
  if (nout <= 0) return BTR_OK;
  BTR_REQUIRE(grad_points, "group_points_grad: null output");
  hipError_t e = hipMemsetAsync(grad_points, 0, sizeof(float) * nout, as_stream(stream));
  if (e != hipSuccess) return fail((int)e, "group_points_grad memset: %s", hipGetErrorString(e));
  const long long ms = (long long)npoints * nsample;
  if (ms <= 0) return BTR_OK;
  BTR_REQUIRE(grad_out && idx, "group_points_grad: null pointer");
  BTR_REQUIRE(ms < 0x7fffffffLL && b < 65536, "group_points_grad: shape too large");
  hipLaunchKernelGGL(group_points_grad_kernel, dim3(cdiv(ms, 256), cdiv(c, kGroupCh), b),
                     dim3(256), 0, as_stream(stream), c, n, (int)ms, grad_out, idx, grad_points);
  return check_launch("group_points_grad");
}

}  // extern "C"
