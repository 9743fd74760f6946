// optimizer.hip -- Adam / AdamW over MANY parameter tensors in one launch.
//
// reference: the training scripts step torch.optim.Adam (train_Votenet_FSB.py:172, :231) /
// AdamW with two parameter groups (train_GF_FSB.py:233-244, :319) once per iteration.  torch's
// fused multi-tensor implementation passes the tensor list through kernel arguments, ~36
// tensors per launch: 3 launches for VoteNet's ~100 tensors, 12 for GroupFree3D's ~400
// (0.76 ms for 14 M parameters whose update moves 400 MB, i.e. ~60 us of HBM time: most blocks
// work on a bias of a few hundred elements).  Here the per-tensor pointers live in a device
// table (parameters / moments / the tensor's own step counter / its group: built once per
// parameter set), the gradient pointers of up to 448 tensors and the groups' learning rates and
// weight decays ride in the kernel arguments (a scheduler that changes lr every iteration --
// train_GF_FSB.py:322 -- costs nothing), and a static chunk map assigns 4096-element chunks to
// workgroups: one launch per 448 tensors.  The bias corrections are evaluated ON THE DEVICE from
// each tensor's step counter (the f32 scalar torch keeps in the optimizer state and has already
// incremented), in double like torch's kernel: no host mirror of the step exists that a graph
// replay, a fallback step or a parameter that joined late could leave stale.
//
// Update rule = at::native fused_adam_utils.cuh (adam_math), f32:
//   g = grad [/ grad_scale];  Adam: g += wd * p;  AdamW: p -= lr * wd * p
//   m = lerp(m, g, 1 - beta1);  v = beta2 * v + (1 - beta2) * g * g
//   p -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
#include <cmath>

#include "common.hpp"

namespace btr {
namespace {

constexpr int kChunk = 4096;   // elements per workgroup

__device__ __forceinline__ float lerpf(float a, float b, float w) {
  // std::lerp as ATen's at::native::lerp evaluates it
  return w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.f - w);
}

struct AdamHyper {   // (1 - beta) are formed in double on the host, as torch's kernel does
  double beta1d, beta2d;
  float beta2, omb1, omb2, eps;
  int decoupled;
};

__device__ __forceinline__ void adam_elem(float &p, float g, float &m, float &v, float lr,
                                          float wd, const AdamHyper &h, float inv_scale,
                                          float bc1, float bc2_sqrt) {
  g *= inv_scale;
  if (wd != 0.f) {
    if (h.decoupled) p -= lr * wd * p;
    else g += p * wd;
  }
  m = lerpf(m, g, h.omb1);
  v = h.beta2 * v + h.omb2 * g * g;
  const float step_size = lr / bc1;
  const float denom = sqrtf(v) / bc2_sqrt + h.eps;
  p -= step_size * m / denom;
}

// (the gradient pointers change every step -- autograd allocates the gradients anew -- and ride in
// the kernel arguments: no upload, no pinned staging buffers)
__global__ __launch_bounds__(256) void adam_multi_kernel(
    const btr_adam_item_t *__restrict__ items, btr_adam_grads_t grads, btr_adam_groups_t groups,
    int tensor0, const int2 *__restrict__ chunk_map, AdamHyper h,
    const float *__restrict__ grad_scale) {
  const int2 cm = chunk_map[blockIdx.x];   // (tensor, first element of the chunk)
  const btr_adam_item_t it = items[cm.x];
  // fused_adam_utils.cuh: 1 - pow(beta, *step) in double, then sqrt for the second moment
  __shared__ float bias[2];
  if (threadIdx.x == 0) {
    const double t = (double)*it.step;
    bias[0] = (float)(1.0 - pow(h.beta1d, t));
    bias[1] = (float)sqrt(1.0 - pow(h.beta2d, t));
  }
  __syncthreads();
  const float bc1 = bias[0], bc2s = bias[1];
  const float lr = groups.lr[it.group], wd = groups.wd[it.group];
  const float *__restrict__ g = grads.g[cm.x - tensor0];
  const float inv_scale = grad_scale ? 1.f / *grad_scale : 1.f;
  const long long e0 = cm.y, e1 = min((long long)cm.y + kChunk, it.n);
  float *__restrict__ p = it.p, *__restrict__ m = it.m, *__restrict__ v = it.v;
  if (it.vec) {
    for (long long e = e0 + 4 * threadIdx.x; e < e1; e += 4 * 256) {
      float4 pp = *reinterpret_cast<float4 *>(p + e);
      const float4 gg = *reinterpret_cast<const float4 *>(g + e);
      float4 mm = *reinterpret_cast<float4 *>(m + e);
      float4 vv = *reinterpret_cast<float4 *>(v + e);
      adam_elem(pp.x, gg.x, mm.x, vv.x, lr, wd, h, inv_scale, bc1, bc2s);
      adam_elem(pp.y, gg.y, mm.y, vv.y, lr, wd, h, inv_scale, bc1, bc2s);
      adam_elem(pp.z, gg.z, mm.z, vv.z, lr, wd, h, inv_scale, bc1, bc2s);
      adam_elem(pp.w, gg.w, mm.w, vv.w, lr, wd, h, inv_scale, bc1, bc2s);
      *reinterpret_cast<float4 *>(p + e) = pp;
      *reinterpret_cast<float4 *>(m + e) = mm;
      *reinterpret_cast<float4 *>(v + e) = vv;
    }
  } else {
    for (long long e = e0 + threadIdx.x; e < e1; e += 256) {
      float pp = p[e], mm = m[e], vv = v[e];
      adam_elem(pp, g[e], mm, vv, lr, wd, h, inv_scale, bc1, bc2s);
      p[e] = pp;
      m[e] = mm;
      v[e] = vv;
    }
  }
}

// ---- total gradient norm over the same table (clip_grad_norm_ folded into the step)
// pass 1: one workgroup per chunk of the chunk map -> that chunk's sum of squares
__global__ __launch_bounds__(256) void grad_sumsq_kernel(
    const btr_adam_item_t *__restrict__ items, btr_adam_grads_t grads, int tensor0,
    const int2 *__restrict__ chunk_map, float *__restrict__ partial) {
  const int2 cm = chunk_map[blockIdx.x];
  const btr_adam_item_t it = items[cm.x];
  const float *__restrict__ g = grads.g[cm.x - tensor0];
  const long long e0 = cm.y, e1 = min((long long)cm.y + kChunk, it.n);
  float s = 0.f;
  if (it.vec) {
    for (long long e = e0 + 4 * threadIdx.x; e < e1; e += 4 * 256) {
      const float4 gg = *reinterpret_cast<const float4 *>(g + e);
      s += gg.x * gg.x + gg.y * gg.y + gg.z * gg.z + gg.w * gg.w;
    }
  } else {
    for (long long e = e0 + threadIdx.x; e < e1; e += 256) s += g[e] * g[e];
  }
  __shared__ float red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// pass 2 (one workgroup, fixed order): out[0] = total norm, out[1] = max(1, (norm + 1e-6) / clip)
__global__ __launch_bounds__(256) void grad_norm_final_kernel(const float *__restrict__ partial,
                                                              int chunks, float clip,
                                                              float *__restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < chunks; i += 256) s += (double)partial[i];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float total = (float)sqrt(red[0]);
    out[0] = total;
    out[1] = fmaxf(1.f, (total + 1e-6f) / clip);
  }
}

}  // namespace
}  // namespace btr

using namespace btr;

extern "C" {

int btr_adam_chunk(void) { return kChunk; }

int btr_adam_multi(int chunks, int tensor0, const btr_adam_item_t *items,
                   const btr_adam_grads_t *grads, const btr_adam_groups_t *groups,
                   const int *chunk_map, double beta1, double beta2, double eps, int decoupled,
                   const float *grad_scale, btr_stream_t stream) {
  if (chunks <= 0) return BTR_OK;
  BTR_REQUIRE(items && grads && groups && chunk_map && tensor0 >= 0 && beta1 >= 0.0 &&
                  beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0,
              "adam_multi: bad arguments (betas %g %g)", beta1, beta2);
  AdamHyper h{};
  h.beta1d = beta1;
  h.beta2d = beta2;
  h.beta2 = (float)beta2;
  h.omb1 = (float)(1.0 - beta1);
  h.omb2 = (float)(1.0 - beta2);
  h.eps = (float)eps;
  h.decoupled = decoupled;
  hipLaunchKernelGGL(adam_multi_kernel, dim3(chunks), dim3(256), 0, as_stream(stream), items,
                     *grads, *groups, tensor0, reinterpret_cast<const int2 *>(chunk_map), h,
                     grad_scale);
  return check_launch("adam_multi");
}

int btr_grad_sumsq_multi(int chunks, int tensor0, const btr_adam_item_t *items,
                         const btr_adam_grads_t *grads, const int *chunk_map, float *partial,
                         btr_stream_t stream) {
  if (chunks <= 0) return BTR_OK;
  BTR_REQUIRE(items && grads && chunk_map && partial && tensor0 >= 0,
              "grad_sumsq_multi: bad arguments");
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(chunks), dim3(256), 0, as_stream(stream), items,
                     *grads, tensor0, reinterpret_cast<const int2 *>(chunk_map), partial);
  return check_launch("grad_sumsq_multi");
}

int btr_grad_norm_final(int chunks, const float *partial, float clip, float *out,
                        btr_stream_t stream) {
  BTR_REQUIRE(chunks > 0 && partial && out && clip > 0.f, "grad_norm_final: bad arguments");
  hipLaunchKernelGGL(grad_norm_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), partial,
                     chunks, clip, out);
  return check_launch("grad_norm_final");
}

}  // extern "C"
