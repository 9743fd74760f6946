// gf_loss.hip -- the per-head part of GroupFree3D's training loss and its gradient.
//
// reference: detection/GroupFree3D/models/loss_helper.py:81-275
// (compute_objectness_loss_based_on_query_points, compute_box_and_sem_cls_loss) with
// models/losses.py (SigmoidFocalClassificationLoss, smoothl1_loss): for every prediction head
// (proposal head + one per decoder layer) the sigmoid focal objectness loss of the query points
// and, on the query points that lie in an object, centre / heading / size / semantic losses
// against that object's labels.  All heads share their targets, so one launch handles every
// (head, scene, query point); the op-by-op form is ~150 torch launches forward and ~200 backward.
//
//   gf_targets_kernel   per scene: objectness label + assigned ground-truth slot of every query
//                       point (through seed_inds / sample_inds), number of positives
//   gf_heads_kernel     one thread per (head, scene, query): the raw head output row is staged
//                       through LDS from the head's (B, C, P) tensor, the seven loss terms are
//                       summed per wave, and the gradient w.r.t. the row is written back in the
//                       same (B, C, P) layout (what the head's backward takes)
//   gf_final_kernel     per-head terms, their sums and the weighted total
// No float atomics: partial sums per workgroup, reduced in a fixed order.
#include <cmath>

#include "common.hpp"

namespace btr {
namespace {

constexpr int kMaxHeads = 8;
constexpr int kMaxC = 192;   // channels of a head output row staged in LDS (64 rows)
constexpr int kTerms = 7;    // obj, center, heading cls, heading reg, size cls, size reg, sem

struct HeadPtrs {
  const float *p[kMaxHeads];
};

__device__ __forceinline__ float sl1(float e, float d) {
  const float a = fabsf(e);
  return a < d ? 0.5f * a * a / d : a - 0.5f * d;
}
__device__ __forceinline__ float sl1_grad(float e, float d) {
  const float a = fabsf(e);
  return a < d ? e / d : (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
}

__global__ __launch_bounds__(256) void gf_targets_kernel(
    int P, int S1, int N, int K2, const int *__restrict__ seed_inds,
    const int *__restrict__ sample_inds, const long long *__restrict__ point_obj_mask,
    const long long *__restrict__ point_instance_label, long long *__restrict__ objectness_label,
    long long *__restrict__ object_assignment, float *__restrict__ npos_part) {
  __shared__ int red[256];
  const int b = blockIdx.x;
  int cnt = 0;
  for (int k = threadIdx.x; k < P; k += 256) {
    const int s = sample_inds[(size_t)b * P + k];
    const int pt = seed_inds[(size_t)b * S1 + s];
    const long long obj = point_obj_mask[(size_t)b * N + pt];
    const long long inst = point_instance_label[(size_t)b * N + pt];
    objectness_label[(size_t)b * P + k] = obj;
    object_assignment[(size_t)b * P + k] = inst < 0 ? (long long)(K2 - 1) : inst;
    cnt += obj != 0 ? 1 : 0;   // (the reference sums the label as a float: labels are 0 / 1)
  }
  red[threadIdx.x] = cnt;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) npos_part[b] = (float)red[0];
}

struct GfDims {
  int b, p, k2, nh, ns, nc, heads, c;
  float w_obj, w_box, w_sem;   // 10 * coef / (num_decoder_layers + 1)
  float d_center, d_heading, d_size;
  int weak;   // centre labels only (loss_helper.py:479-554): see gf_heads_kernel
};

// #positives per scene from given labels (the weakly supervised branch makes its labels itself)
__global__ __launch_bounds__(256) void gf_npos_kernel(int P, const long long *__restrict__ label,
                                                      float *__restrict__ npos_part) {
  __shared__ int red[256];
  const int b = blockIdx.x;
  int cnt = 0;
  for (int k = threadIdx.x; k < P; k += 256) cnt += label[(size_t)b * P + k] != 0 ? 1 : 0;
  red[threadIdx.x] = cnt;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) npos_part[b] = (float)red[0];
}

// log-sum-exp cross entropy of t[o .. o+n) (LDS column `lane`) with target `tg`; the scores are
// replaced by coef * (softmax - onehot)
__device__ __forceinline__ float ce_inplace(float (*t)[64], int lane, int o, int n, int tg,
                                            float coef) {
  float m = -INFINITY;
  for (int i = 0; i < n; ++i) m = fmaxf(m, t[o + i][lane]);
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += expf(t[o + i][lane] - m);
  const float lse = m + logf(s);
  const float loss = lse - t[o + tg][lane];
  const float inv = 1.f / s;
  for (int i = 0; i < n; ++i) {
    const float pr = expf(t[o + i][lane] - m) * inv;
    t[o + i][lane] = coef * (pr - (i == tg ? 1.f : 0.f));
  }
  return loss;
}

__global__ __launch_bounds__(64) void gf_heads_kernel(
    GfDims d, HeadPtrs heads, const float *__restrict__ base_xyz,
    const long long *__restrict__ objectness_label, const long long *__restrict__ assignment,
    const float *__restrict__ npos_part, const float *__restrict__ center_label,
    const long long *__restrict__ heading_class_label,
    const float *__restrict__ heading_residual_label,
    const long long *__restrict__ size_class_label, const float *__restrict__ size_residual_label,
    const long long *__restrict__ sem_cls_label, const float *__restrict__ mean_size,
    float *__restrict__ part, float *__restrict__ grads) {
  __shared__ float t[kMaxC][64];
  const int lane = threadIdx.x;
  const int p0 = blockIdx.x * 64, b = blockIdx.y, h = blockIdx.z;
  const int p = p0 + lane;
  const bool live = p < d.p;
  const float *src = heads.p[h] + (size_t)b * d.c * d.p;
  for (int c = 0; c < d.c; ++c) t[c][lane] = live ? src[(size_t)c * d.p + p] : 0.f;
  float npos = 1e-6f;
  for (int i = 0; i < d.b; ++i) npos += npos_part[i];   // (fixed order)
  float term[kTerms];
#pragma unroll
  for (int i = 0; i < kTerms; ++i) term[i] = 0.f;
  if (live) {
    const size_t q = (size_t)b * d.p + p;
    const float lab = objectness_label[q] != 0 ? 1.f : 0.f;
    const int a = (int)assignment[q];
    const size_t ga = (size_t)b * d.k2 + a;
    // ---- objectness: sigmoid focal loss (alpha 0.25, gamma 2), weight 1 / P, summed / B
    {
      const float x = t[0][lane];
      const float pr = 1.f / (1.f + expf(-x));
      const float aw = lab * 0.25f + (1.f - lab) * 0.75f;
      const float pt = lab * (1.f - pr) + (1.f - lab) * pr;
      const float bce = fmaxf(x, 0.f) - x * lab + log1pf(expf(-fabsf(x)));
      const float w = 1.f / (float)d.p;
      term[0] = aw * pt * pt * bce * w;
      const float dpt = (1.f - 2.f * lab) * pr * (1.f - pr);
      t[0][lane] = d.w_obj / (float)d.b * aw * w * (2.f * pt * dpt * bce + pt * pt * (pr - lab));
    }
    const float cb = d.w_box * lab / npos;   // box terms: masked by the label, / #positives
    const int o_hc = 4, o_hr = 4 + d.nh, o_sc = 4 + 2 * d.nh, o_sr = o_sc + d.ns;
    const int o_sem = o_sr + 3 * d.ns;
    const int sc = (int)size_class_label[ga];
    if (d.weak) {
      // ---- weak labels: the centre with a dead zone of 5 % of the class's mean size
      // (clamp(smooth-L1 - margin, min = 0) per component; autograd's clamp passes the gradient
      // at 0), the size class and the semantic class; no heading, no size residual
      for (int j = 0; j < 3; ++j) {
        const float e = center_label[ga * 3 + j] - (base_xyz[q * 3 + j] + t[1 + j][lane]);
        const float v = sl1(e, d.d_center) - 0.05f * mean_size[sc * 3 + j];
        term[1] += fmaxf(v, 0.f) * lab;
        t[1 + j][lane] = v >= 0.f ? -cb * sl1_grad(e, d.d_center) : 0.f;
      }
      for (int i = 0; i < 2 * d.nh; ++i) t[o_hc + i][lane] = 0.f;
      for (int i = 0; i < 3 * d.ns; ++i) t[o_sr + i][lane] = 0.f;
    } else {
    // ---- centre (smooth-L1 of gt - (base + residual))
    for (int j = 0; j < 3; ++j) {
      const float e = center_label[ga * 3 + j] - (base_xyz[q * 3 + j] + t[1 + j][lane]);
      term[1] += sl1(e, d.d_center) * lab;
      t[1 + j][lane] = -cb * sl1_grad(e, d.d_center);
    }
    // ---- heading class / residual
    const int hc = (int)heading_class_label[ga];
    {
      const float pred = t[o_hr + hc][lane];
      const float e = pred - heading_residual_label[ga] / (3.14159265358979323846f / (float)d.nh);
      term[3] = d.d_heading * sl1(e, d.d_heading) * lab;
      for (int i = 0; i < d.nh; ++i) t[o_hr + i][lane] = 0.f;
      t[o_hr + hc][lane] = cb * d.d_heading * sl1_grad(e, d.d_heading);
    }
    term[2] = ce_inplace(t, lane, o_hc, d.nh, hc, 0.1f * cb) * lab;
    // ---- size residual
    {
      float g3[3];
      for (int j = 0; j < 3; ++j) {
        const float pred = t[o_sr + 3 * sc + j][lane];
        const float e = pred - size_residual_label[ga * 3 + j] / mean_size[sc * 3 + j];
        term[5] += d.d_size * sl1(e, d.d_size) * lab;
        g3[j] = cb * d.d_size * sl1_grad(e, d.d_size);
      }
      for (int i = 0; i < 3 * d.ns; ++i) t[o_sr + i][lane] = 0.f;
      for (int j = 0; j < 3; ++j) t[o_sr + 3 * sc + j][lane] = g3[j];
    }
    }
    // ---- size class
    term[4] = ce_inplace(t, lane, o_sc, d.ns, sc, 0.1f * cb) * lab;
    // ---- semantic class
    term[6] = ce_inplace(t, lane, o_sem, d.nc, (int)sem_cls_label[ga], d.w_sem * lab / npos) * lab;
  }
#pragma unroll
  for (int i = 0; i < kTerms; ++i) {
    float v = term[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    term[i] = v;
  }
  const int blk = (h * d.b + b) * gridDim.x + blockIdx.x;
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < kTerms; ++i) part[(size_t)blk * kTerms + i] = term[i];
  float *dst = grads + ((size_t)h * d.b + b) * d.c * d.p;
  if (live)
    for (int c = 0; c < d.c; ++c) dst[(size_t)c * d.p + p] = t[c][lane];
}

// stats: [h][8] = obj, center, heading cls, heading reg, size cls, size reg, box, sem;
// then [8H + 0..5] = sum obj, sum box, sum sem, weighted total, pos_ratio, neg_ratio
__global__ __launch_bounds__(64) void gf_final_kernel(GfDims d, int blocks_per_scene,
                                                      const float *__restrict__ part,
                                                      const float *__restrict__ npos_part,
                                                      float *__restrict__ stats) {
  __shared__ float hs[kMaxHeads][8];
  const int h = threadIdx.x >> 3, i = threadIdx.x & 7;   // 8 heads x 8 slots
  float npos = 0.f;
  for (int k = 0; k < d.b; ++k) npos += npos_part[k];
  if (h < d.heads && i < kTerms) {
    float s = 0.f;
    const int n = d.b * blocks_per_scene;
    for (int k = 0; k < n; ++k) s += part[((size_t)h * n + k) * kTerms + i];
    hs[h][i] = i == 0 ? s / (float)d.b : s / (npos + 1e-6f);
  }
  __syncthreads();
  if (threadIdx.x < d.heads) {
    const int hh = threadIdx.x;
    const float box = hs[hh][1] + 0.1f * hs[hh][2] + hs[hh][3] + 0.1f * hs[hh][4] + hs[hh][5];
    float *o = stats + hh * 8;
    for (int k = 0; k < 6; ++k) o[k] = hs[hh][k];
    o[6] = box;
    o[7] = hs[hh][6];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float so = 0.f, sb = 0.f, ss = 0.f;
    for (int hh = 0; hh < d.heads; ++hh) {
      so += stats[hh * 8];
      sb += stats[hh * 8 + 6];
      ss += stats[hh * 8 + 7];
    }
    float *o = stats + 8 * d.heads;
    o[0] = so; o[1] = sb; o[2] = ss;
    o[3] = d.w_obj * so + d.w_box * sb + d.w_sem * ss;
    const float total = (float)d.b * (float)d.p;
    o[4] = npos / total;
    o[5] = 1.f - npos / total;
    o[6] = o[3];   // the word the returned loss TENSOR lives on: `loss *= w` leaves o[3] alone
  }
}

// ------------------------------------------------------------------ decode of a PredictHead
// reference: detection/GroupFree3D/models/modules.py:233-262 -- from the head's raw output the
// box centre (base_xyz + residual), the heading / size residuals in metres and the size of the
// arg-max size class; six elementwise / arg-max / gather launches per head there (seven heads per
// step), plus the clones and the concatenation that build the next decoder layer's query position
// (detector.py:204-230).  One wave per proposal here; element (b, p, ch) of the head output lives
// at out[b * sb + p * sp + ch * sc] (the channel-last twin or the (b, c, p) tensor).
struct DecodeArgs {
  int rows, p, nh, ns;
  const float *out;
  long long sb, sp, sc;
  const float *base_xyz, *mean_size;
  float *center, *hres, *sres, *pred_size, *qpos, *qpos_t;
  float hscale;
};
__global__ __launch_bounds__(256) void gf_head_decode_kernel(DecodeArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
  if (row >= a.rows) return;
  const int b = row / a.p, pp = row - b * a.p;
  const float *o = a.out + b * a.sb + pp * a.sp;
  const int o_hres = 4 + a.nh, o_ss = 4 + 2 * a.nh, o_sr = o_ss + a.ns;
  // arg-max of the size scores: first index of the maximum (what torch.argmax returns here)
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int s = lane; s < a.ns; s += 64) {
    const float v = o[(o_ss + s) * a.sc];
    if (v > best || (v == best && s < bi) || (v != v && !(best != best))) {
      best = v;
      bi = s;
    }
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const float ov = __shfl_xor(best, m);
    const int oi = __shfl_xor(bi, m);
    const bool take = (ov != ov) ? (!(best != best) || oi < bi)
                                 : (!(best != best) && (ov > best || (ov == best && oi < bi)));
    if (take) {
      best = ov;
      bi = oi;
    }
  }
  if (bi == 0x7fffffff) bi = 0;
  for (int h = lane; h < a.nh; h += 64)
    a.hres[(size_t)row * a.nh + h] = __fmul_rn(o[(o_hres + h) * a.sc], a.hscale);
  for (int i = lane; i < 3 * a.ns; i += 64)
    a.sres[(size_t)row * 3 * a.ns + i] = __fmul_rn(o[(o_sr + i) * a.sc], a.mean_size[i]);
  if (lane < 6) {
    float v;
    if (lane < 3) {
      v = __fadd_rn(a.base_xyz[(size_t)row * 3 + lane], o[(1 + lane) * a.sc]);
      a.center[(size_t)row * 3 + lane] = v;
    } else {
      const int c = lane - 3, i = bi * 3 + c;
      const float ms = a.mean_size[i];
      v = __fadd_rn(__fmul_rn(o[(o_sr + i) * a.sc], ms), ms);
      a.pred_size[(size_t)row * 3 + c] = v;
    }
    a.qpos[(size_t)row * 6 + lane] = v;
    a.qpos_t[((size_t)b * 6 + lane) * a.p + pp] = v;
  }
}


// ---- sigmoid focal loss of one logit per point, summed (GroupFree3D's objectness of the seed
// points: loss_helper.py:17-78 with SigmoidFocalClassificationLoss, losses.py:21-81):
//   p = sigmoid(x), a = t alpha + (1 - t)(1 - alpha), pt = t (1 - p) + (1 - t) p,
//   bce = max(x, 0) - x t + log1p(exp(-|x|)),   value = scale * sum_i a pt^gamma bce w
// ~40 element-wise / reduction launches forward and ~30 backward as torch ops on 4 096 floats;
// here ONE workgroup computes the value and d value / d x for a unit upstream gradient (fixed
// summation order: thread-strided partial sums, wave shuffles, four waves in order).
__global__ __launch_bounds__(256) void focal_sum_kernel(int n, int period,
                                                        const float *__restrict__ x,
                                                        const long long *__restrict__ label,
                                                        float w, float scale, float gamma,
                                                        float alpha, float *__restrict__ out,
                                                        float *__restrict__ grad) {
  // workgroup g: elements [g n, (g + 1) n) of x, their labels label[i % period] (the same labels
  // for every group: one objectness target for all prediction heads)
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long base = (long long)blockIdx.x * n;
  float acc = 0.f;
  for (int j = tid; j < n; j += 256) {
    const long long i = base + j;
    const float xi = x[i], t = (float)label[i % period];
    const float p = 1.f / (1.f + expf(-xi));
    const float a = t * alpha + (1.f - t) * (1.f - alpha);
    const float pt = t * (1.f - p) + (1.f - t) * p;
    const float e = expf(-fabsf(xi));
    const float bce = fmaxf(xi, 0.f) - xi * t + log1pf(e);
    const float pg = powf(pt, gamma);
    acc += a * pg * bce * w;
    // d pt / dx = (1 - 2 t) p (1 - p);  d bce / dx term by term as autograd differentiates the
    // composition (clamp passes the gradient at x >= 0, |x| has slope 0 at 0): p - t everywhere
    // except at x == 0 exactly, where it is 1 - t
    const float sgn = xi > 0.f ? 1.f : (xi < 0.f ? -1.f : 0.f);
    const float dbce = (xi >= 0.f ? 1.f : 0.f) - t - sgn * e / (1.f + e);
    const float dpt = (1.f - 2.f * t) * p * (1.f - p);
    const float dpg = pt > 0.f ? gamma * powf(pt, gamma - 1.f) * dpt : 0.f;
    grad[i] = scale * w * a * (dpg * bce + pg * dbce);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = scale * (((red[0] + red[1]) + red[2]) + red[3]);
}

}  // namespace
}  // namespace btr

using namespace btr;

extern "C" {

int btr_gf_loss_fwd(const btr_gf_loss_t *dp, const float *const *heads, const float *base_xyz,
                    const int *seed_inds, const int *sample_inds,
                    const long long *point_obj_mask, const long long *point_instance_label,
                    const float *center_label, const long long *heading_class_label,
                    const float *heading_residual_label, const long long *size_class_label,
                    const float *size_residual_label, const long long *sem_cls_label,
                    const float *mean_size, long long *objectness_label,
                    long long *object_assignment, float *npos_part, float *part, float *stats,
                    float *grads, btr_stream_t stream) {
  BTR_REQUIRE(dp && heads && base_xyz && seed_inds && sample_inds && point_obj_mask &&
                  point_instance_label && center_label && heading_class_label &&
                  heading_residual_label && size_class_label && size_residual_label &&
                  sem_cls_label && mean_size && objectness_label && object_assignment &&
                  npos_part && part && stats && grads,
              "gf_loss_fwd: null pointer");
  const btr_gf_loss_t &s = *dp;
  BTR_REQUIRE(s.b > 0 && s.b < 65536 && s.p > 0 && s.k2 > 0 && s.nh > 0 && s.ns > 0 && s.nc > 0 &&
                  s.heads >= 1 && s.heads <= kMaxHeads && s.s1 > 0 && s.n > 0,
              "gf_loss_fwd: bad sizes");
  BTR_REQUIRE(s.c == 4 + 2 * s.nh + 4 * s.ns + s.nc && s.c <= kMaxC,
              "gf_loss_fwd: %d channels for nh=%d ns=%d nc=%d (max %d)", s.c, s.nh, s.ns, s.nc,
              kMaxC);
  hipStream_t hs = as_stream(stream);
  GfDims d{s.b, s.p, s.k2, s.nh, s.ns, s.nc, s.heads, s.c, s.w_obj, s.w_box, s.w_sem,
           s.center_delta, s.heading_delta, s.size_delta, 0};
  HeadPtrs hp{};
  for (int i = 0; i < s.heads; ++i) {
    BTR_REQUIRE(heads[i], "gf_loss_fwd: head %d is null", i);
    hp.p[i] = heads[i];
  }
  hipLaunchKernelGGL(gf_targets_kernel, dim3(s.b), dim3(256), 0, hs, s.p, s.s1, s.n, s.k2,
                     seed_inds, sample_inds, point_obj_mask, point_instance_label,
                     objectness_label, object_assignment, npos_part);
  const int gx = cdiv(s.p, 64);
  hipLaunchKernelGGL(gf_heads_kernel, dim3(gx, s.b, s.heads), dim3(64), 0, hs, d, hp, base_xyz,
                     objectness_label, object_assignment, npos_part, center_label,
                     heading_class_label, heading_residual_label, size_class_label,
                     size_residual_label, sem_cls_label, mean_size, part, grads);
  hipLaunchKernelGGL(gf_final_kernel, dim3(1), dim3(64), 0, hs, d, gx, part, npos_part, stats);
  return check_launch("gf_loss_fwd");
}

// The weakly supervised per-head loss (centre labels only): objectness_label / object_assignment
// (b, p) are INPUTS (the caller made them: nearest labelled centre, positive within 0.3 m).
int btr_gf_loss_weak_fwd(const btr_gf_loss_t *dp, const float *const *heads,
                         const float *base_xyz, const long long *objectness_label,
                         const long long *object_assignment, const float *center_label,
                         const long long *size_class_label, const long long *sem_cls_label,
                         const float *mean_size, float *npos_part, float *part, float *stats,
                         float *grads, btr_stream_t stream) {
  BTR_REQUIRE(dp && heads && base_xyz && objectness_label && object_assignment && center_label &&
                  size_class_label && sem_cls_label && mean_size && npos_part && part && stats &&
                  grads,
              "gf_loss_weak_fwd: null pointer");
  const btr_gf_loss_t &s = *dp;
  BTR_REQUIRE(s.b > 0 && s.b < 65536 && s.p > 0 && s.k2 > 0 && s.nh > 0 && s.ns > 0 && s.nc > 0 &&
                  s.heads >= 1 && s.heads <= kMaxHeads,
              "gf_loss_weak_fwd: bad sizes");
  BTR_REQUIRE(s.c == 4 + 2 * s.nh + 4 * s.ns + s.nc && s.c <= kMaxC,
              "gf_loss_weak_fwd: %d channels for nh=%d ns=%d nc=%d (max %d)", s.c, s.nh, s.ns,
              s.nc, kMaxC);
  hipStream_t hs = as_stream(stream);
  GfDims d{s.b, s.p, s.k2, s.nh, s.ns, s.nc, s.heads, s.c, s.w_obj, s.w_box, s.w_sem,
           s.center_delta, s.heading_delta, s.size_delta, 1};
  HeadPtrs hp{};
  for (int i = 0; i < s.heads; ++i) {
    BTR_REQUIRE(heads[i], "gf_loss_weak_fwd: head %d is null", i);
    hp.p[i] = heads[i];
  }
  hipLaunchKernelGGL(gf_npos_kernel, dim3(s.b), dim3(256), 0, hs, s.p, objectness_label,
                     npos_part);
  const int gx = cdiv(s.p, 64);
  hipLaunchKernelGGL(gf_heads_kernel, dim3(gx, s.b, s.heads), dim3(64), 0, hs, d, hp, base_xyz,
                     objectness_label, object_assignment, npos_part, center_label, nullptr,
                     nullptr, size_class_label, nullptr, sem_cls_label, mean_size, part, grads);
  hipLaunchKernelGGL(gf_final_kernel, dim3(1), dim3(64), 0, hs, d, gx, part, npos_part, stats);
  return check_launch("gf_loss_weak_fwd");
}

// value[g] = scale * sum over group g's n elements of focal(x_i, label_{i % period}) * w and
// grad[i] = d value[g(i)] / d x_i (unit upstream gradient); label in {0, 1}; x holds groups * n
// elements
int btr_focal_sum(int groups, int n, int period, const float *x, const long long *label, float w,
                  float scale, float gamma, float alpha, float *value, float *grad,
                  btr_stream_t stream) {
  if (n <= 0 || groups <= 0) return BTR_OK;
  BTR_REQUIRE(x && label && value && grad && period > 0, "focal_sum: null pointer");
  hipLaunchKernelGGL(focal_sum_kernel, dim3(groups), dim3(256), 0, as_stream(stream), n, period,
                     x, label, w, scale, gamma, alpha, value, grad);
  return check_launch("focal_sum");
}

int btr_gf_loss_part_floats(int b, int p, int heads) { return heads * b * cdiv(p, 64) * kTerms; }

int btr_gf_head_decode(int b, int p, int nh, int ns, const float *out, long long sb, long long sp,
                       long long sc, const float *base_xyz, const float *mean_size, float *center,
                       float *heading_residuals, float *size_residuals, float *pred_size,
                       float *query_pos, float *query_pos_t, btr_stream_t stream) {
  if (b <= 0 || p <= 0) return BTR_OK;
  BTR_REQUIRE(out && base_xyz && mean_size && center && heading_residuals && size_residuals &&
                  pred_size && query_pos && query_pos_t && nh > 0 && ns > 0,
              "gf_head_decode: null pointer or nh=%d ns=%d", nh, ns);
  DecodeArgs a{b * p, p, nh, ns, out, sb, sp, sc, base_xyz, mean_size, center, heading_residuals,
               size_residuals, pred_size, query_pos, query_pos_t, (float)(M_PI / nh)};
  hipLaunchKernelGGL(gf_head_decode_kernel, dim3(cdiv(b * p, 4)), dim3(256), 0, as_stream(stream),
                     a);
  return check_launch("gf_head_decode");
}

}  // extern "C"
