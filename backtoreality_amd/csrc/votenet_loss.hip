// votenet_loss.hip -- the VoteNet loss (detection/Votenet/models/loss_helper.py:336-400 with
// compute_vote_loss :24-69, compute_objectness_loss :111-152, compute_box_and_sem_cls_loss
// :154-228, nn_distance utils/nn_distance.py:34-61) as three kernels instead of ~250 tiny
// torch launches (forward + autograd backward).  This is the caller right after the hot path
// (SURVEY 8f "next #1"); it only matters because everything before it got fast.
//
//   loss_terms_kernel   one workgroup per batch element: nearest-GT assignment, objectness
//                       labels/masks, every per-proposal loss term, the vote term, block sums
//   loss_reduce_kernel  batch sums -> the 13 scalars of get_loss + the normalisers
//   loss_grad_kernel    gradients w.r.t. the raw head output (B, Cout, K), the aggregated
//                       vote xyz and the vote xyz, scaled by dL/d(loss)
// Head channel layout (proposal_module.py:18-50): [objectness 2 | centre offset 3 |
// heading scores NH | heading residuals NH | size scores NS | size residuals NS*3 | sem NC].
#include <algorithm>
#include <cmath>

#include "common.hpp"

namespace btr {

constexpr int kMaxObj = 256;   // K2 (MAX_NUM_OBJ: 64 scannet, 256 matterport)
constexpr int kMaxProp = 1024; // K
constexpr int kNSums = 16;

struct LossDims {
  int B, K, K2, NH, NS, NC, S1, N, Cout;
  // Back-to-Reality (get_loss_DA, loss_helper.py:548-664): per-term weights of
  // (vote, objectness, center, heading_cls, heading_reg, size_cls, size_reg, sem_cls) in
  // loss/10, and vote_mode 1 = compute_weak_vote_loss (:71-109) instead of compute_vote_loss.
  float w[8];
  int vote_mode;
};

__device__ __forceinline__ float huber1(float e) {  // delta = 1 (nn_distance.py:15-32)
  const float a = fabsf(e);
  const float q = fminf(a, 1.f);
  return 0.5f * q * q + (a - q);
}
__device__ __forceinline__ float huber1_grad(float e) {
  return fabsf(e) <= 1.f ? e : (e > 0.f ? 1.f : -1.f);
}

// log-sum-exp of `n` logits net[ch0 + c][k] (stride `cs` between channels)
__device__ __forceinline__ float lse(const float *p, size_t cs, int n, float &mx) {
  mx = p[0];
  for (int c = 1; c < n; ++c) mx = fmaxf(mx, p[c * cs]);
  float s = 0.f;
  for (int c = 0; c < n; ++c) s += expf(p[c * cs] - mx);
  return mx + logf(s);
}

// Cross-entropy of `n` logits against class `pick`: lse - p[pick].  Up to kLogitRegs classes the
// logits are loaded ONCE, all loads in flight together, and the picked one is selected from the
// registers (the two-pass loop above waits for every load before the next: ~110 dependent L2 round
// trips per proposal over the three class blocks, 28 us for a kernel of one workgroup per scene).
// Same operations in the same order as lse(): bit-identical.
constexpr int kLogitRegs = 24;
__device__ __forceinline__ float ce_pick(const float *p, size_t cs, int n, int pick) {
  if (n > kLogitRegs) {
    float mx;
    return lse(p, cs, n, mx) - p[(size_t)pick * cs];
  }
  float v[kLogitRegs];
#pragma unroll
  for (int c = 0; c < kLogitRegs; ++c) v[c] = c < n ? p[c * cs] : -3.0e38f;
  float mx = v[0], got = v[0];
#pragma unroll
  for (int c = 1; c < kLogitRegs; ++c) {
    mx = fmaxf(mx, v[c]);
    got = c == pick ? v[c] : got;
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < kLogitRegs; ++c) s += c < n ? expf(v[c] - mx) : 0.f;
  return mx + logf(s) - got;
}
// ... and its gradient: g[c] = w * (softmax(p)[c] - (c == pick)) for c < n
__device__ __forceinline__ void ce_grad(const float *p, float *g, size_t cs, int n, int pick,
                                        float w) {
  if (n > kLogitRegs) {
    float mx;
    const float l = lse(p, cs, n, mx);
    for (int c = 0; c < n; ++c) g[c * cs] = w * (expf(p[c * cs] - l) - (c == pick ? 1.f : 0.f));
    return;
  }
  float v[kLogitRegs];
#pragma unroll
  for (int c = 0; c < kLogitRegs; ++c) v[c] = c < n ? p[c * cs] : -3.0e38f;
  float mx = v[0];
#pragma unroll
  for (int c = 1; c < kLogitRegs; ++c) mx = fmaxf(mx, v[c]);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < kLogitRegs; ++c) s += c < n ? expf(v[c] - mx) : 0.f;
  const float l = mx + logf(s);
#pragma unroll
  for (int c = 0; c < kLogitRegs; ++c)
    if (c < n) g[c * cs] = w * (expf(v[c] - l) - (c == pick ? 1.f : 0.f));
}

enum {  // block / batch sums
  S_LABEL, S_MASK, S_BOXMASK, S_VOTEMASK, S_OBJ, S_D1C, S_D2C, S_HCLS, S_HREG, S_SCLS, S_SREG,
  S_SEM, S_VOTE, S_ACC, S_D2V
};
// The three independent parts of the loss run as three workgroups per scene (blockIdx.y): the
// per-proposal terms, the centres' second chamfer direction, the vote loss -- one workgroup per
// scene walked them one after the other, 41 us of dependent phases on 8 of 256 CUs.  Every sum
// slot has ONE owner, so the partial sums are bit-identical to the single-workgroup form.
enum { R_PROPOSALS, R_CENTRES, R_VOTES, kLossRoles };
__device__ __forceinline__ int sum_owner(int slot) {
  return (slot == S_D2C || slot == S_BOXMASK) ? R_CENTRES
         : (slot == S_VOTE || slot == S_VOTEMASK || slot == S_D2V) ? R_VOTES : R_PROPOSALS;
}

__global__ __launch_bounds__(256) void loss_terms_kernel(
    LossDims d, const float *__restrict__ net, const float *__restrict__ agg_xyz,
    const float *__restrict__ vote_xyz, const float *__restrict__ seed_xyz,
    const int *__restrict__ seed_inds, const float *__restrict__ vote_label,
    const long long *__restrict__ vote_label_mask, const float *__restrict__ center_label,
    const float *__restrict__ box_label_mask, const long long *__restrict__ heading_class_label,
    const float *__restrict__ heading_residual_label,
    const long long *__restrict__ size_class_label, const float *__restrict__ size_residual_label,
    const long long *__restrict__ sem_cls_label, const float *__restrict__ mean_size,
    long long *__restrict__ objectness_label, float *__restrict__ objectness_mask,
    long long *__restrict__ object_assignment, int *__restrict__ j1c, int *__restrict__ k2c,
    signed char *__restrict__ vote_arg, float *__restrict__ part, int *__restrict__ i2v) {
  __shared__ float gt[kMaxObj * 3];
  __shared__ float cen[kMaxProp * 3];
  __shared__ float red[kNSums][4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int role = blockIdx.y;
  const size_t cs = (size_t)d.K;  // channel stride of net
  const float *nb = net + (size_t)b * d.Cout * d.K;
  float sums[kNSums];
#pragma unroll
  for (int i = 0; i < kNSums; ++i) sums[i] = 0.f;

  for (int i = tid; i < d.K2 * 3; i += 256) gt[i] = center_label[(size_t)b * d.K2 * 3 + i];
  __syncthreads();

  const int oH = 5, oHR = 5 + d.NH, oS = 5 + 2 * d.NH, oSR = oS + d.NS, oC = oS + 4 * d.NS;
  if (role == R_CENTRES)   // the predicted centres, for the scan below
    for (int k = tid; k < d.K; k += 256) {
      const float *a = agg_xyz + ((size_t)b * d.K + k) * 3;
      cen[k * 3 + 0] = a[0] + nb[2 * cs + k];
      cen[k * 3 + 1] = a[1] + nb[3 * cs + k];
      cen[k * 3 + 2] = a[2] + nb[4 * cs + k];
    }
  for (int k = tid; role == R_PROPOSALS && k < d.K; k += 256) {
    const float *a = agg_xyz + ((size_t)b * d.K + k) * 3;
    const float ax = a[0], ay = a[1], az = a[2];
    const float cx = ax + nb[2 * cs + k], cy = ay + nb[3 * cs + k], cz = az + nb[4 * cs + k];
    float d1 = 3.0e38f, d1c = 3.0e38f;
    int i1 = 0, i1c = 0;
    for (int j = 0; j < d.K2; ++j) {
      const float gx = gt[j * 3], gy = gt[j * 3 + 1], gz = gt[j * 3 + 2];
      float ex = ax - gx, ey = ay - gy, ez = az - gz;
      const float q = ex * ex + ey * ey + ez * ez;
      if (q < d1) { d1 = q; i1 = j; }
      ex = cx - gx; ey = cy - gy; ez = cz - gz;
      const float qc = ex * ex + ey * ey + ez * ez;
      if (qc < d1c) { d1c = qc; i1c = j; }
    }
    const float e1 = sqrtf(d1 + 1e-6f);
    const int label = e1 < 0.3f ? 1 : 0;                       // NEAR_THRESHOLD
    const float mask = (e1 < 0.3f || e1 > 0.6f) ? 1.f : 0.f;   // FAR_THRESHOLD
    const size_t o = (size_t)b * d.K + k;
    objectness_label[o] = label;
    objectness_mask[o] = mask;
    object_assignment[o] = i1;
    j1c[o] = i1c;
    const float lab = (float)label;
    sums[S_LABEL] += lab;
    sums[S_MASK] += mask;
    // objectness: weighted CE, weights (0.2, 0.8)
    float mx;
    const float l0 = lse(nb + k, cs, 2, mx);
    const float w = label ? 0.8f : 0.2f;
    sums[S_OBJ] += w * (l0 - nb[label * cs + k]) * mask;
    const int pred = nb[cs + k] > nb[k] ? 1 : 0;
    sums[S_ACC] += (pred == label ? 1.f : 0.f) * mask;
    sums[S_D1C] += d1c * lab;
    // labels of the assigned GT box
    const size_t jo = (size_t)b * d.K2 + i1;
    const int hcl = (int)heading_class_label[jo];
    const int scl = (int)size_class_label[jo];
    const int sem = (int)sem_cls_label[jo];
    sums[S_HCLS] += ce_pick(nb + oH * cs + k, cs, d.NH, hcl) * lab;
    const float htar = heading_residual_label[jo] / (3.14159265358979323846f / (float)d.NH);
    sums[S_HREG] += huber1(nb[(oHR + hcl) * cs + k] - htar) * lab;
    sums[S_SCLS] += ce_pick(nb + oS * cs + k, cs, d.NS, scl) * lab;
    float sreg = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const float tar = size_residual_label[jo * 3 + t] / mean_size[scl * 3 + t];
      sreg += huber1(nb[(oSR + scl * 3 + t) * cs + k] - tar);
    }
    sums[S_SREG] += (sreg / 3.f) * lab;
    sums[S_SEM] += ce_pick(nb + oC * cs + k, cs, d.NC, sem) * lab;
  }
  __syncthreads();
  // nearest predicted centre for every GT box (chamfer direction 2)
  for (int j = tid; role == R_CENTRES && j < d.K2; j += 256) {
    const float gx = gt[j * 3], gy = gt[j * 3 + 1], gz = gt[j * 3 + 2];
    float best = 3.0e38f;
    int bi = 0;
    for (int k = 0; k < d.K; ++k) {
      const float ex = cen[k * 3] - gx, ey = cen[k * 3 + 1] - gy, ez = cen[k * 3 + 2] - gz;
      const float q = ex * ex + ey * ey + ez * ez;
      if (q < best) { best = q; bi = k; }
    }
    const float bm = box_label_mask[(size_t)b * d.K2 + j];
    k2c[(size_t)b * d.K2 + j] = bi;
    sums[S_D2C] += best * bm;
    sums[S_BOXMASK] += bm;
  }
  if (role != R_VOTES) {
  } else if (d.vote_mode == 0) {
    // vote loss (vote_factor 1): min over the 3 GT votes of the L1 distance
    for (int i = tid; i < d.S1; i += 256) {
      const size_t so = (size_t)b * d.S1 + i;
      const int pi = seed_inds[so];
      const float m = (float)vote_label_mask[(size_t)b * d.N + pi];
      const float *vl = vote_label + ((size_t)b * d.N + pi) * 9;
      const float sx = seed_xyz[so * 3], sy = seed_xyz[so * 3 + 1], sz = seed_xyz[so * 3 + 2];
      const float vx = vote_xyz[so * 3], vy = vote_xyz[so * 3 + 1], vz = vote_xyz[so * 3 + 2];
      float best = 3.0e38f;
      int bc = 0;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float q = fabsf(vx - (vl[c * 3] + sx)) + fabsf(vy - (vl[c * 3 + 1] + sy)) +
                        fabsf(vz - (vl[c * 3 + 2] + sz));
        if (q < best) { best = q; bc = c; }
      }
      vote_arg[so] = (signed char)bc;
      sums[S_VOTE] += best * m;
      sums[S_VOTEMASK] += m;
    }
  } else {
    // weak vote loss: L1 chamfer between the votes and ALL GT-centre slots (padding
    // included, like the reference), votes -> nearest centre averaged over every seed,
    // centres -> nearest vote averaged over the real boxes
    for (int i = tid; i < d.S1; i += 256) {
      const size_t so = (size_t)b * d.S1 + i;
      const float vx = vote_xyz[so * 3], vy = vote_xyz[so * 3 + 1], vz = vote_xyz[so * 3 + 2];
      float best = 3.0e38f;
      int bj = 0;
      for (int j = 0; j < d.K2; ++j) {
        const float q = fabsf(vx - gt[j * 3]) + fabsf(vy - gt[j * 3 + 1]) + fabsf(vz - gt[j * 3 + 2]);
        if (q < best) { best = q; bj = j; }
      }
      vote_arg[so] = (signed char)(unsigned char)bj;
      sums[S_VOTE] += best;
    }
    // centres -> nearest vote: `tpg` adjacent lanes share one centre and scan a slice of the
    // seeds each (one thread per centre walks all 1024 seeds alone: 115 us for 8 blocks)
    int tpg = 1;
    while (tpg < 32 && tpg * 2 * d.K2 <= 256) tpg *= 2;
    const float *v = vote_xyz + (size_t)b * d.S1 * 3;
    const int per = (d.S1 + tpg - 1) / tpg;
    for (int j0 = 0; j0 < d.K2; j0 += 256 / tpg) {
      const int j = j0 + tid / tpg, sl = tid % tpg;
      float best = 3.0e38f;
      int bi2 = 0x7fffffff;
      if (j < d.K2) {
        const float gx = gt[j * 3], gy = gt[j * 3 + 1], gz = gt[j * 3 + 2];
        const int i1 = min(d.S1, (sl + 1) * per);
        for (int i = sl * per; i < i1; ++i) {
          const float q = fabsf(v[i * 3] - gx) + fabsf(v[i * 3 + 1] - gy) + fabsf(v[i * 3 + 2] - gz);
          if (q < best) { best = q; bi2 = i; }
        }
      }
      for (int off = 1; off < tpg; off <<= 1) {  // first minimum over the slices
        const float ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi2, off);
        if (ob < best || (ob == best && oi < bi2)) { best = ob; bi2 = oi; }
      }
      if (j < d.K2 && sl == 0) {
        i2v[(size_t)b * d.K2 + j] = bi2;
        sums[S_D2V] += best * box_label_mask[(size_t)b * d.K2 + j];
      }
    }
  }
  // block reduction of the sums
#pragma unroll
  for (int i = 0; i < kNSums; ++i) {
    float v = sums[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) red[i][wave] = v;
  }
  __syncthreads();
  if (tid < kNSums && sum_owner(tid) == role)
    part[(size_t)b * kNSums + tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
}

// stats[0..13] (13: a second copy of the total for the loss tensor) = loss, vote, objectness, center, heading_cls, heading_reg, size_cls, size_reg,
// sem_cls, box, pos_ratio, neg_ratio, obj_acc;  norm[0..3] = 1/(sum+1e-6) of label, mask,
// boxmask, votemask.
__global__ void loss_reduce_kernel(LossDims d, const float *__restrict__ part,
                                   float *__restrict__ stats, float *__restrict__ norm) {
  const int B = d.B, K = d.K;
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s[kNSums];
  for (int i = 0; i < kNSums; ++i) {
    float v = 0.f;
    for (int b = 0; b < B; ++b) v += part[(size_t)b * kNSums + i];
    s[i] = v;
  }
  const float nl = 1.f / (s[S_LABEL] + 1e-6f), nm = 1.f / (s[S_MASK] + 1e-6f);
  const float nb = 1.f / (s[S_BOXMASK] + 1e-6f), nv = 1.f / (s[S_VOTEMASK] + 1e-6f);
  const float vote = d.vote_mode == 0
                         ? s[S_VOTE] * nv
                         : s[S_VOTE] / ((float)B * (float)d.S1) + s[S_D2V] * nb;
  const float obj = s[S_OBJ] * nm;
  const float center = s[S_D1C] * nl + s[S_D2C] * nb;
  const float hcls = s[S_HCLS] * nl, hreg = s[S_HREG] * nl, scls = s[S_SCLS] * nl;
  const float sreg = s[S_SREG] * nl, sem = s[S_SEM] * nl;
  const float box = center + 0.1f * hcls + hreg + 0.1f * scls + sreg;
  const float total = (float)B * (float)K;
  stats[0] = (d.w[0] * vote + d.w[1] * obj + d.w[2] * center + d.w[3] * hcls + d.w[4] * hreg +
              d.w[5] * scls + d.w[6] * sreg + d.w[7] * sem) * 10.f;
  stats[1] = vote; stats[2] = obj; stats[3] = center; stats[4] = hcls; stats[5] = hreg;
  stats[6] = scls; stats[7] = sreg; stats[8] = sem; stats[9] = box;
  stats[10] = s[S_LABEL] / total;
  stats[11] = s[S_MASK] / total - s[S_LABEL] / total;
  stats[12] = s[S_ACC] * nm;
  stats[13] = stats[0];   // the word the returned loss TENSOR lives on: `loss *= w` leaves [0] alone
  norm[0] = nl; norm[1] = nm; norm[2] = nb; norm[3] = nv;
}

__global__ __launch_bounds__(256) void loss_grad_kernel(
    LossDims d, const float *__restrict__ gout, const float *__restrict__ norm,
    const float *__restrict__ net, const float *__restrict__ agg_xyz,
    const float *__restrict__ vote_xyz, const float *__restrict__ seed_xyz,
    const int *__restrict__ seed_inds, const float *__restrict__ vote_label,
    const long long *__restrict__ vote_label_mask, const float *__restrict__ center_label,
    const float *__restrict__ box_label_mask, const long long *__restrict__ heading_class_label,
    const float *__restrict__ heading_residual_label,
    const long long *__restrict__ size_class_label, const float *__restrict__ size_residual_label,
    const long long *__restrict__ sem_cls_label, const float *__restrict__ mean_size,
    const long long *__restrict__ objectness_label, const float *__restrict__ objectness_mask,
    const long long *__restrict__ object_assignment, const int *__restrict__ j1c,
    const int *__restrict__ k2c, const signed char *__restrict__ vote_arg,
    float *__restrict__ dnet, float *__restrict__ dagg, float *__restrict__ dvote,
    const int *__restrict__ i2v) {
  __shared__ float gt[kMaxObj * 3];
  __shared__ float bm[kMaxObj];
  __shared__ int kc[kMaxObj];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int role = blockIdx.y;   // 0: the proposals' gradients, 1: the votes' (two workgroups)
  const size_t cs = (size_t)d.K;
  const float *nb = net + (size_t)b * d.Cout * d.K;
  float *gb = dnet + (size_t)b * d.Cout * d.K;
  const float g10 = 10.f * gout[0];  // d(total)/d(term) = 10 * weight * dL/d(loss)
  const float nl = norm[0], nm = norm[1], nbx = norm[2], nv = norm[3];
  for (int i = tid; i < d.K2 * 3; i += 256) gt[i] = center_label[(size_t)b * d.K2 * 3 + i];
  for (int j = tid; j < d.K2; j += 256) {
    bm[j] = box_label_mask[(size_t)b * d.K2 + j];
    kc[j] = k2c[(size_t)b * d.K2 + j];
  }
  __syncthreads();
  const int oH = 5, oHR = 5 + d.NH, oS = 5 + 2 * d.NH, oSR = oS + d.NS, oC = oS + 4 * d.NS;
  for (int k = tid; role == 0 && k < d.K; k += 256) {
    const size_t o = (size_t)b * d.K + k;
    const int label = (int)objectness_label[o];
    const float lab = (float)label, mask = objectness_mask[o];
    const int i1 = (int)object_assignment[o];
    // objectness
    {
      float mx;
      const float l = lse(nb + k, cs, 2, mx);
      const float w = (label ? 0.8f : 0.2f) * mask * nm * d.w[1] * g10;
      gb[0 * cs + k] = w * (expf(nb[k] - l) - (label == 0 ? 1.f : 0.f));
      gb[1 * cs + k] = w * (expf(nb[cs + k] - l) - (label == 1 ? 1.f : 0.f));
    }
    // centre: both chamfer directions
    const float *a = agg_xyz + o * 3;
    const float cx = a[0] + nb[2 * cs + k], cy = a[1] + nb[3 * cs + k], cz = a[2] + nb[4 * cs + k];
    const int jn = j1c[o];
    float gx = 2.f * (cx - gt[jn * 3]) * lab * nl, gy = 2.f * (cy - gt[jn * 3 + 1]) * lab * nl,
          gz = 2.f * (cz - gt[jn * 3 + 2]) * lab * nl;
    for (int j = 0; j < d.K2; ++j)
      if (kc[j] == k && bm[j] != 0.f) {
        gx += 2.f * (cx - gt[j * 3]) * bm[j] * nbx;
        gy += 2.f * (cy - gt[j * 3 + 1]) * bm[j] * nbx;
        gz += 2.f * (cz - gt[j * 3 + 2]) * bm[j] * nbx;
      }
    gx *= g10 * d.w[2]; gy *= g10 * d.w[2]; gz *= g10 * d.w[2];
    gb[2 * cs + k] = gx; gb[3 * cs + k] = gy; gb[4 * cs + k] = gz;
    dagg[o * 3] = gx; dagg[o * 3 + 1] = gy; dagg[o * 3 + 2] = gz;
    // heading / size / semantic
    const size_t jo = (size_t)b * d.K2 + i1;
    const int hcl = (int)heading_class_label[jo];
    const int scl = (int)size_class_label[jo];
    const int sem = (int)sem_cls_label[jo];
    const float wl = lab * nl * g10;
    {
      ce_grad(nb + oH * cs + k, gb + oH * cs + k, cs, d.NH, hcl, d.w[3] * wl);
      const float htar = heading_residual_label[jo] / (3.14159265358979323846f / (float)d.NH);
      const float ge = huber1_grad(nb[(oHR + hcl) * cs + k] - htar) * wl * d.w[4];
      for (int c = 0; c < d.NH; ++c) gb[(oHR + c) * cs + k] = c == hcl ? ge : 0.f;
    }
    {
      ce_grad(nb + oS * cs + k, gb + oS * cs + k, cs, d.NS, scl, d.w[5] * wl);
      for (int c = 0; c < d.NS; ++c)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          float g = 0.f;
          if (c == scl) {
            const float tar = size_residual_label[jo * 3 + t] / mean_size[scl * 3 + t];
            g = huber1_grad(nb[(oSR + c * 3 + t) * cs + k] - tar) * wl * d.w[6] / 3.f;
          }
          gb[(oSR + c * 3 + t) * cs + k] = g;
        }
    }
    ce_grad(nb + oC * cs + k, gb + oC * cs + k, cs, d.NC, sem, d.w[7] * wl);
  }
  // votes
  if (role != 1) {
  } else if (d.vote_mode == 0) {
    for (int i = tid; i < d.S1; i += 256) {
      const size_t so = (size_t)b * d.S1 + i;
      const int pi = seed_inds[so];
      const float m = (float)vote_label_mask[(size_t)b * d.N + pi] * nv * g10 * d.w[0];
      const int c = vote_arg[so];
      const float *vl = vote_label + ((size_t)b * d.N + pi) * 9 + c * 3;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const float e = vote_xyz[so * 3 + t] - (vl[t] + seed_xyz[so * 3 + t]);
        dvote[so * 3 + t] = m * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
      }
    }
  } else {
    const float w1 = g10 * d.w[0] / ((float)d.B * (float)d.S1), w2 = g10 * d.w[0] * nbx;
    for (int i = tid; i < d.S1; i += 256) {
      const size_t so = (size_t)b * d.S1 + i;
      const int j1 = (unsigned char)vote_arg[so];
      float g[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const float e = vote_xyz[so * 3 + t] - gt[j1 * 3 + t];
        g[t] = w1 * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
      }
      for (int j = 0; j < d.K2; ++j)
        if (i2v[(size_t)b * d.K2 + j] == i && bm[j] != 0.f) {
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const float e = vote_xyz[so * 3 + t] - gt[j * 3 + t];
            g[t] += w2 * bm[j] * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
          }
        }
      dvote[so * 3] = g[0]; dvote[so * 3 + 1] = g[1]; dvote[so * 3 + 2] = g[2];
    }
  }
}

// ---------------------------------------------------------------- the domain-adaptation term
// reference: detection/Votenet/models/loss_helper.py:618-650 (get_loss_DA) and :466-545
// (FocalLoss, alpha = 1) -- per branch
//   coef * mean_{b,k}( e(l)^2 * w ) + coef * mean_b( -(1 - p_t)^gamma * log p_t ),
// l = local_d_pred (b, 1, k) (sigmoid outputs), w = objectness_label, e(l) = l for the source
// branch (domain 0) and 1 - l for the target (domain 1), p = softmax(global_d_pred (b, 2)),
// t = the branch's domain; coef = 0.5 (VoteNet's da_coefficient) or 1 (GroupFree3D's
// loss_helper.py:673-712, the same terms unweighted).  ~25 element-wise / reduction launches forward and ~35 backward as
// torch ops; here ONE workgroup computes the value and the gradient for a unit upstream
// gradient (grads: [d gS (2b) | d lS (b k) | d gT (2b) | d lT (b k)]), the backward scales them.
__global__ __launch_bounds__(256) void domain_loss_kernel(
    int b, int k, float gamma, float coef, const float *__restrict__ gS, const float *__restrict__ lS,
    const long long *__restrict__ wS, const float *__restrict__ gT, const float *__restrict__ lT,
    const long long *__restrict__ wT, float *__restrict__ out, float *__restrict__ grads) {
  __shared__ float red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bk = b * k;
  const float inv_bk = 1.f / (float)bk, inv_b = 1.f / (float)b;
  float *dgS = grads, *dlS = dgS + 2 * b, *dgT = dlS + bk, *dlT = dgT + 2 * b;
  // ---- local terms: 0.5 * mean(l^2 w) and 0.5 * mean((1 - l)^2 w)
  float accS = 0.f, accT = 0.f;
  for (int i = tid; i < bk; i += 256) {
    const float ws = (float)wS[i], wt = (float)wT[i];
    const float ls = lS[i], et = 1.f - lT[i];
    accS += ls * ls * ws;
    accT += et * et * wt;
    dlS[i] = 2.f * coef * ls * ws * inv_bk;      // coef * 2 l w / (b k)
    dlT[i] = -2.f * coef * et * wt * inv_bk;     // coef * 2 (1 - l) (-1) w / (b k)
  }
  // ---- global terms: focal loss on the two-way softmax of each scene
  float focS = 0.f, focT = 0.f;
  for (int i = tid; i < 2 * b; i += 256) {
    const int bi = i >> 1, branch = i & 1;   // branch 0: source (domain 0), 1: target (domain 1)
    const float *g = (branch ? gT : gS) + 2 * bi;
    const float m = fmaxf(g[0], g[1]);
    const float e0 = expf(g[0] - m), e1 = expf(g[1] - m);
    const float inv = 1.f / (e0 + e1);
    const float p0 = e0 * inv, p1 = e1 * inv;
    const float pt = branch ? p1 : p0;
    const float omp = 1.f - pt;
    const float lg = logf(pt);
    const float f = -powf(omp, gamma) * lg;
    // d f / d p_t = gamma (1 - p)^(gamma - 1) log p - (1 - p)^gamma / p
    const float dfdp = gamma * powf(omp, gamma - 1.f) * lg - powf(omp, gamma) / pt;
    // d p_t / d g_j = p_t (delta_tj - p_j)
    const float c = coef * inv_b * dfdp * pt;
    float *dg = (branch ? dgT : dgS) + 2 * bi;
    dg[0] = c * ((branch ? 0.f : 1.f) - p0);
    dg[1] = c * ((branch ? 1.f : 0.f) - p1);
    if (branch) focT += f; else focS += f;
  }
  float vS = coef * (accS * inv_bk + focS * inv_b);
  float vT = coef * (accT * inv_bk + focT * inv_b);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    vS += __shfl_xor(vS, off);
    vT += __shfl_xor(vT, off);
  }
  if (lane == 0) {
    red[0][wave] = vS;
    red[1][wave] = vT;
  }
  __syncthreads();
  if (tid == 0) {
    const float s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const float t = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    out[0] = s + t;
    out[1] = s;
    out[2] = t;
  }
}

}  // namespace btr

using namespace btr;

extern "C" {

// The domain-adaptation loss of a Back-to-Reality step and its gradient for a unit upstream
// gradient: out[3] = (total, source part, target part); grads (4 b + 2 b k floats) =
// [d global_S (b,2) | d local_S (b,k) | d global_T (b,2) | d local_T (b,k)].
int btr_domain_loss(int b, int k, float gamma, float coef, const float *global_S, const float *local_S,
                    const long long *label_S, const float *global_T, const float *local_T,
                    const long long *label_T, float *out, float *grads, btr_stream_t stream) {
  if (b <= 0 || k <= 0) return BTR_OK;
  BTR_REQUIRE(global_S && local_S && label_S && global_T && local_T && label_T && out && grads,
              "domain_loss: null pointer");
  hipLaunchKernelGGL(domain_loss_kernel, dim3(1), dim3(256), 0, as_stream(stream), b, k, gamma,
                     coef, global_S, local_S, label_S, global_T, local_T, label_T, out, grads);
  return check_launch("domain_loss");
}

// Forward of the VoteNet loss.  All label tensors as the reference's batch dict
// (scannet_detection_dataset.py:197-219); `net` is the raw proposal-head output (b, cout, k).
// Workspace outputs (kept for the backward): j1c (b,k) i32, k2c (b,k2) i32, vote_arg (b,s1) i8,
// part (b,16) f32, norm (4) f32.  stats (13) f32: see loss_reduce_kernel.
int btr_votenet_loss_fwd(int b, int k, int k2, int nh, int ns, int nc, int s1, int n, int cout,
                         const float *net, const float *agg_xyz, const float *vote_xyz,
                         const float *seed_xyz, const int *seed_inds, const float *vote_label,
                         const long long *vote_label_mask, const float *center_label,
                         const float *box_label_mask, const long long *heading_class_label,
                         const float *heading_residual_label, const long long *size_class_label,
                         const float *size_residual_label, const long long *sem_cls_label,
                         const float *mean_size, long long *objectness_label,
                         float *objectness_mask, long long *object_assignment, int *j1c, int *k2c,
                         signed char *vote_arg, float *part, float *stats, float *norm,
                         const float *weights8, int vote_mode, int *i2v, btr_stream_t stream) {
  if (b <= 0) return BTR_OK;
  BTR_REQUIRE(k > 0 && k <= kMaxProp && k2 > 0 && k2 <= kMaxObj &&
                  cout == 5 + 2 * nh + 4 * ns + nc,
              "votenet_loss: unsupported sizes (k=%d k2=%d cout=%d)", k, k2, cout);
  BTR_REQUIRE(weights8 && (vote_mode == 0 || i2v), "votenet_loss: weights / i2v missing");
  LossDims d{b, k, k2, nh, ns, nc, s1, n, cout, {0, 0, 0, 0, 0, 0, 0, 0}, vote_mode};
  for (int i = 0; i < 8; ++i) d.w[i] = weights8[i];
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(loss_terms_kernel, dim3(b, kLossRoles), dim3(256), 0, st, d, net, agg_xyz, vote_xyz,
                     seed_xyz, seed_inds, vote_label, vote_label_mask, center_label,
                     box_label_mask, heading_class_label, heading_residual_label,
                     size_class_label, size_residual_label, sem_cls_label, mean_size,
                     objectness_label, objectness_mask, object_assignment, j1c, k2c, vote_arg,
                     part, i2v);
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(64), 0, st, d, part, stats, norm);
  return check_launch("votenet_loss_fwd");
}

// Backward: dnet (b,cout,k), dagg (b,k,3), dvote (b,s1,3) <- gradients scaled by gout[0].
int btr_votenet_loss_bwd(int b, int k, int k2, int nh, int ns, int nc, int s1, int n, int cout,
                         const float *gout, const float *norm, const float *net,
                         const float *agg_xyz, const float *vote_xyz, const float *seed_xyz,
                         const int *seed_inds, const float *vote_label,
                         const long long *vote_label_mask, const float *center_label,
                         const float *box_label_mask, const long long *heading_class_label,
                         const float *heading_residual_label, const long long *size_class_label,
                         const float *size_residual_label, const long long *sem_cls_label,
                         const float *mean_size, const long long *objectness_label,
                         const float *objectness_mask, const long long *object_assignment,
                         const int *j1c, const int *k2c, const signed char *vote_arg, float *dnet,
                         float *dagg, float *dvote, const float *weights8, int vote_mode,
                         const int *i2v, btr_stream_t stream) {
  if (b <= 0) return BTR_OK;
  BTR_REQUIRE(weights8 && (vote_mode == 0 || i2v), "votenet_loss: weights / i2v missing");
  LossDims d{b, k, k2, nh, ns, nc, s1, n, cout, {0, 0, 0, 0, 0, 0, 0, 0}, vote_mode};
  for (int i = 0; i < 8; ++i) d.w[i] = weights8[i];
  hipLaunchKernelGGL(loss_grad_kernel, dim3(b, 2), dim3(256), 0, as_stream(stream), d, gout, norm,
                     net, agg_xyz, vote_xyz, seed_xyz, seed_inds, vote_label, vote_label_mask,
                     center_label, box_label_mask, heading_class_label, heading_residual_label,
                     size_class_label, size_residual_label, sem_cls_label, mean_size,
                     objectness_label, objectness_mask, object_assignment, j1c, k2c, vote_arg,
                     dnet, dagg, dvote, i2v);
  return check_launch("votenet_loss_bwd");
}

}  // extern "C"
