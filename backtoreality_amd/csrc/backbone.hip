// backbone.hip -- whole-backbone entry points: the PointNet++ encoder-decoder of VoteNet /
// GroupFree3D (set-abstraction levels + feature-propagation modules) in three calls:
//   btr_backbone_sampling   everything that depends on coordinates only (FPS pyramid, centre
//                           coordinates, ball queries, 3-NN blend weights) -- may run on a side
//                           stream, a whole training step ahead
//   btr_backbone_forward    SA1 .. SA_L, FP1 .. FP_F
//   btr_backbone_backward   their backward, including the gradient accumulation at the skip
//                           connections
// over five caller-provided arenas (geometry / outputs / saved / scratch / flat gradients) laid
// out by btr_backbone_plan.  Nothing here computes: the functions SEQUENCE the whole-layer calls
// of sa_layer.hip (which sequence the kernels of sa_mlp.hip) plus three small glue kernels.
// Why: issued layer by layer from Python (one autograd node, a dozen allocator calls and
// tensor views per layer) the host needs ~5 ms per step for what the GPU runs in ~4.5 ms.
//
// reference: models/backbone_module.py:83-133 (Pointnet2Backbone.forward), the SA / FP modules
// it calls (pointnet2/pointnet2_modules.py:210-272, 469-514) and their autograd backward.
#include <algorithm>
#include <cstring>

#include "internal.hpp"

namespace btr {
namespace {

constexpr int kMaxLv = BTR_MAX_LEVELS;
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
  size_t ints(size_t n) { return take(n * sizeof(int)); }
};
inline float *at_f(void *base, size_t off) { return (float *)((char *)base + off); }
inline const float *at_f(const void *base, size_t off) {
  return (const float *)((const char *)base + off);
}
inline int *at_i(void *base, size_t off) { return (int *)((char *)base + off); }
inline const int *at_i(const void *base, size_t off) {
  return (const int *)((const char *)base + off);
}

#define BTR_TRY(call)              \
  do {                             \
    const int rc_ = (call);        \
    if (rc_ != BTR_OK) return rc_; \
  } while (0)

// xyz (B, N, 3) and feats (B, N, C) from the cloud (B, N, 3 + C)
__global__ __launch_bounds__(256) void split_cloud_kernel(long long rows, int c,
                                                          const float *__restrict__ cloud,
                                                          float *__restrict__ xyz,
                                                          float *__restrict__ feat) {
  const int w = 3 + c;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * w) return;
  const long long r = t / w;
  const int k = (int)(t - r * w);
  const float v = cloud[t];
  if (k < 3) xyz[r * 3 + k] = v;
  else feat[r * c + (k - 3)] = v;
}

// Input rows of a feature-propagation MLP, channel-last (pointnet2_modules.py:489-506:
// three_interpolate of the known level's features, concatenated with the skip features):
//   X[r][c]      = p1*w1 + p2*w2 + p3*w3 over the 3 nearest known points    c <  c1
//   X[r][c1 + c] = skip[r][c]                                               c <  c2
// (the blend is the expression of three_interpolate_kernel, so the rows equal what
// three_interpolate + cat + a transpose produce).  One thread = 4 channels of one row.
__global__ __launch_bounds__(256) void fp_concat_kernel(
    int n, int m, int c1, int c2, const float *__restrict__ known_cl,
    const int *__restrict__ idx, const float *__restrict__ weight,
    const float *__restrict__ skip_cl, float *__restrict__ X, long long rows) {
  const int q = (c1 + c2) >> 2;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * q) return;
  const long long r = t / q;
  const int c = (int)(t - r * q) * 4;
  float4 v;
  if (c < c1) {
    const long long bi = r / n;
    const int *id = idx + r * 3;
    const float *w = weight + r * 3;
    const float w1 = w[0], w2 = w[1], w3 = w[2];
    const float *k = known_cl + (size_t)bi * m * c1 + c;
    const float4 p1 = *reinterpret_cast<const float4 *>(k + (size_t)id[0] * c1);
    const float4 p2 = *reinterpret_cast<const float4 *>(k + (size_t)id[1] * c1);
    const float4 p3 = *reinterpret_cast<const float4 *>(k + (size_t)id[2] * c1);
    v = make_float4(dot3(p1.x, w1, p2.x, w2, p3.x, w3), dot3(p1.y, w1, p2.y, w2, p3.y, w3),
                    dot3(p1.z, w1, p2.z, w2, p3.z, w3), dot3(p1.w, w1, p2.w, w2, p3.w, w3));
  } else {
    v = *reinterpret_cast<const float4 *>(skip_cl + (size_t)r * c2 + (c - c1));
  }
  *reinterpret_cast<float4 *>(X + (size_t)r * (c1 + c2) + c) = v;
}

__global__ __launch_bounds__(256) void add_inplace_kernel(long long n4, float *__restrict__ acc,
                                                          const float *__restrict__ src) {
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n4;
       t += (long long)gridDim.x * 256) {
    float4 a = reinterpret_cast<float4 *>(acc)[t];
    const float4 s = reinterpret_cast<const float4 *>(src)[t];
    a.x += s.x; a.y += s.y; a.z += s.z; a.w += s.w;
    reinterpret_cast<float4 *>(acc)[t] = a;
  }
}
int add_inplace(float *acc, const float *src, long long n, hipStream_t st) {
  if (n <= 0) return BTR_OK;
  BTR_REQUIRE(n % 4 == 0, "backbone: gradient size %lld not a multiple of 4", n);
  hipLaunchKernelGGL(add_inplace_kernel, dim3((int)std::min<long long>(cdiv(n / 4, 256), 2048)),
                     dim3(256), 0, st, n / 4, acc, src);
  return check_launch("backbone add");
}

// Events of the sequential mode of btr_backbone_sampling (levels >= 2 on a side stream while
// the caller's stream goes on with SA1): per host thread and device, like the weight-gradient
// stream of sa_layer.hip.
struct LevelEvents {
  hipEvent_t fork = nullptr, level[kMaxLv + 1] = {};
  bool ok = false;
};
LevelEvents *level_events() {
  static thread_local LevelEvents ctx[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  LevelEvents &e = ctx[dev];
  if (!e.ok) {
    if (hipEventCreateWithFlags(&e.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
    for (int i = 0; i <= kMaxLv; ++i)
      if (hipEventCreateWithFlags(&e.level[i], hipEventDisableTiming) != hipSuccess)
        return nullptr;
    e.ok = true;
  }
  return &e;
}

// sizes of the levels: points going in (n), centres (m), channels out (c)
struct Dims {
  int n[kMaxLv + 1], c[kMaxLv + 1];   // [0] = the cloud; [l] = after SA l
};
Dims dims_of(const btr_backbone_t &d) {
  Dims s{};
  s.n[0] = d.n;
  s.c[0] = d.c;
  for (int l = 0; l < d.levels; ++l) {
    s.n[l + 1] = d.sa[l].m;
    s.c[l + 1] = d.sa[l].width[d.sa[l].layers - 1];
  }
  return s;
}
// feature-propagation module j: known level L - j (features: SA_L's or FP_{j-1}'s output),
// unknown level L - j - 1 (skip features: SA_{L-j-1}'s output)
inline int fp_known(const btr_backbone_t &d, int j) { return d.levels - j; }
inline int fp_unknown(const btr_backbone_t &d, int j) { return d.levels - j - 1; }
inline int fp_known_c(const btr_backbone_t &d, const Dims &s, int j) {
  return j == 0 ? s.c[d.levels] : d.fp[j - 1].width[d.fp[j - 1].layers - 1];
}

// the prepared compact plan / scatter lists of level l inside the geometry arena
SaGeom level_geom(const btr_backbone_plan_t &p, const void *geom, int l) {
  SaGeom g;
  char *base = (char *)const_cast<void *>(geom);
  if (p.sa[l].compact) {
    g.goff = (int *)(base + p.g_goff[l]);
    g.dims = (int *)(base + p.g_dims[l]);
    g.cidx = (int *)(base + p.g_cidx[l]);
    g.bgrp = (int *)(base + p.g_bgrp[l]);
    g.bw = (float *)(base + p.g_bw[l]);
  }
  if (p.g_scat_bytes[l]) g.scatter_ws = base + p.g_scat[l];
  return g;
}

struct BwdScratch {
  size_t dx[kMaxLv], dk[kMaxLv], df[kMaxLv + 1], ti, bytes;
  // every layer call its OWN scratch (not the largest one shared): the split-K partials of all
  // weight gradients stay alive until the one reduction launch at the end of the call
  size_t layer_fp[kMaxLv], layer_sa[kMaxLv];
  size_t ti_bytes;
};
BwdScratch bwd_scratch(const btr_backbone_t &d, const btr_backbone_plan_t &p) {
  BwdScratch s{};
  const Dims dm = dims_of(d);
  Bump b;
  size_t ti = 0;
  for (int j = 0; j < d.fps; ++j) {
    const int u = fp_unknown(d, j), k = fp_known(d, j);
    s.dx[j] = b.floats((size_t)d.b * d.fp[j].c * dm.n[u]);
    s.dk[j] = b.floats((size_t)d.b * fp_known_c(d, dm, j) * dm.n[k]);
    ti = std::max(ti, ti_grad_workspace_bytes(d.b, dm.n[u], dm.n[k]));
  }
  for (int l = 1; l < d.levels; ++l) s.df[l] = b.floats((size_t)d.b * dm.c[l] * dm.n[l]);
  s.ti_bytes = ti;
  s.ti = b.take(ti);
  for (int j = 0; j < d.fps; ++j) s.layer_fp[j] = b.take(p.fp[j].bwd_scratch_bytes);
  for (int l = 0; l < d.levels; ++l) s.layer_sa[l] = b.take(p.sa[l].bwd_scratch_bytes);
  s.bytes = b.off;
  return s;
}

}  // namespace
}  // namespace btr

using namespace btr;

extern "C" {

int btr_backbone_plan(const btr_backbone_t *dp, btr_backbone_plan_t *p) {
  BTR_REQUIRE(dp && p, "backbone_plan: null pointer");
  const btr_backbone_t &d = *dp;
  BTR_REQUIRE(d.levels >= 1 && d.levels <= kMaxLv && d.fps >= 0 && d.fps < d.levels,
              "backbone_plan: %d levels, %d feature-propagation modules", d.levels, d.fps);
  BTR_REQUIRE(d.b > 0 && d.n > 0 && d.c >= 0, "backbone_plan: bad cloud (b=%d n=%d c=%d)", d.b,
              d.n, d.c);
  std::memset(p, 0, sizeof(*p));
  const Dims dm = dims_of(d);
  for (int l = 0; l < d.levels; ++l) {
    const btr_sa_layer_t &s = d.sa[l];
    BTR_REQUIRE(s.b == d.b && s.n == dm.n[l] && s.c == dm.c[l],
                "backbone_plan: level %d does not continue level %d (n=%d c=%d)", l + 1, l, s.n,
                s.c);
    BTR_REQUIRE(!s.need_dxyz && !s.need_dnew_xyz && (l > 0) == (s.need_dfeat != 0),
                "backbone_plan: level %d: feature gradients for every level but the first, no "
                "coordinate gradients", l + 1);
    BTR_TRY(btr_sa_layer_plan(&s, &p->sa[l]));
  }
  for (int j = 0; j < d.fps; ++j) {
    const btr_pm_chain_t &f = d.fp[j];
    const int u = fp_unknown(d, j);
    const int c1 = fp_known_c(d, dm, j), c2 = dm.c[u];
    BTR_REQUIRE(u >= 1, "backbone_plan: propagation onto the input cloud is not covered");
    BTR_REQUIRE(f.b == d.b && f.n == dm.n[u] && f.c == c1 + c2 && c1 % 4 == 0 && c2 % 4 == 0 &&
                    f.need_dx,
                "backbone_plan: feature-propagation module %d does not fit (n=%d c=%d)", j, f.n,
                f.c);
    BTR_REQUIRE(f.has_bn[f.layers - 1] && f.width[f.layers - 1] % 4 == 0,
                "backbone_plan: feature-propagation MLPs end in BatchNorm + ReLU");
    BTR_TRY(btr_pm_chain_plan(&f, &p->fp[j]));
  }
  {  // geometry arena
    Bump g;
    if (d.c > 0) {
      p->g_xyz = g.floats((size_t)d.b * d.n * 3);
      p->g_feat = g.floats((size_t)d.b * d.n * d.c);
    }
    size_t bq = 0;
    for (int l = 0; l < d.levels; ++l) {
      const btr_sa_layer_t &s = d.sa[l];
      p->g_inds[l] = g.ints((size_t)d.b * s.m);
      p->g_new_xyz[l] = g.floats((size_t)d.b * s.m * 3);
      p->g_idx[l] = g.ints((size_t)d.b * s.m * s.s);
      p->g_fps_ws_bytes[l] = btr_furthest_point_sampling_workspace_bytes(d.b, s.n, s.m);
      p->g_fps_ws[l] = g.take(p->g_fps_ws_bytes[l]);
      // the streaming fall-back (no bucket workspace for a large scene) needs the (b, n) scratch
      p->g_fps_temp[l] = (p->g_fps_ws_bytes[l] == 0 && s.n > 4096) ? g.floats((size_t)d.b * s.n)
                                                                    : (size_t)0;
      // levels 2.. sample the previous level's points in FPS order: the verdict slots of
      // btr_furthest_point_sampling_ordered live in the same scratch slot
      if (l >= 1 && btr_fps_ordered_scratch_bytes(d.b, s.n, s.m))
        p->g_fps_temp[l] = g.floats(btr_fps_ordered_scratch_bytes(d.b, s.n, s.m) / sizeof(float));
      const size_t bk = p->g_fps_ws_bytes[l]
                            ? btr_ball_query_buckets_workspace_bytes(d.b, s.n, s.m, s.s) : 0;
      p->bq_buckets[l] = bk > 0;
      bq = std::max(bq, bk ? bk : btr_ball_query_workspace_bytes(d.b, s.n, s.m, s.s));
      if (p->sa[l].compact) {   // the compact-row plan (csrc/sa_mlp.hip "compact rows")
        const size_t groups = (size_t)d.b * s.m, rows = (size_t)p->sa[l].rows;
        p->g_goff[l] = g.ints(groups + 1);
        p->g_dims[l] = g.ints(2);
        p->g_cidx[l] = g.ints(rows);
        p->g_bgrp[l] = g.ints(rows / 8);
        p->g_bw[l] = g.floats(rows / 8);
        p->g_len[l] = g.ints(groups);
      }
      if (s.need_dfeat && s.c > 0) {   // the backward's inverted neighbour lists
        p->g_scat_bytes[l] = p->sa[l].compact
                                 ? btr_sac_scatter_workspace_bytes(d.b, s.n, p->sa[l].rows)
                                 : btr_sa_scatter_workspace_bytes(d.b, s.n, s.m, s.s);
        p->g_scat[l] = g.take(p->g_scat_bytes[l]);
      }
    }
    for (int j = 0; j < d.fps; ++j) {
      const size_t e = (size_t)d.b * dm.n[fp_unknown(d, j)] * 3;
      p->g_nn_idx[j] = g.ints(e);
      p->g_nn_w[j] = g.floats(e);
      p->g_ti_bytes[j] = ti_grad_workspace_bytes(d.b, dm.n[fp_unknown(d, j)], dm.n[fp_known(d, j)]);
      p->g_ti[j] = g.take(p->g_ti_bytes[j]);
      bq = std::max(bq, e * sizeof(float));   // dist2 of three_nn (not kept)
    }
    p->g_ws_bytes = bq;
    p->g_ws = g.take(bq);
    p->geom_bytes = g.off;
  }
  {  // outputs
    Bump o;
    for (int l = 0; l < d.levels; ++l) {
      const size_t e = (size_t)d.b * dm.c[l + 1] * dm.n[l + 1];
      p->o_sa[l] = o.floats(e);
      p->o_sa_cl[l] = o.floats(e);
    }
    for (int j = 0; j < d.fps; ++j) {
      const size_t e = (size_t)d.b * d.fp[j].width[d.fp[j].layers - 1] * d.fp[j].n;
      p->o_fp[j] = o.floats(e);
      p->o_fp_cl[j] = o.floats(e);
    }
    p->out_bytes = o.off;
  }
  {  // saved for the backward + forward scratch + gradients
    Bump s;
    size_t scratch = 0, g = 0;
    for (int l = 0; l < d.levels; ++l) {
      p->s_sa[l] = s.take(p->sa[l].saved_bytes);
      scratch = std::max(scratch, p->sa[l].fwd_scratch_bytes);
      p->gr_sa[l] = g;
      g += p->sa[l].grads_floats;
    }
    for (int j = 0; j < d.fps; ++j) {
      p->s_fpx[j] = s.floats((size_t)d.b * d.fp[j].n * d.fp[j].c);
      p->s_fp[j] = s.take(p->fp[j].saved_bytes);
      scratch = std::max(scratch, p->fp[j].fwd_scratch_bytes);
      p->gr_fp[j] = g;
      g += p->fp[j].grads_floats;
    }
    p->saved_bytes = s.off;
    p->fwd_scratch_bytes = up(scratch);
    p->grads_floats = g;
  }
  p->bwd_scratch_bytes = bwd_scratch(d, *p).bytes;
  return BTR_OK;
}

int btr_backbone_sampling(const btr_backbone_t *dp, const btr_backbone_plan_t *pp,
                          const float *cloud, void *geom, btr_stream_t stream,
                          btr_stream_t side) {
  BTR_REQUIRE(dp && pp && cloud && geom, "backbone_sampling: null pointer");
  const btr_backbone_t &d = *dp;
  const btr_backbone_plan_t &p = *pp;
  const Dims dm = dims_of(d);
  hipStream_t st = as_stream(stream);
  const float *xyz = cloud;
  if (d.c > 0) {
    const long long rows = (long long)d.b * d.n;
    hipLaunchKernelGGL(split_cloud_kernel, dim3(cdiv(rows * (3 + d.c), 256)), dim3(256), 0, st,
                       rows, d.c, cloud, at_f(geom, p.g_xyz), at_f(geom, p.g_feat));
    xyz = at_f(geom, p.g_xyz);
  }
  LevelEvents *ev = nullptr;
  if (side && side != stream) {
    ev = level_events();
    BTR_REQUIRE(ev, "backbone_sampling: could not create events");
  }
  btr_stream_t cur = stream;
  void *ws = (char *)geom + p.g_ws;
  for (int l = 0; l < d.levels; ++l) {
    const btr_sa_layer_t &s = d.sa[l];
    int *inds = at_i(geom, p.g_inds[l]);
    float *new_xyz = at_f(geom, p.g_new_xyz[l]);
    void *fws = p.g_fps_ws_bytes[l] ? (char *)geom + p.g_fps_ws[l] : nullptr;
    float *temp = p.g_fps_temp[l] ? at_f(geom, p.g_fps_temp[l]) : nullptr;
    const size_t ordered = l >= 1 ? btr_fps_ordered_scratch_bytes(d.b, s.n, s.m) : 0;
    if (ordered && temp)   // xyz = the previous level's new_xyz: expect 0, 1, 2, ... and check it
      BTR_TRY(btr_furthest_point_sampling_ordered(d.b, s.n, s.m, xyz, nullptr, inds, 0, temp,
                                                  ordered, cur));
    else
      BTR_TRY(btr_furthest_point_sampling_ws(d.b, s.n, s.m, xyz, temp, inds, 0, fws,
                                             p.g_fps_ws_bytes[l], cur));
    BTR_TRY(btr_gather_rows(d.b, s.n, s.m, 3, xyz, inds, new_xyz, cur));
    if (p.bq_buckets[l])
      BTR_TRY(btr_ball_query_buckets(d.b, s.n, s.m, d.radius[l], s.s, new_xyz, fws,
                                     at_i(geom, p.g_idx[l]), ws, p.g_ws_bytes, cur));
    else
      BTR_TRY(btr_ball_query_ws(d.b, s.n, s.m, d.radius[l], s.s, new_xyz, xyz,
                                at_i(geom, p.g_idx[l]), ws, p.g_ws_bytes, cur));
    // what forward / backward derive from the ball query alone, while nobody waits for it
    if (p.sa[l].compact)
      BTR_TRY(btr_sac_plan(d.b * s.m, s.s, at_i(geom, p.g_idx[l]), at_i(geom, p.g_len[l]),
                           at_i(geom, p.g_goff[l]), at_i(geom, p.g_dims[l]),
                           at_i(geom, p.g_cidx[l]), at_i(geom, p.g_bgrp[l]),
                           at_f(geom, p.g_bw[l]), cur));
    if (p.g_scat_bytes[l]) {
      void *sw = (char *)geom + p.g_scat[l];
      if (p.sa[l].compact)
        BTR_TRY(sac_scatter_ex(d.b, s.n, s.m, s.c, p.sa[l].k0p, s.use_xyz, nullptr,
                               at_i(geom, p.g_cidx[l]), at_i(geom, p.g_goff[l]), nullptr, sw,
                               p.g_scat_bytes[l], p.sa[l].rows, kScatterBuild, as_stream(cur)));
      else
        BTR_TRY(sa_scatter_ex(d.b, s.n, s.m, s.s, s.c, p.sa[l].k0p, s.use_xyz, s.radius_div,
                              nullptr, at_i(geom, p.g_idx[l]), nullptr, nullptr, nullptr, sw,
                              p.g_scat_bytes[l], kScatterBuild, as_stream(cur)));
    }
    if (ev) {
      if (l == 0) {   // level 1 stays on the caller's stream; the rest forks to the side stream
        (void)hipEventRecord(ev->fork, as_stream(stream));
        (void)hipStreamWaitEvent(as_stream(side), ev->fork, 0);
        cur = side;
      } else {
        (void)hipEventRecord(ev->level[l], as_stream(side));
      }
    }
    xyz = new_xyz;
  }
  for (int j = 0; j < d.fps; ++j) {
    const int u = fp_unknown(d, j), k = fp_known(d, j);
    BTR_TRY(btr_three_nn_weights(d.b, dm.n[u], dm.n[k], at_f(geom, p.g_new_xyz[u - 1]),
                                 at_f(geom, p.g_new_xyz[k - 1]), (float *)ws,
                                 at_i(geom, p.g_nn_idx[j]), at_f(geom, p.g_nn_w[j]), cur));
    BTR_TRY(ti_grad_lists(d.b, 0, dm.n[u], dm.n[k], nullptr, 0, at_i(geom, p.g_nn_idx[j]),
                          nullptr, nullptr, (char *)geom + p.g_ti[j], p.g_ti_bytes[j],
                          as_stream(cur), kScatterBuild));
  }
  if (ev) (void)hipEventRecord(ev->level[d.levels], as_stream(side));   // the 3-NN weights
  return check_launch("backbone_sampling");
}

// btr_backbone_fork_event (header): the event the NEXT btr_backbone_forward of this host thread
// records on its stream once set-abstraction level `level` (1-based) has been issued
namespace {
struct ForkEvent {
  hipEvent_t ev = nullptr;
  int level = 0;
};
inline ForkEvent &fork_event() {
  static thread_local ForkEvent f;
  return f;
}
}  // namespace

void btr_backbone_fork_event(void *event, int level) {
  fork_event().ev = (hipEvent_t)event;
  fork_event().level = level;
}

int btr_backbone_forward(const btr_backbone_t *dp, const btr_backbone_plan_t *pp,
                         const float *cloud, const void *geom, void *out, void *saved,
                         void *scratch, int wait_side, btr_stream_t stream) {
  BTR_REQUIRE(dp && pp && cloud && geom && out && saved && scratch,
              "backbone_forward: null pointer");
  const btr_backbone_t &d = *dp;
  const btr_backbone_plan_t &p = *pp;
  const Dims dm = dims_of(d);
  hipStream_t st = as_stream(stream);
  LevelEvents *ev = nullptr;
  if (wait_side) {
    ev = level_events();
    BTR_REQUIRE(ev, "backbone_forward: no events");
  }
  const float *xyz = d.c > 0 ? at_f(geom, p.g_xyz) : cloud;
  const float *feats = d.c > 0 ? at_f(geom, p.g_feat) : nullptr;
  // ---- the weight preparation of all levels and modules as ONE launch: a collecting pass over
  // the same calls (they add their layers and return), then the launch, then the real pass
  // (instead of six launches, one per call)
  constexpr bool prep_batched = true;
  struct PrepScope {
    bool on;
    ~PrepScope() { if (on) prep_batch_end(); }
  } prep_scope{prep_batched};
  if (prep_batched) {
    prep_batch_begin();
    const float *cx = xyz, *cf = feats;
    for (int l = 0; l < d.levels; ++l) {
      const SaGeom sg = level_geom(p, geom, l);
      BTR_TRY(sa_layer_forward_geom(&d.sa[l], &p.sa[l], cx, at_f(geom, p.g_new_xyz[l]), cf,
                                    at_i(geom, p.g_idx[l]), at_f(out, p.o_sa[l]),
                                    at_f(out, p.o_sa_cl[l]), (char *)saved + p.s_sa[l], scratch,
                                    &sg, stream));
      cx = at_f(geom, p.g_new_xyz[l]);
      cf = at_f(out, p.o_sa_cl[l]);
    }
    for (int j = 0; j < d.fps; ++j)
      BTR_TRY(btr_pm_chain_forward(&d.fp[j], &p.fp[j], nullptr, at_f(saved, p.s_fpx[j]),
                                   at_f(out, p.o_fp[j]), at_f(out, p.o_fp_cl[j]),
                                   (char *)saved + p.s_fp[j], scratch, stream));
    prep_batch_launch(st);
  }
  for (int l = 0; l < d.levels; ++l) {
    if (ev && l > 0) (void)hipStreamWaitEvent(st, ev->level[l], 0);
    const float *new_xyz = at_f(geom, p.g_new_xyz[l]);
    const SaGeom sg = level_geom(p, geom, l);
    BTR_TRY(sa_layer_forward_geom(&d.sa[l], &p.sa[l], xyz, new_xyz, feats,
                                  at_i(geom, p.g_idx[l]), at_f(out, p.o_sa[l]),
                                  at_f(out, p.o_sa_cl[l]), (char *)saved + p.s_sa[l], scratch,
                                  &sg, stream));
    xyz = new_xyz;
    feats = at_f(out, p.o_sa_cl[l]);
    if (fork_event().ev && (l + 1 == fork_event().level || l + 1 == d.levels)) {
      (void)hipEventRecord(fork_event().ev, st);   // (at the last level at the latest)
      fork_event().ev = nullptr;
    }
  }
  if (ev && d.fps > 0) (void)hipStreamWaitEvent(st, ev->level[d.levels], 0);
  for (int j = 0; j < d.fps; ++j) {
    const int u = fp_unknown(d, j), k = fp_known(d, j);
    const int c1 = fp_known_c(d, dm, j), c2 = dm.c[u];
    const float *known_cl = j == 0 ? at_f(out, p.o_sa_cl[d.levels - 1])
                                   : at_f(out, p.o_fp_cl[j - 1]);
    float *x = at_f(saved, p.s_fpx[j]);
    const long long rows = (long long)d.b * dm.n[u];
    hipLaunchKernelGGL(fp_concat_kernel, dim3(cdiv(rows * ((c1 + c2) / 4), 256)), dim3(256), 0,
                       st, dm.n[u], dm.n[k], c1, c2, known_cl, at_i(geom, p.g_nn_idx[j]),
                       at_f(geom, p.g_nn_w[j]), at_f(out, p.o_sa_cl[u - 1]), x, rows);
    BTR_TRY(btr_pm_chain_forward(&d.fp[j], &p.fp[j], nullptr, x, at_f(out, p.o_fp[j]),
                                 at_f(out, p.o_fp_cl[j]), (char *)saved + p.s_fp[j], scratch,
                                 stream));
  }
  return check_launch("backbone_forward");
}

int btr_backbone_backward(const btr_backbone_t *dp, const btr_backbone_plan_t *pp,
                          const void *geom, const void *out, const float *const *dout_sa,
                          const float *const *dout_fp, void *saved, float *grads, void *scratch,
                          btr_stream_t stream) {
  BTR_REQUIRE(dp && pp && geom && out && saved && grads && scratch,
              "backbone_backward: null pointer");
  const btr_backbone_t &d = *dp;
  const btr_backbone_plan_t &p = *pp;
  const Dims dm = dims_of(d);
  const int L = d.levels, F = d.fps;
  hipStream_t st = as_stream(stream);
  const BwdScratch sc = bwd_scratch(d, p);
  // ONE launch for the split-K reductions of every weight gradient of the call (the layer calls'
  // own scopes nest into this one: was one launch per layer call, 6 - 9 per step)
  constexpr bool defer = true;
  if (defer) reduce_batch_begin();
  struct Flush {
    hipStream_t s;
    bool on;
    ~Flush() { if (on) reduce_batch_flush(s); }
  } flush_all{st, defer};
  auto ext_sa = [&](int l) { return dout_sa ? dout_sa[l - 1] : nullptr; };   // level l = 1..L
  auto ext_fp = [&](int j) { return dout_fp ? dout_fp[j] : nullptr; };
  // which levels / modules receive a gradient at all (a level without one is skipped together
  // with everything that only it feeds: its parameters' gradients are then zero)
  bool fp_live[kMaxLv] = {}, sa_live[kMaxLv + 2] = {};
  for (int j = F - 1; j >= 0; --j) fp_live[j] = ext_fp(j) || (j + 1 < F && fp_live[j + 1]);
  for (int l = L; l >= 1; --l) {
    const int j = L - l - 1;   // the module whose skip input level l is
    sa_live[l] = ext_sa(l) || (l == L && F > 0 && fp_live[0]) ||
                 (l < L && (sa_live[l + 1] || (j >= 0 && j < F && fp_live[j])));
  }
  {
    bool all = true;
    for (int j = 0; j < F; ++j) all = all && fp_live[j];
    for (int l = 1; l <= L; ++l) all = all && sa_live[l];
    if (!all) {   // somebody's parameters get no gradient: they must read as zeros
      hipError_t e = hipMemsetAsync(grads, 0, sizeof(float) * p.grads_floats, st);
      if (e != hipSuccess) return fail((int)e, "backbone_backward memset: %s", hipGetErrorString(e));
    }
  }
  // ---- feature-propagation modules, last first
  for (int j = F - 1; j >= 0; --j) {
    if (!fp_live[j]) continue;
    const int u = fp_unknown(d, j), k = fp_known(d, j);
    const int c1 = fp_known_c(d, dm, j);
    const int cout = d.fp[j].width[d.fp[j].layers - 1];
    const float *dout = ext_fp(j);
    if (j + 1 < F && fp_live[j + 1]) {   // dK of the module above (+ an external gradient)
      float *dk = at_f(scratch, sc.dk[j + 1]);
      if (dout) BTR_TRY(add_inplace(dk, dout, (long long)d.b * cout * dm.n[u], st));
      dout = dk;
    }
    float *dx = at_f(scratch, sc.dx[j]);
    BTR_TRY(btr_pm_chain_backward(&d.fp[j], &p.fp[j], at_f(saved, p.s_fpx[j]), dout,
                                  (char *)saved + p.s_fp[j], grads + p.gr_fp[j], dx,
                                  (char *)scratch + sc.layer_fp[j], stream));
    // known level's share: scatter the interpolated channels back through the 3-NN lists
    BTR_TRY(ti_grad_lists(d.b, c1, dm.n[u], dm.n[k], dx, (long long)d.fp[j].c * dm.n[u],
                          at_i(geom, p.g_nn_idx[j]), at_f(geom, p.g_nn_w[j]),
                          at_f(scratch, sc.dk[j]), (char *)const_cast<void *>(geom) + p.g_ti[j],
                          p.g_ti_bytes[j], st, kScatterReduce));
  }
  // ---- set-abstraction levels, last first
  for (int l = L; l >= 1; --l) {
    if (!sa_live[l]) continue;
    const btr_sa_layer_t &s = d.sa[l - 1];
    const long long e = (long long)d.b * dm.c[l] * dm.n[l];
    const float *dout = nullptr;
    if (l == L) {
      if (F > 0 && fp_live[0]) {
        float *dk = at_f(scratch, sc.dk[0]);
        if (ext_sa(l)) BTR_TRY(add_inplace(dk, ext_sa(l), e, st));
        dout = dk;
      } else {
        dout = ext_sa(l);
      }
    } else {
      // df[l] was written by level l+1's backward (its feature gradient + the skip share);
      // when level l+1 is dead, the skip share / external gradient stand alone
      float *df = at_f(scratch, sc.df[l]);
      const int j = L - l - 1;
      const bool skip = j >= 0 && j < F && fp_live[j];
      if (!sa_live[l + 1]) {
        if (skip) {   // the slab (b, c2, n) of dx[j] behind its first c1 channels, made dense
          const int c1 = fp_known_c(d, dm, j);
          hipError_t er = hipMemcpy2DAsync(
              df, sizeof(float) * (size_t)dm.c[l] * dm.n[l],
              at_f(scratch, sc.dx[j]) + (size_t)c1 * dm.n[l],
              sizeof(float) * (size_t)d.fp[j].c * dm.n[l],
              sizeof(float) * (size_t)dm.c[l] * dm.n[l], d.b, hipMemcpyDeviceToDevice, st);
          if (er != hipSuccess) return fail((int)er, "backbone_backward copy: %s", hipGetErrorString(er));
          if (ext_sa(l)) BTR_TRY(add_inplace(df, ext_sa(l), e, st));
          dout = df;
        } else {
          dout = ext_sa(l);
        }
      } else {
        if (ext_sa(l)) BTR_TRY(add_inplace(df, ext_sa(l), e, st));
        dout = df;
      }
    }
    BTR_REQUIRE(dout, "backbone_backward: level %d has no gradient", l);
    // this level's feature gradient goes to level l-1, plus the skip share of the module
    // whose unknown level that is
    float *dfeat = nullptr;
    const float *add = nullptr;
    long long add_bs = 0;
    if (l > 1 && s.need_dfeat) {
      dfeat = at_f(scratch, sc.df[l - 1]);
      const int j = L - (l - 1) - 1;
      if (j >= 0 && j < F && fp_live[j]) {
        add = at_f(scratch, sc.dx[j]) + (size_t)fp_known_c(d, dm, j) * dm.n[l - 1];
        add_bs = (long long)d.fp[j].c * dm.n[l - 1];
      }
    }
    const SaGeom sg = level_geom(p, geom, l - 1);
    BTR_TRY(sa_layer_backward_add(&s, &p.sa[l - 1], at_i(geom, p.g_idx[l - 1]),
                                  at_f(out, p.o_sa[l - 1]), dout, (char *)saved + p.s_sa[l - 1],
                                  grads + p.gr_sa[l - 1], dfeat, nullptr, nullptr,
                                  (char *)scratch + sc.layer_sa[l - 1], add, add_bs, &sg, stream));
  }
  return check_launch("backbone_backward");
}

}  // extern "C"
