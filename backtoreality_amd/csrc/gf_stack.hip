// gf_stack.hip -- the GroupFree3D decoder stack, one call per direction.
//
// reference: detection/GroupFree3D/models/detector.py:161-219 (the loop over the decoder layers:
// position embeddings -> decoder layer -> prediction head -> next query position) and
// detector.py:204-230 (the detached (center, pred_size) query position).
//
// Nothing is computed here that the per-module entry points do not compute: the host walks the
// launches of btr_pm_chain_forward / _backward (position embeddings, prediction heads),
// btr_decoder_layer_forward / _backward and btr_gf_head_decode in the order the module loop of
// the reference runs them, on buffers cut out of one `saved` / `scratch` / `grads` allocation.
// The step was paced by the host: 39 autograd nodes, ~0.1 ms of interpreter each way around ~10
// launches of 5 - 20 us each (tools/host_profile_gf.py), 3 us of which is the launch itself
// (tools/probe/launch_cost.hip).  One node instead of 30 leaves the launches.
//
// Data handed between the modules stays in channel-last rows (the layout every GEMM here reads):
//   x[i]        (b*pq, e)   output of layer i  -> layer i+1, head i            (saved)
//   qpos_cl[i]  (b*pq, e)   self position embedding of layer i                 (saved)
//   kpos_cl[i]  (b*pk, e)   cross position embedding of layer i                (saved)
// Backward, layer i (from the last): head i's chain gives d x[i] (+ the gradient arriving from
// layer i+1, + the caller's gradient of the last output), the layer gives d x[i-1], d qpos_cl[i],
// d key (which is d kpos_cl[i] as well: kp = key + kpos), the two embedding chains take theirs,
// the key gradients of all layers are summed at the end.
//
// Round 6: two lanes.  The launches are 5 - 20 us on 1 024 - 4 096 rows -- most of the chip idles --
// and about a third of them are not on the path from one layer's input to its output:
//   forward   the key position embedding of a layer and the cross-attention's key / value rows
//             (they depend on the seed points only);
//   backward  the weight / bias / LayerNorm parameter gradients, the key rows' input gradient, and
//             the whole backward of the two position-embedding chains (their inputs are
//             coordinates: nothing waits for them).
// Those run on a side stream of the library, per layer, ordered against the main lane by events;
// each lane's per-layer segment is ONE replayed HIP graph (graph_cache.hip: a graph with parallel
// branches takes the runtime's slow path, two linear graphs on two streams do not).  Buffers a
// side segment reads are double-buffered across layers; the main lane waits for side segment
// i + 2 before it reuses them for layer i.  Same launches, same operands, same results as the
// one-stream sequence, which remains for callers inside a stream capture, under the GEMM trace,
// or with BTR_GRAPHS=0.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

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
inline void *at_v(void *base, size_t off) { return (void *)((char *)base + off); }

#define BTR_TRY(call)              \
  do {                             \
    const int rc_ = (call);        \
    if (rc_ != BTR_OK) return rc_; \
  } while (0)

constexpr int kMaxOps = 16;
struct SumArgs {
  const float4 *src[kMaxOps];
  int count;
  long long n4;
  float4 *dst;
};
// dst = src[0] + src[1] + ... (in that order), 16 bytes per lane
__global__ __launch_bounds__(256) void sum_rows_kernel(SumArgs a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n4) return;
  float4 s = a.src[0][i];
  for (int k = 1; k < a.count; ++k) {
    const float4 v = a.src[k][i];
    s.x += v.x;
    s.y += v.y;
    s.z += v.z;
    s.w += v.w;
  }
  a.dst[i] = s;
}
int sum_into(hipStream_t s, long long n, const float *const *src, int count, float *dst) {
  SumArgs a{};
  a.count = count;
  a.n4 = n / 4;
  a.dst = (float4 *)dst;
  for (int k = 0; k < count; ++k) a.src[k] = (const float4 *)src[k];
  hipLaunchKernelGGL(sum_rows_kernel, dim3((unsigned)((a.n4 + 255) / 256)), dim3(256), 0, s, a);
  return check_launch("gf_stack sum");
}

// Backward scratch.  Gradients travel between the modules as channel-last rows -- the layout the
// layer's LayerNorm backward and the chains' GEMMs read -- so no (b, e, p) tensor is formed between
// them (per layer: the head chain's pm_out, the sum kernel, the layer's pm_rows and its three
// rows_to_bcp launches are gone; tests/test_gf_stack_gpu.py: still bit for bit the module loop's
// numbers, the operands are added in the same order).
struct BwdScratch {
  size_t layer[2];            // a decoder layer's own scratch, layers alternate (the side lane reads it)
  size_t head;                // the head chains' (lane 2, before the embedding chains)
  size_t head_last;           // the last layer's head chain's (it runs on the main lane, beside them)
  size_t side;                // the position-embedding chains' (lane 2)
  size_t dhx[BTR_GF_MAX_DECODER_LAYERS];   // (b*pq, e) per layer: gradient of x[i] from head i
  size_t pair[2][2];          // (b*pq, e) x 2: d res1, d qp0 of a layer (their sum = d x[i-1]), ping-pong
  size_t dqp1[2];             // (b*pq, e): with d qp0 the gradient of the query position embedding
  size_t dlast;               // (b*pq, e): the caller's gradient of the last output, as rows
  size_t dkp[BTR_GF_MAX_DECODER_LAYERS];   // (b*pk, e) per layer: d key = d kpos
  size_t ksum;                // (b*pk, e)
  size_t bytes;
};
BwdScratch bwd_scratch(const btr_gf_stack_t &d, const btr_gf_stack_plan_t &p) {
  BwdScratch s{};
  Bump b;
  size_t layer = 0, head = 0, side = 0;
  for (int i = 0; i < d.layers; ++i) {
    layer = std::max(layer, p.layer[i].bwd_scratch_bytes);
    head = std::max(head, p.head[i].bwd_scratch_bytes);
    if (d.has_qpos) side = std::max(side, p.qpos[i].bwd_scratch_bytes);
    if (d.has_kpos) side = std::max(side, p.kpos[i].bwd_scratch_bytes);
  }
  s.layer[0] = b.take(layer);
  s.layer[1] = b.take(layer);
  s.head = b.take(head);
  s.head_last = b.take(head);
  s.side = b.take(side);
  const size_t q = (size_t)d.b * d.pq * d.e, k = (size_t)d.b * d.pk * d.e;
  for (int i = 0; i < d.layers; ++i) s.dhx[i] = b.floats(q);
  for (int a = 0; a < 2; ++a)
    for (int c = 0; c < 2; ++c) s.pair[a][c] = b.floats(q);
  s.dqp1[0] = b.floats(q);
  s.dqp1[1] = b.floats(q);
  s.dlast = b.floats(q);
  for (int i = 0; i < d.layers; ++i) s.dkp[i] = b.floats(k);
  s.ksum = b.floats(k);
  s.bytes = b.off;
  return s;
}
// Forward scratch: the modules' own (they run one after the other), and the side lane's key
// position embedding chains'.
struct FwdScratch {
  size_t main, side, bytes;
};
FwdScratch fwd_scratch(const btr_gf_stack_t &d, const btr_gf_stack_plan_t &p) {
  FwdScratch s{};
  Bump b;
  size_t m = 0, sd = 0;
  for (int i = 0; i < d.layers; ++i) {
    m = std::max(m, p.layer[i].fwd_scratch_bytes);
    m = std::max(m, p.head[i].fwd_scratch_bytes);
    if (d.has_qpos) m = std::max(m, p.qpos[i].fwd_scratch_bytes);
    if (d.has_kpos) {
      m = std::max(m, p.kpos[i].fwd_scratch_bytes);
      sd = std::max(sd, p.kpos[i].fwd_scratch_bytes);
    }
  }
  s.main = b.take(m);
  s.side = b.take(sd);
  s.bytes = b.off;
  return s;
}

// The lanes of a call: 0 = the caller's stream, 1 and 2 = the library's side streams.
// lane == NULL: one stream, launches issued one by one, no events.
struct Lanes {
  hipStream_t main = nullptr;
  SideLane *lane = nullptr;
  uint64_t key = 0;
  // BTR_GF_LANES=1 | 2 | 3 (default 3): how many streams the three lanes are mapped onto (the
  // events order them on any number; 1 = replayed graphs on the caller's stream alone).  With the
  // cross-attention's d k / d v kernel on lane 2 the side work of a layer (115 + 131 + 130 us) is
  // as long as the main lane's (~360 us): on ONE side stream it becomes the critical path (step
  // 8.9 ms against 8.3 - 8.7 with a stream per lane, same box; profiles/r06_gf_lanes_spread.txt
  // has the earlier comparison, when the side work was 290 us and two streams sufficed)
  static int count() {
    static const int n = [] {
      const char *e = getenv("BTR_GF_LANES");
      const int v = e ? atoi(e) : 3;
      return v < 1 ? 1 : (v > 3 ? 3 : v);
    }();
    return n;
  }
  hipStream_t stream(int which) const {
    if (which >= count()) which = count() - 1;
    return lane && which ? lane->s[which - 1] : main;
  }
  hipStream_t other_side() const { return count() >= 3 ? lane->s[1] : lane->s[0]; }
  // BTR_LANE_DEBUG=1: an event pair around every segment; report() waits for the lanes and prints
  // when each segment started and how long it took on the GPU (a profiler's interception makes
  // the host the bottleneck, and the overlap it then shows is the host's, not the queues')
  static bool debug() {
    static const bool on = getenv("BTR_LANE_DEBUG") != nullptr;
    return on;
  }
  struct Mark {
    int segment, which;
    hipEvent_t a, b;
  };
  mutable std::vector<Mark> marks;
  // one segment: a linear sequence of launches on one of the streams
  int run(int segment, int which, const std::function<int(hipStream_t)> &body) const {
    hipStream_t s = stream(which);
    if (!lane) return body(s);
    Mark m{segment, which, nullptr, nullptr};
    if (debug()) {
      (void)hipEventCreate(&m.a);
      (void)hipEventCreate(&m.b);
      (void)hipEventRecord(m.a, s);
    }
    const int rc = graph_run(hash_bytes(key, &segment, sizeof(segment)), s, body, nullptr);
    if (debug()) {
      (void)hipEventRecord(m.b, s);
      marks.push_back(m);
    }
    return rc;
  }
  void signal(int ev, int from) const {
    if (lane) (void)hipEventRecord(lane->ev[ev], stream(from));
  }
  void wait(int ev, int on) const {
    if (lane) (void)hipStreamWaitEvent(stream(on), lane->ev[ev], 0);
  }
  void report(const char *what) const {
    if (!debug() || !lane || marks.empty()) return;
    (void)hipStreamSynchronize(lane->s[0]);
    (void)hipStreamSynchronize(other_side());
    (void)hipStreamSynchronize(main);
    fprintf(stderr, "lanes of %s (us after the first segment's start: start + duration)\n", what);
    for (const Mark &m : marks) {
      float t0 = 0.f, dt = 0.f;
      (void)hipEventElapsedTime(&t0, marks[0].a, m.a);
      (void)hipEventElapsedTime(&dt, m.a, m.b);
      fprintf(stderr, "   %*slane %d segment %3d: %8.1f + %7.1f\n", 10 * m.which, "", m.which,
              m.segment, 1e3f * t0, 1e3f * dt);
    }
    for (const Mark &m : marks) {
      (void)hipEventDestroy(m.a);
      (void)hipEventDestroy(m.b);
    }
    marks.clear();
  }
};
Lanes lanes_of(hipStream_t hs, uint64_t key) {
  Lanes l;
  l.main = hs;
  l.key = key;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  const bool outer = hipStreamIsCapturing(hs, &st) == hipSuccess &&
                     st != hipStreamCaptureStatusNone;
  if (graphs_enabled() && !outer && !graph_capturing() && !gemm_trace_active())
    l.lane = side_lane(hs, Lanes::count() >= 3 ? 2 : 1);
  return l;
}

int check(const btr_gf_stack_t &d) {
  BTR_REQUIRE(d.layers > 0 && d.layers <= BTR_GF_MAX_DECODER_LAYERS, "gf_stack: %d layers",
              d.layers);
  BTR_REQUIRE(d.b > 0 && d.pq > 0 && d.pk > 0 && d.e > 0 && d.e % 4 == 0 && d.head_c > 0,
              "gf_stack: bad sizes");
  for (int i = 0; i < d.layers; ++i) {
    const btr_decoder_layer_t &l = d.layer[i];
    BTR_REQUIRE(l.b == d.b && l.pq == d.pq && l.pk == d.pk && l.e == d.e,
                "gf_stack: layer %d sizes differ from the stack's", i);
    const btr_pm_chain_t &h = d.head[i];
    BTR_REQUIRE(h.b == d.b && h.n == d.pq && h.c == d.e && h.layers >= 1 &&
                    h.width[h.layers - 1] == d.head_c && h.need_dx == 1,
                "gf_stack: head %d is not a (b, e, pq) -> (b, head_c, pq) chain with need_dx", i);
    if (d.has_qpos) {
      const btr_pm_chain_t &c = d.qpos[i];
      BTR_REQUIRE(c.b == d.b && c.n == d.pq && c.layers >= 1 && c.width[c.layers - 1] == d.e &&
                      c.need_dx == 0,
                  "gf_stack: query position embedding %d", i);
    }
    if (d.has_kpos) {
      const btr_pm_chain_t &c = d.kpos[i];
      BTR_REQUIRE(c.b == d.b && c.n == d.pk && c.layers >= 1 && c.width[c.layers - 1] == d.e &&
                      c.need_dx == 0,
                  "gf_stack: key position embedding %d", i);
    }
  }
  return BTR_OK;
}

}  // namespace
}  // namespace btr

using namespace btr;

extern "C" {

// sizeof(btr_gf_stack_t) (0) / sizeof(btr_gf_stack_plan_t) (1): lets a binding written in another
// language check its mirror of the two structs before it hands one over
long long btr_gf_stack_sizeof(int which) {
  return which == 0 ? (long long)sizeof(btr_gf_stack_t) : (long long)sizeof(btr_gf_stack_plan_t);
}

int btr_gf_stack_plan(const btr_gf_stack_t *dp, btr_gf_stack_plan_t *p) {
  BTR_REQUIRE(dp && p, "gf_stack_plan: null pointer");
  const btr_gf_stack_t &d = *dp;
  BTR_TRY(check(d));
  std::memset(p, 0, sizeof(*p));
  Bump sv;
  size_t g = 0;
  const size_t q = (size_t)d.b * d.pq * d.e, k = (size_t)d.b * d.pk * d.e;
  for (int i = 0; i < d.layers; ++i) {
    BTR_TRY(btr_decoder_layer_plan(&d.layer[i], &p->layer[i]));
    BTR_TRY(btr_pm_chain_plan(&d.head[i], &p->head[i]));
    if (d.has_qpos) BTR_TRY(btr_pm_chain_plan(&d.qpos[i], &p->qpos[i]));
    if (d.has_kpos) BTR_TRY(btr_pm_chain_plan(&d.kpos[i], &p->kpos[i]));
    p->s_layer[i] = sv.take(p->layer[i].saved_bytes);
    p->s_head[i] = sv.take(p->head[i].saved_bytes);
    p->s_x[i] = sv.floats(q);
    p->g_layer[i] = g;
    g += p->layer[i].grads_floats;
    p->g_head[i] = g;
    g += p->head[i].grads_floats;
    if (d.has_qpos) {
      p->s_qpos[i] = sv.take(p->qpos[i].saved_bytes);
      p->s_qpos_cl[i] = sv.floats(q);
      p->g_qpos[i] = g;
      g += p->qpos[i].grads_floats;
    }
    if (d.has_kpos) {
      p->s_kpos[i] = sv.take(p->kpos[i].saved_bytes);
      p->s_kpos_cl[i] = sv.floats(k);
      p->g_kpos[i] = g;
      g += p->kpos[i].grads_floats;
    }
  }
  p->saved_bytes = sv.off;
  p->grads_floats = g;
  p->fwd_scratch_bytes = fwd_scratch(d, *p).bytes;
  p->bwd_scratch_bytes = bwd_scratch(d, *p).bytes;
  return BTR_OK;
}

int btr_gf_stack_forward(const btr_gf_stack_t *dp, const btr_gf_stack_plan_t *pp,
                         const float *query_cl, const float *key_cl, const float *qpos0_t,
                         const float *key_xyz_t, const float *base_xyz, const float *mean_size,
                         float *const *head_out, float *const *head_out_cl, float *const *center,
                         float *const *heading_residuals, float *const *size_residuals,
                         float *const *pred_size, float *const *query_pos,
                         float *const *query_pos_t, float *last_bcp, float *last_cl, void *saved,
                         void *scratch, btr_stream_t stream) {
  BTR_REQUIRE(dp && pp && query_cl && key_cl && base_xyz && mean_size && head_out &&
                  head_out_cl && center && heading_residuals && size_residuals && pred_size &&
                  query_pos && query_pos_t && saved && scratch,
              "gf_stack_forward: null pointer");
  const btr_gf_stack_t &d = *dp;
  const btr_gf_stack_plan_t &p = *pp;
  BTR_REQUIRE(!d.has_qpos || qpos0_t, "gf_stack_forward: the first query position is missing");
  BTR_REQUIRE(!d.has_kpos || key_xyz_t, "gf_stack_forward: the key position is missing");
  hipStream_t hs = as_stream(stream);
  const int L = d.layers;
  for (int i = 0; i < L; ++i)
    BTR_REQUIRE(head_out[i] && head_out_cl[i] && center[i] && heading_residuals[i] &&
                    size_residuals[i] && pred_size[i] && query_pos[i] && query_pos_t[i],
                "gf_stack_forward: outputs of layer %d", i);
  // everything the launches depend on: the descriptor (sizes, parameter pointers, seeds), the
  // plan's offsets follow from it; the arguments; the contents of the pointer arrays
  uint64_t key = hash_bytes(0x9f57ac1e5eedull, &d, sizeof(d));
  const void *args[] = {query_cl, key_cl, qpos0_t, key_xyz_t, base_xyz, mean_size,
                        last_bcp, last_cl, saved, scratch};
  key = hash_bytes(key, args, sizeof(args));
  float *const *arrays[] = {head_out, head_out_cl, center, heading_residuals, size_residuals,
                            pred_size, query_pos, query_pos_t};
  for (auto *a : arrays) key = hash_bytes(key, a, sizeof(float *) * L);
  const Lanes ln = lanes_of(hs, key);
  const FwdScratch fsc = fwd_scratch(d, p);
  void *scr_main = at_v(scratch, fsc.main), *scr_side = at_v(scratch, fsc.side);
  // ---- side lane: per layer, the key position embedding and the key / value rows
  bool ahead = ln.lane != nullptr && d.has_kpos;
  for (int i = 0; ahead && i < L; ++i) ahead = decoder_kv_separable(&d.layer[i], &p.layer[i]);
  if (ahead) {
    ln.signal(0, 0);   // the inputs are there
    ln.wait(0, 1);
    for (int i = 0; i < L; ++i) {
      BTR_TRY(ln.run(100 + i, 1, [&](hipStream_t bs) -> int {
        float *kpos_cl = at_f(saved, p.s_kpos_cl[i]);
        BTR_TRY(btr_pm_chain_forward(&d.kpos[i], &p.kpos[i], key_xyz_t, nullptr, nullptr, kpos_cl,
                                     at_v(saved, p.s_kpos[i]), scr_side, (btr_stream_t)bs));
        return decoder_layer_kv(&d.layer[i], &p.layer[i], key_cl, kpos_cl,
                                at_v(saved, p.s_layer[i]), (btr_stream_t)bs);
      }));
      ln.signal(1 + i, 1);
    }
  }
  // ---- main lane: per layer, query position embedding -> layer -> head -> decoded boxes.  The
  // first layer is two segments with the wait for its keys between them (the embedding and the
  // self-attention do not read a key: the side lane's first segment hides behind them); the
  // later layers' keys are long there when the main lane arrives.
  for (int i = 0; i < L; ++i) {
    const float *x = i == 0 ? query_cl : at_f(saved, p.s_x[i - 1]);
    const float *qpos_t = i == 0 ? qpos0_t : query_pos_t[i - 1];
    float *qpos_cl = d.has_qpos ? at_f(saved, p.s_qpos_cl[i]) : nullptr;
    float *kpos_cl = d.has_kpos ? at_f(saved, p.s_kpos_cl[i]) : nullptr;
    float *xo = at_f(saved, p.s_x[i]);
    const bool split = ahead && i == 0;
    auto first = [&](hipStream_t bs, int parts) -> int {
      btr_stream_t stream = (btr_stream_t)bs;
      if (d.has_qpos)
        BTR_TRY(btr_pm_chain_forward(&d.qpos[i], &p.qpos[i], qpos_t, nullptr, nullptr, qpos_cl,
                                     at_v(saved, p.s_qpos[i]), scr_main, stream));
      if (d.has_kpos && !ahead)
        BTR_TRY(btr_pm_chain_forward(&d.kpos[i], &p.kpos[i], key_xyz_t, nullptr, nullptr,
                                     kpos_cl, at_v(saved, p.s_kpos[i]), scr_main, stream));
      return decoder_layer_forward_ex(&d.layer[i], &p.layer[i], x, key_cl, qpos_cl, kpos_cl,
                                      i == L - 1 ? last_bcp : nullptr, xo,
                                      at_v(saved, p.s_layer[i]), scr_main, ahead ? 1 : 0, parts,
                                      stream);
    };
    auto second = [&](hipStream_t bs) -> int {
      btr_stream_t stream = (btr_stream_t)bs;
      BTR_TRY(btr_pm_chain_forward(&d.head[i], &p.head[i], nullptr, xo, head_out[i],
                                   head_out_cl[i], at_v(saved, p.s_head[i]), scr_main, stream));
      const int cp = p.head[i].np[d.head[i].layers - 1];
      BTR_TRY(btr_gf_head_decode(d.b, d.pq, d.nh, d.ns, head_out_cl[i], (long long)d.pq * cp, cp,
                                 1, base_xyz, mean_size, center[i], heading_residuals[i],
                                 size_residuals[i], pred_size[i], query_pos[i], query_pos_t[i],
                                 stream));
      if (i == L - 1 && last_cl)
        (void)hipMemcpyAsync(last_cl, xo, (size_t)d.b * d.pq * d.e * sizeof(float),
                             hipMemcpyDeviceToDevice, bs);
      return check_launch("gf_stack_forward");
    };
    if (split) {
      BTR_TRY(ln.run(50 + i, 0, [&](hipStream_t bs) -> int { return first(bs, kDecoderFwdSelf); }));
      ln.wait(1 + i, 0);
      BTR_TRY(ln.run(i, 0, [&](hipStream_t bs) -> int {
        BTR_TRY(decoder_layer_forward_ex(&d.layer[i], &p.layer[i], x, key_cl, qpos_cl, kpos_cl,
                                         i == L - 1 ? last_bcp : nullptr, xo,
                                         at_v(saved, p.s_layer[i]), scr_main, 1, kDecoderFwdRest,
                                         (btr_stream_t)bs));
        return second(bs);
      }));
    } else {
      if (ahead) ln.wait(1 + i, 0);
      BTR_TRY(ln.run(i, 0, [&](hipStream_t bs) -> int {
        BTR_TRY(first(bs, kDecoderFwdSelf | kDecoderFwdRest));
        return second(bs);
      }));
    }
  }
  ln.report("gf_stack_forward");
  return BTR_OK;
}

int btr_gf_stack_backward(const btr_gf_stack_t *dp, const btr_gf_stack_plan_t *pp,
                          const float *query_cl, const float *key_cl,
                          const float *const *dhead, const float *dlast_bcp, void *saved,
                          float *grads, float *dquery_bcp, float *dkey_bcp, void *scratch,
                          btr_stream_t stream) {
  BTR_REQUIRE(dp && pp && query_cl && key_cl && dhead && saved && grads && scratch,
              "gf_stack_backward: null pointer");
  const btr_gf_stack_t &d = *dp;
  const btr_gf_stack_plan_t &p = *pp;
  hipStream_t hs = as_stream(stream);
  uint64_t key = hash_bytes(0xbac4b0a2d5eedull, &d, sizeof(d));
  const void *args[] = {query_cl, key_cl, dlast_bcp, saved, grads, dquery_bcp, dkey_bcp, scratch};
  key = hash_bytes(key, args, sizeof(args));
  key = hash_bytes(key, dhead, sizeof(float *) * d.layers);
  const Lanes ln = lanes_of(hs, key);
  const BwdScratch sc = bwd_scratch(d, p);
  const long long q = (long long)d.b * d.pq * d.e, k = (long long)d.b * d.pk * d.e;
  const int L = d.layers;
  // Three lanes.  0 (the caller's stream): per layer, the decoder layer's path from its output
  // gradient to its input's.  2: first the head chains' backwards, last layer first -- they
  // depend on the loss only, and layer i's path starts from d x[i] = head i's part + what comes
  // from above -- then, per layer, the key rows' gradient and the two position-embedding chains.
  // 1: per layer, the weight / bias / LayerNorm parameter gradients.
  // Events: 0 the inputs are there; H(i) head chain i is through; M(i) the main segment of layer
  // i; R(i), C(i) the segments of lanes 1 and 2 that read layer i's scratch (the main lane waits
  // for those of layer i + 2 before it reuses the buffers for layer i).
  auto H = [&](int i) { return 1 + i; };
  auto M = [&](int i) { return 1 + L + i; };
  auto R = [&](int i) { return 1 + 2 * L + i; };
  auto C = [&](int i) { return 1 + 3 * L + i; };
  auto Kd = [&](int i) { return 1 + 4 * L + i; };   // the cross-attention's d k / d v are there
  ln.signal(0, 0);
  ln.wait(0, 2);
  for (int i = L - 1; i >= 0; --i) {
    // (the last layer's head chain is what the main lane starts from: on its own stream, without
    // the hand-over; the others run beside it)
    const int hl = i == L - 1 ? 0 : 2;
    BTR_TRY(ln.run(200 + i, hl, [&](hipStream_t bs) -> int {
      if (dhead[i])
        return pm_chain_backward_rows(&d.head[i], &p.head[i], at_f(saved, p.s_x[i]), dhead[i],
                                      nullptr, nullptr, at_v(saved, p.s_head[i]),
                                      grads + p.g_head[i], nullptr, at_f(scratch, sc.dhx[i]),
                                      at_v(scratch, i == L - 1 ? sc.head_last : sc.head),
                                      (btr_stream_t)bs);
      (void)hipMemsetAsync(grads + p.g_head[i], 0, p.head[i].grads_floats * sizeof(float), bs);
      return check_launch("gf_stack_backward");
    }));
    ln.signal(H(i), hl);
  }
  // what reaches x[i] from above: the pair (d res1, d qp0) of layer i+1, or the caller's gradient
  const float *up0 = dlast_bcp ? at_f(scratch, sc.dlast) : nullptr, *up1 = nullptr;
  int flip = 0;
  for (int i = L - 1; i >= 0; --i) {
    const float *g0 = up0, *g1 = up1, *g2 = nullptr;
    float *dhx = at_f(scratch, sc.dhx[i]);
    if (dhead[i]) {
      // (module loop: d x[i] = head's + the layer's / the caller's -- a + b = b + a exactly, and
      // the layer's own two parts are added first as its rows_to_bcp did)
      if (g0) g2 = dhx; else g0 = dhx;
    }
    const bool reached = g0 != nullptr;   // else nothing reaches this layer (nor any below it)
    const float *x_in = i == 0 ? query_cl : at_f(saved, p.s_x[i - 1]);
    const float *qpos_cl = d.has_qpos ? at_f(saved, p.s_qpos_cl[i]) : nullptr;
    const float *kpos_cl = d.has_kpos ? at_f(saved, p.s_kpos_cl[i]) : nullptr;
    float *dkp = at_f(scratch, sc.dkp[i]);
    void *sub = at_v(scratch, sc.layer[i & 1]);
    const DecoderRowsOut out{at_f(scratch, sc.pair[flip][0]), at_f(scratch, sc.pair[flip][1]),
                             at_f(scratch, sc.dqp1[flip]), dkp};
    if (reached) flip ^= 1;
    // ---- lane 0
    ln.wait(H(i), 0);
    if (i + 2 <= L - 1) {   // the buffers of layer i + 2 are free
      ln.wait(R(i + 2), 0);
      ln.wait(C(i + 2), 0);
    }
    BTR_TRY(ln.run(i, 0, [&](hipStream_t bs) -> int {
      btr_stream_t st = (btr_stream_t)bs;
      if (i == L - 1 && dlast_bcp)
        BTR_TRY(btr_pm_rows(d.b, d.pq, d.e, d.e, dlast_bcp, at_f(scratch, sc.dlast), st));
      if (!reached) {
        (void)hipMemsetAsync(grads + p.g_layer[i], 0, p.layer[i].grads_floats * sizeof(float), bs);
        (void)hipMemsetAsync(dkp, 0, (size_t)k * sizeof(float), bs);
        if (d.has_qpos)
          (void)hipMemsetAsync(grads + p.g_qpos[i], 0, p.qpos[i].grads_floats * sizeof(float), bs);
        if (d.has_kpos)
          (void)hipMemsetAsync(grads + p.g_kpos[i], 0, p.kpos[i].grads_floats * sizeof(float), bs);
        return check_launch("gf_stack_backward");
      }
      return decoder_layer_backward_parts(&d.layer[i], &p.layer[i], x_in, key_cl, qpos_cl, kpos_cl,
                                          nullptr, g0, g1, g2, at_v(saved, p.s_layer[i]),
                                          grads + p.g_layer[i], i == 0 ? dquery_bcp : nullptr,
                                          nullptr, nullptr, &out, sub, kDecoderBwdChain, st);
    }));
    ln.signal(M(i), 0);
    // ---- lane 2 first: the cross-attention's d k / d v and the key rows' gradient (88 + 27 us
    // that the layer's input gradient does not wait for).  (Layer 0's lane-2 work goes onto the
    // caller's stream: nothing is left for it to do but wait for the side lanes, and the
    // parameter gradients of layer 0 run beside it instead of in front of it)
    const int cl = i == 0 ? 0 : 2;
    ln.wait(M(i), cl);
    if (reached)
      BTR_TRY(ln.run(500 + i, cl, [&](hipStream_t bs) -> int {
        return decoder_layer_backward_parts(&d.layer[i], &p.layer[i], x_in, key_cl, qpos_cl,
                                            kpos_cl, nullptr, nullptr, nullptr, nullptr,
                                            at_v(saved, p.s_layer[i]), grads + p.g_layer[i],
                                            nullptr, nullptr, nullptr, &out, sub, kDecoderBwdKey,
                                            (btr_stream_t)bs);
      }));
    ln.signal(Kd(i), cl);
    // ---- lane 1: the parameter gradients of the layer (the d k / d v rows among their operands)
    ln.wait(M(i), 1);
    ln.wait(Kd(i), 1);
    if (reached)
      BTR_TRY(ln.run(100 + i, 1, [&](hipStream_t bs) -> int {
        return decoder_layer_backward_parts(&d.layer[i], &p.layer[i], x_in, key_cl, qpos_cl,
                                            kpos_cl, nullptr, nullptr, nullptr, nullptr,
                                            at_v(saved, p.s_layer[i]), grads + p.g_layer[i],
                                            nullptr, nullptr, nullptr, &out, sub, kDecoderBwdRest,
                                            (btr_stream_t)bs);
      }));
    ln.signal(R(i), 1);
    // ---- lane 2 again: the two position-embedding chains
    if (reached)
      BTR_TRY(ln.run(300 + i, cl, [&](hipStream_t bs) -> int {
        btr_stream_t st = (btr_stream_t)bs;
        if (d.has_qpos)
          BTR_TRY(pm_chain_backward_rows(&d.qpos[i], &p.qpos[i], nullptr, nullptr, out.dqp0,
                                         out.dqp1, at_v(saved, p.s_qpos[i]), grads + p.g_qpos[i],
                                         nullptr, nullptr, at_v(scratch, sc.side), st));
        if (d.has_kpos)
          BTR_TRY(pm_chain_backward_rows(&d.kpos[i], &p.kpos[i], nullptr, nullptr, dkp, nullptr,
                                         at_v(saved, p.s_kpos[i]), grads + p.g_kpos[i], nullptr,
                                         nullptr, at_v(scratch, sc.side), st));
        return check_launch("gf_stack_backward");
      }));
    ln.signal(C(i), cl);
    if (reached) {
      up0 = out.dres1;
      up1 = out.dqp0;
    } else {
      up0 = up1 = nullptr;
    }
  }
  // the side streams are in order: their last segments being through means all are
  ln.wait(R(0), 0);
  ln.wait(C(0), 0);
  if (dquery_bcp && !up0)
    (void)hipMemsetAsync(dquery_bcp, 0, (size_t)q * sizeof(float), hs);
  if (dkey_bcp)   // the layers' key gradients, last layer first (autograd's order), then (b, e, pk)
    BTR_TRY(ln.run(99, 0, [&](hipStream_t bs) -> int {
      const float *src[BTR_GF_MAX_DECODER_LAYERS];
      for (int i = 0; i < L; ++i) src[i] = at_f(scratch, sc.dkp[L - 1 - i]);
      BTR_TRY(sum_into(bs, k, src, L, at_f(scratch, sc.ksum)));
      return btr_pm_out(d.b, d.pk, d.e, d.e, at_f(scratch, sc.ksum), nullptr, nullptr, 0, dkey_bcp,
                        nullptr, (btr_stream_t)bs);
    }));
  ln.report("gf_stack_backward");
  return check_launch("gf_stack_backward");
}

}  // extern "C"
