// graph_cache.hip -- launch-bound call sequences of the library as replayed HIP graphs.
//
// The GroupFree3D decoder stack (csrc/gf_stack.hip) is ~280 launches forward and ~420 backward of
// 5 - 20 us each on 1 024 - 4 096 rows: issued one by one they cost the host 2.6 - 3.6 us each and
// the in-order queue 2.5 - 3.5 us of idle time between two dependent kernels; as ONE captured
// graph on ONE stream the launch costs the host 15 us in all and the gap is ~1.1 us
// (tools/probe/graph_cost.hip, profiles/r06_graph_cost.txt).  A graph with parallel branches
// (fork / join through a side stream inside the capture) takes the runtime's slow path -- 3.5 - 4
// us of host time PER NODE -- so everything captured here is a linear chain; overlap comes from
// two linear graphs on two streams (see graph_run's callers).
//
// A graph replays exact kernel arguments, so a call is replayed only when EVERYTHING that
// determines its launches -- sizes, every pointer, every scalar the host passes by value -- hashes
// to a key seen before (the caller hashes its descriptor and arguments).  First sight of a key: the
// launches are issued as always; second sight: captured (nothing runs while capturing),
// instantiated (~10 ms, once) and launched; from then on: launched.  Callers that want hits keep
// their buffers (python: groupfree/fused_stack.py's slots).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <list>
#include <mutex>
#include <unordered_map>

#include "internal.hpp"

namespace btr {
namespace {

struct Cached {
  hipGraphExec_t exec = nullptr;
  hipGraph_t graph = nullptr;
  int sightings = 0;
  bool refused = false;   // a capture of this key failed once: stay eager
  std::list<uint64_t>::iterator lru;
};
struct Cache {
  std::mutex mu;
  std::unordered_map<uint64_t, Cached> map;
  std::list<uint64_t> order;   // most recent first
  long long replays = 0, captures = 0, eager = 0;
};
Cache &cache() {
  static Cache c;
  return c;
}
constexpr size_t kMaxGraphs = 512;   // (a two-branch GroupFree3D step holds 2 x 37 segments)

thread_local int g_capturing = 0;

// one capture stream per device and host thread (captures do not nest and run nothing)
hipStream_t capture_stream() {
  constexpr int kMaxDev = 16;
  static thread_local hipStream_t s[kMaxDev] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
  if (!s[dev] && hipStreamCreateWithFlags(&s[dev], hipStreamNonBlocking) != hipSuccess)
    s[dev] = nullptr;
  return s[dev];
}

void drop(Cached &c) {
  if (c.exec) (void)hipGraphExecDestroy(c.exec);
  if (c.graph) (void)hipGraphDestroy(c.graph);
  c.exec = nullptr;
  c.graph = nullptr;
}

}  // namespace

bool graph_capturing() { return g_capturing > 0; }

bool graphs_enabled() {
  const char *e = getenv("BTR_GRAPHS");   // (read per call: the tests toggle it)
  return !(e && e[0] == '0');
}

uint64_t hash_bytes(uint64_t h, const void *p, size_t n) {
  const unsigned char *b = (const unsigned char *)p;
  // FNV-1a over 8-byte words (the descriptors are arrays of pointers and ints), bytes at the end
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    memcpy(&w, b + i, 8);
    h = (h ^ w) * 0x100000001b3ull;
    h ^= h >> 29;
  }
  for (; i < n; ++i) h = (h ^ b[i]) * 0x100000001b3ull;
  return h;
}

int graph_run(uint64_t key, hipStream_t stream, const std::function<int(hipStream_t)> &body,
              int *how) {
  if (how) *how = 0;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  const bool outer = hipStreamIsCapturing(stream, &st) == hipSuccess &&
                     st != hipStreamCaptureStatusNone;
  if (!graphs_enabled() || outer || g_capturing || gemm_trace_active()) return body(stream);
  Cache &c = cache();
  hipGraphExec_t exec = nullptr;
  bool capture = false;
  {
    std::lock_guard<std::mutex> lock(c.mu);
    auto it = c.map.find(key);
    if (it == c.map.end()) {
      while (c.map.size() >= kMaxGraphs) {   // evict the least recently used key
        const uint64_t old = c.order.back();
        c.order.pop_back();
        drop(c.map[old]);
        c.map.erase(old);
      }
      c.order.push_front(key);
      Cached n;
      n.sightings = 1;
      n.lru = c.order.begin();
      c.map.emplace(key, n);
      ++c.eager;
    } else {
      Cached &e = it->second;
      c.order.splice(c.order.begin(), c.order, e.lru);
      ++e.sightings;
      if (e.exec) {
        exec = e.exec;
        ++c.replays;
      } else if (!e.refused) {
        capture = true;
      } else {
        ++c.eager;
      }
    }
  }
  if (exec) {
    if (how) *how = 2;
    const hipError_t err = hipGraphLaunch(exec, stream);
    BTR_REQUIRE(err == hipSuccess, "graph_run: hipGraphLaunch: %s", hipGetErrorString(err));
    return BTR_OK;
  }
  if (!capture) return body(stream);
  // ---- capture: the body's launches become nodes; nothing of it runs until the launch below
  hipGraph_t graph = nullptr;
  hipGraphExec_t ge = nullptr;
  int rc = BTR_OK;
  // (on a stream of our own: the caller's may be the legacy default stream, which cannot capture;
  // the body only records on the stream it is handed, and the graph is launched on the caller's)
  hipStream_t cs = capture_stream();
  hipError_t err = cs ? hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed)
                      : hipErrorInvalidResourceHandle;
  if (err == hipSuccess) {
    ++g_capturing;
    rc = body(cs);
    --g_capturing;
    err = hipStreamEndCapture(cs, &graph);
    if (err == hipSuccess && rc == BTR_OK && graph)
      err = hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0);
  }
  const bool ok = err == hipSuccess && rc == BTR_OK && ge;
  {
    std::lock_guard<std::mutex> lock(c.mu);
    auto it = c.map.find(key);
    if (it == c.map.end()) {   // evicted meanwhile by another thread
      c.order.push_front(key);
      Cached n;
      n.sightings = 2;
      n.lru = c.order.begin();
      it = c.map.emplace(key, n).first;
    }
    if (ok && !it->second.exec) {
      it->second.exec = ge;
      it->second.graph = graph;
      ++c.captures;
    } else if (!ok) {
      it->second.refused = true;
    }
  }
  if (!ok) {
    static bool told = false;
    if (!told) {   // once: the calls keep working, launch by launch
      told = true;
      fprintf(stderr, "libbtr_pointnet2: HIP-graph capture failed (%s; body rc %d: %s): the call "
                      "sequence stays on single launches\n",
              hipGetErrorString(err), rc, rc != BTR_OK ? btr_last_error() : "ok");
    }
    (void)hipGetLastError();
    if (ge) (void)hipGraphExecDestroy(ge);
    if (graph) (void)hipGraphDestroy(graph);
    if (rc != BTR_OK) return rc;   // the body's own complaint (its message is set)
    return body(stream);           // the capture machinery failed: issue the launches as always
  }
  if (how) *how = 1;
  err = hipGraphLaunch(ge, stream);
  BTR_REQUIRE(err == hipSuccess, "graph_run: hipGraphLaunch after capture: %s",
              hipGetErrorString(err));
  return BTR_OK;
}

// A side stream of a caller's stream, with events for the hand-overs, owned by the library and
// kept for the process's lifetime (one per device and caller stream).
SideLane *side_lane(hipStream_t main, int nstreams) {
  static std::mutex mu;
  static std::unordered_map<uint64_t, SideLane *> lanes;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  const uint64_t k = ((uint64_t)(uintptr_t)main << 6) ^ (uint64_t)dev;
  std::lock_guard<std::mutex> lock(mu);
  auto it = lanes.find(k);
  if (it != lanes.end()) {
    SideLane *l = it->second;   // (a second stream asked for later: made then)
    if (l && nstreams >= 2 && !l->s[1] &&
        hipStreamCreateWithFlags(&l->s[1], hipStreamNonBlocking) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    return l;
  }
  SideLane *l = new SideLane();
  // (stream priorities measured no different: lowest 3.52 ms, default 3.47, highest 3.51 for the
  // decoder stack's backward, profiles/r06_gf_lanes.txt)
  bool ok = hipStreamCreateWithFlags(&l->s[0], hipStreamNonBlocking) == hipSuccess &&
            (nstreams < 2 ||
             hipStreamCreateWithFlags(&l->s[1], hipStreamNonBlocking) == hipSuccess);
  for (int i = 0; ok && i < SideLane::kEvents; ++i)
    ok = hipEventCreateWithFlags(&l->ev[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    delete l;
    l = nullptr;
  }
  lanes.emplace(k, l);
  return l;
}

}  // namespace btr

extern "C" {

// (replays, captures, calls issued launch by launch) since the library was loaded
void btr_graph_stats(long long *replays, long long *captures, long long *eager) {
  btr::Cache &c = btr::cache();
  std::lock_guard<std::mutex> lock(c.mu);
  if (replays) *replays = c.replays;
  if (captures) *captures = c.captures;
  if (eager) *eager = c.eager;
}

// forget every captured graph (tests; a caller about to free the buffers its graphs point into)
void btr_graph_clear(void) {
  btr::Cache &c = btr::cache();
  std::lock_guard<std::mutex> lock(c.mu);
  for (auto &kv : c.map) btr::drop(kv.second);
  c.map.clear();
  c.order.clear();
}

}  // extern "C"
