// ball_query.hip -- radius search "first nsample neighbours in index order" for gfx950.
//
// Replaces src/ball_query_gpu.cu:14-59 of the reference.  Built with -ffp-contract=off: the
// f32 test `d2 < radius2` must round exactly as the reference source writes it.
//
// Brute-force path (this file, v1): lanes = 64 consecutive centres, the point stream is
// wave-uniform (scalar loads broadcast through the scalar cache), the index range is cut into
// S segments scanned by independent waves so the launch fills the chip, and a merge pass
// concatenates the per-segment hit lists in segment order, stops at nsample, and applies the
// reference's padding rule (first hit repeated; empty row = zeros).
#include <algorithm>
#include <cstdlib>

#include "internal.hpp"

namespace btr {

// Scan one segment [k0, k1) for the 64 centres of this wave.
//   DIRECT (S == 1): write the final padded rows into idx.
//   else: write the raw hits of the segment into seg_hits[(row*S + seg)*nsample + i] and the
//         count into seg_cnt[row*S + seg].
template <bool DIRECT>
__global__ __launch_bounds__(64) void bq_scan_kernel(int n, int m, int nsample, int S,
                                                     int seg_len, float radius2,
                                                     const float *__restrict__ new_xyz,
                                                     const float *__restrict__ xyz,
                                                     int *__restrict__ idx,
                                                     int *__restrict__ seg_hits,
                                                     int *__restrict__ seg_cnt) {
  const int bi = blockIdx.z;
  const int seg = blockIdx.y;
  const int j = blockIdx.x * 64 + threadIdx.x;
  const bool active = j < m;
  const int jj = active ? j : m - 1;
  xyz += (size_t)bi * n * 3;
  const float *c = new_xyz + ((size_t)bi * m + jj) * 3;
  const float new_x = c[0], new_y = c[1], new_z = c[2];
  const size_t row = (size_t)bi * m + jj;
  int *out = DIRECT ? idx + row * nsample : seg_hits + (row * S + seg) * nsample;

  const int k0 = seg * seg_len;
  const int k1 = min(n, k0 + seg_len);
  int cnt = active ? 0 : nsample;
  int first = 0;
  for (int kb = k0; kb < k1; kb += 64) {
    if (__all(cnt >= nsample)) break;  // ball_query_gpu.cu:32 early exit, per wave
    const int ke = min(k1, kb + 64);
    for (int k = kb; k < ke; ++k) {
      const float x = xyz[k * 3 + 0], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
      const float d2 = sq3(new_x - x, new_y - y, new_z - z);
      if (d2 < radius2 && cnt < nsample) {
        if (cnt == 0) first = k;
        out[cnt] = k;
        ++cnt;
      }
    }
  }
  if (!active) return;
  if (DIRECT) {
    const int fill = cnt == 0 ? 0 : first;  // ball_query_gpu.cu:39-43 / zero-initialised row
    for (int l = cnt; l < nsample; ++l) out[l] = fill;
  } else {
    seg_cnt[row * S + seg] = cnt;
  }
}

// One thread per output element (row, l): locate the l-th hit across the row's segments.
__global__ __launch_bounds__(256) void bq_merge_kernel(long long rows, int nsample, int S,
                                                       const int *__restrict__ seg_hits,
                                                       const int *__restrict__ seg_cnt,
                                                       int *__restrict__ idx) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * nsample) return;
  const long long row = t / nsample;
  const int l = (int)(t - row * nsample);
  const int *cnt = seg_cnt + row * S;
  const int *hits = seg_hits + row * S * (long long)nsample;
  int acc = 0, val = 0, first = 0;
  bool have_first = false, found = false;
  for (int s = 0; s < S; ++s) {
    const int c = cnt[s];
    if (c > 0 && !have_first) {
      first = hits[(long long)s * nsample];
      have_first = true;
    }
    if (!found && l < acc + c) {
      val = hits[(long long)s * nsample + (l - acc)];
      found = true;
    }
    acc += c;
    if (acc >= nsample) break;
  }
  idx[t] = found ? val : (have_first ? first : 0);
}

// Small scenes: one wave per centre, lanes over 64 consecutive points (coalesced, the scene
// stays in L1/L2), hits compacted in index order with ballot + prefix popcount, early exit
// once nsample hits are found (ball_query_gpu.cu:32).
__global__ __launch_bounds__(256) void bq_wave_kernel(int n, int m, int nsample, float radius2,
                                                      const float *__restrict__ new_xyz,
                                                      const float *__restrict__ xyz,
                                                      int *__restrict__ idx) {
  const int bi = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  xyz += (size_t)bi * n * 3;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int j = blockIdx.x * 4 + wave; j < m; j += gridDim.x * 4) {
    const float *c = new_xyz + ((size_t)bi * m + j) * 3;
    const float new_x = c[0], new_y = c[1], new_z = c[2];
    int *row = idx + ((size_t)bi * m + j) * nsample;
    int cnt = 0, first = 0;
    for (int kb = 0; kb < n && cnt < nsample; kb += 64) {
      const int k = kb + lane;
      bool hit = false;
      if (k < n) {
        const float x = xyz[k * 3 + 0], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
        const float d2 = sq3(new_x - x, new_y - y, new_z - z);
        hit = d2 < radius2;
      }
      const unsigned long long mask = __ballot(hit);
      if (mask) {
        if (cnt == 0) first = kb + __builtin_ctzll(mask);
        const int pos = cnt + __builtin_popcountll(mask & lt);
        if (hit && pos < nsample) row[pos] = k;
        cnt += __builtin_popcountll(mask);
      }
    }
    const int fill = cnt == 0 ? 0 : first;
    for (int l = min(cnt, nsample) + lane; l < nsample; l += 64) row[l] = fill;
  }
}

constexpr int kBqWaveMaxN = 8192;

struct BqPlan {
  int mtiles, S, seg_len;
  size_t hits_bytes, cnt_bytes;
};

static BqPlan bq_plan(int b, int n, int m, int nsample) {
  BqPlan p;
  p.mtiles = cdiv(m, 64);
  if (n < kBqWaveMaxN) {  // wave-per-centre kernel: no scratch
    p.S = 1;
    p.seg_len = std::max(n, 1);
    p.hits_bytes = p.cnt_bytes = 0;
    return p;
  }
  const long long waves = (long long)b * p.mtiles;
  int S = (int)std::max<long long>(1, 2048 / std::max<long long>(1, waves));
  S = std::min(S, std::max(1, n / 512));
  S = std::min(S, 32);
  p.seg_len = cdiv(cdiv(n, S), 64) * 64;
  p.S = cdiv(n, p.seg_len);
  if (p.S <= 1) {
    p.S = 1;
    p.seg_len = std::max(n, 1);
    p.hits_bytes = p.cnt_bytes = 0;
  } else {
    p.hits_bytes = sizeof(int) * (size_t)b * m * p.S * nsample;
    p.cnt_bytes = sizeof(int) * (size_t)b * m * p.S;
  }
  return p;
}

// ball_query_grid.hip
bool bq_grid_supported(int n, int m, int nsample);
size_t bq_grid_workspace_bytes(int b, int n);
int bq_grid_launch(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                   const float *xyz, int *idx, void *ws, size_t ws_bytes, hipStream_t s);

// BTR_BQ_IMPL=brute forces the brute-force scan for large scenes (A/B and cross-checks).
static bool bq_use_grid(int n, int m, int nsample) {
  const char *e = getenv("BTR_BQ_IMPL");
  if (e && e[0] == 'b') return false;
  return bq_grid_supported(n, m, nsample);
}

}  // namespace btr

namespace btr {
hipEvent_t *bq_call_events() {
  static thread_local hipEvent_t ev[2] = {nullptr, nullptr};
  return ev;
}
}  // namespace btr

using namespace btr;

extern "C" {

void btr_ball_query_time_next(void *start_event, void *stop_event) {
  bq_call_events()[0] = (hipEvent_t)start_event;
  bq_call_events()[1] = (hipEvent_t)stop_event;
}

size_t btr_ball_query_workspace_bytes(int b, int n, int m, int nsample) {
  if (b <= 0 || n <= 0 || m <= 0 || nsample <= 0) return 0;
  if (bq_use_grid(n, m, nsample)) return bq_grid_workspace_bytes(b, n);
  const BqPlan p = bq_plan(b, n, m, nsample);
  return p.hits_bytes + p.cnt_bytes;
}

static int ball_query_ws_impl(int b, int n, int m, float radius, int nsample,
                              const float *new_xyz, const float *xyz, int *idx, void *workspace,
                              size_t workspace_bytes, btr_stream_t stream);

int btr_ball_query_ws(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                      const float *xyz, int *idx, void *workspace, size_t workspace_bytes,
                      btr_stream_t stream) {
  hipEvent_t *ev = bq_call_events();
  const bool timed = ev[0] && n > 4096;   // (the hook is for the large-scene query)
  if (timed) (void)hipEventRecord(ev[0], as_stream(stream));
  const int rc = ball_query_ws_impl(b, n, m, radius, nsample, new_xyz, xyz, idx, workspace,
                                    workspace_bytes, stream);
  if (timed) {
    (void)hipEventRecord(ev[1], as_stream(stream));
    ev[0] = ev[1] = nullptr;
  }
  return rc;
}

static int ball_query_ws_impl(int b, int n, int m, float radius, int nsample,
                              const float *new_xyz, const float *xyz, int *idx, void *workspace,
                              size_t workspace_bytes, btr_stream_t stream) {
  if (b <= 0 || m <= 0 || nsample <= 0) return BTR_OK;
  BTR_REQUIRE(idx, "ball_query: null output");
  hipStream_t s = as_stream(stream);
  if (n <= 0) {  // no points: every row stays zero (ball_query.cpp:24-26)
    hipError_t e = hipMemsetAsync(idx, 0, sizeof(int) * (size_t)b * m * nsample, s);
    return e == hipSuccess ? BTR_OK : fail((int)e, "ball_query memset: %s", hipGetErrorString(e));
  }
  BTR_REQUIRE(new_xyz && xyz, "ball_query: null input");
  if (bq_use_grid(n, m, nsample))
    return bq_grid_launch(b, n, m, radius, nsample, new_xyz, xyz, idx, workspace,
                          workspace_bytes, s);
  const BqPlan p = bq_plan(b, n, m, nsample);
  const float radius2 = radius * radius;  // ball_query_gpu.cu:27
  if (n < kBqWaveMaxN) {
    const int gx = std::min(cdiv(m, 4), 1024);
    hipLaunchKernelGGL(bq_wave_kernel, dim3(gx, b), dim3(256), 0, s, n, m, nsample, radius2,
                       new_xyz, xyz, idx);
    return check_launch("ball_query(wave)");
  }
  if (p.S == 1) {
    hipLaunchKernelGGL((bq_scan_kernel<true>), dim3(p.mtiles, 1, b), dim3(64), 0, s, n, m,
                       nsample, 1, p.seg_len, radius2, new_xyz, xyz, idx, (int *)nullptr,
                       (int *)nullptr);
    return check_launch("ball_query");
  }
  BTR_REQUIRE(workspace && workspace_bytes >= p.hits_bytes + p.cnt_bytes,
              "ball_query: workspace of %zu bytes required, got %zu",
              p.hits_bytes + p.cnt_bytes, workspace_bytes);
  int *seg_hits = (int *)workspace;
  int *seg_cnt = (int *)((char *)workspace + p.hits_bytes);
  hipLaunchKernelGGL((bq_scan_kernel<false>), dim3(p.mtiles, p.S, b), dim3(64), 0, s, n, m,
                     nsample, p.S, p.seg_len, radius2, new_xyz, xyz, idx, seg_hits, seg_cnt);
  int rc = check_launch("ball_query(scan)");
  if (rc) return rc;
  const long long rows = (long long)b * m;
  hipLaunchKernelGGL(bq_merge_kernel, dim3(cdiv(rows * nsample, 256)), dim3(256), 0, s, rows,
                     nsample, p.S, seg_hits, seg_cnt, idx);
  return check_launch("ball_query(merge)");
}

int btr_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                   const float *xyz, int *idx, btr_stream_t stream) {
  const size_t ws = btr_ball_query_workspace_bytes(b, n, m, nsample);
  void *w = nullptr;
  hipStream_t s = as_stream(stream);
  if (ws) {
    hipError_t e = hipMallocAsync(&w, ws, s);
    if (e != hipSuccess) return fail((int)e, "ball_query workspace: %s", hipGetErrorString(e));
  }
  const int rc = btr_ball_query_ws(b, n, m, radius, nsample, new_xyz, xyz, idx, w, ws, stream);
  if (w) (void)hipFreeAsync(w, s);
  return rc;
}

}  // extern "C"
