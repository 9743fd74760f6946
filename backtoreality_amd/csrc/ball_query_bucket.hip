// ball_query_bucket.hip -- ball query over the Hilbert-sorted buckets the large-scene FPS has
// already built for the same cloud (fps_bucket.hip): no spatial structure of its own.
//
// SA1 of the benchmark step samples 2048 centres of a 40 000-point scene with the bucketed
// FPS and then ball-queries the SAME cloud around them.  ball_query_grid.hip counting-sorts the
// cloud a second time (47 us for the build, 38 us for the query itself) although the FPS
// workspace still holds the cloud sorted along a Hilbert curve in buckets of 64 points --
// x[64] y[64] z[64] k[64] per bucket, k = original index.  Here:
//   1. the bounding box of every bucket: left in the FPS workspace by the FPS kernel, which
//      needs them for its own pruning (internal.hpp fps_boxes_lookup; bqb_box_kernel -- one wave
//      per bucket: three coalesced 256-byte loads and a wave min/max -- when another FPS kernel
//      ran), and of every SUPER-bucket of 16 consecutive buckets (consecutive along the curve =
//      spatially compact), built by every query workgroup in its prologue;
//   2. bqb_query_kernel, one wave per centre: box test of the <= 64 super-buckets (one per
//      lane, boxes in LDS), then of the 16 buckets of every touched one; the points of the
//      surviving buckets (four coalesced loads per bucket) are tested with the reference's
//      exact f32 expression and mark bit `index` in a per-wave LDS bitmap with a one-bit-per-
//      word summary on top; the summary is walked in index order, so the FIRST nsample indices
//      come out without a sort and without touching the empty 95 % of the bitmap.
// A point can only pass d2 < r^2 if its bucket's box is within r of the centre: the box test
// uses the exact squared distance to the box in f32 with a relative slack of 1e-5 on r^2
// (conservative: it may only admit extra buckets), so the hit set is the reference's, bit for
// bit (ball_query_gpu.cu:14-49).  Workgroups are mapped so that all centres of a scene run
// on one XCD (blockIdx % B = scene): the scene's 640 KB sorted copy is fetched into that L2
// once.
#include <algorithm>
#include <cmath>

#include "internal.hpp"

namespace btr {

constexpr int kSuper = 16;          // buckets per super-bucket
constexpr int kMaxSupers = 128;     // super-buckets per scene the query kernel can hold
constexpr int kBqbWaves = 4;
// candidate buckets loaded per dependent round trip: template parameter kTrip (4; 2 and 8 measured
// the same)

__device__ __forceinline__ float wave_min_f32(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}

// boxes[scene][nb]: one wave per bucket
__global__ __launch_bounds__(256) void bqb_box_kernel(int np, int nb,
                                                      const float *__restrict__ spts,
                                                      Box8 *__restrict__ boxes) {
  const int bi = blockIdx.y, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= nb) return;
  const float *bp = spts + (size_t)bi * np * 4 + (size_t)b * 256 + lane;
  const bool real = __float_as_int(bp[192]) >= 0;  // padding slots carry index -1
  const float x = bp[0], y = bp[64], z = bp[128];
  const float big = 3.0e38f;
  // six reductions interleaved (independent chains)
  float lo[3] = {real ? x : big, real ? y : big, real ? z : big};
  float hi[3] = {real ? x : -big, real ? y : -big, real ? z : -big};
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
    }
  if (lane == 0)
    boxes[(size_t)bi * nb + b] = Box8{lo[0], lo[1], lo[2], 0.f, hi[0], hi[1], hi[2], 0.f};
}

// squared distance from p to the box (0 inside); plain f32, only used as a conservative cull
__device__ __forceinline__ float box_d2(const Box8 &q, float x, float y, float z) {
  const float ex = fmaxf(fmaxf(q.x0 - x, x - q.x1), 0.f);
  const float ey = fmaxf(fmaxf(q.y0 - y, y - q.y1), 0.f);
  const float ez = fmaxf(fmaxf(q.z0 - z, z - q.z1), 0.f);
  return ex * ex + ey * ey + ez * ez;
}

// Dynamic LDS per workgroup: nsup Box8 (super boxes) + kBqbWaves * (words + twords) bitmap
// words (zero on entry, restored to zero after every centre).
template <int kTrip>
__global__ __launch_bounds__(kBqbWaves * 64) void bqb_query_kernel(
    int B, int n, int np, int nb, int nsup, int m, int nsample, int words, int twords,
    float radius2, float cull2, const float *__restrict__ new_xyz,
    const float *__restrict__ spts, const Box8 *__restrict__ boxes, unsigned box_epoch,
    int *__restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int stale;   // some box of this scene does not carry the expected stamp
  Box8 *sb = reinterpret_cast<Box8 *>(smem);
  unsigned *maps = reinterpret_cast<unsigned *>(smem + sizeof(Box8) * nsup);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bi = blockIdx.x % B;            // all workgroups of a scene share blockIdx % B:
  const int chunk = blockIdx.x / B;         // with B a multiple or divisor of 8 -> one XCD set
  const int nchunks = gridDim.x / B;
  unsigned *bm = maps + (size_t)wave * (words + twords);
  unsigned *top = bm + words;
  for (int w = lane; w < words + twords; w += 64) bm[w] = 0u;
  if (threadIdx.x == 0) stale = 0;
  __syncthreads();
  // super-bucket boxes: the union of 16 consecutive bucket boxes, built here (640 B of L2 reads
  // per super-bucket and workgroup; was a launch of its own).  Boxes the FPS kernel left behind
  // (box_epoch != 0) are only trusted when every one of the scene carries that launch's stamp.
  const float *sp = spts + (size_t)bi * np * 4;
  bool bad = false;
  for (int s = threadIdx.x; s < nsup; s += kBqbWaves * 64) {
    const Box8 *q = boxes + (size_t)bi * nb + (size_t)s * kSuper;
    Box8 a = Box8{3.0e38f, 3.0e38f, 3.0e38f, 0.f, -3.0e38f, -3.0e38f, -3.0e38f, 0.f};
    for (int t = 0; t < min(kSuper, nb - s * kSuper); ++t) {
      a.x0 = fminf(a.x0, q[t].x0); a.y0 = fminf(a.y0, q[t].y0); a.z0 = fminf(a.z0, q[t].z0);
      a.x1 = fmaxf(a.x1, q[t].x1); a.y1 = fmaxf(a.y1, q[t].y1); a.z1 = fmaxf(a.z1, q[t].z1);
      bad |= box_epoch != 0u &&
             (__float_as_uint(q[t].p0) != box_epoch ||
              __float_as_uint(q[t].p1) != box_stamp_pos(bi * nb + s * kSuper + t));
    }
    sb[s] = a;
  }
  if (bad) atomicOr(&stale, 1);
  __syncthreads();
  const bool trusted = stale == 0;
  if (!trusted) {
    // the boxes are not the noted launch's: bound every super-bucket by its own (up to 1 024)
    // points instead and cull no single bucket
    for (int s = threadIdx.x; s < nsup; s += kBqbWaves * 64) {
      Box8 a = Box8{3.0e38f, 3.0e38f, 3.0e38f, 0.f, -3.0e38f, -3.0e38f, -3.0e38f, 0.f};
      for (int t = 0; t < min(kSuper, nb - s * kSuper); ++t) {
        const float *bp = sp + (size_t)(s * kSuper + t) * 256;
        for (int i = 0; i < 64; ++i) {
          if (__float_as_int(bp[192 + i]) < 0) continue;
          a.x0 = fminf(a.x0, bp[i]); a.x1 = fmaxf(a.x1, bp[i]);
          a.y0 = fminf(a.y0, bp[64 + i]); a.y1 = fmaxf(a.y1, bp[64 + i]);
          a.z0 = fminf(a.z0, bp[128 + i]); a.z1 = fmaxf(a.z1, bp[128 + i]);
        }
      }
      sb[s] = a;
    }
    __syncthreads();
  }

  const Box8 *bx = boxes + (size_t)bi * nb;
  for (int j = chunk * kBqbWaves + wave; j < m; j += nchunks * kBqbWaves) {
    const float *c = new_xyz + ((size_t)bi * m + j) * 3;
    const float cx = c[0], cy = c[1], cz = c[2];
    // ---- level 1: super-buckets (lane = super-bucket; up to two rounds of 64)
    for (int s0 = 0; s0 < nsup; s0 += 64) {
      const int s = s0 + lane;
      const bool near = s < nsup && box_d2(sb[s], cx, cy, cz) < cull2;
      unsigned long long smask = __ballot(near);
      // ---- level 2: the 16 buckets of each touched super-bucket, four supers per round
      while (smask) {
        int sup[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          sup[q] = smask ? s0 + __builtin_ctzll(smask) : -1;
          smask &= smask - 1;
        }
        const int mys = sup[lane >> 4];
        const int b = mys * kSuper + (lane & 15);
        const bool cand =
            mys >= 0 && b < nb && (!trusted || box_d2(bx[b], cx, cy, cz) < cull2);
        unsigned long long bmask = __ballot(cand);
        // ---- candidates: 64 points per bucket, kTrip buckets per trip.  The kernel is bound by
        // its dependent L2 round trips (SQ counters, profiles/r04_bq_counters.md: 74 % of the
        // wave-cycles parked in s_waitcnt, 18 % issuing): a centre has 10-14 candidate buckets,
        // two per trip were 5-7 round trips in a row, four are 3 (16 loads in flight per lane)
        while (bmask) {
          int bb[kTrip];
          bool on[kTrip];
#pragma unroll
          for (int t = 0; t < kTrip; ++t) {
            on[t] = bmask != 0;
            const int sl = on[t] ? __builtin_ctzll(bmask) : 0;
            bb[t] = on[t] ? __builtin_amdgcn_readlane(b, sl) : bb[0];   // (a valid bucket to read)
            bmask &= bmask - 1;   // (0 stays 0)
          }
          float px[kTrip], py[kTrip], pz[kTrip];
          int pk[kTrip];
#pragma unroll
          for (int t = 0; t < kTrip; ++t) {
            const float *pp = sp + (size_t)bb[t] * 256 + lane;
            px[t] = pp[0];
            py[t] = pp[64];
            pz[t] = pp[128];
            pk[t] = __float_as_int(pp[192]);
          }
#pragma unroll
          for (int t = 0; t < kTrip; ++t) {
            if (on[t] && pk[t] >= 0 && sq3(cx - px[t], cy - py[t], cz - pz[t]) < radius2) {
              atomicOr(&bm[pk[t] >> 5], 1u << (pk[t] & 31));
              atomicOr(&top[pk[t] >> 10], 1u << ((pk[t] >> 5) & 31));
            }
          }
        }
      }
    }
    // ---- first nsample set bits in index order: lane l owns summary words l, l + 64, ...
    int *row = idx + ((size_t)bi * m + j) * nsample;
    int base = 0;          // hits in the summary words of earlier rounds
    int first = 0x7fffffff;
    for (int t0 = 0; t0 < twords; t0 += 64) {
      const int t = t0 + lane;
      unsigned tw = t < twords ? top[t] : 0u;
      int cnt = 0;
      for (unsigned v = tw; v; v &= v - 1) cnt += __builtin_popcount(bm[(t << 5) + __builtin_ctz(v)]);
      int incl = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
      }
      int pos = base + incl - cnt;
      base += __builtin_amdgcn_readlane(incl, 63);
      if (tw) top[t] = 0u;
      while (tw) {
        const int w = (t << 5) + __builtin_ctz(tw);
        tw &= tw - 1;
        unsigned v = bm[w];
        bm[w] = 0u;  // restore the bitmap for the next centre
        while (v) {
          const int k = (w << 5) + __builtin_ctz(v);
          v &= v - 1;
          first = min(first, k);
          if (pos < nsample) row[pos] = k;
          ++pos;
        }
      }
    }
    // padding: the smallest hit fills the rest; no hit -> zeros (ball_query_gpu.cu:39-43)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) first = min(first, __shfl_xor(first, o));
    const int fill = base == 0 ? 0 : first;
    for (int l = min(base, nsample) + lane; l < nsample; l += 64) row[l] = fill;
  }
}

// A second form of the query kernel (all prologue passes in flight, the box tests of four centres
// in one trip, candidate buckets eight per trip with the next trip issued before the current one is
// tested) produced the same output and was SLOWER: 34.6 us against 28.2 at its natural 159
// registers, 31.1 - 32.6 us in three shapes that fit four workgroups per CU; a third form with two
// centres' loads in one basic block 47 - 63 us; centres in Morton order no change; centres drawn
// from a per-scene counter 196 us (returning device-scope atomics serialise across XCDs):
// profiles/r05_bq_form_ab*.txt.  Neither fewer dependent trips nor more loads in flight nor
// locality shorten the kernel, and it is far from the L1 / L2 rates (10.8 / 3.2 TB/s,
// profiles/r05_bq_l2.md) and from the issue rate (0.16 instructions per SIMD-cycle); its time
// follows the number of vector-memory requests.  The second form was removed in round 6.

struct BqbPlan {
  int nb, np, nsup, words, twords;
  size_t box_bytes, sbox_bytes, lds;
};

static BqbPlan bqb_plan(int b, int n) {
  BqbPlan p;
  p.nb = cdiv(n, 64);
  p.np = p.nb * 64;
  p.nsup = cdiv(p.nb, kSuper);
  p.words = cdiv(n, 32);
  p.twords = cdiv(p.words, 32);
  p.box_bytes = sizeof(Box8) * (size_t)b * p.nb;
  p.sbox_bytes = sizeof(Box8) * (size_t)b * p.nsup;
  p.lds = sizeof(Box8) * p.nsup + sizeof(unsigned) * (size_t)kBqbWaves * (p.words + p.twords);
  return p;
}

bool bq_bucket_supported(int n, int m, int nsample) {
  if (n <= 0 || m <= 0 || nsample <= 0) return false;
  const BqbPlan p = bqb_plan(1, n);
  return p.nsup <= kMaxSupers && p.lds <= 120 * 1024;
}

size_t bq_bucket_workspace_bytes(int b, int n) {
  const BqbPlan p = bqb_plan(b, n);
  return p.box_bytes + p.sbox_bytes;
}

// spts: the bucket-SoA copy at the start of the FPS workspace (fps_bucket.hip fps_plan)
int bq_bucket_launch(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                     const void *fps_workspace, int *idx, void *ws, size_t ws_bytes,
                     hipStream_t s) {
  const BqbPlan p = bqb_plan(b, n);
  BTR_REQUIRE(fps_workspace && ws && ws_bytes >= p.box_bytes + p.sbox_bytes,
              "ball_query(buckets): workspace too small");
  const float *spts = (const float *)fps_workspace;
  // bucket boxes: left behind by the FPS kernel that sorted this cloud (internal.hpp), else one
  // wave per bucket here
  unsigned epoch = 0u;   // 0: boxes of the own pass below, nothing to verify
  const Box8 *boxes = fps_boxes_lookup(fps_workspace, b, n, &epoch);
  if (!boxes) {
    Box8 *own = (Box8 *)ws;
    hipLaunchKernelGGL(bqb_box_kernel, dim3(cdiv(p.nb, 4), b), dim3(256), 0, s, p.np, p.nb, spts,
                       own);
    boxes = own;
  }
  // (candidate buckets per L2 trip: 2 / 4 / 8 measured 51 - 53 us per call in the loop at every
  // width -- seven waves per SIMD already hide the trips; four it is)
  const void *fn = (const void *)bqb_query_kernel<4>;
  static size_t lds_set[64] = {};   // (per device: the attribute belongs to the function on a device)
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = dev >= 0 && dev < 64 ? dev : 0;
  if (p.lds > lds_set[dev] && p.lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
    if (e != hipSuccess)
      return fail((int)e, "ball_query(buckets) attr: %s", hipGetErrorString(e));
    lds_set[dev] = p.lds;
  }
  const float radius2 = radius * radius;          // ball_query_gpu.cu:27
  const float cull2 = radius2 * 1.00001f + 1e-30f;  // conservative box cull (see the header)
  // one round of resident workgroups: a workgroup's time is its waves' chains of dependent L2
  // round trips (~4.5 us per centre), so the 2 048 workgroups of the benchmark shape -- 1 792 fit
  // the chip at once (LDS: seven per CU) -- spent a second round on the last 256 of them
  const int by_lds = (int)std::max<size_t>(1, (size_t)(160 * 1024) / std::max<size_t>(p.lds + 64, 1));
  // (at most four per CU: inside the training loop the query shares the chip with the step's
  // kernels and their LDS -- benchmark shape, in the loop: 87 us at seven per CU, 69 at four, 77 at
  // three; alone 43 / 45 / 51.  The Matterport-shaped scenes fit three: 56 -> 37 us in the loop)
  const int per_cu = std::min(by_lds, 4);
  const int budget = per_cu * grid_cus();
  const int chunks = std::max(1, std::min(cdiv(m, kBqbWaves), budget / std::max(1, b)));
  hipLaunchKernelGGL(bqb_query_kernel<4>, dim3(chunks * b), dim3(kBqbWaves * 64), p.lds, s, b, n,
                     p.np, p.nb, p.nsup, m, nsample, p.words, p.twords, radius2, cull2, new_xyz,
                     spts, boxes, epoch, idx);
  return check_launch("ball_query(buckets)");
}

}  // namespace btr

using namespace btr;

extern "C" {

size_t btr_ball_query_buckets_workspace_bytes(int b, int n, int m, int nsample) {
  if (b <= 0 || !bq_bucket_supported(n, m, nsample)) return 0;
  return bq_bucket_workspace_bytes(b, n);
}

int btr_ball_query_buckets(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                           const void *fps_workspace, int *idx, void *workspace,
                           size_t workspace_bytes, btr_stream_t stream) {
  if (b <= 0 || m <= 0 || nsample <= 0) return BTR_OK;
  BTR_REQUIRE(idx && new_xyz, "ball_query(buckets): null pointer");
  BTR_REQUIRE(bq_bucket_supported(n, m, nsample), "ball_query(buckets): n=%d not supported", n);
  hipEvent_t *ev = bq_call_events();   // (btr_ball_query_time_next)
  const bool timed = ev[0] != nullptr;
  if (timed) (void)hipEventRecord(ev[0], as_stream(stream));
  const int rc = bq_bucket_launch(b, n, m, radius, nsample, new_xyz, fps_workspace, idx,
                                  workspace, workspace_bytes, as_stream(stream));
  if (timed) {
    (void)hipEventRecord(ev[1], as_stream(stream));
    ev[0] = ev[1] = nullptr;
  }
  return rc;
}

}  // extern "C"
