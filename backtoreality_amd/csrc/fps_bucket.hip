// fps_bucket.hip -- furthest point sampling for large scenes (n > 4096) on gfx950:
// Hilbert-curve buckets + exact bounding-box pruning, one workgroup (one CU) per scene.
//
// Why: FPS is a chain of m-1 dependent arg-max steps; a 40 000-point scene does not fit one
// CU's registers (640 KB of x,y,z,min-dist), and a cross-CU exchange costs >= 1 us per step
// (MI355X_MICROARCH.md price list), so the per-step work is cut instead:
//   * fps_sortm_*_kernel counting-sort the scene by a 15-bit Hilbert cell (32^3 grid) into
//     buckets of 64 consecutive points = one wave-wide load each;
//   * fps_bucket_kernel keeps, per bucket, its bounding box and its current best key
//     (max min-dist, tie key, coordinates) lane-parallel in VGPRs.  A new sample s only
//     touches buckets whose box lower bound d_box(s) is below the bucket's max min-dist.
// Exactness of the pruning: d_box is evaluated with the SAME f32 expression as the point
// distance, on the per-axis clamp of s to the box.  Every f32 operation involved (subtract,
// square, add) is monotone in |x2 - x1|, so d_box <= d(p) holds for the ROUNDED values of
// every point p of the bucket; if d_box >= max min-dist, min(d, temp) leaves every temp
// unchanged, bit for bit.  The bucket order only affects speed: keys (d2, tie key) are unique
// per point, so the arg-max -- and therefore the result -- is independent of the sort.
// Selection rule / tie-break: identical to sampling.hip (reference sampling_gpu.cu:74-178).
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <vector>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "internal.hpp"

namespace btr {

// Measurement hook (btr_fps_time_next_kernel): an event pair recorded immediately around the
// next launch of the sampling kernel itself (not its sort launches) by this host thread.
hipEvent_t *fps_kernel_events() {
  static thread_local hipEvent_t ev[2] = {nullptr, nullptr};
  return ev;
}

constexpr int kSortThreads = 1024;
constexpr int kGridBits = 5;                 // 32^3 cells (64^3 was measured: 9.4 vs 9.6 touched
constexpr int kCells = 1 << (3 * kGridBits);  // buckets per sample, but +0.3 ms of sort)

__device__ __forceinline__ unsigned spread3(unsigned v) {  // <= 10 bits -> every third bit
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// Cell index of a point on a (2^bits)^3 grid along a space-filling curve.  Hilbert order
// (Skilling's transpose algorithm) instead of Morton order: consecutive cells are always face
// neighbours, so a bucket (64 consecutive points) never straddles one of the Z curve's jumps
// and its bounding box is tighter -- 9.6 instead of 13.1 buckets pass the box test per sample
// (2.63 -> 2.25 ms on 8 x 40000 -> 2048).  The order only affects speed, never the result (see
// the header).

__device__ __forceinline__ int hilbert3(unsigned x0, unsigned x1, unsigned x2, int bits) {
  unsigned X[3] = {x0, x1, x2};
  const unsigned M = 1u << (bits - 1);
  for (unsigned Q = M; Q > 1u; Q >>= 1) {  // inverse undo excess work
    const unsigned P = Q - 1u;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (X[i] & Q) {
        X[0] ^= P;
      } else {
        const unsigned t = (X[0] ^ X[i]) & P;
        X[0] ^= t;
        X[i] ^= t;
      }
    }
  }
  X[1] ^= X[0];  // Gray encode
  X[2] ^= X[1];
  unsigned t = 0;
  for (unsigned Q = M; Q > 1u; Q >>= 1)
    if (X[2] & Q) t ^= Q - 1u;
  X[0] ^= t; X[1] ^= t; X[2] ^= t;
  // transposed index: bit k of X[i] is bit 3k + (2 - i) of the Hilbert index
  return (int)((spread3(X[0]) << 2) | (spread3(X[1]) << 1) | spread3(X[2]));
}

__device__ __forceinline__ int morton_cell(float x, float y, float z, float mnx, float mny,
                                           float mnz, float scale) {
  constexpr int bits = kGridBits, top = (1 << bits) - 1;  // scale = 2^bits / extent
  const int qx = min(top, max(0, (int)((x - mnx) * scale)));
  const int qy = min(top, max(0, (int)((y - mny) * scale)));
  const int qz = min(top, max(0, (int)((z - mnz) * scale)));
  return hilbert3((unsigned)qx, (unsigned)qy, (unsigned)qz, bits);
}

// Bucket storage: structure of arrays per bucket of 64 points -- x[64] y[64] z[64] k[64] (k =
// bits of the original index): every component of a bucket is one fully coalesced 256-byte
// wave load into its own register (a float4-per-point load needs 4 consecutive registers,
// and the register shuffles around a PREFETCHED float4 wait for its load).
__device__ __forceinline__ size_t soa_at(size_t pos, int comp) {
  return (pos >> 6) * 256 + (size_t)comp * 64 + (pos & 63);
}
__device__ __forceinline__ float4 soa_point(const float *sp, size_t pos) {
  return make_float4(sp[soa_at(pos, 0)], sp[soa_at(pos, 1)], sp[soa_at(pos, 2)],
                     sp[soa_at(pos, 3)]);
}

// The counting sort: output spts = the bucket-SoA points {x, y, z, bits(original index)}
// (read-only from here on; index -1 for padding), np = 64 * ceil(n / 64), in curve order.
// The running min-dists are the sampling kernel's own business (round 6: it derives their
// initial values from the points and keeps them in LDS; the tail that does not fit lives in a
// separate global array -- separate because a per-bucket write-back is then 2 full 128-B lines
// instead of 64 dwords strewn over 8 lines: ~490 against ~1050 cycles per dependent bucket load
// with 16 waves doing it, tools/probe/lat_probe.hip).
// MANY workgroups per scene (a one-workgroup version streamed the scene three times through
// a single CU: 157 us on 8 x 40000, all of it on the step's critical path).  meta[b] = 6 order-preserving uint keys (max of ~key(min), max of
// key(max)) accumulated with integer atomics; cells[b][32768] = histogram -> exclusive scan
// -> scatter cursor.  The order inside a cell depends on the atomics and cannot change the
// FPS result (see the header).
__device__ __forceinline__ unsigned f32_key(float f) {  // monotone float -> uint
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

struct SortBox {
  float mnx, mny, mnz, scale;
};
__device__ __forceinline__ SortBox sort_box(const unsigned *__restrict__ meta) {
  const float mnx = key_f32(~meta[0]), mny = key_f32(~meta[1]), mnz = key_f32(~meta[2]);
  const float mxx = key_f32(meta[3]), mxy = key_f32(meta[4]), mxz = key_f32(meta[5]);
  const float ext = fmaxf(fmaxf(mxx - mnx, mxy - mny), mxz - mnz);
  return SortBox{mnx, mny, mnz, ext > 0.f ? (float)(1 << kGridBits) / ext : 0.f};
}

__global__ __launch_bounds__(256) void fps_sortm_bbox_kernel(int n,
                                                             const float *__restrict__ dataset,
                                                             unsigned *__restrict__ meta) {
  __shared__ unsigned red[6][4];
  const int bi = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  dataset += (size_t)bi * n * 3;
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int k = blockIdx.x * 256 + tid; k < n; k += gridDim.x * 256) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = dataset[k * 3 + a];
      mn[a] = fminf(mn[a], v);
      mx[a] = fmaxf(mx[a], v);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
    }
    if (lane == 0) {
      red[a][wave] = ~f32_key(mn[a]);
      red[3 + a][wave] = f32_key(mx[a]);
    }
  }
  __syncthreads();
  if (tid < 6) {
    const unsigned v = max(max(red[tid][0], red[tid][1]), max(red[tid][2], red[tid][3]));
    atomicMax(&meta[(size_t)bi * 8 + tid], v);
  }
}

__global__ __launch_bounds__(256) void fps_sortm_hist_kernel(int n,
                                                             const float *__restrict__ dataset,
                                                             const unsigned *__restrict__ meta,
                                                             int *__restrict__ cells) {
  const int bi = blockIdx.y;
  dataset += (size_t)bi * n * 3;
  const SortBox bx = sort_box(meta + (size_t)bi * 8);
  for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
    const int c = morton_cell(dataset[k * 3], dataset[k * 3 + 1], dataset[k * 3 + 2], bx.mnx,
                              bx.mny, bx.mnz, bx.scale);
    atomicAdd(&cells[(size_t)bi * kCells + c], 1);
  }
}

// exclusive scan of the 32768 cell counts of one scene, in place
__global__ __launch_bounds__(kSortThreads) void fps_sortm_scan_kernel(int *__restrict__ cells) {
  __shared__ int wsum[16];
  int *hist = cells + (size_t)blockIdx.x * kCells;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int PER = kCells / kSortThreads;
  int local[PER];
  int sum = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    local[i] = hist[tid * PER + i];
    sum += local[i];
  }
  int incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int run = incl - sum;
  for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    hist[tid * PER + i] = run;
    run += local[i];
  }
}

__global__ __launch_bounds__(256) void fps_sortm_scatter_kernel(
    int n, int np, const float *__restrict__ dataset, const unsigned *__restrict__ meta,
    int *__restrict__ cells, float4 *__restrict__ spts) {
  const int bi = blockIdx.y;
  dataset += (size_t)bi * n * 3;
  float *sp = (float *)(spts + (size_t)bi * np);
  const SortBox bx = sort_box(meta + (size_t)bi * 8);
  for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
    const float x = dataset[k * 3], y = dataset[k * 3 + 1], z = dataset[k * 3 + 2];
    const int c = morton_cell(x, y, z, bx.mnx, bx.mny, bx.mnz, bx.scale);
    const int pos = atomicAdd(&cells[(size_t)bi * kCells + c], 1);
    sp[soa_at(pos, 0)] = x;
    sp[soa_at(pos, 1)] = y;
    sp[soa_at(pos, 2)] = z;
    sp[soa_at(pos, 3)] = __int_as_float(k);
  }
  // padding of the last bucket: index -1 (kept out of the bucket's bounding box; never
  // competes); the coordinates only have to be finite
  if (blockIdx.x == 0 && n + (int)threadIdx.x < np) {
    const int pos = n + threadIdx.x;
    sp[soa_at(pos, 0)] = dataset[0];
    sp[soa_at(pos, 1)] = dataset[1];
    sp[soa_at(pos, 2)] = dataset[2];
    sp[soa_at(pos, 3)] = __int_as_float(-1);
  }
}

__device__ __forceinline__ unsigned fps_tk2(int k, int bs, int log2bs, int cpb) {
  const unsigned r = log2bs == 0 ? 0u : (__brev((unsigned)(k & (bs - 1))) >> (32 - log2bs));
  return r * (unsigned)cpb + (unsigned)(k >> log2bs);
}

struct TieParams {
  int bs, log2bs, cpb;
};

// max over the wave of `hi`; winner = the lane with that hi and, on exact ties (duplicated
// points), the largest lo.  lo is only looked at when two lanes tie.  Returns the lane.
template <typename LoFn>
__device__ __forceinline__ int wave_argmax(unsigned hi, LoFn lo_of_lane, unsigned &mh) {
  mh = wave_max_u32(hi);
  const unsigned long long cand = __ballot(hi == mh);
  if (__builtin_popcountll(cand) == 1) return __builtin_ctzll(cand);
  const unsigned lo = (hi == mh) ? lo_of_lane() : 0u;
  const unsigned ml = wave_max_u32(lo);
  return __builtin_ctzll(__ballot(hi == mh && lo == ml));
}

struct BSlot {  // (x, y) on an even dword pair, like the float4 point: packed f32 math without
  unsigned hi, lo;  // register shuffles (a shuffle of a prefetched point waits for its load)
  float x, y, z;
  int k;
  int pad0, pad1;
};

__device__ __forceinline__ float rl_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// Initial min-dist of a point: 1e10 when it competes, -1 when the reference skips it (|p|^2 <= 1e-3
// in double against the f32 magnitude, sampling_gpu.cu:105-106) or when it is padding (k < 0).
__device__ __forceinline__ float fps_tmin0(float x, float y, float z, int k) {
  return (k < 0 || (double)sq3(x, y, z) <= 1e-3) ? -1.f : 1e10f;
}

// The running min-dists of the first `lds_pts` points (curve order) live in dynamic LDS: the launch
// already asked for >= 128 KB it never touched (a CU reservation by resource, see
// fps_bucket_launch); with the min-dists there a touched bucket is three coordinate lines + the
// index line from L2, one conflict-free LDS read and, where a min-dist changes, one LDS write --
// no min-dist load, no write-back through L2 (the counters had the kernel at 2.4 x its algorithmic
// bytes: min-dist lines written back again and again).  A 40 000-point scene fits whole (160 000
// B of the CU's 163 840); points beyond lds_pts keep theirs in the global array `tmin`.
extern __shared__ float fps_ltmin[];

// One workgroup of NW waves per scene.  Every lane keeps the state of SL buckets: bucket b
// lives in wave b % NW, lane (b / NW) % 64, slot b / (64 NW) (neighbouring buckets go to
// different waves: the few buckets a late sample touches update in parallel -- each update is
// an L2 round trip plus a wave reduction, so 16 waves beat 4 here: measured 2.85 vs 5.4 ms on
// 8 x 40000 -> 2048).
// TM: where the running min-dists live -- 0: all in the global array, 1: all in LDS, 2: the first
// lds_pts points in LDS, the rest global.  A template parameter, not a run-time branch, for the
// two pure forms: with both paths in one instruction stream the compiler's s_waitcnt at the join
// must cover the path with FEWER loads per trip, i.e. it waits for part of the prefetched
// bucket -- measured: the mixed form 8 % slower than either pure form on a scene that fits.
template <int NW, int SL, int UB, bool PROF = false, int TM = 2>
__global__ __launch_bounds__(NW * 64) void fps_bucket_kernel(int n, int np, int m, int bs,
                                                             int log2bs,
                                                             const float *__restrict__ dataset,
                                                             const float4 *__restrict__ spts,
                                                             float *__restrict__ tmin,
                                                             int *__restrict__ idxs,
                                                             unsigned long long *dbg = nullptr,
                                                             Box8 *__restrict__ boxes = nullptr,
                                                             unsigned box_epoch = 0u,
                                                             int lds_pts = 0) {
  // PROF: s_memtime phase counters (tuning builds only; BTR_FPS_PROF=1 in tools/)
  unsigned long long tph[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = 0, nact = 0, nuse = 0, nchg = 0;
#define BTR_PH(i)                                                  \
  if (PROF) {                                                      \
    const unsigned long long now = __builtin_amdgcn_s_memtime();   \
    tph[i] += now - tlast;                                         \
    tlast = now;                                                   \
  }
  __shared__ BSlot slots[2][NW];

  // latency chain: issue ahead of co-resident streaming waves (without it 4.73 -> 4.77 ms per
  // step in round 4: no reason to make it optional)
  __builtin_amdgcn_s_setprio(3);
  const int bi = blockIdx.x;
  dataset += (size_t)bi * n * 3;
  spts += (size_t)bi * np;
  tmin += (size_t)bi * np;
  idxs += (size_t)bi * m;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nb = np >> 6;
  const TieParams tp{bs, log2bs, (n + bs - 1) >> log2bs};

  const float x0 = dataset[0], y0 = dataset[1], z0 = dataset[2];
  const int lds_bkts = __builtin_amdgcn_readfirstlane(lds_pts >> 6);  // buckets with LDS min-dists

  // ---- initial min-dists, derived from the points (coalesced pass over the bucket-SoA array;
  // it also pulls the scene into L2 for the per-lane box pass below)
  for (int p = tid; p < np; p += NW * 64) {
    const float *bp = (const float *)spts + (size_t)(p >> 6) * 256 + (p & 63);
    const float t = fps_tmin0(bp[0], bp[64], bp[128], __float_as_int(bp[192]));
    if (TM == 1 || (TM == 2 && (p >> 6) < lds_bkts)) fps_ltmin[p] = t;
    else tmin[p] = t;
  }
  __syncthreads();   // (also orders the global tail's stores before the owner waves' loads)

  // ---- per-slot bucket state (registers; every loop over s is fully unrolled)
  float bx0[SL], by0[SL], bz0[SL], bx1[SL], by1[SL], bz1[SL];  // bounding box
  unsigned mhi[SL], mlo[SL];                                   // best key of the bucket
  int mk[SL];                                                  // ... its original index
  float mx[SL], my[SL], mz[SL];                                // ... and coordinates
#pragma unroll
  for (int s = 0; s < SL; ++s) {
    const int myb = (s * 64 + lane) * NW + wave;
    bx0[s] = by0[s] = bz0[s] = bx1[s] = by1[s] = bz1[s] = 0.f;
    mhi[s] = mlo[s] = 0u;
    mk[s] = 0;
    mx[s] = my[s] = mz[s] = 0.f;
    if (myb < nb) {
      const float *bp = (const float *)spts + (size_t)myb * 256;
      float ax0 = bp[0], ax1 = ax0, ay0 = bp[64], ay1 = ay0, az0 = bp[128], az1 = az0;
      bool any = fps_tmin0(ax0, ay0, az0, __float_as_int(bp[192])) >= 0.f;
#pragma unroll 8
      for (int i = 1; i < 64; ++i) {
        const int qk = __float_as_int(bp[192 + i]);
        if (qk < 0) continue;  // padding slot: not in the box
        const float qx = bp[i], qy = bp[64 + i], qz = bp[128 + i];
        ax0 = fminf(ax0, qx); ax1 = fmaxf(ax1, qx);
        ay0 = fminf(ay0, qy); ay1 = fmaxf(ay1, qy);
        az0 = fminf(az0, qz); az1 = fmaxf(az1, qz);
        any |= fps_tmin0(qx, qy, qz, qk) >= 0.f;
      }
      bx0[s] = ax0; bx1[s] = ax1; by0[s] = ay0; by1[s] = ay1; bz0[s] = az0; bz1[s] = az1;
      mhi[s] = any ? __float_as_uint(1e10f) + 1u : 0u;  // competing points start at 1e10
      // for the ball query over the same buckets (internal.hpp Box8): every box carries the
      // launch's epoch and its own position, which the query checks before it trusts it
      if (boxes)
        boxes[(size_t)bi * nb + myb] =
            Box8{ax0, ay0, az0, __uint_as_float(box_epoch),
                 ax1, ay1, az1, __uint_as_float(box_stamp_pos(bi * nb + myb))};
    }
  }

  if (tid == 0) idxs[0] = 0;
  float sx = x0, sy = y0, sz = z0;

  if (PROF) tlast = __builtin_amdgcn_s_memtime();
  int wl = 0;         // this wave's best lane / key: only changes when one of the wave's
  unsigned wh = 0u;   // buckets is touched, so idle waves reuse it
  bool fresh = false;
  for (int j = 1; j < m; ++j) {
    bool touched = false;
    // SL == 1 fast path: the wave best is tracked in SGPRs while the buckets update
    unsigned bh = 0u;
    int bl = 0;
    bool ambig = true;
#pragma unroll
    for (int s = 0; s < SL; ++s) {
      // lower bound of the distance from the sample to the bucket, rounded exactly like the
      // point distance (see the header comment): skip the bucket when it cannot change
      const float cx = fminf(fmaxf(sx, bx0[s]), bx1[s]);
      const float cy = fminf(fmaxf(sy, by0[s]), by1[s]);
      const float cz = fminf(fmaxf(sz, bz0[s]), bz1[s]);
      const float ex = cx - sx, ey = cy - sy, ez = cz - sz;
      const float dbox = sq3(ex, ey, ez);
      const bool active = (__float_as_uint(dbox) + 1u) < mhi[s];  // mhi == 0: none competes
      unsigned long long todo = __ballot(active);
      if (PROF) {
        nact += __builtin_popcountll(todo);
        if (bi == 0 && lane == 0 && j < 2048)  // per-step touched-bucket count of this wave
          ((unsigned char *)(dbg + 4096))[j * NW + wave] =
              (unsigned char)min(255, __builtin_popcountll(todo));
      }
      BTR_PH(0)
      if (todo == 0) continue;
      touched = true;
      // Software-pipelined trips: the NEXT touched bucket's loads are in flight while the
      // current one is reduced, and the arg-max over the lanes whose buckets are NOT touched
      // runs under the first load's L2 latency.
      // (two register sets A/B, loop unrolled by two: loop-carried copies of the prefetched
      // registers would force the wait for the loads to the top of the loop)
      // (the bucket index is wave-uniform: LDS or global min-dists is a scalar branch)
      auto fetch = [&](int b, size_t &o, float4 &p, float &t) {
        const int bkt = (s * 64 + b) * NW + wave;
        const float *bp = (const float *)spts + (size_t)bkt * 256 + lane;
        o = (size_t)bkt * 64 + lane;
        p.x = bp[0];
        p.y = bp[64];
        p.z = bp[128];
        p.w = bp[192];
        if (TM == 1 || (TM == 2 && bkt < lds_bkts)) t = fps_ltmin[bkt * 64 + lane];
        else t = tmin[o];
      };
      auto process = [&](int cb, size_t co, const float4 &p, float t0) {
        if (PROF) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          BTR_PH(5)
        }
        // (plain v_sub_f32 through asm: the SLP vectoriser would pair (y, z) for a packed
        // subtract and shuffle the just-prefetched registers, i.e. wait for the load at once)
        float dx, dy, dz;
        asm("v_sub_f32 %0, %1, %2" : "=v"(dx) : "v"(p.x), "v"(sx));
        asm("v_sub_f32 %0, %1, %2" : "=v"(dy) : "v"(p.y), "v"(sy));
        asm("v_sub_f32 %0, %1, %2" : "=v"(dz) : "v"(p.z), "v"(sz));
        const float d = sq3(dx, dy, dz);
        const bool valid = t0 >= 0.f;
        const float t = valid ? fminf(d, t0) : t0;
        if (t != t0) {
          if (TM == 1 || (TM == 2 && (s * 64 + cb) * NW + wave < lds_bkts)) fps_ltmin[co] = t;
          else tmin[co] = t;
        }
        if (PROF) {
          const unsigned long long ch = __ballot(t != t0);
          nuse += ch ? 1 : 0;
          nchg += __builtin_popcountll(ch);
        }
        const unsigned hi = valid ? __float_as_uint(t) + 1u : 0u;
        const unsigned mh = wave_max_u32(hi);
        const unsigned long long cand = __ballot(hi == mh);
        const int kk = __float_as_int(p.w);
        int w;
        if (__builtin_popcountll(cand) == 1) {
          w = __builtin_ctzll(cand);
        } else {  // exact tie (duplicated points) or an all-skipped bucket
          const unsigned lo = (hi == mh) ? 0xffffffffu - fps_tk2(kk, tp.bs, tp.log2bs, tp.cpb)
                                         : 0u;
          const unsigned ml = wave_max_u32(lo);
          w = __builtin_ctzll(__ballot(hi == mh && lo == ml));
        }
        const int wk = __builtin_amdgcn_readlane(kk, w);  // wave-uniform: scalar tie key
        const unsigned wlo = 0xffffffffu - fps_tk2(wk, tp.bs, tp.log2bs, tp.cpb);
        const float wx = rl_f(p.x, w), wy = rl_f(p.y, w), wz = rl_f(p.z, w);
        const bool mine = lane == cb;  // the lane slot that keeps this bucket's state
        mhi[s] = mine ? mh : mhi[s];
        mlo[s] = mine ? wlo : mlo[s];
        mk[s] = mine ? wk : mk[s];
        mx[s] = mine ? wx : mx[s];
        my[s] = mine ? wy : my[s];
        mz[s] = mine ? wz : mz[s];
        if (SL == 1) {
          if (mh > bh) {
            bh = mh;
            bl = cb;
            ambig = false;
          } else if (mh == bh) {
            ambig = true;
          }
        }
      };
      int bA = __builtin_ctzll(todo), bB = 0;
      todo &= todo - 1;
      size_t oA, oB = 0;
      float4 pA, pB = make_float4(0.f, 0.f, 0.f, 0.f);
      float tA, tB = 0.f;
      fetch(bA, oA, pA, tA);
      if (SL == 1) {
        const unsigned rv = active ? 0u : mhi[0];
        bh = wave_max_u32(rv);
        const unsigned long long rc = __ballot(rv == bh);
        bl = __builtin_ctzll(rc);
        ambig = __builtin_popcountll(rc) > 1;
      }
      // (process() is called separately on the prefetching and the non-prefetching path: at
      // a join the compiler's s_waitcnt would have to cover both, i.e. wait for the prefetch)
      for (;;) {
        if (todo == 0) {
          process(bA, oA, pA, tA);
          break;
        }
        bB = __builtin_ctzll(todo);
        todo &= todo - 1;
        fetch(bB, oB, pB, tB);
        process(bA, oA, pA, tA);
        if (todo == 0) {
          process(bB, oB, pB, tB);
          break;
        }
        bA = __builtin_ctzll(todo);
        todo &= todo - 1;
        fetch(bA, oA, pA, tA);
        process(bB, oB, pB, tB);
      }
    }

    BTR_PH(1)
    // lane best over its slots, wave best over its lanes, block best over the waves
    unsigned lh = mhi[0], ll = mlo[0];
    int lk = mk[0];
    float lx = mx[0], ly = my[0], lz = mz[0];
#pragma unroll
    for (int s = 1; s < SL; ++s) {
      const bool better = mhi[s] > lh || (mhi[s] == lh && mlo[s] > ll);
      lh = better ? mhi[s] : lh;
      ll = better ? mlo[s] : ll;
      lk = better ? mk[s] : lk;
      lx = better ? mx[s] : lx;
      ly = better ? my[s] : ly;
      lz = better ? mz[s] : lz;
    }
    if (SL == 1 && touched && !ambig) {
      wl = bl;
      wh = bh;
    } else if (touched || !fresh) {
      wl = wave_argmax(lh, [&]() { return ll; }, wh);
    }
    fresh = true;
    // One barrier per step: the per-wave slots are double-buffered (a wave can only reach
    // the write of step j+2 after every wave passed the barrier of step j+1, i.e. after all
    // reads of step j), and EVERY wave reduces the NW slots itself with row DPP ops and
    // fetches the winner with one broadcast LDS read -- no second barrier, no readlanes
    // (3.07 -> 2.80 ms on 8 x 40000 -> 2048 against a leader-wave reduction + 2nd barrier).
    BSlot *sl = slots[j & 1];
    if (lane == wl) sl[wave] = BSlot{wh, ll, lx, ly, lz, lk, 0, 0};
    BTR_PH(2)
    lds_barrier();
    BTR_PH(3)
    const int r = lane & 15;
    const unsigned h = r < NW ? sl[r].hi : 0u;
    const unsigned gh = row16_max_u32(h);
    const unsigned long long cand = __ballot(h == gh) & 0xFFFFull;
    int ws;
    if (__builtin_popcountll(cand) == 1) {
      ws = __builtin_ctzll(cand);
    } else {
      const unsigned l = (h == gh && r < NW) ? sl[r].lo : 0u;
      const unsigned gl = row16_max_u32(l);
      ws = __builtin_ctzll(__ballot(h == gh && l == gl) & 0xFFFFull);
    }
    if (__builtin_amdgcn_readfirstlane(gh) == 0u) {  // nothing competes: best=-1, besti=0
      sx = x0; sy = y0; sz = z0;
      if (tid == 0) idxs[j] = 0;
    } else {
      const BSlot win = sl[ws];
      sx = win.x; sy = win.y; sz = win.z;
      if (tid == 0) idxs[j] = win.k;
    }
    BTR_PH(4)
  }
  if (PROF && lane == 0 && dbg) {
    unsigned long long *o = dbg + ((size_t)bi * NW + wave) * 8;
    for (int i = 0; i < 6; ++i) o[i] = tph[i];
    o[6] = nact;
    o[7] = (nuse << 32) | (nchg & 0xffffffffull);
  }
#undef BTR_PH
}

// Two exact alternatives to the owner-wave kernel were built, kept bit-exact in the whole index
// suite through rounds 2 - 5 and measured slower or equal on 8 x 40 000 -> 2 048; they were removed
// in round 6 (the numbers stay in DESIGN.md 7 and profiles/):
//   * a work-queue distribution of the touched buckets (no second trip on the busiest wave, but a
//     second barrier and the queue exchange): 2.71 vs 2.22 ms;
//   * multi-sample rounds (up to 4 samples accepted per round when the runners-up provably
//     survive the update): 2.26 vs 2.23 ms -- the bucket trips per SAMPLE are the same.

struct FpsPlan {
  int nb, np;
  size_t pts_bytes, k_bytes, sort_bytes;  // sort_bytes: cells[b][32768] + meta[b][8]
};

static FpsPlan fps_plan(int b, int n) {
  FpsPlan p;
  p.nb = cdiv(n, 64);
  p.np = p.nb * 64;
  p.pts_bytes = sizeof(float4) * (size_t)b * p.np;
  p.k_bytes = sizeof(int) * (size_t)b * p.np;
  p.sort_bytes = sizeof(int) * (size_t)b * (kCells + 8);
  return p;
}

constexpr int kBucketWaves = 16;
// Scenes whose min-dists do not all fit keep ALL of them global (tools/fps_lds_ab.py, 4 x 80 000
// -> 2 048 alone: 2.84 ms all global, 3.06 ms with the first 40 704 in LDS -- see TM)
constexpr bool kFpsLdsMixedDefault = false;
static bool fps_lds_mixed();
constexpr int kBucketMaxSlots = 2;
constexpr int kBucketMaxN = kBucketMaxSlots * kBucketWaves * 64 * 64;  // 131072 points/scene

bool fps_bucket_supported(int n) { return n > 0 && n <= kBucketMaxN; }

size_t fps_bucket_workspace_bytes(int b, int n) {
  if (b <= 0 || !fps_bucket_supported(n)) return 0;
  const FpsPlan p = fps_plan(b, n);
  return p.pts_bytes + p.k_bytes + p.sort_bytes;
}

int fps_bucket_launch(int b, int n, int m, const float *dataset, int *idxs, int bs, int log2bs,
                      void *workspace, size_t workspace_bytes, hipStream_t s) {
  const FpsPlan p = fps_plan(b, n);
  BTR_REQUIRE(workspace && workspace_bytes >= p.pts_bytes + p.k_bytes + p.sort_bytes,
              "furthest_point_sampling: workspace of %zu bytes required, got %zu",
              p.pts_bytes + p.k_bytes + p.sort_bytes, workspace_bytes);
  float4 *spts = (float4 *)workspace;
  float *sk = (float *)((char *)workspace + p.pts_bytes);  // the min-dist array
  fps_boxes_note(workspace, b, n, nullptr, 0u);   // (set again below by the kernel that writes them)
  {
    int *cells = (int *)((char *)workspace + p.pts_bytes + p.k_bytes);
    unsigned *meta = (unsigned *)(cells + (size_t)b * kCells);
    hipError_t e = hipMemsetAsync(cells, 0, p.sort_bytes, s);
    if (e != hipSuccess) return fail((int)e, "fps sort memset: %s", hipGetErrorString(e));
    const int gx = std::max(1, std::min(cdiv(n, 1024), 64));
    hipLaunchKernelGGL(fps_sortm_bbox_kernel, dim3(gx, b), dim3(256), 0, s, n, dataset, meta);
    hipLaunchKernelGGL(fps_sortm_hist_kernel, dim3(gx, b), dim3(256), 0, s, n, dataset, meta,
                       cells);
    hipLaunchKernelGGL(fps_sortm_scan_kernel, dim3(b), dim3(kSortThreads), 0, s, cells);
    hipLaunchKernelGGL(fps_sortm_scatter_kernel, dim3(gx, b), dim3(256), 0, s, n, p.np, dataset,
                       meta, cells, spts);
  }
  int rc = check_launch("furthest_point_sampling(sort)");
  if (rc) return rc;
  // Dynamic LDS of the launch: the running min-dists (4 B per point, as many as fit) and, beyond
  // them, a reservation.  A workgroup that holds most of its CU's 160 KB keeps every LDS-using
  // workgroup of the other streams OFF that CU: the kernel sits on one CU per scene for 2 ms
  // while the previous batch's training step runs on the other streams; whatever shares those
  // CUs runs at a fraction of its speed (16 high-priority waves beside it), and a launch of
  // equal row chunks ends with its slowest workgroup.  Eight sleeping 1024-thread workgroups
  // alone cost the backbone forward 8 %, VALU-busy ones more than double it
  // (tools/probe/occupant.hip, tools/fps_interference.py); round 4, same box, 20 steps: 4.54 ->
  // 4.37 ms per step with 128 KB held, 64 KB: 4.49.  (A CU-masked queue does not do it: the
  // dispatcher balances workgroups per shader engine, so taking one CU of 32 away slowed every
  // launch on that queue by 16 %; removed in round 6.)
  const int lds_kb = fps_lds_kb(p.np);
  size_t dyn = (size_t)lds_kb << 10;
  if (dyn > 0) {
    // (the attribute belongs to the function ON A DEVICE: set once per device and size.  A
    // runtime that refuses it leaves the min-dists in global memory: same results)
    static int attr_kb[64] = {};   // largest size granted so far, -1: refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
      dyn = 0;
    } else {
      if (attr_kb[dev] >= 0 && attr_kb[dev] < lds_kb) {
        bool ok = true;
        for (const void *f :
             {reinterpret_cast<const void *>(&fps_bucket_kernel<kBucketWaves, 1, 1, false, 0>),
              reinterpret_cast<const void *>(&fps_bucket_kernel<kBucketWaves, 1, 1, false, 1>),
              reinterpret_cast<const void *>(&fps_bucket_kernel<kBucketWaves, 1, 1, false, 2>),
              reinterpret_cast<const void *>(&fps_bucket_kernel<kBucketWaves, 2, 1, false, 0>),
              reinterpret_cast<const void *>(&fps_bucket_kernel<kBucketWaves, 2, 1, false, 2>),
              reinterpret_cast<const void *>(&fps_bucket_kernel<kBucketWaves, 1, 1, true, 0>),
              reinterpret_cast<const void *>(&fps_bucket_kernel<kBucketWaves, 1, 1, true, 1>)})
          ok = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) ==
                   hipSuccess && ok;
        (void)hipGetLastError();
        attr_kb[dev] = ok ? lds_kb : -1;
      }
      if (attr_kb[dev] < lds_kb) dyn = 0;
    }
  }
  int lds_pts = (int)std::min<size_t>((size_t)p.np, (dyn / sizeof(float)) & ~(size_t)63);
  if (!fps_lds_mixed() && lds_pts < p.np) lds_pts = 0;   // (the reservation stays: dyn)
  const int tm = lds_pts >= p.np ? 1 : (lds_pts == 0 ? 0 : 2);
  if (getenv("BTR_FPS_PROF")) {
    // tuning only (tools/fps_prof.py): s_memtime phase counters of the kernel, scene 0
    static unsigned long long *dbg = nullptr;
    if (!dbg) (void)hipMalloc(&dbg, sizeof(unsigned long long) * 64 * 16 * 8);
    if (tm == 1)
      hipLaunchKernelGGL((fps_bucket_kernel<16, 1, 1, true, 1>), dim3(b), dim3(1024), dyn, s, n,
                         p.np, m, bs, log2bs, dataset, spts, sk, idxs, dbg, (Box8 *)nullptr, 0u,
                         lds_pts);
    else
      hipLaunchKernelGGL((fps_bucket_kernel<16, 1, 1, true, 0>), dim3(b), dim3(1024), dyn, s, n,
                         p.np, m, bs, log2bs, dataset, spts, sk, idxs, dbg, (Box8 *)nullptr, 0u, 0);
    (void)hipStreamSynchronize(s);
    unsigned long long h[16 * 8];
    (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[5] = {"bbox-test", "bucket-update", "wave-argmax+slot", "barrier",
                            "block-argmax"};
    {
      static unsigned char cnt[2048 * 16];
      (void)hipMemcpy(cnt, dbg + 4096, sizeof(cnt), hipMemcpyDeviceToHost);
      double tot = 0, mx = 0;
      int hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const int steps = std::min(m, 2048);
      for (int j = 1; j < steps; ++j) {
        int t = 0, mm = 0;
        for (int w = 0; w < 16; ++w) {
          t += cnt[j * 16 + w];
          mm = std::max<int>(mm, cnt[j * 16 + w]);
        }
        tot += t;
        mx += mm;
        hist[std::min(mm, 7)]++;
      }
      fprintf(stderr, "[fps prof] scene 0 (%d of %d min-dists in LDS): touched buckets/step total "
              "%.2f, max over waves %.2f, balanced would be %.2f; hist(max) =", lds_pts, p.np,
              tot / (steps - 1), mx / (steps - 1), tot / (steps - 1) / 16.0);
      for (int i = 0; i < 8; ++i) fprintf(stderr, " %d", hist[i]);
      fprintf(stderr, "\n");
    }
    for (int w = 0; w < 16; w += 5) {
      fprintf(stderr, "[fps prof] scene 0 wave %2d:", w);
      for (int i = 0; i < 5; ++i)
        fprintf(stderr, " %s %.0f", names[i], (double)h[w * 8 + i] / (m - 1));
      fprintf(stderr, " load-wait %.0f cycles/step; touched buckets/step %.2f, of which %.2f "
              "changed a min-dist (%.1f points per useful trip)\n",
              (double)h[w * 8 + 5] / (m - 1), (double)h[w * 8 + 6] / (m - 1),
              (double)(h[w * 8 + 7] >> 32) / (m - 1),
              (double)(h[w * 8 + 7] & 0xffffffffull) / (double)std::max<unsigned long long>(1, h[w * 8 + 7] >> 32));
    }
    return check_launch("furthest_point_sampling(bucket,prof)");
  }
  // owner-wave kernel: 16 waves, one bucket per trip
  hipEvent_t *ev = fps_kernel_events();   // bench.py: event pair around THIS kernel only
  if (ev[0]) (void)hipEventRecord(ev[0], s);
  // the bucket boxes go where the counting sort kept its cells (dead by now): b * nb * 32 bytes
  // of the b * 131 072 there
  Box8 *boxes = reinterpret_cast<Box8 *>((char *)workspace + p.pts_bytes + p.k_bytes);
  static_assert(sizeof(Box8) * (kBucketMaxN / 64) <= sizeof(int) * kCells, "boxes fit the cells");
  // a process-wide launch counter (never 0: 0 means "unstamped boxes" to the query)
  static std::atomic<unsigned> epoch_counter{0x5a000000u};
  unsigned epoch = (epoch_counter.fetch_add(1u) + 1u) & 0x7fffffffu;
  if (epoch == 0u) epoch = (epoch_counter.fetch_add(1u) + 1u) & 0x7fffffffu;
#define BTR_FPS_BUCKET(SLOTS, TMODE)                                                          \
  hipLaunchKernelGGL((fps_bucket_kernel<kBucketWaves, SLOTS, 1, false, TMODE>), dim3(b),         \
                     dim3(kBucketWaves * 64), dyn, s, n, p.np, m, bs, log2bs, dataset, spts, sk, \
                     idxs, (unsigned long long *)nullptr, boxes, epoch, lds_pts)
  if (p.nb <= kBucketWaves * 64) {
    if (tm == 1) BTR_FPS_BUCKET(1, 1); else if (tm == 0) BTR_FPS_BUCKET(1, 0); else BTR_FPS_BUCKET(1, 2);
  } else {
    if (tm == 0) BTR_FPS_BUCKET(2, 0); else BTR_FPS_BUCKET(2, 2);
  }
#undef BTR_FPS_BUCKET
  if (ev[1]) (void)hipEventRecord(ev[1], s);
  ev[0] = ev[1] = nullptr;
  fps_boxes_note(workspace, b, n, boxes, epoch);
  return check_launch("furthest_point_sampling(bucket)");
}

// ---- dynamic LDS of the large-scene FPS launch (fps_bucket_launch): the running min-dists and
// the CU reservation.  fps_lds_reserve_kb(): the FLOOR the launch holds whatever the scene size --
// BTR_FPS_LDS_KB = k (0: none), default 128; in a data-parallel run (WORLD_SIZE > 1) 96: RCCL's
// kernels need up to ~64 KB of LDS per workgroup, and the all-reduce runs while the next batch's
// FPS holds its eight CUs -- with 96 KB taken a collective workgroup still fits beside a scene
// (160 - 97 KB), so a ring of more channels than the 248 free CUs' share never waits for the
// 2 ms sampling chain to end; the step's own LDS-heavy kernels (64 - 128 KB per workgroup pair)
// are still kept off those CUs.  fps_lds_kb(np): what a launch over np points asks for -- on a
// single GPU enough for all np min-dists when that fits beside the kernel's 1 KB of slots (159
// KB: 40 704 points), never less than the floor; with the environment variable or WORLD_SIZE > 1
// exactly the floor (min-dists beyond it stay in global memory).  btr_fps_set_lds_kb(k): a
// process-wide override for tests and A/B runs (k < 0: back to the rules above).
constexpr int kFpsLdsMaxKb = 159;
static std::atomic<int> g_fps_lds_override{-1};
// A scene whose min-dists do not all fit: the first lds_pts of them in LDS and the rest global
// (true), or all of them global as before round 6 (false; the launch still holds its LDS).  With
// the override set the split form runs whenever the size asks for it (tests, A/B).
static bool fps_lds_mixed() { return g_fps_lds_override.load(std::memory_order_relaxed) >= 0 || kFpsLdsMixedDefault; }
int fps_lds_reserve_kb() {
  static const int kb = [] {
    const char *e = getenv("BTR_FPS_LDS_KB");
    const char *w = getenv("WORLD_SIZE");
    const int v = e ? atoi(e) : (w && atoi(w) > 1 ? 96 : 128);
    return v > 0 && v <= kFpsLdsMaxKb ? v : 0;
  }();
  return kb;
}
int fps_lds_kb(int np) {
  const int o = g_fps_lds_override.load(std::memory_order_relaxed);
  if (o >= 0) return std::min(o, kFpsLdsMaxKb);
  static const bool fixed = [] {
    const char *w = getenv("WORLD_SIZE");
    return getenv("BTR_FPS_LDS_KB") != nullptr || (w && atoi(w) > 1);
  }();
  const int floor_kb = fps_lds_reserve_kb();
  if (fixed || floor_kb == 0) return floor_kb;
  const int need = (int)(((size_t)np * sizeof(float) + 1023) >> 10);
  return std::max(floor_kb, std::min(need, kFpsLdsMaxKb));
}

// ---- what the one-round grids count on (internal.hpp)
// CUs left to the collective's kernels when they overlap the step: BTR_COMM_CUS, default 16
// under WORLD_SIZE > 1 with BTR_DP=ddp, else 0.
static int comm_cus() {
  if (const char *e = getenv("BTR_COMM_CUS")) {
    const int v = atoi(e);
    return v >= 0 && v <= 128 ? v : 0;
  }
  const char *w = getenv("WORLD_SIZE");
  const char *dp = getenv("BTR_DP");
  const bool overlapped = dp && dp[0] == 'd' && dp[1] == 'd' && dp[2] == 'p' && dp[3] == 0;
  return (w && atoi(w) > 1 && overlapped) ? 16 : 0;
}
int grid_cus() {
  static const int avail = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0)
      cus = 256;   // (no device at hand: the build check, the CPU-side planning tests)
    (void)hipGetLastError();
    if (const char *e = getenv("BTR_GRID_CUS")) {   // sizing only
      const int v = atoi(e);
      if (v >= 8 && v <= cus) return v;
    }
    // default: the eight CUs the large-scene FPS of the NEXT batch sits on while a step runs (one
    // workgroup per scene) are not counted -- a one-round grid sized for all 256 leaves its last
    // workgroups waiting for a second round beside it (same box, 20 steps: 4.66 -> 4.59 ms
    // pipelined, 6.42 -> 6.47 strictly sequential).  Data-parallel runs whose gradient
    // collective overlaps the backward (BTR_DP=ddp: bucketed all-reduce kernels of RCCL beside
    // the weight-gradient GEMMs) leave comm_cus() more out: RCCL's ring kernels are persistent
    // workgroups (one per channel) that hold their CU's LDS staging buffers for the whole
    // collective.  The flat all-reduce (the default wrapper) runs behind the backward, beside
    // nothing of the step but the FPS: no CUs are set aside for it.
    return std::max(8, cus - 8 - comm_cus());
  }();
  return avail;
}

// ---- which workspaces hold bucket boxes (internal.hpp): the last few launches of this thread
namespace {
struct BoxNote {
  const void *ws = nullptr;
  int b = 0, n = 0;
  const Box8 *boxes = nullptr;
  unsigned epoch = 0u;
};
constexpr int kBoxNotes = 16;
inline BoxNote *box_notes() {
  static thread_local BoxNote notes[kBoxNotes];
  return notes;
}
}  // namespace
void fps_boxes_note(const void *workspace, int b, int n, const Box8 *boxes, unsigned epoch) {
  BoxNote *t = box_notes();
  static thread_local int next = 0;
  for (int i = 0; i < kBoxNotes; ++i)
    if (t[i].ws == workspace) {
      t[i] = BoxNote{workspace, b, n, boxes, epoch};
      return;
    }
  t[next] = BoxNote{workspace, b, n, boxes, epoch};
  next = (next + 1) % kBoxNotes;
}
const Box8 *fps_boxes_lookup(const void *workspace, int b, int n, unsigned *epoch) {
  static const bool off = getenv("BTR_BQ_FPS_BOXES") && getenv("BTR_BQ_FPS_BOXES")[0] == '0';
  *epoch = 0u;
  if (off) return nullptr;
  const BoxNote *t = box_notes();
  for (int i = 0; i < kBoxNotes; ++i)
    if (t[i].ws == workspace && t[i].b == b && t[i].n == n) {
      *epoch = t[i].epoch;
      return t[i].boxes;
    }
  return nullptr;
}

}  // namespace btr

extern "C" int btr_grid_cus(void) { return btr::grid_cus(); }
extern "C" int btr_fps_lds_reserve_kb(void) { return btr::fps_lds_reserve_kb(); }
extern "C" int btr_fps_lds_kb(int points) {
  return btr::fps_lds_kb((points + 63) / 64 * 64);
}
extern "C" void btr_fps_set_lds_kb(int kb) { btr::g_fps_lds_override.store(kb); }

extern "C" void btr_fps_time_next_kernel(void *start_event, void *stop_event) {
  btr::fps_kernel_events()[0] = (hipEvent_t)start_event;
  btr::fps_kernel_events()[1] = (hipEvent_t)stop_event;
}
