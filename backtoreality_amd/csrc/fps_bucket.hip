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
// the header).  BTR_FPS_CURVE=morton: Z order.
__device__ bool g_fps_morton = false;

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
  if (g_fps_morton) return (int)(spread3(qx) | (spread3(qy) << 1) | (spread3(qz) << 2));
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
// (read-only from here on; index -1 for padding) and tmin[np] = 1e10 (competing) or -1
// (skipped by the |p|^2 <= 1e-3 rule, or padding), np = 64 * ceil(n / 64), both in curve order.
// The min-dists live in their OWN array: the per-bucket write-back is then 2 full 128-B lines
// instead of 64 dwords strewn over 8 lines -- with 16 waves doing it the strided form costs
// ~1050 cycles per dependent bucket load, the contiguous one ~490 (tools/probe/lat_probe.hip).
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
    int *__restrict__ cells, float4 *__restrict__ spts, float *__restrict__ tmin) {
  const int bi = blockIdx.y;
  dataset += (size_t)bi * n * 3;
  float *sp = (float *)(spts + (size_t)bi * np);
  tmin += (size_t)bi * np;
  const SortBox bx = sort_box(meta + (size_t)bi * 8);
  for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
    const float x = dataset[k * 3], y = dataset[k * 3 + 1], z = dataset[k * 3 + 2];
    const int c = morton_cell(x, y, z, bx.mnx, bx.mny, bx.mnz, bx.scale);
    const int pos = atomicAdd(&cells[(size_t)bi * kCells + c], 1);
    const float mag = sq3(x, y, z);
    sp[soa_at(pos, 0)] = x;
    sp[soa_at(pos, 1)] = y;
    sp[soa_at(pos, 2)] = z;
    sp[soa_at(pos, 3)] = __int_as_float(k);
    tmin[pos] = ((double)mag <= 1e-3) ? -1.f : 1e10f;  // sampling_gpu.cu:105-106
  }
  // padding of the last bucket: index -1 (kept out of the bucket's bounding box), tmin < 0
  // (never competes); the coordinates only have to be finite
  if (blockIdx.x == 0 && n + (int)threadIdx.x < np) {
    const int pos = n + threadIdx.x;
    sp[soa_at(pos, 0)] = dataset[0];
    sp[soa_at(pos, 1)] = dataset[1];
    sp[soa_at(pos, 2)] = dataset[2];
    sp[soa_at(pos, 3)] = __int_as_float(-1);
    tmin[pos] = -1.f;
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

// One workgroup of NW waves per scene.  Every lane keeps the state of SL buckets: bucket b
// lives in wave b % NW, lane (b / NW) % 64, slot b / (64 NW) (neighbouring buckets go to
// different waves: the few buckets a late sample touches update in parallel -- each update is
// an L2 round trip plus a wave reduction, so 16 waves beat 4 here: measured 2.85 vs 5.4 ms on
// 8 x 40000 -> 2048).
template <int NW, int SL, int UB, bool PROF = false>
__global__ __launch_bounds__(NW * 64) void fps_bucket_kernel(int n, int np, int m, int bs,
                                                             int log2bs,
                                                             const float *__restrict__ dataset,
                                                             const float4 *__restrict__ spts,
                                                             float *__restrict__ tmin,
                                                             int *__restrict__ idxs,
                                                             unsigned long long *dbg = nullptr,
                                                             Box8 *__restrict__ boxes = nullptr,
                                                             unsigned box_epoch = 0u) {
  // PROF: s_memtime phase counters (tuning builds only; BTR_FPS_PROF=1 in tools/)
  unsigned long long tph[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = 0, nact = 0, nuse = 0, nchg = 0;
#define BTR_PH(i)                                                  \
  if (PROF) {                                                      \
    const unsigned long long now = __builtin_amdgcn_s_memtime();   \
    tph[i] += now - tlast;                                         \
    tlast = now;                                                   \
  }
  __shared__ BSlot slots[2][NW];

  // latency chain: issue ahead of co-resident streaming waves -- unless the caller hides this
  // kernel under other work anyway (box_epoch's top bit: see fps_bucket_launch, BTR_FPS_PRIO)
  if (!(box_epoch & 0x80000000u)) __builtin_amdgcn_s_setprio(3);
  box_epoch &= 0x7fffffffu;
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
      const float *tm = tmin + (size_t)myb * 64;
      float ax0 = bp[0], ax1 = ax0, ay0 = bp[64], ay1 = ay0, az0 = bp[128], az1 = az0;
      bool any = tm[0] >= 0.f;
#pragma unroll 8
      for (int i = 1; i < 64; ++i) {
        if (__float_as_int(bp[192 + i]) < 0) continue;  // padding slot: not in the box
        const float qx = bp[i], qy = bp[64 + i], qz = bp[128 + i];
        ax0 = fminf(ax0, qx); ax1 = fmaxf(ax1, qx);
        ay0 = fminf(ay0, qy); ay1 = fmaxf(ay1, qy);
        az0 = fminf(az0, qz); az1 = fmaxf(az1, qz);
        any |= tm[i] >= 0.f;
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
      auto fetch = [&](int b, size_t &o, float4 &p, float &t) {
        const size_t bkt = (size_t)((s * 64 + b) * NW + wave);
        const float *bp = (const float *)spts + bkt * 256 + lane;
        o = bkt * 64 + lane;
        p.x = bp[0];
        p.y = bp[64];
        p.z = bp[128];
        p.w = bp[192];
        t = tmin[o];
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
        if (t != t0) tmin[co] = t;
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

// ------------------------------------------- work-queue kernel (BTR_FPS_IMPL=queue, opt-in)
// MEASURED AND NOT ADOPTED (MI355X, 8 x 40000 -> 2048): 2.71 ms against 2.22 ms for the
// owner-wave kernel above; bit-exact in the whole index suite.  It removes the second trip of
// the busiest wave (items/step per wave 1.17 max instead of 1.6 trips) but pays for it with a
// second barrier, the queue exchange and a reduction over NW + Q candidates: s_memtime phase
// counters (BTR_FPS_PROF=1 BTR_FPS_IMPL=queue), cycles per step on the busiest wave:
// test+push 391, barrier A 669, pop+fetch+untouched-best 639, update 783, barrier B 235,
// reduce+winner 762 = 3503, against 2600 for the owner-wave kernel.  Kept for A/B.
// Same algorithm as fps_bucket_kernel above (bucket boxes, exact pruning, one sample per
// step), other distribution of the per-step work.  There, bucket b is ALWAYS updated by its
// owner wave b % NW: a late sample touches ~10 of 625 buckets, the busiest of the 16 waves makes
// 1.6-2.0 dependent trips (L2 round trip + wave reduction each) while half of the waves make
// none.  Here the owner lanes only run the box test; the touched buckets go through an LDS
// queue and wave w updates queue items w, w + NW, ...: one trip per wave up to NW touched
// buckets.  Per step:
//   test (owner lanes, registers) -> push touched bucket ids (one LDS atomic per wave)
//   -> barrier A -> pop -> load bucket (L2) | under that latency: best UNTOUCHED bucket of the
//   wave's own lanes -> update min-dists, bucket arg-max -> record to LDS -> barrier B
//   -> every wave reduces the NW untouched-bests + Q fresh records -> next sample.
// The per-bucket winner records live in LDS (rec[]), so a record is written once by whoever
// updated the bucket and read by the one wave-uniform broadcast read of the final winner;
// owners keep only the box and the bucket's max key in registers.
struct QRec {  // 32 B: two ds_write_b128 / ds_read_b128
  unsigned hi, lo;
  float x, y, z;
  int k, pad0, pad1;
};

template <int NW, int SL, bool PROF = false>
__global__ __launch_bounds__(NW * 64) void fps_queue_kernel(int n, int np, int m, int bs,
                                                            int log2bs,
                                                            const float *__restrict__ dataset,
                                                            const float4 *__restrict__ spts,
                                                            float *__restrict__ tmin,
                                                            int *__restrict__ idxs,
                                                            unsigned long long *dbg = nullptr) {
  // PROF: s_memtime phase counters (tuning builds only; BTR_FPS_PROF=1)
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0, nitem = 0;
#define BTR_QPH(i)                                                 \
  if (PROF) {                                                      \
    const unsigned long long now = __builtin_amdgcn_s_memtime();   \
    tph[i] += now - tlast;                                         \
    tlast = now;                                                   \
  }
  constexpr int NB = NW * 64 * SL;  // bucket capacity of the workgroup
  __shared__ QRec rec[NB];          // winner record of every bucket
  __shared__ int queue[2][NB];      // touched bucket ids of this step (double-buffered)
  __shared__ unsigned qhi[NB];      // max key of queue item i after its update
  __shared__ uint2 slots[NW];       // (max key, bucket) of each wave's best untouched bucket
  __shared__ int qcount[2];
  constexpr int kOut = 2048;        // samples are collected in LDS and written out in chunks:
  __shared__ int out_idx[kOut];     // no global store on the per-step chain

  __builtin_amdgcn_s_setprio(3);  // latency chain: issue ahead of co-resident streaming waves
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

  // ---- owner state: bounding box + max key of the buckets this lane owns
  float bx0[SL], by0[SL], bz0[SL], bx1[SL], by1[SL], bz1[SL];
  unsigned mhi[SL];
  int myb[SL];
#pragma unroll
  for (int s = 0; s < SL; ++s) {
    myb[s] = (s * 64 + lane) * NW + wave;
    bx0[s] = by0[s] = bz0[s] = bx1[s] = by1[s] = bz1[s] = 0.f;
    mhi[s] = 0u;
    if (myb[s] < nb) {
      const float *bp = (const float *)spts + (size_t)myb[s] * 256;
      const float *tm = tmin + (size_t)myb[s] * 64;
      float ax0 = bp[0], ax1 = ax0, ay0 = bp[64], ay1 = ay0, az0 = bp[128], az1 = az0;
      bool any = tm[0] >= 0.f;
#pragma unroll 8
      for (int i = 1; i < 64; ++i) {
        if (__float_as_int(bp[192 + i]) < 0) continue;  // padding slot: not in the box
        const float qx = bp[i], qy = bp[64 + i], qz = bp[128 + i];
        ax0 = fminf(ax0, qx); ax1 = fmaxf(ax1, qx);
        ay0 = fminf(ay0, qy); ay1 = fmaxf(ay1, qy);
        az0 = fminf(az0, qz); az1 = fmaxf(az1, qz);
        any |= tm[i] >= 0.f;
      }
      bx0[s] = ax0; bx1[s] = ax1; by0[s] = ay0; by1[s] = ay1; bz0[s] = az0; bz1[s] = az1;
      mhi[s] = any ? __float_as_uint(1e10f) + 1u : 0u;  // competing points start at 1e10
    }
  }
  if (tid == 0) {
    out_idx[0] = 0;
    qcount[0] = 0;
    qcount[1] = 0;
  }
  float sx = x0, sy = y0, sz = z0;
  const unsigned long long lt = (1ull << lane) - 1ull;
  lds_barrier();
  if (PROF) tlast = __builtin_amdgcn_s_memtime();

  for (int j = 1; j < m; ++j) {
    if ((j & (kOut - 1)) == 0) {  // flush a full chunk of samples
      lds_barrier();
      for (int i = tid; i < kOut; i += NW * 64) idxs[j - kOut + i] = out_idx[i];
      lds_barrier();
    }
    const int par = j & 1;
    int *q = queue[par];
    // ---- A: box test on the owner lanes (lower bound of the distance from the sample to the
    // bucket, rounded exactly like the point distance: see the header of this file)
    bool act[SL];
    int myslot[SL];
    unsigned lane_hi = 0u;
    int lane_b = 0;
#pragma unroll
    for (int s = 0; s < SL; ++s) {
      const float ex = __builtin_amdgcn_fmed3f(sx, bx0[s], bx1[s]) - sx;
      const float ey = __builtin_amdgcn_fmed3f(sy, by0[s], by1[s]) - sy;
      const float ez = __builtin_amdgcn_fmed3f(sz, bz0[s], bz1[s]) - sz;
      const float dbox = sq3(ex, ey, ez);
      act[s] = (__float_as_uint(dbox) + 1u) < mhi[s];  // mhi == 0: none competes
      // ---- B: push the touched buckets: one LDS atomic per wave and slot
      const unsigned long long mask = __ballot(act[s]);
      myslot[s] = 0;
      if (mask) {
        const int first = __builtin_ctzll(mask);
        int base = 0;
        if (lane == first) base = atomicAdd(&qcount[par], __builtin_popcountll(mask));
        base = __builtin_amdgcn_readlane(base, first);
        myslot[s] = base + __builtin_popcountll(mask & lt);
        if (act[s]) q[myslot[s]] = myb[s];
      }
      // this lane's best UNTOUCHED bucket (its record in rec[] stays valid through the step)
      const unsigned uh = act[s] ? 0u : mhi[s];
      if (s == 0 || uh > lane_hi) {
        lane_hi = uh;
        lane_b = myb[s];
      }
    }
    BTR_QPH(0)
    // Min-dist stores of the previous step must have COMPLETED before another wave may load
    // the same bucket (a bucket is updated by whichever wave pops it): they were issued a full
    // reduction phase ago, so this wait is normally free.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BTR_QPH(1)
    lds_barrier();  // ---- barrier A: the queue of this step is complete
    BTR_QPH(2)
    if (tid == 0) qcount[par ^ 1] = 0;  // next step's counter (everyone finished reading it
                                        // before arriving here)
    const int Q = __builtin_amdgcn_readfirstlane(qcount[par]);

    // ---- C: pop + update.  Item `it` of the queue belongs to wave it % NW.  Software
    // pipelined like the kernel above: the next item's loads are in flight while the current
    // one is reduced; the untouched-best of the wave runs under the first load's latency.
    auto fetch = [&](int it, int &b, float4 &p, float &t) {
      b = __builtin_amdgcn_readfirstlane(q[it]);
      const float *bp = (const float *)spts + (size_t)b * 256 + lane;
      p.x = bp[0];
      p.y = bp[64];
      p.z = bp[128];
      p.w = bp[192];
      t = tmin[(size_t)b * 64 + lane];
    };
    auto process = [&](int it, int b, const float4 &p, float t0) {
      float dx, dy, dz;  // plain v_sub_f32 through asm: see fps_bucket_kernel
      asm("v_sub_f32 %0, %1, %2" : "=v"(dx) : "v"(p.x), "v"(sx));
      asm("v_sub_f32 %0, %1, %2" : "=v"(dy) : "v"(p.y), "v"(sy));
      asm("v_sub_f32 %0, %1, %2" : "=v"(dz) : "v"(p.z), "v"(sz));
      const float d = sq3(dx, dy, dz);
      const bool valid = t0 >= 0.f;
      const float t = valid ? fminf(d, t0) : t0;
      if (t != t0) tmin[(size_t)b * 64 + lane] = t;
      const unsigned hi = valid ? __float_as_uint(t) + 1u : 0u;
      const unsigned mh = wave_max_u32(hi);
      const unsigned long long cand = __ballot(hi == mh);
      const int kk = __float_as_int(p.w);
      int w;
      if (__builtin_popcountll(cand) == 1) {
        w = __builtin_ctzll(cand);
      } else {  // exact tie (duplicated points) or an all-skipped bucket
        const unsigned lo = (hi == mh) ? 0xffffffffu - fps_tk2(kk, tp.bs, tp.log2bs, tp.cpb) : 0u;
        const unsigned ml = wave_max_u32(lo);
        w = __builtin_ctzll(__ballot(hi == mh && lo == ml));
      }
      if (lane == w) {  // the winner lane publishes its own point: no readlanes
        rec[b] = QRec{mh, 0xffffffffu - fps_tk2(kk, tp.bs, tp.log2bs, tp.cpb), p.x, p.y, p.z,
                      kk, 0, 0};
        qhi[it] = mh;
      }
    };
    auto untouched_best = [&]() {
      const unsigned wh = wave_max_u32(lane_hi);
      const unsigned long long c = __ballot(lane_hi == wh);
      int wl;
      if (__builtin_popcountll(c) == 1) {
        wl = __builtin_ctzll(c);
      } else {  // several buckets hold the same max key: the tie key decides
        const unsigned lo = (lane_hi == wh && wh != 0u) ? rec[lane_b].lo : 0u;
        const unsigned ml = wave_max_u32(lo);
        wl = __builtin_ctzll(__ballot(lane_hi == wh && lo == ml));
      }
      if (lane == wl) slots[wave] = make_uint2(wh, (unsigned)lane_b);
    };
    if (PROF) nitem += (Q > wave) ? (unsigned)((Q - wave + NW - 1) / NW) : 0u;
    {
      int it = wave;
      if (it >= Q) {
        untouched_best();
      } else {
        int bA, bB = 0, itA = it, itB = 0;
        float4 pA, pB = make_float4(0.f, 0.f, 0.f, 0.f);
        float tA, tB = 0.f;
        fetch(itA, bA, pA, tA);
        untouched_best();
        BTR_QPH(3)
        if (PROF) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          BTR_QPH(4)
        }
        for (;;) {
          itB = itA + NW;
          if (itB >= Q) {
            process(itA, bA, pA, tA);
            break;
          }
          fetch(itB, bB, pB, tB);
          process(itA, bA, pA, tA);
          itA = itB + NW;
          if (itA >= Q) {
            process(itB, bB, pB, tB);
            break;
          }
          fetch(itA, bA, pA, tA);
          process(itB, bB, pB, tB);
        }
      }
    }
    BTR_QPH(5)
    lds_barrier();  // ---- barrier B: every record of this step is in LDS
    BTR_QPH(6)

    // owners pick up the new max key of their touched buckets (used by the next box test;
    // the read runs under the reduction below)
    unsigned nh[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) nh[s] = act[s] ? qhi[myslot[s]] : mhi[s];

    // ---- D: every wave reduces the NW untouched-bests + the Q fresh records
    unsigned gh = 0u, glo = 0u;
    int gb = 0;
    const int C = NW + Q;
    for (int c0 = 0; c0 < C; c0 += 64) {
      const int ci = c0 + lane;
      unsigned h = 0u;
      int bkt = 0;
      if (ci < NW) {
        const uint2 v = slots[ci];
        h = v.x;
        bkt = (int)v.y;
      } else if (ci < C) {
        h = qhi[ci - NW];
        bkt = q[ci - NW];
      }
      const unsigned ch = wave_max_u32(h);
      const unsigned long long cand = __ballot(h == ch);
      if (c0 == 0 && C <= 64 && __builtin_popcountll(cand) == 1) {  // the common case
        gh = ch;
        gb = __builtin_amdgcn_readlane(bkt, __builtin_ctzll(cand));
        break;
      }
      if (ch == 0u) continue;
      const unsigned lo = (h == ch) ? rec[bkt].lo : 0u;  // ties: the tie key decides
      const unsigned cl = wave_max_u32(lo);
      const int cw = __builtin_ctzll(__ballot(h == ch && lo == cl));
      const int cb = __builtin_amdgcn_readlane(bkt, cw);
      if (ch > gh || (ch == gh && cl > glo)) {
        gh = ch;
        glo = cl;
        gb = cb;
      }
    }
#pragma unroll
    for (int s = 0; s < SL; ++s) mhi[s] = nh[s];
    if (gh == 0u) {  // nothing competes: best=-1, besti=0 in the reference
      sx = x0; sy = y0; sz = z0;
      if (tid == 0) out_idx[j & (kOut - 1)] = 0;
    } else {
      const QRec win = rec[gb];  // wave-uniform address: broadcast read
      sx = win.x; sy = win.y; sz = win.z;
      if (tid == 0) out_idx[j & (kOut - 1)] = win.k;
    }
    BTR_QPH(7)
  }
  lds_barrier();
  {
    const int done = (m - 1) & ~(kOut - 1);  // first sample of the chunk still in LDS
    for (int i = tid; done + i < m; i += NW * 64) idxs[done + i] = out_idx[i];
  }
  if (PROF && lane == 0 && dbg) {
    unsigned long long *o = dbg + ((size_t)bi * NW + wave) * 16;
    for (int i = 0; i < 8; ++i) o[i] = tph[i];
    o[8] = nitem;
  }
#undef BTR_QPH
}

// ------------------------------------- multi-sample rounds (BTR_FPS_IMPL=pm, opt-in)
// Exact FPS emitting SEVERAL samples per synchronisation round.
//
// Round-start invariant: every min-dist is exact for the samples chosen so far and each bucket
// b knows B1(b) = its best key (hi, lo, point) and B2hi(b) = the second-best `hi` inside b.
// Each wave publishes W1 = its best bucket (with that bucket's B2hi) and W2hi = the best `hi`
// among its OTHER buckets.  Sort the NW wave winners: c_1 > c_2 > ...; U = max W2hi bounds
// every point outside the winners' buckets.  c_1 is the next sample.  c_t (t >= 2) is the
// sample after c_1..c_{t-1} if, for every r < t,
//   (i)   hi(c_t) > U and hi(c_t) > 1          (nothing hidden can beat it; temp > 0)
//   (ii)  fminf(d(c_r, c_t), temp(c_t)) == temp(c_t)   (its own min-dist is untouched)
//   (iii) B2hi(bucket(c_r)) < hi(c_t)          (what remains in an accepted bucket is lower)
// because min-dists only decrease: every other point was <= its bucket best <= c_t already,
// points sharing a bucket with an accepted sample are bounded by (iii), the accepted samples
// themselves drop to 0.  The chain stops at the first failure.  All accepted samples are then
// applied in ONE pass over the touched buckets.  Comparisons on `hi` alone are conservative
// (a tie on hi just ends the chain), so the emitted sequence is exactly the sequential one.
struct MSlot {
  unsigned hi, lo;
  int k;
  float x, y, z;
  unsigned b2hi, w2hi;
};

// MEASURED AND NOT ADOPTED (MI355X, 8 x 40000 -> 2048): 2.26 ms against 2.23 ms for the
// one-sample-per-step kernel; bit-exact in the whole index suite.  The scheme above on the
// owner-wave kernel's machinery: touched buckets are updated with the software-pipelined
// two-register-set loop (all accepted samples applied in one pass: t = min over the samples);
// after the first barrier every wave RANKS its own candidate among the NW (one compare per
// lane), the KMAX best go to sorted[rank], and after a second barrier the chain is validated
// with one (earlier, later) pair per lane.  One bucket per lane (n <= NW * 64 * 64 points).
// KMAX = 4: 3.0 samples per round, 681 rounds -- but 7 500 cycles per round (s_memtime,
// BTR_FPS_PROF=1 BTR_FPS_IMPL=pm): box tests against 4 samples 520-840, bucket trips
// 1 800-2 200 (1.9 per wave and round), wave arg-max 380, barrier 1 1 200-2 900 (the wave with
// the most trips), rank 440-640, barrier 2 150-330, chain 1 200-2 100 = 2 500 cycles per SAMPLE
// against 2 600 for one sample per step.  The number of bucket trips per sample is the same
// either way (9.6; each an L2 round trip + three wave reductions), better balanced here (the
// busiest wave makes ~1.2 trips per sample instead of 1.6), and that gain is spent on the second
// barrier and on 16 waves sharing 4 SIMDs for the rank / chain arithmetic (an earlier form
// where every wave extracted and validated the candidates itself, no second barrier, needed
// 9 200 cycles per round: ~280 instructions x 16 waves is issue-bound).  What bounds FPS on
// this machine is the trip: ~1 000 cycles of L2 latency + reductions per touched bucket.
template <int NW, int KMAX, bool PROF = false>
__global__ __launch_bounds__(NW * 64) void fps_bucket_pm_kernel(
    int n, int np, int m, int bs, int log2bs, const float *__restrict__ dataset,
    const float4 *__restrict__ spts, float *__restrict__ tmin, int *__restrict__ idxs,
    unsigned long long *dbg) {
  static_assert(NW <= 16 && KMAX <= NW && KMAX <= 8, "slots are reduced by one 16-lane row");
  __shared__ MSlot slots[NW];      // one slot per wave: its winner this round
  __shared__ MSlot sorted[KMAX];   // the KMAX best of them, in order
  unsigned long long tph[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = 0, ntrip = 0;
#define BTR_PMH(i)                                                 \
  if (PROF) {                                                      \
    const unsigned long long now = __builtin_amdgcn_s_memtime();   \
    tph[i] += now - tlast;                                         \
    tlast = now;                                                   \
  }

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

  float bx0 = 0.f, by0 = 0.f, bz0 = 0.f, bx1 = 0.f, by1 = 0.f, bz1 = 0.f;
  unsigned mhi = 0u, mlo = 0u, mb2 = 0u;
  int mk = 0;
  float mx = 0.f, my = 0.f, mz = 0.f;
  {
    const int myb = lane * NW + wave;
    if (myb < nb) {
      const float *bp = (const float *)spts + (size_t)myb * 256;
      const float *tm = tmin + (size_t)myb * 64;
      float ax0 = bp[0], ax1 = ax0, ay0 = bp[64], ay1 = ay0, az0 = bp[128], az1 = az0;
      bool any = tm[0] >= 0.f;
#pragma unroll 8
      for (int i = 1; i < 64; ++i) {
        if (__float_as_int(bp[192 + i]) < 0) continue;
        const float qx = bp[i], qy = bp[64 + i], qz = bp[128 + i];
        ax0 = fminf(ax0, qx); ax1 = fmaxf(ax1, qx);
        ay0 = fminf(ay0, qy); ay1 = fmaxf(ay1, qy);
        az0 = fminf(az0, qz); az1 = fmaxf(az1, qz);
        any |= tm[i] >= 0.f;
      }
      bx0 = ax0; bx1 = ax1; by0 = ay0; by1 = ay1; bz0 = az0; bz1 = az1;
      mhi = mb2 = any ? __float_as_uint(1e10f) + 1u : 0u;   // (b2 = best: no chain until the
    }                                                         // bucket has been through a pass)
  }
  if (tid == 0) idxs[0] = 0;
  float ax[KMAX], ay[KMAX], az[KMAX];   // samples accepted in the previous round (wave-uniform)
#pragma unroll
  for (int a = 0; a < KMAX; ++a) { ax[a] = x0; ay[a] = y0; az[a] = z0; }
  int nacc = 1;
  unsigned long long rounds = 0;
  int wl = 0;
  unsigned wh = 0u, w2 = 0u;
  bool fresh = false;

  if (PROF) tlast = __builtin_amdgcn_s_memtime();
  for (int j = 1; j < m;) {
    // ---- box test against every accepted sample
    bool active = false;
#pragma unroll
    for (int a = 0; a < KMAX; ++a) {
      if (a < nacc) {
        const float cx = fminf(fmaxf(ax[a], bx0), bx1);
        const float cy = fminf(fmaxf(ay[a], by0), by1);
        const float cz = fminf(fmaxf(az[a], bz0), bz1);
        const float ex = cx - ax[a], ey = cy - ay[a], ez = cz - az[a];
        active |= (__float_as_uint(sq3(ex, ey, ez)) + 1u) < mhi;
      }
    }
    unsigned long long todo = __ballot(active);
    const bool touched = todo != 0;
    if (PROF) ntrip += __builtin_popcountll(todo);
    BTR_PMH(0)
    if (touched) {
      auto fetch = [&](int b, size_t &o, float4 &p, float &t) {
        const size_t bkt = (size_t)(b * NW + wave);
        const float *bp = (const float *)spts + bkt * 256 + lane;
        o = bkt * 64 + lane;
        p.x = bp[0];
        p.y = bp[64];
        p.z = bp[128];
        p.w = bp[192];
        t = tmin[o];
      };
      auto process = [&](int cb, size_t co, const float4 &p, float t0) {
        float dx, dy, dz;
        asm("v_sub_f32 %0, %1, %2" : "=v"(dx) : "v"(p.x), "v"(ax[0]));
        asm("v_sub_f32 %0, %1, %2" : "=v"(dy) : "v"(p.y), "v"(ay[0]));
        asm("v_sub_f32 %0, %1, %2" : "=v"(dz) : "v"(p.z), "v"(az[0]));
        float d = sq3(dx, dy, dz);
#pragma unroll
        for (int a = 1; a < KMAX; ++a) {
          if (a < nacc) {
            const float ex = p.x - ax[a], ey = p.y - ay[a], ez = p.z - az[a];
            d = fminf(d, sq3(ex, ey, ez));
          }
        }
        const bool valid = t0 >= 0.f;
        const float t = valid ? fminf(d, t0) : t0;
        if (t != t0) tmin[co] = t;
        const unsigned hi = valid ? __float_as_uint(t) + 1u : 0u;
        const unsigned mh = wave_max_u32(hi);
        const unsigned long long cand = __ballot(hi == mh);
        const int kk = __float_as_int(p.w);
        int w;
        if (__builtin_popcountll(cand) == 1) {
          w = __builtin_ctzll(cand);
        } else {
          const unsigned lo = (hi == mh) ? 0xffffffffu - fps_tk2(kk, tp.bs, tp.log2bs, tp.cpb)
                                         : 0u;
          const unsigned ml = wave_max_u32(lo);
          w = __builtin_ctzll(__ballot(hi == mh && lo == ml));
        }
        const unsigned b2 = wave_max_u32(lane == w ? 0u : hi);   // second-best hi of the bucket
        const int wk = __builtin_amdgcn_readlane(kk, w);
        const unsigned wlo = 0xffffffffu - fps_tk2(wk, tp.bs, tp.log2bs, tp.cpb);
        const float wx = rl_f(p.x, w), wy = rl_f(p.y, w), wz = rl_f(p.z, w);
        const bool mine = lane == cb;
        mhi = mine ? mh : mhi;
        mlo = mine ? wlo : mlo;
        mb2 = mine ? b2 : mb2;
        mk = mine ? wk : mk;
        mx = mine ? wx : mx;
        my = mine ? wy : my;
        mz = mine ? wz : mz;
      };
      int bA = __builtin_ctzll(todo), bB = 0;
      todo &= todo - 1;
      size_t oA, oB = 0;
      float4 pA, pB = make_float4(0.f, 0.f, 0.f, 0.f);
      float tA, tB = 0.f;
      fetch(bA, oA, pA, tA);
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
    BTR_PMH(1)
    // ---- wave winner and the bound for everything else in this wave
    if (touched || !fresh) {
      wl = wave_argmax(mhi, [&]() { return mlo; }, wh);
      w2 = wave_max_u32(lane == wl ? 0u : mhi);
    }
    fresh = true;
    // (single-buffered: a slot is rewritten after the NEXT round's second barrier at the
    // earliest, `sorted` after the next round's first)
    MSlot *sl = slots;
    if (lane == wl) sl[wave] = MSlot{wh, mlo, mk, mx, my, mz, mb2, w2};
    BTR_PMH(2)
    lds_barrier();
    BTR_PMH(3)

    // ---- phase A: every wave ranks ITS candidate among the NW (one compare per lane); the
    // KMAX best are copied to sorted[rank].  (Doing the whole extraction + validation in every
    // wave was measured first: 16 waves x ~280 scalar-ish instructions on 4 SIMDs is issue-
    // bound, 2 200-5 000 cycles per round; ranking is 10 instructions.)
    const int r = lane & 15;
    unsigned ehi = 0u, elo = 0u, ew2 = 0u;
    if (r < NW) {
      ehi = sl[r].hi;
      elo = sl[r].lo;
      ew2 = sl[r].w2hi;
    }
    const unsigned U = row16_max_u32(ew2);   // bounds every point hidden behind a wave winner
    const unsigned mylo = (unsigned)__builtin_amdgcn_readlane((int)mlo, wl);
    const bool gt = r != wave && r < NW &&
                    (ehi > wh || (ehi == wh && (elo > mylo || (elo == mylo && r < wave))));
    const int rank = __builtin_popcountll(__ballot(gt) & 0xFFFFull);
    if (rank < KMAX && lane == wl) sorted[rank] = MSlot{wh, mlo, mk, mx, my, mz, mb2, w2};
    BTR_PMH(4)
    lds_barrier();
    BTR_PMH(5)
    // ---- phase B: chain validation, one (earlier, later) candidate pair per lane
    const int limit = min(KMAX, m - j);
    // lane p < KMAX*(KMAX-1)/2: pair (q, t), q < t, enumerated by t: (0,1) (0,2) (1,2) (0,3) ...
    int pt = 1, pq = lane;
#pragma unroll
    for (int t = 1; t < KMAX; ++t)
      if (pq >= pt && pt == t) { pq -= t; pt = t + 1; }
    bool okp = true;
    if (pt < KMAX) {
      const MSlot cq = sorted[pq], ct = sorted[pt];
      const float tt = __uint_as_float(ct.hi - 1u);
      const float dx = ct.x - cq.x, dy = ct.y - cq.y, dz = ct.z - cq.z;   // point c_t, sample c_q
      okp = (fminf(sq3(dx, dy, dz), tt) == tt) && (cq.b2hi < ct.hi);       // (ii), (iii)
    }
    // lane 32 + t: condition (i) of candidate t (and a strict drop from candidate t - 1: a tie on
    // hi ends the chain)
    bool oki = true;
    if (lane >= 32 && lane < 32 + KMAX) {
      const int t = lane - 32;
      const unsigned ht = sorted[t].hi;
      const unsigned hp = sorted[t > 0 ? t - 1 : 0].hi;
      oki = t < limit && (t == 0 || (ht > U && ht > 1u && ht < hp));
    }
    const unsigned long long badp = __ballot(!okp), badi = __ballot(!oki) >> 32;
    int A = 1;
#pragma unroll
    for (int t = 1; t < KMAX; ++t) {
      const unsigned long long pairs = ((1ull << t) - 1ull) << (t * (t - 1) / 2);  // (q, t), q < t
      if (A == t && !((badi >> t) & 1ull) && !(badp & pairs)) A = t + 1;
    }
    const MSlot c0 = sorted[0];
    if (c0.hi == 0u) {   // nothing competes: best = -1, besti = 0 in the reference
      A = 1;
      ax[0] = x0; ay[0] = y0; az[0] = z0;
      if (tid == 0) idxs[j] = 0;
    } else {
#pragma unroll
      for (int a = 0; a < KMAX; ++a) {
        const MSlot c = sorted[a < A ? a : 0];
        ax[a] = c.x; ay[a] = c.y; az[a] = c.z;
      }
      if (tid < A) idxs[j + tid] = sorted[tid].k;
    }
    nacc = A;
    j += nacc;
    ++rounds;
    BTR_PMH(6)
  }
  if (PROF) {
    if (lane == 0 && dbg) {
      unsigned long long *o = dbg + 8 + ((size_t)bi * NW + wave) * 16;
      for (int i = 0; i < 7; ++i) o[i] = tph[i];
      o[7] = ntrip;
      o[8] = rounds;
    }
  } else if (dbg && tid == 0) {
    dbg[bi] = rounds;
  }
#undef BTR_PMH
}

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
    static int curve_set = -1;
    const char *cv = getenv("BTR_FPS_CURVE");
    const int want = (cv && cv[0] == 'm') ? 1 : 0;
    if (curve_set != want) {
      const bool flag = want == 1;
      (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_fps_morton), &flag, sizeof(bool), 0,
                                   hipMemcpyHostToDevice, s);
      curve_set = want;
    }
  }
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
                       meta, cells, spts, sk);
  }
  int rc = check_launch("furthest_point_sampling(sort)");
  if (rc) return rc;
  if (getenv("BTR_FPS_PROF") && getenv("BTR_FPS_IMPL") && getenv("BTR_FPS_IMPL")[0] == 'q') {
    static unsigned long long *dbg = nullptr;
    if (!dbg) (void)hipMalloc(&dbg, sizeof(unsigned long long) * 64 * 16 * 16);
    hipLaunchKernelGGL((fps_queue_kernel<16, 1, true>), dim3(b), dim3(1024), 0, s, n, p.np, m, bs,
                       log2bs, dataset, spts, sk, idxs, dbg);
    (void)hipStreamSynchronize(s);
    unsigned long long h[16 * 16];
    (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[8] = {"test+push", "store-wait", "barrierA", "pop+fetch+untouched",
                            "load-wait", "process", "barrierB", "reduce+winner"};
    for (int w = 0; w < 16; w += 5) {
      fprintf(stderr, "[fps queue prof] scene 0 wave %2d:", w);
      double tot = 0;
      for (int i = 0; i < 8; ++i) {
        fprintf(stderr, " %s %.0f", names[i], (double)h[w * 16 + i] / (m - 1));
        tot += (double)h[w * 16 + i] / (m - 1);
      }
      fprintf(stderr, " | total %.0f cycles/step, items/step %.2f\n", tot,
              (double)h[w * 16 + 8] / (m - 1));
    }
    return check_launch("furthest_point_sampling(queue,prof)");
  }
  if (getenv("BTR_FPS_PROF") && !(getenv("BTR_FPS_IMPL") && getenv("BTR_FPS_IMPL")[0] == 'p')) {
    // tuning only: s_memtime phase counters of the default kernel
    static unsigned long long *dbg = nullptr;
    if (!dbg) (void)hipMalloc(&dbg, sizeof(unsigned long long) * 64 * 16 * 8);
    hipLaunchKernelGGL((fps_bucket_kernel<16, 1, 1, true>), dim3(b), dim3(1024), 0, s, n, p.np, m,
                       bs, log2bs, dataset, spts, sk, idxs, dbg);
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
      fprintf(stderr, "[fps prof] scene 0: touched buckets/step total %.2f, max over waves %.2f,"
              " balanced would be %.2f; hist(max) =", tot / (steps - 1), mx / (steps - 1),
              tot / (steps - 1) / 16.0);
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
  {  // BTR_FPS_IMPL=queue: the work-queue kernel (measured slower, see its header)
    const char *e = getenv("BTR_FPS_IMPL");
    if (e && e[0] == 'q') {
      if (p.nb <= kBucketWaves * 64)
        hipLaunchKernelGGL((fps_queue_kernel<kBucketWaves, 1>), dim3(b), dim3(kBucketWaves * 64),
                           0, s, n, p.np, m, bs, log2bs, dataset, spts, sk, idxs);
      else
        hipLaunchKernelGGL((fps_queue_kernel<kBucketWaves, 2>), dim3(b), dim3(kBucketWaves * 64),
                           0, s, n, p.np, m, bs, log2bs, dataset, spts, sk, idxs);
      return check_launch("furthest_point_sampling(queue)");
    }
  }
  {  // BTR_FPS_IMPL=pm: multi-sample rounds on the pipelined machinery (n <= 65 536; measured
     // equal to the one-sample-per-step kernel below, see fps_bucket_pm_kernel)
    const char *e = getenv("BTR_FPS_IMPL");
    const bool pm = e && ((e[0] == 'p' && e[1] == 'm') || e[0] == 'm') &&   // "pm" / "multi"
                    p.nb <= kBucketWaves * 64;
    if (pm && getenv("BTR_FPS_PROF")) {   // tuning: s_memtime phase counters of scene 0
      static unsigned long long *dbgp = nullptr;
      if (!dbgp) (void)hipMalloc(&dbgp, sizeof(unsigned long long) * (8 + 64 * 16 * 16));
      hipLaunchKernelGGL((fps_bucket_pm_kernel<kBucketWaves, 4, true>), dim3(b),
                         dim3(kBucketWaves * 64), 0, s, n, p.np, m, bs, log2bs, dataset, spts, sk,
                         idxs, dbgp);
      (void)hipStreamSynchronize(s);
      unsigned long long h[8 + 16 * 16];
      (void)hipMemcpy(h, dbgp, sizeof(h), hipMemcpyDeviceToHost);
      const char *names[7] = {"box-test", "bucket-trips", "wave-argmax+slot", "barrier-1",
                              "rank", "barrier-2", "chain"};
      for (int w = 0; w < 16; w += 5) {
        const unsigned long long *o = h + 8 + w * 16;
        const double rnd = (double)std::max<unsigned long long>(1, o[8]);
        fprintf(stderr, "[fps pm prof] scene 0 wave %2d:", w);
        double tot = 0;
        for (int i = 0; i < 7; ++i) {
          fprintf(stderr, " %s %.0f", names[i], (double)o[i] / rnd);
          tot += (double)o[i] / rnd;
        }
        fprintf(stderr, " | %.0f cycles/round, %.0f rounds, %.2f trips/round\n", tot, rnd,
                (double)o[7] / rnd);
      }
      return check_launch("furthest_point_sampling(bucket,pm,prof)");
    }
    if (pm) {
      unsigned long long *rd = nullptr;
      if (getenv("BTR_FPS_ROUNDS")) {
        static unsigned long long *dbg3 = nullptr;
        if (!dbg3) (void)hipMalloc(&dbg3, sizeof(unsigned long long) * 4096);
        rd = dbg3;
      }
      hipEvent_t *ev = fps_kernel_events();
      if (ev[0]) (void)hipEventRecord(ev[0], s);
      hipLaunchKernelGGL((fps_bucket_pm_kernel<kBucketWaves, 4>), dim3(b),
                         dim3(kBucketWaves * 64), 0, s, n, p.np, m, bs, log2bs, dataset, spts, sk,
                         idxs, rd);
      if (ev[1]) (void)hipEventRecord(ev[1], s);
      ev[0] = ev[1] = nullptr;
      if (rd) {
        (void)hipStreamSynchronize(s);
        unsigned long long h = 0;
        (void)hipMemcpy(&h, rd, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[fps] scene 0: %llu rounds for %d samples (%.2f samples/round)\n", h,
                m - 1, (double)(m - 1) / (double)h);
      }
      return check_launch("furthest_point_sampling(bucket,pm)");
    }
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
  // BTR_FPS_PRIO=0: no raised wave priority (rides in the epoch's top bit, stripped in the kernel
  // before the boxes are stamped).  With priority 3 the eight FPS workgroups starve whatever
  // shares their CUs; when the pyramid is hidden under a training step anyway, the step's own
  // kernels are what should not wait.
  const char *pe = getenv("BTR_FPS_PRIO");
  const unsigned kflag = (pe && pe[0] == '0') ? 0x80000000u : 0u;
  // BTR_CU_MASK: the sampling kernel alone moves to the stream that owns the reserved CUs (its
  // sort launches and everything behind it stay where they are); fork / join with two events.
  // Not while a HIP graph is captured (the events are not part of the capture's streams).
  hipStream_t ks = s;
  static thread_local hipEvent_t fork_ev = nullptr, join_ev = nullptr;
  if (hipStream_t fs = cu_mask_fps_stream()) {
    hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cst) == hipSuccess && cst == hipStreamCaptureStatusNone) {
      if (!fork_ev) {
        (void)hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&join_ev, hipEventDisableTiming);
      }
      (void)hipEventRecord(fork_ev, s);
      (void)hipStreamWaitEvent(fs, fork_ev, 0);
      ks = fs;
    } else {
      (void)hipGetLastError();
    }
  }
  if (ev[0] && ks != s) (void)hipEventRecord(ev[0], ks);   // (time the kernel on ITS stream)
  // The launch asks for 128 KB (BTR_FPS_LDS_KB = k: k KB, 0: none) of dynamic LDS it never
  // touches, on top of its own 1 KB.  A workgroup that holds most of its CU's 160 KB keeps every
  // LDS-using workgroup of the other streams OFF that CU: a CU reservation by resource.  Why:
  // the kernel sits on one CU per scene for 2 ms while the previous batch's training step runs
  // on the other streams; whatever shares those CUs runs at a fraction of its speed (16
  // high-priority waves beside it), and a launch of equal row chunks ends with its slowest
  // workgroup.  Eight sleeping 1024-thread workgroups alone cost the backbone forward 8 %, VALU-
  // busy ones more than double it (tools/probe/occupant.hip, tools/fps_interference.py).  A
  // CU-masked queue does not do it: the dispatcher balances workgroups per shader engine, so
  // taking one CU of 32 away slows every launch on that queue by 16 % (BTR_CU_MASK).
  // Same box, 20 steps: 4.54 -> 4.37 ms per step; 64 KB: 4.49.
  static const int lds_kb = fps_lds_reserve_kb();
  size_t dyn = (size_t)lds_kb << 10;
  if (dyn > 0) {
    // (the attribute belongs to the function ON A DEVICE: set once per device.  A runtime that
    // refuses it leaves the launch as it was: the reservation is about speed only)
    static signed char attr_state[64] = {};   // 0 unknown, 1 set, -1 refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
      dyn = 0;
    } else {
      if (attr_state[dev] == 0) {
        const hipError_t e1 = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&fps_bucket_kernel<kBucketWaves, 1, 1>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        const hipError_t e2 = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&fps_bucket_kernel<kBucketWaves, 2, 1>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        (void)hipGetLastError();
        attr_state[dev] = (e1 == hipSuccess && e2 == hipSuccess) ? 1 : -1;
      }
      if (attr_state[dev] != 1) dyn = 0;
    }
  }
  if (p.nb <= kBucketWaves * 64)
    hipLaunchKernelGGL((fps_bucket_kernel<kBucketWaves, 1, 1>), dim3(b), dim3(kBucketWaves * 64),
                       dyn, ks, n, p.np, m, bs, log2bs, dataset, spts, sk, idxs,
                       (unsigned long long *)nullptr, boxes, epoch | kflag);
  else
    hipLaunchKernelGGL((fps_bucket_kernel<kBucketWaves, 2, 1>), dim3(b), dim3(kBucketWaves * 64),
                       dyn, ks, n, p.np, m, bs, log2bs, dataset, spts, sk, idxs,
                       (unsigned long long *)nullptr, boxes, epoch | kflag);
  if (ev[1]) (void)hipEventRecord(ev[1], ks);
  ev[0] = ev[1] = nullptr;
  if (ks != s) {
    (void)hipEventRecord(join_ev, ks);
    (void)hipStreamWaitEvent(s, join_ev, 0);
  }
  fps_boxes_note(workspace, b, n, boxes, epoch);
  return check_launch("furthest_point_sampling(bucket)");
}

// ---- the LDS reservation of the large-scene FPS launch (see btr_furthest_point_sampling's
// bucket path): BTR_FPS_LDS_KB = k (0: none), default 128.  In a data-parallel run (WORLD_SIZE
// > 1) the default drops to 96: RCCL's kernels need up to ~64 KB of LDS per workgroup, and the
// all-reduce runs while the next batch's FPS holds its eight CUs -- with 96 KB taken a collective
// workgroup still fits beside a scene (160 - 97 KB), so a ring of more channels than the 248 free
// CUs' share never waits for the 2 ms sampling chain to end; the step's own LDS-heavy kernels (64
// - 128 KB per workgroup pair) are still kept off those CUs.
int fps_lds_reserve_kb() {
  static const int kb = [] {
    const char *e = getenv("BTR_FPS_LDS_KB");
    const char *w = getenv("WORLD_SIZE");
    const int v = e ? atoi(e) : (w && atoi(w) > 1 ? 96 : 128);
    return v > 0 && v <= 156 ? v : 0;
  }();
  return kb;
}

// ---- CU partitioning (internal.hpp)
int cu_mask_reserved() {
  static const int c = [] {
    const char *e = getenv("BTR_CU_MASK");
    const int v = e ? atoi(e) : 0;
    return v > 0 && v <= 8 ? v : 0;
  }();
  return c;
}
// CUs left to the collective's kernels when they overlap the step (see cu_mask_avail_cus):
// BTR_COMM_CUS, default 16 under WORLD_SIZE > 1 with BTR_DP=ddp, else 0.
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
int cu_mask_avail_cus() {
  static const int avail = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0)
      cus = 256;   // (no device at hand: the build check, the CPU-side planning tests)
    (void)hipGetLastError();
    if (const char *e = getenv("BTR_GRID_CUS")) {   // sizing only, no partition
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
    return std::max(8, cus - 8 * std::max(1, cu_mask_reserved()) - comm_cus());
  }();
  return avail;
}
hipStream_t cu_mask_create_stream(bool reserved) {
  const int c = cu_mask_reserved();
  hipStream_t st = nullptr;
  if (c == 0) {
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return st;
  }
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return nullptr;
  const int words = (cus + 31) / 32;
  std::vector<uint32_t> mask(words, 0u);
  for (int i = 0; i < cus; ++i) {
    const bool low = i < 8 * c;   // bits 0 .. 8c-1: CUs 0 .. c-1 of each of the 8 XCDs
    if (low == reserved) mask[i >> 5] |= 1u << (i & 31);
  }
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask.data()) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return st;
}
hipStream_t cu_mask_fps_stream() {
  if (cu_mask_reserved() == 0) return nullptr;
  static thread_local hipStream_t per_dev[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!per_dev[dev]) per_dev[dev] = cu_mask_create_stream(true);
  return per_dev[dev];
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

extern "C" int btr_cu_mask_reserved(void) { return btr::cu_mask_reserved(); }
extern "C" int btr_grid_cus(void) { return btr::cu_mask_avail_cus(); }
extern "C" int btr_fps_lds_reserve_kb(void) { return btr::fps_lds_reserve_kb(); }
extern "C" void *btr_cu_mask_create_stream(int reserved) {
  return (void *)btr::cu_mask_create_stream(reserved != 0);
}

extern "C" void btr_fps_time_next_kernel(void *start_event, void *stop_event) {
  btr::fps_kernel_events()[0] = (hipEvent_t)start_event;
  btr::fps_kernel_events()[1] = (hipEvent_t)stop_event;
}
