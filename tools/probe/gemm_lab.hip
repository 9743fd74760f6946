// gemm_lab.hip -- stand-alone bench of candidate NT GEMM cores against the library kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/probe/gemm_lab.hip \
//         -Lbacktoreality_amd/lib -lbtr_pointnet2 -Wl,-rpath,'$ORIGIN/../../backtoreality_amd/lib' \
//         -o tools/probe/gemm_lab
// C[r][n] = sum_k f(A[r][k]) W[n][k], f = relu(pa*y+pb) (PRO) -- the forward GEMM of a shared-MLP
// layer (reference pytorch_utils.py:11-36 through conv2d 1x1 + BatchNorm + ReLU).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#include "../../include/btr_pointnet2.h"

#pragma clang fp contract(fast)

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

// ---------------------------------------------------------------------------------------------
// v2: barrier-free.  W (BN x K) lives in LDS for the whole kernel; every WAVE owns 32-row strips
// end to end: A rows go global -> registers directly in MFMA fragment order (lane (r, h) holds
// k = c*32 + h*16 + 4q .. +3 of row r), a register ring (the load of strip t+1's group g is
// issued right after the MFMAs that consumed strip t's group g), NJ = BN/32 accumulators.
// No __syncthreads in the loop: waves drift apart and cover each other's epilogues.
template <int BN, int KC /* K / 32 */, bool PRO, bool STATS, int D /* ring depth */, int WPS>
__global__ __launch_bounds__(256, WPS) void gemm_nt_v2(const float *__restrict__ A, int lda,
                                                       const float *__restrict__ W, int ldw,
                                                       float *__restrict__ C, int ldc, int R,
                                                       int N, const float *__restrict__ pa,
                                                       const float *__restrict__ pb,
                                                       float *__restrict__ part) {
  constexpr int K = KC * 32;
  constexpr int NJ = BN / 32;
  constexpr int LD = 36;
  constexpr int G = KC * 4;   // float4 groups per lane per strip
  static_assert(G % D == 0, "ring depth must divide the groups of a strip");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Ws = smem;                          // [KC][BN][LD]
  float *sPa = smem + KC * BN * LD;          // [K]
  float *sPb = sPa + K;
  double *red = reinterpret_cast<double *>(sPb + K);   // [2][4 waves][BN]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int n_blk = blockIdx.y * BN;
  for (int e = tid; e < BN * (K / 4); e += 256) {
    const int n = e / (K / 4), k4 = (e % (K / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n_blk + n < N) v = *reinterpret_cast<const float4 *>(W + (size_t)(n_blk + n) * ldw + k4);
    const int kc = k4 >> 5, kk = k4 & 31;
    *reinterpret_cast<float4 *>(&Ws[(kc * BN + n) * LD + kk]) = v;
  }
  if (PRO)
    for (int i = tid; i < K; i += 256) {
      sPa[i] = pa[i];
      sPb[i] = pb[i];
    }
  __syncthreads();

  const int nstrips = (R + 31) / 32;
  const int wstride = gridDim.x * 4;
  int strip = blockIdx.x * 4 + wave;
  float4 ra[D];
  const float *arow = A + (size_t)(strip * 32 + l31) * lda + h * 16;   // this lane's row
  const size_t astep = (size_t)wstride * 32 * lda;
  bool rowok = strip * 32 + l31 < R;
#pragma unroll
  for (int g = 0; g < D; ++g) {
    ra[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rowok) ra[g] = *reinterpret_cast<const float4 *>(arow + (g >> 2) * 32 + (g & 3) * 4);
  }
  float s1[NJ], s2[NJ];
  double d1[NJ], d2[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    s1[j] = s2[j] = 0.f;
    d1[j] = d2[j] = 0.0;
  }
  int wofs = l31 * LD + h * 16;
  float4 bfc[NJ], fac = make_float4(1.f, 1.f, 1.f, 1.f), fbc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    bfc[j] = *reinterpret_cast<const float4 *>(Ws + wofs + (j * 32) * LD);
  if (PRO) {
    fac = *reinterpret_cast<const float4 *>(&sPa[h * 16]);
    fbc = *reinterpret_cast<const float4 *>(&sPb[h * 16]);
  }
  for (; strip < nstrips; strip += wstride) {
    asm volatile("" : "+v"(wofs));   // keep the W fragment reads inside the loop (no hoisting
    const float *wbase = Ws + wofs;  // of 128 loop-invariant registers, no spills)
    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
    const int r0 = strip * 32;
    const bool nrowok = (strip + wstride) * 32 + l31 < R;
    const float *nrow = arow + astep;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float4 a = ra[g % D];
      // refill this ring slot: group g + D of this strip, or of the next one
      {
        const int gn = g + D;
        ra[g % D] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gn < G) {
          if (rowok)
            ra[g % D] = *reinterpret_cast<const float4 *>(arow + (gn >> 2) * 32 + (gn & 3) * 4);
        } else {
          const int g2 = gn - G;
          if (nrowok)
            ra[g % D] = *reinterpret_cast<const float4 *>(nrow + (g2 >> 2) * 32 + (g2 & 3) * 4);
        }
      }
      // LDS operands of the NEXT group are requested before this group's MFMAs
      float4 bfn[NJ], fan, fbn;
      {
        const int gn = (g + 1) % G;
        const int kc = gn >> 2, t4 = (gn & 3) * 4;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          bfn[j] = *reinterpret_cast<const float4 *>(wbase + (kc * BN + j * 32) * LD + t4);
        if (PRO) {
          fan = *reinterpret_cast<const float4 *>(&sPa[kc * 32 + h * 16 + t4]);
          fbn = *reinterpret_cast<const float4 *>(&sPb[kc * 32 + h * 16 + t4]);
        }
      }
      if (PRO) {
        a.x = fmaxf(fmaf(fac.x, a.x, fbc.x), 0.f);
        a.y = fmaxf(fmaf(fac.y, a.y, fbc.y), 0.f);
        a.z = fmaxf(fmaf(fac.z, a.z, fbc.z), 0.f);
        a.w = fmaxf(fmaf(fac.w, a.w, fbc.w), 0.f);
        if (!rowok) a = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bfc[j].x, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bfc[j].y, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bfc[j].z, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bfc[j].w, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bfc[j] = bfn[j];
      fac = fan;
      fbc = fbn;
    }
    // ---- epilogue: D layout col = lane&31, row = (v&3) + 8*(v>>2) + 4*h
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = n_blk + j * 32 + l31;
      float *cp = C + (size_t)(r0 + 4 * h) * ldc + col;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int rr = (v & 3) + 8 * (v >> 2);
        const float c = acc[j][v];
        if (C != nullptr && r0 + 4 * h + rr < R && col < N) cp[(size_t)rr * ldc] = c;
        if (!STATS && C == nullptr && c == 123456.789f) part[0] = c;   // (keeps acc alive)
        if (STATS) {
          s1[j] += c;
          s2[j] = fmaf(c, c, s2[j]);
        }
      }
      if (STATS) {
        d1[j] += (double)s1[j];
        d2[j] += (double)s2[j];
        s1[j] = s2[j] = 0.f;
      }
    }
    arow = nrow;
    rowok = nrowok;
  }
  if (STATS) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      d1[j] += __shfl_xor(d1[j], 32);
      d2[j] += __shfl_xor(d2[j], 32);
      if (h == 0) {
        red[(0 * 4 + wave) * BN + j * 32 + l31] = d1[j];
        red[(1 * 4 + wave) * BN + j * 32 + l31] = d2[j];
      }
    }
    __syncthreads();
    for (int c = tid; c < 2 * BN; c += 256) {
      const int which = c / BN, col = c % BN;
      double s = 0.0;
      for (int w = 0; w < 4; ++w) s += red[(which * 4 + w) * BN + col];
      if (n_blk + col < N) part[((size_t)blockIdx.x * 2 + which) * N + n_blk + col] = (float)s;
    }
  }
}

template <int BN, int KC, bool PRO, bool STATS, int D, int WPS>
static void launch_v2(const float *A, int lda, const float *W, int ldw, float *C, int ldc, int R,
                      int N, const float *pa, const float *pb, float *part, int gx,
                      hipStream_t st) {
  constexpr int K = KC * 32;
  const size_t lds = sizeof(float) * (KC * BN * 36 + 2 * K) + sizeof(double) * 2 * 4 * BN;
  static bool set = false;
  if (!set) {
    CK(hipFuncSetAttribute((const void *)gemm_nt_v2<BN, KC, PRO, STATS, D, WPS>,
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    set = true;
  }
  hipLaunchKernelGGL((gemm_nt_v2<BN, KC, PRO, STATS, D, WPS>), dim3(gx, (N + BN - 1) / BN),
                     dim3(256), lds, st, A, lda, W, ldw, C, ldc, R, N, pa, pb, part);
}

static float time_it(std::function<void()> fn, int iters = 20) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) fn();
  CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int i = 0; i < iters; ++i) {
    CK(hipEventRecord(e0, 0));
    fn();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}


// pure matrix-pipe rate: NACC independent accumulators, operands in registers, no memory
template <int NACC, int WPS>
__global__ __launch_bounds__(256, WPS) void mfma_peak(float *out, int iters, float seed) {
  f32x16 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
  float a = seed * (threadIdx.x + 1), b = seed * 0.37f * (threadIdx.x + 3);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int j = 0; j < NACC; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
      a = a * 1.0001f + 0.001f;
      b = b * 0.9999f - 0.001f;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) s += acc[j][v];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int WPS>
static void run_peak(int blocks_per_cu) {
  float *out;
  CK(hipMalloc(&out, 256 * 256 * 8 * 4));
  const int iters = 4000;
  auto fn = [&]() {
    hipLaunchKernelGGL((mfma_peak<NACC, WPS>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, out,
                       iters, 0.001f);
  };
  const float t = time_it(fn, 5);
  const double fl = 2.0 * 32 * 32 * 2 * (double)NACC * 4 * iters * 4 * 256 * blocks_per_cu;
  printf("mfma_f32_32x32x2f32 peak: %d acc, %d WG/CU: %.1f TF\n", NACC, blocks_per_cu, fl / t / 1e9);
  hipFree(out);
}

template <int BN, int KC, int D, int WPS>
static void run(int R, int N, bool store, int gx_mult) {
  constexpr int K = KC * 32;
  std::vector<float> hA((size_t)R * K), hW((size_t)N * K), hpa(K), hpb(K);
  srand(1);
  for (auto &v : hA) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto &v : hW) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto &v : hpa) v = (float)rand() / RAND_MAX + 0.5f;
  for (auto &v : hpb) v = (float)rand() / RAND_MAX - 0.5f;
  float *A, *W, *C0, *C1, *pa, *pb, *part0, *part1;
  CK(hipMalloc(&A, hA.size() * 4));
  CK(hipMalloc(&W, hW.size() * 4));
  CK(hipMalloc(&C0, (size_t)R * N * 4));
  CK(hipMalloc(&C1, (size_t)R * N * 4));
  CK(hipMalloc(&pa, K * 4));
  CK(hipMalloc(&pb, K * 4));
  const int grid_ref = btr_sa_gemm_grid(R);
  const int gx = std::min((R + 127) / 128, 256 * gx_mult);
  CK(hipMalloc(&part0, (size_t)grid_ref * 2 * N * 4));
  CK(hipMalloc(&part1, (size_t)gx * 2 * N * 4));
  CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(pa, hpa.data(), K * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(pb, hpb.data(), K * 4, hipMemcpyHostToDevice));
  float *c0 = store ? C0 : nullptr, *c1 = store ? C1 : nullptr;
  auto ref = [&]() {
    if (btr_sa_gemm_nt(R, N, K, A, K, W, K, c0, N, pa, pb, part0, nullptr) != 0) {
      printf("ref failed: %s\n", btr_last_error());
      exit(1);
    }
  };
  auto v2 = [&]() { launch_v2<BN, KC, true, true, D, WPS>(A, K, W, K, c1, N, R, N, pa, pb, part1, gx, 0); };
  ref();
  v2();
  CK(hipDeviceSynchronize());
  // correctness: C and the column statistics
  double worst = 0, worst_s = 0;
  if (store) {
    std::vector<float> h0((size_t)R * N), h1((size_t)R * N);
    CK(hipMemcpy(h0.data(), C0, h0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost));
    double mx = 0;
    for (size_t i = 0; i < h0.size(); ++i) {
      mx = std::max(mx, (double)fabsf(h0[i]));
      worst = std::max(worst, (double)fabsf(h0[i] - h1[i]));
    }
    worst /= mx;
  }
  {
    std::vector<float> p0((size_t)grid_ref * 2 * N), p1((size_t)gx * 2 * N);
    CK(hipMemcpy(p0.data(), part0, p0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(p1.data(), part1, p1.size() * 4, hipMemcpyDeviceToHost));
    for (int w = 0; w < 2; ++w)
      for (int n = 0; n < N; ++n) {
        double a = 0, b = 0;
        for (int g = 0; g < grid_ref; ++g) a += p0[((size_t)g * 2 + w) * N + n];
        for (int g = 0; g < gx; ++g) b += p1[((size_t)g * 2 + w) * N + n];
        worst_s = std::max(worst_s, fabs(a - b) / (fabs(a) + 1e-9));
      }
  }
  const float t0 = time_it(ref), t1 = time_it(v2);
  const double fl = 2.0 * R * N * K;
  printf("D=%d wps=%d R=%7d N=%3d K=%3d store=%d grid x%d: lib %.1f us (%.1f TF)  v2 %.1f us (%.1f TF)  "
         "max rel diff C %.1e stats %.1e\n",
         D, WPS, R, N, K, (int)store, gx_mult, t0 * 1e3, fl / t0 / 1e9, t1 * 1e3, fl / t1 / 1e9, worst,
         worst_s);
  hipFree(A); hipFree(W); hipFree(C0); hipFree(C1); hipFree(pa); hipFree(pb);
  hipFree(part0); hipFree(part1);
}


// ---------------------------------------------------------------------------------------------
// v3 "bf16x6": the f32 product evaluated on the bf16 matrix pipe.  Every f32 operand is split
// into three bf16 pieces a = ah + am + al (round-to-nearest remainders: the split is exact),
// a bf16 x bf16 product is exact in f32, and the six largest of the nine cross terms are
// accumulated in f32:  ah*bh + (ah*bm + am*bh) + (ah*bl + al*bh + am*bm)   (dropped: 2^-24 and
// below relative to ah*bh -- the size of one f32 rounding).  v_mfma_f32_32x32x16_bf16 retires 16
// k per 32 cycles, v_mfma_f32_32x32x2_f32 2 k per 64: six bf16 instructions per 16 k are 2.67x
// the f32-input rate.  Same tiling / pipeline as the library kernel (128 x BN tile, 4 waves,
// K chunks of 32 through LDS), LDS holds three bf16 planes per operand.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct Split4 {
  bf16x4 h, m, l;
};
__device__ __forceinline__ Split4 split4(const float4 v) {
  Split4 s;
  const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 h = (__bf16)f[i];
    const float r1 = f[i] - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    s.h[i] = h;
    s.m[i] = m;
    s.l[i] = (__bf16)r2;
  }
  return s;
}

template <int BN, bool PRO, bool STATS, bool WIDE>
__global__ __launch_bounds__(256, 2) void gemm_nt_v3(const float *__restrict__ A, int lda,
                                                     const float *__restrict__ W, int ldw,
                                                     float *__restrict__ C, int ldc, int R, int N,
                                                     int K, const float *__restrict__ pa,
                                                     const float *__restrict__ pb,
                                                     float *__restrict__ part) {
  constexpr int BM = 128, BK = 32;
  constexpr int LP = 40;            // plane row pitch in bf16 (80 B: conflict-free b128 reads)
  constexpr int WN = BN / 64, WM = 4 / WN;
  constexpr int MI = BM / WM / 32, NJ = 2;
  __shared__ __attribute__((aligned(16))) __bf16 smem[3 * (BM + BN) * LP];
  __bf16(*As)[BM * LP] = reinterpret_cast<__bf16(*)[BM * LP]>(smem);
  __bf16(*Bs)[BN * LP] = reinterpret_cast<__bf16(*)[BN * LP]>(smem + 3 * BM * LP);
  static_assert(3 * (BM + BN) * LP * 2 >= 4 * 32 * 68 * 4, "transpose buffer");
  __shared__ double red[STATS ? 2 * WM * BN : 1];
  __shared__ __attribute__((aligned(16))) float sPa[PRO ? 512 : 4], sPb[PRO ? 512 : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;
  const int n_blk = blockIdx.y * BN;
  float s1[NJ], s2[NJ];
  double d1[NJ], d2[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    s1[j] = s2[j] = 0.f;
    d1[j] = d2[j] = 0.0;
  }
  const int ntiles = (R + BM - 1) / BM;
  const int nkc = (K + BK - 1) / BK;
  const int kq = (tid & 7) * 4;
  const int srow = tid >> 3;
  float4 ra[BM / 32], rb[BN / 32];
  auto fetch = [&](int tile, int kc) {
    const int r0 = tile * BM, kk = kc * BK + kq;
#pragma unroll
    for (int p = 0; p < BM / 32; ++p) {
      const int row = srow + 32 * p;
      ra[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + row < R && kk < K)
        ra[p] = *reinterpret_cast<const float4 *>(A + (size_t)(r0 + row) * lda + kk);
    }
#pragma unroll
    for (int p = 0; p < BN / 32; ++p) {
      const int row = srow + 32 * p;
      rb[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n_blk + row < N && kk < K)
        rb[p] = *reinterpret_cast<const float4 *>(W + (size_t)(n_blk + row) * ldw + kk);
    }
  };
  auto stage = [&](int tile, int kc) {
    const int r0 = tile * BM, kk = kc * BK + kq;
    float4 fa = make_float4(1.f, 1.f, 1.f, 1.f), fb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PRO && kk < K) {
      fa = *reinterpret_cast<const float4 *>(&sPa[kk]);
      fb = *reinterpret_cast<const float4 *>(&sPb[kk]);
    }
#pragma unroll
    for (int p = 0; p < BM / 32; ++p) {
      const int row = srow + 32 * p;
      float4 v = ra[p];
      if (PRO && r0 + row < R && kk < K) {
        v.x = fmaxf(fmaf(fa.x, v.x, fb.x), 0.f);
        v.y = fmaxf(fmaf(fa.y, v.y, fb.y), 0.f);
        v.z = fmaxf(fmaf(fa.z, v.z, fb.z), 0.f);
        v.w = fmaxf(fmaf(fa.w, v.w, fb.w), 0.f);
      }
      const Split4 sp = split4(v);
      *reinterpret_cast<bf16x4 *>(&As[0][row * LP + kq]) = sp.h;
      *reinterpret_cast<bf16x4 *>(&As[1][row * LP + kq]) = sp.m;
      *reinterpret_cast<bf16x4 *>(&As[2][row * LP + kq]) = sp.l;
    }
#pragma unroll
    for (int p = 0; p < BN / 32; ++p) {
      const int row = srow + 32 * p;
      const Split4 sp = split4(rb[p]);
      *reinterpret_cast<bf16x4 *>(&Bs[0][row * LP + kq]) = sp.h;
      *reinterpret_cast<bf16x4 *>(&Bs[1][row * LP + kq]) = sp.m;
      *reinterpret_cast<bf16x4 *>(&Bs[2][row * LP + kq]) = sp.l;
    }
  };
  if (PRO) {
    for (int i = tid; i < K; i += 256) {
      sPa[i] = pa[i];
      sPb[i] = pb[i];
    }
    __syncthreads();
  }
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x, 0);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int r0 = tile * BM;
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    for (int kc = 0; kc < nkc; ++kc) {
      stage(tile, kc);
      __syncthreads();
      if (kc + 1 < nkc)
        fetch(tile, kc + 1);
      else if (tile + (int)gridDim.x < ntiles)
        fetch(tile + gridDim.x, 0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[3][MI], bf[3][NJ];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
            af[q][i] = *reinterpret_cast<const bf16x8 *>(
                &As[q][(wm * (BM / WM) + i * 32 + l31) * LP + ks * 16 + h * 8]);
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            bf[q][j] = *reinterpret_cast<const bf16x8 *>(
                &Bs[q][(wn * 64 + j * 32 + l31) * LP + ks * 16 + h * 8]);
        }
        // smallest terms first: (l,h) (h,l) (m,m) | (m,h) (h,m) | (h,h)
#define BTR_T(QA, QB)                                                                     \
  _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[QA][i], bf[QB][j], acc[i][j], 0, 0, 0);
        BTR_T(2, 0)
        BTR_T(0, 2)
        BTR_T(1, 1)
        BTR_T(1, 0)
        BTR_T(0, 1)
        BTR_T(0, 0)
#undef BTR_T
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int col = n_blk + wn * 64 + j * 32 + l31;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int row = r0 + wm * (BM / WM) + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
          const float c = acc[i][j][v];
          if (!WIDE && C != nullptr && row < R && col < N) C[(size_t)row * ldc + col] = c;
          if (STATS) {
            s1[j] += c;
            s2[j] = fmaf(c, c, s2[j]);
          }
        }
      }
      if (WIDE && C != nullptr) {
        // the wave's 32 x 64 block through its own LDS region: written in the accumulator
        // layout (lanes along columns), read back as rows -> 16-byte stores, 4 rows x 256 B each
        float *T = reinterpret_cast<float *>(smem) + wave * (32 * 68);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int v = 0; v < 16; ++v)
            T[((v & 3) + 8 * (v >> 2) + 4 * h) * 68 + j * 32 + l31] = acc[i][j][v];
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the region is private to the wave
        __builtin_amdgcn_wave_barrier();
        const int rl = lane >> 4, c4 = (lane & 15) * 4;
        const int colw = n_blk + wn * 64 + c4;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const float4 v4 = *reinterpret_cast<const float4 *>(&T[(it * 4 + rl) * 68 + c4]);
          const int row = r0 + wm * (BM / WM) + i * 32 + it * 4 + rl;
          if (row < R && colw < N) *reinterpret_cast<float4 *>(C + (size_t)row * ldc + colw) = v4;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (WIDE && C != nullptr) __syncthreads();   // the regions overlay the staging buffers
    if (STATS) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        d1[j] += (double)s1[j];
        d2[j] += (double)s2[j];
        s1[j] = s2[j] = 0.f;
      }
    }
  }
  if (STATS) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      d1[j] += __shfl_xor(d1[j], 32);
      d2[j] += __shfl_xor(d2[j], 32);
      if (h == 0) {
        red[(0 * WM + wm) * BN + wn * 64 + j * 32 + l31] = d1[j];
        red[(1 * WM + wm) * BN + wn * 64 + j * 32 + l31] = d2[j];
      }
    }
    __syncthreads();
    for (int c = tid; c < 2 * BN; c += 256) {
      const int which = c / BN, col = c % BN;
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < WM; ++w) s += red[(which * WM + w) * BN + col];
      if (n_blk + col < N) part[((size_t)blockIdx.x * 2 + which) * N + n_blk + col] = (float)s;
    }
  }
}

// error of a GEMM result against a float64 host evaluation of sampled entries
static void err_vs_f64(const std::vector<float> &hA, const std::vector<float> &hW,
                       const std::vector<float> &hpa, const std::vector<float> &hpb,
                       const std::vector<float> &c, int R, int N, int K, const char *tag) {
  double worst = 0, mx = 0, sum = 0;
  int cnt = 0;
  for (int t = 0; t < 4000; ++t) {
    const int r = (int)(((long long)t * 7919 + 13) % R), n = (t * 31 + 5) % N;
    double acc = 0;
    for (int k = 0; k < K; ++k) {
      const float y = fmaxf(fmaf(hpa[k], hA[(size_t)r * K + k], hpb[k]), 0.f);
      acc += (double)y * (double)hW[(size_t)n * K + k];
    }
    const double e = fabs((double)c[(size_t)r * N + n] - acc);
    worst = std::max(worst, e);
    mx = std::max(mx, fabs(acc));
    sum += e;
    ++cnt;
  }
  printf("   %s vs float64: max abs err / max |C| = %.2e, mean abs err / max |C| = %.2e\n", tag,
         worst / mx, sum / cnt / mx);
}

template <int BN, bool WIDE>
static void run_v3(int R, int N, int K, bool store) {
  std::vector<float> hA((size_t)R * K), hW((size_t)N * K), hpa(K), hpb(K);
  srand(2);
  for (auto &v : hA) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto &v : hW) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto &v : hpa) v = (float)rand() / RAND_MAX + 0.5f;
  for (auto &v : hpb) v = (float)rand() / RAND_MAX - 0.5f;
  float *A, *W, *C0, *C1, *pa, *pb, *part0, *part1;
  CK(hipMalloc(&A, hA.size() * 4));
  CK(hipMalloc(&W, hW.size() * 4));
  CK(hipMalloc(&C0, (size_t)R * N * 4));
  CK(hipMalloc(&C1, (size_t)R * N * 4));
  CK(hipMalloc(&pa, K * 4));
  CK(hipMalloc(&pb, K * 4));
  const int gx = btr_sa_gemm_grid(R);
  CK(hipMalloc(&part0, (size_t)gx * 2 * N * 4));
  CK(hipMalloc(&part1, (size_t)gx * 2 * N * 4));
  CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(pa, hpa.data(), K * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(pb, hpb.data(), K * 4, hipMemcpyHostToDevice));
  float *c0 = store ? C0 : nullptr, *c1 = store ? C1 : nullptr;
  auto ref = [&]() {
    if (btr_sa_gemm_nt(R, N, K, A, K, W, K, c0, N, pa, pb, part0, nullptr) != 0) exit(1);
  };
  auto v3 = [&]() {
    hipLaunchKernelGGL((gemm_nt_v3<BN, true, true, WIDE>), dim3(gx, (N + BN - 1) / BN), dim3(256), 0, 0,
                       A, K, W, K, c1, N, R, N, K, pa, pb, part1);
  };
  ref();
  v3();
  CK(hipDeviceSynchronize());
  const float t0 = time_it(ref), t1 = time_it(v3);
  const double fl = 2.0 * R * N * K;
  printf("bf16x6 wide=%d R=%7d N=%3d K=%3d store=%d: lib f32 %.1f us (%.1f TF)  v3 %.1f us (%.1f TF f32-equivalent)\n",
         (int)WIDE, R, N, K, (int)store, t0 * 1e3, fl / t0 / 1e9, t1 * 1e3, fl / t1 / 1e9);
  if (store) {
    std::vector<float> h0((size_t)R * N), h1((size_t)R * N);
    CK(hipMemcpy(h0.data(), C0, h0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost));
    err_vs_f64(hA, hW, hpa, hpb, h0, R, N, K, "f32 MFMA");
    err_vs_f64(hA, hW, hpa, hpb, h1, R, N, K, "bf16x6  ");
  }
  hipFree(A); hipFree(W); hipFree(C0); hipFree(C1); hipFree(pa); hipFree(pb);
  hipFree(part0); hipFree(part1);
}

template <int BN, int KC, int D, int WPS>
static void run_loop_only(int R, int N) {
  constexpr int K = KC * 32;
  float *A, *W, *pa, *pb, *part;
  CK(hipMalloc(&A, (size_t)R * K * 4));
  CK(hipMalloc(&W, (size_t)N * K * 4));
  CK(hipMalloc(&pa, K * 4));
  CK(hipMalloc(&pb, K * 4));
  CK(hipMalloc(&part, 1024 * 2 * N * 4));
  CK(hipMemset(A, 0x3c, (size_t)R * K * 4));
  CK(hipMemset(W, 0x3c, (size_t)N * K * 4));
  CK(hipMemset(pa, 0x3c, K * 4));
  CK(hipMemset(pb, 0x3c, K * 4));
  const int gx = std::min((R + 127) / 128, 512);
  auto fn = [&]() { launch_v2<BN, KC, true, false, D, WPS>(A, K, W, K, nullptr, N, R, N, pa, pb, part, gx, 0); };
  const float t = time_it(fn);
  printf("loop only (no stats, no stores) R=%d N=%d K=%d: %.1f us (%.1f TF)\n", R, N, K, t * 1e3,
         2.0 * R * N * K / t / 1e9);
  hipFree(A); hipFree(W); hipFree(pa); hipFree(pb); hipFree(part);
}

int main() {
  run_v3<128, false>(706504, 128, 64, true);
  run_v3<128, true>(706504, 128, 64, true);
  run_v3<64, false>(706504, 64, 64, true);
  run_v3<64, true>(706504, 64, 64, true);
  run_v3<128, false>(114624, 256, 128, true);
  run_v3<128, true>(114624, 256, 128, true);
  run_v3<128, true>(114624, 128, 128, true);
  run_v3<128, true>(114624, 128, 256, true);
  run_v3<128, true>(65536, 256, 128, true);
  run_v3<128, true>(32768, 128, 128, true);
  run_v3<128, true>(8192, 256, 256, true);
  return 0;
}
