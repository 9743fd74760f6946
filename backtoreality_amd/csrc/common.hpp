// common.hpp -- shared helpers for the gfx950 kernels behind include/btr_pointnet2.h.
// Wave = 64 lanes everywhere (CDNA4); no other target is supported.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/btr_pointnet2.h"

namespace btr {

constexpr int kWave = 64;

// Thread-local error text behind btr_last_error().
inline char *err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

inline int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

// The reference printf+exit(-1)s on a failed launch (include/cuda_utils.h:35-44); we report.
inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
  return BTR_OK;
}

inline hipStream_t as_stream(btr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------- wave helpers
// DPP lane exchange inside a row of 16 lanes (VALU-latency, no LDS crossbar trip).
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

// max over each row of 16 lanes, result replicated in all 16 lanes of the row.
__device__ __forceinline__ unsigned row16_max_u32(unsigned v) {
  unsigned t;
  t = dpp_u32<0xB1>(v);  v = v > t ? v : t;  // quad_perm [1,0,3,2]
  t = dpp_u32<0x4E>(v);  v = v > t ? v : t;  // quad_perm [2,3,0,1]
  t = dpp_u32<0x141>(v); v = v > t ? v : t;  // row_half_mirror
  t = dpp_u32<0x140>(v); v = v > t ? v : t;  // row_mirror
  return v;
}

// max over the whole 64-lane wave, returned wave-uniform (SGPR).
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = row16_max_u32(v);
  unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0);
  unsigned b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32);
  unsigned d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  a = a > b ? a : b;
  c = c > d ? c : d;
  return a > c ? a : c;
}

__device__ __forceinline__ float row16_sum_f32(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));
  return v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it
// waits for every outstanding GLOBAL store/load of the wave; kernels whose waves exchange data
// through LDS alone (the FPS arg-max slots) must not pay a memory round trip per barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// ------------------------------------------------------------- the one distance expression
// Every index the path produces (FPS arg-max, ball-query membership, 3-NN order) is decided by
// the f32 rounding of  a*a + b*b + c*c  with a = x2 - x1 ... (reference sampling_gpu.cu:108-109,
// :105 for |p|^2; ball_query_gpu.cu:36-38; interpolate_gpu.cu:38).  The reference is built by
// nvcc -O2 with the default --fmad=true (pointnet2/setup.py:22-25), which CONTRACTS the
// expression; how is decided by the compiler, not the source, so the rounding is a build
// mode here (BTR_FMAD, one library per mode, see build.py / DESIGN.md section 2):
//   1 (default)  fma(c, c, fma(a, a, b*b))  -- the NVPTX/LLVM contraction of ((a*a + b*b) + c*c):
//                an fadd whose FIRST operand is an fmul fuses that operand, else the second
//   2            fma(c, c, fma(b, b, a*a))  -- the left-to-right chain
//   0            ((a*a) + (b*b)) + (c*c)    -- as written, no contraction (--fmad=false)
// All files are compiled with -ffp-contract=off, so nothing else fuses.  Each form is monotone
// non-decreasing in |a|, |b|, |c| (products of equal factors and rounded sums / fmas of
// non-negative terms are), which is what the FPS box-pruning argument needs (fps_bucket.hip).
#ifndef BTR_FMAD
#define BTR_FMAD 1
#endif
__device__ __forceinline__ float sq3(float a, float b, float c) {
#if BTR_FMAD == 1
  return __builtin_fmaf(c, c, __builtin_fmaf(a, a, b * b));
#elif BTR_FMAD == 2
  return __builtin_fmaf(c, c, __builtin_fmaf(b, b, a * a));
#else
  return a * a + b * b + c * c;
#endif
}
// p1*w1 + p2*w2 + p3*w3 (three_interpolate, interpolate_gpu.cu:103-104) under the same rule.
__device__ __forceinline__ float dot3(float p1, float w1, float p2, float w2, float p3,
                                      float w3) {
#if BTR_FMAD == 1
  return __builtin_fmaf(p3, w3, __builtin_fmaf(p1, w1, p2 * w2));
#elif BTR_FMAD == 2
  return __builtin_fmaf(p3, w3, __builtin_fmaf(p2, w2, p1 * w1));
#else
  return p1 * w1 + p2 * w2 + p3 * w3;
#endif
}

}  // namespace btr

#define BTR_REQUIRE(cond, ...)                                              \
  do {                                                                      \
    if (!(cond)) return ::btr::fail(BTR_ERR_INVALID_ARGUMENT, __VA_ARGS__); \
  } while (0)
