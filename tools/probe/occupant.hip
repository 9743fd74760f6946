// A stand-in for the large-scene FPS kernel's footprint, to find out WHAT about it costs the
// backbone forward 240 us when they run side by side (tools/fps_interference.py --occupant):
// `wgs` workgroups of 1024 threads that stay resident for `usec` microseconds and
//   mode 0  sleep (s_sleep): only the wave slots / registers / the workgroup slot are taken
//   mode 1  spin on the VALU: issue slots of their CUs as well
//   mode 2  chase pointers through a buffer that fits L2 (the FPS's access pattern: dependent
//           loads that hit L2), one load in flight per wave
//   mode 3  stream 16-byte loads through the same buffer with many in flight (L2 bandwidth)
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/probe/liboccupant.so tools/probe/occupant.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(1024) void occupant_kernel(int mode, long long ticks,
                                                        const int *__restrict__ chain, int n,
                                                        float *__restrict__ sink) {
  const long long t0 = wall_clock64();   // 100 MHz constant clock
  float acc = (float)threadIdx.x;
  int p = (threadIdx.x * 97 + blockIdx.x * 13) % n;
  const float4 *buf4 = reinterpret_cast<const float4 *>(chain);
  while (wall_clock64() - t0 < ticks) {
    if (mode == 0) {
      __builtin_amdgcn_s_sleep(64);
    } else if (mode == 1) {
#pragma unroll
      for (int i = 0; i < 64; ++i) acc = fmaf(acc, 1.0000001f, 0.5f);
    } else if (mode == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) p = chain[p];
    } else {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 v = buf4[(p + i * 1024) % (n / 4)];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      acc += s.x + s.y + s.z + s.w;
      p = (p + 8192 + 64) % (n / 4);
    }
  }
  if (acc == 12345.678f || p == -7) sink[0] = acc + (float)p;
}

// a single-wave variant (64 threads per workgroup): is it the resident KERNEL or its footprint?
__global__ __launch_bounds__(64) void occupant_small_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
extern "C" int occupant_launch_small(int wgs, int usec, void *stream) {
  hipLaunchKernelGGL(occupant_small_kernel, dim3(wgs), dim3(64), 0, (hipStream_t)stream,
                     (long long)usec * 100);
  return (int)hipGetLastError();
}

// (lds_bytes of unused dynamic LDS, as the FPS launch holds: occupant_launch_lds)
extern "C" int occupant_launch_lds(int wgs, int mode, int usec, const int *chain, int n,
                                   float *sink, void *stream, int lds_bytes) {
  static int set = 0;
  if (lds_bytes > set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&occupant_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    set = lds_bytes;
  }
  hipLaunchKernelGGL(occupant_kernel, dim3(wgs), dim3(1024), lds_bytes, (hipStream_t)stream, mode,
                     (long long)usec * 100, chain, n, sink);
  return (int)hipGetLastError();
}

extern "C" int occupant_launch(int wgs, int mode, int usec, const int *chain, int n, float *sink,
                               void *stream) {
  hipLaunchKernelGGL(occupant_kernel, dim3(wgs), dim3(1024), 0, (hipStream_t)stream, mode,
                     (long long)usec * 100, chain, n, sink);
  return (int)hipGetLastError();
}
