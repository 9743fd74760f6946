// What a kernel resident on a SECOND queue costs the launches of the first:
// K dependent launches on stream A -- (a) one empty workgroup, (b) 8 192 empty workgroups of 256
// threads, (c) 2 048 workgroups streaming 64 MB, (d) 248 x 2 workgroups streaming the same -- timed
// with hipEvents, alone and while ONE / EIGHT sleeping 1 024-thread workgroups sit on stream B.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probe/queue_probe tools/probe/queue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(1024) void sleeper(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
__global__ __launch_bounds__(256) void empty_k(int *p) {
  if (p && threadIdx.x == 9999) p[0] = 1;
}
__global__ __launch_bounds__(256) void stream_k(const float4 *__restrict__ a, float4 *__restrict__ b,
                                                long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4;
       i += (long long)gridDim.x * 256) {
    float4 v = a[i];
    v.x += 1.f;
    b[i] = v;
  }
}

int main() {
  hipStream_t A, Bs;
  hipStreamCreateWithFlags(&A, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&Bs, hipStreamNonBlocking);
  const long long n4 = (64ll << 20) / 16;
  float4 *x, *y;
  hipMalloc(&x, n4 * 16);
  hipMalloc(&y, n4 * 16);
  hipMemset(x, 0, n4 * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int K = 200;
  for (int occ : {0, 1, 8}) {
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        hipDeviceSynchronize();
        if (occ) hipLaunchKernelGGL(sleeper, dim3(occ), dim3(1024), 0, Bs, 100ll * 20000);  // 20 ms
        hipEventRecord(e0, A);
        for (int i = 0; i < K; ++i) {
          if (mode == 0) hipLaunchKernelGGL(empty_k, dim3(1), dim3(256), 0, A, (int *)nullptr);
          if (mode == 1) hipLaunchKernelGGL(empty_k, dim3(8192), dim3(256), 0, A, (int *)nullptr);
          if (mode == 2) hipLaunchKernelGGL(stream_k, dim3(2048), dim3(256), 0, A, x, y, n4);
          if (mode == 3) hipLaunchKernelGGL(stream_k, dim3(496), dim3(256), 0, A, x, y, n4);
        }
        hipEventRecord(e1, A);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        hipDeviceSynchronize();
      }
      const char *names[4] = {"1 empty workgroup", "8192 empty workgroups", "2048 workgroups, 64 MB copy",
                              "496 workgroups, 64 MB copy"};
      printf("sleepers on the other stream: %d | %-28s %7.2f us per launch\n", occ, names[mode],
             1e3f * best / K);
    }
  }
  return 0;
}
