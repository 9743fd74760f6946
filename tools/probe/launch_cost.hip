// Host cost of the HIP calls the library makes per dispatch, on the box at hand:
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_out/launch_cost tools/probe/launch_cost.hip && gpurun_out/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
struct Big { float *p[24]; int n[16]; };
__global__ void k_small(float *p, int n) { if (n < 0) p[0] = 1.f; }
__global__ void k_big(Big b) { if (b.n[0] < 0) b.p[0][0] = 1.f; }
__global__ void k_work(float *p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}
static double now() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
int main() {
  hipStream_t s, s2;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  float *p;
  hipMalloc(&p, 64 << 20);
  hipEvent_t ev;
  hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  Big b{};
  b.p[0] = p;
  const int N = 2000;
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, p, 1);
    double t1 = now();
    hipStreamSynchronize(s);
    double t2 = now();
    printf("small-arg launch: host %.2f us each, to idle %.2f us each\n", (t1 - t0) / N, (t2 - t0) / N);
    t0 = now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, b);
    t1 = now();
    hipStreamSynchronize(s);
    t2 = now();
    printf("256-byte-arg launch: host %.2f us each, to idle %.2f us each\n", (t1 - t0) / N, (t2 - t0) / N);
    t0 = now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_work, dim3(4096), dim3(256), 0, s, p, 1 << 20);
    t1 = now();
    hipStreamSynchronize(s);
    t2 = now();
    printf("4 MB rw kernel: host %.2f us each, to idle %.2f us each\n", (t1 - t0) / N, (t2 - t0) / N);
    t0 = now();
    for (int i = 0; i < N; ++i) {
      hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, p, 1);
      hipEventRecord(ev, s);
      hipStreamWaitEvent(s2, ev, 0);
      hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s2, p, 1);
    }
    t1 = now();
    hipDeviceSynchronize();
    t2 = now();
    printf("launch + record + wait + launch(other stream): host %.2f us per iteration, to idle %.2f\n",
           (t1 - t0) / N, (t2 - t0) / N);
    t0 = now();
    for (int i = 0; i < N; ++i) hipMemsetAsync(p, 0, 4096, s);
    t1 = now();
    hipStreamSynchronize(s);
    t2 = now();
    printf("hipMemsetAsync 4 KB: host %.2f us each, to idle %.2f us each\n", (t1 - t0) / N, (t2 - t0) / N);
    t0 = now();
    for (int i = 0; i < N; ++i) { volatile const char *e = getenv("BTR_SOME_SWITCH"); (void)e; }
    t1 = now();
    printf("getenv: %.3f us each\n", (t1 - t0) / N);
  }
  return 0;
}
