// What a captured HIP graph of small dependent kernels costs against the same launches issued
// one by one, and whether two captured branches (fork / join through a side stream) overlap:
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_out/graph_cost tools/probe/graph_cost.hip && gpurun_out/graph_cost
// The kernels model the GroupFree3D decoder stack's: ~8 us on 40 workgroups (most CUs idle).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
struct Big { float *p[24]; int n[16]; };
__global__ __launch_bounds__(256) void k_spin(Big b, int iters) {
  float v = (float)threadIdx.x;
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  if (v == 123.456f) b.p[0][blockIdx.x] = v;
}
static double now() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 400;
  const int iters = argc > 2 ? atoi(argv[2]) : 3000;
  const int wgs = argc > 3 ? atoi(argv[3]) : 40;
  hipStream_t s, s2;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  float *p;
  CK(hipMalloc(&p, 64 << 20));
  hipEvent_t ev, ev2;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
  Big b{};
  b.p[0] = p;
  auto chain = [&](hipStream_t st, int n) {
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(256), 0, st, b, iters);
  };
  // every `every` kernels of the main chain, `side` kernels are forked onto s2 and joined at the
  // end of the group
  auto forked = [&](int n, int every, int side) {
    for (int i = 0; i < n; i += every) {
      CK(hipEventRecord(ev, s));
      CK(hipStreamWaitEvent(s2, ev, 0));
      chain(s2, side);
      CK(hipEventRecord(ev2, s2));
      chain(s, every);
      CK(hipStreamWaitEvent(s, ev2, 0));
    }
  };
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now();
    chain(s, N);
    double t1 = now();
    CK(hipStreamSynchronize(s));
    double t2 = now();
    printf("eager, one stream, %d kernels: host %.2f us each, to idle %.2f us each\n", N,
           (t1 - t0) / N, (t2 - t0) / N);
    t0 = now();
    forked(N / 2, 10, 10);
    t1 = now();
    CK(hipDeviceSynchronize());
    t2 = now();
    printf("eager, %d kernels as two branches of 10 (fork / join per 10): host %.2f us per kernel, "
           "to idle %.2f us per kernel\n", N, (t1 - t0) / N, (t2 - t0) / N);
  }
  for (int variant = 0; variant < 2; ++variant) {
    hipGraph_t g;
    hipGraphExec_t ge;
    double t0 = now();
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    if (variant == 0) chain(s, N); else forked(N / 2, 10, 10);
    CK(hipStreamEndCapture(s, &g));
    double t1 = now();
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    double t2 = now();
    printf("%s: capture %.1f us, instantiate %.1f us\n", variant ? "two branches" : "one chain",
           t1 - t0, t2 - t1);
    for (int rep = 0; rep < 3; ++rep) {
      const int R = 10;
      t0 = now();
      for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
      t1 = now();
      CK(hipStreamSynchronize(s));
      t2 = now();
      printf("  graph replay: host %.1f us per launch (%.2f per kernel), to idle %.2f us per kernel\n",
             (t1 - t0) / R, (t1 - t0) / R / N, (t2 - t0) / R / N);
    }
    // one replay between eager launches on the same stream (the library's use)
    t0 = now();
    for (int r = 0; r < 10; ++r) {
      chain(s, 5);
      CK(hipGraphLaunch(ge, s));
    }
    t1 = now();
    CK(hipStreamSynchronize(s));
    t2 = now();
    printf("  5 eager + graph, x10: host %.1f us per round, to idle %.1f us per round\n",
           (t1 - t0) / 10, (t2 - t0) / 10);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  {  // two LINEAR graphs on two streams, launched back to back: do they overlap?
    hipGraph_t g[2];
    hipGraphExec_t ge[2];
    hipStream_t st[2] = {s, s2};
    for (int j = 0; j < 2; ++j) {
      CK(hipStreamBeginCapture(st[j], hipStreamCaptureModeRelaxed));
      chain(st[j], N / 2);
      CK(hipStreamEndCapture(st[j], &g[j]));
      CK(hipGraphInstantiate(&ge[j], g[j], nullptr, nullptr, 0));
    }
    for (int rep = 0; rep < 3; ++rep) {
      const int R = 10;
      double t0 = now();
      for (int r = 0; r < R; ++r) {
        CK(hipEventRecord(ev, s));
        CK(hipStreamWaitEvent(s2, ev, 0));
        CK(hipGraphLaunch(ge[1], s2));
        CK(hipEventRecord(ev2, s2));
        CK(hipGraphLaunch(ge[0], s));
        CK(hipStreamWaitEvent(s, ev2, 0));
      }
      double t1 = now();
      CK(hipDeviceSynchronize());
      double t2 = now();
      printf("two linear graphs of %d kernels on two streams (fork / join around them): host %.1f us "
             "per round, to idle %.2f us per kernel\n", N / 2, (t1 - t0) / R, (t2 - t0) / R / N);
    }
    // the same with 6 + 6 short graphs per round (the decoder stack's per-layer segments)
    hipGraph_t h[2];
    hipGraphExec_t he[2];
    for (int j = 0; j < 2; ++j) {
      CK(hipStreamBeginCapture(st[j], hipStreamCaptureModeRelaxed));
      chain(st[j], N / 12);
      CK(hipStreamEndCapture(st[j], &h[j]));
      CK(hipGraphInstantiate(&he[j], h[j], nullptr, nullptr, 0));
    }
    for (int rep = 0; rep < 3; ++rep) {
      const int R = 10;
      double t0 = now();
      for (int r = 0; r < R; ++r)
        for (int seg = 0; seg < 6; ++seg) {
          CK(hipGraphLaunch(he[0], s));
          CK(hipEventRecord(ev, s));
          CK(hipStreamWaitEvent(s2, ev, 0));
          CK(hipGraphLaunch(he[1], s2));
          CK(hipEventRecord(ev2, s2));
        }
      double t1 = now();
      CK(hipDeviceSynchronize());
      double t2 = now();
      printf("6 x (graph of %d on main, event, graph of %d on side): host %.1f us per round, to idle "
             "%.2f us per kernel\n", N / 12, N / 12, (t1 - t0) / R, (t2 - t0) / R / (N / 12 * 12));
    }
  }
  return 0;
}
