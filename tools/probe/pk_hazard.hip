// pk_hazard.hip -- stand-alone reproducer for the packed-f32 incident.
//
// Round 3: with SLP-vectorised code (v_pk_add / mul / fma_f32 with op_sel operands on register
// pairs) the register-resident FPS kernel (csrc/sampling.hip fps_regs_kernel<4, 8>, pyramid
// level 2048 -> 1024) returned a WRONG sample sequence in 1-3 % of its launches, but only while
// kernels of other streams shared its CUs -- never alone.  The library has been built with
// -fno-slp-vectorize and an asm fence around the three coordinate differences since, and
// build.py fails on any packed f32 op in an index-producing object.  This program rebuilds the
// FAILING form (same source, -fslp-vectorize, fence compiled out) beside the shipped form and
// runs both under co-running load, so that the hazard can be re-tested on the next compiler or
// part without the training loop around it:
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics \
//       -fslp-vectorize -DBTR_PK_REPRO_NO_FENCE -DBTR_FMAD=1 -Ibacktoreality_amd/csrc -Iinclude \
//       tools/probe/pk_hazard.hip -o /tmp/pk_hazard_slp          # the form that failed
//   hipcc ... -fno-slp-vectorize -DBTR_FMAD=1 ... -o /tmp/pk_hazard_ok  # the shipped form
//   /tmp/pk_hazard_slp [launches=4000] [co-runners: 0 none | 1 stream copy | 2 + fma waves]
//
// Input: an FPS-ordered prefix (the kernel's own output on a random cloud, gathered), so the
// right answer of every launch is 0, 1, 2, ...; a device kernel counts the launches that differ.
// Prints: launches, wrong launches, first wrong position -- and the count of v_pk_*_f32 in the
// disassembly is what `llvm-objdump -d` of the binary's code object shows (tools/probe/
// pk_hazard.sh does both builds, both runs and the disassembly count).
//
// What was observed when this was written (round 4, ROCm 7.2 clang 22.0.0git, MI355X): see
// profiles/r04_pk_hazard.txt.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

// the kernels under test, from the library source itself (namespace btr)
#include "../../backtoreality_amd/csrc/sampling.hip"

// the library's FPS entry points reference the bucketed kernels of fps_bucket.hip; the probe only
// launches fps_regs_kernel, so give the linker inert definitions
namespace btr {
bool fps_bucket_supported(int) { return false; }
size_t fps_bucket_workspace_bytes(int, int) { return 0; }
int fps_bucket_launch(int, int, int, const float *, int *, int, int, void *, size_t, hipStream_t) {
  return -1;
}
}  // namespace btr

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(2);                                                                 \
    }                                                                          \
  } while (0)

__global__ void gather3(int n, int m, const float *src, const int *idx, float *dst) {
  const int t = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (t < m * 3) dst[(size_t)b * m * 3 + t] = src[((size_t)b * n + idx[(size_t)b * m + t / 3]) * 3 + t % 3];
}

// wrong[0] += 1 when scene-launch (b) is not 0..m-1; wrong[1] = min first wrong position
__global__ void judge(int m, const int *idx, unsigned *wrong) {
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0x7fffffff;
  __syncthreads();
  for (int j = threadIdx.x; j < m; j += 256)
    if (idx[(size_t)blockIdx.x * m + j] != j) atomicMin(&bad, j);
  __syncthreads();
  if (threadIdx.x == 0 && bad != 0x7fffffff) {
    atomicAdd(&wrong[0], 1u);
    atomicMin(&wrong[1], (unsigned)bad);
  }
}

// co-runners: a streaming copy (HBM + all CUs) and dependent-FMA waves (VALU pressure)
__global__ void stream_copy(const float4 *a, float4 *b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    b[i] = a[i];
}
__global__ void fma_waves(float *out, int iters) {
  float x = threadIdx.x * 1e-3f, y = 1.0001f, z = 0.5f, w = 0.25f;
  for (int i = 0; i < iters; ++i) {
    x = x * y + z;
    w = w * y + x;
    z = z * y + w;
  }
  if (x + w + z == 123.f) out[0] = x;
}

int main(int argc, char **argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 4000;
  const int load = argc > 2 ? atoi(argv[2]) : 2;
  const int B = 8, N0 = 4096, N = 2048, M = 1024;
  std::vector<float> h((size_t)B * N0 * 3);
  unsigned s = 12345u;
  for (auto &v : h) {
    s = s * 1664525u + 1013904223u;
    v = 0.5f + 6.0f * (float)(s >> 8) / 16777216.0f;
  }
  float *cloud, *prefix;
  int *idx0, *idx;
  unsigned *wrong;
  CK(hipMalloc(&cloud, h.size() * 4));
  CK(hipMalloc(&prefix, (size_t)B * N * 3 * 4));
  CK(hipMalloc(&idx0, (size_t)B * N * 4));
  CK(hipMalloc(&idx, (size_t)B * M * 4));
  CK(hipMalloc(&wrong, 8));
  CK(hipMemcpy(cloud, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const unsigned init[2] = {0u, 0xffffffffu};
  CK(hipMemcpy(wrong, init, 8, hipMemcpyHostToDevice));
  hipStream_t sa, sb, sc;
  CK(hipStreamCreate(&sa));
  CK(hipStreamCreate(&sb));
  CK(hipStreamCreate(&sc));
  // an FPS-ordered prefix: 4096 -> 2048 by the same kernel family, alone on the chip
  hipLaunchKernelGGL((btr::fps_regs_kernel<4, 16>), dim3(B), dim3(256), 0, sa, N0, N, 512, 9,
                     cloud, idx0, (const int *)nullptr, 0);
  hipLaunchKernelGGL(gather3, dim3((N * 3 + 255) / 256, B), dim3(256), 0, sa, N0, N, cloud, idx0,
                     prefix);
  CK(hipStreamSynchronize(sa));
  // co-runner buffers
  const size_t cn = (size_t)64 << 20;   // 64 Mi float4 = 1 GiB moved per copy launch
  float4 *ca = nullptr, *cb = nullptr;
  float *fo = nullptr;
  if (load >= 1) {
    CK(hipMalloc(&ca, cn * 16 / 4));
    CK(hipMalloc(&cb, cn * 16 / 4));
    CK(hipMemset(ca, 0, cn * 16 / 4));
  }
  CK(hipMalloc(&fo, 4));
  for (int it = 0; it < launches; ++it) {
    if (load >= 1 && it % 4 == 0)
      hipLaunchKernelGGL(stream_copy, dim3(2048), dim3(256), 0, sb, ca, cb, cn / 4);
    if (load >= 2 && it % 2 == 0)
      hipLaunchKernelGGL(fma_waves, dim3(1024), dim3(256), 0, sc, fo, 20000);
    hipLaunchKernelGGL((btr::fps_regs_kernel<4, 8>), dim3(B), dim3(256), 0, sa, N, M, 512, 9,
                       prefix, idx, (const int *)nullptr, 0);
    hipLaunchKernelGGL(judge, dim3(B), dim3(256), 0, sa, M, idx, wrong);
    if (it % 64 == 63) CK(hipDeviceSynchronize());   // bound the queues
  }
  CK(hipDeviceSynchronize());
  unsigned res[2];
  CK(hipMemcpy(res, wrong, 8, hipMemcpyDeviceToHost));
  printf("{\"launches\": %d, \"scene_launches\": %d, \"co_runners\": %d, \"wrong_scene_launches\": %u, "
         "\"first_wrong_position\": %d}\n",
         launches, launches * B, load, res[0], res[0] ? (int)res[1] : -1);
  return 0;
}
