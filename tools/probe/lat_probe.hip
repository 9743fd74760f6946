// lat_probe.hip -- dependent-chain latency of a wave-wide float4 load (1 KB per wave) from a
// scene-sized buffer, the access the bucketed FPS kernel makes once per touched bucket.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/lat_probe.hip -o tools/probe/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// STORE: 0 none, 1 .w only (strided dwords), 2 whole float4, 3 separate contiguous t array,
// 4 = 1 but only every 4th wave stores, 5 = 3 with a nontemporal store
template <int NW, int STORE>
__global__ __launch_bounds__(NW * 64) void probe(float4 *buf, int nb, int iters,
                                                 unsigned long long *out, float *sink,
                                                 float *tarr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  buf += (size_t)blockIdx.x * nb * 64;
  tarr += (size_t)blockIdx.x * nb * 64;
  unsigned b = (wave * 37u + 11u) % nb;
  float acc = 0.f;
  // warm the cache: touch the whole buffer once
  for (int i = threadIdx.x; i < nb * 64; i += NW * 64) acc += buf[i].x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    float4 p = buf[(size_t)b * 64 + lane];
    if (STORE == 1 || (STORE == 4 && (wave & 3) == 0)) buf[(size_t)b * 64 + lane].w = p.w + 1.f;
    if (STORE == 2) {
      p.w += 1.f;
      buf[(size_t)b * 64 + lane] = p;
    }
    if (STORE == 3 || STORE == 5) {
      const float t = tarr[(size_t)b * 64 + lane];
      p.y += t;
      if (STORE == 3) tarr[(size_t)b * 64 + lane] = t + 1.f;
      else __builtin_nontemporal_store(t + 1.f, &tarr[(size_t)b * 64 + lane]);
    }
    // next bucket depends on the loaded value (dependent chain)
    const unsigned v = __builtin_amdgcn_readfirstlane(__float_as_uint(p.x));
    b = (b * 1664525u + 1013904223u + (v & 1u)) % nb;
    acc += p.y;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * NW + wave] = t1 - t0;
  if (acc == 123.456f) sink[0] = acc;
}

template <int NW, int STORE>
static void run(const char *name, int blocks, int nb, int iters) {
  float4 *buf;
  unsigned long long *out;
  float *sink;
  hipMalloc(&buf, sizeof(float4) * 64 * (size_t)nb * blocks);
  hipMemset(buf, 0, sizeof(float4) * 64 * (size_t)nb * blocks);
  hipMalloc(&out, 8 * NW * blocks);
  hipMalloc(&sink, 4);
  float *tarr;
  hipMalloc(&tarr, 4 * 64 * (size_t)nb * blocks);
  hipMemset(tarr, 0, 4 * 64 * (size_t)nb * blocks);
  for (int rep = 0; rep < 2; ++rep)
    hipLaunchKernelGGL((probe<NW, STORE>), dim3(blocks), dim3(NW * 64), 0, 0, buf, nb, iters, out,
                       sink, tarr);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(NW * blocks);
  hipMemcpy(h.data(), out, 8 * NW * blocks, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("%-28s blocks %d waves %2d buffer %5d KB/block: %.0f cycles per dependent load\n", name,
         blocks, NW, nb, s / h.size() / iters);
  hipFree(buf); hipFree(out); hipFree(sink);
}

int main() {
  const int it = 2000;
  run<1, 0>("load only", 1, 16, it);
  run<16, 0>("load only", 8, 640, it);
  run<1, 1>("load + store .w", 1, 640, it);
  run<16, 1>("load + store .w", 8, 640, it);
  run<16, 4>("  ... every 4th wave", 8, 640, it);
  run<16, 2>("load + store float4", 8, 640, it);
  run<16, 3>("load xyz,t + store t[]", 8, 640, it);
  run<16, 5>("  ... nontemporal store", 8, 640, it);
  run<4, 1>("load + store .w", 8, 640, it);
  run<8, 1>("load + store .w", 8, 640, it);
  return 0;
}
