// cu_mask_probe.hip -- which CUs does a stream created with hipExtStreamCreateWithCUMask use?
// A kernel of many small workgroups records (XCC_ID, SE, SH, CU) of every workgroup; the host
// prints, per mask, the distinct CUs per XCC that ran something.  Answers how mask bits map to
// the 8 x 32 CUs of an MI355X before the product pins its sampling stream to a few of them.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(2);                                                                  \
    }                                                                           \
  } while (0)

__global__ void where(unsigned *out, int spin) {
  // s_getreg_b32: simm16 = (size - 1) << 11 | offset << 6 | id;  HW_ID = 4, XCC_ID = 20
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
  volatile float x = 1.f;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;   // keep the CU busy so others fill
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | (hw & 0xffff);
}

static void run(const char *name, hipStream_t st, unsigned *dev, std::vector<unsigned> &host) {
  const int n = (int)host.size();
  hipLaunchKernelGGL(where, dim3(n), dim3(256), 0, st, dev, 20000);
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(host.data(), dev, n * 4, hipMemcpyDeviceToHost));
  std::set<unsigned> cus[16];
  for (unsigned v : host) {
    const unsigned xcc = v >> 16, hw = v & 0xffff;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cus[xcc & 15].insert((se << 8) | (sh << 4) | cu);
  }
  printf("%-28s", name);
  int total = 0;
  for (int x = 0; x < 8; ++x) {
    printf(" xcc%d:%2zu", x, cus[x].size());
    total += (int)cus[x].size();
  }
  printf("  total %d\n", total);
  if (total <= 24) {
    printf("    ");
    for (int x = 0; x < 8; ++x)
      for (unsigned c : cus[x]) printf("[x%d se%u sh%u cu%u] ", x, c >> 8, (c >> 4) & 1, c & 15);
    printf("\n");
  }
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("%s: %d CUs\n", prop.name, prop.multiProcessorCount);
  const int n = 4096;
  unsigned *dev;
  CK(hipMalloc(&dev, n * 4));
  std::vector<unsigned> host(n);
  hipStream_t plain;
  CK(hipStreamCreate(&plain));
  run("unmasked", plain, dev, host);
  struct Case {
    const char *name;
    unsigned m[8];
  };
  const Case cases[] = {
      {"bits 0-7", {0xffu, 0, 0, 0, 0, 0, 0, 0}},
      {"bits 0-15", {0xffffu, 0, 0, 0, 0, 0, 0, 0}},
      {"bits 0-31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}},
      {"bits 32-63", {0, 0xffffffffu, 0, 0, 0, 0, 0, 0}},
      {"bit 0 of every word", {1, 1, 1, 1, 1, 1, 1, 1}},
      {"bits 0,8,16,24 of word 0", {0x01010101u, 0, 0, 0, 0, 0, 0, 0}},
      {"all but bits 0-7", {0xffffff00u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}},
      {"all but bit 0 of every word", {~1u, ~1u, ~1u, ~1u, ~1u, ~1u, ~1u, ~1u}},
  };
  for (const Case &c : cases) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, c.m);
    if (e != hipSuccess) {
      printf("%-28s hipExtStreamCreateWithCUMask: %s\n", c.name, hipGetErrorString(e));
      continue;
    }
    run(c.name, st, dev, host);
    CK(hipStreamDestroy(st));
  }
  return 0;
}
