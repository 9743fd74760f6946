#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  // each lane gives an 8-byte aligned address: row-major image [16 rows r][64 cols c] of u16,
  // element (r, c) = r*64 + c
  int addr_elems;
  if (mode == 0) addr_elems = (l & 15) * 4 + (l >> 4) * 64;      // lane i: 4 contiguous elems
  else addr_elems = (l & 15) * 64 + (l >> 4) * 4;                 // lane i -> row i, cols 4*(l>>4)
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + addr_elems));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  for (int mode = 0; mode < 2; ++mode) {
    k<<<1, 64>>>(d, mode);
    unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
  }
}
