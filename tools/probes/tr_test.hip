#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int e = threadIdx.x; e < 4096; e += 64) lds[e] = e;
  __syncthreads();
  int l = threadIdx.x;
  int off;
  if (mode == 0) off = (l >> 4) * 64 + (l & 15) * 4;            // contiguous [4][16] block per group
  else off = (l >> 4) * 1024 + ((l & 15) >> 2) * 100 + (l & 3) * 4;  // rows at pitch 100 (8B aligned), 16 cols
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 512);
  unsigned short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    k<<<1, 64>>>(d, mode); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); if (l == 19) l = 47; }
  }
  return 0;
}
