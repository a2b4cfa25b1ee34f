#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  unsigned l = threadIdx.x;
  unsigned a = l, b = 100 + l;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[l] = r[0]; out[64 + l] = r[1];
  auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[128 + l] = s[0]; out[192 + l] = s[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4); unsigned h[256];
  k<<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[4] = {"p16 r0", "p16 r1", "p32 r0", "p32 r1"};
  for (int q = 0; q < 4; ++q) { printf("%s:", nm[q]); for (int l = 0; l < 64; l += 8) printf(" [%u..]", h[q * 64 + l]); printf("\n"); }
  return 0;
}
