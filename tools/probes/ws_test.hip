#include <hip/hip_runtime.h>
#include <stdio.h>
#define DEV __device__ __forceinline__
template <int CTRL, int BANK = 0xF>
DEV float dpp(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xF, BANK, false));
}
__global__ void k(float* out) {
  int l = threadIdx.x;
  float v = (float)(1 << (l & 15)) ;  // row-distinct bits
  float a = v + dpp<0xB1>(v, v); out[l] = a;
  float b = a + dpp<0x4E>(a, a); out[64 + l] = b;
  float c = b + dpp<0x141>(b, b); out[128 + l] = c;
  float d = c + dpp<0x140>(c, c); out[192 + l] = d;
}
int main() {
  float* d; hipMalloc(&d, 256 * 4); float h[256];
  k<<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int q = 0; q < 4; ++q) { printf("s%d:", q); for (int l = 0; l < 16; ++l) printf(" %g", h[q * 64 + l]); printf("\n"); }
  return 0;
}
