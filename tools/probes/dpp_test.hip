#include <hip/hip_runtime.h>
#include <stdio.h>
#define DEV __device__ __forceinline__
template <int CTRL, int BANK = 0xF>
DEV float dpp(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xF, BANK, false));
}
template <int M> DEV float xshfl(float v);
template <> DEV float xshfl<1>(float v) { return dpp<0xB1>(v, v); }
template <> DEV float xshfl<2>(float v) { return dpp<0x4E>(v, v); }
template <> DEV float xshfl<4>(float v) { float t = dpp<0x104, 0x5>(v, v); return dpp<0x114, 0xA>(t, v); }
template <> DEV float xshfl<8>(float v) { return dpp<0x128>(v, v); }
DEV float wsum(float v) {
  v += dpp<0xB1>(v, v); v += dpp<0x4E>(v, v); v += dpp<0x141>(v, v); v += dpp<0x140>(v, v);
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  unsigned u = __builtin_bit_cast(unsigned, v), w = u;
  asm volatile("" : "+v"(w));
  auto r = __builtin_amdgcn_permlane16_swap(u, w, false, false);
  v = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
  u = __builtin_bit_cast(unsigned, v); w = u;
  asm volatile("" : "+v"(w));
  auto s = __builtin_amdgcn_permlane32_swap(u, w, false, false);
  return __builtin_bit_cast(float, s[0]) + __builtin_bit_cast(float, s[1]);
}
__global__ void k(float* out) {
  int l = threadIdx.x;
  float v = (float)l;
  out[l] = xshfl<1>(v); out[64 + l] = xshfl<2>(v); out[128 + l] = xshfl<4>(v); out[192 + l] = xshfl<8>(v);
  out[256 + l] = wsum((float)(l * l % 17) + 0.5f);
  out[320 + l] = dpp<0x141>(v, v); out[384 + l] = dpp<0x140>(v, v);
}
int main() {
  float* d; hipMalloc(&d, 448 * 4); float h[448];
  k<<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    if (h[l] != (l ^ 1)) bad++; if (h[64 + l] != (l ^ 2)) bad++; if (h[128 + l] != (l ^ 4)) bad++; if (h[192 + l] != (l ^ 8)) bad++;
  }
  float ref = 0; for (int l = 0; l < 64; ++l) ref += (float)(l * l % 17) + 0.5f;
  for (int l = 0; l < 64; ++l) if (h[256 + l] != ref) bad++;
  printf("bad=%d ref=%g got=%g %g\n", bad, ref, h[256], h[256 + 63]);
  printf("x4:"); for (int l = 0; l < 16; ++l) printf(" %g", h[128 + l]); printf("\nhm:"); for (int l = 0; l < 16; ++l) printf(" %g", h[320 + l]);
  printf("\nm:"); for (int l = 0; l < 16; ++l) printf(" %g", h[384 + l]); printf("\n");
  return 0;
}
