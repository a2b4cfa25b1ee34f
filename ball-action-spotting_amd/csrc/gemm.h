// gemm.h — helpers shared by the MFMA kernels (k_pw.hip, k_conv.hip).
#pragma once
#include "elem.h"

#define PW_KC 32    // k per MFMA step
template <typename T> struct PwLd;           // LDS row pitch (elements): 16-byte aligned rows
template <> struct PwLd<bf16_t> { static const int v = PW_KC + 8; };
template <> struct PwLd<float> { static const int v = PW_KC + 4; };

// reduce-scatter of 32 per-lane partials over the 16 lanes sharing q: afterwards vals[0..1] hold
// the 16-lane totals of entries e0, e0+1 with e0 = 16*b3 + 8*b2 + 4*b1 + 2*b0 (b = bits of i).
MDS_DEV int reduce_scatter32(float (&vals)[32], int i) {
  int e0 = 0;
#pragma unroll
  for (int step = 0; step < 4; ++step) {
    const int half = 16 >> step, mask = 8 >> step;
    const bool bit = (i & mask) != 0;
#pragma unroll
    for (int k = 0; k < half; ++k) {
      float lo = vals[k], hi = vals[k + half];
      float send = bit ? lo : hi;
      float keep = bit ? hi : lo;
      vals[k] = keep + __shfl_xor(send, mask);
    }
    if (bit) e0 += half;
  }
  return e0;
}


// 16-byte aligned fragment loads from LDS (8 consecutive k of one row)
MDS_DEV u16x8 ld_frag(const bf16_t* p) { return *(const u16x8*)p; }
MDS_DEV f32x8 ld_frag(const float* p) {
  f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
  return (f32x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

MDS_DEV void lds_store8_u32(float* dst, const float (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) dst[j] = v[j];
}
MDS_DEV void lds_store8_u32(bf16_t* dst, const float (&v)[8]) {
  uint32_t* d = (uint32_t*)dst;  // pitch is even -> 4-byte aligned
#pragma unroll
  for (int j = 0; j < 4; ++j) d[j] = (uint32_t)f2bf(v[2 * j]) | ((uint32_t)f2bf(v[2 * j + 1]) << 16);
}

