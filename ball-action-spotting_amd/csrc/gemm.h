// gemm.h — helpers shared by the MFMA kernels (k_pw.hip, k_conv.hip).
#pragma once
#include "elem.h"

#define PW_KC 32    // k per MFMA step
template <typename T> struct PwLd;           // LDS row pitch (elements): 16-byte aligned rows
template <> struct PwLd<bf16_t> { static const int v = PW_KC + 8; };
template <> struct PwLd<float> { static const int v = PW_KC + 4; };

// reduce-scatter of N per-lane partials over the 16 lanes sharing q: afterwards vals[0..N/16-1] hold
// the 16-lane totals of entries e0.. with e0 = (N/16) * (8*b3 + 4*b2 + 2*b1 + b0) (b = bits of i).
template <int MASK, int HALF, int N>
MDS_DEV void reduce_scatter_step(float (&vals)[N], bool bit) {
#pragma unroll
  for (int k = 0; k < HALF; ++k) {
    const float lo = vals[k], hi = vals[k + HALF];
    const float send = bit ? lo : hi, keep = bit ? hi : lo;
    vals[k] = keep + __shfl_xor(send, MASK);  // (the DPP row_xor<MASK> costs the MFMA kernels ~60 VGPRs here: spills)
  }
}
MDS_DEV int reduce_scatter32(float (&vals)[32], int i) {
  reduce_scatter_step<8, 16>(vals, (i & 8) != 0);
  reduce_scatter_step<4, 8>(vals, (i & 4) != 0);
  reduce_scatter_step<2, 4>(vals, (i & 2) != 0);
  reduce_scatter_step<1, 2>(vals, (i & 1) != 0);
  return ((i & 8) ? 16 : 0) + ((i & 4) ? 8 : 0) + ((i & 2) ? 4 : 0) + ((i & 1) ? 2 : 0);
}
// reduce-scatter of 16 per-lane partials over the 16 lanes sharing q; returns the entry index
MDS_DEV int reduce_scatter16(float (&vals)[16], int i) {
  reduce_scatter_step<8, 8>(vals, (i & 8) != 0);
  reduce_scatter_step<4, 4>(vals, (i & 4) != 0);
  reduce_scatter_step<2, 2>(vals, (i & 2) != 0);
  reduce_scatter_step<1, 1>(vals, (i & 1) != 0);
  return ((i & 8) ? 8 : 0) + ((i & 4) ? 4 : 0) + ((i & 2) ? 2 : 0) + ((i & 1) ? 1 : 0);
}

template <typename T> struct RawV8;   // 8 consecutive elements as loaded (no conversion yet)
template <> struct RawV8<bf16_t> {
  u16x8 v;
  MDS_DEV void ld(const bf16_t* p) { v = *(const u16x8*)p; }
  MDS_DEV void zero() { v = (u16x8){0, 0, 0, 0, 0, 0, 0, 0}; }
  MDS_DEV void get(float (&o)[8]) const {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = bf2f(v[j]);
  }
  MDS_DEV void st(bf16_t* p) const { *(u16x8*)p = v; }
};
template <> struct RawV8<float> {
  f32x4 a, b;
  MDS_DEV void ld(const float* p) { a = *(const f32x4*)p; b = *(const f32x4*)(p + 4); }
  MDS_DEV void zero() { a = (f32x4){0, 0, 0, 0}; b = a; }
  MDS_DEV void get(float (&o)[8]) const {
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  }
  MDS_DEV void st(float* p) const { *(f32x4*)p = a; *(f32x4*)(p + 4) = b; }
};

template <typename T> struct RawV4;   // 4 consecutive elements as loaded
template <> struct RawV4<bf16_t> {
  u16x4 v;
  MDS_DEV void ld(const bf16_t* p) { v = *(const u16x4*)p; }
  MDS_DEV float get(int j) const { return bf2f(v[j]); }
};
template <> struct RawV4<float> {
  f32x4 v;
  MDS_DEV void ld(const float* p) { v = *(const f32x4*)p; }
  MDS_DEV float get(int j) const { return v[j]; }
};

// exact a / b for 0 <= a < 2^22 with rb = 1.0f / b (cheap replacement for integer division)
MDS_DEV int fdiv(int a, float rb) { return (int)(((float)a + 0.5f) * rb); }

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
  for (int j = 0; j < 4; ++j) d[j] = pack2(v[2 * j], v[2 * j + 1]);
}

