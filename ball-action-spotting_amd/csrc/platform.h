// platform.h — device-side vocabulary shared by every kernel file of libmds_hip.so.
//
// Target: gfx950 (MI355X, CDNA4), wave64, hipcc.  Everything that touches the hardware directly (HIP runtime header, bf16
// conversion, v_exp / v_rcp, MFMA, the transposing LDS read, DPP reductions, kernel launch) lives in two headers found through
// the include path: mds_platform_rt.h and mds_platform_hw.h in this directory.  The CPU test-suite's kernel simulator
// (tests/hipemu/) compiles these same kernel sources with ITS files of those names first on the include path - there is no
// simulator code and no conditional compilation in the product sources.
#pragma once
#include <stdint.h>

#include <mds_platform_rt.h>   // the HIP runtime header (csrc/); the test simulator puts its own file of that name first on the include path

#include "../../include/mds.h"

#define MDS_DEV __device__ __forceinline__
#define MDS_WAVE 64

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ scalar helpers
MDS_DEV float bits2f(uint32_t u) { return __builtin_bit_cast(float, u); }
MDS_DEV uint32_t f2bits(float f) { return __builtin_bit_cast(uint32_t, f); }
MDS_DEV float bf2f(bf16_t v) { return bits2f((uint32_t)v << 16); }
#include <mds_platform_hw.h>   // hardware touch-points: bf16 conversion, v_exp / v_rcp, MFMA, transposing LDS read, DPP, launch
// SiLU/sigmoid run on every element of every activation tensor (several times, because the
// normalised tensor is never materialised): hardware v_exp_f32 / v_rcp_f32 (~1 ulp) instead of
// the multi-instruction libm expansions.
MDS_DEV float sigmoidf_(float x) { return fast_rcp(1.0f + fast_exp(-x)); }
MDS_DEV float siluf_(float x) { return x * sigmoidf_(x); }
// d silu(z)/dz
MDS_DEV float silu_gradf_(float z) { float s = sigmoidf_(z); return s * (1.0f + z * (1.0f - s)); }

template <typename T> struct Elem;
template <> struct Elem<float> {
  static MDS_DEV float ld(const float* p) { return *p; }
  static MDS_DEV void st(float* p, float v) { *p = v; }
  static MDS_DEV float rnd(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static MDS_DEV float ld(const bf16_t* p) { return bf2f(*p); }
  static MDS_DEV void st(bf16_t* p, float v) { *p = f2bf(v); }
  static MDS_DEV float rnd(float v) { return bf2f(f2bf(v)); }
};

// 8 consecutive elements <-> 8 floats (16-byte / 32-byte vector memory ops)
MDS_DEV void load8(const float* p, float (&v)[8]) {
  f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
MDS_DEV void load8(const bf16_t* p, float (&v)[8]) {
  u16x8 a = *(const u16x8*)p;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = bf2f(a[i]);
}
MDS_DEV void store8(float* p, const float (&v)[8]) {
  f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
  *(f32x4*)p = a; *(f32x4*)(p + 4) = b;
}
// 8 floats -> 8 packed bf16: built as 4 dwords so each pair is ONE v_cvt_pk_bf16_f32
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
MDS_DEV u16x8 pack8(const float (&v)[8]) {
  u32x4 w;
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = pack2(v[2 * i], v[2 * i + 1]);
  return __builtin_bit_cast(u16x8, w);
}
MDS_DEV void store8(bf16_t* p, const float (&v)[8]) { *(u16x8*)p = pack8(v); }
MDS_DEV void load4(const float* p, float (&v)[4]) { f32x4 a = *(const f32x4*)p; v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; }
MDS_DEV void load4(const bf16_t* p, float (&v)[4]) { u16x4 a = *(const u16x4*)p; for (int i = 0; i < 4; ++i) v[i] = bf2f(a[i]); }
MDS_DEV void store4(float* p, const float (&v)[4]) { f32x4 a = {v[0], v[1], v[2], v[3]}; *(f32x4*)p = a; }
MDS_DEV void store4(bf16_t* p, const float (&v)[4]) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  u32x2 w = {pack2(v[0], v[1]), pack2(v[2], v[3])};
  *(u32x2*)p = w;
}

// ------------------------------------------------------------------ MFMA tile op
// One wave computes C[16x16] += A[16x32] * B[32x16].  Lane l = (i = l & 15, q = l >> 4) holds
//   A[i][8q .. 8q+7]  and  B[8q .. 8q+7][i]      (8 consecutive k per lane, both operands)
// and C/D element r of lane l is C[row = 4q + r][col = i]     (guide §3 fragment layout).
// bf16: one v_mfma_f32_16x16x32_bf16.   f32: eight v_mfma_f32_16x16x4_f32 (exact fp32; MFMA j
// consumes element j of every lane — any k-permutation is legal as long as A and B agree).
template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef u16x8 type; };
template <> struct Frag<float> { typedef f32x8 type; };


// ------------------------------------------------------------------ transposing LDS read (gfx950)
// ds_read_b64_tr_b16: every lane passes the (8-byte aligned) LDS address of 4 consecutive bf16;
// within a 16-lane group, lane m's four values are row m>>2, columns 4(m&3)..4(m&3)+3 of a
// [4][16] block, and lane i RECEIVES column i of that block (rows 0..3).  The rows may sit at
// arbitrary addresses, so a [pixel][channel] image yields MFMA fragments whose reduction index is
// the pixel — the weight-gradient GEMMs — in one instruction per 4 pixels (measured on hardware
// with tools/probes/tr_test.hip; tests/test_k_conv.py covers it through conv_wgrad).

// Orders one wave's own LDS traffic (lane A's write, lane B's read) without a block barrier: the LDS unit runs a wave's
// DS instructions in order, so only the compiler (and the simulator's lane interleaving) must be held back.

// Split-precision product for the fp32 INFERENCE plans (X3): every fp32 operand is split in registers into two bf16 values,
// a = hi + lo with hi = bf16(a), lo = bf16(a - hi) (16 significant bits together), and a fragment product is three bf16 MFMAs
// lo*hi + hi*lo + hi*hi accumulated in fp32 - the dropped lo*lo term and the residual of the split are <= 2^-16 of a product
// (the exact-fp32 path issues eight v_mfma_f32_16x16x4_f32 per fragment pair at 1/16 of the bf16 rate: 2048 against 3 x 16 clocks;
// the split costs 3 VALU ops per element).  The fragment layout does not change: lane (i, q) holds k = 8q .. 8q+7 of row i in
// both forms.  Training plans (TAIL != EPI) and every bf16 kernel use Mma<T, false> = the plain mma16.
#ifndef MDS_EVAL_X3
#define MDS_EVAL_X3 1     /* 0: inference plans keep the exact fp32 MFMA (library A/B) */
#endif
struct FragX3 { u16x8 hi, lo; };
MDS_DEV FragX3 split_x3(const f32x8& v) {
  u32x4 h, l;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = v[2 * p], b = v[2 * p + 1];
    const uint32_t hh = pack2(a, b);
    const float ha = __builtin_bit_cast(float, hh << 16), hb = __builtin_bit_cast(float, hh & 0xffff0000u);
    h[p] = hh;
    l[p] = pack2(a - ha, b - hb);
  }
  FragX3 r;
  r.hi = __builtin_bit_cast(u16x8, h);
  r.lo = __builtin_bit_cast(u16x8, l);
  return r;
}
template <typename T, bool X3> struct Mma {
  typedef typename Frag<T>::type frag;
  static MDS_DEV frag prep(const typename Frag<T>::type& f) { return f; }
  static MDS_DEV void mma(const frag& a, const frag& b, f32x4& c) { mma16(a, b, c); }
};
template <> struct Mma<float, true> {
  typedef FragX3 frag;
  static MDS_DEV frag prep(const f32x8& f) { return split_x3(f); }
  static MDS_DEV void mma(const frag& a, const frag& b, f32x4& c) {
    mma16(a.lo, b.hi, c);
    mma16(a.hi, b.lo, c);
    mma16(a.hi, b.hi, c);
  }
};

MDS_DEV void frag_from8(u16x8& f, const float (&v)[8]) { f = pack8(v); }
MDS_DEV void frag_from8(f32x8& f, const float (&v)[8]) { f = (f32x8){v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]}; }
MDS_DEV void frag_zero(u16x8& f) { f = (u16x8){0, 0, 0, 0, 0, 0, 0, 0}; }
MDS_DEV void frag_zero(f32x8& f) { f = (f32x8){0, 0, 0, 0, 0, 0, 0, 0}; }

// ------------------------------------------------------------------ wave / block reductions
// Cross-lane exchange inside a 16-lane row runs on the VALU's DPP path (one 4-cycle op: quad_perm
// for xor 1/2, two bank-masked row shifts for xor 4, row_ror:8 for xor 8) — __shfl_xor compiles to
// ds_bpermute_b32, an LDS-crossbar instruction, and the statistic epilogues issued 30+ of them per
// tile.  Semantics checked on hardware (row_shl:n -> lane i receives lane i+n).

// ------------------------------------------------------------------ XCD-aware work assignment
// Workgroup id -> position in a kernel's work sequence.  The hardware hands consecutive workgroup ids to consecutive XCDs
// (id % 8; 8 XCDs, each with its own 4 MiB L2 - MI355X_MICROARCH.md, a speed assumption only: any placement is correct).
// Kernels whose neighbouring work items re-read the same rows (halos of the 3x3 / depthwise tiles, the tiles of one row
// split of a weight gradient) give every XCD a CONTIGUOUS range of the sequence, so the shared rows come from that XCD's L2
// instead of being fetched from HBM once per XCD.  Bijective for any n.
MDS_DEV unsigned xcd_contiguous(unsigned id, unsigned n) {
  const unsigned q = n >> 3, r = n & 7, xcd = id & 7, slot = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// ------------------------------------------------------------------ developer knobs
int mds_knob(int id);   // mds_dev_set() values (0 = default), see include/mds.h

// ------------------------------------------------------------------ error plumbing (C ABI)
void mds_set_error(const char* fmt, ...);
int mds_check_launch(const char* what);
#define MDS_REQUIRE(cond, ...) \
  do { if (!(cond)) { mds_set_error(__VA_ARGS__); return MDS_ERR_BAD_ARG; } } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
