// platform.h — device-side vocabulary shared by every kernel file of libmds_hip.so.
//
// Target: gfx950 (MI355X, CDNA4), wave64, hipcc.  The only other build of these sources is the
// host *simulator* used by the CPU test-suite (tests/hipemu/hipemu.h, -DMDS_EMU): it replaces
// the three hardware touch-points below (HIP runtime header, MFMA builtins, kernel launch) and
// nothing else.  It is test infrastructure, not a fallback: the product library is hipcc-only.
#pragma once
#include <stdint.h>

#ifndef MDS_EMU
#include <hip/hip_runtime.h>
#include <atomic>
#endif

#include "../../include/mds.h"

#define MDS_DEV __device__ __forceinline__
#define MDS_WAVE 64

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ scalar helpers
MDS_DEV float bits2f(uint32_t u) { return __builtin_bit_cast(float, u); }
MDS_DEV uint32_t f2bits(float f) { return __builtin_bit_cast(uint32_t, f); }
MDS_DEV float bf2f(bf16_t v) { return bits2f((uint32_t)v << 16); }
#ifndef MDS_EMU
// round-to-nearest-even in hardware
MDS_DEV bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
// two floats -> one dword of two bf16: exactly ONE v_cvt_pk_bf16_f32 (the scalar form followed by
// shift/or costs three more VALU ops per pair — a quarter of the GEMM epilogues' instructions)
MDS_DEV uint32_t pack2(float lo, float hi) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_){lo, hi}, bf16x2_));
}
#else
MDS_DEV bf16_t f2bf(float f) {  // round-to-nearest-even
  uint32_t u = f2bits(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
MDS_DEV uint32_t pack2(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }
#endif
// SiLU/sigmoid run on every element of every activation tensor (several times, because the
// normalised tensor is never materialised): hardware v_exp_f32 / v_rcp_f32 (~1 ulp) instead of
// the multi-instruction libm expansions.
#ifndef MDS_EMU
MDS_DEV float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
MDS_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#else
MDS_DEV float fast_exp(float x) { return expf(x); }
MDS_DEV float fast_rcp(float x) { return 1.0f / x; }
#endif
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

#ifndef MDS_EMU
typedef __bf16 bf16x8_hw __attribute__((ext_vector_type(8)));
MDS_DEV void mma16(const u16x8& a, const u16x8& b, f32x4& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_hw, a),
                                              __builtin_bit_cast(bf16x8_hw, b), c, 0, 0, 0);
}
MDS_DEV void mma16(const f32x8& a, const f32x8& b, f32x4& c) {
#pragma unroll
  for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
}
#define MDS_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define MDS_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)  /* wave-uniform value -> scalar register */
#define MDS_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
// gfx950 has 160 KiB of LDS per CU; launches above the 64 KiB default opt in once per kernel.
#define MDS_LAUNCH(kernel, grid, block, smem, stream, ...)                                          \
  do {                                                                                              \
    const size_t mds_smem_ = (size_t)(smem);                                                        \
    if (mds_smem_ > 65536) { /* opt-in is idempotent; the high-water mark only avoids repeating it */ \
      static std::atomic<size_t> mds_cur_{0};                                                       \
      if (mds_smem_ > mds_cur_.load(std::memory_order_relaxed)) {                                   \
        (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mds_smem_); \
        mds_cur_.store(mds_smem_, std::memory_order_relaxed);                                       \
      }                                                                                             \
    }                                                                                               \
    hipLaunchKernelGGL(kernel, grid, block, mds_smem_, (hipStream_t)(stream), __VA_ARGS__);        \
  } while (0)
#else  // ---- host simulator (tests only): same contracts, scalar arithmetic
MDS_DEV void mma16_emu(const float (&a)[8], const float (&b)[8], f32x4& c, bool round_bf16) {
  int lane = hipemu::lane_id();
  float* mine = (float*)hipemu::wave_scratch(lane);
  for (int j = 0; j < 8; ++j) { mine[j] = a[j]; mine[8 + j] = b[j]; }
  hipemu::wave_barrier();
  int i = lane & 15, q = lane >> 4;
  for (int r = 0; r < 4; ++r) {
    int row = 4 * q + r, col = i;
    float acc = c[r];
    for (int k = 0; k < 32; ++k) {
      float av = ((float*)hipemu::wave_scratch(row + 16 * (k >> 3)))[k & 7];
      float bv = ((float*)hipemu::wave_scratch(col + 16 * (k >> 3)))[8 + (k & 7)];
      acc += av * bv;
    }
    c[r] = acc;
  }
  (void)round_bf16;
  hipemu::wave_barrier();
}
MDS_DEV void mma16(const u16x8& a, const u16x8& b, f32x4& c) {
  float fa[8], fb[8];
  for (int j = 0; j < 8; ++j) { fa[j] = bf2f(a[j]); fb[j] = bf2f(b[j]); }
  mma16_emu(fa, fb, c, true);
}
MDS_DEV void mma16(const f32x8& a, const f32x8& b, f32x4& c) {
  float fa[8], fb[8];
  for (int j = 0; j < 8; ++j) { fa[j] = a[j]; fb[j] = b[j]; }
  mma16_emu(fa, fb, c, false);
}
#define MDS_SCHED_FENCE() ((void)0)
#define MDS_UNIFORM(x) (x)
#define MDS_DYN_SMEM(name) char* name = hipemu::dyn_smem()
#define MDS_LAUNCH(kernel, grid, block, smem, stream, ...) \
  hipemu::launch(grid, block, smem, [=]() { kernel(__VA_ARGS__); })
#endif

// ------------------------------------------------------------------ transposing LDS read (gfx950)
// ds_read_b64_tr_b16: every lane passes the (8-byte aligned) LDS address of 4 consecutive bf16;
// within a 16-lane group, lane m's four values are row m>>2, columns 4(m&3)..4(m&3)+3 of a
// [4][16] block, and lane i RECEIVES column i of that block (rows 0..3).  The rows may sit at
// arbitrary addresses, so a [pixel][channel] image yields MFMA fragments whose reduction index is
// the pixel — the weight-gradient GEMMs — in one instruction per 4 pixels (measured on hardware
// with tools/probes/tr_test.hip; tests/test_k_conv.py covers it through conv_wgrad).
#ifndef MDS_EMU
MDS_DEV u16x4 lds_tr4(const bf16_t* p) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(u16x4, v);
}
#else
MDS_DEV u16x4 lds_tr4(const bf16_t* p) {
  const int lane = hipemu::lane_id();
  hipemu::wave_scratch(lane)[0] = (uint64_t)(uintptr_t)p;
  hipemu::wave_barrier();
  const int i = lane & 15, g = lane & ~15;
  u16x4 out;
  for (int j = 0; j < 4; ++j) {
    const bf16_t* src = (const bf16_t*)(uintptr_t)hipemu::wave_scratch(g + 4 * j + (i >> 2))[0];
    out[j] = src[i & 3];
  }
  hipemu::wave_barrier();
  return out;
}
#endif

// Orders one wave's own LDS traffic (lane A's write, lane B's read) without a block barrier: the LDS unit runs a wave's
// DS instructions in order, so only the compiler (and the simulator's lane interleaving) must be held back.
#ifndef MDS_EMU
MDS_DEV void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
#else
MDS_DEV void wave_lds_sync() { hipemu::wave_barrier(); }
#endif

MDS_DEV void frag_from8(u16x8& f, const float (&v)[8]) { f = pack8(v); }
MDS_DEV void frag_from8(f32x8& f, const float (&v)[8]) { f = (f32x8){v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]}; }
MDS_DEV void frag_zero(u16x8& f) { f = (u16x8){0, 0, 0, 0, 0, 0, 0, 0}; }
MDS_DEV void frag_zero(f32x8& f) { f = (f32x8){0, 0, 0, 0, 0, 0, 0, 0}; }

// ------------------------------------------------------------------ wave / block reductions
// Cross-lane exchange inside a 16-lane row runs on the VALU's DPP path (one 4-cycle op: quad_perm
// for xor 1/2, two bank-masked row shifts for xor 4, row_ror:8 for xor 8) — __shfl_xor compiles to
// ds_bpermute_b32, an LDS-crossbar instruction, and the statistic epilogues issued 30+ of them per
// tile.  Semantics checked on hardware (row_shl:n -> lane i receives lane i+n).
#ifndef MDS_EMU
template <int CTRL, int BANK = 0xF>
MDS_DEV float dpp_f(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xF, BANK, false));
}
template <int M> MDS_DEV float row_xor(float v);  // value of lane (i ^ M), M < 16
template <> MDS_DEV float row_xor<1>(float v) { return dpp_f<0xB1>(v, v); }
template <> MDS_DEV float row_xor<2>(float v) { return dpp_f<0x4E>(v, v); }
template <> MDS_DEV float row_xor<4>(float v) { return dpp_f<0x114, 0xA>(dpp_f<0x104, 0x5>(v, v), v); }
template <> MDS_DEV float row_xor<8>(float v) { return dpp_f<0x128>(v, v); }
// sum over the 16 lanes that share q = lane >> 4 (i.e. over i = lane & 15); every lane gets it
MDS_DEV float sum_over_i16(float v) {
  v += dpp_f<0xB1>(v, v); v += dpp_f<0x4E>(v, v);
  v += dpp_f<0x141>(v, v);  // row_half_mirror: pairs the two quads of each half row
  v += dpp_f<0x140>(v, v);  // row_mirror: pairs the half rows
  return v;
}
MDS_DEV float wave_sum(float v) {
  v = sum_over_i16(v);
  const int u = __builtin_bit_cast(int, v);
  return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 16))) +
         (__builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(u, 48)));
}
#else
template <int M> MDS_DEV float row_xor(float v) { return __shfl_xor(v, M); }
MDS_DEV float sum_over_i16(float v) {
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
  return v;
}
MDS_DEV float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
#endif

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

// ------------------------------------------------------------------ developer switches
// MDS_*_OLD environment variables select the previous kernel generation of a family for A/B timing.
// They are read ONCE per process (k_misc.hip), never on the launch path.
enum { MDS_SW_DW_OLD = 0, MDS_SW_CONV_OLD, MDS_SW_WG_OLD, MDS_SW_STEM_OLD, MDS_SW_COUNT };
bool mds_switch(int id);
int mds_knob(int id);   // mds_dev_set() values (0 = default), see include/mds.h

// ------------------------------------------------------------------ error plumbing (C ABI)
void mds_set_error(const char* fmt, ...);
int mds_check_launch(const char* what);
#define MDS_REQUIRE(cond, ...) \
  do { if (!(cond)) { mds_set_error(__VA_ARGS__); return MDS_ERR_BAD_ARG; } } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
