// mds_platform_hw.h (test simulator) - host emulation of the hardware touch-points of csrc/platform.h: same contracts,
// scalar arithmetic.  Found instead of csrc/mds_platform_hw.h because tests/hipemu/ is first on the simulator build's include
// path.  Test infrastructure only.
#pragma once
MDS_DEV bf16_t f2bf(float f) {  // round-to-nearest-even
  uint32_t u = f2bits(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
MDS_DEV uint32_t pack2(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }

MDS_DEV void st_coherent4(float* p, const f32x4& v) { for (int j = 0; j < 4; ++j) p[j] = v[j]; }
MDS_DEV f32x4 ld_coherent4(const float* p) { f32x4 v; for (int j = 0; j < 4; ++j) v[j] = p[j]; return v; }
MDS_DEV float fast_exp(float x) { return expf(x); }
MDS_DEV float fast_exp2(float x) { return exp2f(x); }
MDS_DEV float fast_rcp(float x) { return 1.0f / x; }

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
MDS_DEV void mds_wait_stores() {}
#define MDS_CHAIN_PRIO() ((void)0)
#define MDS_SETPRIO(n) ((void)0)
#define MDS_SCHED_FENCE() ((void)0)
#define MDS_SCHED_GROUP(mask, n) ((void)0)
#define MDS_PIN_SGPR(x) ((void)0)
#define MDS_UNIFORM(x) (x)
#define MDS_DYN_SMEM(name) char* name = hipemu::dyn_smem()
#define MDS_LAUNCH(kernel, grid, block, smem, stream, ...) \
  hipemu::launch(grid, block, smem, [=]() { kernel(__VA_ARGS__); })

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

MDS_DEV void wave_lds_sync() { hipemu::wave_barrier(); }

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

// direct-to-LDS pipeline touch-points (k_pwk8.hip): the simulator copies at issue time and the waits are no-ops, so only the
// addressing / slot arithmetic is exercised here - the counted waits themselves are checked on the MI355X
typedef uint32_t lds_t;
MDS_DEV lds_t lds_addr_of(const void* p) { return (lds_t)((const char*)p - hipemu::dyn_smem()); }
MDS_DEV void glds16(const void* gsrc, lds_t dst_wave_uniform) { memcpy(hipemu::dyn_smem() + dst_wave_uniform + 16 * hipemu::lane_id(), gsrc, 16); }
MDS_DEV u16x8 lds_ld16(lds_t addr) { u16x8 v; memcpy(&v, hipemu::dyn_smem() + addr, 16); return v; }
MDS_DEV void lds_st16(lds_t addr, const u16x8& v) { memcpy(hipemu::dyn_smem() + addr, &v, 16); }
template <typename V> MDS_DEV void reg_pin(V&) {}
MDS_DEV void wait_lgkm0() {}
template <int N> MDS_DEV void wait_lgkm() {}
template <int N> MDS_DEV void wait_vm() {}
MDS_DEV void wait_vm_dyn(int) {}
MDS_DEV void raw_barrier() { hipemu::block_barrier(); }
MDS_DEV float ld_uniform(const float* p, int idx) { return p[idx]; }
MDS_DEV f32x8 sld8(const float* p) { f32x8 v; for (int j = 0; j < 8; ++j) v[j] = p[j]; return v; }
MDS_DEV void sreg_pin(f32x8&) {}
MDS_DEV void gld16(u16x8& dst, const void* p) { memcpy(&dst, p, 16); }

inline int mds_cu_count() { return hipemu_cu_count(); }
