// mds_platform_hw.h - the gfx950 hardware touch-points of platform.h (included from there, after the vector typedefs).
#pragma once
// round-to-nearest-even in hardware
MDS_DEV bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
// two floats -> one dword of two bf16: exactly ONE v_cvt_pk_bf16_f32 (the scalar form followed by
// shift/or costs three more VALU ops per pair — a quarter of the GEMM epilogues' instructions)
MDS_DEV uint32_t pack2(float lo, float hi) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_){lo, hi}, bf16x2_));
}

MDS_DEV float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
MDS_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
MDS_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

typedef __bf16 bf16x8_hw __attribute__((ext_vector_type(8)));
MDS_DEV void mma16(const u16x8& a, const u16x8& b, f32x4& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_hw, a),
                                              __builtin_bit_cast(bf16x8_hw, b), c, 0, 0, 0);
}
MDS_DEV void mma16(const f32x8& a, const f32x8& b, f32x4& c) {
#pragma unroll
  for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
}
// Handing data from one workgroup to another INSIDE a launch (the split-K partial tiles of pw_fwd).  The 8 XCDs' L2s are not
// coherent with each other for ordinary accesses.  Measured alternatives: an agent-scope release fence is an L2 write-back per block
// and an acquire fence an L2 invalidate for the whole XCD (a 360-block launch ran 2.5x slower than not splitting); returning atomic
// exchanges are bounded by the atomic units (~0.14 us per block of 4096 exchanges).  What is used: agent-scope atomic STORES (sc1:
// written through, acknowledged at device scope), the block barrier (every lane waits for its acknowledgements), then the ticket;
// the consumer reads with agent-scope atomic loads (sc1), which go to the same coherence point.  No fence, no cache maintenance.
MDS_DEV void st_coherent4(float* p, const f32x4& v) {
  typedef unsigned long long u64;
  // (elements are copied to scalars first: __builtin_bit_cast applied to a vector-element lvalue reads element 0 on this compiler)
  const float f0 = v[0], f1 = v[1], f2 = v[2], f3 = v[3];
  const u64 lo = (u64)__builtin_bit_cast(uint32_t, f0) | ((u64)__builtin_bit_cast(uint32_t, f1) << 32);
  const u64 hi = (u64)__builtin_bit_cast(uint32_t, f2) | ((u64)__builtin_bit_cast(uint32_t, f3) << 32);
  __hip_atomic_store((u64*)p, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store((u64*)(p + 2), hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
MDS_DEV f32x4 ld_coherent4(const float* p) {
  typedef unsigned long long u64;
  const u64 a = __hip_atomic_load((const u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const u64 b = __hip_atomic_load((const u64*)(p + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (f32x4){__builtin_bit_cast(float, (uint32_t)a), __builtin_bit_cast(float, (uint32_t)(a >> 32)),
                 __builtin_bit_cast(float, (uint32_t)b), __builtin_bit_cast(float, (uint32_t)(b >> 32))};
}
extern thread_local void* mds_tl_stop_event;   // k_misc.hip (mds_launch_event)
extern thread_local int mds_tl_stop_uses;      // launches issued with the armed event since it was armed
// all of this lane's outstanding vector-memory operations (loads returned, stores acknowledged): s_waitcnt vmcnt(0)
MDS_DEV void mds_wait_stores() { __builtin_amdgcn_s_waitcnt(0x0F70); }   // gfx9 encoding: vmcnt = 0, expcnt / lgkmcnt = max (no wait)
// kernels of the DEPENDENT CHAIN raise their waves' issue priority (s_setprio): where they share a SIMD with a wave of the second
// stream (weight gradients), the chain's wave is served first.  -DMDS_CHAIN_PRIO_LEVEL=0 builds the library without it (A/B).
#ifndef MDS_CHAIN_PRIO_LEVEL
#define MDS_CHAIN_PRIO_LEVEL 3
#endif
#define MDS_CHAIN_PRIO() __builtin_amdgcn_s_setprio(MDS_CHAIN_PRIO_LEVEL)
#define MDS_SETPRIO(n) __builtin_amdgcn_s_setprio(n)   /* a role's issue priority inside one kernel (producer / consumer waves) */
#define MDS_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define MDS_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)   /* LLVM SchedGroupMask: VALU 2, SALU 4, MFMA 8, VMEM read 0x20, write 0x40, DS read 0x100 */
#define MDS_PIN_SGPR(x) asm volatile("" : "+s"(x))   /* keep a wave-uniform value in scalar registers (no rematerialisation at its uses) */
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
    if (mds_tl_stop_event) { /* armed by mds_launch_event: the kernel's own completion signal is the event */ \
      ++mds_tl_stop_uses;                                                                           \
      hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)mds_smem_, (hipStream_t)(stream), nullptr, (hipEvent_t)mds_tl_stop_event, 0, __VA_ARGS__); \
    } else                                                                                            \
      hipLaunchKernelGGL(kernel, grid, block, mds_smem_, (hipStream_t)(stream), __VA_ARGS__);      \
  } while (0)

MDS_DEV u16x4 lds_tr4(const bf16_t* p) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(u16x4, v);
}

MDS_DEV void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

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

// ------------------------------------------------------------------ direct-to-LDS pipeline (k_pwk8.hip)
// A K-streaming GEMM keeps several stages of operands in flight as LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave
// instruction, destination = wave-uniform LDS address + 16 * lane, no VGPRs).  hipcc drains such loads with s_waitcnt
// vmcnt(0) in front of every LDS access it can see and in front of __syncthreads(); so inside the pipelined loop every LDS
// access is issued through these helpers (inline asm, invisible to that bookkeeping), the barrier is the bare s_barrier,
// and the waits are counted by hand: vmcnt(n) = "at most n of MY loads still in flight" (they retire in issue order).
typedef uint32_t lds_t;   // byte address inside the workgroup's LDS allocation
MDS_DEV lds_t lds_addr_of(const void* p) { return (lds_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
MDS_DEV void glds16(const void* gsrc, lds_t dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)(uintptr_t)dst_wave_uniform, 16, 0, 0);
}
// 16 bytes per lane from global memory into registers, invisible to hipcc's waitcnt bookkeeping (beside LDS-DMA it would wait for
// vmcnt(0) at the first use, i.e. for everything issued since): usable only after the caller's own counted s_waitcnt + reg_pin.
MDS_DEV void gld16(u16x8& dst, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); }
MDS_DEV u16x8 lds_ld16(lds_t addr) {
  u16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
MDS_DEV void lds_st16(lds_t addr, const u16x8& v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// the value is usable below this point only (place after the s_waitcnt that covers its load)
template <typename V> MDS_DEV void reg_pin(V& v) { asm volatile("" : "+v"(v)); }
MDS_DEV void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <int N> MDS_DEV void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> MDS_DEV void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wave-uniform n; values without a case wait for the next smaller one (stricter, never wrong)
MDS_DEV void wait_vm_dyn(int n) {
#define MDS_VM_CASE(k) case k: wait_vm<k>(); break;
  switch (n < 0 ? 0 : (n > 40 ? 40 : n)) {
    MDS_VM_CASE(0) MDS_VM_CASE(1) MDS_VM_CASE(2) MDS_VM_CASE(3) MDS_VM_CASE(4) MDS_VM_CASE(5) MDS_VM_CASE(6) MDS_VM_CASE(7)
    MDS_VM_CASE(8) MDS_VM_CASE(9) MDS_VM_CASE(10) MDS_VM_CASE(11) MDS_VM_CASE(12) MDS_VM_CASE(13) MDS_VM_CASE(14) MDS_VM_CASE(15)
    MDS_VM_CASE(16) MDS_VM_CASE(17) MDS_VM_CASE(18) MDS_VM_CASE(19) MDS_VM_CASE(20) MDS_VM_CASE(21) MDS_VM_CASE(22) MDS_VM_CASE(23)
    MDS_VM_CASE(24) MDS_VM_CASE(25) MDS_VM_CASE(26) MDS_VM_CASE(27) MDS_VM_CASE(28) MDS_VM_CASE(29) MDS_VM_CASE(30) MDS_VM_CASE(31)
    MDS_VM_CASE(32) MDS_VM_CASE(33) MDS_VM_CASE(34) MDS_VM_CASE(35) MDS_VM_CASE(36) MDS_VM_CASE(37) MDS_VM_CASE(38) MDS_VM_CASE(39)
    MDS_VM_CASE(40)
  }
#undef MDS_VM_CASE
}
MDS_DEV void raw_barrier() { __builtin_amdgcn_s_barrier(); }
// eight consecutive floats at a wave-uniform address into SGPRs (s_load_dwordx8); usable after the next lgkmcnt(0) + sreg_pin
MDS_DEV f32x8 sld8(const float* p_uniform) {
  f32x8 v;
  asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(v) : "s"(p_uniform));
  return v;
}
MDS_DEV void sreg_pin(f32x8& v) { asm volatile("" : "+s"(v)); }
// p[idx] for a wave-uniform idx of a table no launch of the same stream is writing: through the scalar cache (s_load), i.e.
// neither a vector register per lane nor an entry on vmcnt
MDS_DEV float ld_uniform(const float* p, int idx) { return ((const __attribute__((address_space(4))) float*)(uintptr_t)p)[idx]; }

// CUs of the current device (launch-shape rules that count rounds of the chip ask once)
inline int mds_cu_count() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  return n;
}
