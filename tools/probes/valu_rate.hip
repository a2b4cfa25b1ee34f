// Probe: issue rates of the VALU instructions a SiLU prologue is made of, and of their packed-fp16 alternatives,
// on gfx950.  One wave per SIMD, 8 independent dependency chains per instruction kind, no memory traffic: cycles (s_memtime ticks at one
// wave per SIMD = shader clocks) per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate.hip -o tools/probes/build/valu_rate && tools/probes/build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(NAME, ASM, CONSTR, TYPE, INIT)                                                    \
  __global__ void NAME(float* out, unsigned long long* ticks, int iters) {                        \
    TYPE r0 = INIT, r1 = INIT, r2 = INIT, r3 = INIT, r4 = INIT, r5 = INIT, r6 = INIT, r7 = INIT; \
    const unsigned long long t0 = __builtin_readcyclecounter();                                   \
    for (int it = 0; it < iters; ++it) {                                                          \
      asm volatile(ASM : "+" CONSTR(r0)); asm volatile(ASM : "+" CONSTR(r1)); asm volatile(ASM : "+" CONSTR(r2)); asm volatile(ASM : "+" CONSTR(r3)); \
      asm volatile(ASM : "+" CONSTR(r4)); asm volatile(ASM : "+" CONSTR(r5)); asm volatile(ASM : "+" CONSTR(r6)); asm volatile(ASM : "+" CONSTR(r7)); \
    }                                                                                             \
    const unsigned long long t1 = __builtin_readcyclecounter();                                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(*(float*)&r0) + (float)(*(float*)&r1) + (float)(*(float*)&r2) + (float)(*(float*)&r3) + \
        (float)(*(float*)&r4) + (float)(*(float*)&r5) + (float)(*(float*)&r6) + (float)(*(float*)&r7);                                           \
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;                                  \
  }
typedef float f2 __attribute__((ext_vector_type(2)));
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %0, %0", "v", float, 0.5f)
KERNEL(k_mul_f32, "v_mul_f32 %0, %0, %0", "v", float, 0.999f)
KERNEL(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %0, %0", "v", double, 0.5)
KERNEL(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %0", "v", double, 0.999)
KERNEL(k_pk_fma_f16, "v_pk_fma_f16 %0, %0, %0, %0", "v", float, 0.5f)
KERNEL(k_pk_mul_f16, "v_pk_mul_f16 %0, %0, %0", "v", float, 0.5f)
KERNEL(k_pk_add_f16, "v_pk_add_f16 %0, %0, %0", "v", float, 0.5f)
KERNEL(k_pk_max_f16, "v_pk_max_f16 %0, %0, %0", "v", float, 0.5f)
KERNEL(k_exp_f32, "v_exp_f32 %0, %0", "v", float, 0.5f)
KERNEL(k_rcp_f32, "v_rcp_f32 %0, %0", "v", float, 1.5f)
KERNEL(k_exp_f16, "v_exp_f16 %0, %0", "v", float, 0.5f)
KERNEL(k_rcp_f16, "v_rcp_f16 %0, %0", "v", float, 1.5f)
KERNEL(k_cvt_pk_bf16_f32, "v_cvt_pk_bf16_f32 %0, %0, %0", "v", float, 0.5f)
KERNEL(k_cvt_pkrtz_f16_f32, "v_cvt_pkrtz_f16_f32 %0, %0, %0", "v", float, 0.5f)
KERNEL(k_cvt_f32_f16, "v_cvt_f32_f16 %0, %0", "v", float, 0.5f)
KERNEL(k_lshlrev_b32, "v_lshlrev_b32 %0, 16, %0", "v", float, 0.5f)
KERNEL(k_and_b32, "v_and_b32 %0, 0xffff0000, %0", "v", float, 0.5f)
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %0, vcc", "v", float, 0.5f)

template <typename K> void run(const char* name, K kern, int elems) {
  const int iters = 20000;
  float* out; unsigned long long* ticks;
  (void)hipMalloc(&out, sizeof(float) * 256 * 256); (void)hipMalloc(&ticks, 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  kern<<<256, 256>>>(out, ticks, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  kern<<<256, 256>>>(out, ticks, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long t; (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 8;
  printf("%-22s %6.2f cycles per wave-instruction  (%d element%s per lane: %5.2f cycles per element; wall %.2f ns per instruction)\n", name, t / n, elems,
         elems > 1 ? "s" : " ", t / n / elems, ms * 1e6 / n);
  (void)hipFree(out); (void)hipFree(ticks);
}
int main() {
  run("v_fma_f32", k_fma_f32, 1); run("v_mul_f32", k_mul_f32, 1); run("v_pk_fma_f32", k_pk_fma_f32, 2); run("v_pk_mul_f32", k_pk_mul_f32, 2);
  run("v_pk_fma_f16", k_pk_fma_f16, 2); run("v_pk_mul_f16", k_pk_mul_f16, 2); run("v_pk_add_f16", k_pk_add_f16, 2); run("v_pk_max_f16", k_pk_max_f16, 2);
  run("v_exp_f32", k_exp_f32, 1); run("v_rcp_f32", k_rcp_f32, 1); run("v_exp_f16", k_exp_f16, 1); run("v_rcp_f16", k_rcp_f16, 1);
  run("v_cvt_pk_bf16_f32", k_cvt_pk_bf16_f32, 2); run("v_cvt_pkrtz_f16_f32", k_cvt_pkrtz_f16_f32, 2); run("v_cvt_f32_f16", k_cvt_f32_f16, 1);
  run("v_lshlrev_b32", k_lshlrev_b32, 1); run("v_and_b32", k_and_b32, 1); run("v_cndmask_b32", k_cndmask, 1);
  return 0;
}
