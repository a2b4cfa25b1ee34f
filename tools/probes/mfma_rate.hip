// Probe: issue interval of v_mfma_f32_16x16x32_bf16 from ONE wave per SIMD vs two, independent accumulators, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o tools/probes/build/mfma_rate && tools/probes/build/mfma_rate
// Prints s_memtime ticks and wall-clock ns per MFMA for 1 / 2 / 4 waves per SIMD, with 0 / 1 / 2 / 3 independent VALU ops after each MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV>
__global__ void k(float* out, unsigned long long* ticks, int iters) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      if (NV >= 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v0));
      if (NV >= 2) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v1));
      if (NV >= 3) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v2));
      if (NV >= 4) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(double*)&acc[7]));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = v0 + v1 + v2;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int NV> void run(int waves_per_simd) {
  const int iters = 20000, blocks = 256, threads = 256 * waves_per_simd;
  float* out; unsigned long long* ticks;
  hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&ticks, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NV><<<blocks, threads>>>(out, ticks, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NV><<<blocks, threads>>>(out, ticks, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 8;
  const double tf = 256.0 * 4 * waves_per_simd * n * 16384 / (ms * 1e-3) / 1e12;
  printf("waves/SIMD %d, %d VALU per MFMA: %.2f ticks per MFMA per wave, %.2f ns per MFMA per wave, tick = %.3f ns (%.2f GHz), chip %.0f TFLOP/s\n",
         waves_per_simd, NV, t / n, ms * 1e6 / n, ms * 1e6 / t, t / (ms * 1e6), tf);
  hipFree(out); hipFree(ticks);
}
int main() {
  for (int w : {1, 2, 4}) { run<0>(w); run<1>(w); run<2>(w); run<3>(w); run<4>(w); }
  return 0;
}
