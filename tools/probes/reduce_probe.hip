// Where does a grouped column reduction over [G][R][C] bf16 lose against a plain read?  (developer probe;
// hipcc --offload-arch=gfx950 -O3 reduce_probe.hip -o reduce_probe && ./reduce_probe)
// Variants: slots in flight, passes per block (grid size), contiguous or strided row ownership, nontemporal loads,
// epilogue on/off (LDS reduce + fp64 atomics), per-element work none / silu.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
__device__ inline float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
struct Cfg { int G, R, C, passes, contiguous, epilogue, work; };
template <int SLOTS, bool NT>
__global__ __launch_bounds__(256) void red(const unsigned short* y, double* out, Cfg c) {
  __shared__ float lds[3 * 1152 > 256 * 8 ? 3 * 1152 : 256 * 8];
  const int cpr = c.C / 8, rpb = 256 / cpr, chunk = threadIdx.x % cpr, rsub = threadIdx.x / cpr;
  const bool valid = rsub < rpb;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (valid) {
    const unsigned short* base = y + (long)blockIdx.y * c.R * c.C + chunk * 8;
    long r0, stride, rend;
    if (c.contiguous) { const long per = (long)rpb * c.passes; r0 = blockIdx.x * per + rsub; stride = rpb; rend = (blockIdx.x + 1) * per < c.R ? (blockIdx.x + 1) * per : c.R; }
    else { r0 = (long)blockIdx.x * rpb + rsub; stride = (long)gridDim.x * rpb; rend = c.R; }
    u16x8 raw[SLOTS];
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) { const long rr = r0 + k * stride < rend ? r0 + k * stride : 0; const u16x8* p = (const u16x8*)(base + rr * c.C); raw[k] = NT ? __builtin_nontemporal_load(p) : *p; }
    for (long r = r0; r < rend; r += SLOTS * stride) {
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) {
        const long rr = r + k * stride, rn = rr + SLOTS * stride < rend ? rr + SLOTS * stride : 0;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bf2f(raw[k][j]);
        const u16x8* p = (const u16x8*)(base + rn * c.C);
        raw[k] = NT ? __builtin_nontemporal_load(p) : *p;
        if (rr < rend) {
          if (c.work) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float z = v[j] * 1.1f + 0.1f; v[j] = z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * z)); }
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
      }
    }
  }
  if (c.epilogue >= 2) {
    // coalesced forms: the block's partial rows go to LDS, thread t owns CHANNEL t (consecutive lanes = consecutive addresses)
    __syncthreads();
    if (valid) for (int j = 0; j < 8; ++j) lds[rsub * c.C + chunk * 8 + j] = acc[j];      // [rpb][C]
    __syncthreads();
    for (int ch = threadIdx.x; ch < c.C; ch += 256) {
      float s = 0.f;
      for (int r = 0; r < rpb; ++r) s += lds[r * c.C + ch];
      if (c.epilogue == 2) atomicAdd(out + (long)blockIdx.y * c.C + ch, (double)s);
      else if (c.epilogue == 3) atomicAdd((float*)out + (long)blockIdx.y * c.C + ch, s);
      else ((float*)out)[((long)blockIdx.y * gridDim.x + blockIdx.x) * c.C + ch] = s;        // 4: plain partial store
    }
  } else if (c.epilogue) {
    __syncthreads();
    if (valid) for (int j = 0; j < 8; ++j) lds[(rsub * cpr + chunk) * 8 + j] = acc[j];
    __syncthreads();
    if (valid && rsub == 0) {
      for (int r = 1; r < rpb; ++r) for (int j = 0; j < 8; ++j) acc[j] += lds[(r * cpr + chunk) * 8 + j];
      for (int j = 0; j < 8; ++j) atomicAdd(out + (long)blockIdx.y * c.C + chunk * 8 + j, (double)acc[j]);
    }
  } else if (acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7] == 1.2345f) out[0] = 1;
}
template <int SLOTS, bool NT>
void run(const unsigned short* y, double* out, Cfg c) {
  const int cpr = c.C / 8, rpb = 256 / cpr;
  const long per = (long)rpb * c.passes;
  const int gb = (int)((c.R + per - 1) / per);
  dim3 grid(gb, c.G);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) red<SLOTS, NT><<<grid, 256>>>(y, out, c);
  (void)hipEventRecord(e0);
  const int reps = 20;
  for (int r = 0; r < reps; ++r) red<SLOTS, NT><<<grid, 256>>>(y, out, c);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 2.0 * c.G * c.R * c.C;
  printf("%dx%dx%d slots=%d nt=%d passes=%3d blocks=%5d contig=%d epi=%d work=%d  %7.1f us %6.0f GB/s\n", c.G, c.R, c.C, SLOTS, (int)NT, c.passes,
         gb * c.G, c.contiguous, c.epilogue, c.work, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
}
int main() {
  // eight different tensors in rotation would defeat the 256 MB last-level cache; here ONE tensor per shape, as kbench does -
  // and a 1 GB one ("big") for the cold number
  for (int shape = 0; shape < 3; ++shape) {
    const int G = shape == 2 ? 200 : 20, R = shape == 1 ? 920 : 3680, C = shape == 1 ? 1152 : 672;
    unsigned short* y; double* out;
    (void)hipMalloc(&y, 2L * G * R * C); (void)hipMalloc(&out, 8L * G * C * 512);
    (void)hipMemset(y, 0x3c, 2L * G * R * C); (void)hipMemset(out, 0, 8L * G * C);
    for (int work = 0; work < 2; ++work)
      for (int epi = 0; epi < 5; ++epi)
        for (int contig = 0; contig < 1; ++contig)
          for (int passes : {8, 16, 32, 64}) {
            Cfg c = {G, R, C, passes, contig, epi, work};
            run<4, false>(y, out, c);
            if (passes == 32) { run<4, true>(y, out, c); }
          }
    (void)hipFree(y); (void)hipFree(out);
  }
  return 0;
}
