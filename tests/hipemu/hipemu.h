// hipemu.h — TEST INFRASTRUCTURE ONLY.
//
// A small host-side simulator of the HIP execution-model subset that csrc/*.hip uses, so that
// the *same kernel sources* can be stepped through on a CPU-only CI box (index math, LDS tiling,
// barrier placement, MFMA fragment bookkeeping).  It is force-included (-include) when building
// libmds_emu.so (tests/hipemu first on its include path) and is never part of the product library (libmds_hip.so), which
// is hipcc/gfx950 only.  Nothing here is a fallback: the Python product path cannot load it.
//
// Model: one OS thread; every HIP thread of a block is a fiber (own stack, hand-rolled context switch); blocks run one after
// another.  __syncthreads() and the wave-level operations (shuffles, MFMA) are rendezvous
// points: a fiber that reaches one yields to the scheduler until all live fibers of the block
// (or of its 64-lane wave) have arrived.  A round in which no fiber can make progress is a
// divergent-barrier deadlock and aborts with a message.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0

inline dim3 threadIdx, blockIdx, blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

namespace hipemu {

enum { RUNNING = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3, READY = 4 };

// Minimal x86-64 fiber switch (callee-saved registers + stack pointer).  ucontext's swapcontext
// issues two sigprocmask syscalls per switch, which dominated the simulator's run time.
typedef void* fctx_t;
__attribute__((naked, noinline)) static void fiber_switch(fctx_t* from, fctx_t* to) {
  asm volatile(
      "pushq %rbp\n pushq %rbx\n pushq %r12\n pushq %r13\n pushq %r14\n pushq %r15\n"
      "movq %rsp, (%rdi)\n"
      "movq (%rsi), %rsp\n"
      "popq %r15\n popq %r14\n popq %r13\n popq %r12\n popq %rbx\n popq %rbp\n"
      "ret\n");
}
inline fctx_t fiber_make(char* stack, size_t size, void (*entry)()) {
  uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
  void** sp = (void**)top;
  *--sp = nullptr;        // fake return address of `entry` (it never returns)
  *--sp = (void*)entry;   // popped by fiber_switch's ret
  for (int i = 0; i < 6; ++i) *--sp = nullptr;  // rbp rbx r12 r13 r14 r15
  return (fctx_t)sp;
}

struct Fiber {
  fctx_t ctx = nullptr;
  char* stack = nullptr;
  int state = READY;
  unsigned tid = 0;
  uint64_t wait_gen = 0;
};

struct BlockState {
  std::vector<Fiber> fibers;
  int n = 0, nwaves = 0;
  int cur = 0;
  fctx_t sched = nullptr;
  int block_arrived = 0, done = 0;
  uint64_t block_gen = 1;
  std::vector<int> wave_arrived, wave_done;
  std::vector<uint64_t> wave_gen;
  std::function<void()> body;
  // rendezvous scratch for wave ops: [wave][lane][16 x 8 bytes]
  std::vector<uint64_t> scratch;
};

inline BlockState* g_blk = nullptr;
inline std::vector<char> g_dyn_smem;
inline char* dyn_smem() { return g_dyn_smem.data(); }

static const size_t kStack = 256 * 1024;

inline void set_tid(unsigned tid) {
  threadIdx.x = tid % blockDim.x;
  threadIdx.y = (tid / blockDim.x) % blockDim.y;
  threadIdx.z = tid / (blockDim.x * blockDim.y);
}

inline void yield_to_sched() {
  BlockState* b = g_blk;
  Fiber& f = b->fibers[b->cur];
  fiber_switch(&f.ctx, &b->sched);
}

inline int lane_id() { return g_blk->cur & 63; }
inline int wave_id() { return g_blk->cur >> 6; }
inline int wave_live(BlockState* b, int w) {
  int sz = (w == b->nwaves - 1) ? (b->n - 64 * w) : 64;
  return sz - b->wave_done[w];
}

inline void block_barrier() {
  BlockState* b = g_blk;
  Fiber& f = b->fibers[b->cur];
  b->block_arrived++;
  if (b->block_arrived == b->n - b->done) {
    b->block_arrived = 0;
    b->block_gen++;
    return;
  }
  f.wait_gen = b->block_gen;
  f.state = WAIT_BLOCK;
  yield_to_sched();
}

inline void wave_barrier() {
  BlockState* b = g_blk;
  int w = wave_id();
  Fiber& f = b->fibers[b->cur];
  b->wave_arrived[w]++;
  if (b->wave_arrived[w] == wave_live(b, w)) {
    b->wave_arrived[w] = 0;
    b->wave_gen[w]++;
    return;
  }
  f.wait_gen = b->wave_gen[w];
  f.state = WAIT_WAVE;
  yield_to_sched();
}

inline uint64_t* wave_scratch(int lane) {
  BlockState* b = g_blk;
  return &b->scratch[((size_t)wave_id() * 64 + lane) * 16];
}

inline void trampoline() {
  BlockState* b = g_blk;
  b->body();
  Fiber& f = b->fibers[b->cur];
  f.state = DONE;
  b->done++;
  b->wave_done[b->cur >> 6]++;
  fiber_switch(&f.ctx, &b->sched);
  abort();  // a finished fiber is never resumed
}

inline void run_block(BlockState& b) {
  g_blk = &b;
  b.block_arrived = 0; b.done = 0; b.block_gen = 1;
  std::fill(b.wave_arrived.begin(), b.wave_arrived.end(), 0);
  std::fill(b.wave_done.begin(), b.wave_done.end(), 0);
  std::fill(b.wave_gen.begin(), b.wave_gen.end(), 1);
  for (int i = 0; i < b.n; ++i) {
    Fiber& f = b.fibers[i];
    f.state = READY; f.tid = i; f.wait_gen = 0;
    f.ctx = fiber_make(f.stack, kStack, trampoline);
  }
  while (b.done < b.n) {
    bool progressed = false;
    for (int i = 0; i < b.n; ++i) {
      Fiber& f = b.fibers[i];
      if (f.state == DONE) continue;
      if (f.state == WAIT_BLOCK) {
        // released either by a generation bump or because everyone still alive has arrived
        if (b.block_gen == f.wait_gen) {
          if (b.block_arrived == b.n - b.done && b.block_arrived > 0) { b.block_arrived = 0; b.block_gen++; }
          else continue;
        }
      } else if (f.state == WAIT_WAVE) {
        int w = i >> 6;
        if (b.wave_gen[w] == f.wait_gen) {
          if (b.wave_arrived[w] == wave_live(&b, w) && b.wave_arrived[w] > 0) { b.wave_arrived[w] = 0; b.wave_gen[w]++; }
          else continue;
        }
      }
      b.cur = i;
      set_tid(i);
      f.state = RUNNING;
      fiber_switch(&b.sched, &f.ctx);
      progressed = true;
    }
    if (!progressed) {
      fprintf(stderr, "hipemu: deadlock (divergent barrier?) block=(%u,%u,%u) done=%d/%d\n",
              blockIdx.x, blockIdx.y, blockIdx.z, b.done, b.n);
      abort();
    }
  }
  g_blk = nullptr;
}

inline void launch(dim3 grid, dim3 block, size_t smem, std::function<void()> body) {
  static BlockState b;
  int n = block.x * block.y * block.z;
  if ((int)b.fibers.size() < n) {
    size_t old = b.fibers.size();
    b.fibers.resize(n);
    for (size_t i = old; i < (size_t)n; ++i) b.fibers[i].stack = (char*)malloc(kStack);
  }
  b.n = n;
  b.nwaves = (n + 63) / 64;
  b.wave_arrived.assign(b.nwaves, 0);
  b.wave_done.assign(b.nwaves, 0);
  b.wave_gen.assign(b.nwaves, 1);
  b.scratch.assign((size_t)b.nwaves * 64 * 16, 0);
  b.body = body;
  if (g_dyn_smem.size() < smem + 64) g_dyn_smem.resize(smem + 64);
  blockDim = block; gridDim = grid;
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        blockIdx = dim3(x, y, z);
        run_block(b);
      }
}

template <typename T>
inline T shfl_src(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "shfl payload");
  uint64_t raw = 0;
  memcpy(&raw, &v, sizeof(T));
  wave_scratch(lane_id())[0] = raw;
  wave_barrier();
  BlockState* b = g_blk;
  int w = wave_id();
  int sz = (w == b->nwaves - 1) ? (b->n - 64 * w) : 64;
  T out = v;
  if (src_lane >= 0 && src_lane < sz) {
    uint64_t r = wave_scratch(src_lane)[0];
    memcpy(&out, &r, sizeof(T));
  }
  wave_barrier();
  return out;
}

}  // namespace hipemu

inline void __syncthreads() { hipemu::block_barrier(); }
template <typename T> inline T __shfl_xor(T v, int mask, int width = 64) { (void)width; return hipemu::shfl_src(v, hipemu::lane_id() ^ mask); }
template <typename T> inline T __shfl_down(T v, unsigned d, int width = 64) { (void)width; return hipemu::shfl_src(v, hipemu::lane_id() + (int)d); }
template <typename T> inline T __shfl(T v, int src, int width = 64) { (void)width; return hipemu::shfl_src(v, src); }

inline void __threadfence() {}
inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
inline int atomicExch(int* p, int v) { int o = *p; *p = v; return o; }
inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }

inline const char* hipGetErrorString(int) { return "hipemu"; }
inline int hipGetLastError() { return 0; }
inline int hipemu_cu_count() { return 256; }      // the simulated device: an MI355X's 256 CUs
inline int hipPeekAtLastError() { return 0; }
