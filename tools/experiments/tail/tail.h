// tail.h — "the launch that produced a BatchNorm layer's statistics also finalizes the layer" (mds_tail_t).
//
// 144 finalize launches per step used to sit between dependent kernels: 5-6 us each of launch + one memory round trip,
// with the chip idle.  With a tail, the producing launch carries ONE extra workgroup (the last linear block id, so it
// is dispatched last) that waits until every wave of the working blocks has ARRIVED, then runs the finalize itself.
//
// Protocol (MI355X_MICROARCH.md, inter-workgroup visibility): the payload are fp32 atomics (performed at agent scope,
// never cached in a per-XCD L2); every wave drains its own (`s_waitcnt vmcnt(0)`), the block meets at a barrier and one
// thread adds 1 to one of the ticket counters - no returned value, the block simply ends.  The poller polls the ticket with relaxed agent-scope
// loads, issues ONE agent-scope acquire, `__syncthreads()`, and reads the slots.  Working blocks never wait on
// anything, so there is no forward-progress assumption beyond "the poller eventually gets a slot"; its spin is bounded
// (a protocol bug costs wrong statistics that the parity tests catch, never a hung GPU).  The poller re-arms the ticket.
#pragma once
#include "platform.h"

// Arrivals are spread over MDS_TAIL_WORDS counters, one 64-byte line each: a few thousand agent-scope atomics on ONE
// address serialise at ~7 ns apiece (measured: +76 us per launch with one counter and an arrival per wave).
#define MDS_TAIL_WORDS 32
#define MDS_TAIL_STRIDE 16   // ints between counters

// number of arrivals the poller waits for: ONE per working block (the grid has one extra column of blocks)
MDS_DEV unsigned mds_tail_expected() { return (gridDim.x - 1) * gridDim.y * gridDim.z; }
// true for the blocks of the extra column; exactly one of them (the last linear id) is the poller
MDS_DEV bool mds_tail_extra_block(const mds_tail_t& t) { return t.ticket != 0 && blockIdx.x == gridDim.x - 1; }
MDS_DEV bool mds_tail_is_poller() { return blockIdx.y == gridDim.y - 1 && blockIdx.z == gridDim.z - 1; }

// every thread of a working block, once, after the block's last statistics atomic (no early returns before it)
MDS_DEV void mds_tail_arrive(const mds_tail_t& t) {
  if (!t.ticket) return;
#ifndef MDS_EMU
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's atomics are performed
#endif
  __syncthreads();                                    // ... and every other wave's of the block
  if (threadIdx.x == 0) {
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    int* w = t.ticket + (lin % MDS_TAIL_WORDS) * MDS_TAIL_STRIDE;
#ifndef MDS_EMU
    __hip_atomic_fetch_add(w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *w += 1;
#endif
  }
}

// the poller block: wait for all arrivals, make them visible to plain loads of this CU
MDS_DEV void mds_tail_wait(const mds_tail_t& t, unsigned expected) {
#ifndef MDS_EMU
  if (threadIdx.x < 64) {   // one wave polls: lane k reads counter k, the wave sums
    int spins = 0;
    for (;;) {
      int v = threadIdx.x < MDS_TAIL_WORDS ? __hip_atomic_load(t.ticket + threadIdx.x * MDS_TAIL_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
      if ((unsigned)v >= expected || ++spins >= (1 << 20)) break;
      __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
#else
  (void)expected;
#endif
  __syncthreads();
}
MDS_DEV void mds_tail_rearm(const mds_tail_t& t) {
  __syncthreads();
  if (threadIdx.x < MDS_TAIL_WORDS) {
#ifndef MDS_EMU
    __hip_atomic_store(t.ticket + threadIdx.x * MDS_TAIL_STRIDE, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    t.ticket[threadIdx.x * MDS_TAIL_STRIDE] = 0;
#endif
  }
}

// mds_bn_finalize for every channel by ONE 256-thread block.  Producers with a tail add their statistics to the first
// MDS_TAIL_SLOTS slots only (the others stay zero for the stand-alone kernel), so a thread's loads for three of its
// channels (c, c + 256, c + 512) fit in ~60 registers and go out together: one memory round trip per 768 channels
// (the tail must not raise the register allocation of the kernel that hosts it).
#define MDS_TAIL_SLOTS 8
#define MDS_TAIL_MAXC 4096
MDS_DEV void bn_finalize_all(const mds_bn_finalize_args& a) {
  constexpr int NCH = 3;
#pragma unroll 1
  for (int cb = 0; cb < a.C; cb += 256 * NCH) {
  float gam[NCH], bet[NCH], rm[NCH], rv[NCH], v[NCH][MDS_TAIL_SLOTS], w[NCH][MDS_TAIL_SLOTS];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = cb + threadIdx.x + 256 * j;
    if (c < a.C) {
      gam[j] = a.gamma[c]; bet[j] = a.beta[c];
      rm[j] = a.running_mean ? a.running_mean[c] : 0.f;
      rv[j] = a.running_mean ? a.running_var[c] : 1.f;
      if (a.training) {
#pragma unroll
        for (int k = 0; k < MDS_TAIL_SLOTS; ++k) {
          v[j][k] = a.stats[(long)(k * 2 + 0) * a.C + c];
          w[j][k] = a.stats[(long)(k * 2 + 1) * a.C + c];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = cb + threadIdx.x + 256 * j;
    if (c < a.C) {
      float mean, var;
      if (a.training) {
        double s = 0.0, ss = 0.0;
#pragma unroll
        for (int k = 0; k < MDS_TAIL_SLOTS; ++k) { s += v[j][k]; ss += w[j][k]; }
        const double inv = 1.0 / (double)a.count;
        const double m = s * inv;
        double vv = ss * inv - m * m;
        if (vv < 0.0) vv = 0.0;
        mean = (float)m;
        var = (float)vv;
        if (a.running_mean) {
          const float unb = a.count > 1 ? (float)(vv * (double)a.count / (double)(a.count - 1)) : var;
          a.running_mean[c] = (1.0f - a.momentum) * rm[j] + a.momentum * mean;
          a.running_var[c] = (1.0f - a.momentum) * rv[j] + a.momentum * unb;
        }
      } else {
        mean = rm[j];
        var = rv[j];
      }
      const float rstd = 1.0f / sqrtf(var + a.eps);
      const float sc = gam[j] * rstd;
      a.out[0 * a.C + c] = sc;
      a.out[1 * a.C + c] = bet[j] - mean * sc;
      a.out[2 * a.C + c] = mean;
      a.out[3 * a.C + c] = rstd;
    }
  }
  }
  if (threadIdx.x == 0 && a.training && a.num_batches_tracked) *a.num_batches_tracked += 1;
}

// body of the extra column of blocks; returns after the finalize (callers `return` right after)
MDS_DEV void mds_tail_run(const mds_tail_t& t) {
  if (!mds_tail_is_poller()) return;
  mds_tail_wait(t, mds_tail_expected());
  if (t.fin) bn_finalize_all(*(const mds_bn_finalize_args*)t.fin);
  mds_tail_rearm(t);
}
