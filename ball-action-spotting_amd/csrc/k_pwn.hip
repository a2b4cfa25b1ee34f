// k_pwn.hip — mds_pw_fwd for the N-STREAMING 1x1 GEMMs (bf16): short K (96 / 128 / 192), wide N - the MBConv / 3D expansions
// (cin -> mid, no prologue, statistics) and the data gradients of the projections (dy3[M][cout] * W -> u2[M][mid]) at <= 40 k rows.
//
// At 18 400 rows these launches are 50 MB problems that the general kernel (k_pw.hip) runs as 2592 blocks of 64 x 128 outputs,
// each re-staging its 24 KB of x and 48 KB of filter through registers and LDS (187 MB of L2 -> CU traffic for a 50 MB problem) and
// each ending in its own statistics flush: 25 - 29 us, 1.7 - 2.0 TB/s.  Here a block owns BM rows and ALL of N:
//   * the block's x rows live in REGISTERS for the whole kernel, as the MFMA B fragments of each wave (every wave holds all BM
//     rows x K: loaded once, straight from global memory, 16 bytes per lane);
//   * N is walked in 64-column stages, wave w takes the 16 columns 16 w of a stage: its filter fragments come straight from the
//     fragment-major copy (MDS_PACK_FRAG_*: 1 KiB per fragment, lane-linear, L2 resident) one stage ahead into a second register
//     set - nothing but the output tile ever touches LDS;
//   * the stage's BM x 64 outputs leave through a double-buffered LDS tile as 16-byte row segments (128 contiguous bytes per row),
//     ONE barrier per stage; the column sums of a wave's 16 columns are reduced over its lanes and parked in LDS - no atomics
//     inside the loop, one coalesced fp64 atomic per channel at the end (the flush form of elem.h).
// Bound: the output write (10 KB per stage and block; 1300 cycles per stage at 8 B/clk/CU against 480 cycles of MFMA per wave).
#include <stdlib.h>
#include "gemm.h"

namespace {
// MFW: 16-row fragments per block (BM = 16 MFW); KS: 32-channel k-steps (K = 32 KS, or 32 KS - 16 with the last half step zero
// padded by the fragment copy and masked on the x side); STATS: forward statistics
template <int MFW, int KS, bool STATS>
__global__ __launch_bounds__(256, MFW <= 3 ? 4 : 2) void pwn_kernel(mds_pw_fwd_args a) {
  MDS_CHAIN_PRIO();
  constexpr int BM = 16 * MFW;
  constexpr int TP = 64 + 8;                             // staged row pitch in elements: 144 B = 16 B x odd
  __shared__ __attribute__((aligned(16))) bf16_t tile[2][BM * TP];
  MDS_DYN_SMEM(smem);                                    // [2][N] floats: column sums / sums of squares of this block's rows
  float* colsum = (float*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const long m0 = (long)blockIdx.x * BM;
  const int K = a.K, N = a.N, NFT = N >> 4, NST = N >> 6;
  const bf16_t* x = (const bf16_t*)a.x;
  // ---- this wave's copy of the block's rows: B fragments, row 16 mf + i, channels 32 ks + 8 q .. + 7
  u16x8 xf[MFW][KS];
#pragma unroll
  for (int mf = 0; mf < MFW; ++mf) {
    long row = m0 + 16 * mf + i;
    if (row >= a.M) row = a.M - 1;                       // rows past M: finite values, never stored, masked in the sums
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = 32 * ks + 8 * q;
      if (k < K) xf[mf][ks] = *(const u16x8*)(x + row * K + k);
      else xf[mf][ks] = (u16x8){0, 0, 0, 0, 0, 0, 0, 0};  // (K = 32 KS - 16: the fragment copy's half step is zero there too)
    }
  }
  // ---- filter fragments of (k-step ks, column fragment nf): 1 KiB at ((ks * NFT + nf) * 512 + lane * 8) elements.  Requested TWO
  // stages ahead into a three-set register ring through loads hipcc does not see (gld16: it would wait for vmcnt(0) - i.e. also
  // for the previous stage's global stores - at the first use after the loop's back edge: 3000 cycles per stage); the wait is
  // counted by hand.  vmcnt(2 KS) = "all but the 2 KS most recent vector-memory operations are done": the fragments of stage st
  // were followed by at least the 2 KS loads of stages st + 1 and st + 2 (and by stage st - 1's stores: the count only errs on
  // the strict side).
  const bf16_t* wbase = (const bf16_t*)a.w_frag + lane * 8;
  u16x8 wf[3][KS];
  auto load_w = [&](int st, u16x8 (&dst)[KS]) {          // always KS loads (past the end: the last stage again), so that the count holds
    const int nf = 4 * (st < NST ? st : NST - 1) + wave;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) gld16(dst[ks], wbase + ((long)ks * NFT + nf) * 512);
  };
  bf16_t* y = (bf16_t*)a.y;
  // rows of this lane that exist: bit mf set when row m0 + 16 mf + i < M
  unsigned rok = 0;
#pragma unroll
  for (int mf = 0; mf < MFW; ++mf) rok |= (m0 + 16 * mf + i < a.M ? 1u : 0u) << mf;

  auto stage = [&](int st, u16x8 (&w)[KS], u16x8 (&w2)[KS]) {
    load_w(st + 2, w2);                                  // into the set stage st - 1 has left
    wait_vm<2 * KS>();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) reg_pin(w[ks]);
    f32x4 acc[MFW];
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf) acc[mf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int mf = 0; mf < MFW; ++mf) mma16(w[ks], xf[mf][ks], acc[mf]);   // acc[r] = y[row 16 mf + i][col 16 wave + 4 q + r]
    bf16_t* tl = tile[st & 1];
    float s[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf) {
      const float v[4] = {acc[mf][0], acc[mf][1], acc[mf][2], acc[mf][3]};
      store4(tl + (16 * mf + i) * TP + 16 * wave + 4 * q, v);
      if (STATS && ((rok >> mf) & 1u)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[r] += v[r]; ss[r] += v[r] * v[r]; }
      }
    }
    if (STATS) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t0 = sum_over_i16(s[r]), t1 = sum_over_i16(ss[r]);
        if (i == 0) {                                    // every column of a stage is owned by exactly one wave: plain stores
          const int c = 64 * st + 16 * wave + 4 * q + r;
          colsum[c] = t0; colsum[N + c] = t1;
        }
      }
    }
    // the tile is complete; (st - 1)'s readers are past their LDS loads.  The bare barrier behind an LDS-only wait:
    // __syncthreads() also drains vmcnt, i.e. it would wait for the filter ring and for the previous stage's global stores
    wait_lgkm0();
    raw_barrier();
    for (int e = tid; e < BM * 8; e += 256) {            // 16-byte row segments: 8 per row = 128 contiguous bytes
      const int r = e >> 3, sg = e & 7;
      const long m = m0 + r;
      if (m < a.M) *(u16x8*)(y + m * N + 64 * st + 8 * sg) = *(const u16x8*)(tl + r * TP + 8 * sg);
    }
  };
  load_w(0, wf[0]);
  load_w(1, wf[1]);
  for (int st = 0; st < NST; st += 3) {
    stage(st, wf[0], wf[2]);
    if (st + 1 < NST) stage(st + 1, wf[1], wf[0]);
    if (st + 2 < NST) stage(st + 2, wf[2], wf[1]);
  }
  wait_vm<0>();
  if (STATS) {
    __syncthreads();
    double* sl = a.stats + (long)(blockIdx.x % MDS_STAT_SLOTS) * 2 * N;
    for (int e = tid; e < 2 * N; e += 256) atomicAdd(sl + e, (double)colsum[e]);   // thread = channel: coalesced fp64 atomics
  }
}

bool pwn_shape_ok(long M, int K, int N, int dtype) {
  return dtype == MDS_BF16 && (K == 96 || K == 192 || K == 128) && N % 64 == 0 && N >= 256 && N > K && M < 4294967295L;
}
}  // namespace

// does a launch of this shape take the N-streaming kernel when it is given the fragment-major filter copy?
int pwn_wants_frag(long M, int K, int N, int dtype, int data_gradient) {
  const int knob = mds_knob(MDS_KNOB_PWN);
  if (knob == 1 || !pwn_shape_ok(M, K, N, dtype)) return 0;
  if (knob == 2) return 1;
  if ((knob == 0 && data_gradient) || (knob == 4 && !data_gradient)) return 0;   // rule: forward launches (see k_pwk8.hip); 3: both
  return M >= 4096 && M <= 40000;
}

// 1 = not taken (another kernel runs), 0 = launched, < 0 = error
int pw_fwd_n_try(const mds_pw_fwd_args* a, mds_stream_t stream) {
  if (!a->w_frag || a->epi.mode != MDS_EPI_NONE || a->split > 1 || a->pro.mode != MDS_PRO_NONE || a->residual ||
      a->post.mode != MDS_POST_NONE)
    return 1;
  const bool dg = a->stats == nullptr;
  if (!pwn_wants_frag(a->M, a->K, a->N, a->dtype, dg)) return 1;
  const int KS = (a->K + 31) / 32;
  // rows per tile: one round of blocks where possible (18 400 rows: 80-row tiles, 230 blocks); 64-row tiles otherwise
  const int cus = 256;
  const int BM = (cdiv(a->M, 80) <= 2 * cus && cdiv(cdiv(a->M, 80), cus) * 80 <= cdiv(cdiv(a->M, 64), cus) * 64) ? 80 : 64;
  const int kb = mds_knob(MDS_KNOB_PWK_BM);
  const int BMk = (kb == 32 || kb == 48 || kb == 64 || kb == 80) ? kb : BM;      // (A/B and tests: the tile-rows knob of the K-streaming kernel applies here too)
  const dim3 grid(cdiv(a->M, BMk)), block(256);
  const size_t smem = (size_t)2 * a->N * sizeof(float);
#define PWN_GO(MFW, KS_, ST) MDS_LAUNCH((pwn_kernel<MFW, KS_, ST>), grid, block, smem, stream, *a)
#define PWN_K(MFW, ST) do { if (KS == 3) PWN_GO(MFW, 3, ST); else if (KS == 4) PWN_GO(MFW, 4, ST); else PWN_GO(MFW, 6, ST); } while (0)
  if (BMk == 32) { if (a->stats) PWN_K(2, true); else PWN_K(2, false); }
  else if (BMk == 48) { if (a->stats) PWN_K(3, true); else PWN_K(3, false); }
  else if (BMk == 80) { if (a->stats) PWN_K(5, true); else PWN_K(5, false); }
  else { if (a->stats) PWN_K(4, true); else PWN_K(4, false); }
#undef PWN_K
#undef PWN_GO
  return mds_check_launch("pw_fwd (N-streaming)");
}
