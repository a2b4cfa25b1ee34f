// k_pwk8.hip — mds_pw_fwd for the K-STREAMING 1x1 GEMMs (bf16), wave-specialised: K >= N, 64 < N <= 192 - the MBConv / 3D
// projections (mid -> cout, BN + SiLU + gate prologue) and the data gradients of the expansions (dy1[M][mid] * W -> dx[M][cin]).
//
// One block = 8 waves on one CU, two per SIMD, with DIFFERENT jobs, so that the phases of the GEMM overlap by construction
// instead of by instruction scheduling inside one wave (the four-wave form of this kernel, git 07b0ff9: every wave issued loads, then read
// fragments, then issued MFMAs, then ran the prologue's VALU work - nothing overlapped, profiles/r05_pwk_v3_trace.txt):
//   * waves 0-3, CONSUMERS: own the accumulators (BM rows x all of N, split WM x WN), read x fragments from the LDS ring
//     (ds_read_b128, conflict-free swizzle) one k-step ahead into a second register set, fetch the filter fragments of
//     their own columns straight from the fragment-major copy (MDS_PACK_FRAG_*: one coalesced 16-byte load per lane, L2
//     resident) three k-steps ahead into a four-set register ring, and issue MFMAs - with the next k-step's reads and
//     loads interleaved BETWEEN the MFMAs in program order.  Their vmcnt only ever counts filter loads.
//   * waves 4-7, PRODUCERS: stream x global -> LDS (global_load_lds_dwordx4, 1 KiB per instruction, no registers) DX
//     stages ahead - their vmcnt only ever counts these, so the waits are exact and HBM latency is covered by DX - 2
//     stages in flight - and, with a prologue, transform the raw stage IN PLACE two stages ahead of its use: producer w
//     owns k-octets 2w, 2w + 1 of a stage, lane r row r, so scale / shift / gate are wave-uniform (s_load, scalar cache).
//     Their VALU work shares a SIMD with a consumer's MFMAs: the two pipes run side by side.
//   * ONE s_barrier per 64-channel stage for all eight waves; the consumers take theirs in the middle of the stage (they
//     only need the barrier before touching the NEXT stage's slot), so they run half a stage behind the producers.
//   * LDS image of a stage (as in the four-wave form): row r at r * 128, k-octet o in 16-byte slot o ^ (r & 7) (on the SOURCE address
//     of the DMA); ring of DX + 1 slots.
//   * epilogue: accumulators -> LDS (fp32, the whole tile at once: the ring is free), then thread = (row group, 8-column
//     octet) over all 512 threads: 16-byte coalesced operand loads and stores, column sums row group -> LDS -> one
//     coalesced fp64 atomic per channel.
#include <stdlib.h>
#include <type_traits>
#include "gemm.h"

namespace {
template <int V> using ic = std::integral_constant<int, V>;

// MFW x WM row fragments, NFW x WN column fragments (WM * WN = 4 consumer waves); DX = prefetch distance of the x stream
// in 64-channel stages.  PRO: mds_pro_t mode.  TAIL: 0 = forward (statistics), 1 = data gradient (residual, mds_poststat_t).
template <int MFW, int WM, int NFW, int WN, int DX, int PRO, int TAIL>
__global__ __launch_bounds__(512) void pwk8_kernel(mds_pw_fwd_args a) {
  MDS_CHAIN_PRIO();
  static_assert(WM * WN == 4, "four consumer waves");
  constexpr bool XF = PRO != MDS_PRO_NONE;
  constexpr int NEED = XF ? 2 : 1;                       // stages ahead of the consumers' stage whose x must have LANDED at a barrier
  static_assert(DX >= NEED + 2, "at least one DMA batch stays in flight over a barrier");
  constexpr bool HAS_BN = PRO == MDS_PRO_AFFINE || PRO == MDS_PRO_BN_SILU || PRO == MDS_PRO_BN_SILU_GATE;
  constexpr bool HAS_ACT = PRO == MDS_PRO_BN_SILU || PRO == MDS_PRO_BN_SILU_GATE;
  constexpr bool HAS_GATE = PRO == MDS_PRO_BN_SILU_GATE || PRO == MDS_PRO_GATE;
  constexpr int BM = 16 * MFW * WM, BNP = 16 * NFW * WN;
  constexpr int NB = BM / 8;                             // 1 KiB DMA blocks (8 rows x 128 B) per stage
  constexpr int XS = BM * 128, RX = DX + 1;              // bytes per ring slot, ring length
  constexpr int DW = 3;                                  // filter prefetch distance in k-steps (four register sets)
  MDS_DYN_SMEM(smem);
  const lds_t xring = lds_addr_of(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);
  const long m0 = (long)blockIdx.x * BM;
  const int K = a.K, N = a.N, S = (K + 63) >> 6, KST = (K + 31) >> 5, NFT = (N + 15) >> 4;
  const bool ktail = (K & 63) != 0;                      // the last stage holds 32 channels (K % 64 == 32)
  f32x4 acc[MFW][NFW];
  const int NOCT = N >> 3, RG = 512 / NOCT;                // 8-column octets per row, row groups
  const int c = tid % NOCT, rg = tid / NOCT;
  constexpr int RGMIN = 512 / (BNP / 8), JMAX = (BM + RGMIN - 1) / RGMIN;
  bf16_t* y = (bf16_t*)a.y;
  constexpr bool DG = TAIL == 1;
  const bool post = DG && a.post.mode != MDS_POST_NONE;
  double* const sdst = DG ? (post ? a.post.stats : nullptr) : a.stats;
  float cs[8], css[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { cs[j] = 0.f; css[j] = 0.f; }
  float pb[4][8];                                          // DG: scale, shift, mean, rstd of this thread's octet
  u16x8 rres[JMAX], rys[JMAX];
  float rmk[JMAX];
  // DG: every epilogue operand of the thread's rows is requested HERE, before the K loop - the loads are older than anything the
  // loop counts on vmcnt (loads return in order), cost JMAX * 8 + 36 registers through the loop and take ~2 us of exposed
  // latency off the end of a kernel whose K loop is only ~12 us at 18 400 rows
  if (DG && rg < RG) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) pb[t][j] = post ? a.post.bn[(long)t * N + 8 * c + j] : 0.f;
#pragma unroll
    for (int jj = 0; jj < JMAX; ++jj) {
      const int rr = rg + RG * jj;
      long m = m0 + rr;
      if (!(rr < BM && m < a.M)) m = 0;
      rres[jj] = a.residual ? *(const u16x8*)((const bf16_t*)a.residual + m * N + 8 * c) : (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
      rys[jj] = post ? *(const u16x8*)((const bf16_t*)a.post.y + m * N + 8 * c) : (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
      rmk[jj] = (post && a.post.mode == MDS_POST_MASK) ? a.post.mask[(unsigned)m / (unsigned)a.post.rows_per_group] : 1.0f;
    }
  }
#ifdef PWK_TRACE   /* experiment builds: cycle stamps of block PWK_TRACE, [step][wave][phase] in LDS behind the ring, dumped through the (unused) split_part pointer */
  const bool trc = blockIdx.x == PWK_TRACE && a.split_part != nullptr;
  const lds_t trc_base = xring + RX * XS;
#define PWK_STAMP(s_, ph_) do { if (trc && lane == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); \
    asm volatile("ds_write_b64 %0, %1" ::"v"(trc_base + (((s_) * 8 + wave) * 8 + (ph_)) * 8), "v"(t_) : "memory"); } } while (0)
#else
#define PWK_STAMP(s_, ph_) ((void)0)
#endif

  if (wave >= 4) {
    // ================================================================ PRODUCER
    const int pw = wave - 4;
    const bf16_t* x = (const bf16_t*)a.x;
    // Lane l of a DMA instruction fills 16-byte slot l of a 1 KiB block = 8 rows x 128 B: row l >> 3, slot l & 7 of the row,
    // which holds logical k-octet (l & 7) ^ (row & 7) - the swizzle sits on the SOURCE address.
    const int lr = lane >> 3, lo = (lane & 7) ^ lr;
    constexpr int XLMAX = (NB + 3) / 4;
    const int xlw = (NB - pw + 3) / 4;                     // DMA instructions of this wave per stage: blocks pw, pw + 4, ...
    const bf16_t* src[XLMAX];
#pragma unroll
    for (int j = 0; j < XLMAX; ++j) {
      long row = m0 + 8 * (pw + 4 * j) + lr;
      if (row >= a.M) row = a.M - 1;                       // rows past M: finite values, masked in the epilogue
      src[j] = x + row * K + 8 * lo;
    }
    const int tadj = (ktail && lo >= 4) ? -32 : 0;         // half stage: the octets past K re-read the valid half (never used)
    int isl = 0;                                           // ring slot of the next stage to issue
    auto issue_x = [&](int s) {
      const lds_t slot = xring + isl * XS;
      const int go = 64 * s + (s == S - 1 ? tadj : 0);
#pragma unroll
      for (int j = 0; j < XLMAX; ++j)
        if (pw + 4 * j < NB) glds16(src[j] + go, slot + (pw + 4 * j) * 1024);
      if (++isl == RX) isl = 0;
    };
    // gate groups of the tile's rows: rows < gsplit -> g0, else g1 (a tile spans at most two: rows_per_group >= BM)
    int g0 = 0, g1 = 0, gsplit = BM;
    if (HAS_GATE) {
      const long rpg = a.pro.rows_per_group;
      g0 = (int)(m0 / rpg);
      const long nb = (long)(g0 + 1) * rpg;                // first row of the next group
      gsplit = nb - m0 < BM ? (int)(nb - m0) : BM;
      g1 = nb < a.M ? g0 + 1 : g0;
    }
    const float* gate0 = HAS_GATE ? a.pro.gate + (long)g0 * K : nullptr;
    const float* gate1 = HAS_GATE ? a.pro.gate + (long)g1 * K : nullptr;
    constexpr int NRP = (BM + 63) / 64;                    // 64-row passes of the transform
    f32x8 tsc[2], tsh[2], tga[2];                          // tables of the NEXT transform (SGPRs)
    const bool two_groups = HAS_GATE && gsplit < BM;
    auto tables = [&](int s) {                             // request the tables of stage s (waited for with the next lgkmcnt(0))
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        int k = 64 * s + 16 * pw + 8 * o;
        if (k > K - 8) k = K - 8;                          // (half stage: the unused octets read a valid table entry)
        if (HAS_BN) { tsc[o] = sld8(a.pro.scale + k); tsh[o] = sld8(a.pro.shift + k); }
        if (HAS_GATE) tga[o] = sld8(gate0 + k);
      }
    };
    auto pin_tables = [&]() {
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        if (HAS_BN) { sreg_pin(tsc[o]); sreg_pin(tsh[o]); }
        if (HAS_GATE) sreg_pin(tga[o]);
      }
    };
    // One 8-channel piece, all lanes on one gate row, written stage by stage over the 8 elements (8 independent exps, then 8
    // reciprocals).  What it costs (profiles/r05_pwk8_trace_*.txt; the same numbers with the consumers' MFMAs compiled out, with
    // two producers per SIMD they double): plain VALU 4 cycles, packed fp32 and 64-bit moves 8, v_exp_f32 / v_rcp_f32 16 per
    // wave instruction - 8 unpack + 4 mov + 4 pk_fma + 8 mul + 8 exp + 8 add + 8 rcp + 8 pk_mul + 4 cvt = 496 cycles per piece,
    // half of it the two transcendentals; four pieces per stage at 80 rows + ~500 cycles of LDS / table latency = the 2500
    // cycles the trace shows.  The transform is bound by the SIMD's VALU throughput, not by scheduling.
    auto piece = [&](const u16x8& rv, const f32x8& sc, const f32x8& sh, const f32x8& ga) {
      float v[8], e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = bf2f(rv[j]);
        if (HAS_BN) v[j] = v[j] * sc[j] + sh[j];
      }
      if (HAS_ACT) {
        // (the pins are what holds the stages apart: a scheduling fence alone does not stop the IR passes from sinking every
        // element's arithmetic down to its use)
#pragma unroll
        for (int j = 0; j < 8; ++j) { e[j] = v[j] * -1.4426950408889634f; reg_pin(e[j]); }
#pragma unroll
        for (int j = 0; j < 8; ++j) { e[j] = fast_exp2(e[j]); reg_pin(e[j]); }
#pragma unroll
        for (int j = 0; j < 8; ++j) { e[j] = 1.0f + e[j]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) { e[j] = fast_rcp(e[j]); reg_pin(e[j]); }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= e[j];
      }
      if (HAS_GATE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= ga[j];
      }
      return pack8(v);
    };
    auto piece2 = [&](const u16x8& rv, const f32x8& sc, const f32x8& sh, const float* ga) {   // per-lane gate row
      float v[8], g[8];
      load8f(ga, g);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float z = bf2f(rv[j]);
        if (HAS_BN) z = z * sc[j] + sh[j];
        if (HAS_ACT) z = siluf_(z);
        v[j] = z * g[j];
      }
      return pack8(v);
    };
    auto transform = [&](lds_t slot, int s) {              // raw x stage s -> activation, in place (tables requested a step ago)
      u16x8 rv[NRP][2];
      lds_t roff[NRP][2];
#pragma unroll
      for (int p = 0; p < NRP; ++p) {
        const int r = lane + 64 * p;
        if (r < BM) {
#pragma unroll
          for (int o = 0; o < 2; ++o) {
            roff[p][o] = slot + r * 128 + (((2 * pw + o) ^ (r & 7)) << 4);
            rv[p][o] = lds_ld16(roff[p][o]);
          }
        }
      }
      wait_lgkm0();                                        // (covers the tables too)
      pin_tables();
#pragma unroll
      for (int p = 0; p < NRP; ++p) {
        const int r = lane + 64 * p;
        if (r < BM) {
          reg_pin(rv[p][0]); reg_pin(rv[p][1]);
          // the gate row is wave-uniform (scalar tables) unless the 64-row pass reaches into the tile's second group (one tile
          // in ~11): those lanes then read their row's gate values themselves (ordinary loads: hipcc drains vmcnt for them)
          const bool all_a = !HAS_GATE || gsplit >= 64 * (p + 1) || gsplit >= BM;
#pragma unroll
          for (int o = 0; o < 2; ++o) {
            u16x8 out;
            if (all_a) out = piece(rv[p][o], tsc[o], tsh[o], tga[o]);
            else {
              int k = 64 * s + 16 * pw + 8 * o;
              if (k > K - 8) k = K - 8;
              out = piece2(rv[p][o], tsc[o], tsh[o], (r >= gsplit ? gate1 : gate0) + k);
            }
            lds_st16(roff[p][o], out);
          }
        }
      }
    };
    // how many of this wave's DMA instructions may stay in flight while batch b must have landed, at step t
    auto allowed = [&](int t, int b) {
      int last = (t > 0 ? t : 0) - 1 + DX;
      if (last > S - 1) last = S - 1;
      const int n = last - b;
      return n > 0 ? n * xlw : 0;
    };

    for (int s = 0; s < DX && s < S; ++s) issue_x(s);
    if (XF) tables(0);
    int tsl = 0;                                           // ring slot of the next stage to transform
    for (int t = -NEED; t < S; ++t) {
      const int b = t + NEED;                              // this stage's x must have landed (XF: it is transformed during this step)
      PWK_STAMP(t + NEED, 0);
      if (b < S) wait_vm_dyn(allowed(t, b));
      wait_lgkm0();                                        // this wave's transform stores of the previous step
      PWK_STAMP(t + NEED, 1);
      raw_barrier();
      PWK_STAMP(t + NEED, 2);
      if (t >= 0 && t + DX < S) issue_x(t + DX);           // into the slot stage t - 1 has left
      PWK_STAMP(t + NEED, 3);
      if (XF && b < S) {
        transform(xring + tsl * XS, b);
        if (b + 1 < S) tables(b + 1);
        if (++tsl == RX) tsl = 0;
      }
      PWK_STAMP(t + NEED, 4);
    }
    PWK_STAMP(S + NEED, 0);
    wait_vm<0>();
    wait_lgkm0();
  } else {
    // ================================================================ CONSUMER
    const int i = lane & 15, q = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf)
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // filter stream: fragment-major copy, fragment (ks, nf) = 1 KiB at ((ks * NFT + nf) * 512 + lane * 8) elements
    const bf16_t* wfp[NFW];
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf) {
      int f = wn * NFW + nf;
      if (f >= NFT) f = NFT - 1;                           // columns past N: computed, never stored
      wfp[nf] = (const bf16_t*)a.w_frag + (long)f * 512 + lane * 8;
    }
    const long wstep = (long)NFT * 512;                    // elements per k-step
    u16x8 wf[4][NFW];                                      // [k-step & 3][column fragment]
    int wk = 0;                                            // next k-step to request
    auto load_w = [&](auto Pc) {                           // always NFW loads (past the end: the last k-step again), so that the counts hold
      constexpr int P = decltype(Pc)::value;
      const int kk = wk < KST ? wk : KST - 1;
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) gld16(wf[P][nf], wfp[nf] + kk * wstep);
      ++wk;
    };
    // fragment (16 rows, k-step ks) of a slot: row i at i * 128, octet 4 ks + q in 16-byte slot (4 ks + q) ^ (i & 7)
    const lds_t fo0 = wm * MFW * 2048 + i * 128 + ((q ^ (i & 7)) << 4), fo1 = wm * MFW * 2048 + i * 128 + (((4 + q) ^ (i & 7)) << 4);
    u16x8 xf[2][MFW];                                      // [k-step of the stage][row fragment]
    int csx = 0;                                           // ring slot of the stage being consumed

    load_w(ic<0>()); load_w(ic<1>()); load_w(ic<2>());
    for (int t = -NEED; t < 0; ++t) raw_barrier();
    // (x(0) is complete and visible: its step's barrier is behind us)
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf) xf[0][mf] = lds_ld16(xring + mf * 2048 + fo0);

    // one k-step: wait for its operands, then MFMAs with the NEXT k-step's fragment reads (from `nxt`, if any) and the filter
    // request of k-step + DW spread between them
    auto kstep = [&](auto Pc, auto KSc, lds_t nxt, int s) {
      constexpr int P = decltype(Pc)::value, KS = decltype(KSc)::value;
      PWK_STAMP(s + NEED, 3 * KS);
      wait_vm<(DW - 1) * NFW>();
      wait_lgkm0();
      PWK_STAMP(s + NEED, 3 * KS + 1);
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) reg_pin(wf[P][nf]);
#pragma unroll
      for (int mf = 0; mf < MFW; ++mf) reg_pin(xf[KS][mf]);
      MDS_SCHED_FENCE();
      load_w(ic<(P + DW) & 3>());
      MDS_SCHED_FENCE();
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
        for (int mf = 0; mf < MFW; ++mf) {
          mma16(wf[P][nf], xf[KS][mf], acc[mf][nf]);     // acc[r] = y[m = i][n = 4q + r]
          // one fragment read of the next k-step behind each of the first MFW MFMAs (past the end of the stream: a stale slot,
          // finite or not, never multiplied)
          if (nf == 0) xf[KS ^ 1][mf] = lds_ld16(nxt + mf * 2048 + (KS ? fo0 : fo1));
          MDS_SCHED_FENCE();
        }
      PWK_STAMP(s + NEED, 3 * KS + 2);
    };
    auto stage = [&](int s, auto SPc) {
      constexpr int SP = decltype(SPc)::value;
      const lds_t xb = xring + csx * XS;
      int nsx = csx + 1;
      if (nsx == RX) nsx = 0;
      const lds_t xn = xring + nsx * XS;
      const bool last = s == S - 1, two = !(ktail && last);
      kstep(ic<2 * SP>(), ic<0>(), xb, s);
      raw_barrier();                                       // stage s + 1 is complete in its slot; the producers may refill stage s - 1's
      if (two) kstep(ic<2 * SP + 1>(), ic<1>(), xn, s);      // (a half stage is always the last)
      csx = nsx;
    };
    for (int s = 0; s < S; s += 2) {
      stage(s, ic<0>());
      if (s + 1 < S) stage(s + 1, ic<1>());
    }
    PWK_STAMP(S + NEED, 0);
    wait_vm<0>();
    wait_lgkm0();
  }
#ifdef PWK_TRACE
  raw_barrier();
  if (trc) {
    const unsigned long long* tl = (const unsigned long long*)(smem + RX * XS);
    for (int e = tid; e < (S + NEED + 1) * 64; e += 512) ((unsigned long long*)a.split_part)[e] = tl[e];
  }
  PWK_STAMP(S + NEED + 1, 0);
#endif

  // ---- epilogue: the whole tile in row-major order through LDS
  raw_barrier();                                           // every consumer is past its last fragment read: the ring is free
  constexpr int SP = BNP + 4;                              // floats per staged row (784 B at 192 columns: 16 B x odd)
  float* stage_f = (float*)smem;                           // [BM][SP]
  if (wave < 4) {
    const int i = lane & 15, q = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf)
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf)
        *(f32x4*)(stage_f + (16 * (wm * MFW + mf) + i) * SP + 16 * (wn * NFW + nf) + 4 * q) = acc[mf][nf];
  }
  __syncthreads();
  if (rg < RG) {
#pragma unroll
    for (int jj = 0; jj < JMAX; ++jj) {
      const int rr = rg + RG * jj;
      const long m = m0 + rr;
      if (rr < BM && m < a.M) {
        const f32x4 lo4 = *(const f32x4*)(stage_f + rr * SP + 8 * c), hi4 = *(const f32x4*)(stage_f + rr * SP + 8 * c + 4);
        float v[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        if (DG) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            v[j] += bf2f(rres[jj][j]);
            if (post) {
              const float ys = bf2f(rys[jj][j]);
              if (a.post.mode == MDS_POST_SILU) v[j] *= silu_gradf_(ys * pb[0][j] + pb[1][j]);   // g replaces u in memory
              const float g = Elem<bf16_t>::rnd(v[j]) * rmk[jj];                                 // the sums see what later readers will read
              cs[j] += g;
              css[j] += g * ((ys - pb[2][j]) * pb[3][j]);
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) { cs[j] += v[j]; css[j] += v[j] * v[j]; }
        }
        *(u16x8*)(y + m * N + 8 * c) = pack8(v);
      }
    }
  }
  if (sdst) {
    float* red = stage_f + BM * SP;                        // [RG][2][N]
    if (rg < RG) {
      float* rp = red + (long)rg * 2 * N + 8 * c;
      *(f32x4*)rp = (f32x4){cs[0], cs[1], cs[2], cs[3]};
      *(f32x4*)(rp + 4) = (f32x4){cs[4], cs[5], cs[6], cs[7]};
      *(f32x4*)(rp + N) = (f32x4){css[0], css[1], css[2], css[3]};
      *(f32x4*)(rp + N + 4) = (f32x4){css[4], css[5], css[6], css[7]};
    }
    __syncthreads();
    double* sl = sdst + (long)(blockIdx.x % MDS_STAT_SLOTS) * 2 * N;
    for (int e = tid; e < 2 * N; e += 512) {               // thread = channel: one coalesced fp64 atomic per instruction
      float t = 0.f;
      for (int g = 0; g < RG; ++g) t += red[g * 2 * N + e];
      atomicAdd(sl + e, (double)t);
    }
  }
}

// LDS bytes of a launch: the x ring, or the epilogue's staged tile + row-group sums
template <int MFW, int WM, int NFW, int WN, int DX>
size_t pwk8_smem(int N) {
  const int BM = 16 * MFW * WM, BNP = 16 * NFW * WN;
  size_t ring = (size_t)(DX + 1) * BM * 128;
#ifdef PWK_TRACE
  ring += 16 * 1024;
#endif
  const int RG = 512 / (N >> 3);
  const size_t epi = ((size_t)BM * (BNP + 4) + (size_t)RG * 2 * N) * 4;
  return ring > epi ? ring : epi;
}

bool pwk8_shape_ok(long M, int K, int N, int dtype) {
  return dtype == MDS_BF16 && K % 32 == 0 && K >= 64 && N % 16 == 0 && N <= 192 && N > 64 && M < 4294967295L;
}
}  // namespace

// does a launch of this shape take the K-streaming kernel when it is given the fragment-major filter copy?  (the planner asks
// before it schedules the extra MDS_PACK_FRAG_* job)
extern "C" int mds_pw_fwd_wants_frag(long M, int K, int N, int dtype, int data_gradient) {
  const int knob = mds_knob(MDS_KNOB_PWK);
  if (knob == 1 || !pwk8_shape_ok(M, K, N, dtype)) return 0;
  if (knob == 2) return 1;
  // Rule: the 192-column layers at <= 40 k rows (stage 5, the 3D blocks, the 2D projection: one round of one-block-per-CU tiles; the
  // 73 600-row layers of stages 3 / 4 take two or three rounds of such tiles and lose to the general kernel's 3-4 blocks per CU,
  // profiles/r05_pwk8_kbench.txt) - and FORWARD launches only.  The data-gradient form is 7-20 % faster alone (24.6 vs 31.2 us at
  // 18 400 x 1152 -> 192) and 0.13 ms per step SLOWER inside the step (profiles/r05_pwk8_instep.txt): a 512-thread block at ~236
  // VGPRs needs a whole CU, so it cannot share CUs with the weight-gradient stream that runs beside the backward chain; the forward
  // has nothing beside it.  Knob 3 = data gradients too, 4 = data gradients only (A/B).
  if ((knob == 0 && data_gradient) || (knob == 4 && !data_gradient)) return 0;
  return K >= N && K >= 128 && N > 128 && M >= 4096 && M <= 40000;
}

// 1 = not taken (the general kernel runs), 0 = launched, < 0 = error
int pw_fwd_k_try(const mds_pw_fwd_args* a, mds_stream_t stream) {
  if (!a->w_frag || a->epi.mode != MDS_EPI_NONE || a->split > 1) return 1;
  const int N = a->N, mode = a->pro.mode;
  const bool post = a->post.mode != MDS_POST_NONE;
  const bool dg = a->form ? a->form == 2 : (post || a->residual != nullptr);      // (the planner's own flag when it gave one)
  if (!dg && (post || a->residual != nullptr)) return 1;                             // a forward launch with a fused residual / post sums: the general kernel
  if (!mds_pw_fwd_wants_frag(a->M, a->K, a->N, a->dtype, dg)) return 1;
  if (dg && (mode != MDS_PRO_NONE || a->stats)) return 1;
  // tile shapes: 192 columns = 4 waves x 3 fragments, 128 columns = 4 x 2, 96 columns = 2 x 3 with two wave rows.  Row count:
  // one block per CU (512 threads, ~200 VGPRs), so the launch runs in ceil(blocks / 256) rounds - the smallest rounds * BM wins
  // (18 400 rows: 80-row tiles, 230 blocks, one round; 73 600 rows: 96-row tiles, 767 blocks, three rounds)
  const int shape = N > 128 ? 0 : (N > 96 ? 1 : 2);
  static int cus = 0;      // one block per CU: the tile-height rule counts rounds of the chip
  if (cus == 0) cus = mds_cu_count();
  auto cost = [&](int bm) { const long blocks = cdiv(a->M, bm); return (long)cdiv(blocks, cus) * bm; };
  int BM;
  if (shape == 2) BM = 96;
  else { BM = 64; for (int bm : {80, 96, 128}) if (cost(bm) < cost(BM) && !(dg && bm > (shape == 0 ? 80 : 96))) BM = bm; }   // (dg: the prefetched epilogue operands of a taller tile do not fit beside the accumulators)
  const int kb = mds_knob(MDS_KNOB_PWK_BM);
  if (kb > 0 && shape != 2 && (kb == 64 || kb == 80 || ((kb == 96 || kb == 128) && !(dg && kb > (shape == 0 ? 80 : 96))))) BM = kb;
  const bool gated = mode == MDS_PRO_BN_SILU_GATE || mode == MDS_PRO_GATE;
  if (gated && a->pro.rows_per_group < BM) return 1;     // a tile spans at most two gate rows
  const dim3 grid(cdiv(a->M, BM)), block(512);
#define PWK_GO(MFW, WM, NFW, WN, DX, PRO, TAIL)                                                                  \
  do { const size_t smem = pwk8_smem<MFW, WM, NFW, WN, DX>(N);                                                   \
       MDS_LAUNCH((pwk8_kernel<MFW, WM, NFW, WN, DX, PRO, TAIL>), grid, block, smem, stream, *a); } while (0)
#define PWK_ROWS(NFW, WN, DX, PRO, TAIL)                                                                         \
  do { if (BM == 64) PWK_GO(4, 1, NFW, WN, DX, PRO, TAIL); else if (BM == 80) PWK_GO(5, 1, NFW, WN, DX, PRO, TAIL); \
       else if (BM == 96) PWK_GO(6, 1, NFW, WN, DX, PRO, TAIL); else PWK_GO(8, 1, NFW, WN, DX, PRO, TAIL); } while (0)
#define PWK_SHAPE(DX, PRO, TAIL)                                                                                 \
  do { if (shape == 0) PWK_ROWS(3, 4, DX, PRO, TAIL); else if (shape == 1) PWK_ROWS(2, 4, DX, PRO, TAIL);        \
       else PWK_GO(3, 2, 3, 2, DX, PRO, TAIL); } while (0)
  if (dg) {
    if (shape == 2) PWK_GO(3, 2, 3, 2, 6, MDS_PRO_NONE, 1);
    else if (BM == 64) { if (shape == 0) PWK_GO(4, 1, 3, 4, 6, MDS_PRO_NONE, 1); else PWK_GO(4, 1, 2, 4, 6, MDS_PRO_NONE, 1); }
    else if (BM == 80) { if (shape == 0) PWK_GO(5, 1, 3, 4, 6, MDS_PRO_NONE, 1); else PWK_GO(5, 1, 2, 4, 6, MDS_PRO_NONE, 1); }
    else PWK_GO(6, 1, 2, 4, 6, MDS_PRO_NONE, 1);
  }
  else switch (mode) {
    case MDS_PRO_NONE: PWK_SHAPE(6, MDS_PRO_NONE, 0); break;
    case MDS_PRO_AFFINE: PWK_SHAPE(7, MDS_PRO_AFFINE, 0); break;
    case MDS_PRO_BN_SILU: PWK_SHAPE(7, MDS_PRO_BN_SILU, 0); break;
    case MDS_PRO_BN_SILU_GATE: PWK_SHAPE(7, MDS_PRO_BN_SILU_GATE, 0); break;
    default: PWK_SHAPE(7, MDS_PRO_GATE, 0); break;
  }
#undef PWK_SHAPE
#undef PWK_ROWS
#undef PWK_GO
  return mds_check_launch("pw_fwd (K-streaming)");
}
