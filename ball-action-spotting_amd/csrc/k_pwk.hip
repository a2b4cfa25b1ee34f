// k_pwk.hip — mds_pw_fwd for the K-STREAMING 1x1 GEMMs (bf16): K >= N, 64 < N <= 192 - the MBConv / 3D projections
// (mid -> cout, BN + SiLU + gate prologue) and the data gradients of the expansions (dy1[M][mid] * W -> dx[M][cin]).
//
// Why a second kernel.  The general kernel (k_pw.hip) tiles N in 128 columns (N = 192: two n-tiles, the second half
// padding, x read twice), stages BOTH operands through registers into LDS per 64-channel chunk and reads them back as
// fragments: per 64 rows x 64 channels a CU moves 32 KiB through its one texture-address path and ~100 KiB through
// LDS for 96 MFMAs - every unit is as busy as the matrix pipe and the phases of a block do not overlap
// (profiles/r05_pwk_v1_trace.txt: a 2500-cycle dependent chain per 32-channel stage).  Here:
//   * a block owns BM rows and ALL of N: x is read once, the output tile leaves once;
//   * the FILTER never touches LDS: mds_pack_weights writes a fragment-major copy (MDS_PACK_FRAG_*: the 1 KiB a wave's
//     A operand of one (k-step, 16-column) MFMA needs is contiguous, lane-linear), each wave fetches the fragments of its
//     own columns with one coalesced 16-byte load per lane, a stage ahead, into a second register set;
//   * x goes global -> LDS directly (global_load_lds_dwordx4, no staging registers) in 64-channel stages = full 128-byte
//     lines, into a ring; the waits are counted (s_waitcnt vmcnt(n), n = the instructions that may stay in flight), the
//     barrier is the bare s_barrier, and every LDS access inside the loop is inline asm (mds_platform_hw.h) - hipcc
//     would otherwise drain the ring with vmcnt(0) at each LDS access it can see;
//   * ONE barrier per stage.  With a prologue the raw x stage is transformed IN PLACE one stage ahead of its use, under
//     the MFMAs of the current stage: wave w owns k-octets 2w, 2w + 1 of the stage and lane r row r, so scale / shift /
//     gate of the octets are wave-uniform and arrive through the scalar cache (s_load) - nothing on vmcnt, no tables in LDS;
//   * LDS image of a stage: row r at r * 128, k-octet o in 16-byte slot o ^ (r & 7) (applied on the SOURCE address of the
//     DMA): conflict-free ds_read_b128 fragments;
//   * the epilogue works on the tile in row-major order, 32 rows per wave row at a time: accumulators -> LDS (fp32), then
//     thread = (row group, 8-column octet): residual / BatchNorm-backward operands are 16-byte coalesced loads (all
//     requested before the first pass), the output leaves as 16-byte row segments, and the column sums go row group ->
//     LDS -> ONE coalesced fp64 atomic per channel (the flush form that paid in the reduce kernels, elem.h).
#include <stdlib.h>
#include <type_traits>
#include "gemm.h"

// MFW x WM row fragments, NFW x WN column fragments (WM * WN = 4 waves); DX = prefetch distance of the x stream in
// 64-channel stages.  PRO: mds_pro_t mode.  TAIL: 0 = forward (statistics), 1 = data gradient (residual, mds_poststat_t).
template <int MFW, int WM, int NFW, int WN, int DX, int PRO, int TAIL>
__global__ __launch_bounds__(256, 2) void pwk_kernel(mds_pw_fwd_args a) {
  MDS_CHAIN_PRIO();
  static_assert(WM * WN == 4 && MFW % 2 == 0, "four waves; the epilogue takes two row fragments per pass");
  constexpr bool XF = PRO != MDS_PRO_NONE;
  static_assert(DX >= (XF ? 3 : 2), "the transform runs one stage ahead of its use, one DMA batch stays in flight");
  constexpr bool HAS_BN = PRO == MDS_PRO_AFFINE || PRO == MDS_PRO_BN_SILU || PRO == MDS_PRO_BN_SILU_GATE;
  constexpr bool HAS_ACT = PRO == MDS_PRO_BN_SILU || PRO == MDS_PRO_BN_SILU_GATE;
  constexpr bool HAS_GATE = PRO == MDS_PRO_BN_SILU_GATE || PRO == MDS_PRO_GATE;
  constexpr int BM = 16 * MFW * WM, BNP = 16 * NFW * WN;
  constexpr int XL = BM / 32;                            // DMA instructions (8 rows x 128 B) per wave and stage
  constexpr int NW = 2 * NFW;                            // filter fragment loads per wave and stage
  constexpr int XS = BM * 128, RX = DX + 1;              // bytes per ring slot, ring length
  MDS_DYN_SMEM(smem);
  const lds_t xring = lds_addr_of(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const long m0 = (long)blockIdx.x * BM;
  const int K = a.K, N = a.N, S = (K + 63) >> 6, KST = (K + 31) >> 5, NFT = (N + 15) >> 4;
  const bool ktail = (K & 63) != 0;                      // the last stage holds 32 channels (K % 64 == 32)
  const bf16_t* x = (const bf16_t*)a.x;

  // ---- x stream.  Lane l of a DMA instruction fills 16-byte slot l of a 1 KiB block = 8 rows x 128 B: row l >> 3, slot
  // l & 7 of the row, which holds logical k-octet (l & 7) ^ (row & 7) - the swizzle sits on the SOURCE address.
  const int lr = lane >> 3, lo = (lane & 7) ^ lr;
  const bf16_t* src[XL];
#pragma unroll
  for (int j = 0; j < XL; ++j) {
    long row = m0 + 8 * (wave + 4 * j) + lr;
    if (row >= a.M) row = a.M - 1;                         // rows past M: finite values, masked in the epilogue
    src[j] = x + row * K + 8 * lo;
  }
  const int tadj = (ktail && lo >= 4) ? -32 : 0;           // half stage: the octets past K re-read the valid half (never used)
  int isl = 0;                                             // ring slot of the next stage to issue
  auto issue_x = [&](int s) {
    const lds_t slot = xring + isl * XS;
    const int go = 64 * s + (s == S - 1 ? tadj : 0);
#pragma unroll
    for (int j = 0; j < XL; ++j) glds16(src[j] + go, slot + (wave + 4 * j) * 1024);
    if (++isl == RX) isl = 0;
  };

  // ---- filter stream: fragment-major copy, fragment (ks, nf) = 1 KiB at ((ks * NFT + nf) * 512 + lane * 8) elements
  const bf16_t* wfp[NFW];
#pragma unroll
  for (int nf = 0; nf < NFW; ++nf) {
    int f = wn * NFW + nf;
    if (f >= NFT) f = NFT - 1;                             // columns past N: computed, never stored
    wfp[nf] = (const bf16_t*)a.w_frag + (long)f * 512 + lane * 8;
  }
  u16x8 wf[2][2][NFW];                                     // [register set][k-step of the stage][column fragment]
  auto load_w = [&](int s, auto Pc) {
    constexpr int P = decltype(Pc)::value;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      int kk = 2 * s + ks;
      if (kk >= KST) kk = KST - 1;                         // half stage: a valid fragment again (its MFMAs are skipped)
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) gld16(wf[P][ks][nf], wfp[nf] + (long)kk * NFT * 512);
    }
  };

  // ---- prologue transform of a raw x stage (XF), in place, one stage ahead of its use: wave w owns k-octets 2w, 2w + 1 of
  // the stage and lane r row r; the octets' scale / shift / gate rows are fetched through the scalar cache a stage earlier
  int g0 = 0, g1 = 0, gsplit = BM;                         // groups of the tile's rows: rows < gsplit -> g0, else g1
  if (HAS_GATE) {
    const long rpg = a.pro.rows_per_group;
    g0 = (int)(m0 / rpg);
    const long nb = (long)(g0 + 1) * rpg;                  // first row of the next group
    gsplit = nb - m0 < BM ? (int)(nb - m0) : BM;
    g1 = nb < a.M ? g0 + 1 : g0;
  }
  const bool two_groups = HAS_GATE && gsplit < BM;
  const float* gate0 = HAS_GATE ? a.pro.gate + (long)g0 * K : nullptr;
  const float* gate1 = HAS_GATE ? a.pro.gate + (long)g1 * K : nullptr;
  f32x8 tsc[2], tsh[2], tga[2];                            // tables of the NEXT transform (SGPRs)
  auto tables = [&](int s) {                               // request the tables of stage s (waited for with the next lgkmcnt(0))
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      int k = 64 * s + 16 * wave + 8 * o;
      if (k > K - 8) k = K - 8;                            // (half stage: the unused octets read a valid table entry)
      if (HAS_BN) { tsc[o] = sld8(a.pro.scale + k); tsh[o] = sld8(a.pro.shift + k); }
      if (HAS_GATE) tga[o] = sld8(gate0 + k);
    }
  };
  u16x8 rv[BM / 64][2];
  lds_t roff[BM / 64][2];
  auto transform_read = [&](lds_t slot) {                  // request this thread's raw pieces
#pragma unroll
    for (int p = 0; p < BM / 64; ++p)
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const int r = lane + 64 * p;
        roff[p][o] = slot + r * 128 + (((2 * wave + o) ^ (r & 7)) << 4);
        rv[p][o] = lds_ld16(roff[p][o]);
      }
  };
  auto transform_apply = [&](int s) {                      // (after the lgkmcnt(0) that covers transform_read and tables)
    f32x8 tgb[2];
    if (two_groups) {                                      // a gate boundary inside the tile (one tile in ~14): second gate row, fetched here
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        int k = 64 * s + 16 * wave + 8 * o;
        if (k > K - 8) k = K - 8;
        tgb[o] = sld8(gate1 + k);
      }
      wait_lgkm0();
#pragma unroll
      for (int o = 0; o < 2; ++o) sreg_pin(tgb[o]);
    }
#pragma unroll
    for (int p = 0; p < BM / 64; ++p)
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const int r = lane + 64 * p;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float z = bf2f(rv[p][o][j]);
          if (HAS_BN) z = z * tsc[o][j] + tsh[o][j];
          if (HAS_ACT) z = siluf_(z);
          if (HAS_GATE) z *= (two_groups && r >= gsplit) ? tgb[o][j] : tga[o][j];
          v[j] = z;
        }
        lds_st16(roff[p][o], pack8(v));
      }
  };
  auto pin_tables = [&]() {
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      if (HAS_BN) { sreg_pin(tsc[o]); sreg_pin(tsh[o]); }
      if (HAS_GATE) sreg_pin(tga[o]);
    }
  };

#ifdef PWK_TRACE   /* experiment builds: s_memtime stamps of block PWK_TRACE, [stage][wave][phase] in LDS behind the ring, dumped through the (unused) split_part pointer */
  const bool trc = blockIdx.x == PWK_TRACE && a.split_part != nullptr;
  const lds_t trc_base = xring + RX * XS;
#define PWK_STAMP(s_, ph_) do { if (trc && lane == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); \
    asm volatile("ds_write_b64 %0, %1" ::"v"(trc_base + (((s_) * 4 + wave) * 8 + (ph_)) * 8), "v"(t_) : "memory"); } } while (0)
#else
#define PWK_STAMP(s_, ph_) ((void)0)
#endif
  // ---- accumulate
  f32x4 acc[MFW][NFW];
#pragma unroll
  for (int mf = 0; mf < MFW; ++mf)
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // fragment (16 rows, k-step ks) of a slot: row i at i * 128, octet 4 ks + q in 16-byte slot (4 ks + q) ^ (i & 7)
  const lds_t fo0 = wm * MFW * 2048 + i * 128 + ((q ^ (i & 7)) << 4), fo1 = wm * MFW * 2048 + i * 128 + (((4 + q) ^ (i & 7)) << 4);

  load_w(0, std::integral_constant<int, 0>());
  wait_vm<0>();                                            // (before anything can copy these registers)
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf) reg_pin(wf[0][ks][nf]);
  for (int s = 0; s < DX && s < S; ++s) issue_x(s);
  int csx = 0;                                             // ring slot of the stage being consumed
  if (XF) {
    tables(0);
    wait_vm<0>();
    wait_lgkm0();
    raw_barrier();
    pin_tables();
    transform_read(xring);
    wait_lgkm0();
#pragma unroll
    for (int p = 0; p < BM / 64; ++p) { reg_pin(rv[p][0]); reg_pin(rv[p][1]); }
    transform_apply(0);
    if (S > 1) tables(1);
  }
  auto stage = [&](int s, auto Pc) {
    constexpr int P = decltype(Pc)::value;
    // on entry: filter fragments of stage s are in (or on their way to) wf[P]; x of stage s (XF: and raw s + 1) was issued
    PWK_STAMP(s, 0);
    if (s >= 1 && s + DX <= S) wait_vm<(DX - (XF ? 2 : 1)) * (NW + XL)>();   // steady state: stage s - 1 issued its full batch
    else wait_vm<0>();
    wait_lgkm0();                                          // (XF: this wave's transform stores of stage s, the next tables)
    PWK_STAMP(s, 1);
    raw_barrier();                                         // x of stage s (XF: raw s + 1) has landed for everyone; everyone is past stage s - 1
    PWK_STAMP(s, 2);
    if (s + 1 < S) load_w(s + 1, std::integral_constant<int, P ^ 1>());
    if (s + DX < S) issue_x(s + DX);                       // into the slot stage s - 1 has left
    PWK_STAMP(s, 3);
    const lds_t xb = xring + csx * XS;
    const bool ks1 = !(ktail && s == S - 1);
    u16x8 xf0[MFW], xf1[MFW];
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf) xf0[mf] = lds_ld16(xb + mf * 2048 + fo0);
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf) xf1[mf] = lds_ld16(xb + mf * 2048 + fo1);
    const bool xform = XF && s + 1 < S;
    if (xform) {
      pin_tables();
      int nsx = csx + 1;
      if (nsx == RX) nsx = 0;
      transform_read(xring + nsx * XS);
    }
    wait_lgkm0();
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf) { reg_pin(xf0[mf]); reg_pin(xf1[mf]); }
    if (xform) {
#pragma unroll
      for (int p = 0; p < BM / 64; ++p) { reg_pin(rv[p][0]); reg_pin(rv[p][1]); }
    }
    // the filter fragments of this stage (requested a stage ago): everything younger may stay in flight
    if (s >= 1) { if (s + DX < S) wait_vm<NW + 2 * XL>(); else wait_vm<0>(); }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) reg_pin(wf[P][ks][nf]);
    PWK_STAMP(s, 4);
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf)
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) mma16(wf[P][0][nf], xf0[mf], acc[mf][nf]);   // acc[r] = y[m = i][n = 4q + r]
    if (ks1) {
#pragma unroll
      for (int mf = 0; mf < MFW; ++mf)
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf) mma16(wf[P][1][nf], xf1[mf], acc[mf][nf]);
    }
    PWK_STAMP(s, 5);
    if (xform) {                                           // the next stage's raw x -> activation, under this stage's MFMAs
      transform_apply(s + 1);
      if (s + 2 < S) tables(s + 2);
    }
    PWK_STAMP(s, 6);
    if (++csx == RX) csx = 0;
  };
  for (int s = 0; s < S; s += 2) {
    stage(s, std::integral_constant<int, 0>());
    if (s + 1 < S) stage(s + 1, std::integral_constant<int, 1>());
  }
  PWK_STAMP(S, 0);
#ifdef PWK_TRACE
  wait_lgkm0();
  raw_barrier();
  if (trc) {
    const unsigned long long* tl = (const unsigned long long*)(smem + RX * XS);
    for (int e = tid; e < (S + 1) * 32; e += 256) ((unsigned long long*)a.split_part)[e] = tl[e];
  }
#endif

  // ---- epilogue: the tile in row-major order through LDS, 32 rows per wave row at a time
  wait_vm<0>();
  wait_lgkm0();
  raw_barrier();                                         // every wave is past its last fragment read: the ring is free
  constexpr int SP = BNP + 4;                            // floats per staged row (784 B at 192 columns: 16 B x odd)
  constexpr int NPASS = MFW / 2, RP = 32 * WM;           // passes, staged rows per pass
  float* stage_f = (float*)smem;                         // [RP][SP]
  const int NOCT = N >> 3, RG = 256 / NOCT;              // 8-column octets per row, row groups
  const int c = tid % NOCT, rg = tid / NOCT;
  constexpr int RGMIN = 256 / (BNP / 8), JMAX = (RP + RGMIN - 1) / RGMIN;
  bf16_t* y = (bf16_t*)a.y;
  constexpr bool DG = TAIL == 1;
  const bool post = DG && a.post.mode != MDS_POST_NONE;
  double* const sdst = DG ? (post ? a.post.stats : nullptr) : a.stats;
  float cs[8], css[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { cs[j] = 0.f; css[j] = 0.f; }
  // staged row rr of pass p -> tile row
  auto tile_row = [&](int p, int rr) { return ((rr >> 5) * MFW + 2 * p) * 16 + (rr & 31); };
  float pb[4][8];                                        // DG: scale, shift, mean, rstd of this thread's octet
  u16x8 rres[NPASS][JMAX], rys[NPASS][JMAX];
  float rmk[NPASS][JMAX];
  if (DG && rg < RG) {                                   // every operand of the thread's rows is requested up front
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) pb[t][j] = post ? a.post.bn[(long)t * N + 8 * c + j] : 0.f;
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
#pragma unroll
      for (int jj = 0; jj < JMAX; ++jj) {
        const int rr = rg + RG * jj;
        long m = m0 + tile_row(p, rr);
        if (!(rr < RP && m < a.M)) m = 0;
        rres[p][jj] = a.residual ? *(const u16x8*)((const bf16_t*)a.residual + m * N + 8 * c) : (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
        rys[p][jj] = post ? *(const u16x8*)((const bf16_t*)a.post.y + m * N + 8 * c) : (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
        rmk[p][jj] = (post && a.post.mode == MDS_POST_MASK) ? a.post.mask[(unsigned)m / (unsigned)a.post.rows_per_group] : 1.0f;
      }
  }
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    if (p) __syncthreads();                              // the previous pass's rows have been read
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf)
        *(f32x4*)(stage_f + (32 * wm + 16 * h + i) * SP + 16 * (wn * NFW + nf) + 4 * q) = acc[2 * p + h][nf];
    __syncthreads();
    if (rg < RG) {
#pragma unroll
      for (int jj = 0; jj < JMAX; ++jj) {
        const int rr = rg + RG * jj;
        const long m = m0 + tile_row(p, rr);
        if (rr < RP && m < a.M) {
          const f32x4 lo4 = *(const f32x4*)(stage_f + rr * SP + 8 * c), hi4 = *(const f32x4*)(stage_f + rr * SP + 8 * c + 4);
          float v[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
          if (DG) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              v[j] += bf2f(rres[p][jj][j]);
              if (post) {
                const float ys = bf2f(rys[p][jj][j]);
                if (a.post.mode == MDS_POST_SILU) v[j] *= silu_gradf_(ys * pb[0][j] + pb[1][j]);   // g replaces u in memory
                const float g = Elem<bf16_t>::rnd(v[j]) * rmk[p][jj];                              // the sums see what later readers will read
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
  }
  if (sdst) {
    float* red = stage_f + RP * SP;                      // [RG][2][N]
    if (rg < RG) {
      float* rp = red + (long)rg * 2 * N + 8 * c;
      *(f32x4*)rp = (f32x4){cs[0], cs[1], cs[2], cs[3]};
      *(f32x4*)(rp + 4) = (f32x4){cs[4], cs[5], cs[6], cs[7]};
      *(f32x4*)(rp + N) = (f32x4){css[0], css[1], css[2], css[3]};
      *(f32x4*)(rp + N + 4) = (f32x4){css[4], css[5], css[6], css[7]};
    }
    __syncthreads();
    double* sl = sdst + (long)(blockIdx.x % MDS_STAT_SLOTS) * 2 * N;
    for (int e = tid; e < 2 * N; e += 256) {             // thread = channel: one coalesced fp64 atomic per instruction
      float t = 0.f;
      for (int g = 0; g < RG; ++g) t += red[g * 2 * N + e];
      atomicAdd(sl + e, (double)t);
    }
  }
}

// LDS bytes of a launch: the x ring, or the epilogue's staged rows + row-group sums
template <int MFW, int WM, int NFW, int WN, int DX>
static size_t pwk_smem(int N) {
  const int BM = 16 * MFW * WM, BNP = 16 * NFW * WN;
  size_t ring = (size_t)(DX + 1) * BM * 128;
#ifdef PWK_TRACE
  ring += 12 * 1024;
#endif
  const int RG = 256 / (N >> 3);
  const size_t epi = ((size_t)32 * WM * (BNP + 4) + (size_t)RG * 2 * N) * 4;
  return ring > epi ? ring : epi;
}

static bool pwk_shape_ok(long M, int K, int N, int dtype) {
  return dtype == MDS_BF16 && K % 32 == 0 && K >= 64 && N % 16 == 0 && N <= 192 && N > 64 && M < 4294967295L;
}
// does a launch of this shape take the K-streaming kernel when it is given the fragment-major filter copy?  (the planner asks
// before it schedules the extra MDS_PACK_FRAG_* job)
extern "C" int mds_pw_fwd_wants_frag(long M, int K, int N, int dtype) {
  const int knob = mds_knob(MDS_KNOB_PWK);
  if (knob == 1 || !pwk_shape_ok(M, K, N, dtype)) return 0;
  return knob == 2 || (K >= N && K >= 128 && M >= 4096);
}

// 1 = not taken (the general kernel runs), 0 = launched, < 0 = error
int pw_fwd_k_try(const mds_pw_fwd_args* a, mds_stream_t stream) {
  if (!a->w_frag || a->epi.mode != MDS_EPI_NONE || a->split > 1 || !mds_pw_fwd_wants_frag(a->M, a->K, a->N, a->dtype)) return 1;
  const int N = a->N, mode = a->pro.mode;
  const bool post = a->post.mode != MDS_POST_NONE;
  const bool dg = post || a->residual != nullptr;
  if (dg && (mode != MDS_PRO_NONE || a->stats)) return 1;
  if (!(mode == MDS_PRO_NONE || mode == MDS_PRO_BN_SILU_GATE || mode == MDS_PRO_GATE || mode == MDS_PRO_BN_SILU || mode == MDS_PRO_AFFINE)) return 1;
  const bool wide = N > 128, mid = !wide && N > 96;    // 192-column / 128-column tiles of 64 rows; 96-column tiles of 128 rows
  const int BM = (wide || mid) ? 64 : 128;
  const bool gated = mode == MDS_PRO_BN_SILU_GATE || mode == MDS_PRO_GATE;
  if (gated && a->pro.rows_per_group < BM) return 1;   // a tile spans at most two gate rows
  const dim3 grid(cdiv(a->M, BM)), block(256);
#define PWK_GO(MFW, WM, NFW, WN, DX, PRO, TAIL)                                                                  \
  do { const size_t smem = pwk_smem<MFW, WM, NFW, WN, DX>(N);                                                    \
       MDS_LAUNCH((pwk_kernel<MFW, WM, NFW, WN, DX, PRO, TAIL>), grid, block, smem, stream, *a); } while (0)
#define PWK_SHAPE(DX, PRO, TAIL)                                                                                 \
  do { if (wide) PWK_GO(4, 1, 3, 4, DX, PRO, TAIL); else if (mid) PWK_GO(4, 1, 2, 4, DX, PRO, TAIL);             \
       else PWK_GO(4, 2, 3, 2, DX, PRO, TAIL); } while (0)
  if (dg) PWK_SHAPE(2, MDS_PRO_NONE, 1);
  else switch (mode) {
    case MDS_PRO_NONE: PWK_SHAPE(2, MDS_PRO_NONE, 0); break;
    case MDS_PRO_AFFINE: PWK_SHAPE(3, MDS_PRO_AFFINE, 0); break;
    case MDS_PRO_BN_SILU: PWK_SHAPE(3, MDS_PRO_BN_SILU, 0); break;
    case MDS_PRO_BN_SILU_GATE: PWK_SHAPE(3, MDS_PRO_BN_SILU_GATE, 0); break;
    default: PWK_SHAPE(3, MDS_PRO_GATE, 0); break;
  }
#undef PWK_SHAPE
#undef PWK_GO
  return mds_check_launch("pw_fwd (K-streaming)");
}
