// k_pwk.hip — mds_pw_fwd for the K-STREAMING 1x1 GEMMs (bf16): K >= N, N <= 192 - the MBConv / 3D projections
// (mid -> cout, BN + SiLU + gate prologue) and the data gradients of the expansions (dy1[M][mid] * W -> dx[M][cin]).
//
// Why a second kernel.  The general kernel (k_pw.hip) walks K in 64-channel chunks with ONE chunk of loads in flight per
// block, tiles N in 128 columns (N = 192: two n-tiles, the second half padding, x read twice) and stages through
// registers: at 18 400 x 1152 -> 192 a block is 18 dependent memory round trips and the launch runs at 1.2 TB/s of its
// operands.  This one is built around what such a launch needs:
//   * a block owns BM rows and ALL of N: x is read once, the output tile leaves once;
//   * both operands go global -> LDS directly (global_load_lds_dwordx4, no staging registers), in 32-channel stages,
//     into two rings with separate depths: wave 0 (..NXL-1) issues the x stream - HBM latency, small stages, deep ring -
//     and the other waves the filter stream - L2 latency, large stages, shallow ring.  vmcnt retires in issue order per
//     WAVE, so giving the two streams to different waves is what lets their depths differ;
//   * every wait is counted: s_waitcnt vmcnt(n) with n = the loads of the stages that may stay in flight, the barrier
//     is the bare s_barrier, and every LDS access inside the loop is inline asm (mds_platform_hw.h) - hipcc would
//     otherwise drain the rings with vmcnt(0) at each access it can see;
//   * one barrier per stage.  With a prologue the raw x stage is transformed one stage AHEAD of its use: wave w owns
//     k-octet w of the stage and lane r row r, so scale / shift / gate of the octet are wave-uniform and arrive through
//     the scalar cache (s_load) - no vector loads, no LDS tables, nothing on vmcnt;
//   * the LDS image of a 16-row x 32-channel fragment block is the 1 KiB a DMA instruction writes (lane-linear); the
//     16-byte slot of (row r, octet o) is 4 r + (o ^ ((4 - (r >> 2)) & 3)) - applied on the SOURCE address of the DMA -
//     which makes the ds_read_b128 of a fragment and of a transform pass conflict-free for every lane group;
//   * the epilogue works on the tile in row-major order: accumulators -> LDS (fp32), then thread = (row group, 8-column
//     octet): residual / BatchNorm-backward operands are 16-byte coalesced loads, the output leaves as 16-byte row
//     segments, and the column sums go row group -> LDS -> ONE coalesced fp64 atomic per channel (the flush form that
//     paid in the reduce kernels, elem.h block_reduce_channels).
#include <stdlib.h>
#include "gemm.h"

// MFW x WM row fragments, NFW x WN column fragments (WM * WN = 4 waves); NXL = waves that issue the x stream.
// PRO: mds_pro_t mode.  TAIL: 0 = forward (statistics), 1 = data gradient (residual, mds_poststat_t).
template <int MFW, int WM, int NFW, int WN, int NXL, int PRO, int TAIL>
__global__ __launch_bounds__(256, 2) void pwk_kernel(mds_pw_fwd_args a, int DX, int DW) {
  MDS_CHAIN_PRIO();
  static_assert(WM * WN == 4 && NXL >= 1 && NXL <= 3, "four waves, both streams have an issuer");
  constexpr bool XF = PRO != MDS_PRO_NONE;
  constexpr bool HAS_BN = PRO == MDS_PRO_AFFINE || PRO == MDS_PRO_BN_SILU || PRO == MDS_PRO_BN_SILU_GATE;
  constexpr bool HAS_ACT = PRO == MDS_PRO_BN_SILU || PRO == MDS_PRO_BN_SILU_GATE;
  constexpr bool HAS_GATE = PRO == MDS_PRO_BN_SILU_GATE || PRO == MDS_PRO_GATE;
  constexpr int BM = 16 * MFW * WM, NXB = MFW * WM, NWB = NFW * WN, BNP = 16 * NWB;
  constexpr int NWL = 4 - NXL, XL = (NXB + NXL - 1) / NXL, WL = (NWB + NWL - 1) / NWL;
  constexpr int XS = NXB * 1024, WS = NWB * 1024;       // bytes per ring slot
  constexpr int LMAX = XL > WL ? XL : WL;
  MDS_DYN_SMEM(smem);
  const lds_t lds0 = lds_addr_of(smem);
  const int RX = DX + 1, RW = DW + 1;
  const lds_t xring = lds0, wring = lds0 + RX * XS, xbuf = wring + RW * WS;
  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const long m0 = (long)blockIdx.x * BM;
  const int K = a.K, N = a.N, S = K >> 5;
  const bf16_t* x = (const bf16_t*)a.x;
  const bf16_t* w = (const bf16_t*)a.w;

  // ---- the two DMA streams: lane l of an instruction fills slot l of a 1 KiB fragment block = (row l >> 2, octet below).
  // ONE code path for both roles (per-wave values, no role branches around the loads: two structurally equal branches get
  // merged by the compiler into one body over POINTERS to the state variables, which then live in scratch memory).
  const bool xloader = wave < NXL;
  const int lr = lane >> 2, lo = (lane & 3) ^ ((4 - (lr >> 2)) & 3);
  const int nl = xloader ? XL : WL;                        // DMA instructions of this wave per stage
  const int D = xloader ? DX : DW, R = D + 1;              // this wave's prefetch distance and ring length (stages)
  const lds_t rbase = xloader ? xring : wring;
  const int SB = xloader ? XS : WS;
  const bf16_t* src[LMAX];
  int dblk[LMAX];
#pragma unroll
  for (int j = 0; j < LMAX; ++j) {
    const int lw = xloader ? wave : wave - NXL, stride = xloader ? NXL : NWL, nb = xloader ? NXB : NWB;
    int b = lw + stride * j;
    if (b >= nb) b = nb - 1;                               // (padding instruction: same block again, same bytes)
    long row = (xloader ? m0 : 0) + 16 * b + lr;
    const long lim = xloader ? a.M : (long)N;
    if (row >= lim) row = lim - 1;                         // rows past M / columns past N: finite values, never stored
    src[j] = (xloader ? x : w) + row * K + 8 * lo;
    dblk[j] = b * 1024;
  }
  int isl = 0;                                             // ring slot of the next stage to issue
#define PWK_ISSUE(s_)                                                                          \
  do { const lds_t slot_ = rbase + isl * SB;                                                   \
       _Pragma("unroll") for (int j = 0; j < LMAX; ++j)                                        \
         if (j < nl) glds16(src[j] + 32 * (s_), slot_ + dblk[j]);                              \
       if (++isl == R) isl = 0; } while (0)
  // wait until this wave's loads of stage `need_` have landed, given that stages 0 .. s_ + D - 1 (clamped to S) are issued
#define PWK_WAIT(s_, need_)                                                                    \
  do { int issued_ = (s_) + D; if (issued_ > S) issued_ = S;                                   \
       wait_vm_dyn((issued_ - 1 - (need_)) * nl); } while (0)

  // ---- prologue transform of one raw x stage (XF): wave = k-octet, lane = row; the octet's scale / shift / gate rows are
  // wave-uniform and come through the scalar cache
  int g0 = 0, g1 = 0, gsplit = BM;                         // groups of the tile's rows: rows < gsplit -> g0, else g1
  if (HAS_GATE) {
    const long rpg = a.pro.rows_per_group;
    g0 = (int)(m0 / rpg);
    const long nb = (long)(g0 + 1) * rpg;                  // first row of the next group
    gsplit = nb - m0 < BM ? (int)(nb - m0) : BM;
    g1 = nb < a.M ? g0 + 1 : g0;
  }
  const float* gate0 = HAS_GATE ? a.pro.gate + (long)g0 * K : nullptr;
  const float* gate1 = HAS_GATE ? a.pro.gate + (long)g1 * K : nullptr;
  auto transform = [&](int s, lds_t raw, lds_t dst) {
    const int k = 32 * s + 8 * wave;
    float sc[8], sh[8], ga[8], gb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = HAS_BN ? ld_uniform(a.pro.scale, k + j) : 1.f;
      sh[j] = HAS_BN ? ld_uniform(a.pro.shift, k + j) : 0.f;
      ga[j] = HAS_GATE ? ld_uniform(gate0, k + j) : 1.f;
      gb[j] = HAS_GATE ? ld_uniform(gate1, k + j) : 1.f;
    }
#pragma unroll
    for (int p = 0; p < BM / 64; ++p) {
      const int r = lane + 64 * p, rr = r & 15;
      const lds_t off = (r >> 4) * 1024 + (rr * 4 + (wave ^ ((4 - (rr >> 2)) & 3))) * 16;
      u16x8 rv = lds_ld16(raw + off);
      wait_lgkm0();
      reg_pin(rv);
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float z = bf2f(rv[j]);
        if (HAS_BN) z = z * sc[j] + sh[j];
        if (HAS_ACT) z = siluf_(z);
        if (HAS_GATE) z *= (r < gsplit ? ga[j] : gb[j]);
        v[j] = z;
      }
      lds_st16(dst + off, pack8(v));
    }
  };

#ifdef PWK_TRACE   /* experiment builds: s_memtime stamps of block PWK_TRACE, [stage][wave][phase] in LDS behind the rings, dumped through the (unused) split_part pointer */
  const bool trc = blockIdx.x == PWK_TRACE && a.split_part != nullptr;
  const lds_t trc_base = xbuf + (XF ? 2 * XS : 0);
  int cur_stage = 0;
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
  const lds_t fo = (i * 4 + (q ^ ((4 - (i >> 2)) & 3))) * 16;
  const lds_t fox = wm * MFW * 1024 + fo, fow = wn * NFW * 1024 + fo;
  auto mfma_stage = [&](lds_t xb, lds_t wb) {
    u16x8 wf[NFW], xf[MFW];
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf) wf[nf] = lds_ld16(wb + fow + nf * 1024);
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf) xf[mf] = lds_ld16(xb + fox + mf * 1024);
    wait_lgkm0();
    PWK_STAMP(cur_stage, 6);
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf) reg_pin(wf[nf]);
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf) reg_pin(xf[mf]);
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf)
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) mma16(wf[nf], xf[mf], acc[mf][nf]);   // acc[r] = y[m = i][n = 4q + r]
  };

  for (int s = 0; s < D && s < S; ++s) PWK_ISSUE(s);
  int csx = 0, csw = 0;                                    // ring slots of the stage being consumed
  if (XF) {
    if (xloader) PWK_WAIT(0, 0);
    raw_barrier();
    transform(0, xring, xbuf);
  }
  for (int s = 0; s < S; ++s) {
    // x stream: stage s (XF: the raw stage s + 1, transformed below); filter stream: stage s
    const int need = (xloader && XF) ? s + 1 : s;
#ifdef PWK_TRACE
    cur_stage = s;
#endif
    PWK_STAMP(s, 0);
    if (need < S) PWK_WAIT(s, need);
    wait_lgkm0();                                          // (XF: this wave's transform stores of stage s)
    PWK_STAMP(s, 1);
    raw_barrier();                                         // the stage has landed for everyone; everyone is past stage s - 1
    PWK_STAMP(s, 2);
    if (s + D < S) PWK_ISSUE(s + D);                       // into the slot stage s - 1 has left
    PWK_STAMP(s, 3);
    if (XF) {
      int nsx = csx + 1;
      if (nsx == RX) nsx = 0;
      if (s + 1 < S) transform(s + 1, xring + nsx * XS, xbuf + ((s + 1) & 1) * XS);
      PWK_STAMP(s, 4);
      mfma_stage(xbuf + (s & 1) * XS, wring + csw * WS);
      csx = nsx;
    } else {
      PWK_STAMP(s, 4);
      mfma_stage(xring + csx * XS, wring + csw * WS);
      if (++csx == RX) csx = 0;
    }
    PWK_STAMP(s, 5);
    if (++csw == RW) csw = 0;
  }
  PWK_STAMP(S, 0);
#ifdef PWK_TRACE
  wait_lgkm0();
  raw_barrier();
  if (trc) {
    const unsigned long long* tl = (const unsigned long long*)(smem + (trc_base - lds0));
    for (int e = tid; e < (S + 1) * 32; e += 256) ((unsigned long long*)a.split_part)[e] = tl[e];
  }
#endif
#undef PWK_ISSUE
#undef PWK_WAIT

  // ---- epilogue: the tile in row-major order through LDS
  wait_vm<0>();
  wait_lgkm0();
  raw_barrier();                                         // every wave is past its last fragment read: the rings are free
  constexpr int SP = BNP + 4;                            // floats per staged row (784 B at 192 columns: 16 B x odd)
  float* stage = (float*)smem;                           // [BM][SP]
#pragma unroll
  for (int mf = 0; mf < MFW; ++mf)
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf)
      *(f32x4*)(stage + (16 * (wm * MFW + mf) + i) * SP + 16 * (wn * NFW + nf) + 4 * q) = acc[mf][nf];
  __syncthreads();
  const int NOCT = N >> 3, RG = 256 / NOCT;              // 8-column octets per row, row groups
  const int c = tid % NOCT, rg = tid / NOCT;
  constexpr int RGMIN = 256 / (BNP / 8), JMAX = (BM + RGMIN - 1) / RGMIN;
  bf16_t* y = (bf16_t*)a.y;
  constexpr bool DG = TAIL == 1;
  const bool post = DG && a.post.mode != MDS_POST_NONE;
  double* const sdst = DG ? (post ? a.post.stats : nullptr) : a.stats;
  float cs[8], css[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { cs[j] = 0.f; css[j] = 0.f; }
  if (rg < RG) {
    float pb[4][8];                                      // DG: scale, shift, mean, rstd of this thread's octet
    u16x8 rres[JMAX], rys[JMAX];
    float rmk[JMAX];
    if (DG) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) pb[t][j] = post ? a.post.bn[(long)t * N + 8 * c + j] : 0.f;
#pragma unroll
      for (int jj = 0; jj < JMAX; ++jj) {                // every operand of the thread's rows is requested up front
        const int r = rg + RG * jj;
        long m = m0 + r;
        const bool ok = r < BM && m < a.M;
        if (!ok) m = 0;
        rres[jj] = a.residual ? *(const u16x8*)((const bf16_t*)a.residual + m * N + 8 * c) : (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
        rys[jj] = post ? *(const u16x8*)((const bf16_t*)a.post.y + m * N + 8 * c) : (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
        rmk[jj] = (post && a.post.mode == MDS_POST_MASK) ? a.post.mask[(unsigned)m / (unsigned)a.post.rows_per_group] : 1.0f;
      }
    }
#pragma unroll
    for (int jj = 0; jj < JMAX; ++jj) {
      const int r = rg + RG * jj;
      const long m = m0 + r;
      if (r < BM && m < a.M) {
        const f32x4 lo4 = *(const f32x4*)(stage + r * SP + 8 * c), hi4 = *(const f32x4*)(stage + r * SP + 8 * c + 4);
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
    float* red = stage + BM * SP;                        // [RG][2][N]
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

// LDS bytes of a launch: the rings (+ the two transformed x stages), or the epilogue's staged tile + row-group sums
template <int MFW, int WM, int NFW, int WN>
static size_t pwk_smem(int DX, int DW, bool xf, int N) {
  const int BM = 16 * MFW * WM, NXB = MFW * WM, NWB = NFW * WN, BNP = 16 * NWB;
  size_t ring = (size_t)(DX + 1) * NXB * 1024 + (size_t)(DW + 1) * NWB * 1024 + (xf ? 2 * NXB * 1024 : 0);
#ifdef PWK_TRACE
  ring += 12 * 1024;
#endif
  const int RG = 256 / (N >> 3);
  const size_t epi = ((size_t)BM * (BNP + 4) + (size_t)RG * 2 * N) * 4;
  return ring > epi ? ring : epi;
}

// 1 = not taken (the general kernel runs), 0 = launched, < 0 = error
int pw_fwd_k_try(const mds_pw_fwd_args* a, mds_stream_t stream) {
  const int knob = mds_knob(MDS_KNOB_PWK);
  if (knob == 1 || a->dtype != MDS_BF16 || a->epi.mode != MDS_EPI_NONE || a->split > 1) return 1;
  const int K = a->K, N = a->N, mode = a->pro.mode;
  if (K % 32 || K < 64 || N % 16 || N > 192 || N <= 64) return 1;
  const bool post = a->post.mode != MDS_POST_NONE;
  const bool dg = post || a->residual != nullptr;
  if (dg && (mode != MDS_PRO_NONE || a->stats)) return 1;
  if (!(mode == MDS_PRO_NONE || mode == MDS_PRO_BN_SILU_GATE || mode == MDS_PRO_GATE || mode == MDS_PRO_BN_SILU || mode == MDS_PRO_AFFINE)) return 1;
  const bool wide = N > 128, mid = !wide && N > 96;    // 192-column / 128-column tiles of 64 rows; 96-column tiles of 128 rows
  const int BM = (wide || mid) ? 64 : 128;
  const bool gated = mode == MDS_PRO_BN_SILU_GATE || mode == MDS_PRO_GATE;
  if (gated && a->pro.rows_per_group < BM) return 1;   // a tile spans at most two gate rows
  if (knob != 2 && (K < N || K < 128 || a->M < 4096)) return 1;
  // prefetch distances (stages of 32 channels): two blocks per CU share 160 KiB of LDS
  int DX = mds_knob(MDS_KNOB_PWK_DX), DW = mds_knob(MDS_KNOB_PWK_DW);
  if (DX <= 0) DX = 4;
  if (DW <= 0) DW = wide ? 3 : 4;
  const int S = K >> 5;
  if (DX > S) DX = S;
  if (DW > S) DW = S;
  if (DX < 2) DX = 2;
  if (DX > 10) DX = 10;
  if (DW > 10) DW = 10;
  const bool xf = mode != MDS_PRO_NONE;
  const dim3 grid(cdiv(a->M, BM)), block(256);
#define PWK_GO(MFW, WM, NFW, WN, NXL, PRO, TAIL)                                                                 \
  do { const size_t smem = pwk_smem<MFW, WM, NFW, WN>(DX, DW, xf, N);                                            \
       MDS_REQUIRE(smem <= 160 * 1024, "pw_fwd: K-streaming kernel needs %zu bytes of LDS", smem);               \
       MDS_LAUNCH((pwk_kernel<MFW, WM, NFW, WN, NXL, PRO, TAIL>), grid, block, smem, stream, *a, DX, DW); } while (0)
#define PWK_SHAPE(PRO, TAIL)                                                                                     \
  do { if (wide) PWK_GO(4, 1, 3, 4, 1, PRO, TAIL); else if (mid) PWK_GO(4, 1, 2, 4, 1, PRO, TAIL);               \
       else PWK_GO(4, 2, 3, 2, 2, PRO, TAIL); } while (0)
  if (dg) PWK_SHAPE(MDS_PRO_NONE, 1);
  else switch (mode) {
    case MDS_PRO_NONE: PWK_SHAPE(MDS_PRO_NONE, 0); break;
    case MDS_PRO_AFFINE: PWK_SHAPE(MDS_PRO_AFFINE, 0); break;
    case MDS_PRO_BN_SILU: PWK_SHAPE(MDS_PRO_BN_SILU, 0); break;
    case MDS_PRO_BN_SILU_GATE: PWK_SHAPE(MDS_PRO_BN_SILU_GATE, 0); break;
    default: PWK_SHAPE(MDS_PRO_GATE, 0); break;
  }
#undef PWK_SHAPE
#undef PWK_GO
  return mds_check_launch("pw_fwd (K-streaming)");
}
