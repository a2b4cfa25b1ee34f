// k_pwg.hip — mds_pwg_fwd: the prologue-free 1x1-convolution GEMM (bf16), the fast path of the
// inverted-residual blocks (stages 3-5, 3D tail, projections) in both directions.
//
//   y[m][n] = sum_p sum_k x_p[m][k] * w_p[g(m)][n][k] (+ bias[n]) (+ residual[m][n])
//
// Why a second GEMM kernel: the register-staged kernel of k_pw.hip serialises, per tile, a global-load round
// trip for every 64-wide K chunk and the epilogue behind it; at K = 96..1152 and 18 k - 74 k rows the launch
// is a chain of exposed latencies (10 % MFMA-busy, 45 % issue-stalled, round-2 PMC).  Here
//   * operands go global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPRs),
//     into an NS-deep ring of {x chunk [BM][64], w chunk [BN][64]} stages, with COUNTED vmcnt waits: NS-2
//     stages stay in flight across every barrier;
//   * a block is persistent over its range of tiles, so the ring already holds the next tile's first chunks
//     while the current tile's epilogue stores drain — no per-tile start-up bubble;
//   * LDS rows are 128 B with the 16-byte slot index XOR-ed with (row & 7): ds_read_b128 fragment reads are
//     conflict-free for every lane group; the swizzle is applied on the SOURCE address of the direct load
//     (the LDS image of such a load is lane-linear), the 8 lanes of a row still cover one 128-byte line;
//   * everything a prologue used to do is expressed in the operands: per-group weight sets (SE gate /
//     DropPath mask folded into the weights), a second operand pair and a bias row (BatchNorm backward in
//     its linear form), zero-padded K in the packed weights (x lanes past K fetch a zero page).
// Block = 4 waves as 2 x 2, wave tile (BM/2) x (BN/2), v_mfma_f32_16x16x32_bf16, accumulators
// acc[r] = y[m = i][n = 4q + r] (the product is formed transposed so a lane owns 4 consecutive channels).
#include <stdlib.h>
#include "gemm.h"

#define PWG_KC 64   // k per stage: one 128-byte row per operand row

struct PwgGeom { int tiles_per_group, ntn, ntiles, per_block, nst0, nst, dbg; };
// position in the tile sequence (n-tile fastest, then m-tile within the row group, then group): advanced
// incrementally — the 64-bit divisions of a direct tile -> coordinates map cost more scalar time than a stage's MFMAs
struct PwgCur {
  int nt, mtg, grp;
  MDS_DEV void init(int t, const PwgGeom& g) { nt = t % g.ntn; const int mt = t / g.ntn; mtg = mt % g.tiles_per_group; grp = mt / g.tiles_per_group; }
  MDS_DEV void next(const PwgGeom& g) {
    if (++nt == g.ntn) { nt = 0; if (++mtg == g.tiles_per_group) { mtg = 0; ++grp; } }
  }
};

// EPI: 0 = store (+ forward statistics) only — no loads in the epilogue, so nothing ever makes the compiler (or the
//          in-order vmcnt) drain the ring between tiles;  1 = + bias / residual;  2 = + BatchNorm-backward sums (post).
template <int BM, int BN, int NS, int EPI>
__global__ __launch_bounds__(384, 1) void pwg_kernel(mds_pwg_args a, PwgGeom g) {
  constexpr bool POST = EPI == 2;
  // Wave roles: waves 0-3 multiply and store (2 x 2 over the tile), waves 4-5 only issue the direct-to-LDS loads.
  // The counted vmcnt wait on a stage must see ONLY loads: vmcnt retires in order, so a wave that also has the
  // epilogue's stores and atomics in its queue would sit out their whole memory round trip at every tile boundary
  // (measured: 5 us per tile, 60 % of an expansion layer, when all four waves did both jobs).
  constexpr int NPROD = 2;
  constexpr int ROWS = BM + BN, NXI = BM / (8 * NPROD), NWI = BN / (8 * NPROD), NI = NXI + NWI;   // DMA wave-instructions per producer wave and stage
  constexpr int MFW = BM / 32, NFW = BN / 32;                                    // 16x16 fragments per wave
  constexpr int STAGE = ROWS * PWG_KC;                                           // elements per ring stage
  static_assert(NS >= 3 && BM % 32 == 0 && BN % 32 == 0, "ring geometry");
  constexpr int OPITCH = BN + 16;                    // bytes per staged output row: (BN/2) bf16 + 16 (bank spread)
  constexpr int OWAVE = 32 * OPITCH;                 // a wave stages 32 rows x BN/2 columns at a time
  constexpr int CPR = BN / 16;                       // 16-byte chunks per staged row
  MDS_DYN_SMEM(smem);
  bf16_t* ring = (bf16_t*)smem;   // [NS][ROWS][64], then [4 waves][32][OPITCH] output staging

  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);
  const bool producer = wave >= 4;
  const int pw = wave - 4;                          // producer index (row groups pw, pw + NPROD, ...)
  const int i = lane & 15, q = lane >> 4;
  const int wm = (wave >> 1) & 1, wn = wave & 1;
  const int lrow = lane >> 3;                       // DMA: row of the 8-row group this lane fills
  const int lcol = 8 * ((lane & 7) ^ lrow);         // ... and the LOGICAL k offset (elements) it fetches (source-side swizzle)
  const int N = a.N;
  const int tbeg = blockIdx.x * g.per_block;
  int tend = tbeg + g.per_block;
  if (tend > g.ntiles) tend = g.ntiles;
  const int total = (tend - tbeg) * g.nst;          // stages this block consumes
  const bf16_t* zeros = (const bf16_t*)a.zeros;

  // ------------------------------------------------------------------ the stage stream (issue side)
  // cursor of the next stage to ISSUE: tile, chunk within the tile, and the per-lane source pointers
  PwgCur icur, ccur;
  icur.init(tbeg, g);
  ccur = icur;
  int ic = 0;
  const bf16_t* xsrc[NXI];
  const bf16_t* wsrc[NWI];
  bool xok[NXI], wok[NWI];
  int ikrem = 0;   // elements of K left from this chunk's start for the current pair (K tail -> zero page)
  auto tile_coords = [&](const PwgCur& c, int& grp, long& m0, long& mend, int& n0) {
    n0 = c.nt * BN;
    grp = c.grp;
    m0 = (long)c.grp * a.rows_per_group + (long)c.mtg * BM;
    mend = (long)(c.grp + 1) * a.rows_per_group;
  };
  auto set_pair = [&](const PwgCur& t, int pair) {   // per-lane pointers of the first chunk of `pair` of tile t
    int grp, n0; long m0, mend;
    tile_coords(t, grp, m0, mend, n0);
    const bf16_t* x = (const bf16_t*)(pair ? a.x1 : a.x0);
    const bf16_t* w = (const bf16_t*)(pair ? a.w1 : a.w0);
    const int K = pair ? a.K1 : a.K0, Kp = (K + 63) & ~63;
    if (pair ? a.wg1 : a.wg0) w += (long)grp * N * Kp;
#pragma unroll
    for (int j = 0; j < NXI; ++j) {
      const long m = m0 + 8 * (pw + NPROD * j) + lrow;
      xok[j] = m < mend;
      xsrc[j] = x + (xok[j] ? m : m0) * K + lcol;
    }
#pragma unroll
    for (int j = 0; j < NWI; ++j) {
      const int n = n0 + 8 * (pw + NPROD * j) + lrow;
      wok[j] = n < N;                                      // rows past N read the zero page
      wsrc[j] = w + (long)(wok[j] ? n : 0) * Kp + lcol;
    }
    ikrem = K;
  };
  auto issue = [&](int js) {   // DMA of stage js into ring slot js % NS
    bf16_t* slot = ring + (js % NS) * STAGE;
    if (js < total) {
      if (ic == 0) set_pair(icur, 0);
      else if (ic == g.nst0) set_pair(icur, 1);
      const bool kin = lcol < ikrem;   // this lane's 8 k are inside K
#pragma unroll
      for (int j = 0; j < NXI; ++j) {
        glds16((xok[j] && kin) ? xsrc[j] : zeros, slot + 8 * (pw + NPROD * j) * PWG_KC);
        xsrc[j] += PWG_KC;
      }
#pragma unroll
      for (int j = 0; j < NWI; ++j) {
        glds16(wok[j] ? wsrc[j] : zeros, slot + (BM + 8 * (pw + NPROD * j)) * PWG_KC);
        wsrc[j] += PWG_KC;
      }
      ikrem -= PWG_KC;
      if (++ic == g.nst) { ic = 0; icur.next(g); }
    } else {
      // past the end of the stream: keep the vmcnt arithmetic uniform with harmless fetches of the zero page
#pragma unroll
      for (int j = 0; j < NI; ++j) glds16(zeros, slot + 8 * (pw + NPROD * j) * PWG_KC);
    }
  };

  // ------------------------------------------------------------------ consume side
  f32x4 acc[MFW][NFW];
  auto zero_acc = [&]() {
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf)
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();
  const int sw = i & 7;   // (row & 7) of every fragment row this lane reads (tile row bases are multiples of 8)
#ifndef MDS_EMU
  // Fragment reads are inline-asm ds_read_b128: for a compiler-visible LDS read hipcc inserts s_waitcnt vmcnt(0)
  // (it cannot prove the read does not alias an in-flight direct-to-LDS load), which would drain the ring every
  // stage.  Both k-steps of a stage are requested up front; the second one's data arrives under the first one's MFMAs.
  const uint32_t ring_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t fa0 = (uint32_t)(i * 128 + ((q ^ sw) << 4)), fa1 = (uint32_t)(i * 128 + (((q ^ sw) ^ 4) << 4));
  const uint32_t xbase = (BM / 2) * wm * 128, wbase = (BM + (BN / 2) * wn) * 128;
#endif
  auto compute = [&](int js) {
#ifndef MDS_EMU
    const uint32_t sb = ring_addr + (uint32_t)(js % NS) * (STAGE * 2);
    u16x8 xf[2][MFW], wf[2][NFW];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint32_t ax = sb + xbase + (ks ? fa1 : fa0), aw = sb + wbase + (ks ? fa1 : fa0);
#pragma unroll
      for (int mf = 0; mf < MFW; ++mf) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[ks][mf]) : "v"(ax), "n"(2048 * mf));
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[ks][nf]) : "v"(aw), "n"(2048 * nf));
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MFW + NFW) : "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int mf = 0; mf < MFW; ++mf) asm volatile("" : "+v"(xf[ks][mf]));   // consumers stay below the wait
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) asm volatile("" : "+v"(wf[ks][nf]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
        for (int mf = 0; mf < MFW; ++mf) mma16(wf[ks][nf], xf[ks][mf], acc[mf][nf]);
    }
#else
    const bf16_t* st = ring + (js % NS) * STAGE;
#pragma unroll
    for (int ks = 0; ks < PWG_KC / 32; ++ks) {
      const int off = 8 * ((4 * ks + q) ^ sw);
      u16x8 xf[MFW];
#pragma unroll
      for (int mf = 0; mf < MFW; ++mf) xf[mf] = ld_frag(st + ((BM / 2) * wm + 16 * mf + i) * PWG_KC + off);
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) {
        const u16x8 wf = ld_frag(st + (BM + (BN / 2) * wn + 16 * nf + i) * PWG_KC + off);
#pragma unroll
        for (int mf = 0; mf < MFW; ++mf) mma16(wf, xf[mf], acc[mf][nf]);
      }
    }
#endif
  };

#ifndef MDS_EMU
  const uint32_t out_addr = ring_addr + (uint32_t)(NS * STAGE * 2 + wave * OWAVE);
#endif
  auto epilogue = [&](const PwgCur& t) {
    int grp, n0; long m0, mend;
    tile_coords(t, grp, m0, mend, n0);
    bf16_t* y = (bf16_t*)a.y;
    const int nbase = n0 + (BN / 2) * wn + 4 * q;   // this lane's first channel (fragment nf adds 16*nf)
    float bs[NFW][4];
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) bs[nf][r] = (EPI >= 1 && a.bias && nbase + 16 * nf + r < N) ? a.bias[nbase + 16 * nf + r] : 0.f;
    float ps[NFW][4], pss[NFW][4];
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) { ps[nf][r] = 0.f; pss[nf][r] = 0.f; }
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf) {
      const long m = m0 + (BM / 2) * wm + 16 * mf + i;
      const bool ok = m < mend;
      const long mrow = ok ? m : m0;
      bf16_t* yrow = y + mrow * N + nbase;
      float mk = 1.0f;
      if (POST && a.post.mode == MDS_POST_MASK) mk = a.post.mask[(unsigned)mrow / (unsigned)a.post.rows_per_group];
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) {
        const int n = nbase + 16 * nf;
        if (n >= N) continue;     // N is a multiple of 16 and nbase of 4: a fragment is valid or not as a whole
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[mf][nf][r] + bs[nf][r];
        if (EPI >= 1 && a.residual) {
          float rr[4];
          load4((const bf16_t*)a.residual + mrow * N + n, rr);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += rr[r];
        }
        if (POST) {
          float ys[4], mu[4], rs[4];
          load4((const bf16_t*)a.post.y + mrow * N + n, ys);
          load4(a.post.bn + 2 * (long)N + n, mu);
          load4(a.post.bn + 3 * (long)N + n, rs);
          if (a.post.mode == MDS_POST_SILU) {
            float sc[4], sh[4];
            load4(a.post.bn + n, sc);
            load4(a.post.bn + (long)N + n, sh);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= silu_gradf_(ys[r] * sc[r] + sh[r]);
          }
          if (ok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float gq = bf2f(f2bf(v[r])) * mk;
              ps[nf][r] += gq;
              pss[nf][r] += gq * ((ys[r] - mu[r]) * rs[r]);
            }
          }
        } else if (ok) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { ps[nf][r] += v[r]; pss[nf][r] += v[r] * v[r]; }
        }
#ifdef MDS_EMU
        if (ok) store4(yrow + 16 * nf, v);
#else
        {   // stage the 8 bytes (inline asm: a compiler-visible LDS access would make hipcc drain the ring)
          typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
          const u32x2_ pk = {pack2(v[0], v[1]), pack2(v[2], v[3])};
          const uint32_t oa = out_addr + (uint32_t)((16 * (mf & 1) + i) * OPITCH + (16 * nf + 4 * q) * 2);
          asm volatile("ds_write_b64 %0, %1" ::"v"(oa), "v"(pk) : "memory");
        }
#endif
      }
#ifndef MDS_EMU
      if (mf & 1) {   // 32 rows staged: read them back as whole 16-byte chunks along the rows and store coalesced
        // (the direct form — 8 bytes per lane, 16 rows per instruction — made the epilogue 60 % of an expansion layer)
        const long r0 = m0 + (BM / 2) * wm + 16 * (mf - 1);
        u32x4 od[CPR / 2];
#pragma unroll
        for (int k = 0; k < CPR / 2; ++k) {
          const int c = lane + 64 * k, row = c / CPR, cc2 = c - row * CPR;
          const uint32_t ra = out_addr + (uint32_t)(row * OPITCH + 16 * cc2);
          asm volatile("ds_read_b128 %0, %1" : "=v"(od[k]) : "v"(ra));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < CPR / 2; ++k) {
          asm volatile("" : "+v"(od[k]));
          const int c = lane + 64 * k, row = c / CPR, cc2 = c - row * CPR;
          const int n = n0 + (BN / 2) * wn + 8 * cc2;
          if (r0 + row < mend && n < N && !(g.dbg & 8)) *(u32x4*)(y + (r0 + row) * N + n) = od[k];
        }
      }
#endif
    }
    float* const stat_dst = POST ? a.post.stats : a.stats;
    if (stat_dst && !(g.dbg & 16)) {
      float* sl = stat_dst + (long)((blockIdx.x * 4 + wave) % MDS_STAT_SLOTS) * 2 * N;
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s1 = sum_over_i16(ps[nf][r]), s2 = sum_over_i16(pss[nf][r]);
          const int n = nbase + 16 * nf + r;
          if (i == ((4 * nf + r) & 15) && n < N) {   // spread the atomics of a 16-lane group over its lanes
            atomicAdd(sl + n, s1);
            atomicAdd(sl + N + n, s2);
          }
        }
    }
    zero_acc();
  };

  // ------------------------------------------------------------------ the pipeline: ONE barrier per stage for all six waves
  static_assert((NS - 2) * NI < 64, "vmcnt immediate");
  if (producer) {
    for (int js = 0; js < NS - 1; ++js) issue(js);
    for (int js = 0; js < total; ++js) {
      MDS_WAIT_VMCNT((NS - 2) * NI);   // this wave's part of stage js has landed (NS-2 younger stages stay in flight)
      MDS_RAW_BARRIER();               // ... and the other producer's; the consumers are done reading stage js-1
      if (!(g.dbg & 2)) issue(js + NS - 1);   // refill the slot of stage js-1
    }
    MDS_WAIT_VMCNT(0);
  } else {
    int cc = 0;
    for (int js = 0; js < total; ++js) {
      MDS_RAW_BARRIER();
      if (!(g.dbg & 1)) compute(js);
      if (++cc == g.nst) {
        if (!(g.dbg & 4)) epilogue(ccur);
        cc = 0; ccur.next(g);
      }
    }
  }
}

extern "C" int mds_pwg_fwd(const mds_pwg_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->M > 0 && a->N > 0 && a->N % 16 == 0, "pwg_fwd: bad dims (N must be a multiple of 16)");
  MDS_REQUIRE(a->groups >= 1 && a->rows_per_group > 0 && (long)a->groups * a->rows_per_group == a->M, "pwg_fwd: groups * rows_per_group != M");
  MDS_REQUIRE(a->npairs == 1 || a->npairs == 2, "pwg_fwd: npairs");
  MDS_REQUIRE(a->x0 && a->w0 && a->K0 > 0 && a->K0 % 8 == 0 && a->y && a->zeros, "pwg_fwd: pair 0 / output / zero page");
  MDS_REQUIRE(a->npairs == 1 || (a->x1 && a->w1 && a->K1 > 0 && a->K1 % 8 == 0), "pwg_fwd: pair 1");
  MDS_REQUIRE(a->M < 4294967295L, "pwg_fwd: M must be below 2^32 rows");
  const bool post = a->post.mode != MDS_POST_NONE;
  if (post) {
    MDS_REQUIRE(a->post.y && a->post.bn && a->post.stats && !a->stats, "pwg_fwd: post statistics need y, bn, stats (and no forward stats)");
    MDS_REQUIRE(a->post.mode != MDS_POST_MASK || (a->post.mask && a->post.rows_per_group > 0), "pwg_fwd: post mask");
  }
  const int nst0 = (a->K0 + 63) / 64, nst = nst0 + (a->npairs == 2 ? (a->K1 + 63) / 64 : 0);
  // column tile: 192 when it divides the padded width better (N = 192, 576, 1152 ...), else 128
  const int pad128 = cdiv(a->N, 128) * 128, pad192 = cdiv(a->N, 192) * 192;
  const int BN = pad192 < pad128 ? 192 : 128;
  const int BM = 128;
  PwgGeom g;
  g.tiles_per_group = cdiv(a->rows_per_group, BM);
  g.ntn = cdiv(a->N, BN);
  MDS_REQUIRE((long)a->groups * g.tiles_per_group * g.ntn < 2000000000L, "pwg_fwd: too many tiles");
  g.ntiles = a->groups * g.tiles_per_group * g.ntn;
  g.nst0 = nst0; g.nst = nst;
  g.dbg = getenv("MDS_PWG_DBG") ? atoi(getenv("MDS_PWG_DBG")) : 0;
  int blocks = 256;                       // one persistent block per CU
  if (g.ntiles < blocks) blocks = g.ntiles;
  g.per_block = (g.ntiles + blocks - 1) / blocks;
  blocks = (g.ntiles + g.per_block - 1) / g.per_block;
#define PWG_GO(BN_, NS_)                                                                                          \
  do { const size_t smem = (size_t)NS_ * (128 + BN_) * PWG_KC * sizeof(bf16_t) + 4 * 32 * (BN_ + 16);            \
       if (post) MDS_LAUNCH((pwg_kernel<128, BN_, NS_, 2>), dim3(blocks), dim3(384), smem, stream, *a, g);        \
       else if (a->bias || a->residual) MDS_LAUNCH((pwg_kernel<128, BN_, NS_, 1>), dim3(blocks), dim3(384), smem, stream, *a, g); \
       else MDS_LAUNCH((pwg_kernel<128, BN_, NS_, 0>), dim3(blocks), dim3(384), smem, stream, *a, g); } while (0)
  if (BN == 192) PWG_GO(192, 3); else PWG_GO(128, 4);
#undef PWG_GO
  return mds_check_launch("pwg_fwd");
}

// ------------------------------------------------------------------ weight packing (per-step, tiny)
__global__ void pwg_pack_kernel(mds_pwg_pack_args a) {
  const int Kp = (a.K + 63) & ~63;
  const long total = (long)a.groups * a.N * Kp;
  bf16_t* dst = (bf16_t*)a.dst;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e % Kp);
    const long r = e / Kp;
    const int n = (int)(r % a.N), grp = (int)(r / a.N);
    float v = 0.f;
    if (k < a.K) {
      v = a.transposed ? a.src[(long)k * a.N + n] : a.src[(long)n * a.K + k];
      if (a.kscale) v *= a.kscale[(long)grp * a.K + k];
      if (a.nscale) v *= a.nscale[n];
    }
    dst[e] = f2bf(v);
  }
}
extern "C" int mds_pwg_pack(const mds_pwg_pack_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->src && a->dst && a->N > 0 && a->K > 0 && a->groups >= 1, "pwg_pack: bad args");
  const long total = (long)a->groups * a->N * ((a->K + 63) & ~63);
  int blocks = cdiv(total, 256 * 4);
  if (blocks > 1024) blocks = 1024;
  MDS_LAUNCH(pwg_pack_kernel, dim3(blocks), dim3(256), 0, stream, *a);
  return mds_check_launch("pwg_pack");
}
