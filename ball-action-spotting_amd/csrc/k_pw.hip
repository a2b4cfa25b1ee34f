// k_pw.hip — 1x1 convolutions as MFMA GEMMs over channels-last rows.
//
//   mds_pw_fwd   : y[M][N] = pro(x)[M][K] * w[N][K]^T (+residual) (+per-channel sum/sumsq of y)
//   mds_pw_wgrad : dw[N][K] += dy[M][N]^T * pro(x)[M][K]
//
// Roofline: these GEMMs are skinny (K, N <= 1152, M up to 4.7 M rows): bytes/row = 2(K+N),
// flops/row = 2KN -> 25..165 FLOP/B, below the 312 FLOP/B ridge of MI355X, i.e. HBM-bound IF the
// MFMA side sustains ~1 PFLOP/s; with K this short the kernel is really latency/issue bound, so:
//   * 128x128 block tile, 2x2 waves of 64x64 (16 accumulator fragments, 8 LDS fragment reads per
//     16 MFMAs), K staged 256 bytes per row at a time (128 bf16 / 64 fp32) -> 4 k-steps per barrier;
//   * every global load of a K-chunk is issued at once into raw registers, and the NEXT chunk's
//     loads are in flight while the current chunk's MFMAs run (register software pipeline);
//   * LDS row pitch 160 B (32 B x odd): conflict-free for the lane groups of ds_read_b128.
#include <stdlib.h>
#include "gemm.h"

#define PW_BM 128
#ifndef MDS_PW_GPRE3
#define MDS_PW_GPRE3 1   /* the gate row of a staged vector travels with it in the BN + SiLU + gate prologue too (64-row tiles) */
#endif
#define PW_BN 128
#define PW_SP 72   /* output staging pitch (elements): 64 columns + 16 bytes */

template <typename T> struct PwCfg;
// LDS row pitch = 160 B: with ds_read_b128's real lane groups ({0-3,12-15,20-27}, ...) a pitch of
// 32 B x odd is the conflict-free family; the former 144 B cost 28 of 64 lanes a replay.
template <> struct PwCfg<bf16_t> { static const int KC = 64, LD = 80; };
template <> struct PwCfg<float> { static const int KC = 32, LD = 40; };

// WN = waves along N: 2 -> 128x128 tile (2x2 waves of 64x64), two blocks per CU;
//                      1 -> 128x64 tile (4x1 waves of 32x64), ~160 VGPRs, three blocks per CU.
// BM = rows per tile: 128, or 64 for the small-M layers (stage 5 / 3D: 18400 rows are only 144 tiles of 128 —
//      fewer blocks than CUs); WN = 2 only.
// TAIL: 0 = plain epilogue, 1 = POST (BatchNorm-backward sums of the NEXT layer over the output tile, mds_poststat_t),
//       2 = EPI (mds_epi_t).
// (Measured and removed, DESIGN 5: the x operand formed on load as dy = A*g + B*y + D, a second operand pair + bias row for the
//  linear form of BatchNorm backward, two K chunks in flight for the K-heavy layers of the TRAINING step.)
// DEEP (fp32 inference plans, 64-row tiles, NONE / GATE prologue): TWO K chunks in flight.  A one-image launch has fewer blocks than
//   CUs, so nothing hides a chunk's load: the K loop ran at one memory round trip per 32-channel chunk (~1.4 us; 36 of them in a
//   1152 -> 192 projection).  Two register sets, every load issued unconditionally from a clamped address (zeroed in registers
//   when staged) so that no branch sits between a refill and its wait; an odd chunk count runs one all-zero phantom chunk.
// SPLIT (inference plans, small M): grid.z blocks share an output tile, each over its own K range; partial tiles go to
//   split_part[z][M][N] (fp32), the last block to finish the tile (ticket) adds them in z order and runs the epilogue.
template <typename T, int PRO, int WN, int BM, int TAIL, bool SPLIT = false, bool DEEP = false>
__global__ __launch_bounds__(256, (WN == 2 && BM == 128) || TAIL == 1 || SPLIT || DEEP ? 2 : 3) void pw_fwd_kernel(mds_pw_fwd_args a) {
  MDS_CHAIN_PRIO();
  static_assert(!DEEP || (TAIL != 1 && (PRO == MDS_PRO_NONE || PRO == MDS_PRO_GATE) && WN == 2 && BM == 64), "DEEP variants");
  static_assert(!SPLIT || (TAIL != 1 && (PRO == MDS_PRO_NONE || PRO == MDS_PRO_GATE) && WN == 2 && BM == 64), "SPLIT variants");
  constexpr bool POST = TAIL == 1, EPI = TAIL == 2;
  typedef typename Frag<T>::type frag_t;
  typedef Mma<T, EPI && sizeof(T) == 4 && MDS_EVAL_X3> MM;   // inference plans in fp32: split-bf16 products (platform.h)
  constexpr int KC = PwCfg<T>::KC, LD = PwCfg<T>::LD, VPR = KC / 8, RPP = 256 / VPR, NL = BM / RPP;
  constexpr int BN = 64 * WN, MFW = (WN == 2 ? BM / 32 : BM / 64), NLW = BN / RPP;   // tile columns, m-fragments per wave, filter rows per thread
  MDS_DYN_SMEM(smem);
  T* xs = (T*)smem;                          // [BM][LD]
  T* ws = xs + BM * LD;                      // [BN][LD]
  // bf16 output staging, a private [16][PW_SP] region per wave: in the MFMA layout a lane holds 4 columns (8 bytes) of
  // 16 different rows, and stores issued that way write 32-byte pieces (~2.9 TB/s of output on every GEMM-shaped kernel
  // here, fill reaches 6.9); through LDS a lane stores 16 bytes of a 128-byte row segment (k_pwr.hip measured it first)
  constexpr bool STG = sizeof(T) == 2;
  T* stg = ws + BN * LD + (threadIdx.x >> 6) * 16 * PW_SP;

  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);   // scalar: "nf < nfr" must be a scalar branch, not an exec-mask dance per fragment
  const int i = lane & 15, q = lane >> 4;
  const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
  const long m0 = (long)blockIdx.x * BM;
  const int N = a.N;
  const int K = a.K;
  // SPLIT: this block's K range [KB, KE) (whole chunks); K stays the row pitch
  const int kper = SPLIT ? ((K + (int)gridDim.z * PwCfg<T>::KC - 1) / ((int)gridDim.z * PwCfg<T>::KC)) * PwCfg<T>::KC : K;
  const int KB = SPLIT ? (int)blockIdx.z * kper : 0;
  const int KE = SPLIT ? (KB + kper < K ? KB + kper : K) : K;
  const T* x = (const T*)a.x;
  const T* w = (const T*)a.w;
  T* y = (T*)a.y;
  const int svec = tid % VPR, srow = tid / VPR;  // staging: this thread's k-offset and first row
  int grow[NL];  // squeeze-excite gate / DropPath-mask row of each staged row (one division per row per block)
  if (PRO == MDS_PRO_BN_SILU_GATE || PRO == MDS_PRO_GATE) {
    const long rpg = a.pro.rows_per_group;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const long m = m0 + srow + RPP * l;
      grow[l] = (int)((m < a.M ? m : a.M - 1) / rpg);
    }
  }

  const T* xrow[NL];   // this thread's rows of the tile
  bool xok[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const long m = m0 + srow + RPP * l;
    xok[l] = m < a.M;
    xrow[l] = x + (xok[l] ? m : 0) * K + 8 * svec;
  }
  for (int n0 = blockIdx.y * BN; n0 < N; n0 += gridDim.y * BN) {
    int nfr = (N - n0 - 64 * wn) >> 4;  // valid 16-column fragments of this wave
    nfr = nfr < 0 ? 0 : (nfr > 4 ? 4 : nfr);
    f32x4 acc[MFW][4];
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf)
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // POST: the next BatchNorm's table entries and this lane's post.y fragments are requested HERE, ahead of the K loop
    // (they were three exposed memory round trips in the epilogue: the table, then one per fragment row)
    RawV4<T> rys[POST ? MFW : 1][4];
    float pb[4] = {0.f, 0.f, 0.f, 0.f};
    if (EPI && tid < BN && n0 + tid < N) { pb[0] = a.epi.scale[n0 + tid]; pb[1] = a.epi.shift[n0 + tid]; }   // same for the output transform's table
    if (POST) {
      if (tid < BN && n0 + tid < N) {
#pragma unroll
        for (int k = 0; k < 4; ++k) pb[k] = a.post.bn[(long)k * N + n0 + tid];
      }
      const int nfr_ = (N - n0 - 64 * wn) >> 4;
#pragma unroll
      for (int mf = 0; mf < (POST ? MFW : 1); ++mf) {
        const long m = m0 + 16 * MFW * wm + 16 * mf + i;
        const T* ysrow = (const T*)a.post.y + (m < a.M ? m : 0) * N + n0 + (nfr_ > 0 ? 64 * wn : 0) + 4 * q;   // clamped: finite values
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) rys[mf][nf].ld(ysrow + (nf < nfr_ ? 16 * nf : 0));
      }
    }

    constexpr int NS = DEEP ? 2 : 1;
    RawV8<T> rx[NS][NL], rw[NS][NLW];
    // the squeeze-excite gate row of every staged x vector travels WITH it (same issue point): loaded inside the staging
    // loop it was one exposed L2 round trip per K chunk - 18 of them in the 1152 -> 192 projections
    // (the variants that would spill with 8 more registers per row keep the in-loop load: 128-row tiles, BN_SILU_GATE)
    constexpr bool GATED = PRO == MDS_PRO_BN_SILU_GATE || PRO == MDS_PRO_GATE;
    constexpr bool GPRE = (PRO == MDS_PRO_GATE || (PRO == MDS_PRO_BN_SILU_GATE && MDS_PW_GPRE3)) && NL <= 2;
    float rg[NS][GPRE ? NL : 1][8];
    const T* wrow[NLW];   // this thread's filter rows of the n-tile (row pointers hoisted out of the k-loop)
    bool wok[NLW];
#pragma unroll
    for (int l = 0; l < NLW; ++l) {
      const int n = n0 + srow + RPP * l;
      wok[l] = n < N;
      wrow[l] = w + (long)(wok[l] ? n : 0) * K + 8 * svec;
    }
    // the prologue's scale / shift of a chunk's 8 channels travel with the chunk's loads too (K-heavy projections with the BN + SiLU +
    // gate prologue: they were an exposed round trip per K chunk after the barrier)
    constexpr bool SPRE = PRO == MDS_PRO_BN_SILU_GATE && MDS_PW_GPRE3 && NL <= 2;
    float psc[SPRE ? 8 : 1], psh[SPRE ? 8 : 1];
    auto issue = [&](int kc, RawV8<T> (&tx)[NL], RawV8<T> (&tw)[NLW], float (&tg)[GPRE ? NL : 1][8]) {  // all global loads of one K-chunk
      const bool kok = kc + 8 * svec < KE;
      if (DEEP) {   // straight-line: rows are clamped in xrow / wrow, channels past the range read channel 0; zeroed when staged
        const int ko = kok ? kc : -8 * svec;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          tx[l].ld(xrow[l] + ko);
          if (GPRE) load8f(a.pro.gate + (long)grow[l] * K + (kok ? kc + 8 * svec : 0), tg[l]);
        }
#pragma unroll
        for (int l = 0; l < NLW; ++l) tw[l].ld(wrow[l] + ko);
        return;
      }
      if (SPRE) {
        const int kp = kok ? kc + 8 * svec : 0;
        load8f(a.pro.scale + kp, (float (&)[8])psc); load8f(a.pro.shift + kp, (float (&)[8])psh);
      }
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        if (xok[l] && kok) tx[l].ld(xrow[l] + kc); else tx[l].zero();
        if (GPRE) load8f(a.pro.gate + (long)grow[l] * K + (kok ? kc + 8 * svec : 0), tg[l]);   // grow is clamped: always legal
      }
#pragma unroll
      for (int l = 0; l < NLW; ++l) {
        if (wok[l] && kok) tw[l].ld(wrow[l] + kc); else tw[l].zero();
      }
    };
    auto chunk = [&](int kc, RawV8<T> (&tx)[NL], RawV8<T> (&tw)[NLW], float (&tg)[GPRE ? NL : 1][8]) {
      const int kk = kc + 8 * svec;
      const bool kin = kk < KE;
      __syncthreads();  // previous chunk's fragment reads are done
      if (PRO == MDS_PRO_NONE) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          if (DEEP && !(xok[l] && kin)) tx[l].zero();
          tx[l].st(xs + (srow + RPP * l) * LD + 8 * svec);
        }
      } else {
        float sc[8], sh[8];
        if (SPRE) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { sc[j] = psc[SPRE ? j : 0]; sh[j] = psh[SPRE ? j : 0]; }
        } else if (PRO != MDS_PRO_GATE && kin) { load8f(a.pro.scale + kk, sc); load8f(a.pro.shift + kk, sh); }
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          const int r = srow + RPP * l;
          const long m = m0 + r;
          float v[8];
          tx[l].get(v);
          if (m < a.M && kin) {
            if (PRO != MDS_PRO_GATE) {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                float z = v[j] * sc[j] + sh[j];
                v[j] = (PRO == MDS_PRO_AFFINE) ? z : siluf_(z);
              }
            }
            if (GPRE) {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] *= tg[l][j];
            } else if (GATED) {
              float g[8];
              load8f(a.pro.gate + (long)grow[l] * K + kk, g);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] *= g[j];
            }
          } else if (DEEP) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
          }
          store8(xs + r * LD + 8 * svec, v);
        }
      }
#pragma unroll
      for (int l = 0; l < NLW; ++l) {
        if (DEEP && !(wok[l] && kin)) tw[l].zero();
        tw[l].st(ws + (srow + RPP * l) * LD + 8 * svec);
      }
      __syncthreads();
      if (DEEP) issue(kc + 2 * KC, tx, tw, tg);            // refill this set (past the range: clamped reads, never staged as data)
      else if (kc + KC < KE) issue(kc + KC, tx, tw, tg);    // in flight while the MFMAs below run
      const int ksteps = (KE - kc >= KC) ? KC / 32 : ((KE - kc + 31) >> 5);
#pragma unroll
      for (int ks = 0; ks < KC / 32; ++ks) {
        if (DEEP || ks < ksteps) {   // DEEP: no branch between a refill and its wait (channels past the range are staged as zeros)
          typename MM::frag xf[MFW];
#pragma unroll
          for (int mf = 0; mf < MFW; ++mf) xf[mf] = MM::prep(ld_frag(xs + (16 * MFW * wm + 16 * mf + i) * LD + 32 * ks + 8 * q));
#pragma unroll
          for (int nf = 0; nf < 4; ++nf) {   // no "nf < nfr" test here: filter rows past N are staged as zeros,
            const typename MM::frag wf = MM::prep(ld_frag(ws + (64 * wn + 16 * nf + i) * LD + 32 * ks + 8 * q));   // and a branch-free k-loop schedules better
#pragma unroll
            for (int mf = 0; mf < MFW; ++mf) MM::mma(wf, xf[mf], acc[mf][nf]);  // acc[r] = y[m = i][n = 4q + r]
          }
        }
      }
    };
    if (!SPLIT || KB < KE) issue(KB, rx[0], rw[0], rg[0]);
    if (DEEP) {
      issue(KB + KC, rx[NS - 1], rw[NS - 1], rg[NS - 1]);
      for (int kc = KB; kc < KE; kc += 2 * KC) {
        chunk(kc, rx[0], rw[0], rg[0]);
        chunk(kc + KC, rx[NS - 1], rw[NS - 1], rg[NS - 1]);
      }
    } else {
      for (int kc = KB; kc < KE; kc += KC) chunk(kc, rx[0], rw[0], rg[0]);
    }
    if (SPLIT) {
      // partial tile -> split_part[z]; ticket; the last block of the tile sums the partials in z order (its own included:
      // the order is fixed, so the result does not depend on which block came last) and goes on to the epilogue
      const long MN = a.M * (long)N;
#pragma unroll
      for (int mf = 0; mf < MFW; ++mf) {
        const long m = m0 + 16 * MFW * wm + 16 * mf + i;
        if (m < a.M) {
          float* prow = a.split_part + (long)blockIdx.z * MN + m * N + n0 + 64 * wn + 4 * q;
#pragma unroll
          for (int nf = 0; nf < 4; ++nf)
            if (nf < nfr) st_coherent4(prow + 16 * nf, acc[mf][nf]);
        }
      }
      int* lastp = (int*)smem;    // (the operand tiles are dead: every wave passed the K loop's last barrier)
      mds_wait_stores();          // every lane waits for the acknowledgements of its device-scope stores: EXPLICIT vmcnt(0) - the
                                  // memory model does not oblige the compiler to put one in front of a workgroup barrier
      __syncthreads();
      if (tid == 0) {
        int* tk = a.split_ticket + ((long)blockIdx.x * ((N + BN - 1) / BN) + n0 / BN) * MDS_PW_SPLIT_TICKET_STRIDE;
        const int t = atomicAdd(tk, 1);
        const int last = t == (int)gridDim.z - 1;
        if (last) atomicExch(tk, 0);   // ready for the next launch
        *lastp = last;
      }
      __syncthreads();
      const bool last = *lastp != 0;
      __syncthreads();            // (smem is reused by the epilogue tables)
      if (!last) continue;
#pragma unroll
      for (int mf = 0; mf < MFW; ++mf)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // four partial tiles are requested at a time (one memory round trip per four splits, not one per split and fragment row)
      for (int z0 = 0; z0 < (int)gridDim.z; z0 += 4) {
        f32x4 t[4][MFW][4];
#pragma unroll
        for (int zz = 0; zz < 4; ++zz)
#pragma unroll
          for (int mf = 0; mf < MFW; ++mf) {
            const long m = m0 + 16 * MFW * wm + 16 * mf + i;
            const bool ok = m < a.M && z0 + zz < (int)gridDim.z;
            const float* prow = a.split_part + (long)(z0 + zz) * MN + m * N + n0 + 64 * wn + 4 * q;
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) t[zz][mf][nf] = (ok && nf < nfr) ? ld_coherent4(prow + 16 * nf) : (f32x4){0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
        for (int zz = 0; zz < 4; ++zz)      // z order: the sum does not depend on which block came last
#pragma unroll
          for (int mf = 0; mf < MFW; ++mf)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) acc[mf][nf] += t[zz][mf][nf];
      }
    }

    // ---- epilogue: residual, store, statistics.  One row-validity test per m-fragment (not per
    //      store): the per-lane `m < M` guard around every 8-byte store used to cost an exec-mask
    //      save/restore and a branch per fragment — 1/4 of the epilogue's instructions.
    float ps[16], pss[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { ps[e] = 0.f; pss[e] = 0.f; }
    float* pbn = (float*)smem;   // POST: [4][BN] scale, shift, mean, rstd of the tile's columns (over the finished x/w tiles)
    if (POST) {
      __syncthreads();           // every wave is done with the fragment reads of the last chunk
      if (tid < BN) {
#pragma unroll
        for (int k = 0; k < 4; ++k) pbn[k * BN + tid] = pb[k];
      }
      __syncthreads();
    }
    if (EPI) {                   // scale / shift of the tile's columns
      __syncthreads();
      if (tid < BN) { pbn[tid] = pb[0]; pbn[BN + tid] = pb[1]; }
      __syncthreads();
    }
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf) {
      const long m = m0 + 16 * MFW * wm + 16 * mf + i;
      const bool ok = m < a.M;
      T* yrow = y + (ok ? m : 0) * N + n0 + 64 * wn + 4 * q;
      float v[4][4];
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[nf][r] = acc[mf][nf][r];
      if (EPI) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
          const float* pc = pbn + 64 * wn + 16 * nf + 4 * q;
          const f32x4 sc = *(const f32x4*)pc, sh = *(const f32x4*)(pc + BN);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z = v[nf][r] * sc[r] + sh[r];
            v[nf][r] = a.epi.mode == MDS_EPI_BN_SILU ? siluf_(z) : z;
          }
        }
      }
      if (a.residual && ok) {
        const T* rrow = (const T*)a.residual + m * N + n0 + 64 * wn + 4 * q;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
          if (nf < nfr) {
            float rr[4];
            load4(rrow + 16 * nf, rr);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[nf][r] += rr[r];
          }
        }
      }
      if (POST) {
        // u of the next BatchNorm backward is this tile: g, sum g, sum g*xhat (rows past M: v == 0 -> g == 0)
        const float mk = (a.post.mode == MDS_POST_MASK) ? a.post.mask[(unsigned)(ok ? m : 0) / (unsigned)a.post.rows_per_group] : 1.0f;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
          if (nf < nfr) {
            const float ys[4] = {rys[mf][nf].get(0), rys[mf][nf].get(1), rys[mf][nf].get(2), rys[mf][nf].get(3)};
            const float* pc = pbn + 64 * wn + 16 * nf + 4 * q;
            const f32x4 mu = *(const f32x4*)(pc + 2 * BN), rs = *(const f32x4*)(pc + 3 * BN);
            if (a.post.mode == MDS_POST_SILU) {
              const f32x4 sc = *(const f32x4*)pc, sh = *(const f32x4*)(pc + BN);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[nf][r] *= silu_gradf_(ys[r] * sc[r] + sh[r]);   // g replaces u in memory
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float g = Elem<T>::rnd(v[nf][r]) * mk;      // the sums see what later readers will read
              ps[nf * 4 + r] += g;
              pss[nf * 4 + r] += g * ((ys[r] - mu[r]) * rs[r]);
            }
          }
        }
      }
      if (STG) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) store4(stg + i * PW_SP + 16 * nf + 4 * q, v[nf]);
        wave_lds_sync();
        T* ybase = y + (m0 + 16 * MFW * wm + 16 * mf) * N + n0 + 64 * wn;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int vv = lane + 64 * t, row = vv >> 3, c = vv & 7;
          if (m0 + 16 * MFW * wm + 16 * mf + row < a.M && (c >> 1) < nfr) {
            RawV8<T> o;
            o.ld(stg + row * PW_SP + 8 * c);
            o.st(ybase + (long)row * N + 8 * c);
          }
        }
        wave_lds_sync();   // the next fragment row's staging writes come after these reads
      } else if (ok) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
          if (nf < nfr) store4(yrow + 16 * nf, v[nf]);
      }
      if (!POST) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)   // rows past M hold zeros (zero-filled operand): no guard needed
#pragma unroll
          for (int r = 0; r < 4; ++r) { ps[nf * 4 + r] += v[nf][r]; pss[nf * 4 + r] += v[nf][r] * v[nf][r]; }
      }
    }
    if (POST ? (a.post.stats != nullptr) : (a.stats != nullptr)) {
      // after the reduce-scatter every lane of the wave holds ONE column's partial sums: add them to
      // the slot straight away (no LDS staging, no block barrier at the end of every tile)
      const int e = reduce_scatter16(ps, i);
      reduce_scatter16(pss, i);
      const int n = n0 + 64 * wn + 16 * (e >> 2) + 4 * q + (e & 3);
      if (n < N) {
        const long so = (long)((blockIdx.x + wm) % MDS_STAT_SLOTS) * 2 * N;
        if (POST) {   // backward sums: fp64 slots (include/mds.h)
          atomicAdd(a.post.stats + so + n, (double)ps[0]);
          atomicAdd(a.post.stats + so + N + n, (double)pss[0]);
        } else {
          atomicAdd(a.stats + so + n, (double)ps[0]);
          atomicAdd(a.stats + so + N + n, (double)pss[0]);
        }
      }
    }
  }
}

int pw_fwd_wres_try(const mds_pw_fwd_args* a, mds_stream_t stream);   // k_pwr.hip: short-K wide-N layers; 1 = not taken
int pw_fwd_k_try(const mds_pw_fwd_args* a, mds_stream_t stream);      // k_pwk8.hip: K-streaming narrow-N layers (bf16); 1 = not taken

int c3_pw_try(const mds_pw_fwd_args* a, mds_stream_t stream);      // k_c3.hip

extern "C" int mds_pw_fwd(const mds_pw_fwd_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->M > 0 && a->K > 0 && a->N > 0, "pw_fwd: bad dims");
  MDS_REQUIRE(a->K % 8 == 0 && a->N % 16 == 0, "pw_fwd: K=%d must be a multiple of 8, N=%d of 16", a->K, a->N);
  MDS_REQUIRE(a->x && a->w && a->y, "pw_fwd: null pointer");
  MDS_REQUIRE(a->pro.mode == MDS_PRO_NONE || a->pro.mode == MDS_PRO_GATE || (a->pro.scale && a->pro.shift), "pw_fwd: prologue needs scale/shift");
  MDS_REQUIRE((a->pro.mode != MDS_PRO_BN_SILU_GATE && a->pro.mode != MDS_PRO_GATE) || (a->pro.gate && a->pro.rows_per_group > 0), "pw_fwd: gate prologue");
  const bool post = a->post.mode != MDS_POST_NONE;
  const bool epi = a->epi.mode != MDS_EPI_NONE;
  if (epi) {
    MDS_REQUIRE(a->epi.scale && a->epi.shift && !a->stats && !post, "pw_fwd: an output transform needs scale/shift and excludes statistics and post statistics");
    MDS_REQUIRE(a->pro.mode == MDS_PRO_NONE || a->pro.mode == MDS_PRO_GATE || a->pro.mode == MDS_PRO_BN_SILU, "pw_fwd: an output transform takes the NONE, GATE or BN_SILU prologue");
  }
  if (post) {
    MDS_REQUIRE(a->post.y && a->post.bn && a->post.stats && !a->stats, "pw_fwd: post statistics need y, bn, stats (and no forward stats)");
    MDS_REQUIRE(a->post.mode != MDS_POST_MASK || (a->post.mask && a->post.rows_per_group > 0), "pw_fwd: post mask");
    MDS_REQUIRE(a->pro.mode == MDS_PRO_NONE && a->M < 4294967295L, "pw_fwd: post statistics are a data-gradient feature (no forward prologue)");
  }
  const bool split = a->split > 1;
  if (split) {
    MDS_REQUIRE(a->split <= MDS_PW_MAX_SPLIT && a->split_part && a->split_ticket, "pw_fwd: split-K needs split <= %d, a partial buffer and tickets", MDS_PW_MAX_SPLIT);
    MDS_REQUIRE(!post && !a->stats && (a->pro.mode == MDS_PRO_NONE || a->pro.mode == MDS_PRO_GATE),
                "pw_fwd: split-K is an inference-plan feature (NONE / GATE prologue, no statistics, no post statistics)");
  }
  if (!split && c3_pw_try(a, stream)) return mds_check_launch("pw_fwd");      // k_c3.hip: the large prologue-free bf16 launches
  if (!split) { const int rc = pw_fwd_k_try(a, stream); if (rc <= 0) return rc; }
  if (!split) { const int rc = pw_fwd_wres_try(a, stream); if (rc <= 0) return rc; }
  // 128x64 tiles for the narrow projections (N <= 64: half of a 128-column tile would be padding;
  // 154 -> 119 us at 1.18 M x 128 -> 32); wider N measured 5-20 % slower with them despite 3 blocks/CU
  const int wn = (a->N <= 64 && !split) ? 1 : 2;
  const int BN = 64 * wn;
  // 64-row tiles below 400 k rows: twice the blocks for the stage-3..5 / 3D layers (isolated: -10...25 %;
  // inside the step, where the weight-gradient stream fills the idle CUs anyway, +1 %)
  const long bar64 = mds_knob(MDS_KNOB_PW_BM64) > 0 ? 1000L * mds_knob(MDS_KNOB_PW_BM64) : 400000;
  const int bm = (wn == 2 && a->M <= bar64) ? 64 : PW_BM;
  MDS_REQUIRE(!split || bm == 64, "pw_fwd: split-K is for small M (64-row tiles)");
  const int mt = cdiv(a->M, bm), nt = cdiv(a->N, BN);
  int gy = 1;
  if (nt > 1) { gy = cdiv(mds_knob(MDS_KNOB_PW_GY) > 0 ? mds_knob(MDS_KNOB_PW_GY) : 1536, mt); if (gy > nt) gy = nt; if (gy < 1) gy = 1; }
  dim3 grid(mt, gy, split ? a->split : 1), block(256);
  // two K chunks in flight (DEEP) for the fp32 inference launches that leave CUs without a second block to hide a chunk's load:
  // 64-row tiles, an output transform or split-K (inference plans only), >= 3 chunks in a block's K range, at most one block per
  // CU (measured on one 736 x 1280 frame: the 120 ... 232-block split projections -0.6 ... -2.9 us per launch, the 348 / 360-block
  // expansions +0.3 ... +0.9 us - two blocks on a CU already hide each other's loads).  MDS_KNOBS=17=1: off
  const int kchunks = cdiv(split ? cdiv(a->K, a->split) : a->K, PwCfg<float>::KC);
  const bool deep = a->dtype == MDS_F32 && (epi || split) && wn == 2 && bm == 64 && kchunks >= 3 &&
                    (long)mt * gy * (split ? a->split : 1) <= 256 && mds_knob(MDS_KNOB_PW_DEEP) != 1;
#define PW_GOSPLIT(T, PRO, TAIL_) \
  do { const size_t smem = (size_t)(bm + BN) * PwCfg<T>::LD * sizeof(T) + (sizeof(T) == 2 ? 4 * 16 * PW_SP * 2 : 0); \
       if (sizeof(T) == 4 && deep) MDS_LAUNCH((pw_fwd_kernel<T, PRO, 2, 64, TAIL_, true, sizeof(T) == 4>), grid, block, smem, stream, *a); \
       else MDS_LAUNCH((pw_fwd_kernel<T, PRO, 2, 64, TAIL_, true>), grid, block, smem, stream, *a); } while (0)
#define PW_GODEEP(T, PRO) \
  do { const size_t smem = (size_t)(bm + BN) * PwCfg<T>::LD * sizeof(T) + (sizeof(T) == 2 ? 4 * 16 * PW_SP * 2 : 0); \
       MDS_LAUNCH((pw_fwd_kernel<T, PRO, 2, 64, 2, false, sizeof(T) == 4>), grid, block, smem, stream, *a); } while (0)
#define PW_GO2(T, PRO, TAIL_) \
  do { const size_t smem = (size_t)(bm + BN) * PwCfg<T>::LD * sizeof(T) + (sizeof(T) == 2 ? 4 * 16 * PW_SP * 2 : 0); \
       if (wn == 2 && bm == 64) MDS_LAUNCH((pw_fwd_kernel<T, PRO, 2, 64, TAIL_>), grid, block, smem, stream, *a); \
       else if (wn == 2) MDS_LAUNCH((pw_fwd_kernel<T, PRO, 2, 128, TAIL_>), grid, block, smem, stream, *a); \
       else MDS_LAUNCH((pw_fwd_kernel<T, PRO, 1, 128, TAIL_>), grid, block, smem, stream, *a); } while (0)
#define PW_GO(T, PRO) PW_GO2(T, PRO, 0)
  MDS_DISPATCH_DTYPE(a->dtype, T, {
    if (split) {
      if (epi) { if (a->pro.mode == MDS_PRO_GATE) PW_GOSPLIT(T, MDS_PRO_GATE, 2); else PW_GOSPLIT(T, MDS_PRO_NONE, 2); }
      else { if (a->pro.mode == MDS_PRO_GATE) PW_GOSPLIT(T, MDS_PRO_GATE, 0); else PW_GOSPLIT(T, MDS_PRO_NONE, 0); }
    }
    else if (post) PW_GO2(T, MDS_PRO_NONE, 1);
    else if (epi && deep && sizeof(T) == 4 && (a->pro.mode == MDS_PRO_GATE || a->pro.mode == MDS_PRO_NONE)) {
      if (a->pro.mode == MDS_PRO_GATE) PW_GODEEP(T, MDS_PRO_GATE); else PW_GODEEP(T, MDS_PRO_NONE);
    }
    else if (epi) { if (a->pro.mode == MDS_PRO_GATE) PW_GO2(T, MDS_PRO_GATE, 2); else if (a->pro.mode == MDS_PRO_BN_SILU) PW_GO2(T, MDS_PRO_BN_SILU, 2); else PW_GO2(T, MDS_PRO_NONE, 2); }
    else switch (a->pro.mode) {
      case MDS_PRO_NONE: PW_GO(T, MDS_PRO_NONE); break;
      case MDS_PRO_AFFINE: PW_GO(T, MDS_PRO_AFFINE); break;
      case MDS_PRO_BN_SILU: PW_GO(T, MDS_PRO_BN_SILU); break;
      case MDS_PRO_BN_SILU_GATE: PW_GO(T, MDS_PRO_BN_SILU_GATE); break;
      case MDS_PRO_GATE: PW_GO(T, MDS_PRO_GATE); break;
      default: mds_set_error("pw_fwd: prologue mode %d", a->pro.mode); return MDS_ERR_BAD_ARG;
    }
  });
#undef PW_GO
#undef PW_GO2
#undef PW_GOSPLIT
#undef PW_GODEEP
  return mds_check_launch("pw_fwd");
}

// split-K factor for a shape.  Measured on the fp32 inference shapes of one 736 x 1280 frame (tools/probes/split_bench.py, us):
//   920 x 1152 -> 192 (30 tiles, 36 chunks): 56 unsplit, 38 / 32 / 30 / 30 / 32 / 40 at split 2 / 3 / 4 / 6 / 8 / 12
//   3680 x 672 -> 112 (58 tiles, 21 chunks): 34 unsplit, 27 / 24 / 24 / 34 / 30 / 36
//   4600 x 576 -> 192 (144 tiles): 31 unsplit, 31-37 split - no gain;  920 x 192 -> 1152 (6 chunks): 13 unsplit, 24+ split - a loss.
// What bounds the split form is the last block's pass over the partial tiles (~10 us at 6 splits) and ~6 us of ticket + barrier, so:
// only launches of < 100 tiles that walk >= 16 chunks, four splits.  MDS_KNOB_PW_SPLIT: 1 = never, n >= 2 = that factor instead of 4.
extern "C" int mds_pw_fwd_split(long M, int K, int N, int dtype) {
  const int knob = mds_knob(MDS_KNOB_PW_SPLIT);
  if (knob == 1 || M <= 0 || K <= 0 || N <= 0) return 1;
  const int kc = dtype == MDS_BF16 ? PwCfg<bf16_t>::KC : PwCfg<float>::KC;
  const long chunks = cdiv(K, kc), tiles = cdiv(M, MDS_PW_SPLIT_TILE_ROWS) * cdiv(N, 128);
  if (tiles >= 100 || chunks < 16) return 1;
  long s = knob >= 2 ? knob : 4;
  if (s > chunks / 3) s = chunks / 3;
  if (s > MDS_PW_MAX_SPLIT) s = MDS_PW_MAX_SPLIT;
  return s < 2 ? 1 : (int)s;
}

// ------------------------------------------------------------------------------------ wgrad
// dw[n][k] += sum_m dy[m][n] * pro(x)[m][k]: the MFMA reduction index is the row m, so both operands
// are needed "transposed" (8 consecutive m per lane).  The staging pass transposes while writing to
// LDS: element (channel c, row m) lives at c*LDT + 8*((m>>3) ^ ((c>>3)&7)) + (m&7) — bf16 rows are
// written as (m, m+1) pairs in one dword, the XOR spreads the 8 channel-chunks of a wave over
// distinct banks, and every fragment is ONE aligned 16-byte LDS read (was 8 two-byte reads, which
// made this kernel LDS-issue bound).
#define WG_ROWS 64  // rows staged per step (2 MFMA k-steps of 32 rows)
// Output tile = (64*NF output channels) x (16*KF input channels); wave w owns n-fragments
// {NF*w .. NF*w+NF-1} x all KF k-fragments (see the launcher for the measured choice of NF, KF).
template <typename T> struct WgCfg;
template <> struct WgCfg<bf16_t> { static const int MW = 2, LDT = WG_ROWS + 8; };
template <> struct WgCfg<float> { static const int MW = 1, LDT = WG_ROWS + 4; };
template <int LDT> MDS_DEV int wg_off(int c, int m) { return c * LDT + 8 * ((m >> 3) ^ ((c >> 3) & 7)) + (m & 7); }
MDS_DEV void wg_put(bf16_t* base, int off, float v0, float v1) {
  *(uint32_t*)(base + off) = pack2(v0, v1);
}
MDS_DEV void wg_put(float* base, int off, float v0, float) { base[off] = v0; }

template <typename T, int PRO, int NF, int KF>
__global__ __launch_bounds__(256, 2) void pw_wgrad_kernel(mds_pw_wgrad_args a, int rows_per_block) {
  typedef typename Frag<T>::type frag_t;
  constexpr int MW = WgCfg<T>::MW, LDT = WgCfg<T>::LDT;
  constexpr int NT = 64 * NF, KT = 16 * KF, XCH = KT / 8, YCH = NT / 8;
  constexpr int XN = (WG_ROWS / MW) * XCH, YN = (WG_ROWS / MW) * YCH;   // staging items (row-words x chunks)
  constexpr int XI = (XN + 255) / 256, YI = (YN + 255) / 256;
  MDS_DYN_SMEM(smem);
  T* xsT = (T*)smem;          // [KT][LDT]  channel-major
  T* dsT = xsT + KT * LDT;    // [NT][LDT]
  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int K = a.K, N = a.N;
  const int ntiles_k = (K + KT - 1) / KT;
  const int n0 = (blockIdx.y / ntiles_k) * NT, kt0 = (blockIdx.y % ntiles_k) * KT;
  const long mbeg = (long)blockIdx.x * rows_per_block;
  long mend = mbeg + rows_per_block;
  if (mend > a.M) mend = a.M;
  const T* x = (const T*)a.x;
  const T* dy = (const T*)a.dy;
  const int kfr = (K - kt0 >= KT) ? KF : ((K - kt0 + 15) >> 4);

  // when 256 % XCH == 0 a thread keeps the same 8-channel slice for every staged item: its BN
  // scale/shift live in registers for the whole kernel
  constexpr bool FIXED_CH = (256 % XCH) == 0;
  float sc[8], sh[8];
  if (FIXED_CH && PRO != MDS_PRO_NONE && PRO != MDS_PRO_GATE) {
    const int kx = kt0 + 8 * (tid % XCH);
    if (kx < K) { load8f(a.pro.scale + kx, sc); load8f(a.pro.shift + kx, sh); }
  }

  f32x4 acc[NF][KF];
#pragma unroll
  for (int u = 0; u < NF; ++u)
#pragma unroll
    for (int v = 0; v < KF; ++v) acc[u][v] = (f32x4){0.f, 0.f, 0.f, 0.f};

  RawV8<T> rx[XI][MW], ry[YI][MW];
  auto issue = [&](long mb) {   // every global load of one 64-row step
#pragma unroll
    for (int p = 0; p < XI; ++p) {
      const int it = tid + 256 * p, kx = kt0 + 8 * (it % XCH);
#pragma unroll
      for (int h = 0; h < MW; ++h) {
        const long m = mb + (it / XCH) * MW + h;
        if (it < XN && m < mend && kx < K) rx[p][h].ld(x + m * K + kx); else rx[p][h].zero();
      }
    }
#pragma unroll
    for (int p = 0; p < YI; ++p) {
      const int it = tid + 256 * p, n = n0 + 8 * (it % YCH);
#pragma unroll
      for (int h = 0; h < MW; ++h) {
        const long m = mb + (it / YCH) * MW + h;
        if (it < YN && m < mend && n < N) ry[p][h].ld(dy + m * N + n); else ry[p][h].zero();
      }
    }
  };
  issue(mbeg);
  for (long mb = mbeg; mb < mend; mb += WG_ROWS) {
    __syncthreads();  // previous step's fragment reads are done
#pragma unroll
    for (int p = 0; p < XI; ++p) {
      const int it = tid + 256 * p, xc = it % XCH, kx = kt0 + 8 * xc;
      if (it < XN) {
        float v[MW][8];
#pragma unroll
        for (int h = 0; h < MW; ++h) {
          const long m = mb + (it / XCH) * MW + h;
          rx[p][h].get(v[h]);
          if (PRO != MDS_PRO_NONE && m < mend && kx < K) {
            if (PRO != MDS_PRO_GATE) {
              float scl[8], shl[8];
              if (!FIXED_CH) { load8f(a.pro.scale + kx, scl); load8f(a.pro.shift + kx, shl); }
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                float z = v[h][j] * (FIXED_CH ? sc[j] : scl[j]) + (FIXED_CH ? sh[j] : shl[j]);
                v[h][j] = (PRO == MDS_PRO_AFFINE) ? z : siluf_(z);
              }
            }
            if (PRO == MDS_PRO_BN_SILU_GATE || PRO == MDS_PRO_GATE) {
              float g[8];
              load8f(a.pro.gate + (long)((unsigned)m / (unsigned)a.pro.rows_per_group) * K + kx, g);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[h][j] *= g[j];
            }
          }
        }
        const int ml = (it / XCH) * MW;
#pragma unroll
        for (int j = 0; j < 8; ++j) wg_put(xsT, wg_off<LDT>(8 * xc + j, ml), v[0][j], v[MW - 1][j]);
      }
    }
#pragma unroll
    for (int p = 0; p < YI; ++p) {
      const int it = tid + 256 * p;
      if (it < YN) {
        float v[MW][8];
#pragma unroll
        for (int h = 0; h < MW; ++h) {
          ry[p][h].get(v[h]);
        }
        const int ml = (it / YCH) * MW, yc = it % YCH;
#pragma unroll
        for (int j = 0; j < 8; ++j) wg_put(dsT, wg_off<LDT>(8 * yc + j, ml), v[0][j], v[MW - 1][j]);
      }
    }
    __syncthreads();
    if (mb + WG_ROWS < mend) issue(mb + WG_ROWS);   // next step's loads fly under this step's MFMAs
#pragma unroll
    for (int ks = 0; ks < WG_ROWS / 32; ++ks) {
      const int g = 4 * ks + q;  // this lane's group of 8 rows (the MFMA k index)
      frag_t yf[NF];
#pragma unroll
      for (int u = 0; u < NF; ++u) {
        const int c = 16 * (NF * wave + u) + i;
        yf[u] = ld_frag(dsT + c * LDT + 8 * (g ^ ((c >> 3) & 7)));
      }
#pragma unroll
      for (int v = 0; v < KF; ++v) {
        if (v < kfr) {   // k-fragments past K hold zeros: skip them (uniform)
          const int c = 16 * v + i;
          const frag_t xf = ld_frag(xsT + c * LDT + 8 * (g ^ ((c >> 3) & 7)));
#pragma unroll
          for (int u = 0; u < NF; ++u) mma16(yf[u], xf, acc[u][v]);  // acc[r] = dw[n = 4q + r][k = i]
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NF; ++u)
#pragma unroll
    for (int v = 0; v < KF; ++v) {
      const int k = kt0 + 16 * v + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 16 * (NF * wave + u) + 4 * q + r;
        if (n < N && k < K) atomicAdd(a.dw + (long)n * K + k, acc[u][v][r]);
      }
    }
}

// bf16 variant on the gfx950 transposing LDS read: the staged images stay in the natural
// [row][channel] order (one 16-byte LDS store per loaded vector instead of eight transposing dword
// stores, and no bf16->fp32->bf16 round trip without a prologue); a fragment — 8 rows of one
// channel — is two ds_read_b64_tr_b16.  Row pitches are odd multiples of 32 bytes so the 8 rows a
// 32-lane LDS cycle touches sit in distinct bank groups.
// (Several four-wave groups per block - 1.3-2x faster alone, slower inside the step - were measured in round 3 and removed: DESIGN 5.)
template <int PRO, int NF, int KF>
__global__ __launch_bounds__(256, 2) void pw_wgrad_tr_kernel(mds_pw_wgrad_args a, int rows_per_block, int dbg) {
  typedef bf16_t T;
  constexpr int NT = 64 * NF, KT = 16 * KF, XCH = KT / 8, YCH = NT / 8;
  constexpr int LDX = KT + 16, LDY = NT + 16;           // elements; (KT+16)*2 B = odd * 32 B for KT % 32 == 0
  constexpr int XN = WG_ROWS * XCH, YN = WG_ROWS * YCH;  // staging items (rows x 8-channel chunks)
  constexpr int XI = (XN + 255) / 256, YI = (YN + 255) / 256;
  MDS_DYN_SMEM(smem);
  T* xs = (T*)smem;                                   // [WG_ROWS][LDX]
  T* ds = xs + WG_ROWS * LDX;                         // [WG_ROWS][LDY]
  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int K = a.K, N = a.N;
  const int ntiles_k = (K + KT - 1) / KT;
  // Block -> (row split, output tile), XCD-aware: consecutive block ids go to consecutive
  // XCDs, so row split s and ALL its output tiles are given to XCD s % 8 - the tiles of one split re-read the same x / dy rows,
  // which then come from that XCD's L2 once.  (Measured on 112 -> 672, 73 600 rows: 56 us when a split's 12 tiles were
  // spread over the XCDs, 41 us when they shared one.)
  const int ntiles = ntiles_k * ((N + NT - 1) / NT), slot = blockIdx.x >> 3;
  const int tile_id = slot % ntiles, split_id = (int)(blockIdx.x & 7) + 8 * (slot / ntiles);
  const int n0 = (tile_id / ntiles_k) * NT, kt0 = (tile_id % ntiles_k) * KT;
  const long mbeg = (long)split_id * rows_per_block;
  long mend = mbeg + rows_per_block;
  if (mend > a.M) mend = a.M;
  if (mbeg >= a.M) return;   // (row splits are rounded up to a multiple of 8; whole block, before any barrier)
  const T* x = (const T*)a.x;
  const T* dy = (const T*)a.dy;
  const int kfr = (K - kt0 >= KT) ? KF : ((K - kt0 + 15) >> 4);
  constexpr bool FIXED_CH = (256 % XCH) == 0;
  float sc[8], sh[8];
  if (FIXED_CH && PRO != MDS_PRO_NONE && PRO != MDS_PRO_GATE) {
    const int kx = kt0 + 8 * (tid % XCH);
    if (kx < K) { load8f(a.pro.scale + kx, sc); load8f(a.pro.shift + kx, sh); }
  }
  f32x4 acc[NF][KF];
#pragma unroll
  for (int u = 0; u < NF; ++u)
#pragma unroll
    for (int v = 0; v < KF; ++v) acc[u][v] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-thread staging constants: row within a step, channel, validity, running pointers (the 64-bit
  // address arithmetic per load and step was most of this loop's instruction count)
  RawV8<T> rx[XI], ry[YI];
  int xr[XI], yr[YI];
  bool xok[XI], yok[YI];
  const T* px[XI];
  const T* py[YI];
#pragma unroll
  for (int p = 0; p < XI; ++p) {
    const int it = tid + 256 * p, kx = kt0 + 8 * (it % XCH);
    xr[p] = it / XCH;
    xok[p] = it < XN && kx < K;
    px[p] = x + (mbeg + xr[p]) * K + (xok[p] ? kx : 0);
  }
#pragma unroll
  for (int p = 0; p < YI; ++p) {
    const int it = tid + 256 * p, n = n0 + 8 * (it % YCH);
    yr[p] = it / YCH;
    yok[p] = it < YN && n < N;
    py[p] = dy + (mbeg + yr[p]) * N + (yok[p] ? n : 0);
  }
  const long xstep = (long)WG_ROWS * K, ystep = (long)WG_ROWS * N;
  // the squeeze-excite gate row of every staged x item is requested WITH the item (same issue point): loaded inside the
  // staging loop it cost one exposed L2 round trip per 64-row step - with <= 2 blocks per CU nothing hides it, and it was
  // most of the gated projections' weight-gradient time (94 us per launch inside the step against 56 us ungated)
  constexpr bool GATED = PRO == MDS_PRO_BN_SILU_GATE || PRO == MDS_PRO_GATE;
  float rg[GATED ? XI : 1][8];
  auto issue = [&](long mb) {
    const int left = (int)(mend - mb);   // rows of this step that exist
#pragma unroll
    for (int p = 0; p < XI; ++p) {
      if (xok[p] && xr[p] < left) rx[p].ld(px[p]); else rx[p].zero();
      px[p] += xstep;
      if (GATED) {
        const long m = mb + xr[p];
        const int kx = kt0 + 8 * ((tid + 256 * p) % XCH);
        load8f(a.pro.gate + (long)((unsigned)(m < mend ? m : mbeg) / (unsigned)a.pro.rows_per_group) * K + (kx < K ? kx : 0), rg[p]);
      }
    }
#pragma unroll
    for (int p = 0; p < YI; ++p) {
      if (yok[p] && yr[p] < left) ry[p].ld(py[p]); else ry[p].zero();
      py[p] += ystep;
    }
  };
  issue(mbeg);
  // fragment rows of a 32-row k-step: 16-lane group q reads rows ra..ra+3 and ra+8..ra+11
  const int ra = 16 * (q >> 1) + 4 * (q & 1);
  const int lrow = (i >> 2), lcol = 4 * (i & 3);  // this lane's part of the 4x16 block it helps to gather
  for (long mb = mbeg; mb < mend; mb += WG_ROWS) {
    __syncthreads();  // previous step's fragment reads are done
#pragma unroll
    for (int p = 0; p < XI; ++p) {
      const int it = tid + 256 * p, xc = it % XCH, kx = kt0 + 8 * xc, ml = it / XCH;
      if (it < XN) {
        if (PRO == MDS_PRO_NONE) {
          rx[p].st(xs + ml * LDX + 8 * xc);
        } else {
          const long m = mb + ml;
          float v[8];
          rx[p].get(v);
          if (m < mend && kx < K) {
            if (PRO != MDS_PRO_GATE) {
              float scl[8], shl[8];
              if (!FIXED_CH) { load8f(a.pro.scale + kx, scl); load8f(a.pro.shift + kx, shl); }
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                float z = v[j] * (FIXED_CH ? sc[j] : scl[j]) + (FIXED_CH ? sh[j] : shl[j]);
                v[j] = (PRO == MDS_PRO_AFFINE) ? z : siluf_(z);
              }
            }
            if (GATED) {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] *= rg[p][j];
            }
          }
          store8(xs + ml * LDX + 8 * xc, v);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < YI; ++p) {
      const int it = tid + 256 * p;
      if (it < YN) ry[p].st(ds + (it / YCH) * LDY + 8 * (it % YCH));
    }
    __syncthreads();
    if (mb + WG_ROWS < mend && !(dbg & 4)) issue(mb + WG_ROWS);   // next step's loads fly under this step's MFMAs (rows past mend: zeros)
#pragma unroll
    for (int ks = 0; ks < WG_ROWS / 32; ++ks) {
      if (dbg & 2) break;
      const int r0 = 32 * ks + ra + lrow;
      u16x8 yf[NF];
#pragma unroll
      for (int u = 0; u < NF; ++u) {
        const T* pcol = ds + 16 * (NF * wave + u) + lcol;
        const u16x4 lo = lds_tr4(pcol + r0 * LDY), hi = lds_tr4(pcol + (r0 + 8) * LDY);
        yf[u] = (u16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int v = 0; v < KF; ++v) {
        if (v < kfr) {
          const T* pcol = xs + 16 * v + lcol;
          const u16x4 lo = lds_tr4(pcol + r0 * LDX), hi = lds_tr4(pcol + (r0 + 8) * LDX);
          const u16x8 xf = (u16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
          for (int u = 0; u < NF; ++u) mma16(yf[u], xf, acc[u][v]);  // acc[r] = dw[n = 4q + r][k = i]
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NF; ++u)
#pragma unroll
    for (int v = 0; v < KF; ++v) {
      const int k = kt0 + 16 * v + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 16 * (NF * wave + u) + 4 * q + r;
        if (n < N && k < K && !(dbg & 1)) atomicAdd(a.dw + (long)n * K + k, acc[u][v][r]);
      }
    }
}

extern "C" int mds_pw_wgrad(const mds_pw_wgrad_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->M > 0 && a->K > 0 && a->N > 0, "pw_wgrad: bad dims");
  MDS_REQUIRE(a->K % 8 == 0 && a->N % 8 == 0, "pw_wgrad: K, N must be multiples of 8");
  MDS_REQUIRE(a->x && a->dy && a->dw, "pw_wgrad: null pointer");
  MDS_REQUIRE(a->pro.mode == MDS_PRO_NONE || a->pro.mode == MDS_PRO_GATE || (a->pro.scale && a->pro.shift), "pw_wgrad: prologue needs scale/shift");
  MDS_REQUIRE((a->pro.mode != MDS_PRO_BN_SILU_GATE && a->pro.mode != MDS_PRO_GATE) || (a->pro.gate && a->pro.rows_per_group > 0), "pw_wgrad: gate prologue");
  MDS_REQUIRE(a->M < 2147483647L, "pw_wgrad: M too large");
  // Tile shape (NF, KF) = (2, 4): 128 x 64.  Wider tiles — (2,12) full-K, (2,8), (3,8) — cut the PMC
  // fetch from 2.6x to 1.5x of the algorithmic bytes but measured 15-80 % SLOWER (fewer blocks,
  // 2 instead of 3-4 waves/SIMD): the re-reads are L2/Infinity-Cache hits and occupancy matters more.
  const int NT = 128, KT = 64;
  const int tiles = cdiv(a->N, NT) * cdiv(a->K, KT);
  // Split-M blocks: every block ends with one atomic per output value, so fewer, longer blocks win until the chip runs
  // dry.  Timed ALONE the optimum is ~400-500 blocks in total (1024: +25...50 %) - but these launches run on the second
  // stream beside the dependent chain, where every block they hold is a CU slot the critical kernel does not get:
  // inside the training step 128 blocks in total measured best (448: +0.25 ms per step, 896: +0.8 ms, 32...224: flat).
  // The multi-million-row layers still want <= 2048 rows per block.
  const bool tr = a->dtype == MDS_BF16;     // the transposing-LDS-read kernel; fp32 takes the generic one
  long want_blocks = (mds_knob(MDS_KNOB_WG_BLOCKS) > 0 ? mds_knob(MDS_KNOB_WG_BLOCKS) : 192) / tiles;   // budget re-swept with the XCD-aligned splits: 128 / 192 / 224 / 288 -> 13.88 / 13.77 / 13.78 / 13.78 ms per step (was 128)
  if (want_blocks < a->M / 2048) want_blocks = a->M / 2048;
  if (tr && !(mds_knob(MDS_KNOB_WG_DBG) & 32)) {   // row splits in multiples of 8: split s and all its output tiles run on XCD s % 8 (see the kernel)
    want_blocks = (want_blocks + 4) / 8 * 8;
    if (want_blocks < 8) want_blocks = 8;
  }
  if (want_blocks < 1) want_blocks = 1;
  long rpb = (a->M + want_blocks - 1) / want_blocks;
  rpb = ((rpb + WG_ROWS - 1) / WG_ROWS) * WG_ROWS;
  if (rpb < 4 * WG_ROWS) rpb = 4 * WG_ROWS;
  dim3 grid(cdiv(a->M, rpb), tiles), block(256);
  if (tr) grid = dim3((unsigned)((cdiv(a->M, rpb) + 7) / 8 * 8 * tiles), 1);
#define WG_GO(T, PRO) \
  do { const size_t smem_ = (size_t)(KT + NT) * WgCfg<T>::LDT * sizeof(T); \
       MDS_LAUNCH((pw_wgrad_kernel<T, PRO, 2, 4>), grid, block, smem_, stream, *a, (int)rpb); } while (0)
#define WGT_GO(PRO) \
  do { const size_t smem_ = (size_t)WG_ROWS * (KT + 16 + NT + 16) * sizeof(bf16_t); \
       MDS_LAUNCH((pw_wgrad_tr_kernel<PRO, 2, 4>), grid, block, smem_, stream, *a, (int)rpb, mds_knob(MDS_KNOB_WG_DBG)); } while (0)
  if (tr) {
    switch (a->pro.mode) {
      case MDS_PRO_NONE: WGT_GO(MDS_PRO_NONE); break;
      case MDS_PRO_AFFINE: WGT_GO(MDS_PRO_AFFINE); break;
      case MDS_PRO_BN_SILU: WGT_GO(MDS_PRO_BN_SILU); break;
      case MDS_PRO_BN_SILU_GATE: WGT_GO(MDS_PRO_BN_SILU_GATE); break;
      case MDS_PRO_GATE: WGT_GO(MDS_PRO_GATE); break;
      default: mds_set_error("pw_wgrad: prologue mode %d", a->pro.mode); return MDS_ERR_BAD_ARG;
    }
    return mds_check_launch("pw_wgrad");
  }
#undef WGT_GO
  MDS_DISPATCH_DTYPE(a->dtype, T, {
    switch (a->pro.mode) {
      case MDS_PRO_NONE: WG_GO(T, MDS_PRO_NONE); break;
      case MDS_PRO_AFFINE: WG_GO(T, MDS_PRO_AFFINE); break;
      case MDS_PRO_BN_SILU: WG_GO(T, MDS_PRO_BN_SILU); break;
      case MDS_PRO_BN_SILU_GATE: WG_GO(T, MDS_PRO_BN_SILU_GATE); break;
      case MDS_PRO_GATE: WG_GO(T, MDS_PRO_GATE); break;
      default: mds_set_error("pw_wgrad: prologue mode %d", a->pro.mode); return MDS_ERR_BAD_ARG;
    }
  });
#undef WG_GO
  return mds_check_launch("pw_wgrad");
}
