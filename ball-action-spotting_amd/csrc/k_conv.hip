// k_conv.hip — dense 3x3 convolutions (EfficientNetV2 stages 0-2) as MFMA implicit GEMMs.
//
// One kernel, parameterised by a tap list, serves the stride-1 'same' conv, the TF-SAME stride-2
// conv, and both data gradients (see mds_conv_fwd_args).  A block owns a 16x16 (stride 1) or 8x16
// (stride 2) patch of output sub-grid points and 64 output channels.  Per 32-channel input chunk
// the input patch (+halo) is staged in LDS once — with the producer's BN+SiLU applied on the way in
// and zero padding applied after it — together with the weights of ALL taps, so the 9-tap MFMA
// loop runs without a barrier: 2 barriers per chunk, 0.5 LDS fragment reads per MFMA.
// Arithmetic intensity 100-600 FLOP/B: MFMA-bound layers.
// The thin layers (whole filter slab + whole-channel patch in <= 76 KB of LDS) take the persistent
// kernel conv_fwd_p_kernel instead; conv_wgrad_kernel is the weight gradient of both.
#include <stdlib.h>
#include "gemm.h"

#define CV_TB 16
#define CV_TAPPAD 12  /* MDS_MAX_TAPS rounded up so the float tables after the tap tables stay 16-byte aligned */
#define CV_BN 64

template <typename T> struct CvLd;   // LDS pitch of one staged pixel / weight row: 32 channels + 16 B
template <> struct CvLd<bf16_t> { static const int v = 48; };   // 96 B: 32 B x odd (see PwCfg)
template <> struct CvLd<float> { static const int v = 40; };     // 160 B

template <typename T, int PRO, int IS, int NFR>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? 1 : 2) void conv_fwd_kernel(mds_conv_fwd_args a, int dymin, int dxmin, int TH, int TW, int tg) {   // fp32 (parity path): one block per CU, 512 registers - two spilled 100-286 VGPRs
  MDS_CHAIN_PRIO();
  typedef typename Frag<T>::type frag_t;
  constexpr int LD = CvLd<T>::v, BN = 16 * NFR;
  constexpr int MF = (IS == 1) ? 4 : 2;        // 16-pixel row fragments per wave
  constexpr int TA = 4 * MF;                   // sub-grid rows per block
  constexpr int MAXX = (IS == 1) ? 6 : 9;      // input-patch vectors per thread (18x18 | 17x33 pixels x 4)
  constexpr int MAXW = (MDS_MAX_TAPS * BN * 4 + 255) / 256;
  MDS_DYN_SMEM(smem);
  const int npix = TH * TW;
  const float rTW = 1.0f / (float)TW;
  T* xs = (T*)smem;                                    // [npix][LD]   32-channel chunk of the patch
  T* ws = xs + npix * LD;                              // [tg taps][BN][LD]
  int* toff = (int*)(ws + tg * BN * LD);               // [ntaps] LDS element offset of each tap
  int* twi = toff + CV_TAPPAD;                         // [ntaps] weight slot of each tap (tables padded: pes stays 16-byte aligned)
  float* pes = (float*)(twi + CV_TAPPAD);              // [BN] output-transform scale / shift of the current N-tile (mds_epi_t)
  float* peh = pes + BN;
  const int emode = a.epi.mode;

  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int b0 = blockIdx.x * CV_TB, a0 = blockIdx.y * TA, img = blockIdx.z;
  const int Cin = a.Cin, Cout = a.Cout;
  const T* x = (const T*)a.x + (long)img * a.IH * a.IW * Cin;
  const T* w = (const T*)a.w;
  T* y = (T*)a.y;
  if (tid < a.ntaps) {  // per-lane indices into the argument arrays are global loads: do them once
    toff[tid] = ((a.dy[tid] - dymin) * TW + (a.dx[tid] - dxmin)) * LD;
    twi[tid] = a.wi[tid];
  }
  __syncthreads();
  int xbase[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) xbase[mf] = ((MF * wave + mf) * IS * TW + i * IS) * LD + 8 * q;
  const int wbase = i * LD + 8 * q;
  const int nx = npix * 4;
  const int ch = tid & 3;                       // this thread's 8-channel slice of every chunk (256 % 4 == 0)

  for (int n0 = 0; n0 < Cout; n0 += BN) {
    const int nfr = (Cout - n0 >= BN) ? NFR : ((Cout - n0) >> 4);
    const int wrows = nfr * 16;
    if (emode != MDS_EPI_NONE && tid < BN) {   // (visible after the first stage barrier; the previous N-tile ended with one)
      pes[tid] = n0 + tid < Cout ? a.epi.scale[n0 + tid] : 0.f;
      peh[tid] = n0 + tid < Cout ? a.epi.shift[n0 + tid] : 0.f;
    }
    f32x4 acc[MF][NFR];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NFR; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Software pipeline over (k-chunk, tap group): the global loads of the NEXT stage - its weights, and the
    // input patch when it opens a new chunk - are issued right after the current stage's LDS image is complete
    // and fly under its MFMAs.  Addresses are clamped (always legal); masks are applied at the LDS store.
    // per-thread element offsets (32-bit, relative to the wave-uniform image / filter base: one VGPR per load)
    float sc[8], sh[8];
    RawV8<T> rx[MAXX];
    unsigned xoff[MAXX], okx = 0;
#pragma unroll
    for (int l = 0; l < MAXX; ++l) {
      const int it = tid + 256 * l, pix = (it < nx ? it : nx - 1) >> 2;
      const int ty = fdiv(pix, rTW), tx = pix - ty * TW;
      const int iy = a0 * IS + dymin + ty, ix = b0 * IS + dxmin + tx;
      okx |= ((iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW) ? 1u : 0u) << l;
      const int cy = iy < 0 ? 0 : (iy >= a.IH ? a.IH - 1 : iy), cx = ix < 0 ? 0 : (ix >= a.IW ? a.IW - 1 : ix);
      xoff[l] = (unsigned)((cy * a.IW + cx) * Cin) * (unsigned)sizeof(T);   // BYTE offsets: base(SGPR) + 32-bit VGPR addressing
    }
    bool kin = true;      // this thread's 8 channels of the staged chunk exist (Cin % 32 != 0 tails)
    auto issue_x = [&](int kc) {
      const int kk = kc + 8 * ch;
      const unsigned kl = kk < Cin ? kk : 0;
      kin = kk < Cin;
      if (PRO != MDS_PRO_NONE) { load8f(a.pro.scale + kl, sc); load8f(a.pro.shift + kl, sh); }
#pragma unroll
      for (int l = 0; l < MAXX; ++l)
        rx[l].ld((const T*)((const char*)x + (xoff[l] + kl * (unsigned)sizeof(T))));
    };
    RawV8<T> rw[MAXW];
    unsigned woff[MAXW], okw = 0;
    bool kinw = true;
    auto plan_w = [&](int t0) {            // offsets / row masks of one tap group (once per N-tile when all taps fit)
      const int tn = a.ntaps - t0 < tg ? a.ntaps - t0 : tg;
      okw = 0;
#pragma unroll
      for (int l = 0; l < MAXW; ++l) {
        const int it = tid + 256 * l, rr = (it < tn * BN * 4 ? it : 0) >> 2;   // rr = tl * BN + r
        const int tl = rr / BN, r = rr - tl * BN;
        okw |= ((it < tn * BN * 4 && r < wrows) ? 1u : 0u) << l;
        woff[l] = (unsigned)(((n0 + (r < wrows ? r : 0)) * a.wtaps + twi[t0 + tl]) * Cin) * (unsigned)sizeof(T);
      }
    };
    const bool one_group = tg >= a.ntaps;
    plan_w(0);
    auto issue_w = [&](int kc, int t0) {   // all weight loads of one tap group
      const int kk = kc + 8 * ch;
      const unsigned kl = kk < Cin ? kk : 0;
      kinw = kk < Cin;
      if (!one_group) plan_w(t0);
      const int tn = a.ntaps - t0 < tg ? a.ntaps - t0 : tg;
#pragma unroll
      for (int l = 0; l < MAXW; ++l)
        rw[l].ld((const T*)((const char*)w + (woff[l] + kl * (unsigned)sizeof(T))));
    };
    RawV4<T> rres[MF][NFR];
    long orow[MF];        // output row of this lane's pixel per fragment (clamped: always a legal address)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int aa = a0 + MF * wave + mf, bb = b0 + i;
      orow[mf] = ((long)img * a.OH + (a.oy0 + (aa < a.A ? aa : a.A - 1) * a.os)) * a.OW + (a.ox0 + (bb < a.B ? bb : a.B - 1) * a.os);
    }
    auto load_res = [&]() {
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NFR; ++nf) rres[mf][nf].ld((const T*)a.residual + (orow[mf] * Cout + (n0 + 16 * nf + 4 * q < Cout ? n0 + 16 * nf + 4 * q : 0)));
    };
    issue_x(0);
    issue_w(0, 0);
    for (int kc = 0; kc < Cin; kc += 32) {
      for (int t0 = 0; t0 < a.ntaps; t0 += tg) {
        const int tn = a.ntaps - t0 < tg ? a.ntaps - t0 : tg;
        __syncthreads();  // previous fragment reads are done
        if (t0 == 0) {
#pragma unroll
          for (int l = 0; l < MAXX; ++l) {
            const int it = tid + 256 * l;
            if (it < nx) {
              const int pix = it >> 2;
              const bool ok = ((okx >> l) & 1u) && kin;
              if (PRO == MDS_PRO_NONE) {
                if (!ok) rx[l].zero();
                rx[l].st(xs + pix * LD + 8 * ch);
              } else {
                float v[8];
                rx[l].get(v);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  float z = v[j] * sc[j] + sh[j];
                  z = (PRO == MDS_PRO_AFFINE) ? z : siluf_(z);
                  v[j] = ok ? z : 0.f;  // zero padding AFTER the activation
                }
                store8(xs + pix * LD + 8 * ch, v);
                MDS_SCHED_FENCE();  // keep one vector's exp/rcp temporaries live at a time (register budget)
              }
            }
          }
        }
#pragma unroll
        for (int l = 0; l < MAXW; ++l) {
          const int it = tid + 256 * l;
          if (it < tn * BN * 4) {
            if (!(((okw >> l) & 1u) && kinw)) rw[l].zero();
            rw[l].st(ws + (it >> 2) * LD + 8 * ch);
          }
        }
        __syncthreads();
        if (t0 + tg < a.ntaps) issue_w(kc, t0 + tg);
        else if (kc + 32 < Cin) { issue_x(kc + 32); issue_w(kc + 32, 0); }
        else if (a.residual && NFR <= 2) load_res();   // last stage of the tile: the residual operand flies under its MFMAs
        for (int tl = 0; tl < tn; ++tl) {
          const int xo = toff[t0 + tl];
          const T* wt = ws + tl * BN * LD + wbase;
          frag_t xf[MF], wf[NFR];   // all NFR fragments, no branch in this loop: rows past the layer's Cout are staged as zeros
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) xf[mf] = ld_frag(xs + xbase[mf] + xo);
#pragma unroll
          for (int nf = 0; nf < NFR; ++nf) wf[nf] = ld_frag(wt + 16 * nf * LD);
#pragma unroll
          for (int nf = 0; nf < NFR; ++nf)
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) mma16(wf[nf], xf[mf], acc[mf][nf]);
        }
      }
    }

    if (a.residual && NFR > 2) load_res();   // (register budget: with 3-4 column fragments the loads are batched here instead)
    float ps[16], pss[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { ps[e] = 0.f; pss[e] = 0.f; }
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int aa = a0 + MF * wave + mf, bb = b0 + i;
      const bool valid = aa < a.A && bb < a.B;
      const long row = orow[mf];
#pragma unroll
      for (int nf = 0; nf < NFR; ++nf) {
        if (nf < nfr && valid) {
          const int n = n0 + 16 * nf + 4 * q;
          float v[4] = {acc[mf][nf][0], acc[mf][nf][1], acc[mf][nf][2], acc[mf][nf][3]};
          if (emode != MDS_EPI_NONE) {
            const f32x4 es = *(const f32x4*)(pes + 16 * nf + 4 * q), eh = *(const f32x4*)(peh + 16 * nf + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float z = v[r] * es[r] + eh[r];
              v[r] = emode == MDS_EPI_BN_SILU ? siluf_(z) : z;
            }
          }
          if (a.residual) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += rres[mf][nf].get(r);
          }
          store4(y + row * Cout + n, v);
#pragma unroll
          for (int r = 0; r < 4; ++r) { ps[nf * 4 + r] += v[r]; pss[nf * 4 + r] += v[r] * v[r]; }
        }
      }
    }
    if (a.stats) {   // one column per lane after the reduce-scatter: straight to the slot (no block barrier)
      const int e = reduce_scatter16(ps, i);
      reduce_scatter16(pss, i);
      const int n = n0 + 16 * (e >> 2) + 4 * q + (e & 3);
      if (16 * (e >> 2) < BN && n < Cout) {
        const int slot = (blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * 7 + wave) % MDS_STAT_SLOTS;
        double* st = a.stats + (long)slot * 2 * Cout;
        atomicAdd(st + n, (double)ps[0]);
        atomicAdd(st + Cout + n, (double)pss[0]);
      }
    }
    __syncthreads();  // the next N-tile restages LDS
  }
}

// ------------------------------------------------------------------------------------ persistent forward
// Layers whose filter slab (all taps x Cin of an N-tile of 16*NFR output channels) and one whole-channel input
// patch fit in LDS.  These layers stream 150-450 MB at 100-600 FLOP/B: the bound is HBM, and what decides
// whether a CU gets its share of it is the number of bytes it keeps IN FLIGHT (Little: ~13 B/clk x ~5000 clk).
// So a block keeps its filter slab RESIDENT ([n][k = tap*Cin + ch], one 16-byte fragment read per MFMA
// operand), walks a contiguous range of tiles, and holds the raw patches of the NEXT TWO tiles in two register
// sets while the MFMAs of the current one run; nothing else in the loop touches vmcnt (prologue coefficients
// come from an LDS table, there is no residual operand here), so the in-order counter never drains a prefetch.
// The k index is flattened over (tap, channel): Cin = 16 needs 5 k-steps for 9 taps instead of 9 half-empty ones.
// Tap GROUPS (stride-2 data gradient): the 1+2+2+4 taps of the four output parities are evaluated from ONE
// staged patch of dy and written to their own output sub-grids - one read of dy instead of four.
// LDS row pitch (elements): the next byte pitch that is 32 (mod 64) - conflict-free for ds_read_b128
MDS_DEV int cvp_pitch(int elems, int esz) { const int b = elems * esz; return (b + ((96 - b % 64) % 64)) / esz; }
static inline int cvp_pitch_h(int elems, int esz) { const int b = elems * esz; return (b + ((96 - b % 64) % 64)) / esz; }

#define CVQ_MAXX 6   // patch vectors (8 channels of one pixel) a thread holds per register set
struct CvqGeom {
  int dymin, dxmin, TH, TW, tiles_a, tiles_b, tpb, KS;
  int ng;                 // tap groups (1 = plain conv)
  int gks[5];             // k-steps [gks[g], gks[g+1]) belong to group g
  int goy[4], gox[4], gA[4], gB[4];
};

template <typename T, int MF, int NFR> struct CvqOcc { static const int v = (sizeof(T) == 4 && MF * NFR >= 8) ? 1 : ((sizeof(T) == 4 || MF * NFR >= 8 || NFR == 4) ? 2 : 3); };   // fp32 wide tiles: one block per CU (two spilled 119-151 VGPRs)

// X3 (fp32 inference plans, a.epi.mode != NONE): split-bf16 products, see Mma<float, true> in platform.h
template <typename T, bool HASPRO, int IS, int MF, int NFR, bool X3 = false>
__global__ __launch_bounds__(256, (CvqOcc<T, MF, NFR>::v)) void conv_fwd_q_kernel(mds_conv_fwd_args a, CvqGeom gq) {
  MDS_CHAIN_PRIO();
  typedef Mma<T, X3> MM;
  constexpr int TA = 4 * MF, BNQ = 16 * NFR, MAXX = CVQ_MAXX;
  MDS_DYN_SMEM(smem);
  const int Cin = a.Cin, Cout = a.Cout, K = a.ntaps * Cin, KS = gq.KS, TW = gq.TW;
  const int LDX = cvp_pitch(Cin, sizeof(T)), LDW = cvp_pitch(KS * 32, sizeof(T));
  const int npix = gq.TH * TW, cpp = Cin >> 3, nitems = npix * cpp;
  T* xs = (T*)smem;                        // [npix][LDX]
  T* ws = xs + npix * LDX;                 // [BNQ][LDW]
  double* st_s = (double*)(ws + BNQ * LDW);  // [BNQ] block-level statistic sums: fp64, so the order the four waves add in does not matter
  double* st_ss = st_s + BNQ;
  float* psc = (float*)(st_ss + BNQ);       // [Cin] prologue scale / shift
  float* psh = psc + Cin;
  float* pes = psh + Cin;                  // [BNQ] output-transform scale / shift (mds_epi_t)
  float* peh = pes + BNQ;
  int* ktab = (int*)(peh + BNQ);           // [KS*4] LDS offset (tap shift + channel) of each 8-wide k chunk
  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.y * BNQ;
  const int nfr = (Cout - n0 >= BNQ) ? NFR : ((Cout - n0) >> 4);
  const T* w = (const T*)a.w;
  T* y = (T*)a.y;
  const float rTW = 1.0f / (float)TW, rcpp = 1.0f / (float)cpp;
  const int tiles_ab = gq.tiles_a * gq.tiles_b;

  for (int c = tid; c < KS * 4; c += 256) {
    const int k = 8 * c;
    int off = 0;
    if (k < K) {
      const int t = k / Cin, ch = k - t * Cin;
      off = ((a.dy[t] - gq.dymin) * TW + (a.dx[t] - gq.dxmin)) * LDX + ch;
    }
    ktab[c] = off;
  }
  for (int e = tid; e < BNQ * KS * 4; e += 256) {
    const int n = e / (KS * 4), c = e - n * (KS * 4), k = 8 * c;
    RawV8<T> r;
    r.zero();
    if (n < nfr * 16 && k < K) {
      const int t = k / Cin, ch = k - t * Cin;
      r.ld(w + ((long)(n0 + n) * a.wtaps + a.wi[t]) * Cin + ch);
    }
    r.st(ws + n * LDW + k);
  }
  if (tid < BNQ) { st_s[tid] = 0.0; st_ss[tid] = 0.0; }
  if (HASPRO) {
    for (int c = tid; c < Cin; c += 256) { psc[c] = a.pro.scale[c]; psh[c] = a.pro.shift[c]; }
  }
  const bool silu = a.pro.mode == MDS_PRO_BN_SILU;
  const int emode = a.epi.mode;
  if (emode != MDS_EPI_NONE && tid < BNQ) {
    pes[tid] = n0 + tid < Cout ? a.epi.scale[n0 + tid] : 0.f;
    peh[tid] = n0 + tid < Cout ? a.epi.shift[n0 + tid] : 0.f;
  }

  const long total_tiles = (long)a.N * tiles_ab;
  long tl = (long)xcd_contiguous(blockIdx.x, gridDim.x) * gq.tpb;      // consecutive tile rows (shared halo rows) on one XCD
  long tl_end = tl + gq.tpb;
  if (tl_end > total_tiles) tl_end = total_tiles;
  int xbase[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) xbase[mf] = ((MF * wave + mf) * IS * TW + i * IS) * LDX;
  const int wbase = i * LDW + 8 * q;

  auto origin = [&](long t, int& img, int& a0, int& b0) {
    img = (int)(t / tiles_ab);
    const int rem = (int)(t - (long)img * tiles_ab);
    a0 = (rem / gq.tiles_b) * TA; b0 = (rem % gq.tiles_b) * CV_TB;
  };
  // every issue() executes the same number of loads (tiles past the end re-read the last one): the compiler's
  // vmcnt bookkeeping then never has a path with fewer loads outstanding, i.e. never waits on the younger set
  auto issue = [&](long t, RawV8<T>(&rx)[MAXX], unsigned& okx) {
    int img, a0, b0;
    origin(t < tl_end ? t : tl_end - 1, img, a0, b0);
    const T* x = (const T*)a.x + (long)img * a.IH * a.IW * Cin;
    okx = 0;
#pragma unroll
    for (int l = 0; l < MAXX; ++l) {
      const int it0 = tid + 256 * l, it = it0 < nitems ? it0 : nitems - 1;
      const int pix = fdiv(it, rcpp), c8 = it - pix * cpp;
      const int ty = fdiv(pix, rTW), tx = pix - ty * TW;
      const int iy = a0 * IS + gq.dymin + ty, ix = b0 * IS + gq.dxmin + tx;
      const bool ok = iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
      okx |= (ok ? 1u : 0u) << l;
      const int cy = iy < 0 ? 0 : (iy >= a.IH ? a.IH - 1 : iy), cx = ix < 0 ? 0 : (ix >= a.IW ? a.IW - 1 : ix);
      rx[l].ld(x + ((long)cy * a.IW + cx) * Cin + 8 * c8);
    }
  };
  auto stage = [&](RawV8<T>(&rx)[MAXX], unsigned okx) {
#pragma unroll
    for (int l = 0; l < MAXX; ++l) {
      const int it = tid + 256 * l;
      if (it < nitems) {
        const int pix = fdiv(it, rcpp), c8 = it - pix * cpp;
        const bool ok = (okx >> l) & 1u;
        if (!HASPRO) {
          if (!ok) rx[l].zero();
          rx[l].st(xs + pix * LDX + 8 * c8);
        } else {
          float v[8];
          rx[l].get(v);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float z = v[j] * psc[8 * c8 + j] + psh[8 * c8 + j];
            z = silu ? siluf_(z) : z;
            v[j] = ok ? z : 0.f;  // zero padding AFTER the activation
          }
          store8(xs + pix * LDX + 8 * c8, v);
          MDS_SCHED_FENCE();
        }
      }
    }
  };
  float ps[4 * NFR], pss[4 * NFR];
#pragma unroll
  for (int e = 0; e < 4 * NFR; ++e) { ps[e] = 0.f; pss[e] = 0.f; }

  auto compute = [&](long t) {
    int img, a0, b0;
    origin(t, img, a0, b0);
    for (int g = 0; g < gq.ng; ++g) {
      f32x4 acc[MF][NFR];
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NFR; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int s0 = gq.gks[g], s1 = gq.gks[g + 1];
      int xo_next = ktab[4 * s0 + q];
      for (int s = s0; s < s1; ++s) {
        const int xo = xo_next;   // the tap-offset lookup of step s+1 is issued a step ahead
        xo_next = ktab[4 * (s + 1 < s1 ? s + 1 : s) + q];
        typename MM::frag xf[MF], wf[NFR];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) xf[mf] = MM::prep(ld_frag(xs + xbase[mf] + xo));
#pragma unroll
        for (int nf = 0; nf < NFR; ++nf) wf[nf] = MM::prep(ld_frag(ws + wbase + 16 * nf * LDW + 32 * s));   // rows past Cout are zeros
#pragma unroll
        for (int nf = 0; nf < NFR; ++nf)
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) MM::mma(wf[nf], xf[mf], acc[mf][nf]);
      }
      const int gA = gq.gA[g], gB = gq.gB[g], goy = gq.goy[g], gox = gq.gox[g];
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int aa = a0 + MF * wave + mf, bb = b0 + i;
        const bool valid = aa < gA && bb < gB;
        const long row = ((long)img * a.OH + (goy + aa * a.os)) * a.OW + (gox + bb * a.os);
#pragma unroll
        for (int nf = 0; nf < NFR; ++nf) {
          if (nf < nfr && valid) {
            const int n = n0 + 16 * nf + 4 * q;
            float v[4] = {acc[mf][nf][0], acc[mf][nf][1], acc[mf][nf][2], acc[mf][nf][3]};
            if (emode != MDS_EPI_NONE) {
              const f32x4 es = *(const f32x4*)(pes + 16 * nf + 4 * q), eh = *(const f32x4*)(peh + 16 * nf + 4 * q);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float z = v[r] * es[r] + eh[r];
                v[r] = emode == MDS_EPI_BN_SILU ? siluf_(z) : z;
              }
            }
            store4(y + row * Cout + n, v);
#pragma unroll
            for (int r = 0; r < 4; ++r) { ps[nf * 4 + r] += v[r]; pss[nf * 4 + r] += v[r] * v[r]; }
          }
        }
      }
    }
  };

  RawV8<T> rA[MAXX], rB[MAXX];
  unsigned okA = 0, okB = 0;
  issue(tl, rA, okA);
  issue(tl + 1, rB, okB);
  for (; tl < tl_end; tl += 2) {
    __syncthreads();          // the previous tile's fragment reads are done (first pass: slab + tables staged)
    stage(rA, okA);
    __syncthreads();
    issue(tl + 2, rA, okA);   // two tiles ahead: a full tile period (+ the other set's) to land
    compute(tl);
    if (tl + 1 < tl_end) {
      __syncthreads();
      stage(rB, okB);
      __syncthreads();
      issue(tl + 3, rB, okB);
      compute(tl + 1);
    }
  }
  if (a.stats) {
    float p16[16], q16[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { p16[e] = e < 4 * NFR ? ps[e < 4 * NFR ? e : 0] : 0.f; q16[e] = e < 4 * NFR ? pss[e < 4 * NFR ? e : 0] : 0.f; }
    const int e = reduce_scatter16(p16, i);
    reduce_scatter16(q16, i);
    const int nl = 16 * (e >> 2) + 4 * q + (e & 3);
    if (nl < BNQ) {
      atomicAdd(&st_s[nl], (double)p16[0]);
      atomicAdd(&st_ss[nl], (double)q16[0]);
    }
    __syncthreads();
    if (tid < BNQ && n0 + tid < Cout) {
      double* st = a.stats + (long)(blockIdx.x % MDS_STAT_SLOTS) * 2 * Cout;
      atomicAdd(st + n0 + tid, st_s[tid]);
      atomicAdd(st + Cout + n0 + tid, st_ss[tid]);
    }
  }
}

int c3_try(const mds_conv_fwd_args* a, mds_stream_t stream);   // k_c3.hip: 1 = launched there

static int tap_extent(const int* d, int n, int* dmin) {
  int lo = d[0], hi = d[0];
  for (int t = 1; t < n; ++t) { if (d[t] < lo) lo = d[t]; if (d[t] > hi) hi = d[t]; }
  *dmin = lo;
  return hi - lo;
}

extern "C" int mds_conv_fwd(const mds_conv_fwd_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->N > 0 && a->A > 0 && a->B > 0, "conv_fwd: bad dims");
  MDS_REQUIRE(a->Cin % 8 == 0 && a->Cout % 16 == 0, "conv_fwd: Cin=%d %% 8, Cout=%d %% 16", a->Cin, a->Cout);
  MDS_REQUIRE(a->ntaps >= 1 && a->ntaps <= MDS_MAX_TAPS && a->wtaps >= 1, "conv_fwd: ntaps");
  MDS_REQUIRE((a->is == 1 || a->is == 2) && a->os >= 1, "conv_fwd: strides");
  MDS_REQUIRE(a->x && a->w && a->y, "conv_fwd: null pointer");
  MDS_REQUIRE(a->pro.mode == MDS_PRO_NONE || a->pro.mode == MDS_PRO_AFFINE || a->pro.mode == MDS_PRO_BN_SILU, "conv_fwd: prologue mode");
  MDS_REQUIRE(a->pro.mode == MDS_PRO_NONE || (a->pro.scale && a->pro.shift), "conv_fwd: prologue needs scale/shift");
  MDS_REQUIRE(a->oy0 + (a->A - 1) * a->os < a->OH && a->ox0 + (a->B - 1) * a->os < a->OW, "conv_fwd: sub-grid exceeds output");
  MDS_REQUIRE(a->epi.mode == MDS_EPI_NONE || (a->epi.scale && a->epi.shift && !a->stats), "conv_fwd: an output transform needs scale/shift and excludes statistics");
  if (c3_try(a, stream)) return mds_check_launch("conv_fwd");    // bf16 layers whose filter slice fits the consumers' registers (k_c3.hip)
  MDS_REQUIRE(a->post.mode == MDS_POST_NONE, "conv_fwd: post statistics only where mds_conv_dgrad_post_ok() says so");
  int dymin, dxmin;
  const int eh = tap_extent(a->dy, a->ntaps, &dymin), ew = tap_extent(a->dx, a->ntaps, &dxmin);
  const int TA = a->is == 1 ? 16 : 8;
  const int TH = (TA - 1) * a->is + eh + 1, TW = (CV_TB - 1) * a->is + ew + 1;
  dim3 block(256);
  const int ng = a->ngroups > 1 ? a->ngroups : 1;
  MDS_REQUIRE(ng <= 4, "conv_fwd: at most 4 tap groups");
  if (ng > 1) {
    int nt_ = 0;
    for (int g = 0; g < ng; ++g) {
      MDS_REQUIRE(a->g_ntaps[g] >= 1 && (a->g_ntaps[g] * a->Cin) % 32 == 0, "conv_fwd: a tap group needs ntaps*Cin %% 32 == 0");
      MDS_REQUIRE(a->g_A[g] > 0 && a->g_B[g] > 0 && a->g_oy0[g] + (a->g_A[g] - 1) * a->os < a->OH && a->g_ox0[g] + (a->g_B[g] - 1) * a->os < a->OW,
                  "conv_fwd: group sub-grid exceeds output");
      nt_ += a->g_ntaps[g];
    }
    MDS_REQUIRE(nt_ == a->ntaps && a->is == 1 && !a->residual && !a->stats, "conv_fwd: tap groups partition the tap list (is = 1, no residual / statistics)");
  }
  if (!a->residual) {  // persistent variant when a filter slab + one whole-channel input patch fit in LDS
    const int KS = cdiv(a->ntaps * a->Cin, 32);
    const int esz = a->dtype == MDS_BF16 ? 2 : 4;
    const int co16 = a->Cout / 16;
    const int nfr_try[3] = {co16 >= 3 ? 4 : co16, co16 >= 3 ? 2 : (co16 == 2 ? 1 : 0), co16 >= 3 ? 1 : 0};
    const int mf_try[3] = {a->is == 1 ? 4 : 2, a->is == 1 ? 2 : 1, a->is == 1 ? 1 : 0};
    int MFs = 0, NFRs = 0, THq = 0, TWq = 0;
    size_t smem = 0;
    for (int lim = 0; lim < (ng > 1 ? 2 : 1) && !MFs; ++lim) {   // tap groups have no other kernel: they may take the whole LDS
      for (int ni = 0; ni < 3 && !MFs; ++ni) {
        for (int mi = 0; mi < 3 && !MFs; ++mi) {
          const int nf = nfr_try[ni], mf = mf_try[mi];
          if (!nf || !mf) continue;
          const int th = (4 * mf - 1) * a->is + eh + 1, tw = (CV_TB - 1) * a->is + ew + 1;
          const size_t sm = ((size_t)th * tw * cvp_pitch_h(a->Cin, esz) + (size_t)16 * nf * cvp_pitch_h(KS * 32, esz)) * esz +
                            6 * 16 * nf * sizeof(float) + 2 * (size_t)a->Cin * sizeof(float) + (size_t)KS * 16;
          if (th * tw * (a->Cin / 8) <= CVQ_MAXX * 256 && sm <= (lim ? 160 : 76) * 1024) { MFs = mf; NFRs = nf; THq = th; TWq = tw; smem = sm; }
        }
      }
    }
    MDS_REQUIRE(MFs || ng == 1, "conv_fwd: tap groups need a filter slab + patch that fit in LDS (Cin=%d Cout=%d)", a->Cin, a->Cout);
    if (MFs) {
      CvqGeom gq;
      gq.dymin = dymin; gq.dxmin = dxmin; gq.TH = THq; gq.TW = TWq; gq.KS = KS; gq.ng = ng;
      int Amax = 0, Bmax = 0, k0 = 0;
      for (int g = 0; g < ng; ++g) {
        gq.gks[g] = k0;
        if (ng > 1) { gq.goy[g] = a->g_oy0[g]; gq.gox[g] = a->g_ox0[g]; gq.gA[g] = a->g_A[g]; gq.gB[g] = a->g_B[g]; k0 += a->g_ntaps[g] * a->Cin / 32; }
        else { gq.goy[g] = a->oy0; gq.gox[g] = a->ox0; gq.gA[g] = a->A; gq.gB[g] = a->B; k0 = KS; }
        Amax = gq.gA[g] > Amax ? gq.gA[g] : Amax; Bmax = gq.gB[g] > Bmax ? gq.gB[g] : Bmax;
      }
      gq.gks[ng] = k0;
      for (int g = ng; g < 4; ++g) { gq.gks[g + 1] = k0; gq.goy[g] = gq.gox[g] = gq.gA[g] = gq.gB[g] = 0; }
      gq.tiles_a = cdiv(Amax, 4 * MFs); gq.tiles_b = cdiv(Bmax, CV_TB);
      const long total = (long)a->N * gq.tiles_a * gq.tiles_b;
      const int nt = cdiv(a->Cout, 16 * NFRs);
      int occ = (esz == 4 || MFs * NFRs >= 8 || NFRs == 4) ? 2 : 3;
      while (occ > 1 && smem * occ > 160 * 1024) --occ;
      // grid.x of the persistent kernel (grid.y = nt).  Was 256 * occ * 2 / nt ("two balanced rounds"); swept inside the step with
      // MDS_KNOB_CONV_BLOCKS: 384 / 512 / 768 / 1024 / 1536 / 2048 -> 13.91 / 13.77 / 13.83 / 13.91 / 13.96 / 13.98 ms (former rule 13.83)
      long want = 512;
      (void)occ;
      if (mds_knob(MDS_KNOB_CONV_BLOCKS) > 0) want = mds_knob(MDS_KNOB_CONV_BLOCKS);
      if (want < 1) want = 1;
      gq.tpb = (int)cdiv(total, want < total ? want : total);
      dim3 pgrid(cdiv(total, gq.tpb), nt);
      const bool x3 = MDS_EVAL_X3 && a->dtype == MDS_F32 && a->epi.mode != MDS_EPI_NONE;
#define CVQ_GO3(T, HP, IS_, MF_, NF_)                                                                                           \
  do {                                                                                                                          \
    if (x3) MDS_LAUNCH((conv_fwd_q_kernel<T, HP, IS_, MF_, NF_, (sizeof(T) == 4)>), pgrid, block, smem, stream, *a, gq);       \
    else MDS_LAUNCH((conv_fwd_q_kernel<T, HP, IS_, MF_, NF_, false>), pgrid, block, smem, stream, *a, gq);                     \
  } while (0)
#define CVQ_GO2(T, HP, IS_, MF_) do { if (NFRs == 4) CVQ_GO3(T, HP, IS_, MF_, 4); else if (NFRs == 2) CVQ_GO3(T, HP, IS_, MF_, 2); else CVQ_GO3(T, HP, IS_, MF_, 1); } while (0)
#define CVQ_GO(T, HP)                                                                   \
  do {                                                                                  \
    if (a->is == 1 && MFs == 4) CVQ_GO2(T, HP, 1, 4);                                   \
    else if (a->is == 1 && MFs == 2) CVQ_GO2(T, HP, 1, 2);                              \
    else if (a->is == 1) CVQ_GO2(T, HP, 1, 1);                                          \
    else if (MFs == 2) CVQ_GO2(T, HP, 2, 2);                                            \
    else CVQ_GO2(T, HP, 2, 1);                                                          \
  } while (0)
      MDS_DISPATCH_DTYPE(a->dtype, T, {
        if (a->pro.mode == MDS_PRO_NONE) CVQ_GO(T, false); else CVQ_GO(T, true);
      });
#undef CVQ_GO
#undef CVQ_GO2
#undef CVQ_GO3
      return mds_check_launch("conv_fwd");
    }
  }
  MDS_REQUIRE(ng == 1, "conv_fwd: tap groups need Cin % 32 == 0, K = 9 Cin < 1152 (the persistent kernel) and no residual operand");
  MDS_REQUIRE((long)a->IH * a->IW * a->Cin * (a->dtype == MDS_BF16 ? 2 : 4) < 4294967296L && (long)a->Cout * a->wtaps * a->Cin * 4 < 4294967296L,
              "conv_fwd: one image / the filter must stay below 4 GB (32-bit byte offsets)");
  dim3 grid(cdiv(a->B, CV_TB), cdiv(a->A, TA), a->N);
  const int co16 = a->Cout / 16;
  const int NFRg = co16 <= 4 ? co16 : ((co16 % 4 == 0 || co16 % 3) ? 4 : 3);   // output channels per pass: all of them up to 64
#define CV_GO2(T, PRO, NF)                                                                              \
  do {                                                                                                  \
    const int LD = CvLd<T>::v;                                                                          \
    int tg = a->ntaps;  /* taps staged per barrier group: all of them unless LDS (two blocks per CU) says no */   \
    while (tg > 1 && (size_t)(TH * TW + tg * 16 * NF) * LD * sizeof(T) > 76 * 1024) --tg;               \
    const size_t smem = (size_t)(TH * TW + tg * 16 * NF) * LD * sizeof(T) + 2 * CV_TAPPAD * sizeof(int) + 2 * 16 * NF * sizeof(float); \
    if (a->is == 1) MDS_LAUNCH((conv_fwd_kernel<T, PRO, 1, NF>), grid, block, smem, stream, *a, dymin, dxmin, TH, TW, tg); \
    else MDS_LAUNCH((conv_fwd_kernel<T, PRO, 2, NF>), grid, block, smem, stream, *a, dymin, dxmin, TH, TW, tg); \
  } while (0)
#define CV_GO(T, PRO) do { if (NFRg == 1) CV_GO2(T, PRO, 1); else if (NFRg == 2) CV_GO2(T, PRO, 2); else if (NFRg == 3) CV_GO2(T, PRO, 3); else CV_GO2(T, PRO, 4); } while (0)
  MDS_DISPATCH_DTYPE(a->dtype, T, {
    switch (a->pro.mode) {
      case MDS_PRO_NONE: CV_GO(T, MDS_PRO_NONE); break;
      case MDS_PRO_AFFINE: CV_GO(T, MDS_PRO_AFFINE); break;
      default: CV_GO(T, MDS_PRO_BN_SILU); break;
    }
  });
#undef CV_GO
#undef CV_GO2
  return mds_check_launch("conv_fwd");
}

#define CV_TA 8   // spatial tile of the weight-gradient kernel below
// ------------------------------------------------------------------------------------ wgrad
// dw[co][ci][tap] += sum over output points of dy[p][co] * pro(x)[p*is + tap][ci].
// The MFMA reduction index is the output point; a block walks `tiles_per_block` 8x16 patches and
// keeps all 9 x Cin x 64 accumulators in registers (wave w owns taps w, w+4, w+8).
#define CW_COT 64
// LDS images are [pixel][channel].  bf16 fragments (8 pixels of one channel per lane) come from two
// transposing reads; a 32-lane LDS cycle touches 8 consecutive pixels x 32 bytes, conflict-free when
// the pixel pitch is an odd multiple of 32 bytes.  fp32 has no transposing read: scalar gathers,
// odd dword pitch.
template <typename T> struct CwCfg;
template <> struct CwCfg<float> {
  static constexpr int LDY = CW_COT + 2;
  static MDS_DEV int ldx(int Cin) { return Cin + 2; }
  static MDS_DEV f32x8 frag(const float* base, int pitch, int pix_a, int pix_b, int pstep, int c0, int i) {
    f32x8 f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f[j] = base[(pix_a + j * pstep) * pitch + c0 + i];
      f[4 + j] = base[(pix_b + j * pstep) * pitch + c0 + i];
    }
    return f;
  }
  static MDS_DEV void put8(float* dst, const float (&v)[8]) { lds_store8_u32(dst, v); }
  static MDS_DEV void putraw(float* dst, const RawV8<float>& r) { float v[8]; r.get(v); lds_store8_u32(dst, v); }
};
template <> struct CwCfg<bf16_t> {
  static constexpr int LDY = CW_COT + 16;
  static MDS_DEV int ldx(int Cin) { return Cin == 32 ? 48 : Cin; }
  static MDS_DEV u16x8 frag(const bf16_t* base, int pitch, int pix_a, int pix_b, int pstep, int c0, int i) {
    const int off = (i >> 2) * pstep * pitch + c0 + 4 * (i & 3);
    const u16x4 lo = lds_tr4(base + pix_a * pitch + off), hi = lds_tr4(base + pix_b * pitch + off);
    return (u16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
  static MDS_DEV void put8(bf16_t* dst, const float (&v)[8]) { store8(dst, v); }
  static MDS_DEV void putraw(bf16_t* dst, const RawV8<bf16_t>& r) { r.st(dst); }
};
static inline int cw_ldx(int dtype, int Cin) { return dtype == MDS_BF16 ? (Cin == 32 ? 48 : Cin) : Cin + 2; }
static inline int cw_ldy(int dtype) { return dtype == MDS_BF16 ? CW_COT + 16 : CW_COT + 2; }

template <typename T, int PRO, int CIFR, int COFR, int XL>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(mds_conv_wgrad_args a, int dymin, int dxmin, int TH, int TW,
                                                         int tiles_a, int tiles_b, int tiles_per_block) {
  typedef typename Frag<T>::type frag_t;
  constexpr int NU = (9 * CIFR + 3) / 4;  // (tap, 16-input-channel fragment) units owned by a wave
  MDS_DYN_SMEM(smem);
  const int Cin = a.Cin, Cout = a.Cout;
  const int LDX = CwCfg<T>::ldx(Cin);
  constexpr int LDY = CwCfg<T>::LDY;
  T* xs = (T*)smem;            // [TH*TW][LDX]
  T* dys = xs + TH * TW * LDX;  // [128][LDY]
  float* flush = (float*)(dys + 128 * LDY);  // [16][Cin*wtaps] filter-gradient staging
  float* psc = flush + 16 * Cin * a.wtaps;   // [Cin] prologue scale / shift (a load consumed inside the staging loop would
  float* psh = psc + Cin;                    //       expose an L2 round trip per item: vmcnt retires in order)
  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int co0 = blockIdx.y * CW_COT;
  const int cofr = (Cout - co0 >= 16 * COFR) ? COFR : ((Cout - co0) >> 4);
  const int cpp = Cin >> 3;  // 8-channel chunks per pixel
  const int npix = TH * TW;
  const float rcpp = 1.0f / (float)cpp, rTW = 1.0f / (float)TW;
  const int tiles_ab = tiles_a * tiles_b;
  const long total_tiles = (long)a.N * tiles_ab;
  long tl = (long)xcd_contiguous(blockIdx.x, gridDim.x) * tiles_per_block;
  long tl_end = tl + tiles_per_block;
  if (tl_end > total_tiles) tl_end = total_tiles;
  if (PRO != MDS_PRO_NONE) {
    for (int c = tid; c < Cin; c += 256) { psc[c] = a.pro.scale[c]; psh[c] = a.pro.shift[c]; }
  }
  // unit u = wave + 4k -> (tap u / CIFR, fragment u % CIFR); tap offsets are wave-uniform scalars
  // read ONCE (a per-lane index into the kernel-argument arrays inside the loop compiles to a
  // dependent global load per tap — that alone was 2/3 of this kernel's time)
  int utoff[NU], ukc[NU], uwi[NU];
  bool uok[NU];
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const int u = wave + 4 * k;
    uok[k] = u < a.ntaps * CIFR;
    const int t = uok[k] ? u / CIFR : 0;
    ukc[k] = 16 * (u % CIFR);
    utoff[k] = (a.dy[t] - dymin) * TW + a.dx[t] - dxmin;
    uwi[k] = a.wi[t];
  }

  f32x4 acc[NU][COFR];
#pragma unroll
  for (int k = 0; k < NU; ++k)
#pragma unroll
    for (int cf = 0; cf < COFR; ++cf) acc[k][cf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Register software pipeline: the global loads of tile t+1 are in flight while tile t's MFMAs run.  Loads are never
  // conditional (clamped addresses, masks applied at the LDS store) and XL is sized to the layer (3 items per thread
  // for the stride-1 32-channel layers, 9 for the largest stride-2 patch).
  const int nitems = npix * cpp;
  auto tile_origin = [&](long t, int& img, int& a0, int& b0) {
    img = (int)(t / tiles_ab);
    const int rem = (int)(t - (long)img * tiles_ab);
    a0 = (rem / tiles_b) * CV_TA; b0 = (rem % tiles_b) * CV_TB;
  };
  auto issue = [&](long t, RawV8<T>(&rx)[XL], RawV8<T>(&ry)[4], unsigned& okx, unsigned& oky) {
    int img, a0, b0;
    tile_origin(t < tl_end ? t : tl_end - 1, img, a0, b0);   // tiles past the end re-read the last one (never staged)
    const T* x = (const T*)a.x + (long)img * a.IH * a.IW * Cin;
    const T* dy = (const T*)a.dyt + (long)img * a.OH * a.OW * Cout;
    okx = 0; oky = 0;
#pragma unroll
    for (int l = 0; l < XL; ++l) {
      const int it0 = tid + 256 * l, it = it0 < nitems ? it0 : nitems - 1;
      const int pix = fdiv(it, rcpp), ch = it - pix * cpp;
      const int ty = fdiv(pix, rTW), tx = pix - ty * TW;
      const int iy = a0 * a.is + dymin + ty, ix = b0 * a.is + dxmin + tx;
      okx |= ((iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW) ? 1u : 0u) << l;
      const int cy = iy < 0 ? 0 : (iy >= a.IH ? a.IH - 1 : iy), cx = ix < 0 ? 0 : (ix >= a.IW ? a.IW - 1 : ix);
      rx[l].ld(x + (unsigned)((cy * a.IW + cx) * Cin + 8 * ch));
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int it = tid + 256 * p;
      const int pt = it >> 3, ch = it & 7;
      const int aa = a0 + (pt >> 4), bb = b0 + (pt & 15);
      const int co = co0 + 8 * ch;
      const bool ok = aa < a.OH && bb < a.OW && co < Cout;
      oky |= (ok ? 1u : 0u) << p;
      ry[p].ld(dy + (unsigned)(((aa < a.OH ? aa : a.OH - 1) * a.OW + (bb < a.OW ? bb : a.OW - 1)) * Cout + (co < Cout ? co : 0)));
    }
  };
  auto stage = [&](RawV8<T>(&rx)[XL], RawV8<T>(&ry)[4], unsigned okx, unsigned oky) {
#pragma unroll
    for (int l = 0; l < XL; ++l) {
      const int it = tid + 256 * l;
      if (it < nitems) {
        const int pix = fdiv(it, rcpp), ch = it - pix * cpp;
        const bool ok = (okx >> l) & 1u;
        if (PRO == MDS_PRO_NONE) {
          if (!ok) rx[l].zero();
          CwCfg<T>::putraw(xs + pix * LDX + 8 * ch, rx[l]);
        } else {
          float v[8];
          rx[l].get(v);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float z = v[j] * psc[8 * ch + j] + psh[8 * ch + j];
            z = (PRO == MDS_PRO_AFFINE) ? z : siluf_(z);
            v[j] = ok ? z : 0.f;   // zero padding stays zero
          }
          CwCfg<T>::put8(xs + pix * LDX + 8 * ch, v);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int it = tid + 256 * p;
      if (!((oky >> p) & 1u)) ry[p].zero();
      CwCfg<T>::putraw(dys + (it >> 3) * LDY + 8 * (it & 7), ry[p]);
    }
  };
  auto compute = [&]() {
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
      // this 16-lane group's 8 output points: row al, columns ca..ca+3 and cb..cb+3
      const int al = 2 * s + (q >> 1), ca = 4 * (q & 1), cb = 8 + ca;
      frag_t yf[COFR];
#pragma unroll
      for (int cf = 0; cf < COFR; ++cf) yf[cf] = CwCfg<T>::frag(dys, LDY, al * 16 + ca, al * 16 + cb, 1, 16 * cf, i);
      const int prow = al * a.is * TW;
#pragma unroll
      for (int k = 0; k < NU; ++k) {
        if (uok[k]) {
          const int pb = prow + utoff[k];
          const frag_t xf = CwCfg<T>::frag(xs, LDX, pb + ca * a.is, pb + cb * a.is, a.is, ukc[k], i);
#pragma unroll
          for (int cf = 0; cf < COFR; ++cf) mma16(yf[cf], xf, acc[k][cf]);  // acc[r] = dw[co = 4q + r][ci = i]
        }
      }
    }
  };
  // two register sets when they fit (XL <= 5): the loads of tiles t+1 and t+2 fly while tile t computes
  constexpr bool TWO = XL <= 5;
  RawV8<T> rxA[XL], ryA[4], rxB[XL], ryB[4];
  unsigned oxA = 0, oyA = 0, oxB = 0, oyB = 0;
  if (TWO) {
    issue(tl, rxA, ryA, oxA, oyA);
    issue(tl + 1, rxB, ryB, oxB, oyB);
    for (; tl < tl_end; tl += 2) {
      __syncthreads();
      stage(rxA, ryA, oxA, oyA);
      __syncthreads();
      issue(tl + 2, rxA, ryA, oxA, oyA);
      compute();
      if (tl + 1 < tl_end) {
        __syncthreads();
        stage(rxB, ryB, oxB, oyB);
        __syncthreads();
        issue(tl + 3, rxB, ryB, oxB, oyB);
        compute();
      }
    }
  } else {
    issue(tl, rxA, ryA, oxA, oyA);
    for (; tl < tl_end; ++tl) {
      __syncthreads();
      stage(rxA, ryA, oxA, oyA);
      __syncthreads();
      issue(tl + 1, rxA, ryA, oxA, oyA);
      compute();
    }
  }
  // flush: per 16-output-channel slab, transpose the accumulators through LDS into the parameter's
  // OIHW order and add them with coalesced atomics (one L2 transaction per 16 lanes, not per lane)
  const int slab = Cin * a.wtaps;  // floats per output channel
#pragma unroll
  for (int cf = 0; cf < COFR; ++cf) {
    if (cf >= cofr) break;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      if (uok[k]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) flush[(4 * q + r) * slab + (ukc[k] + i) * a.wtaps + uwi[k]] = acc[k][cf][r];
      }
    }
    __syncthreads();
    float* dst = a.dw + (long)(co0 + 16 * cf) * slab;
    for (int e = tid; e < 16 * slab; e += 256) atomicAdd(dst + e, flush[e]);
  }
}

int c3w_try(const mds_conv_wgrad_args* a, mds_stream_t stream);      // k_c3.hip: the large stride-1 bf16 launches, row streaming; 1 = launched
extern "C" int mds_conv_wgrad(const mds_conv_wgrad_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->N > 0 && a->OH > 0 && a->OW > 0, "conv_wgrad: bad dims");
  MDS_REQUIRE(a->Cin % 16 == 0 && a->Cin <= 48 && a->Cout % 16 == 0, "conv_wgrad: needs Cin in {16,32,48}, Cout %% 16 (Cin=%d Cout=%d)", a->Cin, a->Cout);
  MDS_REQUIRE(a->ntaps >= 1 && a->ntaps <= MDS_MAX_TAPS && a->is >= 1, "conv_wgrad: taps/stride");
  MDS_REQUIRE(a->x && a->dyt && a->dw, "conv_wgrad: null pointer");
  MDS_REQUIRE(a->pro.mode == MDS_PRO_NONE || a->pro.mode == MDS_PRO_AFFINE || a->pro.mode == MDS_PRO_BN_SILU, "conv_wgrad: prologue mode");
  MDS_REQUIRE(a->pro.mode == MDS_PRO_NONE || (a->pro.scale && a->pro.shift), "conv_wgrad: prologue needs scale/shift");
  if (c3w_try(a, stream)) return mds_check_launch("conv_wgrad");
  int dymin, dxmin;
  const int eh = tap_extent(a->dy, a->ntaps, &dymin), ew = tap_extent(a->dx, a->ntaps, &dxmin);
  const int TH = (CV_TA - 1) * a->is + eh + 1, TW = (CV_TB - 1) * a->is + ew + 1;
  const int tiles_a = cdiv(a->OH, CV_TA), tiles_b = cdiv(a->OW, CV_TB);
  MDS_REQUIRE(TH * TW * (a->Cin / 8) <= 9 * 256, "conv_wgrad: input patch %dx%dx%d exceeds the staging registers", TH, TW, a->Cin);
  MDS_REQUIRE((long)a->IH * a->IW * a->Cin < 2147483647L && (long)a->OH * a->OW * a->Cout < 2147483647L, "conv_wgrad: one image must stay below 2^31 elements");
  const int xl = cdiv(TH * TW * (a->Cin / 8), 256);
  const long total = (long)a->N * tiles_a * tiles_b;
  const int cot = cdiv(a->Cout, CW_COT);
  // one block per CU (the accumulators take the register file); every block ends with an atomic per
  // filter value, so: two rounds of the chip for the big layers, one for the small (measured)
  long want = (total >= 4096 ? 512 : 256) / cot;   // (re-measured inside the training step: 64 / 128 lose 12 % / 2 %, 256...1024 are flat)
  if (want < 1) want = 1;
  int tpb = (int)((total + want - 1) / want);
  if (tpb < 1) tpb = 1;
  dim3 grid(cdiv(total, tpb), cot), block(256);
  MDS_REQUIRE(a->ntaps <= 9, "conv_wgrad: at most 9 taps");
  const int cifr = a->Cin >> 4;
  const size_t smem_elems = (size_t)TH * TW * cw_ldx(a->dtype, a->Cin) + 128 * cw_ldy(a->dtype);
  const size_t smem_flush = (size_t)16 * a->Cin * a->wtaps * 4 + 2 * (size_t)a->Cin * 4;
#define CW_GO5(T, PRO, CI, CO, XL_)                                                                    \
  MDS_LAUNCH((conv_wgrad_kernel<T, PRO, CI, CO, XL_>), grid, block, smem_elems * sizeof(T) + smem_flush, stream, *a, dymin, \
             dxmin, TH, TW, tiles_a, tiles_b, tpb)
#define CW_GO4(T, PRO, CI, CO) do { if (xl <= 3) CW_GO5(T, PRO, CI, CO, 3); else if (xl <= 5) CW_GO5(T, PRO, CI, CO, 5); else CW_GO5(T, PRO, CI, CO, 9); } while (0)
#define CW_GO3(T, PRO, CI) do { if (a->Cout == 16) CW_GO4(T, PRO, CI, 1); else CW_GO4(T, PRO, CI, 4); } while (0)
#define CW_GO(T, PRO) do { if (cifr == 1) CW_GO3(T, PRO, 1); else if (cifr == 2) CW_GO3(T, PRO, 2); else CW_GO3(T, PRO, 3); } while (0)
  MDS_DISPATCH_DTYPE(a->dtype, T, {
    switch (a->pro.mode) {
      case MDS_PRO_NONE: CW_GO(T, MDS_PRO_NONE); break;
      case MDS_PRO_AFFINE: CW_GO(T, MDS_PRO_AFFINE); break;
      default: CW_GO(T, MDS_PRO_BN_SILU); break;
    }
  });
#undef CW_GO4
#undef CW_GO5
#undef CW_GO3
#undef CW_GO
  return mds_check_launch("conv_wgrad");
}
