// k_pwr.hip — mds_pw_fwd for the SHORT-K, WIDE-N layers (the MBConv expansions 96->384, 112->672, 192->1152 and the
// data gradients of the matching projections): filter tile resident in LDS, block loops over row tiles.
//
// With K <= 192 a 64x128 output tile is only 24..48 MFMAs per wave, so in the general kernel (k_pw.hip: tile loop over
// N, K staged in 64-wide chunks) the per-tile overheads were as long as the arithmetic: two barriers and a filter
// re-stage per chunk, and a reduce-scatter + atomics of the column statistics per tile (the same GEMM ran 38 us with
// statistics and 28 us without).  Here a block owns ONE n-tile for its whole life:
//   * the [BN][K] filter tile is staged into LDS once; every row tile after that stages only its [64][K] rows — the
//     whole K at once, a contiguous 64*K-element chunk of x, prefetched into registers while the previous tile computes;
//   * column sums (forward statistics, or the BatchNorm-backward sums of mds_poststat_t) stay in registers across the
//     row tiles: one reduce-scatter + atomics per BLOCK;
//   * the grid is sized to be co-resident (2 blocks per CU) with gridDim.x a multiple of 8, so the nt blocks that read
//     the same row tile sit on the same XCD at about the same time and share it through that XCD's L2.
#include <stdlib.h>
#include <type_traits>
#include "gemm.h"

// TAIL: 0 = forward statistics, 1 = POST (mds_poststat_t).  NF = 16-column fragments per wave (tile = 32*NF columns:
// 128, or 96 for N = 672).  KS = ceil(K / 32) = k-steps = row-tile vectors per thread.  D = row tiles in flight.
//
// vmcnt discipline: the steady-state loop is STRAIGHT-LINE in its vector-memory instructions - every load is issued
// unconditionally from a clamped address (invalid vectors are zeroed in registers when staged), the tile index of a
// refill is clamped instead of tested, and full tiles (all 64 rows, all NF fragments) take an epilogue without row /
// column guards.  Only then can the compiler wait for "the loads issued D tiles ago" with s_waitcnt vmcnt(N > 0); with a
// branch around any load or store it falls back to vmcnt(0), which drains the ring and the output stores at every tile
// (that version ran no faster with three tiles in flight than with one).
template <typename T, int TAIL, int NF, int KS, int D>
__global__ __launch_bounds__(256, 2) void pw_fwd_wres_kernel(mds_pw_fwd_args a, int LDK, int MT) {
  MDS_CHAIN_PRIO();
  constexpr bool POST = TAIL == 1;
  constexpr int YM = POST ? 2 : 1, YN = POST ? NF : 1;
  typedef typename Frag<T>::type frag_t;
  constexpr int BM = 64, BN = 32 * NF;
  MDS_DYN_SMEM(smem);
  T* ws = (T*)smem;                        // [BN][LDK]  filter tile, resident
  T* xs = ws + BN * LDK;                   // [BM][LDK]  current row tile
  float* pbn = (float*)(xs + BM * LDK);    // POST: [4][BN] scale, shift, mean, rstd of the tile's columns
  // bf16 output staging, one private [32][SP] region per wave: the MFMA layout holds 4 columns (8 bytes) of 16 different
  // rows per lane, and stores issued that way wrote 32-byte pieces - every GEMM-shaped kernel of this library plateaued at
  // ~2.9 TB/s of output while fill reaches 6.9.  Through LDS a lane stores 16 bytes and a row segment is contiguous.
  constexpr bool STG = sizeof(T) == 2;
  constexpr int SP = 16 * NF + 8, VPW = 2 * NF;
  T* stg = (T*)(pbn + (POST ? 4 * BN : 0)) + (threadIdx.x >> 6) * 32 * SP;

  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int K = a.K, N = a.N, G = gridDim.x;
  const int n0 = blockIdx.y * BN;
  const T* x = (const T*)a.x;
  const T* w = (const T*)a.w;
  T* y = (T*)a.y;
  const int VPR = K >> 3, nvec = BM * VPR;
  int nfr = (N - n0 - 16 * NF * wn) >> 4;   // valid 16-column fragments of this wave
  nfr = nfr < 0 ? 0 : (nfr > NF ? NF : nfr);
  const int MTF = (int)(a.M / BM);           // full row tiles

  // this thread's vectors of a row tile: the tile is one contiguous chunk of x, vector v = tid + 256 j
  int lrow[KS], loff[KS], goff[KS];
  {
    int row = tid / VPR, kv = tid - row * VPR;
    const int drow = 256 / VPR, dkv = 256 - drow * VPR;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const bool in = tid + 256 * j < nvec;
      lrow[j] = in ? row : BM;               // BM: never a valid row
      loff[j] = row * LDK + 8 * kv;
      goff[j] = in ? 8 * (tid + 256 * j) : 0;
      row += drow; kv += dkv;
      if (kv >= VPR) { kv -= VPR; ++row; }
    }
  }
  RawV8<T> rx[D][KS];
  RawV4<T> rys[D][YM][YN];
  auto issue = [&](int mt, RawV8<T> (&r)[KS]) {   // unconditional loads; vectors past the tile / past M read element 0 of the tile
    mt = mt < MT ? mt : MT - 1;
    const long m0 = (long)mt * BM;
    const T* xb = x + m0 * K;
    const int rows = (int)(a.M - m0 < BM ? a.M - m0 : BM);
#pragma unroll
    for (int j = 0; j < KS; ++j) r[j].ld(xb + (lrow[j] < rows ? goff[j] : 0));
  };
  auto issue_ys = [&](int mt, RawV4<T> (&r)[YM][YN]) {   // post.y fragments of a tile (clamped rows / columns: finite values)
    if (!POST) return;
    mt = mt < MT ? mt : MT - 1;
#pragma unroll
    for (int mf = 0; mf < YM; ++mf) {
      const long m = (long)mt * BM + 32 * wm + 16 * mf + i;
      const T* ysrow = (const T*)a.post.y + (m < a.M ? m : 0) * N + n0 + 16 * NF * wn + 4 * q;
#pragma unroll
      for (int nf = 0; nf < YN; ++nf) r[mf][nf].ld(ysrow + (nf < nfr ? 16 * nf : 0) - (nfr == 0 ? 16 * NF * wn : 0));
    }
  };
  const int mt0 = blockIdx.x;
#pragma unroll
  for (int d = 0; d < D; ++d) issue(mt0 + d * G, rx[d]);
  issue_ys(mt0, rys[0]);

  // LDS: zero everything once (the k-padding up to 32*KS and the filter rows past N must read as zeros), then the filter
  {
    const int tot16 = (int)((BN + BM) * LDK * sizeof(T) / 16);
    f32x4* z = (f32x4*)smem;
    for (int e = tid; e < tot16; e += 256) z[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  {
    int row = tid / VPR, kv = tid - row * VPR;
    const int drow = 256 / VPR, dkv = 256 - drow * VPR;
    for (int v = tid; v < BN * VPR; v += 256) {
      const int n = n0 + row;
      if (n < N) {
        RawV8<T> r;
        r.ld(w + (long)n * K + 8 * kv);
        r.st(ws + row * LDK + 8 * kv);
      }
      row += drow; kv += dkv;
      if (kv >= VPR) { kv -= VPR; ++row; }
    }
    if (POST && tid < BN) {
      const int n = n0 + tid;
#pragma unroll
      for (int k = 0; k < 4; ++k) pbn[k * BN + tid] = n < N ? a.post.bn[(long)k * N + n] : 0.f;
    }
  }

  float ps[16], pss[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) { ps[e] = 0.f; pss[e] = 0.f; }
  const int xrow0 = (32 * wm + i) * LDK + 8 * q, wrow0 = (16 * NF * wn + i) * LDK + 8 * q;

  // FULL: all 64 rows valid and all NF fragments of this wave inside N - no guards anywhere in the tile
  auto tile = [&](auto full_tag, int mt, RawV8<T> (&rt)[KS], RawV4<T> (&ysc)[YM][YN], RawV4<T> (&ysn)[YM][YN]) {
    constexpr bool FULL = decltype(full_tag)::value;
    const long m0 = (long)mt * BM;
    __syncthreads();   // the previous tile's fragment reads are done (first trip: the zero fill / filter stage is ordered)
    {
      const int rows = FULL ? BM : (int)(a.M - m0 < BM ? a.M - m0 : BM);
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        if (!FULL && lrow[j] >= rows) rt[j].zero();
        if (lrow[j] < BM) rt[j].st(xs + loff[j]);
      }
    }
    __syncthreads();
    issue(mt + D * G, rt);        // refill this ring slot (tile index clamped: the tail re-reads the last tile)
    issue_ys(mt + G, ysn);

    f32x4 acc[2][NF];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      frag_t xf[2];
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) xf[mf] = ld_frag(xs + xrow0 + 16 * mf * LDK + 32 * ks);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        frag_t wf = ld_frag(ws + wrow0 + 16 * nf * LDK + 32 * ks);
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) mma16(wf, xf[mf], acc[mf][nf]);   // acc[r] = y[m = i][n = 4q + r]
      }
    }

#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      const long m = m0 + 32 * wm + 16 * mf + i;
      const bool ok = FULL || m < a.M;
      const long rowoff = (ok ? m : 0) * N + n0 + 16 * NF * wn + 4 * q;
      float v[NF][4];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[nf][r] = acc[mf][nf][r];
      if (POST) {   // u of the next BatchNorm backward is this tile (rows past M: v == 0 -> g == 0)
        const float mk = (a.post.mode == MDS_POST_MASK) ? a.post.mask[(unsigned)(ok ? m : 0) / (unsigned)a.post.rows_per_group] : 1.0f;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          if (FULL || nf < nfr) {
            const float ys[4] = {ysc[mf][nf].get(0), ysc[mf][nf].get(1), ysc[mf][nf].get(2), ysc[mf][nf].get(3)};
            const float* pc = pbn + 16 * NF * wn + 16 * nf + 4 * q;
            const f32x4 mu = *(const f32x4*)(pc + 2 * BN), rs = *(const f32x4*)(pc + 3 * BN);
            if (a.post.mode == MDS_POST_SILU) {
              const f32x4 sc = *(const f32x4*)pc, sh = *(const f32x4*)(pc + BN);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[nf][r] *= silu_gradf_(ys[r] * sc[r] + sh[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float g = Elem<T>::rnd(v[nf][r]) * mk;
              ps[nf * 4 + r] += g;
              pss[nf * 4 + r] += g * ((ys[r] - mu[r]) * rs[r]);
            }
          }
        }
      }
      if (FULL && STG) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) store4(stg + (16 * mf + i) * SP + 16 * nf + 4 * q, v[nf]);
      } else if (FULL) {
        T* yrow = y + rowoff;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) store4(yrow + 16 * nf, v[nf]);
      } else if (ok) {
        T* yrow = y + rowoff;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
          if (nf < nfr) store4(yrow + 16 * nf, v[nf]);
      }
      if (!POST) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)   // rows past M and columns past N hold zeros (zero-filled operands)
#pragma unroll
          for (int r = 0; r < 4; ++r) { ps[nf * 4 + r] += v[nf][r]; pss[nf * 4 + r] += v[nf][r] * v[nf][r]; }
      }
    }
    if (FULL && STG) {   // the wave's 32 x 16NF outputs, 16 bytes per lane, row segments contiguous
      wave_lds_sync();
      T* ybase = y + (m0 + 32 * wm) * N + n0 + 16 * NF * wn;
#pragma unroll
      for (int t = 0; t < (32 * VPW) / 64; ++t) {
        const int vv = lane + 64 * t, row = vv / VPW, c = vv - row * VPW;
        RawV8<T> o;
        o.ld(stg + row * SP + 8 * c);
        o.st(ybase + (long)row * N + 8 * c);
      }
      wave_lds_sync();   // the next tile's staging writes come after these reads
    }
  };
  int base = mt0;
  if (nfr == NF) {   // steady state: D full tiles per trip, no branch inside
    for (; base + (D - 1) * G < MTF; base += D * G) {
#pragma unroll
      for (int d = 0; d < D; ++d) tile(std::true_type{}, base + d * G, rx[d], rys[d], rys[(d + 1) % D]);
    }
  }
  for (; base < MT; base += D * G) {   // the last trips, the ragged last row tile, blocks on a ragged n-tile
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (base + d * G < MT) tile(std::false_type{}, base + d * G, rx[d], rys[d], rys[(d + 1) % D]);
  }

  if (POST ? (a.post.stats != nullptr) : (a.stats != nullptr)) {   // once per block: every lane ends up with one column's partial sums
    const int e = reduce_scatter16(ps, i);
    reduce_scatter16(pss, i);
    const int n = n0 + 16 * NF * wn + 16 * (e >> 2) + 4 * q + (e & 3);
    if ((e >> 2) < nfr && n < N) {
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

// Launches the filter-resident kernel when the layer qualifies; returns 1 when it does not (the caller falls through to
// the general kernel).  Arguments were validated by mds_pw_fwd.
int pw_fwd_wres_try(const mds_pw_fwd_args* a, mds_stream_t stream) {
  const int knob = mds_knob(MDS_KNOB_PW_WRES);
  if (knob == 1) return 1;
  const bool forced = knob == 2;                    // tests: take this kernel at any M, 8 blocks per n-tile
  // (the POST form of this kernel - BatchNorm-backward sums of the next layer in the epilogue - needed 137-667 spilled VGPRs and
  //  no layer of the network reaches it: post statistics belong to block-INPUT gradients, whose N is narrow.  General kernel.)
  if (a->pro.mode != MDS_PRO_NONE || a->epi.mode != MDS_EPI_NONE || a->residual || a->post.mode != MDS_POST_NONE) return 1;
  const int K = a->K, N = a->N;
  const int KS = (K + 31) / 32;
  if (K < 72 || (KS != 3 && KS != 4 && KS != 6) || N < 128) return 1;
  const int NF = (N % 128 != 0 && N % 96 == 0) ? 3 : 4;
  const int BN = 32 * NF;
  const size_t esz = a->dtype == MDS_BF16 ? 2 : 4;
  size_t pitch = (size_t)KS * 32 * esz;
  if ((pitch / 32) % 2 == 0) pitch += 32;           // 32 B x odd: conflict-free ds_read_b128 lane groups
  const bool post = false;
  const size_t smem = (size_t)(BN + 64) * pitch + (post ? 4 * BN * sizeof(float) : 0) + (esz == 2 ? 4 * 32 * (16 * NF + 8) * 2 : 0);
  if (smem > 160 * 1024) return 1;
  const int bpc = smem <= 80 * 1024 ? 2 : 1;        // co-resident blocks per CU
  const long MT = cdiv(a->M, 64);
  const int nt = cdiv(N, BN);
  int cap = forced ? 8 : (256 * bpc / nt) & ~7;     // (three co-resident blocks measured 5 % slower than two)
  if (cap < 8) cap = 8;
  // too few row tiles per block to amortise the filter stage: measured inside the step, the 18400-row layers
  // (192 -> 1152, 192 -> 576: 5 tiles per block) lose 2-3 us to it, the 73600-row layers gain 10-30 %
  if (!forced && MT < (knob == 3 ? 2L : 6L) * (512 / nt)) return 1;   // (knob 3: the lower bar, for A/B runs)
  const long tpb = cdiv(MT, cap);
  int gx = (int)((cdiv(MT, tpb) + 7) & ~7L);
  if (gx > cap) gx = cap;
  if (gx > MT) gx = (int)MT;
  const int LDK = (int)(pitch / esz);
  dim3 grid(gx, nt), block(256);
#define PWR_GO3(T, TAIL_, NF_, KS_) MDS_LAUNCH((pw_fwd_wres_kernel<T, TAIL_, NF_, KS_, (sizeof(T) == 4 && KS_ == 6) ? 2 : 3>), grid, block, smem, stream, *a, LDK, (int)MT)   /* fp32 K = 192: two row tiles in flight (three spilled 35-103 VGPRs) */
#define PWR_GO2(T, TAIL_, NF_) do { if (KS == 3) PWR_GO3(T, TAIL_, NF_, 3); else if (KS == 4) PWR_GO3(T, TAIL_, NF_, 4); else PWR_GO3(T, TAIL_, NF_, 6); } while (0)
#define PWR_GO(T, TAIL_) do { if (NF == 3) PWR_GO2(T, TAIL_, 3); else PWR_GO2(T, TAIL_, 4); } while (0)
  MDS_DISPATCH_DTYPE(a->dtype, T, {
    PWR_GO(T, 0);
  });
#undef PWR_GO
#undef PWR_GO2
#undef PWR_GO3
  return mds_check_launch("pw_fwd");
}
