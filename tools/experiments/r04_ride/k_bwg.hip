// k_bwg.hip — BatchNorm-backward apply pass with a 1x1 weight gradient riding on it (include/mds.h: mds_bn_bwd_apply_wg).
//
// Why: in the backward of an inverted-residual block (reference: /root/reference/src/models/multidim_stacker.py:124-134 and
// the timm twin) the two 1x1 weight gradients each re-read a WIDE tensor ([M][mid], 42-300 MB) that a streaming pass of the
// BatchNorm backward has just had in registers:
//   BN1 apply   forms dy1[M][mid]                      = the wide operand of dW(conv_pw)  = dy1^T x       (x: block input, narrow)
//   BN2 apply   reads y2[M][mid], evaluates sigmoid(z2) -> silu(z2)*gate is one multiply away
//                                                      = the wide operand of dW(conv_pwl) = (a2 gate)^T dy3   (dy3: narrow)
// A block owns (row slab, CW-channel chunk of the wide tensor): every wide element is loaded once, dy is stored exactly as
// mds_bn_bwd_apply stores it, and the 64 x CW tile goes through LDS into the matrix cores against the slab's rows of the
// narrow operand (K <= 192 columns = the whole narrow width; the C / CW chunk blocks of a slab sit on one XCD and re-read
// those rows from its L2).  Roofline: HBM - 3 wide passes (g, y in; dy out), 48 KB per 64-row step and block against 24 MFMAs
// per wave (~0.2 us); the accumulator tile leaves once per block as a plain store into part[slab] (no atomics), and
// mds_wg_finish adds the slabs in order on the second stream.
//
// Pipeline: a ring of register sets for the wide pair (3-4 steps requested ahead) + one set for the narrow operand; the
// steady-state loop is branch-free (full 64-row steps only, refills from clamped step indices) so that the compiler can
// count its vmcnt waits ("a register ring only exists if the compiler can count it"); the ragged last step of a
// slab runs once, guarded, after the loop.
#include "gemm.h"

#define BWG_ROWS 64

// LDS row pitch (elements): 32 B x odd, the conflict-free family of ds_read_b64_tr_b16 / ds_read_b128
static constexpr int bwg_pitch(int w, int esz) {
  int p = w + 16 / esz;
  while (((p * esz / 32) & 1) == 0 || (p * esz) % 32 != 0) p += 16 / esz;
  return p;
}

template <typename T> struct BwgFrag;
template <> struct BwgFrag<bf16_t> {
  // fragment of 32 rows (the MFMA k index) x 16 columns starting at `col16`: lane (i, q) gets column i of rows ra..ra+3, ra+8..ra+11
  static MDS_DEV u16x8 ld(const bf16_t* tile, int pitch, int ks, int col16, int i, int q) {
    const int ra = 16 * (q >> 1) + 4 * (q & 1), r0 = 32 * ks + ra + (i >> 2);
    const bf16_t* p = tile + col16 + 4 * (i & 3);
    const u16x4 lo = lds_tr4(p + r0 * pitch), hi = lds_tr4(p + (r0 + 8) * pitch);
    return (u16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
};
template <> struct BwgFrag<float> {
  // exact-fp32 MFMA j consumes element j of every lane: lane (i, q) supplies rows 8q + j (any k permutation is legal as long
  // as both operands use it)
  static MDS_DEV f32x8 ld(const float* tile, int pitch, int ks, int col16, int i, int q) {
    const float* p = tile + (32 * ks + 8 * q) * pitch + col16 + i;
    return (f32x8){p[0], p[pitch], p[2 * pitch], p[3 * pitch], p[4 * pitch], p[5 * pitch], p[6 * pitch], p[7 * pitch]};
  }
};

template <typename T> MDS_DEV void bwg_st8(T* gp, T* lp, const float (&dv)[8], const float (&wv)[8], bool same);
template <> MDS_DEV void bwg_st8<bf16_t>(bf16_t* gp, bf16_t* lp, const float (&dv)[8], const float (&wv)[8], bool same) {
  const u16x8 d = pack8(dv);
  if (gp) *(u16x8*)gp = d;
  *(u16x8*)lp = same ? d : pack8(wv);
}
template <> MDS_DEV void bwg_st8<float>(float* gp, float* lp, const float (&dv)[8], const float (&wv)[8], bool same) {
  if (gp) store8(gp, dv);
  store8(lp, same ? dv : wv);
}

// Wave roles.  vmcnt retires in order, so a wave that waits for ANY load has waited for every older one: a wave that both
// streams the wide pair (HBM: the bytes that must stay in flight) and fetches the narrow rows (L2 hits, consumed one step
// later) can never hold more than one step of wide rows outstanding (first version: 3.3 TB/s; a six-deep wide ring in the
// same waves: no better - every wait for the narrow rows drained it).  So the block is two groups of four waves with their
// own counters: the WIDE waves keep a ring of NS steps of (g, y) rows requested ahead, form dy, store it and stage the wide
// operand; the NARROW waves fetch and stage the narrow operand's rows one step ahead; after the barrier all eight waves
// multiply (wave = (half of the wide fragments, every fourth narrow fragment)).  Both groups execute the same barrier sequence.
template <typename T, int CW, int K, bool SE>
__global__ __launch_bounds__(512, (sizeof(T) == 4 || SE) ? 1 : 2) void bn_bwd_apply_wg_kernel(mds_bn_bwd_apply_wg_args a, int nchunks, int splits, int rows_per_slab, long gr) {
  constexpr int ES = sizeof(T);
  constexpr int LW = bwg_pitch(CW, ES), LX = bwg_pitch(K, ES);
  constexpr int WCH = CW / 8, WR = (256 / WCH) >= 32 ? 32 : 16, WP = BWG_ROWS / WR;   // 8-channel chunks per row, rows per pass, passes
  constexpr bool ALLW = WCH * WR == 256;
  constexpr int XV = K / 8, XP = (XV + 3) / 4;
  constexpr int NFW = CW / 16, NFH = NFW / 2, KF = K / 16, KFW = (KF + 3) / 4;
  constexpr int NS = ES == 2 ? (CW == 64 ? 4 : 2) : (CW == 64 ? 2 : 1);      // wide ring depth (registers: NS * WP * 2 vectors)
  typedef typename Frag<T>::type frag_t;
  MDS_DYN_SMEM(smem);
  T* ws = (T*)smem;                 // [64][LW]  the wide operand of the weight gradient (dy, or silu(z)*gate)
  T* xs = ws + BWG_ROWS * LW;       // [64][LX]  the narrow operand's rows
  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6), role = MDS_UNIFORM(tid >> 8), t = tid & 255;
  const int i = lane & 15, q = lane >> 4;
  // block -> (slab, chunk): the chunk blocks of one slab are consecutive on ONE XCD (they share the narrow operand's rows)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int chunk = slot % nchunks, slab = (slot / nchunks) * 8 + xcd;
  if (slab >= a.slabs) return;      // (slabs are rounded up to a multiple of 8 for the mapping; whole block, before any barrier)
  const int grp = slab / splits;
  const long mbeg = (long)grp * gr + (long)(slab % splits) * rows_per_slab;
  long mend = mbeg + rows_per_slab;
  if (mend > (long)(grp + 1) * gr) mend = (long)(grp + 1) * gr;
  const int C = a.C;
  const long nrows = mend - mbeg;
  const int nfull = (int)(nrows / BWG_ROWS), nrem = (int)(nrows - (long)nfull * BWG_ROWS);

  const int wh = (wave >> 2) & 1, wx = wave & 3;   // this wave's MFMA share: wide fragments [wh * NFH, wh * NFH + NFH), narrow fragments wx + 4 jv
  f32x4 acc[NFH][KFW];
#pragma unroll
  for (int u = 0; u < NFH; ++u)
#pragma unroll
    for (int v = 0; v < KFW; ++v) acc[u][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto mfma_step = [&]() {
#pragma unroll
    for (int ks = 0; ks < BWG_ROWS / 32; ++ks) {
      // every fragment of the k-step is requested before the first MFMA (one LDS latency per k-step, not one per operand)
      frag_t wf[NFH], xf[KFW];
#pragma unroll
      for (int u = 0; u < NFH; ++u) wf[u] = BwgFrag<T>::ld(ws, LW, ks, 16 * (wh * NFH + u), i, q);
#pragma unroll
      for (int jv = 0; jv < KFW; ++jv) {
        const int v = wx + 4 * jv;
        xf[jv] = BwgFrag<T>::ld(xs, LX, ks, 16 * ((KF % 4 == 0 || v < KF) ? v : 0), i, q);
      }
#pragma unroll
      for (int jv = 0; jv < KFW; ++jv) {
        if (KF % 4 == 0 || wx + 4 * jv < KF) {       // wave-uniform
#pragma unroll
          for (int u = 0; u < NFH; ++u) mma16(wf[u], xf[jv], acc[u][jv]);   // acc[r] = P[c = 16(wh NFH + u) + 4q + r][k = 16v + i]
        }
      }
    }
  };

  if (role == 0) {
    // ------------------------------------------------------------ WIDE waves
    const int wc = t % WCH, wr = t / WCH;
    const bool wact = ALLW || wr < WR;
    const int c0 = chunk * CW + 8 * (wact ? wc : 0);
    float cA[8], cB[8], cD[8], sc[8], sh[8], ga[8], dp[8];
    load8f(a.lin + c0, cA); load8f(a.lin + C + c0, cB); load8f(a.lin + 2 * C + c0, cD);
    if (SE) {
      load8f(a.bn + c0, sc); load8f(a.bn + C + c0, sh);
      load8f(a.g.gate + (long)grp * C + c0, ga); load8f(a.g.dpooled + (long)grp * C + c0, dp);
    }
    // one row of the wide pair -> dy (stored) and the weight gradient's wide operand
    auto form = [&](const float (&u)[8], const float (&yv)[8], float (&dv)[8], float (&wv)[8]) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (SE) {
          const float z = yv[j] * sc[j] + sh[j], s = sigmoidf_(z);
          const float gg = (u[j] * ga[j] + dp[j]) * (s * (1.0f + z * (1.0f - s)));
          dv[j] = cA[j] * gg + cB[j] * yv[j] + cD[j];
          wv[j] = z * s * ga[j];
        } else {
          dv[j] = cA[j] * u[j] + cB[j] * yv[j] + cD[j];
        }
      }
    };
    const int wstride = WR * C;
    const int wlds = (wact ? wr : 0) * LW + 8 * wc;
    struct Wide { RawV8<T> g[WP], y[WP]; };
    const long wbase = (mbeg + (wact ? wr : 0)) * C + c0;
    const T* pg = (const T*)a.g.u + wbase;
    const T* py = (const T*)a.y + wbase;
    T* pd = (T*)a.dy + wbase;           // running pointer of the NEXT step to store
    int wreq = 0;                       // next step to request (clamped to the last full step: re-requested rows are never consumed)
    auto issue_w = [&](Wide& R) {
      const long o = (long)(wreq < nfull ? wreq : nfull - 1) * BWG_ROWS * C;
      ++wreq;
      if (wact) {
#pragma unroll
        for (int p = 0; p < WP; ++p) { R.g[p].ld(pg + o + p * wstride); R.y[p].ld(py + o + p * wstride); }
      }
    };
    auto step = [&](Wide& R) {
      __syncthreads();     // the previous step's fragment reads are done
      if (wact) {
#pragma unroll
        for (int p = 0; p < WP; ++p) {
          float u[8], yv[8], dv[8], wv[8];
          R.g[p].get(u); R.y[p].get(yv);
          form(u, yv, dv, wv);
          bwg_st8<T>(pd + p * wstride, ws + wlds + p * WR * LW, dv, wv, !SE);
        }
      }
      issue_w(R);          // this set's registers are free again: request step s + NS
      pd += (long)BWG_ROWS * C;
      __syncthreads();
      mfma_step();
    };
    if (nfull > 0) {
      Wide W[NS];
#pragma unroll
      for (int r = 0; r < NS; ++r) issue_w(W[r]);
      int s = 0;
      for (; s + NS <= nfull; s += NS) {
#pragma unroll
        for (int r = 0; r < NS; ++r) step(W[r]);
      }
#pragma unroll
      for (int r = 0; r < NS - 1; ++r)
        if (s + r < nfull) step(W[r]);
    }
    if (nrem > 0) {   // the ragged last step of the slab: guarded, not pipelined
      const long mb = mbeg + (long)nfull * BWG_ROWS;
      __syncthreads();
      if (wact) {
#pragma unroll
        for (int p = 0; p < WP; ++p) {
          const int lr = wr + WR * p;
          const bool ok = lr < nrem;
          const long off = (mb + (ok ? lr : 0)) * C + c0;
          float u[8], yv[8], dv[8], wv[8];
          load8((const T*)a.g.u + off, u); load8((const T*)a.y + off, yv);
          form(u, yv, dv, wv);
          if (!ok) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { dv[j] = 0.f; wv[j] = 0.f; }
          }
          bwg_st8<T>(ok ? (T*)a.dy + off : (T*)nullptr, ws + wlds + p * WR * LW, dv, wv, !SE);
        }
      }
      __syncthreads();
      mfma_step();
    }
  } else {
    // ------------------------------------------------------------ NARROW waves: four threads per row, thread j takes the
    // 16-byte vectors j, j + 4, ... of its row (one base address + immediate offsets); vectors past the row alias the previous
    // one of the same thread (same data to the same LDS address: harmless, and the loop stays branch-free)
    const int xr_ = t >> 2, xj = t & 3;
    const int xlast = (xj + 4 * (XP - 1) < XV) ? 32 * (XP - 1) : 32 * (XP - 2);    // element offset of the last vector (XV % 4 != 0: an alias)
    const int xoff0 = xr_ * K + 8 * xj, xlds0 = xr_ * LX + 8 * xj;
    const T* px = (const T*)a.x + mbeg * K;
    int xreq = 0;
    struct Nar { RawV8<T> v[XP]; };
    auto issue_x = [&](Nar& R) {
      const long o = (long)(xreq < nfull ? xreq : nfull - 1) * BWG_ROWS * K + xoff0;
      ++xreq;
#pragma unroll
      for (int p = 0; p < XP; ++p) R.v[p].ld(px + o + (p == XP - 1 ? xlast : 32 * p));
    };
    auto step = [&](Nar& R) {
      __syncthreads();
#pragma unroll
      for (int p = 0; p < XP; ++p) R.v[p].st(xs + xlds0 + (p == XP - 1 ? xlast : 32 * p));
      issue_x(R);          // two steps ahead
      __syncthreads();
      mfma_step();
    };
    if (nfull > 0) {
      Nar X[2];
      issue_x(X[0]); issue_x(X[1]);
      int s = 0;
      for (; s + 2 <= nfull; s += 2) { step(X[0]); step(X[1]); }
      if (s < nfull) step(X[0]);
    }
    if (nrem > 0) {
      const long mb = mbeg + (long)nfull * BWG_ROWS;
      __syncthreads();
#pragma unroll
      for (int p = 0; p < XP; ++p) {
        const bool ok = xr_ < nrem;
        const int e = (p == XP - 1 ? xlast : 32 * p);
        RawV8<T> r;
        r.ld((const T*)a.x + mb * K + (ok ? xoff0 : 8 * xj) + e);
        if (!ok) r.zero();
        r.st(xs + xlds0 + e);
      }
      __syncthreads();
      mfma_step();
    }
  }

  float* part = a.part + ((long)slab * C + (long)chunk * CW) * K;
#pragma unroll
  for (int u = 0; u < NFH; ++u)
#pragma unroll
    for (int jv = 0; jv < KFW; ++jv) {
      const int v = wx + 4 * jv;
      if (KF % 4 == 0 || v < KF) {
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(long)(16 * (wh * NFH + u) + 4 * q + r) * K + 16 * v + i] = acc[u][jv][r];
      }
    }
}

// slab geometry shared by the launcher and mds_bn_bwd_apply_wg_slabs
struct BwgGeo { int cw, nchunks, groups, splits, rows_per_slab, slabs; long gr; };
static BwgGeo bwg_geo(long M, int C, long group_rows, bool se, int dtype) {
  BwgGeo g;
  g.cw = (C % 64 == 0) ? 64 : 96;
  g.nchunks = C / g.cw;
  g.gr = group_rows > 0 ? group_rows : M;
  g.groups = (int)(M / g.gr);
  const int target = mds_knob(MDS_KNOB_BWG_BLOCKS) > 0 ? mds_knob(MDS_KNOB_BWG_BLOCKS)   : ((se || dtype == MDS_F32) ? 256 : 512);   // one resident round: 1 (SE / fp32) or 2 blocks per CU
  int want = target / g.nchunks;                       // slabs in all
  if (want < 1) want = 1;
  int splits = (want + g.groups / 2) / g.groups;       // per group
  if (splits < 1) splits = 1;
  long rps = (g.gr + splits - 1) / splits;
  rps = (rps + BWG_ROWS - 1) / BWG_ROWS * BWG_ROWS;
  g.rows_per_slab = (int)rps;
  g.splits = (int)((g.gr + rps - 1) / rps);
  g.slabs = g.groups * g.splits;
  return g;
}
static bool bwg_dims_ok(long M, int C, int K, long group_rows) {
  return M > 0 && C > 0 && (C % 64 == 0 || C % 96 == 0) && (K == 48 || K == 96 || K == 112 || K == 192) && group_rows >= 0 &&
         (group_rows == 0 || M % group_rows == 0) && M < 2147483647L;
}
extern "C" int mds_bn_bwd_apply_wg_slabs(long M, int C, int K, long group_rows, int wide_act, int dtype) {
  if (!bwg_dims_ok(M, C, K, group_rows)) return 0;
  return bwg_geo(M, C, group_rows, wide_act == 1, dtype).slabs;
}

extern "C" int mds_bn_bwd_apply_wg(const mds_bn_bwd_apply_wg_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && bwg_dims_ok(a->M, a->C, a->K, a->group_rows), "bn_bwd_apply_wg: bad dims (C a multiple of 64 or 96; K in 48, 96, 112, 192)");
  MDS_REQUIRE(a->g.u && a->y && a->bn && a->lin && a->dy && a->x && a->part, "bn_bwd_apply_wg: null pointer");
  MDS_REQUIRE(a->g.mode == MDS_G_PLAIN || a->g.mode == MDS_G_SE_SILU, "bn_bwd_apply_wg: gradient source must be PLAIN or SE_SILU");
  const bool se = a->g.mode == MDS_G_SE_SILU;
  MDS_REQUIRE(se == (a->wide_act == 1) && (a->wide_act == 0 || a->wide_act == 1), "bn_bwd_apply_wg: wide_act 1 goes with SE_SILU (its gate), 0 with PLAIN");
  MDS_REQUIRE(!se || (a->g.gate && a->g.dpooled && a->g.rows_per_group > 0 && a->group_rows == a->g.rows_per_group),
              "bn_bwd_apply_wg: SE_SILU needs gate, dpooled and group_rows == rows_per_group");
  const BwgGeo g = bwg_geo(a->M, a->C, a->group_rows, se, a->dtype);
  MDS_REQUIRE(a->slabs == g.slabs, "bn_bwd_apply_wg: slabs must come from mds_bn_bwd_apply_wg_slabs");
  const dim3 grid((unsigned)((g.slabs + 7) / 8 * 8 * g.nchunks)), block(512);
#define BWG_GO(T, CW, K_, SE_)                                                                                     \
  MDS_LAUNCH((bn_bwd_apply_wg_kernel<T, CW, K_, SE_>), grid, block, (size_t)BWG_ROWS * (bwg_pitch(CW, sizeof(T)) + bwg_pitch(K_, sizeof(T))) * sizeof(T), \
             stream, *a, g.nchunks, g.splits, g.rows_per_slab, g.gr)
#define BWG_K(T, CW, SE_)                                                                \
  do {                                                                                   \
    switch (a->K) {                                                                      \
      case 48: BWG_GO(T, CW, 48, SE_); break;                                            \
      case 96: BWG_GO(T, CW, 96, SE_); break;                                            \
      case 112: BWG_GO(T, CW, 112, SE_); break;                                          \
      default: BWG_GO(T, CW, 192, SE_); break;                                           \
    }                                                                                    \
  } while (0)
#define BWG_CW(T, SE_) do { if (g.cw == 64) BWG_K(T, 64, SE_); else BWG_K(T, 96, SE_); } while (0)
  MDS_DISPATCH_DTYPE(a->dtype, T, do { if (se) BWG_CW(T, true); else BWG_CW(T, false); } while (0));
  return mds_check_launch("bn_bwd_apply_wg");
}

// dw (+)= sum_s part[s][C][K], slabs added in order (deterministic); four partial sums in flight per thread
__global__ __launch_bounds__(256) void wg_finish_kernel(mds_wg_finish_args a) {
  const long n = (long)a.C * a.K, idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const float* p = a.part + idx;
  int s = 0;
  for (; s + 4 <= a.slabs; s += 4) {
    const float v0 = p[(long)s * n], v1 = p[(long)(s + 1) * n], v2 = p[(long)(s + 2) * n], v3 = p[(long)(s + 3) * n];
    s0 += v0; s1 += v1; s2 += v2; s3 += v3;
  }
  for (; s < a.slabs; ++s) s0 += p[(long)s * n];
  const float t = (s0 + s1) + (s2 + s3);
  const long dst = a.transpose ? (idx % a.K) * (long)a.C + idx / a.K : idx;
  a.dw[dst] += t;
}
extern "C" int mds_wg_finish(const mds_wg_finish_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->C > 0 && a->K > 0 && a->slabs > 0 && a->part && a->dw, "wg_finish: bad args");
  MDS_LAUNCH(wg_finish_kernel, dim3(cdiv((long)a->C * a->K, 256)), dim3(256), 0, stream, *a);
  return mds_check_launch("wg_finish");
}
