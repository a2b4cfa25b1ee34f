// k_pw.hip — 1x1 convolutions as MFMA GEMMs over channels-last rows.
//
//   mds_pw_fwd   : y[M][N] = pro(x)[M][K] * w[N][K]^T (+residual) (+per-channel sum/sumsq of y)
//   mds_pw_wgrad : dw[N][K] += dy[M][N]^T * pro(x)[M][K]
//
// Roofline: these GEMMs are skinny (K, N <= 1152, M up to 4.7 M rows): bytes/row = 2(K+N),
// flops/row = 2KN -> 25..165 FLOP/B, below the 312 FLOP/B ridge of MI355X, i.e. HBM-bound; the
// design goal is one pass over x and y with 16-byte accesses and MFMA work hidden behind it.
#include "gemm.h"

#define PW_BM 128   // rows per block tile
#define PW_BNT 128  // output channels per n-tile (8 MFMA column fragments)

template <typename T, int PRO>
__global__ __launch_bounds__(256) void pw_fwd_kernel(mds_pw_fwd_args a) {
  typedef typename Frag<T>::type frag_t;
  const int LD = PwLd<T>::v;
  MDS_DYN_SMEM(smem);
  T* xs = (T*)smem;                         // [PW_BM][LD]
  T* ws = xs + PW_BM * LD;                  // [PW_BNT][LD]
  float* psc = (float*)(ws + PW_BNT * LD);  // [Kpad] scale
  const int Kpad = (a.K + 31) & ~31;
  float* psh = psc + Kpad;                  // [Kpad] shift
  float* st_s = psh + Kpad;                 // [PW_BNT]
  float* st_ss = st_s + PW_BNT;             // [PW_BNT]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const long m0 = (long)blockIdx.x * PW_BM;
  const int K = a.K, N = a.N;
  const T* x = (const T*)a.x;
  const T* w = (const T*)a.w;
  T* y = (T*)a.y;

  if (PRO != MDS_PRO_NONE) {
    for (int k = tid; k < Kpad; k += 256) {
      psc[k] = k < K ? a.pro.scale[k] : 0.f;
      psh[k] = k < K ? a.pro.shift[k] : 0.f;
    }
  }
  // staging coordinates: 4 threads per row (8 k each), 64 rows per pass, 2 passes
  const int schunk = tid & 3, srow = tid >> 2;

  // n-tiles are spread over gridDim.y when there are too few row tiles to fill 256 CUs
  for (int n0 = blockIdx.y * PW_BNT; n0 < N; n0 += gridDim.y * PW_BNT) {
    const int nfr = (N - n0 >= PW_BNT) ? 8 : ((N - n0) >> 4);
    f32x4 acc[2][8];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int nf = 0; nf < 8; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.stats && tid < PW_BNT) { st_s[tid] = 0.f; st_ss[tid] = 0.f; }

    for (int k0 = 0; k0 < K; k0 += PW_KC) {
      __syncthreads();  // previous step's fragment reads done (also orders psc/psh, st_* init)
      const int kk = k0 + 8 * schunk;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int r = srow + 64 * p;
        const long m = m0 + r;
        float v[8];
        if (m < a.M && kk < K) {
          load8(x + m * K + kk, v);
          if (PRO != MDS_PRO_NONE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              float z = v[j] * psc[kk + j] + psh[kk + j];
              v[j] = (PRO == MDS_PRO_AFFINE) ? z : siluf_(z);
            }
            if (PRO == MDS_PRO_BN_SILU_GATE) {
              float g[8];
              load8f(a.pro.gate + (m / a.pro.rows_per_group) * K + kk, g);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] *= g[j];
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
        store8(xs + r * LD + 8 * schunk, v);
        // weights: rows n0 + r
        const int n = n0 + r;
        if (n < N && kk < K) {
          load8(w + (long)n * K + kk, v);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
        store8(ws + r * LD + 8 * schunk, v);
      }
      __syncthreads();
      frag_t xf[2];
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) xf[mf] = ld_frag(xs + (32 * wave + 16 * mf + i) * LD + 8 * q);
#pragma unroll
      for (int nf = 0; nf < 8; ++nf) {
        if (nf < nfr) {
          frag_t wf = ld_frag(ws + (16 * nf + i) * LD + 8 * q);
          mma16(wf, xf[0], acc[0][nf]);  // acc[r] = y[m = i][n = 4q + r]
          mma16(wf, xf[1], acc[1][nf]);
        }
      }
    }

    // ---- epilogue: residual, store, statistics
    float part_s[32], part_ss[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) { part_s[e] = 0.f; part_ss[e] = 0.f; }
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      const long m = m0 + 32 * wave + 16 * mf + i;
#pragma unroll
      for (int nf = 0; nf < 8; ++nf) {
        if (nf < nfr) {
          const int n = n0 + 16 * nf + 4 * q;
          float v[4] = {acc[mf][nf][0], acc[mf][nf][1], acc[mf][nf][2], acc[mf][nf][3]};
          if (m < a.M) {
            if (a.residual) {
              float rr[4];
              load4((const T*)a.residual + m * N + n, rr);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += rr[r];
            }
            store4(y + m * N + n, v);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            part_s[nf * 4 + r] += v[r];
            part_ss[nf * 4 + r] += v[r] * v[r];
          }
        }
      }
    }
    if (a.stats) {
      int e0 = reduce_scatter32(part_s, i);
      reduce_scatter32(part_ss, i);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int e = e0 + t;
        const int nl = 16 * (e >> 2) + 4 * q + (e & 3);
        atomicAdd(&st_s[nl], part_s[t]);
        atomicAdd(&st_ss[nl], part_ss[t]);
      }
      __syncthreads();
      if (tid < PW_BNT && n0 + tid < N) {
        float* st = a.stats + (long)(blockIdx.x % MDS_STAT_SLOTS) * 2 * N;
        atomicAdd(st + n0 + tid, st_s[tid]);
        atomicAdd(st + N + n0 + tid, st_ss[tid]);
      }
    }
  }
}

template <typename T>
static size_t pw_fwd_smem(int K) {
  int Kpad = (K + 31) & ~31;
  return (size_t)(PW_BM + PW_BNT) * PwLd<T>::v * sizeof(T) + (size_t)(2 * Kpad + 2 * PW_BNT) * sizeof(float);
}

extern "C" int mds_pw_fwd(const mds_pw_fwd_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->M > 0 && a->K > 0 && a->N > 0, "pw_fwd: bad dims");
  MDS_REQUIRE(a->K % 8 == 0 && a->N % 16 == 0, "pw_fwd: K=%d must be a multiple of 8, N=%d of 16", a->K, a->N);
  MDS_REQUIRE(a->x && a->w && a->y, "pw_fwd: null pointer");
  MDS_REQUIRE(a->pro.mode == MDS_PRO_NONE || (a->pro.scale && a->pro.shift), "pw_fwd: prologue needs scale/shift");
  MDS_REQUIRE(a->pro.mode != MDS_PRO_BN_SILU_GATE || (a->pro.gate && a->pro.rows_per_group > 0), "pw_fwd: gate prologue");
  const int mt = cdiv(a->M, PW_BM), nt = cdiv(a->N, PW_BNT);
  dim3 grid(mt, (mt < 2048 && nt > 1) ? nt : 1), block(256);
#define PW_GO(T, PRO) MDS_LAUNCH((pw_fwd_kernel<T, PRO>), grid, block, pw_fwd_smem<T>(a->K), stream, *a)
  MDS_DISPATCH_DTYPE(a->dtype, T, {
    switch (a->pro.mode) {
      case MDS_PRO_NONE: PW_GO(T, MDS_PRO_NONE); break;
      case MDS_PRO_AFFINE: PW_GO(T, MDS_PRO_AFFINE); break;
      case MDS_PRO_BN_SILU: PW_GO(T, MDS_PRO_BN_SILU); break;
      case MDS_PRO_BN_SILU_GATE: PW_GO(T, MDS_PRO_BN_SILU_GATE); break;
      default: mds_set_error("pw_fwd: prologue mode %d", a->pro.mode); return MDS_ERR_BAD_ARG;
    }
  });
#undef PW_GO
  return mds_check_launch("pw_fwd");
}

// ------------------------------------------------------------------------------------ wgrad
#define WG_ROWS 64  // rows staged per step (2 MFMA k-steps of 32 rows)
#define WG_NT 128   // output-channel tile
#define WG_KT 64    // input-channel tile
#define WG_LDX (WG_KT + 2)   // pitch = 2 (mod 8): transposed fragment reads are bank-conflict free
#define WG_LDY (WG_NT + 2)

template <typename T, int PRO>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(mds_pw_wgrad_args a, int rows_per_block) {
  typedef typename Frag<T>::type frag_t;
  MDS_DYN_SMEM(smem);
  T* xs = (T*)smem;               // [WG_ROWS][WG_LDX]
  T* ds = xs + WG_ROWS * WG_LDX;  // [WG_ROWS][WG_LDY]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int K = a.K, N = a.N;
  const int ntiles_k = (K + WG_KT - 1) / WG_KT;
  const int n0 = (blockIdx.y / ntiles_k) * WG_NT, kt0 = (blockIdx.y % ntiles_k) * WG_KT;
  const long mbeg = (long)blockIdx.x * rows_per_block;
  long mend = mbeg + rows_per_block;
  if (mend > a.M) mend = a.M;
  const T* x = (const T*)a.x;
  const T* dy = (const T*)a.dy;

  // x staging: 8 chunks/row, 32 rows/pass, 2 passes ; dy staging: 16 chunks/row, 16 rows/pass, 4 passes
  const int xc = tid & 7, xr = tid >> 3;
  const int yc = tid & 15, yr = tid >> 4;
  float sc[8], sh[8];
  const int kx = kt0 + 8 * xc;
  if (PRO != MDS_PRO_NONE && kx < K) { load8f(a.pro.scale + kx, sc); load8f(a.pro.shift + kx, sh); }

  f32x4 acc[2][4];  // wave owns n-fragments {2*wave, 2*wave+1} x 4 k-fragments
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (long mb = mbeg; mb < mend; mb += WG_ROWS) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = xr + 32 * p;
      const long m = mb + r;
      float v[8];
      if (m < mend && kx < K) {
        load8(x + m * K + kx, v);
        if (PRO != MDS_PRO_NONE) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float z = v[j] * sc[j] + sh[j];
            v[j] = (PRO == MDS_PRO_AFFINE) ? z : siluf_(z);
          }
          if (PRO == MDS_PRO_BN_SILU_GATE) {
            float g[8];
            load8f(a.pro.gate + (m / a.pro.rows_per_group) * K + kx, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= g[j];
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
      }
      lds_store8_u32(xs + r * WG_LDX + 8 * xc, v);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = yr + 16 * p;
      const long m = mb + r;
      const int n = n0 + 8 * yc;
      float v[8];
      if (m < mend && n < N) {
        load8(dy + m * N + n, v);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
      }
      lds_store8_u32(ds + r * WG_LDY + 8 * yc, v);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < WG_ROWS / 32; ++ks) {
      const int rb = 32 * ks + 8 * q;  // this lane's 8 rows (the MFMA k index)
      frag_t xf[4], yf[2];
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[v][j] = xs[(rb + j) * WG_LDX + 16 * v + i];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) yf[u][j] = ds[(rb + j) * WG_LDY + 16 * (2 * wave + u) + i];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) mma16(yf[u], xf[v], acc[u][v]);  // acc[r] = dw[n = 4q + r][k = i]
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int k = kt0 + 16 * v + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 16 * (2 * wave + u) + 4 * q + r;
        if (n < N && k < K) atomicAdd(a.dw + (long)n * K + k, acc[u][v][r]);
      }
    }
}

extern "C" int mds_pw_wgrad(const mds_pw_wgrad_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->M > 0 && a->K > 0 && a->N > 0, "pw_wgrad: bad dims");
  MDS_REQUIRE(a->K % 8 == 0 && a->N % 8 == 0, "pw_wgrad: K, N must be multiples of 8");
  MDS_REQUIRE(a->x && a->dy && a->dw, "pw_wgrad: null pointer");
  MDS_REQUIRE(a->pro.mode == MDS_PRO_NONE || (a->pro.scale && a->pro.shift), "pw_wgrad: prologue needs scale/shift");
  MDS_REQUIRE(a->pro.mode != MDS_PRO_BN_SILU_GATE || (a->pro.gate && a->pro.rows_per_group > 0), "pw_wgrad: gate prologue");
  const int tiles = cdiv(a->N, WG_NT) * cdiv(a->K, WG_KT);
  long want_blocks = 1024 / tiles;
  if (want_blocks < 1) want_blocks = 1;
  long rpb = (a->M + want_blocks - 1) / want_blocks;
  rpb = ((rpb + WG_ROWS - 1) / WG_ROWS) * WG_ROWS;
  if (rpb < 4 * WG_ROWS) rpb = 4 * WG_ROWS;
  dim3 grid(cdiv(a->M, rpb), tiles), block(256);
#define WG_GO(T, PRO) MDS_LAUNCH((pw_wgrad_kernel<T, PRO>), grid, block, (size_t)WG_ROWS * (WG_LDX + WG_LDY) * sizeof(T), stream, *a, (int)rpb)
  MDS_DISPATCH_DTYPE(a->dtype, T, {
    switch (a->pro.mode) {
      case MDS_PRO_NONE: WG_GO(T, MDS_PRO_NONE); break;
      case MDS_PRO_AFFINE: WG_GO(T, MDS_PRO_AFFINE); break;
      case MDS_PRO_BN_SILU: WG_GO(T, MDS_PRO_BN_SILU); break;
      case MDS_PRO_BN_SILU_GATE: WG_GO(T, MDS_PRO_BN_SILU_GATE); break;
      default: mds_set_error("pw_wgrad: prologue mode %d", a->pro.mode); return MDS_ERR_BAD_ARG;
    }
  });
#undef WG_GO
  return mds_check_launch("pw_wgrad");
}
