// k_dw.hip — depthwise 3x3 (2D, stride 1 / TF-SAME stride 2) and 3x3x3 (3D) convolutions.
//
// HBM/LDS-bound VALU kernels (9 or 27 MACs per element): a thread owns 8 channels (one 16-byte
// vector) of a 4-pixel strip, filter taps sit transposed in LDS ([tap][C]) and are read as
// broadcast b128s, the producer's BN+SiLU is applied while loading (zero padding after it), and
// the per-channel BatchNorm sums of the output are accumulated in registers, then one LDS
// reduction + one atomic per channel per block.
#include "elem.h"

#define DW_WS 4  // output (fwd) / input (bwd) pixels per thread along W

template <typename T>
MDS_DEV void load_act8(const T* p, int mode, const float (&sc)[8], const float (&sh)[8], float (&v)[8]) {
  load8(p, v);
  apply_pro8(mode, v, sc, sh);
}

template <typename T, int S>
__global__ __launch_bounds__(256) void dw_fwd_kernel(mds_dw_fwd_args a) {
  MDS_DYN_SMEM(smem);
  float* wl = (float*)smem;                  // [ntap][C]
  float* red = wl + a.kt * 9 * a.C;          // [256*8*2]
  const int C = a.C, ntap = a.kt * 9;
  for (int e = threadIdx.x; e < ntap * C; e += 256) {
    int t = e / C, c = e - t * C;
    wl[e] = a.w[(long)c * ntap + t];
  }
  const RowMap m = rowmap(C);
  const int c0 = m.chunk * 8;
  float sc[8], sh[8];
  if (m.valid && a.pro.mode != MDS_PRO_NONE) { load8f(a.pro.scale + c0, sc); load8f(a.pro.shift + c0, sh); }
  __syncthreads();
  float st[2][8] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
  const int strips_w = (a.OW + DW_WS - 1) / DW_WS;
  const long nstrips = (long)a.N * a.T * a.OH * strips_w;
  const T* x = (const T*)a.x;
  T* y = (T*)a.y;
  const int tpad = a.kt == 3 ? 1 : 0;
  const int NSEG = (DW_WS - 1) * S + 3;
  if (m.valid) {
    for (long sidx = (long)blockIdx.x * m.rpb + m.rsub; sidx < nstrips; sidx += (long)gridDim.x * m.rpb) {
      const int sw = (int)(sidx % strips_w);
      long r = sidx / strips_w;
      const int oy = (int)(r % a.OH); r /= a.OH;
      const int ot = (int)(r % a.T);
      const int n = (int)(r / a.T);
      const int ox0 = sw * DW_WS;
      float acc[DW_WS][8];
#pragma unroll
      for (int o = 0; o < DW_WS; ++o)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[o][j] = 0.f;
      for (int dt = 0; dt < a.kt; ++dt) {
        const int it = ot + dt - tpad;
        if (it < 0 || it >= a.T) continue;
        for (int ky = 0; ky < 3; ++ky) {
          const int iy = oy * S + ky - a.pad_t;
          if (iy < 0 || iy >= a.IH) continue;
          const T* xrow = x + (((long)(n * a.T + it) * a.IH + iy) * a.IW) * C + c0;
          float seg[NSEG][8];
#pragma unroll
          for (int s = 0; s < NSEG; ++s) {
            const int ix = ox0 * S - a.pad_l + s;
            if (ix >= 0 && ix < a.IW) {
              load_act8<T>(xrow + (long)ix * C, a.pro.mode, sc, sh, seg[s]);
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) seg[s][j] = 0.f;
            }
          }
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            float w8[8];
            load8f(wl + ((dt * 3 + ky) * 3 + kx) * C + c0, w8);
#pragma unroll
            for (int o = 0; o < DW_WS; ++o)
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[o][j] += seg[o * S + kx][j] * w8[j];
          }
        }
      }
      T* yrow = y + (((long)(n * a.T + ot) * a.OH + oy) * a.OW) * C + c0;
#pragma unroll
      for (int o = 0; o < DW_WS; ++o) {
        if (ox0 + o < a.OW) {
          store8(yrow + (long)(ox0 + o) * C, acc[o]);
#pragma unroll
          for (int j = 0; j < 8; ++j) { st[0][j] += acc[o][j]; st[1][j] += acc[o][j] * acc[o][j]; }
        }
      }
    }
  }
  if (a.stats) {
    block_reduce_rows<2>(st, m, red);
    if (m.valid && m.rsub == 0) {
      float* sp = a.stats + (long)(blockIdx.x % MDS_STAT_SLOTS) * 2 * C;
#pragma unroll
      for (int j = 0; j < 8; ++j) { atomicAdd(sp + c0 + j, st[0][j]); atomicAdd(sp + C + c0 + j, st[1][j]); }
    }
  }
}

static int dw_blocks(long nstrips, int C) {
  long b = (nstrips + rows_per_pass(C) - 1) / rows_per_pass(C);
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

extern "C" int mds_dw_fwd(const mds_dw_fwd_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->N > 0 && a->T > 0 && a->C % 8 == 0 && a->C <= 2048, "dw_fwd: bad dims");
  MDS_REQUIRE(a->kt == 1 || a->kt == 3, "dw_fwd: kt must be 1 or 3");
  MDS_REQUIRE(a->stride == 1 || a->stride == 2, "dw_fwd: stride");
  MDS_REQUIRE(a->x && a->w && a->y, "dw_fwd: null pointer");
  MDS_REQUIRE(a->pro.mode == MDS_PRO_NONE || (a->pro.scale && a->pro.shift), "dw_fwd: prologue");
  MDS_REQUIRE(a->pro.mode != MDS_PRO_BN_SILU_GATE, "dw_fwd: gate prologue unsupported");
  const long nstrips = (long)a->N * a->T * a->OH * ((a->OW + DW_WS - 1) / DW_WS);
  const size_t smem = ((size_t)a->kt * 9 * a->C + 256 * 8 * 2) * sizeof(float);
  dim3 grid(dw_blocks(nstrips, a->C)), block(256);
  MDS_DISPATCH_DTYPE(a->dtype, T, {
    if (a->stride == 1) MDS_LAUNCH((dw_fwd_kernel<T, 1>), grid, block, smem, stream, *a);
    else MDS_LAUNCH((dw_fwd_kernel<T, 2>), grid, block, smem, stream, *a);
  });
  return mds_check_launch("dw_fwd");
}

// ------------------------------------------------------------------------------------ backward
// Thread owns 8 channels of a 4-pixel strip of the INPUT.  Outputs g = (dgrad) * silu'(z) (the
// gradient wrt the BN output of the producing 1x1 conv), the BN-backward sums of g, and the
// filter gradient.  For the 3x3x3 case the strip loop runs once per temporal tap so that only
// 9x8 filter-gradient accumulators are live (the tensors are small: 4x5x23x40x576).
template <typename T, int S, int PL>
__global__ __launch_bounds__(256) void dw_bwd_kernel(mds_dw_bwd_args a) {
  MDS_DYN_SMEM(smem);
  float* wl = (float*)smem;          // [ntap][C]
  float* red = wl + a.kt * 9 * a.C;  // [256*8*3]
  const int C = a.C, ntap = a.kt * 9;
  for (int e = threadIdx.x; e < ntap * C; e += 256) {
    int t = e / C, c = e - t * C;
    wl[e] = a.w[(long)c * ntap + t];
  }
  const RowMap m = rowmap(C);
  const int c0 = m.chunk * 8;
  float sc[8], sh[8], mu[8], rs[8];
  if (m.valid) {
    load8f(a.pro.scale + c0, sc); load8f(a.pro.shift + c0, sh);
    load8f(a.mean + c0, mu); load8f(a.rstd + c0, rs);
  }
  __syncthreads();
  const int strips_w = (a.IW + DW_WS - 1) / DW_WS;
  const long nstrips = (long)a.N * a.T * a.IH * strips_w;
  const T* x = (const T*)a.x;
  const T* dy = (const T*)a.dy;
  T* g = (T*)a.g;
  const int tpad = a.kt == 3 ? 1 : 0;
  // dy columns that can touch input columns ix0 .. ix0+3:  ox = (ix + PL - kx) / S
  const int NSEG = (S == 1) ? DW_WS + 2 : DW_WS / 2 + 2;
  float st[2][8] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};

  for (int dtw = 0; dtw < a.kt; ++dtw) {
    float dwacc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) dwacc[t][j] = 0.f;
    if (m.valid) {
      for (long sidx = (long)blockIdx.x * m.rpb + m.rsub; sidx < nstrips; sidx += (long)gridDim.x * m.rpb) {
        const int sw = (int)(sidx % strips_w);
        long r = sidx / strips_w;
        const int iy = (int)(r % a.IH); r /= a.IH;
        const int it = (int)(r % a.T);
        const int n = (int)(r / a.T);
        const int ix0 = sw * DW_WS;
        // activation of the forward input at this strip
        float act[DW_WS][8], z[DW_WS][8], xv[DW_WS][8];
        const T* xrow = x + (((long)(n * a.T + it) * a.IH + iy) * a.IW) * C + c0;
#pragma unroll
        for (int o = 0; o < DW_WS; ++o) {
          if (ix0 + o < a.IW) {
            load8(xrow + (long)(ix0 + o) * C, xv[o]);
#pragma unroll
            for (int j = 0; j < 8; ++j) { z[o][j] = xv[o][j] * sc[j] + sh[j]; act[o][j] = siluf_(z[o][j]); }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { xv[o][j] = 0.f; z[o][j] = 0.f; act[o][j] = 0.f; }
          }
        }
        float da[DW_WS][8];
#pragma unroll
        for (int o = 0; o < DW_WS; ++o)
#pragma unroll
          for (int j = 0; j < 8; ++j) da[o][j] = 0.f;
        for (int dt = 0; dt < a.kt; ++dt) {
          const bool do_da = (dtw == 0), do_dw = (dt == dtw);
          if (!do_da && !do_dw) continue;
          const int ot = it - dt + tpad;
          if (ot < 0 || ot >= a.T) continue;
          for (int ky = 0; ky < 3; ++ky) {
            const int num = iy + a.pad_t - ky;
            if (num < 0 || (num % S) != 0) continue;
            const int oy = num / S;
            if (oy >= a.OH) continue;
            const T* drow = dy + (((long)(n * a.T + ot) * a.OH + oy) * a.OW) * C + c0;
            const int seg_lo = (S == 1) ? (ix0 + PL - 2) : (ix0 / 2 - 1);
            float seg[NSEG][8];
#pragma unroll
            for (int s = 0; s < NSEG; ++s) {
              const int ox = seg_lo + s;
              if (ox >= 0 && ox < a.OW) {
                load8(drow + (long)ox * C, seg[s]);
              } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) seg[s][j] = 0.f;
              }
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              float w8[8];
              load8f(wl + ((dt * 3 + ky) * 3 + kx) * C + c0, w8);
#pragma unroll
              for (int o = 0; o < DW_WS; ++o) {
                // ox = (ix0 + o + PL - kx) / S must be an integer; ix0 is a multiple of 4
                if (((o + PL - kx) % S) != 0) continue;
                const int idx = (S == 1) ? (o - kx + 2) : ((o + PL - kx + 2) / 2);  // relative to seg_lo
                if (do_da) {
#pragma unroll
                  for (int j = 0; j < 8; ++j) da[o][j] += seg[idx][j] * w8[j];
                }
                if (do_dw) {
#pragma unroll
                  for (int j = 0; j < 8; ++j) dwacc[ky * 3 + kx][j] += seg[idx][j] * act[o][j];
                }
              }
            }
          }
        }
        if (dtw == 0) {
          T* grow = g + (((long)(n * a.T + it) * a.IH + iy) * a.IW) * C + c0;
#pragma unroll
          for (int o = 0; o < DW_WS; ++o) {
            if (ix0 + o < a.IW) {
              float gv[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                gv[j] = da[o][j] * silu_gradf_(z[o][j]);
                st[0][j] += gv[j];
                st[1][j] += gv[j] * ((xv[o][j] - mu[j]) * rs[j]);
              }
              store8(grow + (long)(ix0 + o) * C, gv);
            }
          }
        }
      }
    }
    // flush this temporal tap's 9 filter-gradient rows
#pragma unroll
    for (int rnd = 0; rnd < 3; ++rnd) {
      float part[3][8];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) part[t][j] = dwacc[rnd * 3 + t][j];
      block_reduce_rows<3>(part, m, red);
      if (m.valid && m.rsub == 0) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int j = 0; j < 8; ++j)
            atomicAdd(a.dw + (long)(c0 + j) * ntap + dtw * 9 + rnd * 3 + t, part[t][j]);
      }
    }
  }
  block_reduce_rows<2>(st, m, red);
  if (m.valid && m.rsub == 0) {
    float* sp = a.stats + (long)(blockIdx.x % MDS_STAT_SLOTS) * 2 * C;
#pragma unroll
    for (int j = 0; j < 8; ++j) { atomicAdd(sp + c0 + j, st[0][j]); atomicAdd(sp + C + c0 + j, st[1][j]); }
  }
}

extern "C" int mds_dw_bwd(const mds_dw_bwd_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->N > 0 && a->T > 0 && a->C % 8 == 0 && a->C <= 2048, "dw_bwd: bad dims");
  MDS_REQUIRE(a->kt == 1 || a->kt == 3, "dw_bwd: kt must be 1 or 3");
  MDS_REQUIRE(a->stride == 1 || a->stride == 2, "dw_bwd: stride");
  MDS_REQUIRE(a->x && a->dy && a->w && a->g && a->dw && a->stats && a->mean && a->rstd, "dw_bwd: null pointer");
  MDS_REQUIRE(a->pro.mode == MDS_PRO_BN_SILU && a->pro.scale && a->pro.shift, "dw_bwd: needs the BN+SiLU prologue of the forward");
  MDS_REQUIRE(a->stride == 2 ? (a->pad_l == 0 || a->pad_l == 1) : (a->pad_l == 1), "dw_bwd: pad_l=%d unsupported for stride %d", a->pad_l, a->stride);
  const long nstrips = (long)a->N * a->T * a->IH * ((a->IW + DW_WS - 1) / DW_WS);
  const size_t smem = ((size_t)a->kt * 9 * a->C + 256 * 8 * 3) * sizeof(float);
  int nb = dw_blocks(nstrips, a->C);
  if (nb > 512) nb = 512;
  dim3 grid(nb), block(256);
  MDS_DISPATCH_DTYPE(a->dtype, T, {
    if (a->stride == 1) MDS_LAUNCH((dw_bwd_kernel<T, 1, 1>), grid, block, smem, stream, *a);
    else if (a->pad_l == 0) MDS_LAUNCH((dw_bwd_kernel<T, 2, 0>), grid, block, smem, stream, *a);
    else MDS_LAUNCH((dw_bwd_kernel<T, 2, 1>), grid, block, smem, stream, *a);
  });
  return mds_check_launch("dw_bwd");
}
