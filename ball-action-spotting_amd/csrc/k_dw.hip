// k_dw.hip — depthwise 3x3 (2D, stride 1 / TF-SAME stride 2) and 3x3x3 (3D) convolutions.
//
// HBM-bound VALU kernels (9 or 27 MACs per element).  A thread owns V channels (one 16-byte
// vector in the forward, one 8/16-byte vector in the backward) of a 4-pixel strip; filter taps
// sit transposed in LDS ([tap][C]) and are read as broadcast vectors; the producer's BN+SiLU is
// applied while loading (zero padding after it); input vectors are consumed one at a time so the
// live register set stays small enough for 3-4 waves per SIMD (latency hiding is occupancy here);
// BatchNorm sums of the output are accumulated in registers, then one LDS reduction + one atomic
// per channel per block.
#include "elem.h"

#define DW_WS 4  // output (fwd) / input (bwd) pixels per thread along W

// ---- V-wide channel vectors
template <int V> struct Vec;
template <> struct Vec<8> {
  template <typename T> static MDS_DEV void ld(const T* p, float (&v)[8]) { load8(p, v); }
  template <typename T> static MDS_DEV void st(T* p, const float (&v)[8]) { store8(p, v); }
};
template <> struct Vec<4> {
  template <typename T> static MDS_DEV void ld(const T* p, float (&v)[4]) { load4(p, v); }
  template <typename T> static MDS_DEV void st(T* p, const float (&v)[4]) { store4(p, v); }
};

template <int V>
struct VMap {  // like RowMap, for a channel span [cbeg, cbeg + span) walked V channels per thread
  int cpr, rpb, chunk, rsub, c0;
  bool valid;
};
template <int V>
MDS_DEV VMap<V> vmap(int cbeg, int span) {
  VMap<V> m;
  m.cpr = span / V;
  m.rpb = 256 / m.cpr;
  m.chunk = threadIdx.x % m.cpr;
  m.rsub = threadIdx.x / m.cpr;
  m.valid = m.rsub < m.rpb;
  m.c0 = cbeg + m.chunk * V;
  return m;
}
template <int NV, int V>
MDS_DEV void block_reduce_v(float (&acc)[NV][V], const VMap<V>& m, float* red) {
  __syncthreads();
  if (m.valid) {
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int j = 0; j < V; ++j) red[((m.rsub * NV + v) * m.cpr + m.chunk) * V + j] = acc[v][j];
  }
  __syncthreads();
  if (m.valid && m.rsub == 0) {
    for (int r = 1; r < m.rpb; ++r)
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[v][j] += red[((r * NV + v) * m.cpr + m.chunk) * V + j];
  }
}
template <int V>
MDS_DEV void ldv(const float* p, float (&v)[V]) {
#pragma unroll
  for (int j = 0; j < V; j += 4) {
    f32x4 a = *(const f32x4*)(p + j);
    v[j] = a[0]; v[j + 1] = a[1]; v[j + 2] = a[2]; v[j + 3] = a[3];
  }
}

// ------------------------------------------------------------------------------------ forward
template <typename T, int S>
__global__ __launch_bounds__(256, 3) void dw_fwd_kernel(mds_dw_fwd_args a) {
  const int V = 8;
  MDS_DYN_SMEM(smem);
  float* wl = (float*)smem;          // [ntap][C]
  float* red = wl + a.kt * 9 * a.C;  // [256*8*2]
  const int C = a.C, ntap = a.kt * 9;
  for (int e = threadIdx.x; e < ntap * C; e += 256) {
    int t = e / C, c = e - t * C;
    wl[e] = a.w[(long)c * ntap + t];
  }
  const VMap<V> m = vmap<V>(0, C);
  const int c0 = m.c0;
  const int mode = a.pro.mode;
  float sc[V], sh[V];
  if (m.valid && mode != MDS_PRO_NONE) { ldv<V>(a.pro.scale + c0, sc); ldv<V>(a.pro.shift + c0, sh); }
  __syncthreads();
  float st[2][V];
#pragma unroll
  for (int j = 0; j < V; ++j) { st[0][j] = 0.f; st[1][j] = 0.f; }
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
      float acc[DW_WS][V];
#pragma unroll
      for (int o = 0; o < DW_WS; ++o)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[o][j] = 0.f;
      for (int dt = 0; dt < a.kt; ++dt) {
        const int it = ot + dt - tpad;
        if (it < 0 || it >= a.T) continue;
        for (int ky = 0; ky < 3; ++ky) {
          const int iy = oy * S + ky - a.pad_t;
          if (iy < 0 || iy >= a.IH) continue;
          const T* xrow = x + (((long)(n * a.T + it) * a.IH + iy) * a.IW) * C + c0;
          float w3[3][V];
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) ldv<V>(wl + ((dt * 3 + ky) * 3 + kx) * C + c0, w3[kx]);
#pragma unroll
          for (int s = 0; s < NSEG; ++s) {
            const int ix = ox0 * S - a.pad_l + s;
            if (ix < 0 || ix >= a.IW) continue;  // zero padding (after the activation)
            float v[V];
            Vec<V>::ld(xrow + (long)ix * C, v);
            if (mode != MDS_PRO_NONE) {
#pragma unroll
              for (int j = 0; j < V; ++j) {
                float z = v[j] * sc[j] + sh[j];
                v[j] = (mode == MDS_PRO_AFFINE) ? z : siluf_(z);
              }
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              if ((s - kx) >= 0 && ((s - kx) % S) == 0 && (s - kx) / S < DW_WS) {
                const int o = (s - kx) / S;
#pragma unroll
                for (int j = 0; j < V; ++j) acc[o][j] += v[j] * w3[kx][j];
              }
            }
          }
        }
      }
      T* yrow = y + (((long)(n * a.T + ot) * a.OH + oy) * a.OW) * C + c0;
#pragma unroll
      for (int o = 0; o < DW_WS; ++o) {
        if (ox0 + o < a.OW) {
          Vec<V>::st(yrow + (long)(ox0 + o) * C, acc[o]);
#pragma unroll
          for (int j = 0; j < V; ++j) { st[0][j] += acc[o][j]; st[1][j] += acc[o][j] * acc[o][j]; }
        }
      }
    }
  }
  if (a.stats) {
    block_reduce_v<2, V>(st, m, red);
    if (m.valid && m.rsub == 0) {
      float* sp = a.stats + (long)(blockIdx.x % MDS_STAT_SLOTS) * 2 * C;
#pragma unroll
      for (int j = 0; j < V; ++j) { atomicAdd(sp + c0 + j, st[0][j]); atomicAdd(sp + C + c0 + j, st[1][j]); }
    }
  }
}

static int dw_blocks(long nstrips, int rows_pp, int cap) {
  long b = (nstrips + rows_pp - 1) / rows_pp;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
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
  dim3 grid(dw_blocks(nstrips, rows_per_pass(a->C), 2048)), block(256);
  MDS_DISPATCH_DTYPE(a->dtype, T, {
    if (a->stride == 1) MDS_LAUNCH((dw_fwd_kernel<T, 1>), grid, block, smem, stream, *a);
    else MDS_LAUNCH((dw_fwd_kernel<T, 2>), grid, block, smem, stream, *a);
  });
  return mds_check_launch("dw_fwd");
}

// ------------------------------------------------------------------------------------ backward
// Thread owns 4 channels of a 4-pixel strip of the INPUT.  Outputs g = (dgrad) * silu'(z) (the
// gradient wrt the BN output of the producing 1x1 conv), the BN-backward sums of g, and the
// filter gradient (9 x 4 register accumulators).  blockIdx.y splits the channels when C/4 > 256.
// For the 3x3x3 case the strip loop runs once per temporal tap for the filter gradient
// (the 3D tensors are small: 4x5x23x40x576).
template <typename T, int S, int PL>
__global__ __launch_bounds__(256, 3) void dw_bwd_kernel(mds_dw_bwd_args a, int span) {
  const int V = 4;
  MDS_DYN_SMEM(smem);
  const int C = a.C, ntap = a.kt * 9;
  const int cbeg = blockIdx.y * span;
  float* wl = (float*)smem;        // [ntap][span]
  float* red = wl + ntap * span;   // [256*4*3]
  float* dwt = red + 256 * 4 * 3;  // [span][9] filter-gradient staging
  for (int e = threadIdx.x; e < ntap * span; e += 256) {
    int t = e / span, c = e - t * span;
    wl[e] = a.w[(long)(cbeg + c) * ntap + t];
  }
  const VMap<V> m = vmap<V>(cbeg, span);
  const int c0 = m.c0, cl = m.chunk * V;
  float sc[V], sh[V], mu[V], rs[V];
  if (m.valid) {
    ldv<V>(a.pro.scale + c0, sc); ldv<V>(a.pro.shift + c0, sh);
    ldv<V>(a.mean + c0, mu); ldv<V>(a.rstd + c0, rs);
  }
  __syncthreads();
  const int strips_w = (a.IW + DW_WS - 1) / DW_WS;
  const long nstrips = (long)a.N * a.T * a.IH * strips_w;
  const T* x = (const T*)a.x;
  const T* dy = (const T*)a.dy;
  T* g = (T*)a.g;
  const int tpad = a.kt == 3 ? 1 : 0;
  const int NSEG = (S == 1) ? DW_WS + 2 : DW_WS / 2 + 2;  // dy columns touching the strip
  float st[2][V];
#pragma unroll
  for (int j = 0; j < V; ++j) { st[0][j] = 0.f; st[1][j] = 0.f; }

  for (int dtw = 0; dtw < a.kt; ++dtw) {
    float dwacc[9][V];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < V; ++j) dwacc[t][j] = 0.f;
    if (m.valid) {
      for (long sidx = (long)blockIdx.x * m.rpb + m.rsub; sidx < nstrips; sidx += (long)gridDim.x * m.rpb) {
        const int sw = (int)(sidx % strips_w);
        long r = sidx / strips_w;
        const int iy = (int)(r % a.IH); r /= a.IH;
        const int it = (int)(r % a.T);
        const int n = (int)(r / a.T);
        const int ix0 = sw * DW_WS;
        float xv[DW_WS][V], act[DW_WS][V], da[DW_WS][V];
        const T* xrow = x + (((long)(n * a.T + it) * a.IH + iy) * a.IW) * C + c0;
#pragma unroll
        for (int o = 0; o < DW_WS; ++o) {
          if (ix0 + o < a.IW) {
            Vec<V>::ld(xrow + (long)(ix0 + o) * C, xv[o]);
#pragma unroll
            for (int j = 0; j < V; ++j) act[o][j] = siluf_(xv[o][j] * sc[j] + sh[j]);
          } else {
#pragma unroll
            for (int j = 0; j < V; ++j) { xv[o][j] = 0.f; act[o][j] = 0.f; }
          }
#pragma unroll
          for (int j = 0; j < V; ++j) da[o][j] = 0.f;
        }
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
            float w3[3][V];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) ldv<V>(wl + ((dt * 3 + ky) * 3 + kx) * span + cl, w3[kx]);
#pragma unroll
            for (int s = 0; s < NSEG; ++s) {
              const int ox = seg_lo + s;
              if (ox < 0 || ox >= a.OW) continue;
              float v[V];
              Vec<V>::ld(drow + (long)ox * C, v);
#pragma unroll
              for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int o = 0; o < DW_WS; ++o) {
                  // ox = (ix0 + o + PL - kx) / S must be an integer equal to seg_lo + s (ix0 % 4 == 0)
                  const bool hit = (S == 1) ? ((o - kx + 2) == s)
                                            : ((((o + PL - kx) % 2) == 0) && (((o + PL - kx + 2) / 2) == s));
                  if (hit) {
                    if (do_da) {
#pragma unroll
                      for (int j = 0; j < V; ++j) da[o][j] += v[j] * w3[kx][j];
                    }
                    if (do_dw) {
#pragma unroll
                      for (int j = 0; j < V; ++j) dwacc[ky * 3 + kx][j] += v[j] * act[o][j];
                    }
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
              float gv[V];
#pragma unroll
              for (int j = 0; j < V; ++j) {
                gv[j] = da[o][j] * silu_gradf_(xv[o][j] * sc[j] + sh[j]);
                st[0][j] += gv[j];
                st[1][j] += gv[j] * ((xv[o][j] - mu[j]) * rs[j]);
              }
              Vec<V>::st(grow + (long)(ix0 + o) * C, gv);
            }
          }
        }
      }
    }
    // flush this temporal tap's 9 filter-gradient rows: block-reduce, transpose through LDS into
    // the parameter's [C][kt*9] order, then coalesced atomics (uncoalesced ones cost one L2
    // transaction per lane and dominated this kernel)
#pragma unroll
    for (int rnd = 0; rnd < 3; ++rnd) {
      float part[3][V];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < V; ++j) part[t][j] = dwacc[rnd * 3 + t][j];
      block_reduce_v<3, V>(part, m, red);
      if (m.valid && m.rsub == 0) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int j = 0; j < V; ++j) dwt[(cl + j) * 9 + rnd * 3 + t] = part[t][j];
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < span * 9; e += 256) {
      const int c = e / 9, t = e - c * 9;
      atomicAdd(a.dw + (long)(cbeg + c) * ntap + dtw * 9 + t, dwt[e]);
    }
    __syncthreads();
  }
  block_reduce_v<2, V>(st, m, red);
  if (m.valid && m.rsub == 0) {
    float* sp = a.stats + (long)(blockIdx.x % MDS_STAT_SLOTS) * 2 * C;
#pragma unroll
    for (int j = 0; j < V; ++j) { atomicAdd(sp + c0 + j, st[0][j]); atomicAdd(sp + C + c0 + j, st[1][j]); }
  }
}

extern "C" int mds_dw_bwd(const mds_dw_bwd_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->N > 0 && a->T > 0 && a->C % 8 == 0 && a->C <= 2048, "dw_bwd: bad dims");
  MDS_REQUIRE(a->kt == 1 || a->kt == 3, "dw_bwd: kt must be 1 or 3");
  MDS_REQUIRE(a->stride == 1 || a->stride == 2, "dw_bwd: stride");
  MDS_REQUIRE(a->x && a->dy && a->w && a->g && a->dw && a->stats && a->mean && a->rstd, "dw_bwd: null pointer");
  MDS_REQUIRE(a->pro.mode == MDS_PRO_BN_SILU && a->pro.scale && a->pro.shift, "dw_bwd: needs the BN+SiLU prologue of the forward");
  MDS_REQUIRE(a->stride == 2 ? (a->pad_l == 0 || a->pad_l == 1) : (a->pad_l == 1), "dw_bwd: pad_l=%d unsupported for stride %d", a->pad_l, a->stride);
  int nsplit = 1;
  while (a->C / (4 * nsplit) > 256 || a->C % (4 * nsplit) != 0) {
    ++nsplit;
    MDS_REQUIRE(nsplit <= 8, "dw_bwd: cannot split C=%d", a->C);
  }
  const int span = a->C / nsplit;
  const long nstrips = (long)a->N * a->T * a->IH * ((a->IW + DW_WS - 1) / DW_WS);
  const size_t smem = ((size_t)a->kt * 9 * span + 256 * 4 * 3 + 9 * span) * sizeof(float);
  dim3 grid(dw_blocks(nstrips, 256 / (span / 4), 512 / nsplit), nsplit), block(256);
  MDS_DISPATCH_DTYPE(a->dtype, T, {
    if (a->stride == 1) MDS_LAUNCH((dw_bwd_kernel<T, 1, 1>), grid, block, smem, stream, *a, span);
    else if (a->pad_l == 0) MDS_LAUNCH((dw_bwd_kernel<T, 2, 0>), grid, block, smem, stream, *a, span);
    else MDS_LAUNCH((dw_bwd_kernel<T, 2, 1>), grid, block, smem, stream, *a, span);
  });
  return mds_check_launch("dw_bwd");
}
