// k_dw.hip — depthwise 3x3 (2D, stride 1 / TF-SAME stride 2) and 3x3x3 (3D) convolutions.
//
// HBM-bound VALU kernels (9 or 27 MACs per element).  Two generations live here:
//   * the REGISTER SLIDING-WINDOW kernels (dw2_*, dw2s_*, dw3_*: second half of the file) serve every
//     layer of the headline configuration: a lane owns a channel pair, a half-wave a 128-byte line, a
//     thread walks a strip of rows keeping the activated window in registers — no LDS tiles, no
//     barriers, 3-4 waves per SIMD (2.4-3.4x the tiled kernels);
//   * the LDS-TILED kernels (dw_fwd_kernel / dw_bwd_kernel, first half) remain for the 3x3x3 case
//     with T != 5 (frozen-encoder configuration, 11 slices): a block owns an 8x16 pixel patch of a
//     128-byte channel slab, issues all global loads of the patch (+halo) back to back, applies the
//     producer's BN+SiLU once per element on the way into LDS (zero padding after the activation) and
//     walks the T slices with a 3-slot ring.
// In both, BatchNorm sums and filter gradients are reduced in the block (shuffles / LDS) and written
// with coalesced atomics in the parameter's own order.
#include <stdlib.h>
#include "elem.h"

template <typename T> struct DwCfg { static const int CC = 128 / sizeof(T); };  // channels per slab

// ---- raw (storage-typed) V-channel vectors: loaded first, converted when consumed
template <typename T, int V> struct Raw;
template <int V> struct Raw<bf16_t, V> {
  typedef unsigned short vt __attribute__((ext_vector_type(V)));
  vt v;
  MDS_DEV void ld(const bf16_t* p) { v = *(const vt*)p; }
  MDS_DEV void get(float (&o)[V]) const {
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = bf2f(v[j]);
  }
};
template <int V> struct Raw<float, V> {
  f32x4 v[V / 4];
  MDS_DEV void ld(const float* p) {
#pragma unroll
    for (int k = 0; k < V / 4; ++k) v[k] = *(const f32x4*)(p + 4 * k);
  }
  MDS_DEV void get(float (&o)[V]) const {
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = v[j >> 2][j & 3];
  }
};
template <typename T, int V>
MDS_DEV void stv(T* p, const float (&v)[V]) {
  if (V == 8) store8(p, (const float(&)[8])v);
  else store4(p, (const float(&)[4])v);
}
template <int V>
MDS_DEV void ldv(const float* p, float (&v)[V]) {
#pragma unroll
  for (int j = 0; j < V; j += 4) {
    f32x4 a = *(const f32x4*)(p + j);
    v[j] = a[0]; v[j + 1] = a[1]; v[j + 2] = a[2]; v[j + 3] = a[3];
  }
}
MDS_DEV int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// sum over the lanes of a wave that share (lane % NCH)
template <int NCH>
MDS_DEV float sum_same_chunk(float v) {
#pragma unroll
  for (int m = NCH; m < 64; m <<= 1) v += __shfl_xor(v, m);
  return v;
}

// ------------------------------------------------------------------------------------ forward
template <typename T, int S, int KT>
__global__ __launch_bounds__(256) void dw_fwd_kernel(mds_dw_fwd_args a, int nchunks, int tpb) {
  constexpr int V = 8, CC = DwCfg<T>::CC, NCH = CC / V, NPT = 256 / NCH;
  constexpr int TOH = 8, TOW = 16, WS = TOH * TOW / NPT, NSTR = TOW / WS;
  constexpr int TH = (TOH - 1) * S + 3, TW = (TOW - 1) * S + 3, NPIX = TH * TW;
  constexpr int MAXL = (NPIX * NCH + 255) / 256;
  constexpr int NSEG = (WS - 1) * S + 3, NTAP = KT * 9;
  MDS_DYN_SMEM(smem);
  T* tile = (T*)smem;                             // [KT][NPIX][CC]
  float* wl = (float*)(tile + KT * NPIX * CC);    // [NTAP][CC]
  double* st_l = (double*)(wl + NTAP * CC);        // [2][CC] fp64: the waves' order of arrival does not matter
  const int tid = threadIdx.x, ch = tid % NCH, pt = tid / NCH;
  const int C = a.C;
  const int cz = blockIdx.z % nchunks, n = blockIdx.z / nchunks;
  const int cbeg = cz * CC, c0 = cbeg + ch * V;
  const bool cvalid = c0 < C;
  const int oy0 = blockIdx.y * TOH;
  const int iy0 = oy0 * S - a.pad_t;
  int ox0 = 0, ix0 = 0;  // set per tile: a block walks `tpb` consecutive tiles along W
  for (int e = tid; e < NTAP * CC; e += 256) {
    const int t = e / CC, c = e - t * CC;
    wl[e] = (cbeg + c < C) ? a.w[(long)(cbeg + c) * NTAP + t] : 0.f;
  }
  if (tid < 2 * CC) st_l[tid] = 0.0;
  const int mode = a.pro.mode;
  float sc[V], sh[V];
  if (cvalid && mode != MDS_PRO_NONE) { ldv<V>(a.pro.scale + c0, sc); ldv<V>(a.pro.shift + c0, sh); }
  const T* x = (const T*)a.x;
  T* y = (T*)a.y;
  const int r = pt / NSTR, sx = pt % NSTR;
  float st[2][V];
#pragma unroll
  for (int j = 0; j < V; ++j) { st[0][j] = 0.f; st[1][j] = 0.f; }

  auto load_slice = [&](int it, int slot) {
    const T* plane = x + ((long)(n * a.T + it) * a.IH * a.IW) * C + (cvalid ? c0 : 0);
    T* dst = tile + (long)slot * NPIX * CC + ch * V;
    Raw<T, V> raw[MAXL];
#pragma unroll
    for (int l = 0; l < MAXL; ++l) {
      const int pix = pt + NPT * l;
      const int ty = pix / TW, tx = pix - ty * TW;
      if (pix < NPIX)
        raw[l].ld(plane + ((long)clampi(iy0 + ty, 0, a.IH - 1) * a.IW + clampi(ix0 + tx, 0, a.IW - 1)) * C);
    }
#pragma unroll
    for (int l = 0; l < MAXL; ++l) {
      const int pix = pt + NPT * l;
      const int ty = pix / TW, tx = pix - ty * TW;
      if (pix < NPIX) {
        const int iy = iy0 + ty, ix = ix0 + tx;
        const bool ok = cvalid && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
        float v[V];
        raw[l].get(v);
        if (mode != MDS_PRO_NONE) {
#pragma unroll
          for (int j = 0; j < V; ++j) {
            float z = v[j] * sc[j] + sh[j];
            v[j] = (mode == MDS_PRO_AFFINE) ? z : siluf_(z);
          }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = ok ? v[j] : 0.f;  // zero padding AFTER the activation
        stv<T, V>(dst + (long)pix * CC, v);
      }
    }
  };

  auto compute = [&](int ot) {
    float acc[WS][V];
#pragma unroll
    for (int o = 0; o < WS; ++o)
#pragma unroll
      for (int j = 0; j < V; ++j) acc[o][j] = 0.f;
#pragma unroll
    for (int dt = 0; dt < KT; ++dt) {
      const int it = ot + dt - (KT == 3 ? 1 : 0);
      if (it < 0 || it >= a.T) continue;
      const T* src = tile + (long)(KT == 3 ? (it % 3) : 0) * NPIX * CC + ch * V;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        float w3[3][V];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) ldv<V>(wl + ((dt * 3 + ky) * 3 + kx) * CC + ch * V, w3[kx]);
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
          Raw<T, V> rv;
          rv.ld(src + (long)((r * S + ky) * TW + sx * WS * S + s) * CC);
          float v[V];
          rv.get(v);
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            if ((s - kx) >= 0 && ((s - kx) % S) == 0 && (s - kx) / S < WS) {
              const int o = (s - kx) / S;
#pragma unroll
              for (int j = 0; j < V; ++j) acc[o][j] += v[j] * w3[kx][j];
            }
          }
        }
      }
    }
    const int oy = oy0 + r;
    if (cvalid && oy < a.OH) {
      T* yrow = y + (((long)(n * a.T + ot) * a.OH + oy) * a.OW) * C + c0;
#pragma unroll
      for (int o = 0; o < WS; ++o) {
        const int ox = ox0 + sx * WS + o;
        if (ox < a.OW) {
          stv<T, V>(yrow + (long)ox * C, acc[o]);
#pragma unroll
          for (int j = 0; j < V; ++j) { st[0][j] += acc[o][j]; st[1][j] += acc[o][j] * acc[o][j]; }
        }
      }
    }
  };

  for (int tt = 0; tt < tpb; ++tt) {
    ox0 = (blockIdx.x * tpb + tt) * TOW;
    if (ox0 >= a.OW) break;
    ix0 = ox0 * S - a.pad_l;
    if (KT == 3) {
      load_slice(0, 0);
      for (int ot = 0; ot < a.T; ++ot) {
        if (ot + 1 < a.T) load_slice(ot + 1, (ot + 1) % 3);
        __syncthreads();
        compute(ot);
        __syncthreads();
      }
    } else {
      for (int ot = 0; ot < a.T; ++ot) {
        load_slice(ot, 0);
        __syncthreads();
        compute(ot);
        __syncthreads();
      }
    }
  }
  if (a.stats) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float s = sum_same_chunk<NCH>(st[k][j]);
        if ((tid & 63) < NCH) atomicAdd(&st_l[k * CC + ch * V + j], (double)s);
      }
    __syncthreads();
    if (tid < 2 * CC) {
      const int k = tid / CC, c = tid - k * CC;
      if (cbeg + c < C) {
        const int slot = (blockIdx.x + blockIdx.y * gridDim.x + n * 5) % MDS_STAT_SLOTS;
        atomicAdd(a.stats + ((long)slot * 2 + k) * C + cbeg + c, st_l[tid]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------ 2D stride 1
// The 3x3 stride-1 layers (all but two of the 2D depthwise convolutions) use a register
// sliding-window formulation instead of LDS tiles: a lane owns a PAIR of channels (one dword of
// bf16), a half-wave owns 64 consecutive channels (one 128-byte line per pixel), and a thread
// walks a strip of R rows x L columns left to right keeping the (R+2) x 3 activated window in
// registers.  Per column it issues R+2 loads (next column prefetched before the current one is
// consumed), applies the producer's BN+SiLU once per element ((R+2)/R re-activation at the band
// seams — same overhead as a tile halo) and does 9R packed FMAs.  ~100 VGPRs -> 4 waves/SIMD, no
// __syncthreads in the main loop.  A block is 8 strips x 32 channel pairs; BN sums and filter
// gradients are reduced over the 8 strips in LDS and flushed with coalesced atomics.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T> struct Pair;
template <> struct Pair<bf16_t> {
  typedef uint32_t raw_t;
  static MDS_DEV raw_t ld(const bf16_t* p) { return *(const uint32_t*)p; }
  static MDS_DEV f32x2 up(raw_t u) { return (f32x2){bits2f(u << 16), bits2f(u & 0xffff0000u)}; }
  static MDS_DEV void st(bf16_t* p, f32x2 v) { *(uint32_t*)p = pack2(v[0], v[1]); }
  static MDS_DEV raw_t pk(f32x2 v) { return pack2(v[0], v[1]); }
  static MDS_DEV void str(bf16_t* p, raw_t r) { *(uint32_t*)p = r; }
};
template <> struct Pair<float> {
  typedef f32x2 raw_t;
  static MDS_DEV raw_t ld(const float* p) { return *(const f32x2*)p; }
  static MDS_DEV f32x2 up(raw_t u) { return u; }
  static MDS_DEV void st(float* p, f32x2 v) { *(f32x2*)p = v; }
  static MDS_DEV raw_t pk(f32x2 v) { return v; }
  static MDS_DEV void str(float* p, raw_t r) { *(f32x2*)p = r; }
};
MDS_DEV f32x2 splat2(float v) { return (f32x2){v, v}; }
MDS_DEV f32x2 sigmoid2(f32x2 z) {
  f32x2 t = z * splat2(-1.4426950408889634f);
  f32x2 d = (f32x2){fast_exp2(t[0]), fast_exp2(t[1])} + splat2(1.0f);
  return (f32x2){fast_rcp(d[0]), fast_rcp(d[1])};
}

// eval-mode output transform of a channel pair (mds_epi_t): act(acc*scale + shift)
MDS_DEV f32x2 epi2(f32x2 v, int mode, f32x2 sc, f32x2 sh) {
  if (mode == MDS_EPI_NONE) return v;
  v = v * sc + sh;
  return mode == MDS_EPI_BN_SILU ? v * sigmoid2(v) : v;
}

struct DwStrips { int nchunks, nseg, L, nbands, spt, swap; long nstrips; };

// Squeeze-excite pooling of an inference plan inside the producing pass (mds_dw_fwd_args.pool): `sp` is this thread's sum of
// the STORED outputs of its strip (what mds_se_pool would read back), `img` the strip's batch element.  The 8 strips of a block
// are reduced in LDS; a block whose strips straddle two images flushes once per image.
MDS_DEV void dw_pool_flush(float (&red)[8][4][32], f32x2 sp, int img, bool strip_ok, double* pool, float inv, int C, int cbeg) {
  __shared__ int img_s[8];
  const int tid = threadIdx.x, cp = tid & 31, sl = tid >> 5;
  red[sl][0][cp] = sp[0]; red[sl][1][cp] = sp[1];
  if (cp == 0) img_s[sl] = strip_ok ? img : -1;
  __syncthreads();
  if (tid < 64 && cbeg + tid < C) {
    float t = 0.f;
    int cur = -1;
    for (int s = 0; s < 8; ++s) {
      const int n = img_s[s];
      if (n < 0) continue;
      if (n != cur) {
        if (cur >= 0) atomicAdd(pool + (long)cur * C + cbeg + tid, (double)(t * inv));
        t = 0.f; cur = n;
      }
      t += red[s][tid & 1][tid >> 1];
    }
    if (cur >= 0) atomicAdd(pool + (long)cur * C + cbeg + tid, (double)(t * inv));
  }
}
// Workgroup -> (strip block, channel chunk).  Consecutive workgroup ids go to consecutive XCDs (id % 8), each with its own L2.
// Neighbouring strips share halo rows / columns ((R + 2) / R x (L + 2) / L = 1.5x the compulsory reads when every strip fetches
// its own halo from HBM - the PMC FETCH_SIZE of round 2 showed 1.6x): XCD x takes a CONTIGUOUS range of the (chunk, strip)
// sequence, so the strips that share a halo run on the same XCD at about the same time and the halo comes from its L2.
// MDS_KNOB_DW_ORDER = 1 / 2 keep the former orders (channel chunk fastest / strip fastest, no remap) for A/B runs.
struct DwBlock { int bx, chunk; };
MDS_DEV DwBlock dw_block(const DwStrips& g) {
  DwBlock b;
  if (g.swap == 1) { b.bx = blockIdx.y; b.chunk = blockIdx.x; return b; }
  if (g.swap == 2) { b.bx = blockIdx.x; b.chunk = blockIdx.y; return b; }
  const unsigned nb = gridDim.x * gridDim.y, id = blockIdx.x + gridDim.x * blockIdx.y;
  const unsigned logical = xcd_contiguous(id, nb);
  b.chunk = (int)(logical / gridDim.x);
  b.bx = (int)(logical - (unsigned)b.chunk * gridDim.x);
  return b;
}

#ifndef MDS_DW2F_OCC
#define MDS_DW2F_OCC 3   /* blocks per CU the bf16 / two-row variants of dw2_fwd are compiled for (library A/B) */
#endif
// POOL: the squeeze-excite pooling of inference plans (mds_dw_fwd_args.pool) - a template flag, so that the training kernels do not carry its sums
template <typename T, int R, bool POOL = false>
// (fp32 six-row bands at 168 VGPRs spilled 56-63 registers: 1.3-1.5 TB/s in the fp32 inference plans; two blocks per CU for that variant)
__global__ __launch_bounds__(256, (sizeof(T) == 4 && R == 6) ? 2 : MDS_DW2F_OCC) void dw2_fwd_kernel(mds_dw_fwd_args a, DwStrips g) {
  MDS_CHAIN_PRIO();
  constexpr int NR = R + 2;
  typedef Pair<T> P;
  typedef typename P::raw_t raw_t;
  __shared__ float red[8][4][32];
  const int tid = threadIdx.x, cp = tid & 31, sl = tid >> 5;
  const DwBlock db = dw_block(g);   // (strip block, channel chunk) of this workgroup, XCD-aware
  const int bx = db.bx;
  const int C = a.C, cbeg = db.chunk * 64, c0 = cbeg + 2 * cp;
  const bool cvalid = c0 < C;
  const int mode = a.pro.mode;
  const int emode = a.epi.mode;
  f32x2 esc = splat2(1.f), esh = splat2(0.f);
  if (emode != MDS_EPI_NONE && c0 < a.C) { esc = *(const f32x2*)(a.epi.scale + c0); esh = *(const f32x2*)(a.epi.shift + c0); }
  f32x2 s1 = splat2(0.f), s2 = splat2(0.f), sp = splat2(0.f);
  int pimg = 0;
  bool pok = false;
  f32x2 w[3][3], sc = splat2(1.f), sh = splat2(0.f);
  if (cvalid) {
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t / 3][t % 3] = (f32x2){a.w[(long)c0 * 9 + t], a.w[(long)(c0 + 1) * 9 + t]};
    if (mode != MDS_PRO_NONE) { sc = *(const f32x2*)(a.pro.scale + c0); sh = *(const f32x2*)(a.pro.shift + c0); }
  }
  for (int k = 0; k < g.spt; ++k) {
    const long strip = ((long)bx * g.spt + k) * 8 + sl;
    if (strip < g.nstrips) { pimg = (int)(strip / ((long)g.nseg * g.nbands)); pok = true; }
    if (!cvalid || strip >= g.nstrips) continue;
    const int seg = (int)(strip % g.nseg);
    const long bt = strip / g.nseg;
    const int band = (int)(bt % g.nbands), img = (int)(bt / g.nbands);
    const int oy0 = band * R, ox0 = seg * g.L;
    const int nout = (a.OW - ox0 < g.L) ? a.OW - ox0 : g.L;
    const T* xim = (const T*)a.x + (long)img * a.IH * a.IW * C + c0;
    T* yim = (T*)a.y + ((long)img * a.OH + oy0) * a.OW * C + c0;
    int roff[NR], rok = 0;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int iy = oy0 - 1 + j;
      rok |= (iy >= 0 && iy < a.IH) ? (1 << j) : 0;
      roff[j] = clampi(iy, 0, a.IH - 1) * a.IW * C;
    }
    auto ldcol = [&](int ix, raw_t (&raw)[NR]) {
      const int xo = clampi(ix, 0, a.IW - 1) * C;
#pragma unroll
      for (int j = 0; j < NR; ++j) raw[j] = P::ld(xim + roff[j] + xo);
    };
    f32x2 win[NR][3];
    auto push = [&](int ix, const raw_t (&raw)[NR]) {  // slide the window one column to the right
      const bool cok = ix >= 0 && ix < a.IW;
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        f32x2 v = P::up(raw[j]);
        if (mode != MDS_PRO_NONE) {
          v = v * sc + sh;
          if (mode != MDS_PRO_AFFINE) v = v * sigmoid2(v);
        }
        const bool ok = cok && ((rok >> j) & 1);
        win[j][0] = win[j][1]; win[j][1] = win[j][2];
        win[j][2] = ok ? v : splat2(0.f);  // zero padding AFTER the activation
      }
    };
    // Three columns of raw loads are always in flight per thread (a ring of three register sets): at 4 bytes
    // per lane and load, one column ahead kept ~24 KB per CU outstanding - a quarter of what HBM latency x
    // bandwidth asks for.  Loads are never conditional (clamped addresses), so the in-order vmcnt counts stay exact.
    raw_t raw[3][NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) { win[j][1] = splat2(0.f); win[j][2] = splat2(0.f); }
    ldcol(ox0 - 1, raw[0]); ldcol(ox0, raw[1]); ldcol(ox0 + 1, raw[2]);
    push(ox0 - 1, raw[0]); ldcol(ox0 + 2, raw[0]);
    push(ox0, raw[1]); ldcol(ox0 + 3, raw[1]);
    auto step = [&](int o, raw_t (&rw)[NR]) {
      push(ox0 + o + 1, rw);
      ldcol(ox0 + o + 4, rw);   // (clamped: always a legal address)
      if (o < nout) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          f32x2 acc = splat2(0.f);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc += win[r + ky][kx] * w[ky][kx];
          if (oy0 + r < a.OH) {
            const raw_t pk = P::pk(epi2(acc, emode, esc, esh));
            P::str(yim + ((long)r * a.OW + ox0 + o) * C, pk);
            s1 += acc; s2 += acc * acc; if (POOL) sp += P::up(pk);
          }
        }
      }
    };
    for (int o = 0; o < nout; o += 3) {
      step(o, raw[2]);
      step(o + 1, raw[0]);
      step(o + 2, raw[1]);
    }
  }
  if (a.stats) {
    red[sl][0][cp] = s1[0]; red[sl][1][cp] = s1[1]; red[sl][2][cp] = s2[0]; red[sl][3][cp] = s2[1];
    __syncthreads();
    if (tid < 128) {
      const int kk = tid >> 6, c = tid & 63;
      float t = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) t += red[s][kk * 2 + (c & 1)][c >> 1];
      if (cbeg + c < C) atomicAdd(a.stats + ((long)(bx % MDS_STAT_SLOTS) * 2 + kk) * C + cbeg + c, (double)t);
    }
  }
  if (POOL && a.pool) dw_pool_flush(red, sp, pimg, pok, a.pool, a.pool_inv, C, cbeg);
}

// backward, same decomposition over INPUT pixels: window = dy rows iy-1..iy+R, cols ix-1..ix+1.
//   da[iy][ix]   = sum_{ky,kx} dy[iy+1-ky][ix+1-kx] * w[ky][kx]
//   dw[ky][kx]  += act[iy][ix] * dy[iy+1-ky][ix+1-kx]           (each input pixel owned by one thread)
template <typename T, int R>
__global__ __launch_bounds__(256, 2) void dw2_bwd_kernel(mds_dw_bwd_args a, DwStrips g) {
  MDS_CHAIN_PRIO();
  constexpr int NR = R + 2;
  typedef Pair<T> P;
  typedef typename P::raw_t raw_t;
  __shared__ float dwl[8][9][64];
  __shared__ float red[8][4][32];
  const int tid = threadIdx.x, cp = tid & 31, sl = tid >> 5;
  const DwBlock db = dw_block(g);   // (strip block, channel chunk) of this workgroup, XCD-aware
  const int bx = db.bx;
  const int C = a.C, cbeg = db.chunk * 64, c0 = cbeg + 2 * cp;
  const bool cvalid = c0 < C;
  f32x2 s1 = splat2(0.f), s2 = splat2(0.f);
  f32x2 w[3][3], dwacc[3][3], sc = splat2(0.f), sh = splat2(0.f), mu = splat2(0.f), rs = splat2(0.f);
#pragma unroll
  for (int t = 0; t < 9; ++t) dwacc[t / 3][t % 3] = splat2(0.f);
  if (cvalid) {
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t / 3][t % 3] = (f32x2){a.w[(long)c0 * 9 + t], a.w[(long)(c0 + 1) * 9 + t]};
    sc = *(const f32x2*)(a.pro.scale + c0); sh = *(const f32x2*)(a.pro.shift + c0);
    mu = *(const f32x2*)(a.mean + c0); rs = *(const f32x2*)(a.rstd + c0);
  }
  for (int k = 0; k < g.spt; ++k) {
    const long strip = ((long)bx * g.spt + k) * 8 + sl;
    if (!cvalid || strip >= g.nstrips) continue;
    const int seg = (int)(strip % g.nseg);
    const long bt = strip / g.nseg;
    const int band = (int)(bt % g.nbands), img = (int)(bt / g.nbands);
    const int iy0 = band * R, ix0 = seg * g.L;
    const int ncol = (a.IW - ix0 < g.L) ? a.IW - ix0 : g.L;
    const long ibase = (long)img * a.IH * a.IW * C + c0;  // OH == IH, OW == IW
    const T* xim = (const T*)a.x + ibase;
    const T* dyim = (const T*)a.dy + ibase;
    T* gim = (T*)a.g + ibase;
    int xoff[R], doff[NR], dok = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) xoff[r] = clampi(iy0 + r, 0, a.IH - 1) * a.IW * C;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int oy = iy0 - 1 + j;
      dok |= (oy >= 0 && oy < a.OH) ? (1 << j) : 0;
      doff[j] = clampi(oy, 0, a.OH - 1) * a.OW * C;
    }
    auto lddy = [&](int ox, raw_t (&raw)[NR]) {
      const int xo = clampi(ox, 0, a.OW - 1) * C;
#pragma unroll
      for (int j = 0; j < NR; ++j) raw[j] = P::ld(dyim + doff[j] + xo);
    };
    auto ldx = [&](int ix, raw_t (&raw)[R]) {
      const int xo = clampi(ix, 0, a.IW - 1) * C;
#pragma unroll
      for (int r = 0; r < R; ++r) raw[r] = P::ld(xim + xoff[r] + xo);
    };
    f32x2 dyw[NR][3];
    auto push = [&](int ox, const raw_t (&raw)[NR]) {
      const bool cok = ox >= 0 && ox < a.OW;
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const bool ok = cok && ((dok >> j) & 1);
        dyw[j][0] = dyw[j][1]; dyw[j][1] = dyw[j][2];
        dyw[j][2] = ok ? P::up(raw[j]) : splat2(0.f);
      }
    };
    // ring of three register sets: three columns of dy and x loads in flight per thread (see dw2_fwd_kernel)
    raw_t rdy[3][NR], rx[3][R];
#pragma unroll
    for (int j = 0; j < NR; ++j) { dyw[j][1] = splat2(0.f); dyw[j][2] = splat2(0.f); }
    lddy(ix0 - 1, rdy[0]); lddy(ix0, rdy[1]); lddy(ix0 + 1, rdy[2]); ldx(ix0, rx[2]);
    push(ix0 - 1, rdy[0]); lddy(ix0 + 2, rdy[0]); ldx(ix0 + 1, rx[0]);
    push(ix0, rdy[1]); lddy(ix0 + 3, rdy[1]); ldx(ix0 + 2, rx[1]);
    auto step = [&](int i, raw_t (&cdy)[NR], raw_t (&cxr)[R]) {
      push(ix0 + i + 1, cdy);
      f32x2 xv[R];
#pragma unroll
      for (int r = 0; r < R; ++r) xv[r] = P::up(cxr[r]);
      lddy(ix0 + i + 4, cdy); ldx(ix0 + i + 3, cxr);  // prefetch (clamped)
      if (i < ncol) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const bool rok = iy0 + r < a.IH;
          const f32x2 z = xv[r] * sc + sh;
          const f32x2 sg = sigmoid2(z);
          const f32x2 act = rok ? z * sg : splat2(0.f);
          f32x2 da = splat2(0.f);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const f32x2 d = dyw[r + 2 - ky][2 - kx];
              da += d * w[ky][kx];
              dwacc[ky][kx] += d * act;
            }
          if (rok) {
            const f32x2 gv = da * (sg * (splat2(1.0f) + z * (splat2(1.0f) - sg)));
            P::st(gim + xoff[r] + (long)(ix0 + i) * C, gv);
            s1 += gv; s2 += gv * (xv[r] - mu);
          }
        }
      }
    };
    for (int i = 0; i < ncol; i += 3) {
      step(i, rdy[2], rx[2]);
      step(i + 1, rdy[0], rx[0]);
      step(i + 2, rdy[1], rx[1]);
    }
  }
  s2 *= rs;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    *(f32x2*)&dwl[sl][t][2 * cp] = dwacc[t / 3][t % 3];
  }
  red[sl][0][cp] = s1[0]; red[sl][1][cp] = s1[1]; red[sl][2][cp] = s2[0]; red[sl][3][cp] = s2[1];
  __syncthreads();
  for (int e = tid; e < 64 * 9; e += 256) {
    const int c = e / 9, t = e - c * 9;
    float v = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) v += dwl[s][t][c];
    if (cbeg + c < C) atomicAdd(a.dw + (long)cbeg * 9 + e, v);
  }
  if (tid < 128) {
    const int kk = tid >> 6, c = tid & 63;
    float t = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) t += red[s][kk * 2 + (c & 1)][c >> 1];
    if (cbeg + c < C) atomicAdd(a.stats + ((long)(bx % MDS_STAT_SLOTS) * 2 + kk) * C + cbeg + c, (double)t);
  }
}

// ------------------------------------------------------------------------------------ 3x3x3, T = 5
// Same register sliding window for the 3D blocks of the headline configuration (5 slices per stack):
// a thread owns one output ROW of all five slices of its channel pair and walks L columns; the
// 5 x 3 x 3 activated window stays in registers (each input value is activated 3x, once per output
// row that needs it), the 27 taps of the channel pair are read from LDS once per column.  The LDS-tiled
// kernel above had 216-324 blocks of five barrier-separated slices each for the whole launch.
#define DW3_T 5
template <typename T, bool POOL = false>
__global__ __launch_bounds__(256, 2) void dw3_fwd_kernel(mds_dw_fwd_args a, DwStrips g) {
  MDS_CHAIN_PRIO();
  constexpr int TT = DW3_T;
  typedef Pair<T> P;
  typedef typename P::raw_t raw_t;
  __shared__ f32x2 wl[27][32];
  __shared__ float red[8][4][32];
  const int tid = threadIdx.x, cp = tid & 31, sl = tid >> 5;
  const DwBlock db = dw_block(g);   // (strip block, channel chunk) of this workgroup, XCD-aware
  const int bx = db.bx;
  const int C = a.C, cbeg = db.chunk * 64, c0 = cbeg + 2 * cp;
  const bool cvalid = c0 < C;
  const int mode = a.pro.mode;
  const int emode = a.epi.mode;
  f32x2 esc = splat2(1.f), esh = splat2(0.f);
  if (emode != MDS_EPI_NONE && c0 < a.C) { esc = *(const f32x2*)(a.epi.scale + c0); esh = *(const f32x2*)(a.epi.shift + c0); }
  for (int e = tid; e < 27 * 32; e += 256) {
    const int t = e >> 5, c = cbeg + 2 * (e & 31);
    wl[t][e & 31] = c < C ? (f32x2){a.w[(long)c * 27 + t], a.w[(long)(c + 1) * 27 + t]} : splat2(0.f);
  }
  __syncthreads();
  f32x2 s1 = splat2(0.f), s2 = splat2(0.f), sp = splat2(0.f), sc = splat2(1.f), sh = splat2(0.f);
  int pimg = 0;
  bool pok = false;
  if (cvalid && mode != MDS_PRO_NONE) { sc = *(const f32x2*)(a.pro.scale + c0); sh = *(const f32x2*)(a.pro.shift + c0); }
  const int tstr = a.IH * a.IW * C;   // slice stride (elements)
  for (int k = 0; k < g.spt; ++k) {
    const long strip = ((long)bx * g.spt + k) * 8 + sl;
    if (strip < g.nstrips) { pimg = (int)(strip / ((long)g.nseg * g.nbands)); pok = true; }
    if (!cvalid || strip >= g.nstrips) continue;
    const int seg = (int)(strip % g.nseg);
    const long bt = strip / g.nseg;
    const int oy = (int)(bt % g.nbands), n = (int)(bt / g.nbands);
    const int ox0 = seg * g.L;
    const int nout = (a.OW - ox0 < g.L) ? a.OW - ox0 : g.L;
    const T* xim = (const T*)a.x + (long)n * TT * tstr + c0;
    T* yim = (T*)a.y + (long)n * TT * tstr + (long)oy * a.OW * C + c0;
    int roff[3], rok = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int iy = oy - 1 + j;
      rok |= (iy >= 0 && iy < a.IH) ? (1 << j) : 0;
      roff[j] = clampi(iy, 0, a.IH - 1) * a.IW * C;
    }
    auto ldcol = [&](int ix, raw_t (&raw)[TT][3]) {
      const int xo = clampi(ix, 0, a.IW - 1) * C;
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int j = 0; j < 3; ++j) raw[t][j] = P::ld(xim + t * tstr + roff[j] + xo);
    };
    f32x2 win[TT][3][3];
    auto push = [&](int ix, const raw_t (&raw)[TT][3]) {
      const bool cok = ix >= 0 && ix < a.IW;
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          f32x2 v = P::up(raw[t][j]);
          if (mode != MDS_PRO_NONE) {
            v = v * sc + sh;
            if (mode != MDS_PRO_AFFINE) v = v * sigmoid2(v);
          }
          const bool ok = cok && ((rok >> j) & 1);
          win[t][j][0] = win[t][j][1]; win[t][j][1] = win[t][j][2];
          win[t][j][2] = ok ? v : splat2(0.f);
        }
    };
    raw_t raw[TT][3], cur[TT][3];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int j = 0; j < 3; ++j) { win[t][j][1] = splat2(0.f); win[t][j][2] = splat2(0.f); }
    ldcol(ox0 - 1, raw); push(ox0 - 1, raw);
    ldcol(ox0, raw); push(ox0, raw);
    ldcol(ox0 + 1, raw);
#pragma unroll 1
    for (int o = 0; o < nout; ++o) {
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int j = 0; j < 3; ++j) cur[t][j] = raw[t][j];
      ldcol(ox0 + o + 2, raw);
      push(ox0 + o + 1, cur);
      f32x2 acc[TT];
#pragma unroll
      for (int t = 0; t < TT; ++t) acc[t] = splat2(0.f);
#pragma unroll
      for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const f32x2 wv = wl[(dt * 3 + ky) * 3 + kx][cp];
#pragma unroll
            for (int ot = 0; ot < TT; ++ot) {
              const int it = ot + dt - 1;
              if (it >= 0 && it < TT) acc[ot] += win[it][ky][kx] * wv;
            }
          }
#pragma unroll
      for (int ot = 0; ot < TT; ++ot) {
        const raw_t pk = P::pk(epi2(acc[ot], emode, esc, esh));
        P::str(yim + (long)ot * tstr + (long)(ox0 + o) * C, pk);
        s1 += acc[ot]; s2 += acc[ot] * acc[ot]; if (POOL) sp += P::up(pk);
      }
    }
  }
  if (a.stats) {
    red[sl][0][cp] = s1[0]; red[sl][1][cp] = s1[1]; red[sl][2][cp] = s2[0]; red[sl][3][cp] = s2[1];
    __syncthreads();
    if (tid < 128) {
      const int kk = tid >> 6, c = tid & 63;
      float t = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) t += red[s][kk * 2 + (c & 1)][c >> 1];
      if (cbeg + c < C) atomicAdd(a.stats + ((long)(bx % MDS_STAT_SLOTS) * 2 + kk) * C + cbeg + c, (double)t);
    }
  }
  if (POOL && a.pool) dw_pool_flush(red, sp, pimg, pok, a.pool, a.pool_inv, C, cbeg);
}

// ------------------------------------------------------------------------------------ 3x3x3, any T (time chunks)
// The same register sliding window for stacks of any length (the 33-frame configuration: T = 11).  A strip is (image, TIME CHUNK of
// TO output slices, output row, column segment): the thread keeps the activated window of the chunk's TO + 2 input slices (the
// chunk and one halo slice on either side; slices outside the stack count as zeros), so the 27 taps need no boundary tests and
// an input slice is read by at most two chunks.  TO = 4: 54 window registers pairs (T = 11 -> chunks of 4 / 4 / 3 outputs,
// 6 + 6 + 4 input slices for 11: 1.45 x, the second reading from L2).  Replaces the LDS-tiled kernel at the top of this file for
// these shapes: 166 -> 84 us at 4 x 11 x 23 x 40 x 576.  (The BACKWARD counterpart - chunks of three input slices under a five-slice
// dy window - measured 238 us against the LDS-tiled kernel's 183: four times the blocks, each ending in its 64 x 27 filter-gradient
// atomics, and 42 spilled registers; not adopted, profiles/LOG.md.)
#define DW3G_TO 4
template <typename T>
__global__ __launch_bounds__(256, 2) void dw3g_fwd_kernel(mds_dw_fwd_args a, DwStrips g, int nchunks_t) {
  MDS_CHAIN_PRIO();
  constexpr int TO = DW3G_TO, TW = TO + 2;
  typedef Pair<T> P;
  typedef typename P::raw_t raw_t;
  __shared__ f32x2 wl[27][32];
  __shared__ float red[8][4][32];
  const int tid = threadIdx.x, cp = tid & 31, sl = tid >> 5;
  const DwBlock db = dw_block(g);   // (strip block, channel chunk) of this workgroup, XCD-aware
  const int bx = db.bx;
  const int C = a.C, cbeg = db.chunk * 64, c0 = cbeg + 2 * cp;
  const bool cvalid = c0 < C;
  const int mode = a.pro.mode;
  for (int e = tid; e < 27 * 32; e += 256) {
    const int t = e >> 5, c = cbeg + 2 * (e & 31);
    wl[t][e & 31] = c < C ? (f32x2){a.w[(long)c * 27 + t], a.w[(long)(c + 1) * 27 + t]} : splat2(0.f);
  }
  __syncthreads();
  f32x2 s1 = splat2(0.f), s2 = splat2(0.f), sc = splat2(1.f), sh = splat2(0.f);
  if (cvalid && mode != MDS_PRO_NONE) { sc = *(const f32x2*)(a.pro.scale + c0); sh = *(const f32x2*)(a.pro.shift + c0); }
  const int tstr = a.IH * a.IW * C;   // slice stride (elements)
  for (int k = 0; k < g.spt; ++k) {
    const long strip = ((long)bx * g.spt + k) * 8 + sl;
    if (!cvalid || strip >= g.nstrips) continue;
    const int seg = (int)(strip % g.nseg);
    const long bt = strip / g.nseg;
    const int oy = (int)(bt % g.nbands), nc = (int)(bt / g.nbands);   // "image" of the strip geometry = (stack, time chunk)
    const int n = nc / nchunks_t, t0 = (nc - n * nchunks_t) * TO;
    const int ox0 = seg * g.L;
    const int nout = (a.OW - ox0 < g.L) ? a.OW - ox0 : g.L;
    const T* xim = (const T*)a.x + (long)n * a.T * tstr + c0;
    T* yim = (T*)a.y + (long)n * a.T * tstr + (long)oy * a.OW * C + c0;
    int roff[3], rok = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int iy = oy - 1 + j;
      rok |= (iy >= 0 && iy < a.IH) ? (1 << j) : 0;
      roff[j] = clampi(iy, 0, a.IH - 1) * a.IW * C;
    }
    long toff[TW];                       // window slot s = input slice t0 - 1 + s (clamped; tok: inside the stack)
    int tok = 0;
#pragma unroll
    for (int s_ = 0; s_ < TW; ++s_) {
      const int it = t0 - 1 + s_;
      tok |= (it >= 0 && it < a.T) ? (1 << s_) : 0;
      toff[s_] = (long)clampi(it, 0, a.T - 1) * tstr;
    }
    auto ldcol = [&](int ix, raw_t (&raw)[TW][3]) {
      const int xo = clampi(ix, 0, a.IW - 1) * C;
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int j = 0; j < 3; ++j) raw[t][j] = P::ld(xim + toff[t] + roff[j] + xo);
    };
    f32x2 win[TW][3][3];
    auto push = [&](int ix, const raw_t (&raw)[TW][3]) {
      const bool cok = ix >= 0 && ix < a.IW;
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          f32x2 v = P::up(raw[t][j]);
          if (mode != MDS_PRO_NONE) {
            v = v * sc + sh;
            if (mode != MDS_PRO_AFFINE) v = v * sigmoid2(v);
          }
          const bool ok = cok && ((rok >> j) & 1) && ((tok >> t) & 1);
          win[t][j][0] = win[t][j][1]; win[t][j][1] = win[t][j][2];
          win[t][j][2] = ok ? v : splat2(0.f);
        }
    };
    raw_t raw[TW][3], cur[TW][3];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int j = 0; j < 3; ++j) { win[t][j][1] = splat2(0.f); win[t][j][2] = splat2(0.f); }
    ldcol(ox0 - 1, raw); push(ox0 - 1, raw);
    ldcol(ox0, raw); push(ox0, raw);
    ldcol(ox0 + 1, raw);
#pragma unroll 1
    for (int o = 0; o < nout; ++o) {
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int j = 0; j < 3; ++j) cur[t][j] = raw[t][j];
      ldcol(ox0 + o + 2, raw);
      push(ox0 + o + 1, cur);
      f32x2 acc[TO];
#pragma unroll
      for (int t = 0; t < TO; ++t) acc[t] = splat2(0.f);
#pragma unroll
      for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const f32x2 wv = wl[(dt * 3 + ky) * 3 + kx][cp];
#pragma unroll
            for (int ot = 0; ot < TO; ++ot) acc[ot] += win[ot + dt][ky][kx] * wv;   // output slice t0 + ot reads slices t0 + ot - 1 .. + 1 = slots ot .. ot + 2
          }
#pragma unroll
      for (int ot = 0; ot < TO; ++ot) {
        if (t0 + ot < a.T) {
          P::str(yim + (long)(t0 + ot) * tstr + (long)(ox0 + o) * C, P::pk(acc[ot]));
          s1 += acc[ot]; s2 += acc[ot] * acc[ot];
        }
      }
    }
  }
  if (a.stats) {
    red[sl][0][cp] = s1[0]; red[sl][1][cp] = s1[1]; red[sl][2][cp] = s2[0]; red[sl][3][cp] = s2[1];
    __syncthreads();
    if (tid < 128) {
      const int kk = tid >> 6, c = tid & 63;
      float t = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) t += red[s][kk * 2 + (c & 1)][c >> 1];
      if (cbeg + c < C) atomicAdd(a.stats + ((long)(bx % MDS_STAT_SLOTS) * 2 + kk) * C + cbeg + c, (double)t);
    }
  }
}

// backward: window of dy [5 slices][3 rows][3 cols] around the thread's input row; per input pixel
//   da[it] = sum_{dt,ky,kx} dy[it+1-dt][iy+1-ky][ix+1-kx] * w[dt][ky][kx],   dw[tap] += act[it] * (same dy)
#ifndef MDS_DW3B_OCC
#define MDS_DW3B_OCC 2     /* bf16: two blocks per CU with 38 spilled VGPRs (round 3: already at two blocks per CU; one block per CU halves the waves in flight of a kernel bound by its loads) */
#endif
template <typename T>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? 1 : MDS_DW3B_OCC) void dw3_bwd_kernel(mds_dw_bwd_args a, DwStrips g) {
  MDS_CHAIN_PRIO();
  constexpr int TT = DW3_T;
  typedef Pair<T> P;
  typedef typename P::raw_t raw_t;
  __shared__ f32x2 wl[27][32];
  __shared__ float dwl[8][27][64];
  __shared__ float red[8][4][32];
  const int tid = threadIdx.x, cp = tid & 31, sl = tid >> 5;
  const DwBlock db = dw_block(g);   // (strip block, channel chunk) of this workgroup, XCD-aware
  const int bx = db.bx;
  const int C = a.C, cbeg = db.chunk * 64, c0 = cbeg + 2 * cp;
  const bool cvalid = c0 < C;
  for (int e = tid; e < 27 * 32; e += 256) {
    const int t = e >> 5, c = cbeg + 2 * (e & 31);
    wl[t][e & 31] = c < C ? (f32x2){a.w[(long)c * 27 + t], a.w[(long)(c + 1) * 27 + t]} : splat2(0.f);
  }
  __syncthreads();
  f32x2 s1 = splat2(0.f), s2 = splat2(0.f), sc = splat2(0.f), sh = splat2(0.f), mu = splat2(0.f), rs = splat2(0.f);
  f32x2 dwacc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) dwacc[t] = splat2(0.f);
  if (cvalid) {
    sc = *(const f32x2*)(a.pro.scale + c0); sh = *(const f32x2*)(a.pro.shift + c0);
    mu = *(const f32x2*)(a.mean + c0); rs = *(const f32x2*)(a.rstd + c0);
  }
  const int tstr = a.IH * a.IW * C;
  for (int k = 0; k < g.spt; ++k) {
    const long strip = ((long)bx * g.spt + k) * 8 + sl;
    if (!cvalid || strip >= g.nstrips) continue;
    const int seg = (int)(strip % g.nseg);
    const long bt = strip / g.nseg;
    const int iy = (int)(bt % g.nbands), n = (int)(bt / g.nbands);
    const int ix0 = seg * g.L;
    const int ncol = (a.IW - ix0 < g.L) ? a.IW - ix0 : g.L;
    const long nbase = (long)n * TT * tstr + c0;
    const T* xim = (const T*)a.x + nbase + (long)iy * a.IW * C;
    const T* dyim = (const T*)a.dy + nbase;
    T* gim = (T*)a.g + nbase + (long)iy * a.IW * C;
    int doff[3], dok = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int oy = iy - 1 + j;
      dok |= (oy >= 0 && oy < a.OH) ? (1 << j) : 0;
      doff[j] = clampi(oy, 0, a.OH - 1) * a.OW * C;
    }
    auto lddy = [&](int ox, raw_t (&raw)[TT][3]) {
      const int xo = clampi(ox, 0, a.OW - 1) * C;
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int j = 0; j < 3; ++j) raw[t][j] = P::ld(dyim + t * tstr + doff[j] + xo);
    };
    auto ldx = [&](int ix, raw_t (&raw)[TT]) {
      const int xo = clampi(ix, 0, a.IW - 1) * C;
#pragma unroll
      for (int t = 0; t < TT; ++t) raw[t] = P::ld(xim + t * tstr + xo);
    };
    f32x2 dyw[TT][3][3];
    auto push = [&](int ox, const raw_t (&raw)[TT][3]) {
      const bool cok = ox >= 0 && ox < a.OW;
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const bool ok = cok && ((dok >> j) & 1);
          dyw[t][j][0] = dyw[t][j][1]; dyw[t][j][1] = dyw[t][j][2];
          dyw[t][j][2] = ok ? P::up(raw[t][j]) : splat2(0.f);
        }
    };
    raw_t rdy[TT][3], cdy[TT][3], rx[TT], cx[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int j = 0; j < 3; ++j) { dyw[t][j][1] = splat2(0.f); dyw[t][j][2] = splat2(0.f); }
    lddy(ix0 - 1, rdy); push(ix0 - 1, rdy);
    lddy(ix0, rdy); push(ix0, rdy);
    lddy(ix0 + 1, rdy); ldx(ix0, rx);
#pragma unroll 1
    for (int i = 0; i < ncol; ++i) {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        cx[t] = rx[t];
#pragma unroll
        for (int j = 0; j < 3; ++j) cdy[t][j] = rdy[t][j];
      }
      lddy(ix0 + i + 2, rdy); ldx(ix0 + i + 1, rx);
      push(ix0 + i + 1, cdy);
      f32x2 xv[TT], z[TT], sg[TT], act[TT], da[TT];
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        xv[t] = P::up(cx[t]);
        z[t] = xv[t] * sc + sh;
        sg[t] = sigmoid2(z[t]);
        act[t] = z[t] * sg[t];
        da[t] = splat2(0.f);
      }
#pragma unroll
      for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int tap = (dt * 3 + ky) * 3 + kx;
            const f32x2 wv = wl[tap][cp];
#pragma unroll
            for (int it = 0; it < TT; ++it) {
              const int ot = it + 1 - dt;
              if (ot >= 0 && ot < TT) {
                const f32x2 d = dyw[ot][2 - ky][2 - kx];
                da[it] += d * wv;
                dwacc[tap] += d * act[it];
              }
            }
          }
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const f32x2 gv = da[t] * (sg[t] * (splat2(1.0f) + z[t] * (splat2(1.0f) - sg[t])));
        P::st(gim + (long)t * tstr + (long)(ix0 + i) * C, gv);
        s1 += gv; s2 += gv * (xv[t] - mu);
      }
    }
  }
  s2 *= rs;
#pragma unroll
  for (int t = 0; t < 27; ++t) *(f32x2*)&dwl[sl][t][2 * cp] = dwacc[t];
  red[sl][0][cp] = s1[0]; red[sl][1][cp] = s1[1]; red[sl][2][cp] = s2[0]; red[sl][3][cp] = s2[1];
  __syncthreads();
  for (int e = tid; e < 64 * 27; e += 256) {
    const int c = e / 27, t = e - c * 27;
    float v = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) v += dwl[s][t][c];
    if (cbeg + c < C) atomicAdd(a.dw + (long)cbeg * 27 + e, v);
  }
  if (tid < 128) {
    const int kk = tid >> 6, c = tid & 63;
    float t = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) t += red[s][kk * 2 + (c & 1)][c >> 1];
    if (cbeg + c < C) atomicAdd(a.stats + ((long)(bx % MDS_STAT_SLOTS) * 2 + kk) * C + cbeg + c, (double)t);
  }
}

// ------------------------------------------------------------------------------------ 3x3 stride 2 (TF-SAME)
// Sliding window for the two stride-2 layers.  Forward: R = 3 output rows need 7 input rows; an output
// column consumes two new input columns (window col 0 <- old col 2).
template <typename T, bool POOL = false>
__global__ __launch_bounds__(256, 3) void dw2s_fwd_kernel(mds_dw_fwd_args a, DwStrips g) {
  MDS_CHAIN_PRIO();
  constexpr int R = 3, NR = 2 * R + 1;
  typedef Pair<T> P;
  typedef typename P::raw_t raw_t;
  __shared__ float red[8][4][32];
  const int tid = threadIdx.x, cp = tid & 31, sl = tid >> 5;
  const DwBlock db = dw_block(g);   // (strip block, channel chunk) of this workgroup, XCD-aware
  const int bx = db.bx;
  const int C = a.C, cbeg = db.chunk * 64, c0 = cbeg + 2 * cp;
  const bool cvalid = c0 < C;
  const int mode = a.pro.mode;
  const int emode = a.epi.mode;
  f32x2 esc = splat2(1.f), esh = splat2(0.f);
  if (emode != MDS_EPI_NONE && c0 < a.C) { esc = *(const f32x2*)(a.epi.scale + c0); esh = *(const f32x2*)(a.epi.shift + c0); }
  f32x2 s1 = splat2(0.f), s2 = splat2(0.f), sp = splat2(0.f);
  int pimg = 0;
  bool pok = false;
  f32x2 w[3][3], sc = splat2(1.f), sh = splat2(0.f);
  if (cvalid) {
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t / 3][t % 3] = (f32x2){a.w[(long)c0 * 9 + t], a.w[(long)(c0 + 1) * 9 + t]};
    if (mode != MDS_PRO_NONE) { sc = *(const f32x2*)(a.pro.scale + c0); sh = *(const f32x2*)(a.pro.shift + c0); }
  }
  for (int k = 0; k < g.spt; ++k) {
    const long strip = ((long)bx * g.spt + k) * 8 + sl;
    if (strip < g.nstrips) { pimg = (int)(strip / ((long)g.nseg * g.nbands)); pok = true; }
    if (!cvalid || strip >= g.nstrips) continue;
    const int seg = (int)(strip % g.nseg);
    const long bt = strip / g.nseg;
    const int band = (int)(bt % g.nbands), img = (int)(bt / g.nbands);
    const int oy0 = band * R, ox0 = seg * g.L;
    const int nout = (a.OW - ox0 < g.L) ? a.OW - ox0 : g.L;
    const T* xim = (const T*)a.x + (long)img * a.IH * a.IW * C + c0;
    T* yim = (T*)a.y + ((long)img * a.OH + oy0) * a.OW * C + c0;
    int roff[NR], rok = 0;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int iy = 2 * oy0 - a.pad_t + j;
      rok |= (iy >= 0 && iy < a.IH) ? (1 << j) : 0;
      roff[j] = clampi(iy, 0, a.IH - 1) * a.IW * C;
    }
    auto ldcol = [&](int ix, raw_t (&raw)[NR]) {
      const int xo = clampi(ix, 0, a.IW - 1) * C;
#pragma unroll
      for (int j = 0; j < NR; ++j) raw[j] = P::ld(xim + roff[j] + xo);
    };
    f32x2 win[NR][3];
    auto activate = [&](int ix, const raw_t (&raw)[NR], int slot) {
      const bool cok = ix >= 0 && ix < a.IW;
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        f32x2 v = P::up(raw[j]);
        if (mode != MDS_PRO_NONE) {
          v = v * sc + sh;
          if (mode != MDS_PRO_AFFINE) v = v * sigmoid2(v);
        }
        const bool ok = cok && ((rok >> j) & 1);
        v = ok ? v : splat2(0.f);
        if (slot == 1) win[j][1] = v; else win[j][2] = v;
      }
    };
    raw_t ra[NR], rb[NR], ca[NR], cb[NR];
    const int ixs = 2 * ox0 - a.pad_l;           // first input column of the strip
    ldcol(ixs, ra); activate(ixs, ra, 2);
    ldcol(ixs + 1, ra); ldcol(ixs + 2, rb);
#pragma unroll 1
    for (int o = 0; o < nout; ++o) {
      const int ixb = ixs + 2 * o;
#pragma unroll
      for (int j = 0; j < NR; ++j) { ca[j] = ra[j]; cb[j] = rb[j]; win[j][0] = win[j][2]; }
      ldcol(ixb + 3, ra); ldcol(ixb + 4, rb);    // prefetch the next output's two new columns (clamped)
      activate(ixb + 1, ca, 1);
      activate(ixb + 2, cb, 2);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        f32x2 acc = splat2(0.f);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) acc += win[2 * r + ky][kx] * w[ky][kx];
        if (oy0 + r < a.OH) {
          const raw_t pk = P::pk(epi2(acc, emode, esc, esh));
          P::str(yim + ((long)r * a.OW + ox0 + o) * C, pk);
          s1 += acc; s2 += acc * acc; if (POOL) sp += P::up(pk);
        }
      }
    }
  }
  if (a.stats) {
    red[sl][0][cp] = s1[0]; red[sl][1][cp] = s1[1]; red[sl][2][cp] = s2[0]; red[sl][3][cp] = s2[1];
    __syncthreads();
    if (tid < 128) {
      const int kk = tid >> 6, c = tid & 63;
      float t = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) t += red[s][kk * 2 + (c & 1)][c >> 1];
      if (cbeg + c < C) atomicAdd(a.stats + ((long)(bx % MDS_STAT_SLOTS) * 2 + kk) * C + cbeg + c, (double)t);
    }
  }
  if (POOL && a.pool) dw_pool_flush(red, sp, pimg, pok, a.pool, a.pool_inv, C, cbeg);
}

// Backward over INPUT pixels: a thread owns 4 input rows x pairs of input columns; the dy values that
// reach them are a 3-row x 2-column window (slides one dy column per input-column pair).  With
// iy = 4b + r, ix = 2c + p:   tap ky hits iff (r + PT - ky) is even -> window row (r+PT-ky+2)/2 - PT,
// tap kx iff (p + PL - kx) is even -> window column (p+PL-kx+2)/2 - PL   (all compile-time).
template <typename T, int PT, int PL>
__global__ __launch_bounds__(256, 2) void dw2s_bwd_kernel(mds_dw_bwd_args a, DwStrips g) {
  MDS_CHAIN_PRIO();
  constexpr int R = 4;
  typedef Pair<T> P;
  typedef typename P::raw_t raw_t;
  __shared__ float dwl[8][9][64];
  __shared__ float red[8][4][32];
  const int tid = threadIdx.x, cp = tid & 31, sl = tid >> 5;
  const DwBlock db = dw_block(g);   // (strip block, channel chunk) of this workgroup, XCD-aware
  const int bx = db.bx;
  const int C = a.C, cbeg = db.chunk * 64, c0 = cbeg + 2 * cp;
  const bool cvalid = c0 < C;
  f32x2 s1 = splat2(0.f), s2 = splat2(0.f);
  f32x2 w[3][3], dwacc[3][3], sc = splat2(0.f), sh = splat2(0.f), mu = splat2(0.f), rs = splat2(0.f);
#pragma unroll
  for (int t = 0; t < 9; ++t) dwacc[t / 3][t % 3] = splat2(0.f);
  if (cvalid) {
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t / 3][t % 3] = (f32x2){a.w[(long)c0 * 9 + t], a.w[(long)(c0 + 1) * 9 + t]};
    sc = *(const f32x2*)(a.pro.scale + c0); sh = *(const f32x2*)(a.pro.shift + c0);
    mu = *(const f32x2*)(a.mean + c0); rs = *(const f32x2*)(a.rstd + c0);
  }
  for (int k = 0; k < g.spt; ++k) {
    const long strip = ((long)bx * g.spt + k) * 8 + sl;
    if (!cvalid || strip >= g.nstrips) continue;
    const int seg = (int)(strip % g.nseg);
    const long bt = strip / g.nseg;
    const int band = (int)(bt % g.nbands), img = (int)(bt / g.nbands);
    const int iy0 = band * R, ix0 = seg * g.L;          // g.L is even
    const int npair = ((a.IW - ix0 < g.L ? a.IW - ix0 : g.L) + 1) >> 1;
    const T* xim = (const T*)a.x + (long)img * a.IH * a.IW * C + c0;
    T* gim = (T*)a.g + (long)img * a.IH * a.IW * C + c0;
    const T* dyim = (const T*)a.dy + (long)img * a.OH * a.OW * C + c0;
    const int oyb = iy0 / 2 - 1 + PT;
    int xoff[R], doff[3], dok = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) xoff[r] = clampi(iy0 + r, 0, a.IH - 1) * a.IW * C;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int oy = oyb + j;
      dok |= (oy >= 0 && oy < a.OH) ? (1 << j) : 0;
      doff[j] = clampi(oy, 0, a.OH - 1) * a.OW * C;
    }
    auto lddy = [&](int ox, raw_t (&raw)[3]) {
      const int xo = clampi(ox, 0, a.OW - 1) * C;
#pragma unroll
      for (int j = 0; j < 3; ++j) raw[j] = P::ld(dyim + doff[j] + xo);
    };
    auto ldx = [&](int ix, raw_t (&raw)[R][2]) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int xo = clampi(ix + p, 0, a.IW - 1) * C;
#pragma unroll
        for (int r = 0; r < R; ++r) raw[r][p] = P::ld(xim + xoff[r] + xo);
      }
    };
    f32x2 dyw[3][2];
    auto push = [&](int ox, const raw_t (&raw)[3]) {
      const bool cok = ox >= 0 && ox < a.OW;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const bool ok = cok && ((dok >> j) & 1);
        dyw[j][0] = dyw[j][1];
        dyw[j][1] = ok ? P::up(raw[j]) : splat2(0.f);
      }
    };
    raw_t rdy[3], cdy[3], rx[R][2], cx[R][2];
    const int cc0 = ix0 >> 1;
#pragma unroll
    for (int j = 0; j < 3; ++j) dyw[j][1] = splat2(0.f);
    lddy(cc0 - 1 + PL, rdy); push(cc0 - 1 + PL, rdy);
    lddy(cc0 + PL, rdy); ldx(ix0, rx);
#pragma unroll 1
    for (int c = 0; c < npair; ++c) {
#pragma unroll
      for (int j = 0; j < 3; ++j) cdy[j] = rdy[j];
#pragma unroll
      for (int r = 0; r < R; ++r) { cx[r][0] = rx[r][0]; cx[r][1] = rx[r][1]; }
      lddy(cc0 + c + 1 + PL, rdy); ldx(ix0 + 2 * c + 2, rx);   // prefetch (clamped)
      push(cc0 + c + PL, cdy);
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int iy = iy0 + r, ix = ix0 + 2 * c + p;
          const bool ok = iy < a.IH && ix < a.IW;
          const f32x2 xv = P::up(cx[r][p]);
          const f32x2 z = xv * sc + sh;
          const f32x2 sg = sigmoid2(z);
          const f32x2 act = ok ? z * sg : splat2(0.f);
          f32x2 da = splat2(0.f);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            if (((r + PT - ky + 2) & 1) == 0) {
              const int jr = (r + PT - ky + 2) / 2 - PT;
#pragma unroll
              for (int kx = 0; kx < 3; ++kx) {
                if (((p + PL - kx + 2) & 1) == 0) {
                  const int jc = (p + PL - kx + 2) / 2 - PL;
                  const f32x2 d = dyw[jr][jc];
                  da += d * w[ky][kx];
                  dwacc[ky][kx] += d * act;
                }
              }
            }
          }
          if (ok) {
            const f32x2 gv = da * (sg * (splat2(1.0f) + z * (splat2(1.0f) - sg)));
            P::st(gim + xoff[r] + (long)ix * C, gv);
            s1 += gv; s2 += gv * (xv - mu);
          }
        }
    }
  }
  s2 *= rs;
#pragma unroll
  for (int t = 0; t < 9; ++t) *(f32x2*)&dwl[sl][t][2 * cp] = dwacc[t / 3][t % 3];
  red[sl][0][cp] = s1[0]; red[sl][1][cp] = s1[1]; red[sl][2][cp] = s2[0]; red[sl][3][cp] = s2[1];
  __syncthreads();
  for (int e = tid; e < 64 * 9; e += 256) {
    const int c = e / 9, t = e - c * 9;
    float v = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) v += dwl[s][t][c];
    if (cbeg + c < C) atomicAdd(a.dw + (long)cbeg * 9 + e, v);
  }
  if (tid < 128) {
    const int kk = tid >> 6, c = tid & 63;
    float t = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) t += red[s][kk * 2 + (c & 1)][c >> 1];
    if (cbeg + c < C) atomicAdd(a.stats + ((long)(bx % MDS_STAT_SLOTS) * 2 + kk) * C + cbeg + c, (double)t);
  }
}

// strips per launch: aim at >= one full round of the chip (256 CUs x 16 waves) before lengthening strips
static DwStrips dw_strips(int images, int H, int W, int C, int R, int want_L = 0, bool even_L = false) {
  DwStrips g;
  g.nchunks = cdiv(C, 64);
  g.nbands = cdiv(H, R);
  int L = 16;
  if (want_L) L = want_L;
  else if ((long)images * g.nbands * cdiv(W, 16) * g.nchunks < 8192) L = 8;
  g.nseg = cdiv(W, L);
  g.L = cdiv(W, g.nseg);
  if (even_L) g.L = (g.L + 1) & ~1;   // stride-2 backward walks input-column PAIRS
  g.nseg = cdiv(W, g.L);
  g.nstrips = (long)images * g.nbands * g.nseg;
  g.spt = 1;   // strips per thread: 2 and 4 measured slower at every layer shape
  // optional grid order (MDS_KNOB_DW_ORDER = 1): channel chunk fastest - the blocks in flight together then cover ALL
  // chunks of the same pixels, so a pixel's whole channel row is fetched at about the same time
  g.swap = mds_knob(MDS_KNOB_DW_ORDER);                // 0: XCD-aware remap (dw_block), 1 / 2: the former orders
  return g;
}

// Strip length of the 3x3 stride-1 sliding-window kernels: the LONGEST strips (least halo: a strip of L columns reads L + 2) that
// still give 2.5 blocks per CU.  Measured at 46 x 80 x 672 (20 images): forward 71.5 / 67.3 / 64.3 / 62.8 / 63.6 us and backward
// 99.6 / 91.2 / 89.7 / 82.7 / 89.2 us at L = 8 / 16 / 20 / 27 / 40 (forward 1760 ... 440 blocks, backward 2640 ... 660);
// at 23 x 40 x 1152 everything from L = 8 to 20 is within the noise and L = 40 (180 blocks) loses 40 %.
static int dw2_len(int images, int H, int W, int C, int R) {
  if (mds_knob(MDS_KNOB_DW2_L)) return mds_knob(MDS_KNOB_DW2_L) == 1 ? 0 : mds_knob(MDS_KNOB_DW2_L);   // 1: the former rule (dw_strips)
  const long chunks = cdiv(C, 64), bands = cdiv(H, R);
  int nseg = cdiv(W, 32);      // (never longer than 32 columns: with 132 images - the 33-frame configuration - whole 80-column rows measured 0.6 % slower per step)
  while (cdiv(W, nseg + 1) >= 8 && cdiv((long)images * bands * nseg, 8) * chunks < (mds_knob(MDS_KNOB_DW2_BLOCKS) > 0 ? mds_knob(MDS_KNOB_DW2_BLOCKS) : 640)) ++nseg;
  return cdiv(W, nseg);
}

// Strip length of the 3x3x3 sliding-window kernels.  They hold two blocks per CU (230-256 VGPRs), i.e. 512 blocks in flight, and a
// launch of the headline shape (4 x 23 x 40 x 576 x 5 slices) is a few hundred blocks: the time is set by the number of ROUNDS, not by
// the strip length (forward 40 / 39 / 42 / 33 / 43 us, backward 66 / 77 / 78 / 54 / 81 us at L = 16-20 / 4 / 8 / 10 / 20: 207, 1035, 522,
// 414, 207 blocks - L = 8 is one block past a full round).  So: the shortest strips (>= 5 columns) that still fit ONE round.
static int dw3_len(int images, int H, int W, int C, int former) {
  if (mds_knob(MDS_KNOB_DW3_L)) return mds_knob(MDS_KNOB_DW3_L) == 1 ? former : mds_knob(MDS_KNOB_DW3_L);   // 1: the former fixed length
  const long chunks = cdiv(C, 64);
  int best = 1;
  for (int nseg = 1; nseg <= 8 && cdiv(W, nseg) >= 5; ++nseg)
    if (cdiv((long)images * H * nseg, 8) * chunks <= 512) best = nseg;
  return cdiv(W, best);
}

static dim3 dw_grid(DwStrips& g) {
  const int sb = cdiv(g.nstrips, 8 * g.spt);
  if (sb >= 65536 && g.swap == 1) g.swap = 2;   // grid.y limit
  return g.swap == 1 ? dim3(g.nchunks, sb) : dim3(sb, g.nchunks);
}

extern "C" int mds_dw_fwd(const mds_dw_fwd_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->N > 0 && a->T > 0 && a->C % 8 == 0, "dw_fwd: bad dims");
  MDS_REQUIRE(a->kt == 1 || a->kt == 3, "dw_fwd: kt must be 1 or 3");
  MDS_REQUIRE(a->stride == 1 || a->stride == 2, "dw_fwd: stride");
  MDS_REQUIRE(!(a->kt == 3 && a->stride == 2), "dw_fwd: 3x3x3 is stride 1 only");
  MDS_REQUIRE(a->kt == 3 || a->T >= 1, "dw_fwd: T");
  MDS_REQUIRE(a->x && a->w && a->y, "dw_fwd: null pointer");
  MDS_REQUIRE(a->pro.mode == MDS_PRO_NONE || (a->pro.scale && a->pro.shift), "dw_fwd: prologue");
  MDS_REQUIRE(a->pro.mode != MDS_PRO_BN_SILU_GATE, "dw_fwd: gate prologue unsupported");
  MDS_REQUIRE((long)a->N * 64 < 65536, "dw_fwd: grid.z");
  MDS_REQUIRE(a->epi.mode == MDS_EPI_NONE || (a->epi.scale && a->epi.shift && !a->stats && (a->kt == 1 || a->T == DW3_T)),
              "dw_fwd: an output transform needs scale/shift, no statistics, and a sliding-window kernel (kt == 1, or T == %d)", DW3_T);
  MDS_REQUIRE(!a->pool || (a->epi.mode != MDS_EPI_NONE && !a->stats && a->pool_inv > 0.f &&
                           ((a->kt == 1 && a->T == 1) || (a->kt == 3 && a->T == DW3_T))),
              "dw_fwd: pooling needs an output transform, no statistics, pool_inv, and a sliding-window kernel (kt == 1 with T == 1, or kt == 3 with T == %d)", DW3_T);
  if (a->kt == 1 && a->stride == 1) {
    MDS_REQUIRE(a->pad_t == 1 && a->pad_l == 1 && a->OH == a->IH && a->OW == a->IW, "dw_fwd: stride-1 geometry");
    // one or two images (the frame-by-frame predictor): 6-row bands with the shortest strips are 50-110 blocks - under one wave per
    // SIMD, every column a full memory round trip.  Two-row bands give three times the threads ((R + 2) / R = 2x the row reads,
    // served by L2).  MDS_KNOB_DW2_R: 1 = always six rows.
    const int images = a->N * a->T;
    const bool small = mds_knob(MDS_KNOB_DW2_R) != 1 &&
                       cdiv((long)images * cdiv(a->OH, 6) * cdiv(a->OW, 8), 8) * cdiv(a->C, 64) < 256;
    const int R = small ? 2 : 6;
    DwStrips g = dw_strips(images, a->OH, a->OW, a->C, R, dw2_len(images, a->OH, a->OW, a->C, R));
    dim3 grid = dw_grid(g), block(256);
    if (a->pool) {
      if (small) MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH((dw2_fwd_kernel<T, 2, true>), grid, block, 0, stream, *a, g));
      else MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH((dw2_fwd_kernel<T, 6, true>), grid, block, 0, stream, *a, g));
    } else {
      if (small) MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH((dw2_fwd_kernel<T, 2>), grid, block, 0, stream, *a, g));
      else MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH((dw2_fwd_kernel<T, 6>), grid, block, 0, stream, *a, g));
    }
    return mds_check_launch("dw_fwd");
  }
  if (a->kt == 1 && a->stride == 2) {
    DwStrips g = dw_strips(a->N * a->T, a->OH, a->OW, a->C, 3);
    dim3 grid = dw_grid(g), block(256);
    if (a->pool) MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH((dw2s_fwd_kernel<T, true>), grid, block, 0, stream, *a, g));
    else MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH((dw2s_fwd_kernel<T, false>), grid, block, 0, stream, *a, g));
    return mds_check_launch("dw_fwd");
  }
  if (a->kt == 3 && a->T == DW3_T) {
    DwStrips g = dw_strips(a->N, a->OH, a->OW, a->C, 1, dw3_len(a->N, a->OH, a->OW, a->C, 20));
    dim3 grid = dw_grid(g), block(256);
    if (a->pool) MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH((dw3_fwd_kernel<T, true>), grid, block, 0, stream, *a, g));
    else MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH((dw3_fwd_kernel<T, false>), grid, block, 0, stream, *a, g));
    return mds_check_launch("dw_fwd");
  }
  if (a->kt == 3 && a->stride == 1 && a->dtype == MDS_BF16 && a->epi.mode == MDS_EPI_NONE && !a->pool && mds_knob(MDS_KNOB_DW3G) != 1) {   // any other T: time chunks (fp32 would spill: it keeps the LDS-tiled kernel)
    const int nct = cdiv(a->T, DW3G_TO);
    // strips of ~10 columns (4 x 11 x 23 x 40 x 576: 112 / 94 / 84 / 94 / 96 us at 40 / 20 / 10 / 8 / 5 columns - two to three rounds
    // of blocks suit this kernel better than one round of long strips)
    const int want = mds_knob(MDS_KNOB_DW3_L) > 1 ? mds_knob(MDS_KNOB_DW3_L) : cdiv(a->OW, (a->OW + 5) / 10 > 0 ? (a->OW + 5) / 10 : 1);
    DwStrips g = dw_strips(a->N * nct, a->OH, a->OW, a->C, 1, want);
    dim3 grid = dw_grid(g), block(256);
    MDS_LAUNCH((dw3g_fwd_kernel<bf16_t>), grid, block, 0, stream, *a, g, nct);      // (bf16 only: no fp32 instantiation exists)
    return mds_check_launch("dw_fwd");
  }
  MDS_DISPATCH_DTYPE(a->dtype, T, {
    const int CC = DwCfg<T>::CC;
    const int nchunks = cdiv(a->C, CC);
    const int tiles_x = cdiv(a->OW, 16), tpb = 1;  // forward: parallelism beats amortisation
    dim3 grid(cdiv(tiles_x, tpb), cdiv(a->OH, 8), a->N * nchunks), block(256);
    const int S = a->stride, TH = 7 * S + 3, TW = 15 * S + 3;
    const size_t smem = (size_t)a->kt * TH * TW * CC * sizeof(T) + ((size_t)a->kt * 9 * CC + 4 * CC) * sizeof(float);
    if (a->kt == 3) MDS_LAUNCH((dw_fwd_kernel<T, 1, 3>), grid, block, smem, stream, *a, nchunks, tpb);
    else if (S == 1) MDS_LAUNCH((dw_fwd_kernel<T, 1, 1>), grid, block, smem, stream, *a, nchunks, tpb);
    else MDS_LAUNCH((dw_fwd_kernel<T, 2, 1>), grid, block, smem, stream, *a, nchunks, tpb);
  });
  return mds_check_launch("dw_fwd");
}

// ------------------------------------------------------------------------------------ backward
// Block = 8x16 INPUT pixels of a channel slab; thread = 4 channels x strip(s) of 4 pixels.
// Produces g = (dgrad) * silu'(z) (gradient wrt the BN output of the producing 1x1 conv), the
// BN-backward sums of g, and the filter gradient.  dy (+halo) is staged raw in LDS; x needs no
// halo and stays in registers.
template <typename T, int S, int PL, int KT>
__global__ __launch_bounds__(256) void dw_bwd_kernel(mds_dw_bwd_args a, int nchunks, int tpb) {
  constexpr int V = 4, CC = DwCfg<T>::CC, NCH = CC / V, NPT = 256 / NCH;
  constexpr int TIH = 8, TIW = 16, SPT = (TIH * TIW / 4) / NPT;  // strips of 4 pixels per thread
  constexpr int DTH = (S == 1) ? TIH + 2 : TIH / 2 + 2, DTW = (S == 1) ? TIW + 2 : TIW / 2 + 2, DNPIX = DTH * DTW;
  constexpr int MAXL = (DNPIX * NCH + 255) / 256;
  constexpr int NSEG = (S == 1) ? 6 : 4, NTAP = KT * 9;
  MDS_DYN_SMEM(smem);
  T* dyt = (T*)smem;                             // [KT][DNPIX][CC]
  float* wl = (float*)(dyt + KT * DNPIX * CC);   // [NTAP][CC]
  float* dwl = wl + NTAP * CC;                   // [CC][NTAP]
  double* st_l = (double*)(dwl + NTAP * CC);      // [2][CC] fp64
  const int tid = threadIdx.x, ch = tid % NCH, pt = tid / NCH;
  const int C = a.C;
  const int cz = blockIdx.z % nchunks, n = blockIdx.z / nchunks;
  const int cbeg = cz * CC, c0 = cbeg + ch * V;
  const bool cvalid = c0 < C;
  const int iy0 = blockIdx.y * TIH;
  const int oy_lo = iy0 / S - 1;
  int ix0 = 0, ox_lo = 0;  // set per tile: a block walks `tpb` consecutive tiles along W
  for (int e = tid; e < NTAP * CC; e += 256) {
    const int t = e / CC, c = e - t * CC;
    wl[e] = (cbeg + c < C) ? a.w[(long)(cbeg + c) * NTAP + t] : 0.f;
    dwl[e] = 0.f;
  }
  if (tid < 2 * CC) st_l[tid] = 0.0;
  float sc[V], sh[V], mu[V], rs[V];
  if (cvalid) {
    ldv<V>(a.pro.scale + c0, sc); ldv<V>(a.pro.shift + c0, sh);
    ldv<V>(a.mean + c0, mu); ldv<V>(a.rstd + c0, rs);
  }
  const T* x = (const T*)a.x;
  const T* dy = (const T*)a.dy;
  T* g = (T*)a.g;
  float st[2][V], dwacc[NTAP][V];
#pragma unroll
  for (int j = 0; j < V; ++j) { st[0][j] = 0.f; st[1][j] = 0.f; }
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int j = 0; j < V; ++j) dwacc[t][j] = 0.f;

  auto load_slice = [&](int ot, int slot) {
    const T* plane = dy + ((long)(n * a.T + ot) * a.OH * a.OW) * C + (cvalid ? c0 : 0);
    T* dst = dyt + (long)slot * DNPIX * CC + ch * V;
    Raw<T, V> raw[MAXL];
#pragma unroll
    for (int l = 0; l < MAXL; ++l) {
      const int pix = pt + NPT * l;
      const int ty = pix / DTW, tx = pix - ty * DTW;
      if (pix < DNPIX)
        raw[l].ld(plane + ((long)clampi(oy_lo + ty, 0, a.OH - 1) * a.OW + clampi(ox_lo + tx, 0, a.OW - 1)) * C);
    }
#pragma unroll
    for (int l = 0; l < MAXL; ++l) {
      const int pix = pt + NPT * l;
      const int ty = pix / DTW, tx = pix - ty * DTW;
      if (pix < DNPIX) {
        const int oy = oy_lo + ty, ox = ox_lo + tx;
        const bool ok = cvalid && oy >= 0 && oy < a.OH && ox >= 0 && ox < a.OW;
        float v[V];
        raw[l].get(v);
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = ok ? v[j] : 0.f;
        stv<T, V>(dst + (long)pix * CC, v);
      }
    }
  };

  auto compute = [&](int it, const Raw<T, V> (&xraw)[SPT][4]) {
#pragma unroll
    for (int sp = 0; sp < SPT; ++sp) {
      const int sid = pt + NPT * sp, r = sid / 4, sx = sid % 4;
      const int iy = iy0 + r, ixb = ix0 + 4 * sx;
      float xv[4][V], act[4][V], da[4][V];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const bool ok = cvalid && iy < a.IH && ixb + o < a.IW;
        xraw[sp][o].get(xv[o]);
#pragma unroll
        for (int j = 0; j < V; ++j) {
          act[o][j] = ok ? siluf_(xv[o][j] * sc[j] + sh[j]) : 0.f;
          da[o][j] = 0.f;
        }
      }
#pragma unroll
      for (int dt = 0; dt < KT; ++dt) {
        const int ot = it - dt + (KT == 3 ? 1 : 0);
        if (ot < 0 || ot >= a.T) continue;
        const T* src = dyt + (long)(KT == 3 ? (ot % 3) : 0) * DNPIX * CC + ch * V;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int num = iy + a.pad_t - ky;
          const int oy = num / S;
          if (num < 0 || (num % S) != 0 || oy >= a.OH) continue;
          const int trow = oy - oy_lo;
          float w3[3][V];
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) ldv<V>(wl + ((dt * 3 + ky) * 3 + kx) * CC + ch * V, w3[kx]);
#pragma unroll
          for (int s = 0; s < NSEG; ++s) {
            // dy column ox = seg_lo + s, seg_lo = ixb + PL - 2 (S=1) | ixb/2 - 1 (S=2); tile col = ox - ox_lo
            const int tcol = (S == 1) ? (4 * sx + PL - 1 + s) : (2 * sx + s);
            Raw<T, V> rv;
            rv.ld(src + (long)(trow * DTW + tcol) * CC);
            float v[V];
            rv.get(v);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
              for (int o = 0; o < 4; ++o) {
                const bool hit = (S == 1) ? ((o - kx + 2) == s)
                                          : ((((o + PL - kx) % 2) == 0) && (((o + PL - kx + 2) / 2) == s));
                if (hit) {
#pragma unroll
                  for (int j = 0; j < V; ++j) {
                    da[o][j] += v[j] * w3[kx][j];
                    dwacc[(dt * 3 + ky) * 3 + kx][j] += v[j] * act[o][j];
                  }
                }
              }
          }
        }
      }
      if (cvalid && iy < a.IH) {
        T* grow = g + (((long)(n * a.T + it) * a.IH + iy) * a.IW) * C + c0;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          if (ixb + o < a.IW) {
            float gv[V];
#pragma unroll
            for (int j = 0; j < V; ++j) {
              gv[j] = da[o][j] * silu_gradf_(xv[o][j] * sc[j] + sh[j]);
              st[0][j] += gv[j];
              st[1][j] += gv[j] * ((xv[o][j] - mu[j]) * rs[j]);
            }
            stv<T, V>(grow + (long)(ixb + o) * C, gv);
          }
        }
      }
    }
  };

  auto load_x = [&](int it, Raw<T, V> (&xraw)[SPT][4]) {
    const T* plane = x + ((long)(n * a.T + it) * a.IH * a.IW) * C + (cvalid ? c0 : 0);
#pragma unroll
    for (int sp = 0; sp < SPT; ++sp) {
      const int sid = pt + NPT * sp, r = sid / 4, sx = sid % 4;
#pragma unroll
      for (int o = 0; o < 4; ++o)
        xraw[sp][o].ld(plane + ((long)clampi(iy0 + r, 0, a.IH - 1) * a.IW + clampi(ix0 + 4 * sx + o, 0, a.IW - 1)) * C);
    }
  };

  for (int tt = 0; tt < tpb; ++tt) {
    ix0 = (blockIdx.x * tpb + tt) * TIW;
    if (ix0 >= a.IW) break;
    ox_lo = ix0 / S - 1;
    if (KT == 3) {
      load_slice(0, 0);
      for (int it = 0; it < a.T; ++it) {
        Raw<T, V> xraw[SPT][4];
        load_x(it, xraw);
        if (it + 1 < a.T) load_slice(it + 1, (it + 1) % 3);
        __syncthreads();
        compute(it, xraw);
        __syncthreads();
      }
    } else {
      for (int it = 0; it < a.T; ++it) {
        Raw<T, V> xraw[SPT][4];
        load_x(it, xraw);
        load_slice(it, 0);
        __syncthreads();
        compute(it, xraw);
        __syncthreads();
      }
    }
  }
  // reductions: same-chunk lanes of the wave by shuffles, then LDS, then coalesced global atomics
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float s = sum_same_chunk<NCH>(dwacc[t][j]);
      if ((tid & 63) < NCH) atomicAdd(&dwl[(ch * V + j) * NTAP + t], s);
    }
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float s = sum_same_chunk<NCH>(st[k][j]);
      if ((tid & 63) < NCH) atomicAdd(&st_l[k * CC + ch * V + j], (double)s);
    }
  __syncthreads();
  for (int e = tid; e < NTAP * CC; e += 256) {
    if (cbeg + e / NTAP < C) atomicAdd(a.dw + (long)cbeg * NTAP + e, dwl[e]);
  }
  if (tid < 2 * CC) {
    const int k = tid / CC, c = tid - k * CC;
    if (cbeg + c < C) {
      const int slot = (blockIdx.x + blockIdx.y * gridDim.x + n * 5) % MDS_STAT_SLOTS;
      atomicAdd(a.stats + ((long)slot * 2 + k) * C + cbeg + c, st_l[tid]);
    }
  }
}

extern "C" int mds_dw_bwd(const mds_dw_bwd_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->N > 0 && a->T > 0 && a->C % 8 == 0, "dw_bwd: bad dims");
  MDS_REQUIRE(a->kt == 1 || a->kt == 3, "dw_bwd: kt must be 1 or 3");
  MDS_REQUIRE(a->stride == 1 || a->stride == 2, "dw_bwd: stride");
  MDS_REQUIRE(!(a->kt == 3 && a->stride == 2), "dw_bwd: 3x3x3 is stride 1 only");
  MDS_REQUIRE(a->x && a->dy && a->w && a->g && a->dw && a->stats && a->mean && a->rstd, "dw_bwd: null pointer");
  MDS_REQUIRE(a->pro.mode == MDS_PRO_BN_SILU && a->pro.scale && a->pro.shift, "dw_bwd: needs the BN+SiLU prologue of the forward");
  MDS_REQUIRE(a->stride == 2 ? (a->pad_l == 0 || a->pad_l == 1) : (a->pad_l == 1 && a->pad_t == 1), "dw_bwd: pad=(%d,%d) unsupported for stride %d", a->pad_t, a->pad_l, a->stride);
  MDS_REQUIRE((long)a->N * 64 < 65536, "dw_bwd: grid.z");
  if (a->kt == 1 && a->stride == 1) {
    MDS_REQUIRE(a->OH == a->IH && a->OW == a->IW, "dw_bwd: stride-1 geometry");
    DwStrips g = dw_strips(a->N * a->T, a->IH, a->IW, a->C, 4, dw2_len(a->N * a->T, a->IH, a->IW, a->C, 4));
    dim3 grid = dw_grid(g), block(256);
    MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH((dw2_bwd_kernel<T, 4>), grid, block, 0, stream, *a, g));
    return mds_check_launch("dw_bwd");
  }
  if (a->kt == 1 && a->stride == 2) {
    MDS_REQUIRE((a->pad_t == 0 || a->pad_t == 1), "dw_bwd: pad_t");
    DwStrips g = dw_strips(a->N * a->T, a->IH, a->IW, a->C, 4, 16, true);
    dim3 grid = dw_grid(g), block(256);
    MDS_DISPATCH_DTYPE(a->dtype, T, {
      if (a->pad_t == 0 && a->pad_l == 0) MDS_LAUNCH((dw2s_bwd_kernel<T, 0, 0>), grid, block, 0, stream, *a, g);
      else if (a->pad_t == 0) MDS_LAUNCH((dw2s_bwd_kernel<T, 0, 1>), grid, block, 0, stream, *a, g);
      else if (a->pad_l == 0) MDS_LAUNCH((dw2s_bwd_kernel<T, 1, 0>), grid, block, 0, stream, *a, g);
      else MDS_LAUNCH((dw2s_bwd_kernel<T, 1, 1>), grid, block, 0, stream, *a, g);
    });
    return mds_check_launch("dw_bwd");
  }
  if (a->kt == 3 && a->T == DW3_T) {
    DwStrips g = dw_strips(a->N, a->IH, a->IW, a->C, 1, dw3_len(a->N, a->IH, a->IW, a->C, 16));
    dim3 grid = dw_grid(g), block(256);
    MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH(dw3_bwd_kernel<T>, grid, block, 0, stream, *a, g));
    return mds_check_launch("dw_bwd");
  }
  MDS_DISPATCH_DTYPE(a->dtype, T, {
    const int CC = DwCfg<T>::CC;
    const int nchunks = cdiv(a->C, CC);
    const int tiles_x = cdiv(a->IW, 16);
    // the per-block fixed cost (tap staging, 36-value wave reductions, atomics) is amortised over a
    // whole row band in 2D (measured: 342 -> 215 us at 46x80x672); 3D blocks already walk T slices
    int tpb = a->kt == 3 ? 2 : (tiles_x < 8 ? tiles_x : 8);
    dim3 grid(cdiv(tiles_x, tpb), cdiv(a->IH, 8), a->N * nchunks), block(256);
    const int dnpix = a->stride == 1 ? 10 * 18 : 6 * 10;
    const size_t smem = (size_t)a->kt * dnpix * CC * sizeof(T) + ((size_t)2 * a->kt * 9 * CC + 4 * CC) * sizeof(float);
    if (a->kt == 3) MDS_LAUNCH((dw_bwd_kernel<T, 1, 1, 3>), grid, block, smem, stream, *a, nchunks, tpb);
    else if (a->stride == 1) MDS_LAUNCH((dw_bwd_kernel<T, 1, 1, 1>), grid, block, smem, stream, *a, nchunks, tpb);
    else if (a->pad_l == 0) MDS_LAUNCH((dw_bwd_kernel<T, 2, 0, 1>), grid, block, smem, stream, *a, nchunks, tpb);
    else MDS_LAUNCH((dw_bwd_kernel<T, 2, 1, 1>), grid, block, smem, stream, *a, nchunks, tpb);
  });
  return mds_check_launch("dw_bwd");
}
