// k_stem.hip — the stem: 3x3 stride-2 TF-SAME convolution of the fp32 frame triple.
//
// Input is the (b*S, 3, H, W) fp32 view of the frame stack (multidim_stacker.py:214): three
// grayscale planes per image.  K = 27 (padded to 32) is a single MFMA k-step, so the kernel is a
// pure streaming pass: 226 MB of frames in, 301 MB of bf16 activations out per batch-4 step
// (15 FLOP/B -> HBM-bound).  im2col fragments are gathered straight from global memory (the
// planes are read with unit stride along W by neighbouring lanes), weights live in registers.
#include <stdlib.h>
#include "gemm.h"

template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(mds_stem_fwd_args a) {
  MDS_CHAIN_PRIO();
  typedef typename Frag<T>::type frag_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int i = lane & 15, q = lane >> 4;
  // XCD-aware: consecutive block ids go to consecutive XCDs, each with its own L2.  A wave's 16-pixel group reads 33 input
  // columns of 3 rows: the boundary line of a block's four groups and two of the three rows are shared with the neighbouring
  // blocks - on eight different XCDs each of them fetched the shared lines from HBM again (PMC round 4: 416 MB fetched for a
  // 226 MB input).  Every XCD now walks a contiguous range of the group sequence.
  const int gw = xcd_contiguous(blockIdx.x, gridDim.x) * 4 + (tid >> 6), nw = gridDim.x * 4;
  const T* w = (const T*)a.w;
  frag_t wf[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    if (16 * f + i < a.Cout) wf[f] = ld_frag(w + (16 * f + i) * 32 + 8 * q);
    else frag_zero(wf[f]);
  }
  int tp[8], tky[8], tkx[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * q + j;
    tp[j] = k < 27 ? k / 9 : -1;
    tky[j] = (k % 9) / 3;
    tkx[j] = k % 3;
  }
  const int gpr = (a.OW + 15) / 16;  // 16-pixel groups per output row
  const long ngroups = (long)a.N * a.OH * gpr;
  float s_[8], ss_[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s_[e] = 0.f; ss_[e] = 0.f; }
  T* y = (T*)a.y;
  const int emode = a.epi.mode;       // eval-mode output transform (mds_epi_t): this lane's 2 x 4 output channels
  float es[2][4], eh[2][4];
  if (emode != MDS_EPI_NONE) {
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int oc = 16 * f + 4 * q + rr;
        es[f][rr] = oc < a.Cout ? a.epi.scale[oc] : 0.f;
        eh[f][rr] = oc < a.Cout ? a.epi.shift[oc] : 0.f;
      }
  }
  for (long gi = gw; gi < ngroups; gi += nw) {
    const int gx = (int)(gi % gpr);
    long r = gi / gpr;
    const int oy = (int)(r % a.OH);
    const int n = (int)(r / a.OH);
    const int ox = gx * 16 + i;
    frag_t xf;
    float xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int iy = oy * 2 + tky[j] - a.pad_t, ix = ox * 2 + tkx[j] - a.pad_l;
      xv[j] = 0.f;
      if (tp[j] >= 0 && ox < a.OW && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
        if (a.ingest.u8) {   // raw uint8 frame: pad + /255 (+ mirrored re-read for the TTA copies) while gathering
          const int ns = n < a.ingest.nsrc ? n : n - a.ingest.nsrc;
          const int sy = iy - a.ingest.pad_top, sx = (n < a.ingest.nsrc ? ix : a.W - 1 - ix) - a.ingest.pad_left;
          if (sy >= 0 && sy < a.ingest.src_h && sx >= 0 && sx < a.ingest.src_w)
            xv[j] = (float)a.ingest.u8[(((long)ns * 3 + tp[j]) * a.ingest.src_h + sy) * a.ingest.src_w + sx] * a.ingest.scale;
        } else {
          xv[j] = a.x[(((long)n * 3 + tp[j]) * a.H + iy) * a.W + ix];
        }
      }
    }
    frag_from8(xf, xv);
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    mma16(wf[0], xf, acc[0]);  // acc[r] = y[pixel i][oc = 4q + r]
    mma16(wf[1], xf, acc[1]);
    if (ox < a.OW) {
      const long row = ((long)n * a.OH + oy) * a.OW + ox;
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int oc = 16 * f + 4 * q;
        if (oc < a.Cout) {
          float v[4] = {acc[f][0], acc[f][1], acc[f][2], acc[f][3]};
          if (emode != MDS_EPI_NONE) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const float z = v[rr] * es[f][rr] + eh[f][rr];
              v[rr] = emode == MDS_EPI_BN_SILU ? siluf_(z) : z;
            }
          }
          store4(y + row * a.Cout + oc, v);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) { s_[f * 4 + rr] += v[rr]; ss_[f * 4 + rr] += v[rr] * v[rr]; }
        }
      }
    }
  }
  if (a.stats) {
    double* st = a.stats + (long)(blockIdx.x % MDS_STAT_SLOTS) * 2 * a.Cout;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = sum_over_i16(s_[e]), ss = sum_over_i16(ss_[e]);
      const int oc = 16 * (e >> 2) + 4 * q + (e & 3);
      if (i == 0 && oc < a.Cout) { atomicAdd(st + oc, (double)s); atomicAdd(st + a.Cout + oc, (double)ss); }
    }
  }
}

static int stem_fwd_tiled(const mds_stem_fwd_args* a, mds_stream_t stream);
extern "C" int mds_stem_fwd(const mds_stem_fwd_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->N > 0 && a->H > 0 && a->W > 0 && a->OH > 0 && a->OW > 0, "stem_fwd: bad dims");
  MDS_REQUIRE(a->Cout % 16 == 0 && a->Cout <= 32, "stem_fwd: Cout=%d must be 16 or 32", a->Cout);
  MDS_REQUIRE((a->x || a->ingest.u8) && a->w && a->y, "stem_fwd: null pointer");
  MDS_REQUIRE(!a->ingest.u8 || (a->ingest.nsrc > 0 && a->ingest.src_h > 0 && a->ingest.src_w > 0 && a->N <= 2 * a->ingest.nsrc),
              "stem_fwd: ingest needs nsrc, src_h, src_w and N <= 2 * nsrc");
  if (a->dtype == MDS_BF16 && a->x && !a->ingest.u8 && a->epi.mode == MDS_EPI_NONE && mds_knob(MDS_KNOB_STEM_FWD) != 1)
    return stem_fwd_tiled(a, stream);                  // the training forward (k_stem.hip, below)
  const long ngroups = (long)a->N * a->OH * ((a->OW + 15) / 16);
  long nb = (ngroups + 3) / 4;
  if (nb > 4096) nb = 4096;
  MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH(stem_fwd_kernel<T>, dim3((unsigned)nb), dim3(256), 0, stream, *a));
  return mds_check_launch("stem_fwd");
}

// weight gradient: dw[oc][27] += sum_pixels dy[pixel][oc] * x27[pixel][k]; the MFMA reduction
// index runs over 32 consecutive output pixels of one row.
template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(mds_stem_wgrad_args a) {
  typedef typename Frag<T>::type frag_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int i = lane & 15, q = lane >> 4;
  const int gw = blockIdx.x * 4 + (tid >> 6), nw = gridDim.x * 4;
  const T* dy = (const T*)a.dy;
  int tp[2], tky[2], tkx[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int k = 16 * g + i;
    tp[g] = k < 27 ? k / 9 : -1;
    tky[g] = (k % 9) / 3;
    tkx[g] = k % 3;
  }
  const int gpr = (a.OW + 31) / 32;
  const long ngroups = (long)a.N * a.OH * gpr;
  f32x4 acc[2][2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int g = 0; g < 2; ++g) acc[f][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (long gi = gw; gi < ngroups; gi += nw) {
    const int gx = (int)(gi % gpr);
    long r = gi / gpr;
    const int oy = (int)(r % a.OH);
    const int n = (int)(r / a.OH);
    const int oxb = gx * 32 + 8 * q;  // this lane's 8 pixels
    const long rowb = ((long)n * a.OH + oy) * a.OW;
    frag_t yf[2], xf[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      float yv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ox = oxb + j, oc = 16 * f + i;
        yv[j] = (ox < a.OW && oc < a.Cout) ? Elem<T>::ld(dy + (rowb + ox) * a.Cout + oc) : 0.f;
      }
      frag_from8(yf[f], yv);
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int iy = oy * 2 + tky[g] - a.pad_t;
      float xv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ox = oxb + j, ix = ox * 2 + tkx[g] - a.pad_l;
        xv[j] = 0.f;
        if (tp[g] >= 0 && ox < a.OW && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
          xv[j] = a.x[(((long)n * 3 + tp[g]) * a.H + iy) * a.W + ix];
      }
      frag_from8(xf[g], xv);
    }
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int g = 0; g < 2; ++g) mma16(yf[f], xf[g], acc[f][g]);  // acc[r] = dw[oc = 4q + r][k = i]
  }
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int k = 16 * g + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int oc = 16 * f + 4 * q + r;
        if (k < 27 && oc < a.Cout) atomicAdd(a.dw + oc * 27 + k, acc[f][g][r]);
      }
    }
}

// bf16 variant, tiled through LDS (the gather kernel above issues 32 scalar loads per MFMA group and
// runs at 0.9 TB/s).  Tile = 8 output rows x 32 output columns; per (plane, input row) three arrays
// A_kx[j] = x[2(ox0+j) + kx - pad_l] are staged (9 coalesced 8-byte loads -> three 16-byte LDS
// stores per item), so a tap's fragment — 8 consecutive output columns of one (plane, ky, kx) — is ONE
// aligned 16-byte LDS read; dy is staged as loaded ([pixel][oc]) and read with the transposing
// ds_read_b64_tr_b16.  Persistent blocks, next tile's loads in flight under the MFMAs.
#define SW_ROWS 8
#define SW_COLS 32
#define SW_PITCH 48   // elements: 96 B = 32 B x odd (conflict-free for ds_read_b128 and the tr reads)
template <bool DYP>
__global__ __launch_bounds__(256, DYP ? 2 : 3) void stem_wgrad_tiled_kernel(mds_stem_wgrad_args a, int tiles_a, int tiles_b, int tiles_per_block) {
  typedef bf16_t T;
  constexpr int IR = 2 * SW_ROWS + 1;                  // input rows of a tile
  constexpr int NX = IR * 3 * (SW_COLS / 8);           // x staging items: (input row, plane, 8-column group)
  MDS_DYN_SMEM(smem);
  T* xa = (T*)smem;                                    // [IR][3 planes][3 kx][SW_PITCH]
  T* dys = xa + IR * 9 * SW_PITCH;                     // [SW_ROWS * SW_COLS][SW_PITCH]
  T* zrow = dys + SW_ROWS * SW_COLS * SW_PITCH;        // [SW_PITCH] zeros (k = 27..31)
  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const T* dy = (const T*)(DYP ? a.dyp.g.u : a.dy);
  const long ydiff = DYP ? (const T*)a.dyp.y - dy : 0;   // the BatchNorm input y has u's layout
  // dy = A*g + B*y + D, g = u or u*silu'(y*scale + shift): this thread stages the same 8 channels of every pixel
  float cA[8], cB[8], cD[8], csc[8], csh[8];
  if (DYP) {
    const int c = 8 * (tid & 3), C = a.Cout;
    if (c < C) {
      load8f(a.dyp.lin + c, cA); load8f(a.dyp.lin + C + c, cB); load8f(a.dyp.lin + 2 * C + c, cD);
      load8f(a.dyp.bn + c, csc); load8f(a.dyp.bn + C + c, csh);
    }
  }
  const bool gsilu = DYP && a.dyp.g.mode == MDS_G_SILU;
  if (tid < SW_PITCH) zrow[tid] = 0;
  // this lane's two taps n = 16g + i -> (plane, ky, kx): LDS element offset of its row within the tile image
  int noff[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int n = 16 * g + i;
    const int pl = n / 9, ky = (n % 9) / 3, kx = n % 3;
    noff[g] = n < 27 ? ((ky * 3 + pl) * 3 + kx) * SW_PITCH : -1;   // + (2 r) * 9 * SW_PITCH per output row r
  }
  const long total = (long)a.N * tiles_a * tiles_b;
  long tl = (long)xcd_contiguous(blockIdx.x, gridDim.x) * tiles_per_block, tl_end = tl + tiles_per_block;
  if (tl_end > total) tl_end = total;
  f32x4 acc[2][2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int g = 0; g < 2; ++g) acc[f][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 rx[9];
  RawV8<T> ry[4], ryy[DYP ? 4 : 1];
  unsigned yok = 0;
  auto origin = [&](long t, int& img, int& oy0, int& ox0) {
    img = (int)(t / (tiles_a * tiles_b));
    const int rem = (int)(t - (long)img * tiles_a * tiles_b);
    oy0 = (rem / tiles_b) * SW_ROWS; ox0 = (rem % tiles_b) * SW_COLS;
  };
  auto issue = [&](long t) {
    int img, oy0, ox0;
    origin(t, img, oy0, ox0);
    if (tid < NX) {
      const int grp = tid & 3, pl = (tid >> 2) % 3, ir = tid / 12;
      const int iy = oy0 * 2 - a.pad_t + ir;
      const int iyc = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
      const float* row = a.x + (((long)img * 3 + pl) * a.H + iyc) * a.W;
      const int ix0 = 2 * (ox0 + 8 * grp) - a.pad_l;
      if (iy >= 0 && iy < a.H && ix0 >= 0 && ix0 + 17 < a.W) {   // interior: nine 8-byte loads
#pragma unroll
        for (int j = 0; j < 9; ++j) rx[j] = *(const f32x2*)(row + ix0 + 2 * j);
      } else {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
          const int ix = ix0 + 2 * j;
          const bool ok0 = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W, ok1 = iy >= 0 && iy < a.H && ix + 1 >= 0 && ix + 1 < a.W;
          const int c0 = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix), c1 = ix + 1 < 0 ? 0 : (ix + 1 >= a.W ? a.W - 1 : ix + 1);
          const float v0 = row[c0], v1 = row[c1];
          rx[j] = (f32x2){ok0 ? v0 : 0.f, ok1 ? v1 : 0.f};
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int it = tid + 256 * p, px = it >> 2, ch = it & 3;
      const int oy = oy0 + (px >> 5), ox = ox0 + (px & 31);
      const bool ok = oy < a.OH && ox < a.OW && 8 * ch < a.Cout;
      if (ok) ry[p].ld(dy + (((long)img * a.OH + oy) * a.OW + ox) * a.Cout + 8 * ch);
      else ry[p].zero();
      if (DYP) {
        if (p == 0) yok = 0;
        yok |= (ok ? 1u : 0u) << p;
        if (ok) ryy[p].ld(dy + ydiff + (((long)img * a.OH + oy) * a.OW + ox) * a.Cout + 8 * ch);
        else ryy[p].zero();
      }
    }
  };
  if (tl < tl_end) issue(tl);
  for (; tl < tl_end; ++tl) {
    __syncthreads();
    if (tid < NX) {
      const int grp = tid & 3, pl = (tid >> 2) % 3, ir = tid / 12;
      T* base = xa + ((ir * 3 + pl) * 3) * SW_PITCH + 8 * grp;
      float v0[8], v1[8], v2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { v0[j] = rx[j][0]; v1[j] = rx[j][1]; v2[j] = rx[j + 1][0]; }
      store8(base, v0); store8(base + SW_PITCH, v1); store8(base + 2 * SW_PITCH, v2);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int it = tid + 256 * p;
      if (!DYP) {
        ry[p].st(dys + (it >> 2) * SW_PITCH + 8 * (it & 3));
      } else {
        float u[8], yv[8], v[8];
        ry[p].get(u);
        ryy[p].get(yv);
        const bool ok = (yok >> p) & 1u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float g = u[j];
          if (gsilu) g *= silu_gradf_(yv[j] * csc[j] + csh[j]);
          v[j] = ok ? cA[j] * g + cB[j] * yv[j] + cD[j] : 0.f;   // pixels past the image contribute nothing (D != 0)
        }
        store8(dys + (it >> 2) * SW_PITCH + 8 * (it & 3), v);
      }
    }
    __syncthreads();
    if (tl + 1 < tl_end) issue(tl + 1);
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = 2 * wave + rr;  // output row of the tile handled by this wave
      // A: dy^T fragments (8 pixels of one oc): pixels r*32 + 16(q>>1) + 4(q&1) + {0..3, 8..11}
      const int p0 = r * SW_COLS + 16 * (q >> 1) + 4 * (q & 1) + (i >> 2);
      u16x8 yf[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const T* pc = dys + 16 * f + 4 * (i & 3);
        const u16x4 lo = lds_tr4(pc + p0 * SW_PITCH), hi = lds_tr4(pc + (p0 + 8) * SW_PITCH);
        yf[f] = (u16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
      // B: x fragments: the same pixel permutation -> columns 16(q>>1) + 4(q&1) + {0..3} and +8
      const int cb = 16 * (q >> 1) + 4 * (q & 1);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const T* rowp = noff[g] >= 0 ? xa + (2 * r) * 9 * SW_PITCH + noff[g] : zrow;
        const int c = noff[g] >= 0 ? cb : 0;
        const u16x4 lo = *(const u16x4*)(rowp + c), hi = *(const u16x4*)(rowp + c + (noff[g] >= 0 ? 8 : 4));
        const u16x8 xf = (u16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int f = 0; f < 2; ++f) mma16(yf[f], xf, acc[f][g]);  // acc[r] = dw[oc = 16f + 4q + r][k = 16g + i]
      }
    }
  }
  // the 864 filter gradients are shared by every block: sum the four waves in LDS first, then one
  // atomic per value and block (per-lane atomics from 1024 blocks = 4800 contended adds per address,
  // which was most of this kernel's time)
  __syncthreads();
  float* red = (float*)smem;   // [4 waves][32 oc][32 k]
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 32 + 16 * f + 4 * q + r) * 32 + 16 * g + i] = acc[f][g][r];
  __syncthreads();
  for (int e = tid; e < 32 * 32; e += 256) {
    const int oc = e >> 5, k = e & 31;
    if (k < 27 && oc < a.Cout) atomicAdd(a.dw + oc * 27 + k, (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]));
  }
}

// bf16 training forward, tiled through LDS (round 5).  The gather kernel at the top of this file issues 8 scalar 4-byte loads per
// lane and 16-pixel group - 64 lanes x every second column of 4 (row, plane) pairs: ~20 cache lines per instruction, bound by the
// texture-address path at 2.8 TB/s.  Here a block stages the fp32 input of an 8 x 32-pixel output tile ONCE with coalesced 8-byte
// loads, as the three pre-shifted bf16 arrays A_kx[j] = x[2 (ox0 + j) + kx - pad_l] per (input row, plane) that the tiled weight
// gradient uses, and a lane gathers its 8 taps of one pixel from LDS (2-byte reads; consecutive lanes = consecutive columns).
// Persistent blocks, the next tile's loads in flight under this tile's MFMAs and stores.  Same arithmetic as the gather kernel:
// the input is rounded to bf16 once (RNE), products on the bf16 matrix core, fp32 accumulation, statistics of the fp32 values.
__global__ __launch_bounds__(256, 3) void stem_fwd_tiled_kernel(mds_stem_fwd_args a, int tiles_a, int tiles_b, int tiles_per_block) {
  MDS_CHAIN_PRIO();
  typedef bf16_t T;
  constexpr int IR = 2 * SW_ROWS + 1;                  // input rows of a tile
  constexpr int NX = IR * 3 * (SW_COLS / 8);           // x staging items: (input row, plane, 8-column group)
  __shared__ __attribute__((aligned(16))) T xa[IR * 9 * SW_PITCH + SW_PITCH];   // [IR][3 planes][3 kx][SW_PITCH] + a row of zeros
  T* zrow = xa + IR * 9 * SW_PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const T* w = (const T*)a.w;
  u16x8 wf[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    if (16 * f + i < a.Cout) wf[f] = ld_frag(w + (16 * f + i) * 32 + 8 * q);
    else frag_zero(wf[f]);
  }
  if (tid < SW_PITCH) zrow[tid] = 0;
  // this lane's eight taps k = 8q + j -> (plane, ky, kx): LDS element offset of the tap's array within the tile image (row 2r of
  // the tile is added per output row r); k >= 27: the row of zeros
  int toff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * q + j, pl = k / 9, ky = (k % 9) / 3, kx = k % 3;
    toff[j] = k < 27 ? ((ky * 3 + pl) * 3 + kx) * SW_PITCH : -1;
  }
  const long total = (long)a.N * tiles_a * tiles_b;
  long tl = (long)xcd_contiguous(blockIdx.x, gridDim.x) * tiles_per_block, tl_end = tl + tiles_per_block;
  if (tl_end > total) tl_end = total;
  float s_[8], ss_[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s_[e] = 0.f; ss_[e] = 0.f; }
  T* y = (T*)a.y;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 rx[9];
  auto origin = [&](long t, int& img, int& oy0, int& ox0) {
    img = (int)(t / (tiles_a * tiles_b));
    const int rem = (int)(t - (long)img * tiles_a * tiles_b);
    oy0 = (rem / tiles_b) * SW_ROWS; ox0 = (rem % tiles_b) * SW_COLS;
  };
  auto issue = [&](long t) {
    int img, oy0, ox0;
    origin(t, img, oy0, ox0);
    if (tid < NX) {
      const int grp = tid & 3, pl = (tid >> 2) % 3, ir = tid / 12;
      const int iy = oy0 * 2 - a.pad_t + ir;
      const int iyc = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
      const float* row = a.x + (((long)img * 3 + pl) * a.H + iyc) * a.W;
      const int ix0 = 2 * (ox0 + 8 * grp) - a.pad_l;
      if (iy >= 0 && iy < a.H && ix0 >= 0 && ix0 + 17 < a.W) {   // interior: nine 8-byte loads
#pragma unroll
        for (int j = 0; j < 9; ++j) rx[j] = *(const f32x2*)(row + ix0 + 2 * j);
      } else {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
          const int ix = ix0 + 2 * j;
          const bool ok0 = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W, ok1 = iy >= 0 && iy < a.H && ix + 1 >= 0 && ix + 1 < a.W;
          const int c0 = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix), c1 = ix + 1 < 0 ? 0 : (ix + 1 >= a.W ? a.W - 1 : ix + 1);
          const float v0 = row[c0], v1 = row[c1];
          rx[j] = (f32x2){ok0 ? v0 : 0.f, ok1 ? v1 : 0.f};
        }
      }
    }
  };
  if (tl < tl_end) issue(tl);
  for (; tl < tl_end; ++tl) {
    __syncthreads();                                   // the previous tile's fragments have been read
    if (tid < NX) {
      const int grp = tid & 3, pl = (tid >> 2) % 3, ir = tid / 12;
      T* base = xa + ((ir * 3 + pl) * 3) * SW_PITCH + 8 * grp;
      float v0[8], v1[8], v2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { v0[j] = rx[j][0]; v1[j] = rx[j][1]; v2[j] = rx[j + 1][0]; }
      store8(base, v0); store8(base + SW_PITCH, v1); store8(base + 2 * SW_PITCH, v2);
    }
    __syncthreads();
    if (tl + 1 < tl_end) issue(tl + 1);
    int img, oy0, ox0;
    origin(tl, img, oy0, ox0);
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = 2 * wave + rr, oy = oy0 + r;       // output row of the tile handled by this wave
#pragma unroll
      for (int h = 0; h < 2; ++h) {                    // its two 16-pixel groups
        const int col = 16 * h + i, ox = ox0 + col;
        const T* rowp = xa + (2 * r) * 9 * SW_PITCH + col;
        u16x8 xf;
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[j] = toff[j] >= 0 ? rowp[toff[j]] : zrow[0];
        f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        mma16(wf[0], xf, acc[0]);                      // acc[r] = y[pixel i][oc = 4q + r]
        mma16(wf[1], xf, acc[1]);
        if (oy < a.OH && ox < a.OW) {
          const long row = ((long)img * a.OH + oy) * a.OW + ox;
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const int oc = 16 * f + 4 * q;
            if (oc < a.Cout) {
              float v[4] = {acc[f][0], acc[f][1], acc[f][2], acc[f][3]};
              store4(y + row * a.Cout + oc, v);
#pragma unroll
              for (int e = 0; e < 4; ++e) { s_[f * 4 + e] += v[e]; ss_[f * 4 + e] += v[e] * v[e]; }
            }
          }
        }
      }
    }
  }
  if (a.stats) {
    double* st = a.stats + (long)(blockIdx.x % MDS_STAT_SLOTS) * 2 * a.Cout;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = sum_over_i16(s_[e]), ss = sum_over_i16(ss_[e]);
      const int oc = 16 * (e >> 2) + 4 * q + (e & 3);
      if (i == 0 && oc < a.Cout) { atomicAdd(st + oc, (double)s); atomicAdd(st + a.Cout + oc, (double)ss); }
    }
  }
}

static int stem_fwd_tiled(const mds_stem_fwd_args* a, mds_stream_t stream) {
  const int tiles_a = cdiv(a->OH, SW_ROWS), tiles_b = cdiv(a->OW, SW_COLS);
  const long total = (long)a->N * tiles_a * tiles_b;
  const int tpb = (int)cdiv(total, total < 768 ? total : 768);   // three blocks per CU
  MDS_LAUNCH(stem_fwd_tiled_kernel, dim3(cdiv(total, tpb)), dim3(256), 0, stream, *a, tiles_a, tiles_b, tpb);
  return mds_check_launch("stem_fwd");
}

extern "C" int mds_stem_wgrad(const mds_stem_wgrad_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->N > 0 && a->H > 0 && a->W > 0 && a->OH > 0 && a->OW > 0, "stem_wgrad: bad dims");
  MDS_REQUIRE(a->Cout % 16 == 0 && a->Cout <= 32, "stem_wgrad: Cout=%d must be 16 or 32", a->Cout);
  const bool dyp = a->dyp.mode != 0;
  MDS_REQUIRE(a->x && (a->dy || dyp) && a->dw, "stem_wgrad: null pointer");
  if (dyp) {
    MDS_REQUIRE(a->dtype == MDS_BF16, "stem_wgrad: the dy prologue is a bf16 feature");
    MDS_REQUIRE(a->dyp.g.u && a->dyp.y && a->dyp.bn && a->dyp.lin && (a->dyp.g.mode == MDS_G_PLAIN || a->dyp.g.mode == MDS_G_SILU),
                "stem_wgrad: dy prologue needs u, y, bn, lin and a PLAIN or SILU gradient source");
  }
  if (a->dtype == MDS_BF16) {
    const int tiles_a = cdiv(a->OH, SW_ROWS), tiles_b = cdiv(a->OW, SW_COLS);
    const long total = (long)a->N * tiles_a * tiles_b;
    const int tpb = (int)cdiv(total, total < 768 ? total : 768);   // three blocks per CU
    const size_t smem = (size_t)((2 * SW_ROWS + 1) * 9 + SW_ROWS * SW_COLS + 1) * SW_PITCH * sizeof(bf16_t);
    if (dyp) MDS_LAUNCH(stem_wgrad_tiled_kernel<true>, dim3(cdiv(total, tpb)), dim3(256), smem, stream, *a, tiles_a, tiles_b, tpb);
    else MDS_LAUNCH(stem_wgrad_tiled_kernel<false>, dim3(cdiv(total, tpb)), dim3(256), smem, stream, *a, tiles_a, tiles_b, tpb);
    return mds_check_launch("stem_wgrad");
  }
  const long ngroups = (long)a->N * a->OH * ((a->OW + 31) / 32);
  long nb = (ngroups + 3) / 4;
  if (nb > 512) nb = 512;
  MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH(stem_wgrad_kernel<T>, dim3((unsigned)nb), dim3(256), 0, stream, *a));
  return mds_check_launch("stem_wgrad");
}
