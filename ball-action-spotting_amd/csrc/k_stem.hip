// k_stem.hip — the stem: 3x3 stride-2 TF-SAME convolution of the fp32 frame triple.
//
// Input is the (b*S, 3, H, W) fp32 view of the frame stack (multidim_stacker.py:214): three
// grayscale planes per image.  K = 27 (padded to 32) is a single MFMA k-step, so the kernel is a
// pure streaming pass: 226 MB of frames in, 301 MB of bf16 activations out per batch-4 step
// (15 FLOP/B -> HBM-bound).  im2col fragments are gathered straight from global memory (the
// planes are read with unit stride along W by neighbouring lanes), weights live in registers.
#include "gemm.h"

template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(mds_stem_fwd_args a) {
  typedef typename Frag<T>::type frag_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int i = lane & 15, q = lane >> 4;
  const int gw = blockIdx.x * 4 + (tid >> 6), nw = gridDim.x * 4;
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
      if (tp[j] >= 0 && ox < a.OW && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
        xv[j] = a.x[(((long)n * 3 + tp[j]) * a.H + iy) * a.W + ix];
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
          store4(y + row * a.Cout + oc, v);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) { s_[f * 4 + rr] += v[rr]; ss_[f * 4 + rr] += v[rr] * v[rr]; }
        }
      }
    }
  }
  if (a.stats) {
    float* st = a.stats + (long)(blockIdx.x % MDS_STAT_SLOTS) * 2 * a.Cout;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = sum_over_i16(s_[e]), ss = sum_over_i16(ss_[e]);
      const int oc = 16 * (e >> 2) + 4 * q + (e & 3);
      if (i == 0 && oc < a.Cout) { atomicAdd(st + oc, s); atomicAdd(st + a.Cout + oc, ss); }
    }
  }
}

extern "C" int mds_stem_fwd(const mds_stem_fwd_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->N > 0 && a->H > 0 && a->W > 0 && a->OH > 0 && a->OW > 0, "stem_fwd: bad dims");
  MDS_REQUIRE(a->Cout % 16 == 0 && a->Cout <= 32, "stem_fwd: Cout=%d must be 16 or 32", a->Cout);
  MDS_REQUIRE(a->x && a->w && a->y, "stem_fwd: null pointer");
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

extern "C" int mds_stem_wgrad(const mds_stem_wgrad_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->N > 0 && a->H > 0 && a->W > 0 && a->OH > 0 && a->OW > 0, "stem_wgrad: bad dims");
  MDS_REQUIRE(a->Cout % 16 == 0 && a->Cout <= 32, "stem_wgrad: Cout=%d must be 16 or 32", a->Cout);
  MDS_REQUIRE(a->x && a->dy && a->dw, "stem_wgrad: null pointer");
  const long ngroups = (long)a->N * a->OH * ((a->OW + 31) / 32);
  long nb = (ngroups + 3) / 4;
  if (nb > 512) nb = 512;
  MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH(stem_wgrad_kernel<T>, dim3((unsigned)nb), dim3(256), 0, stream, *a));
  return mds_check_launch("stem_wgrad");
}
