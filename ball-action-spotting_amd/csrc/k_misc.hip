// k_misc.hip — error plumbing and the latency-class kernels (tiny per-channel / per-group math).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include "elem.h"

static thread_local char g_err[512] = "";

void mds_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int mds_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    mds_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return MDS_ERR_LAUNCH;
  }
  return 0;
}
#include <atomic>
static std::atomic<int> g_knob[MDS_KNOB_COUNT];
int mds_knob(int id) { return id >= 0 && id < MDS_KNOB_COUNT ? g_knob[id].load(std::memory_order_relaxed) : 0; }
thread_local void* mds_tl_stop_event = nullptr;   // read by MDS_LAUNCH (mds_platform_hw.h)
thread_local int mds_tl_stop_uses = 0;
// -> the number of launches that were issued with the PREVIOUSLY armed event (0: the armed op launched nothing, the event still
// names an older kernel and must not be waited on)
extern "C" int mds_launch_event(void* event) {
  const int used = mds_tl_stop_uses;
  mds_tl_stop_event = event;
  mds_tl_stop_uses = 0;
  return used;
}
extern "C" int mds_dev_set(int knob, int value) {
  MDS_REQUIRE(knob >= 0 && knob < MDS_KNOB_COUNT, "mds_dev_set: unknown knob %d", knob);
  g_knob[knob].store(value, std::memory_order_relaxed);
  return 0;
}
extern "C" int mds_version(void) { return MDS_VERSION; }
extern "C" const char* mds_last_error(void) { return g_err; }

// ------------------------------------------------------------------ parameter packing
template <typename T>
__global__ void pack_kernel(const mds_pack_job* jobs, int njobs) {
  const mds_pack_job jb = jobs[blockIdx.y];
  const int O = jb.O, I = jb.I, taps = jb.taps;
  T* dst = (T*)jb.dst;
  if (jb.kind == MDS_PACK_STEM) {
    // src [O][27] -> dst [O][32]
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < O * 32; e += gridDim.x * blockDim.x) {
      int o = e >> 5, k = e & 31;
      Elem<T>::st(dst + e, k < 27 ? jb.src[o * 27 + k] : 0.0f);
    }
    return;
  }
  if (jb.kind == MDS_PACK_FRAG_OI || jb.kind == MDS_PACK_FRAG_IO) {
    // thread = one lane slot (16 bytes) of one fragment: (ks, nf, lane) -> w[n = 16 nf + (lane & 15)][k = 32 ks + 8 (lane >> 4) .. + 7]
    const bool io = jb.kind == MDS_PACK_FRAG_IO;
    const int N = io ? I : O, K = io ? O : I, NFT = (N + 15) >> 4, slots = ((K + 31) >> 5) * NFT * 64;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < slots; e += gridDim.x * blockDim.x) {
      const int lane = e & 63, f = e >> 6, nf = f % NFT, ks = f / NFT;
      const int n = 16 * nf + (lane & 15), k0 = 32 * ks + 8 * (lane >> 4);
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        v[j] = (n < N && k < K) ? (io ? jb.src[(size_t)k * I + n] : jb.src[(size_t)n * I + k]) : 0.0f;
      }
      store8(dst + (size_t)e * 8, v);
    }
    return;
  }
  const int total = O * I * taps;
  if (jb.kind == MDS_PACK_IO_F32 || (jb.kind == MDS_PACK_IO_FLIP && taps == 1)) {
    // [O][I] -> [I][O] (the data-gradient pack of every 1x1 filter, the squeeze-excite w2 copy): 32 x 32 tiles through LDS so
    // that both the fp32 rows read and the packed rows written are contiguous per wave half (a lane-per-destination gather read
    // one 64-byte line per lane)
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int tI = (I + 31) >> 5, nt = tI * ((O + 31) >> 5);
    const bool f32 = jb.kind == MDS_PACK_IO_F32;
    for (int t = blockIdx.x; t < nt; t += gridDim.x) {
      const int i0 = (t % tI) * 32, o0 = (t / tI) * 32;
      for (int r = ty; r < 32; r += 8) {
        const int o = o0 + r, i = i0 + tx;
        tile[r][tx] = (o < O && i < I) ? jb.src[(size_t)o * I + i] : 0.0f;
      }
      __syncthreads();
      for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, o = o0 + tx;
        if (i < I && o < O) {
          if (f32) ((float*)jb.dst)[(size_t)i * O + o] = tile[tx][r];
          else Elem<T>::st(dst + (size_t)i * O + o, tile[tx][r]);
        }
      }
      __syncthreads();
    }
    return;
  }
  if (jb.kind == MDS_PACK_OI && taps == 1 && (total & 3) == 0 && (((uintptr_t)jb.src | (uintptr_t)dst) & 15) == 0) {   // (a parameter that is a view at an odd offset takes the scalar path)
    // 1x1 filters (3/4 of the parameters): the packed order IS the parameter's order - a vectorised cast, no index arithmetic
    for (int e = (blockIdx.x * blockDim.x + threadIdx.x) * 4; e < total; e += gridDim.x * blockDim.x * 4) {
      float v[4];
      load4(jb.src + e, v);
      store4(dst + e, v);
    }
    return;
  }
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    // e indexes the destination
    if (jb.kind == MDS_PACK_OI) {
      int i = e % I, t = (e / I) % taps, o = e / (I * taps);
      Elem<T>::st(dst + e, jb.src[(o * I + i) * taps + t]);
    } else {  // MDS_PACK_IO_FLIP: dst [I][taps][O], tap flipped
      int o = e % O, t = (e / O) % taps, i = e / (O * taps);
      Elem<T>::st(dst + e, jb.src[(o * I + i) * taps + (taps - 1 - t)]);
    }
  }
}
extern "C" int mds_pack_weights(const mds_pack_job* jobs_dev, int njobs, int max_elems, int dtype,
                                mds_stream_t stream) {
  MDS_REQUIRE(jobs_dev && njobs > 0, "pack_weights: empty job table");
  int bx = cdiv(max_elems, 256 * 4);
  if (bx > 64) bx = 64;
  if (bx < 1) bx = 1;
  MDS_DISPATCH_DTYPE(dtype, T, MDS_LAUNCH(pack_kernel<T>, dim3(bx, njobs), dim3(256), 0, stream, jobs_dev, njobs));
  return mds_check_launch("pack_weights");
}

// ------------------------------------------------------------------ BN finalize (fwd)
// 72 + 72 of these run per step between dependent kernels, so their latency is on the critical
// path: a block handles 32 channels, 8 threads per channel each summing 4 of the 32 statistic slots
// (all 8 loads independent and coalesced along c), then one LDS hop.
#define FIN_CH 32
static_assert(MDS_STAT_SLOTS == 32, "finalize kernels sum 8 groups of 4 slots");
template <typename S>   // S = float (forward statistics) or double (backward statistics)
MDS_DEV void fin_slot_sums(const S* stats, int C, int c, int sg, double (&red)[2][8][FIN_CH], int cl) {
  double s = 0.0, ss = 0.0;
  if (c < C) {
    S v[4], w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = stats[((sg * 4 + k) * 2 + 0) * C + c];
      w[k] = stats[((sg * 4 + k) * 2 + 1) * C + c];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { s += v[k]; ss += w[k]; }
  }
  red[0][sg][cl] = s; red[1][sg][cl] = ss;
}
__global__ __launch_bounds__(256) void bn_finalize_kernel(mds_bn_finalize_args a) {
  MDS_CHAIN_PRIO();
  __shared__ double red[2][8][FIN_CH];
  const int cl = threadIdx.x % FIN_CH, sg = threadIdx.x / FIN_CH;
  const int c = blockIdx.x * FIN_CH + cl;
  const bool owner = sg == 0 && c < a.C;
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.training && a.num_batches_tracked) *a.num_batches_tracked += 1;
  // the owner's parameter loads are requested before the statistics arrive: one memory round trip
  float gam = 0.f, bet = 0.f, rm = 0.f, rv = 1.f;
  if (owner) {
    gam = a.gamma[c]; bet = a.beta[c];
    if (a.running_mean) { rm = a.running_mean[c]; rv = a.running_var[c]; }
  }
  if (a.training) fin_slot_sums(a.stats, a.C, c, sg, red, cl);
  __syncthreads();
  if (!owner) return;
  float mean, var;
  if (a.training) {
    double s = 0.0, ss = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { s += red[0][k][cl]; ss += red[1][k][cl]; }
    const double inv = 1.0 / (double)a.count;
    double m = s * inv;
    double v = ss * inv - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m;
    var = (float)v;
    if (a.running_mean) {
      float unb = a.count > 1 ? (float)(v * (double)a.count / (double)(a.count - 1)) : var;
      a.running_mean[c] = (1.0f - a.momentum) * rm + a.momentum * mean;
      a.running_var[c] = (1.0f - a.momentum) * rv + a.momentum * unb;
    }
  } else {
    mean = rm;
    var = rv;
  }
  float rstd = 1.0f / sqrtf(var + a.eps);
  float sc = gam * rstd;
  a.out[0 * a.C + c] = sc;
  a.out[1 * a.C + c] = bet - mean * sc;
  a.out[2 * a.C + c] = mean;
  a.out[3 * a.C + c] = rstd;
}
extern "C" int mds_bn_finalize(const mds_bn_finalize_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->C > 0 && a->out && a->gamma && a->beta, "bn_finalize: bad args");
  MDS_REQUIRE(a->training ? (a->stats != 0 && a->count > 0) : (a->running_mean && a->running_var),
              "bn_finalize: missing stats / running buffers");
  MDS_LAUNCH(bn_finalize_kernel, dim3(cdiv(a->C, FIN_CH)), dim3(256), 0, stream, *a);
  return mds_check_launch("bn_finalize");
}

// eval-mode BatchNorm of every layer of a plan in one launch (grid.y = layer)
__global__ __launch_bounds__(256) void bn_eval_table_kernel(const mds_bn_eval_job* jobs) {
  const mds_bn_eval_job jb = jobs[blockIdx.y];
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < jb.C; c += gridDim.x * blockDim.x) {
    const float mean = jb.running_mean[c], rstd = 1.0f / sqrtf(jb.running_var[c] + jb.eps);
    const float sc = jb.gamma[c] * rstd;
    jb.out[c] = sc;
    jb.out[jb.C + c] = jb.beta[c] - mean * sc;
    jb.out[2 * jb.C + c] = mean;
    jb.out[3 * jb.C + c] = rstd;
  }
}
extern "C" int mds_bn_eval_table(const mds_bn_eval_job* jobs_dev, int njobs, int max_c, mds_stream_t stream) {
  MDS_REQUIRE(jobs_dev && njobs > 0 && max_c > 0, "bn_eval_table: empty table");
  MDS_LAUNCH(bn_eval_table_kernel, dim3(cdiv(max_c, 256), njobs), dim3(256), 0, stream, jobs_dev);
  return mds_check_launch("bn_eval_table");
}

// ------------------------------------------------------------------ BN finalize (bwd)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(mds_bn_bwd_finalize_args a) {
  MDS_CHAIN_PRIO();
  __shared__ double red[2][8][FIN_CH];
  const int cl = threadIdx.x % FIN_CH, sg = threadIdx.x / FIN_CH;
  const int c = blockIdx.x * FIN_CH + cl;
  const bool owner = sg == 0 && c < a.C;
  // everything the channel's owner needs is requested up front: ONE memory round trip per launch
  float gam = 0.f, rstd = 0.f, mean = 0.f, dg = 0.f, db = 0.f;
  if (owner) {
    gam = a.gamma[c]; rstd = a.bn[3 * a.C + c]; mean = a.bn[2 * a.C + c];
    if (a.dgamma) dg = a.dgamma[c];
    if (a.dbeta) db = a.dbeta[c];
  }
  double fmean = 0.0;                                   // the batch mean in fp64 (coef64), from the forward pass's own sums
  if (a.coef64 && a.fwd_stats && a.batch_stats) {
    fin_slot_sums(a.fwd_stats, a.C, c, sg, red, cl);
    __syncthreads();
    if (owner) {
#pragma unroll
      for (int k = 0; k < 8; ++k) fmean += red[0][k][cl];
      fmean /= (double)a.count;
    }
    __syncthreads();
  } else if (owner) fmean = (double)mean;
  fin_slot_sums(a.stats, a.C, c, sg, red, cl);
  __syncthreads();
  if (!owner) return;
  double sg_ = 0.0, sgx = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) { sg_ += red[0][k][cl]; sgx += red[1][k][cl]; }
  const float inv = 1.0f / (float)a.count;
  if (a.dgamma) a.dgamma[c] = dg + (float)sgx;
  if (a.dbeta) a.dbeta[c] = db + (float)sg_;
  a.coef[0 * a.C + c] = gam * rstd;
  const float k0 = gam * rstd, k1 = a.batch_stats ? (float)sg_ * inv : 0.0f, k2 = a.batch_stats ? (float)sgx * inv : 0.0f;
  a.coef[1 * a.C + c] = k1;
  a.coef[2 * a.C + c] = k2;
  if (a.coef64) {
    const double invd = 1.0 / (double)a.count;
    a.coef64[0 * a.C + c] = a.batch_stats ? sg_ * invd : 0.0;
    a.coef64[1 * a.C + c] = a.batch_stats ? sgx * invd : 0.0;
    a.coef64[2 * a.C + c] = fmean;
  }
  if (a.lin) {   // dy = k0*(g - k1 - (y - mean)*rstd*k2) = A*g + B*y + D
    a.lin[0 * a.C + c] = k0;
    a.lin[1 * a.C + c] = -k0 * k2 * rstd;
    a.lin[2 * a.C + c] = k0 * (k2 * rstd * mean - k1);
  }
}
extern "C" int mds_bn_bwd_finalize(const mds_bn_bwd_finalize_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->C > 0 && a->stats && a->gamma && a->bn && a->coef && a->count > 0,
              "bn_bwd_finalize: bad args");
  MDS_LAUNCH(bn_bwd_finalize_kernel, dim3(cdiv(a->C, FIN_CH)), dim3(256), 0, stream, *a);
  return mds_check_launch("bn_bwd_finalize");
}

// ------------------------------------------------------------------ SE FCs
// Latency-class: groups <= 44, C <= 1152, R <= 64.  These used to be 2 + 4 dependent launches of
// loops over R or over the groups, 6-22 us each (60 us of pure latency per SE block and step).
// Now a block owns (group g, 256 channels); the R hidden values of its group are (re)computed by
// the block itself in LDS — R dot products of length C, a few thousand MACs — which removes the
// cross-block dependency, so the forward is ONE launch and the backward two.
#define SE_RMAX 64
#define SE_CCH 256

// out[r] = sum_c w[r][c] * v[c] for all r < R at once.  Wave k owns rows r = k, k + 4, ...; a lane owns 16-byte chunks of the
// channel axis.  Every load of a row batch (RH rows x JU chunks per lane) is issued before the first FMA, so a [48][1152]
// matrix is two memory round trips (it was one per 256 channels with 4-byte loads: five, plus an LDS hop across the waves).
// C % 4 == 0; rows past R read row R - 1 (a legal dummy), chunks past C are weighted by zero.
MDS_DEV f32x4 se_ld4(const float* p) { return *(const f32x4*)p; }
MDS_DEV f32x4 se_ld4(const double* p) {
  const f64x2 a = *(const f64x2*)p, b = *(const f64x2*)(p + 2);
  return (f32x4){(float)a[0], (float)a[1], (float)b[0], (float)b[1]};
}
template <int RB, typename V>  // RB = R rounded up to 16; V: float or double vector
MDS_DEV void se_matvec_rc(const float* w, const V* v, int R, int C, float* out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int RW = RB / 4, RH = RW > 6 ? RW / 2 : RW, JU = 5;
  const int C4 = C >> 2;
  float acc[RW];
#pragma unroll
  for (int k = 0; k < RW; ++k) acc[k] = 0.f;
  for (int fb = 0; fb < C4; fb += 64 * JU) {
    f32x4 vv[JU];
    int fo[JU];
#pragma unroll
    for (int j = 0; j < JU; ++j) {
      const int f = fb + lane + 64 * j;
      fo[j] = f < C4 ? 4 * f : 0;
      vv[j] = se_ld4(v + fo[j]);
      if (f >= C4) vv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int h = 0; h < RW; h += RH) {
      f32x4 wv[RH][JU];
#pragma unroll
      for (int rr = 0; rr < RH; ++rr) {
        const int r = wave + 4 * (h + rr);
        const float* wr = w + (long)(r < R ? r : R - 1) * C;
#pragma unroll
        for (int j = 0; j < JU; ++j) wv[rr][j] = se_ld4(wr + fo[j]);
      }
#pragma unroll
      for (int rr = 0; rr < RH; ++rr)
#pragma unroll
        for (int j = 0; j < JU; ++j) {
          const f32x4 p = wv[rr][j] * vv[j];
          acc[h + rr] += (p[0] + p[1]) + (p[2] + p[3]);
        }
    }
  }
#pragma unroll
  for (int k = 0; k < RW; ++k) {
    const float s = wave_sum(acc[k]);
    if (lane == 0) out[wave + 4 * k] = s;     // out: >= RB floats
  }
  __syncthreads();
}
#define SE_DISPATCH_RB(R, ...)                                      \
  do {                                                              \
    if ((R) <= 16) { constexpr int RB = 16; __VA_ARGS__; }          \
    else if ((R) <= 32) { constexpr int RB = 32; __VA_ARGS__; }     \
    else if ((R) <= 48) { constexpr int RB = 48; __VA_ARGS__; }     \
    else { constexpr int RB = 64; __VA_ARGS__; }                    \
  } while (0)
template <int RB>
__global__ __launch_bounds__(256) void se_fc_fwd_kernel(mds_se_fc_fwd_args a) {
  MDS_CHAIN_PRIO();
  __shared__ float hid[SE_RMAX], act[SE_RMAX];
  const int g = blockIdx.x, c = blockIdx.y * SE_CCH + threadIdx.x;
  const bool cok = c < a.C;
  // this thread's column of w2 and its bias do not depend on the hidden vector: requested first
  float w2c[RB], s = 0.f;
  if (cok) s = a.b2[c];
#pragma unroll
  for (int r = 0; r < RB; ++r)
    w2c[r] = (cok && r < a.R) ? (a.w2t ? a.w2t[(long)r * a.C + c] : a.w2[(long)c * a.R + r]) : 0.f;
  se_matvec_rc<RB>(a.w1, a.pooled + (long)g * a.C, a.R, a.C, hid);
  if (threadIdx.x < a.R) {
    const float h = hid[threadIdx.x] + a.b1[threadIdx.x];
    act[threadIdx.x] = siluf_(h);
    if (blockIdx.y == 0) a.hidden[g * a.R + threadIdx.x] = h;
  }
  __syncthreads();
  if (!cok) return;
#pragma unroll
  for (int r = 0; r < RB; ++r) s += w2c[r] * (r < a.R ? act[r] : 0.f);
  a.gate[(long)g * a.C + c] = sigmoidf_(s);
}
extern "C" int mds_se_fc_fwd(const mds_se_fc_fwd_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->groups > 0 && a->C > 0 && a->C % 4 == 0 && a->R > 0 && a->R <= SE_RMAX, "se_fc_fwd: bad dims (C % 4 == 0, R <= %d)", SE_RMAX);
  MDS_REQUIRE(a->pooled && a->w1 && a->b1 && a->w2 && a->b2 && a->hidden && a->gate, "se_fc_fwd: null pointer");
  MDS_REQUIRE((((uintptr_t)a->pooled | (uintptr_t)a->w1 | (uintptr_t)a->w2t) & 15) == 0, "se_fc_fwd: pooled / w1 / w2t must be 16-byte aligned (16-byte vector loads)");
  SE_DISPATCH_RB(a->R, MDS_LAUNCH(se_fc_fwd_kernel<RB>, dim3(a->groups, cdiv(a->C, SE_CCH)), dim3(256), 0, stream, *a));
  return mds_check_launch("se_fc_fwd");
}

// Backward, launch A — block (g, 256 channels):
//   dhpre[g][r] = silu'(hidden[g][r]) * sum_c de[g][c] * w2[c][r],   de = dgate * gate * (1 - gate)
//     (thread-per-channel partial sums over the contiguous w2 rows, reduced by wave shuffles + LDS)
//   dpooled[g][c] = sum_r dhpre[g][r] * w1[r][c] / rows
//   BatchNorm-backward sums of g = (u*gate + dpooled)*silu'(z) from the per-group partials of
//   mds_se_bwd_reduce:  sum g = sum_grp gate*A1 + dpooled*A3 ;  sum g*xh = sum_grp gate*A2 + dpooled*A4
template <int RB>
__global__ __launch_bounds__(256) void se_bwd_a_kernel(mds_se_fc_bwd_args a) {
  MDS_CHAIN_PRIO();
  __shared__ float part[4][SE_RMAX], dh[SE_RMAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x, C = a.C, R = a.R;
  const int c = blockIdx.y * SE_CCH + tid;
  const bool cok = c < C;
  // Everything that does not depend on dh is requested FIRST (the kernel is a chain of memory round
  // trips): this thread's column of w1, its gate, and the BatchNorm partial sums of its channel.
  float w1c[RB], gt_c = 0.f, bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < RB; ++r) w1c[r] = (cok && r < R) ? a.w1[(long)r * C + c] : 0.f;
  const bool bn = a.bnsums && a.bn_stats;
  if (cok) {
    gt_c = a.gate[(long)g * C + c];
    if (bn) {
      const float* b = a.bnsums + (long)g * a.bn_nblk * 4 * C + c;
#pragma unroll 8
      for (int k = 0; k < a.bn_nblk; ++k, b += 4 * C) {
        bs[0] += b[0]; bs[1] += b[C]; bs[2] += b[2 * C]; bs[3] += b[3 * C];
      }
    }
  }
  if (a.w2t) {  // de[c] staged once, then the block-wide [R][C] mat-vec
    __shared__ __attribute__((aligned(16))) float de_s[2048];
    for (int cc = tid; cc < C; cc += 256) {
      const float gt = a.gate[(long)g * C + cc];
      de_s[cc] = a.dgate[(long)g * C + cc] * gt * (1.0f - gt);
    }
    __syncthreads();
    se_matvec_rc<RB>(a.w2t, de_s, R, C, dh);
    if (tid < R) {
      const float v = dh[tid] * silu_gradf_(a.hidden[g * R + tid]);
      dh[tid] = v;
      if (blockIdx.y == 0) a.scratch[g * R + tid] = v;
    }
  } else {      // thread per channel over the contiguous w2 rows (strided across lanes: slow path)
    float acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = 0.f;
    for (int cc = tid; cc < C; cc += 256) {
      const float gt = a.gate[(long)g * C + cc];
      const float de = a.dgate[(long)g * C + cc] * gt * (1.0f - gt);
      const float* w = a.w2 + (long)cc * R;
#pragma unroll
      for (int r = 0; r < RB; ++r) acc[r] += de * w[r < R ? r : R - 1];
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const float s = wave_sum(acc[r]);
      if (lane == 0) part[wave][r] = s;
    }
    __syncthreads();
    if (tid < R) {
      const float s = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
      const float v = s * silu_gradf_(a.hidden[g * R + tid]);
      dh[tid] = v;
      if (blockIdx.y == 0) a.scratch[g * R + tid] = v;
    }
  }
  __syncthreads();
  if (!cok) return;
  float dp = 0.f;
#pragma unroll
  for (int r = 0; r < RB; ++r) dp += (r < R ? dh[r] : 0.f) * w1c[r];
  dp /= (float)a.rows_per_group;
  a.dpooled[(long)g * C + c] = dp;
  if (bn) {
    atomicAdd(a.bn_stats + c, (double)gt_c * bs[0] + (double)dp * bs[2]);
    atomicAdd(a.bn_stats + C + c, (double)gt_c * bs[1] + (double)dp * bs[3]);
  }
}
// launch B — parameter gradients: dw2[c][r], dw1[r][c] (thread per (r, c), c fastest); r == 0
// threads also do db2[c]; block 0 db1
MDS_DEV void se_bwd_b_body(const mds_se_fc_bwd_args& a) {
  const int G = a.groups, C = a.C, R = a.R;
  if (blockIdx.x == 0) {
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
      float s = 0.f;
      for (int g = 0; g < G; ++g) s += a.scratch[g * R + r];
      a.db1[r] += s;
    }
  }
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= R * C) return;
  const int r = e / C, c = e % C;
  float s2 = 0.f, s1 = 0.f, db2 = 0.f;
#pragma unroll 4
  for (int g = 0; g < G; ++g) {
    const float gt = a.gate[(long)g * C + c];
    const float de = a.dgate[(long)g * C + c] * gt * (1.0f - gt);
    db2 += de;
    s2 += de * siluf_(a.hidden[g * R + r]);
    s1 += a.scratch[g * R + r] * a.pooled[(long)g * C + c];
  }
  a.dw2[(long)c * R + r] += s2;
  a.dw1[(long)r * C + c] += s1;
  if (r == 0) a.db2[c] += db2;
}
__global__ __launch_bounds__(256) void se_bwd_b_kernel(mds_se_fc_bwd_args a) { se_bwd_b_body(a); }
// the same for a device-resident table of layers (grid.y = layer): the parameter gradients of the squeeze-excite layers are leaves
// of the backward - nothing downstream reads them before the optimizer - so the planner defers them and launches one table per
// gradient bucket instead of one 36 us latency-bound kernel per layer (20 per step)
__global__ __launch_bounds__(256) void se_bwd_b_table_kernel(const mds_se_fc_bwd_args* jobs) {
  const mds_se_fc_bwd_args a = jobs[blockIdx.y];
  if ((long)blockIdx.x * 256 >= (long)a.R * a.C) return;
  se_bwd_b_body(a);
}
static int se_fc_bwd_check(const mds_se_fc_bwd_args* a) {
  MDS_REQUIRE(a && a->groups > 0 && a->C > 0 && a->C % 4 == 0 && a->C <= 2048 && a->R > 0 && a->R <= SE_RMAX && a->rows_per_group > 0, "se_fc_bwd: bad dims (R <= %d, C % 4 == 0, C <= 2048)", SE_RMAX);
  MDS_REQUIRE(a->scratch && a->dgate && a->gate && a->hidden && a->pooled && a->dpooled, "se_fc_bwd: null pointer");
  MDS_REQUIRE(!(a->bnsums && a->bn_stats) || a->bn_nblk > 0, "se_fc_bwd: bn_nblk");
  MDS_REQUIRE((((uintptr_t)a->pooled | (uintptr_t)a->w1 | (uintptr_t)a->w2 | (uintptr_t)a->w2t | (uintptr_t)a->dgate | (uintptr_t)a->gate) & 15) == 0,
              "se_fc_bwd: pooled / w1 / w2 / w2t / dgate / gate must be 16-byte aligned (16-byte vector loads)");
  return 0;
}
extern "C" int mds_se_fc_bwd_data(const mds_se_fc_bwd_args* a, mds_stream_t stream) {
  if (int rc = se_fc_bwd_check(a)) return rc;
  SE_DISPATCH_RB(a->R, MDS_LAUNCH(se_bwd_a_kernel<RB>, dim3(a->groups, cdiv(a->C, SE_CCH)), dim3(256), 0, stream, *a));
  return mds_check_launch("se_fc_bwd_data");
}
extern "C" int mds_se_fc_bwd_params(const mds_se_fc_bwd_args* a, mds_stream_t stream) {
  if (int rc = se_fc_bwd_check(a)) return rc;
  MDS_REQUIRE(a->dw1 && a->db1 && a->dw2 && a->db2, "se_fc_bwd_params: null gradient pointer");
  MDS_LAUNCH(se_bwd_b_kernel, dim3(cdiv((long)a->R * a->C, 256)), dim3(256), 0, stream, *a);
  return mds_check_launch("se_fc_bwd_params");
}
extern "C" int mds_se_fc_bwd_params_table(const mds_se_fc_bwd_table_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->jobs && a->njobs > 0 && a->max_rc > 0, "se_fc_bwd_params_table: empty table");
  MDS_LAUNCH(se_bwd_b_table_kernel, dim3(cdiv(a->max_rc, 256), a->njobs), dim3(256), 0, stream, (const mds_se_fc_bwd_args*)a->jobs);
  return mds_check_launch("se_fc_bwd_params_table");
}
extern "C" int mds_se_fc_bwd(const mds_se_fc_bwd_args* a, mds_stream_t stream) {
  if (int rc = mds_se_fc_bwd_data(a, stream)) return rc;
  return mds_se_fc_bwd_params(a, stream);
}

// ------------------------------------------------------------------ head (dropout mask + Linear)
__global__ __launch_bounds__(256) void head_fwd_kernel(mds_head_fwd_args a) {
  // one block per (sample, class): 256 threads walk the F = 1280 features (it was one wave: 20 dependent trips, 15 us on the chain)
  // with `probs`: the first block of each group of `tta` samples does the whole group and writes the mean of their sigmoids
  __shared__ float part[4];
  const int b0 = blockIdx.x / a.NC, k = blockIdx.x % a.NC;
  const int tta = (a.probs && a.tta > 1) ? a.tta : 1;
  if (b0 % tta) return;
  float psum = 0.f;
  for (int t = 0; t < tta && b0 + t < a.B; ++t) {
    const int b = b0 + t;
    float s = 0.f;
    for (int f = threadIdx.x; f < a.F; f += 256) {
      float v = a.pooled[(long)b * a.F + f];
      if (a.mask) v *= a.mask[(long)b * a.F + f];
      s += v * a.w[(long)k * a.F + f];
    }
    s = wave_sum(s);
    __syncthreads();          // (the previous sample's partials have been read)
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    const float logit = ((part[0] + part[1]) + (part[2] + part[3])) + a.b[k];
    if (threadIdx.x == 0) a.logits[b * a.NC + k] = logit;
    psum += sigmoidf_(logit);
  }
  if (a.probs && threadIdx.x == 0) a.probs[(b0 / tta) * a.NC + k] = psum / (float)tta;
}
extern "C" int mds_head_fwd(const mds_head_fwd_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->B > 0 && a->F > 0 && a->NC > 0, "head_fwd: bad dims");
  MDS_REQUIRE(!a->probs || (a->tta >= 1 && a->B % a->tta == 0), "head_fwd: probs needs tta >= 1 dividing B");
  MDS_LAUNCH(head_fwd_kernel, dim3(a->B * a->NC), dim3(256), 0, stream, *a);
  return mds_check_launch("head_fwd");
}

__global__ void head_bwd_kernel(mds_head_bwd_args a) {
  for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < a.F; f += gridDim.x * blockDim.x) {
    for (int b = 0; b < a.B; ++b) {
      float s = 0.f;
      for (int k = 0; k < a.NC; ++k) s += a.dlogits[b * a.NC + k] * a.w[(long)k * a.F + f];
      float mk = a.mask ? a.mask[(long)b * a.F + f] : 1.0f;
      a.dpooled[(long)b * a.F + f] = s * mk;
    }
    for (int k = 0; k < a.NC; ++k) {
      float s = 0.f;
      for (int b = 0; b < a.B; ++b) {
        float mk = a.mask ? a.mask[(long)b * a.F + f] : 1.0f;
        s += a.dlogits[b * a.NC + k] * a.pooled[(long)b * a.F + f] * mk;
      }
      a.dw[(long)k * a.F + f] += s;
    }
  }
  if (blockIdx.x == 0) {
    for (int k = threadIdx.x; k < a.NC; k += blockDim.x) {
      float s = 0.f;
      for (int b = 0; b < a.B; ++b) s += a.dlogits[b * a.NC + k];
      a.db[k] += s;
    }
  }
}
extern "C" int mds_head_bwd(const mds_head_bwd_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->B > 0 && a->F > 0 && a->NC > 0, "head_bwd: bad dims");
  MDS_LAUNCH(head_bwd_kernel, dim3(cdiv(a->F, 256)), dim3(256), 0, stream, *a);
  return mds_check_launch("head_bwd");
}

