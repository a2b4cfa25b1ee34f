// k_misc.hip — error plumbing and the latency-class kernels (tiny per-channel / per-group math).
#include <stdarg.h>
#include <stdio.h>
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
  const int total = O * I * taps;
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
__global__ void bn_finalize_kernel(mds_bn_finalize_args a) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && a.training && a.num_batches_tracked) *a.num_batches_tracked += 1;
  if (c >= a.C) return;
  float mean, var;
  if (a.training) {
    double s = 0.0, ss = 0.0;
    for (int k = 0; k < MDS_STAT_SLOTS; ++k) {
      s += a.stats[(k * 2 + 0) * a.C + c];
      ss += a.stats[(k * 2 + 1) * a.C + c];
    }
    double m = s / (double)a.count;
    double v = ss / (double)a.count - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m;
    var = (float)v;
    if (a.running_mean) {
      float unb = a.count > 1 ? (float)(v * (double)a.count / (double)(a.count - 1)) : var;
      a.running_mean[c] = (1.0f - a.momentum) * a.running_mean[c] + a.momentum * mean;
      a.running_var[c] = (1.0f - a.momentum) * a.running_var[c] + a.momentum * unb;
    }
  } else {
    mean = a.running_mean[c];
    var = a.running_var[c];
  }
  float rstd = 1.0f / sqrtf(var + a.eps);
  float sc = a.gamma[c] * rstd;
  a.out[0 * a.C + c] = sc;
  a.out[1 * a.C + c] = a.beta[c] - mean * sc;
  a.out[2 * a.C + c] = mean;
  a.out[3 * a.C + c] = rstd;
}
extern "C" int mds_bn_finalize(const mds_bn_finalize_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->C > 0 && a->out && a->gamma && a->beta, "bn_finalize: bad args");
  MDS_REQUIRE(a->training ? (a->stats != 0 && a->count > 0) : (a->running_mean && a->running_var),
              "bn_finalize: missing stats / running buffers");
  MDS_LAUNCH(bn_finalize_kernel, dim3(cdiv(a->C, 128)), dim3(128), 0, stream, *a);
  return mds_check_launch("bn_finalize");
}

// ------------------------------------------------------------------ BN finalize (bwd)
__global__ void bn_bwd_finalize_kernel(mds_bn_bwd_finalize_args a) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.C) return;
  double sg = 0.0, sgx = 0.0;
  for (int k = 0; k < MDS_STAT_SLOTS; ++k) {
    sg += a.stats[(k * 2 + 0) * a.C + c];
    sgx += a.stats[(k * 2 + 1) * a.C + c];
  }
  if (a.dgamma) a.dgamma[c] += (float)sgx;
  if (a.dbeta) a.dbeta[c] += (float)sg;
  a.coef[0 * a.C + c] = a.gamma[c] * a.bn[3 * a.C + c];
  a.coef[1 * a.C + c] = (float)(sg / (double)a.count);
  a.coef[2 * a.C + c] = (float)(sgx / (double)a.count);
}
extern "C" int mds_bn_bwd_finalize(const mds_bn_bwd_finalize_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->C > 0 && a->stats && a->gamma && a->bn && a->coef && a->count > 0,
              "bn_bwd_finalize: bad args");
  MDS_LAUNCH(bn_bwd_finalize_kernel, dim3(cdiv(a->C, 128)), dim3(128), 0, stream, *a);
  return mds_check_launch("bn_bwd_finalize");
}

// ------------------------------------------------------------------ SE FCs
// Latency-class: groups <= 44, C <= 1152, R <= 48.  One wave per (group, r) dot product for the
// C-long reductions (coalesced, wave_sum); one thread per (group, c) / per c for the R-long ones.
__global__ void se_fc1_kernel(mds_se_fc_fwd_args a) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= a.groups * a.R) return;
  const int g = e / a.R, r = e % a.R;
  const float* w = a.w1 + (long)r * a.C;
  const float* p = a.pooled + (long)g * a.C;
  float s = 0.f;
  for (int c = lane; c < a.C; c += MDS_WAVE) s += w[c] * p[c];
  s = wave_sum(s);
  if (lane == 0) a.hidden[e] = s + a.b1[r];
}
__global__ void se_fc2_kernel(mds_se_fc_fwd_args a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.groups * a.C) return;
  const int g = e / a.C, c = e % a.C;
  const float* w = a.w2 + (long)c * a.R;
  const float* h = a.hidden + (long)g * a.R;
  float s = a.b2[c];
  for (int r = 0; r < a.R; ++r) s += w[r] * siluf_(h[r]);
  a.gate[e] = sigmoidf_(s);
}
extern "C" int mds_se_fc_fwd(const mds_se_fc_fwd_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->groups > 0 && a->C > 0 && a->R > 0, "se_fc_fwd: bad dims");
  MDS_REQUIRE(a->pooled && a->w1 && a->b1 && a->w2 && a->b2 && a->hidden && a->gate, "se_fc_fwd: null pointer");
  MDS_LAUNCH(se_fc1_kernel, dim3(cdiv(a->groups * a->R, 4)), dim3(256), 0, stream, *a);
  MDS_LAUNCH(se_fc2_kernel, dim3(cdiv((long)a->groups * a->C, 256)), dim3(256), 0, stream, *a);
  return mds_check_launch("se_fc_fwd");
}

// dhpre[g][r] = silu'(hidden) * sum_c de[g][c] * w2[c][r],  de = dgate * gate * (1 - gate)
__global__ void se_bwd1_kernel(mds_se_fc_bwd_args a) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= a.groups * a.R) return;
  const int g = e / a.R, r = e % a.R;
  float s = 0.f;
  for (int c = lane; c < a.C; c += MDS_WAVE) {
    const float gt = a.gate[(long)g * a.C + c];
    s += a.dgate[(long)g * a.C + c] * gt * (1.0f - gt) * a.w2[(long)c * a.R + r];
  }
  s = wave_sum(s);
  if (lane == 0) a.scratch[e] = s * silu_gradf_(a.hidden[e]);
}
// dpooled[g][c] = sum_r dhpre[g][r] * w1[r][c] / rows        (thread per (g, c))
__global__ void se_bwd2_kernel(mds_se_fc_bwd_args a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.groups * a.C) return;
  const int g = e / a.C, c = e % a.C;
  float dp = 0.f;
  for (int r = 0; r < a.R; ++r) dp += a.scratch[g * a.R + r] * a.w1[(long)r * a.C + c];
  a.dpooled[e] = dp / (float)a.rows_per_group;
}
// dw2[c][r], dw1[r][c] (thread per (r, c), c fastest); r == 0 threads also do db2[c]; block 0 db1
__global__ void se_bwd3_kernel(mds_se_fc_bwd_args a) {
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
// BatchNorm-backward sums of g = (u*gate + dpooled)*silu'(z) from the per-group partials of
// mds_se_bwd_reduce:  sum g = sum_grp gate*A1 + dpooled*A3 ;  sum g*xh = sum_grp gate*A2 + dpooled*A4
__global__ void se_bwd4_kernel(mds_se_fc_bwd_args a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (group, channel)
  if (e >= a.groups * a.C) return;
  const int g = e / a.C, c = e % a.C;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  const float* b = a.bnsums + (long)g * a.bn_nblk * 4 * a.C + c;
  for (int k = 0; k < a.bn_nblk; ++k, b += 4 * a.C) {
    s[0] += b[0]; s[1] += b[a.C]; s[2] += b[2 * a.C]; s[3] += b[3 * a.C];
  }
  const float gt = a.gate[e], dp = a.dpooled[e];
  atomicAdd(a.bn_stats + c, gt * s[0] + dp * s[2]);
  atomicAdd(a.bn_stats + a.C + c, gt * s[1] + dp * s[3]);
}
extern "C" int mds_se_fc_bwd(const mds_se_fc_bwd_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->groups > 0 && a->C > 0 && a->R > 0 && a->rows_per_group > 0, "se_fc_bwd: bad dims");
  MDS_REQUIRE(a->scratch && a->dgate && a->gate && a->hidden && a->pooled && a->dpooled, "se_fc_bwd: null pointer");
  MDS_LAUNCH(se_bwd1_kernel, dim3(cdiv(a->groups * a->R, 4)), dim3(256), 0, stream, *a);
  MDS_LAUNCH(se_bwd2_kernel, dim3(cdiv((long)a->groups * a->C, 256)), dim3(256), 0, stream, *a);
  MDS_LAUNCH(se_bwd3_kernel, dim3(cdiv((long)a->R * a->C, 256)), dim3(256), 0, stream, *a);
  if (a->bnsums && a->bn_stats) {
    MDS_REQUIRE(a->bn_nblk > 0, "se_fc_bwd: bn_nblk");
    MDS_LAUNCH(se_bwd4_kernel, dim3(cdiv((long)a->groups * a->C, 256)), dim3(256), 0, stream, *a);
  }
  return mds_check_launch("se_fc_bwd");
}

// ------------------------------------------------------------------ head (dropout mask + Linear)
__global__ void head_fwd_kernel(mds_head_fwd_args a) {
  int b = blockIdx.x / a.NC, k = blockIdx.x % a.NC;
  float s = 0.f;
  for (int f = threadIdx.x; f < a.F; f += MDS_WAVE) {
    float v = a.pooled[(long)b * a.F + f];
    if (a.mask) v *= a.mask[(long)b * a.F + f];
    s += v * a.w[(long)k * a.F + f];
  }
  s = wave_sum(s);
  if (threadIdx.x == 0) a.logits[b * a.NC + k] = s + a.b[k];
}
extern "C" int mds_head_fwd(const mds_head_fwd_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->B > 0 && a->F > 0 && a->NC > 0, "head_fwd: bad dims");
  MDS_LAUNCH(head_fwd_kernel, dim3(a->B * a->NC), dim3(MDS_WAVE), 0, stream, *a);
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
