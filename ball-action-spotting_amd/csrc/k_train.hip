// k_train.hip — SURVEY 8(f) N2: the rest of a training step as single launches.
//   mds_focal_fwd_bwd : sigmoid focal loss, value and gradient in one pass        (src/losses.py:34-50)
//   mds_multi_adamw   : AdamW over every parameter tensor in one launch           (src/argus_models.py:62)
//   mds_multi_ema     : ModelEma.update over the whole state_dict in one launch   (src/ema.py:47-55)
// torch runs these as ~15 + 9 + ~1000 launches per step (rocprof, round 2: 157 torch launches / 0.83 ms per step
// with the fused AdamW, before any EMA); all three are HBM-bound streams over at most 4 x 27 MB.
#include "elem.h"

// ------------------------------------------------------------------ focal loss
__global__ __launch_bounds__(256) void focal_kernel(mds_focal_args a) {
  __shared__ float red[4];
  float part = 0.f;
  const float inv = a.reduction == MDS_REDUCE_MEAN ? 1.0f / (float)a.n : 1.0f;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < a.n; e += (long)gridDim.x * blockDim.x) {
    const float x = a.x[e], t = a.t[e];
    const float p = 1.0f / (1.0f + expf(-x));
    // ce = max(x,0) - x*t + log(1 + exp(-|x|))   (BCEWithLogits, numerically stable);  d ce / dx = p - t
    const float ce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
    const float pt = p * t + (1.0f - p) * (1.0f - t);
    const float om = 1.0f - pt;                       // >= 0
    const float mod = om > 0.f ? powf(om, a.gamma) : 0.f;
    // d(1 - p_t)/dx = -(2t - 1) * p * (1 - p)
    const float dmod = om > 0.f ? a.gamma * powf(om, a.gamma - 1.0f) * (-(2.0f * t - 1.0f) * p * (1.0f - p)) : 0.f;
    float l = ce * mod, dl = (p - t) * mod + ce * dmod;
    if (a.alpha >= 0.f) {
      const float at = a.alpha * t + (1.0f - a.alpha) * (1.0f - t);
      l *= at; dl *= at;
    }
    a.dx[e] = dl * inv;
    if (a.reduction == MDS_REDUCE_NONE) a.loss[e] = l;
    else part += l;
  }
  if (a.reduction != MDS_REDUCE_NONE) {
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(a.loss, ((red[0] + red[1]) + (red[2] + red[3])) * inv);
  }
}
extern "C" int mds_focal_fwd_bwd(const mds_focal_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->n > 0 && a->x && a->t && a->loss && a->dx, "focal_fwd_bwd: bad args");
  MDS_REQUIRE(a->reduction >= MDS_REDUCE_NONE && a->reduction <= MDS_REDUCE_SUM, "focal_fwd_bwd: reduction");
  int blocks = cdiv(a->n, 256);
  if (blocks > 1024) blocks = 1024;
  MDS_LAUNCH(focal_kernel, dim3(blocks), dim3(256), 0, stream, *a);
  return mds_check_launch("focal_fwd_bwd");
}

// ------------------------------------------------------------------ multi-tensor AdamW / EMA
// One block per chunk of MDS_OPT_CHUNK elements of one tensor (the chunk list is built once on the host): 16-byte
// accesses when the four streams are 16-byte aligned at the chunk start (always for the flat buffers; parameters
// are separate torch allocations, 256-byte aligned), scalar tail otherwise.
__global__ __launch_bounds__(256) void adamw_kernel(mds_adamw_args a) {
  const bool skip = a.found_inf && *a.found_inf != 0.f;     // GradScaler: skip the step
  float bias1 = a.bias1, bias2 = a.bias2;
  if (a.step_in) {                                          // device step counter: a skipped step does not count
    const float t = *a.step_in + (skip ? 0.f : 1.f);
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.step_out = t;
    // in double, as torch.optim.AdamW's host arithmetic: 1 - powf(0.999f, 1) is 0.00099998713, not 0.001 (1.3e-5 off at t = 1)
    const double b1 = a.beta1_d != 0.0 ? a.beta1_d : (double)a.beta1, b2 = a.beta2_d != 0.0 ? a.beta2_d : (double)a.beta2;
    bias1 = (float)(1.0 - pow(b1, (double)t)); bias2 = (float)(1.0 - pow(b2, (double)t));
  }
  if (skip) return;
  const int ti = a.chunks[2 * blockIdx.x], c0 = a.chunks[2 * blockIdx.x + 1];
  const mds_opt_tensor T = a.table[ti];
  const long beg = (long)c0 * MDS_OPT_CHUNK;
  long end = beg + MDS_OPT_CHUNK;
  if (end > T.n) end = T.n;
  float* p = T.p;
  const float* g = a.gbase + T.goff;
  float* m = a.exp_avg + T.soff;
  float* v = a.exp_avg_sq + T.soff;
  const float decay = 1.0f - a.lr * a.weight_decay, step = a.lr / bias1, rb2 = 1.0f / sqrtf(bias2);
  const float inv_scale = a.grad_scale ? 1.0f / *a.grad_scale : 1.0f;    // GradScaler: unscale on load (torch multiplies by 1/scale too)
  auto upd = [&](float& pp, float gg, float& mm, float& vv) {
    gg *= inv_scale;
    pp *= decay;
    mm = a.beta1 * mm + (1.0f - a.beta1) * gg;
    vv = a.beta2 * vv + (1.0f - a.beta2) * gg * gg;
    pp -= step * mm / (sqrtf(vv) * rb2 + a.eps);
  };
  const bool al = ((((uintptr_t)(p + beg)) | ((uintptr_t)(g + beg)) | ((uintptr_t)(m + beg)) | ((uintptr_t)(v + beg))) & 15) == 0;
  if (al) {
    const long nv = (end - beg) >> 2;
    for (long e = threadIdx.x; e < nv; e += 256) {
      const long o = beg + 4 * e;
      f32x4 pp = *(f32x4*)(p + o), mm = *(f32x4*)(m + o), vv = *(f32x4*)(v + o);
      const f32x4 gg = *(const f32x4*)(g + o);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float pj = pp[j], mj = mm[j], vj = vv[j];
        upd(pj, gg[j], mj, vj);
        pp[j] = pj; mm[j] = mj; vv[j] = vj;
      }
      *(f32x4*)(p + o) = pp; *(f32x4*)(m + o) = mm; *(f32x4*)(v + o) = vv;
    }
    for (long o = beg + 4 * nv + threadIdx.x; o < end; o += 256) upd(p[o], g[o], m[o], v[o]);
  } else {
    for (long o = beg + threadIdx.x; o < end; o += 256) upd(p[o], g[o], m[o], v[o]);
  }
}
extern "C" int mds_multi_adamw(const mds_adamw_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->table && a->chunks && a->nchunks > 0 && a->exp_avg && a->exp_avg_sq, "multi_adamw: bad args");
  MDS_REQUIRE(a->step_in || (a->bias1 > 0.f && a->bias2 > 0.f), "multi_adamw: bias corrections must be positive (step >= 1)");
  MDS_REQUIRE(!a->step_in || (a->step_out && a->step_out != a->step_in), "multi_adamw: the device step counter needs two distinct scalars");
  MDS_LAUNCH(adamw_kernel, dim3(a->nchunks), dim3(256), 0, stream, *a);
  return mds_check_launch("multi_adamw");
}

// SGD with momentum / Nesterov over the same tables (torch.optim.SGD: weight decay added to the gradient, buffer
// initialised WITH the first gradient, dampening applied from the second step on)
__global__ __launch_bounds__(256) void sgd_kernel(mds_sgd_args a) {
  const bool skip = a.found_inf && *a.found_inf != 0.f;
  bool first = a.first != 0;
  if (a.step_in) {
    const float t0 = *a.step_in;
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.step_out = t0 + (skip ? 0.f : 1.f);
    first = t0 == 0.f;
  }
  if (skip) return;
  const int ti = a.chunks[2 * blockIdx.x], c0 = a.chunks[2 * blockIdx.x + 1];
  const mds_opt_tensor T = a.table[ti];
  const long beg = (long)c0 * MDS_OPT_CHUNK;
  long end = beg + MDS_OPT_CHUNK;
  if (end > T.n) end = T.n;
  float* p = T.p;
  const float* g = a.gbase + T.goff;
  const bool mom = a.momentum != 0.f;
  float* b = mom ? a.momentum_buf + T.soff : p;      // (never dereferenced without momentum)
  const float inv_scale = a.grad_scale ? 1.0f / *a.grad_scale : 1.0f;
  const float keep = 1.0f - a.dampening;
  auto upd = [&](float& pp, float gg, float& bb) {
    gg = gg * inv_scale + a.weight_decay * pp;
    if (mom) {
      bb = first ? gg : a.momentum * bb + keep * gg;
      gg = a.nesterov ? gg + a.momentum * bb : bb;
    }
    pp -= a.lr * gg;
  };
  const bool al = ((((uintptr_t)(p + beg)) | ((uintptr_t)(g + beg)) | ((uintptr_t)(b + beg))) & 15) == 0;
  long o0 = beg;
  if (al) {
    const long nv = (end - beg) >> 2;
    for (long e = threadIdx.x; e < nv; e += 256) {
      const long o = beg + 4 * e;
      f32x4 pp = *(f32x4*)(p + o), bb = mom ? *(f32x4*)(b + o) : (f32x4){0.f, 0.f, 0.f, 0.f};
      const f32x4 gg = *(const f32x4*)(g + o);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float pj = pp[j], bj = bb[j];
        upd(pj, gg[j], bj);
        pp[j] = pj; bb[j] = bj;
      }
      *(f32x4*)(p + o) = pp;
      if (mom) *(f32x4*)(b + o) = bb;
    }
    o0 = beg + 4 * nv;
  }
  for (long o = o0 + threadIdx.x; o < end; o += 256) {
    float bj = mom ? b[o] : 0.f;
    upd(p[o], g[o], bj);
    if (mom) b[o] = bj;
  }
}
extern "C" int mds_multi_sgd(const mds_sgd_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->table && a->chunks && a->nchunks > 0, "multi_sgd: bad args");
  MDS_REQUIRE(a->momentum == 0.f || a->momentum_buf, "multi_sgd: momentum needs a buffer");
  MDS_REQUIRE(!a->nesterov || (a->momentum > 0.f && a->dampening == 0.f), "multi_sgd: Nesterov momentum requires a momentum and zero dampening");
  MDS_LAUNCH(sgd_kernel, dim3(a->nchunks), dim3(256), 0, stream, *a);
  return mds_check_launch("multi_sgd");
}

__global__ __launch_bounds__(256) void ema_kernel(mds_ema_args a) {
  const int ti = a.chunks[2 * blockIdx.x], c0 = a.chunks[2 * blockIdx.x + 1];
  const mds_opt_tensor T = a.table[ti];
  const long beg = (long)c0 * MDS_OPT_CHUNK;
  long end = beg + MDS_OPT_CHUNK;
  if (end > T.n) end = T.n;
  float* e_ = T.p;
  const float* s = a.gbase + T.goff;
  const float d = a.decay, od = 1.0f - a.decay;
  const bool al = ((((uintptr_t)(e_ + beg)) | ((uintptr_t)(s + beg))) & 15) == 0;
  long o0 = beg;
  if (al) {
    const long nv = (end - beg) >> 2;
    for (long e = threadIdx.x; e < nv; e += 256) {
      const long o = beg + 4 * e;
      f32x4 ev = *(f32x4*)(e_ + o);
      const f32x4 sv = *(const f32x4*)(s + o);
#pragma unroll
      for (int j = 0; j < 4; ++j) ev[j] = d * ev[j] + od * sv[j];
      *(f32x4*)(e_ + o) = ev;
    }
    o0 = beg + 4 * nv;
  }
  for (long o = o0 + threadIdx.x; o < end; o += 256) e_[o] = d * e_[o] + od * s[o];
}
extern "C" int mds_multi_ema(const mds_ema_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->table && a->chunks && a->nchunks > 0, "multi_ema: bad args");
  MDS_LAUNCH(ema_kernel, dim3(a->nchunks), dim3(256), 0, stream, *a);
  return mds_check_launch("multi_ema");
}
