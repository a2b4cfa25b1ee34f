// k_elem.hip — HBM-bound row-streaming kernels: 16-byte vector loads, per-channel parameters
// in registers, LDS + wave reductions for the per-channel / per-group sums.
#include <type_traits>
#include "gemm.h"

// ------------------------------------------------------------------ bn_res
template <typename T>
__global__ __launch_bounds__(256) void bn_res_kernel(mds_bn_res_args a) {
  MDS_CHAIN_PRIO();
  const RowMap m = rowmap(a.C);
  if (!m.valid) return;
  const int c0 = m.chunk * 8;
  float sc[8], sh[8];
  load8f(a.scale + c0, sc);
  load8f(a.shift + c0, sh);
  const T* y = (const T*)a.y;
  const T* s = (const T*)a.shortcut;
  T* o = (T*)a.out;
  for (long row = (long)blockIdx.x * m.rpb + m.rsub; row < a.M; row += (long)gridDim.x * m.rpb) {
    float v[8];
    load8(y + row * a.C + c0, v);
    float mk = a.mask ? a.mask[(unsigned)row / (unsigned)a.rows_per_group] : 1.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float z = v[j] * sc[j] + sh[j];
      if (a.act) z = siluf_(z);
      v[j] = z * mk;
    }
    if (s) {
      float r[8];
      load8(s + row * a.C + c0, r);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += r[j];
    }
    store8(o + row * a.C + c0, v);
  }
}
extern "C" int mds_bn_res(const mds_bn_res_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->M < 4294967295L, "bn_res: M must be below 2^32 rows");
  MDS_REQUIRE(a && a->M > 0 && a->C > 0 && a->C % 8 == 0 && a->C <= 2048, "bn_res: bad dims M=%ld C=%d", a ? a->M : 0, a ? a->C : 0);
  MDS_REQUIRE(a->y && a->out && a->scale && a->shift, "bn_res: null pointer");
  MDS_REQUIRE(!a->mask || a->rows_per_group > 0, "bn_res: mask needs rows_per_group");
  MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH(bn_res_kernel<T>, dim3(stream_blocks(a->M, a->C)), dim3(256), 0, stream, *a));
  return mds_check_launch("bn_res");
}

// ------------------------------------------------------------------ grouped reductions
// grid = (blocks_per_group, groups); block walks rows of its group.
template <typename T>
__global__ __launch_bounds__(256) void se_pool_kernel(mds_se_pool_args a) {
  MDS_CHAIN_PRIO();
  __shared__ float red[256 * 8];
  const RowMap m = rowmap(a.C, gridDim.z, blockIdx.z);
  const int c0 = m.c0;
  float acc[1][8] = {{0, 0, 0, 0, 0, 0, 0, 0}};
  if (m.valid) {
    float sc[8], sh[8];
    const bool prebn = a.scale != 0;    // raw conv output: BN + SiLU here; else y already is the activation (mds_epi_t producer)
    if (prebn) { load8f(a.scale + c0, sc); load8f(a.shift + c0, sh); }
    const T* y = (const T*)a.y + (long)blockIdx.y * a.rows_per_group * a.C;
    T* act = a.act ? (T*)a.act + (long)blockIdx.y * a.rows_per_group * a.C : (T*)0;
    // four row slots per thread, ROLLING: a slot is refilled (next trip's row) as soon as its registers are converted, so three
    // to four 16-byte loads stay in flight per thread all the time (issue-all / wait-all / compute had the memory pipe idle
    // during the compute phase and the VALU idle during the wait).  Rows past the end read row 0 again (clamped: no branch
    // around a load - a conditional load costs a vmcnt(0)).
    const long R = a.rows_per_group, stride = (long)gridDim.x * m.rpb, r0 = (long)blockIdx.x * m.rpb + m.rsub;
    RawV8<T> raw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const long rr = r0 + k * stride; raw[k].ld(y + (rr < R ? rr : 0) * a.C + c0); }
    for (long r = r0; r < R; r += 4 * stride) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long rr = r + k * stride, rn = rr + 4 * stride;
        float v[8];
        raw[k].get(v);
        raw[k].ld(y + (rn < R ? rn : 0) * a.C + c0);
        if (rr < R) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { if (prebn) v[j] = siluf_(v[j] * sc[j] + sh[j]); acc[0][j] += v[j]; }
          if (act) store8(act + rr * a.C + c0, v);
        }
      }
    }
  }
  const float inv = 1.0f / (float)a.rows_per_group;
  double* pooled = a.pooled + (long)blockIdx.y * a.C + m.cbase;
  block_reduce_channels<1>(acc, m, red, [&](int, int ch, float s) { atomicAdd(pooled + ch, (double)(s * inv)); });
}
// blocks per group of a reduce kernel: every block ends with O(C) atomics / partial stores, so it
// must own enough rows to amortise them (8 passes made the C = 1152 layers tail-bound: 1.2 TB/s)
static int reduce_passes() { return mds_knob(MDS_KNOB_REDUCE_PASSES) > 0 ? mds_knob(MDS_KNOB_REDUCE_PASSES) : 32; }   // measured 8 / 16 / 32 / 64 / 128 at the four layer shapes
// (fwd: the pooling pass of the FORWARD, which has nothing beside it and no second operand stream - swept alone in round 5 with
// tools/fwd_time.py: 4 / 8 / 12 / 16 / 24 / 32 passes -> 4.131 / 4.122 / 4.118 / 4.108 / 4.140 / 4.172 ms of forward; the backward's
// se_bwd_reduce keeps 32, swept inside the step)
static inline int group_blocks(long rows, int C, bool fwd = false) {
  const long per = (long)rows_per_pass(C, row_slices(C)) * ((fwd && mds_knob(MDS_KNOB_REDUCE_PASSES) == 0) ? 16 : reduce_passes());
  const long b = (rows + per - 1) / per;
  return (int)(b > 64 ? 64 : (b < 1 ? 1 : b));
}
extern "C" int mds_se_pool(const mds_se_pool_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->groups > 0 && a->rows_per_group > 0 && a->C % 8 == 0 && a->C <= 2048, "se_pool: bad dims");
  MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH(se_pool_kernel<T>, dim3(group_blocks(a->rows_per_group, a->C, true), a->groups, row_slices(a->C)), dim3(256), 0, stream, *a));
  return mds_check_launch("se_pool");
}

#ifndef MDS_SEBR_OCC
#define MDS_SEBR_OCC 3
#endif
// row slots of se_bwd_reduce: four spill 13 registers at three blocks per CU (+0.2 ms per step), at two blocks per CU they are
// as fast as two slots at three (13.45 against 13.46 ms; 13.52 with the issue-all / wait-all loop)
#ifndef MDS_SEBR_SLOTS
#define MDS_SEBR_SLOTS 2
#endif
template <typename T>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? 2 : MDS_SEBR_OCC) void se_bwd_reduce_kernel(mds_se_bwd_reduce_args a) {
  MDS_CHAIN_PRIO();
  __shared__ float red[256 * 8];
  __shared__ float red2[256 * 8 * 2];
  const RowMap m = rowmap(a.C, gridDim.z, blockIdx.z);
  const int c0 = m.c0;
  float acc[1][8] = {{0, 0, 0, 0, 0, 0, 0, 0}};
  float bs[4][8];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) bs[k][j] = 0.f;
  const bool fuse_bn = a.bnsums != 0;
  if (m.valid) {
    float sc[8], sh[8], mu[8], rs[8];
    const bool raw = a.scale != 0;  // raw conv output: re-apply BN+SiLU; else y IS the activation
    if (raw) { load8f(a.scale + c0, sc); load8f(a.shift + c0, sh); }
    if (fuse_bn) { load8f(a.mean + c0, mu); load8f(a.rstd + c0, rs); }
    const long base = (long)blockIdx.y * a.rows_per_group * a.C;
    const T* y = (const T*)a.y + base;
    const T* u = (const T*)a.u + base;
    // rolling row slots (see se_pool_kernel)
    const long R = a.rows_per_group, stride = (long)gridDim.x * m.rpb, rb = (long)blockIdx.x * m.rpb + m.rsub;
    constexpr int NSL = MDS_SEBR_SLOTS;
    RawV8<T> rv[NSL], ru[NSL];
#pragma unroll
    for (int k = 0; k < NSL; ++k) {
      const long rr = rb + k * stride < R ? rb + k * stride : 0;
      rv[k].ld(y + rr * a.C + c0);
      ru[k].ld(u + rr * a.C + c0);
    }
    for (long r0 = rb; r0 < R; r0 += NSL * stride) {
#pragma unroll
      for (int k = 0; k < NSL; ++k) {
        const long rr = r0 + k * stride, rn = rr + NSL * stride < R ? rr + NSL * stride : 0;
        float v[8], uu[8];
        rv[k].get(v);
        ru[k].get(uu);
        rv[k].ld(y + rn * a.C + c0);
        ru[k].ld(u + rn * a.C + c0);
        if (rr >= R) continue;
        if (fuse_bn) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float z = v[j] * sc[j] + sh[j];
            const float sg = sigmoidf_(z);
            const float sp = sg * (1.0f + z * (1.0f - sg));     // silu'(z)
            const float xh = (v[j] - mu[j]) * rs[j];
            acc[0][j] += uu[j] * (z * sg);
            const float us = uu[j] * sp;
            bs[0][j] += us; bs[1][j] += us * xh; bs[2][j] += sp; bs[3][j] += sp * xh;
          }
        } else {
          if (raw) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = siluf_(v[j] * sc[j] + sh[j]);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[0][j] += uu[j] * v[j];
        }
      }
    }
  }
  double* dgate = a.dgate + (long)blockIdx.y * a.C + m.cbase;
  block_reduce_channels<1>(acc, m, red, [&](int, int ch, float s) { atomicAdd(dgate + ch, (double)s); });
  if (fuse_bn) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float part[2][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { part[0][j] = bs[2 * h][j]; part[1][j] = bs[2 * h + 1][j]; }
      block_reduce_rows<2>(part, m, red2);
      if (m.valid && m.rsub == 0) {
        float* dst = a.bnsums + (((long)blockIdx.y * gridDim.x + blockIdx.x) * 4 + 2 * h) * a.C + c0;
        store8(dst, part[0]);
        store8(dst + a.C, part[1]);
      }
    }
  }
}
extern "C" int mds_se_bwd_reduce_blocks(long rows_per_group, int C) { return group_blocks(rows_per_group, C); }
extern "C" int mds_se_bwd_reduce(const mds_se_bwd_reduce_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->groups > 0 && a->rows_per_group > 0 && a->C % 8 == 0 && a->C <= 2048, "se_bwd_reduce: bad dims");
  MDS_REQUIRE(!a->bnsums || (a->scale && a->shift && a->mean && a->rstd), "se_bwd_reduce: BN fusion needs raw y + scale/shift/mean/rstd");
  MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH(se_bwd_reduce_kernel<T>, dim3(group_blocks(a->rows_per_group, a->C), a->groups, row_slices(a->C)), dim3(256), 0, stream, *a));
  return mds_check_launch("se_bwd_reduce");
}

// ------------------------------------------------------------------ BN backward reduce / apply
// The fp32 instantiation (parity plans) accumulates in fp64 per thread and per block: the sums of the residual-stream BatchNorms
// cancel to ~1e-3 of their absolute mass, and fp32 partials over a block's ~10^4 rows were the whole error of the worst batch-4
// bias gradient (1.2 - 1.5e-3 of the float64 oracle, tests/test_fullsize_gpu.py).  bf16 plans keep fp32 partials (their inputs
// carry 2^-9 already).
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(mds_bn_bwd_reduce_args a) {
  MDS_CHAIN_PRIO();
  typedef typename std::conditional<std::is_same<T, float>::value, double, float>::type acc_t;
  __shared__ acc_t red[256 * 8 * 2];
  const RowMap m = rowmap(a.C, gridDim.y, blockIdx.y);
  const int c0 = m.c0;
  acc_t acc[2][8] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
  if (m.valid) {
    float sc[8], sh[8], mu[8], rs[8];
    load8f(a.bn + 0 * a.C + c0, sc);
    load8f(a.bn + 1 * a.C + c0, sh);
    load8f(a.bn + 2 * a.C + c0, mu);
    load8f(a.bn + 3 * a.C + c0, rs);
    const T* y = (const T*)a.y;
    const T* ug = (const T*)a.g.u;
    // 4 rows per trip, all 8 loads issued together: the grid is capped at 512 blocks (atomic tail), so a single
    // row in flight per thread left 16 KB per CU outstanding - a latency-bound 3 TB/s
    // rolling row slots (see se_pool_kernel)
    const long stride = (long)gridDim.x * m.rpb, rb = (long)blockIdx.x * m.rpb + m.rsub;
    RawV8<T> ry[4], ru[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long rr = rb + k * stride < a.M ? rb + k * stride : 0;
      ry[k].ld(y + rr * a.C + c0);
      ru[k].ld(ug + rr * a.C + c0);
    }
    for (long row = rb; row < a.M; row += 4 * stride) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long rr = row + k * stride, rn = rr + 4 * stride < a.M ? rr + 4 * stride : 0;
        float v[8], z[8], u[8], g[8];
        ry[k].get(v);
        ru[k].get(u);
        ry[k].ld(y + rn * a.C + c0);
        ru[k].ld(ug + rn * a.C + c0);
        if (rr < a.M) {
#pragma unroll
          for (int j = 0; j < 8; ++j) z[j] = v[j] * sc[j] + sh[j];
          eval_g_u(a.g, rr, c0, a.C, z, u, g);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            acc[0][j] += (acc_t)g[j];
            acc[1][j] += (acc_t)g[j] * (acc_t)((v[j] - mu[j]) * rs[j]);
          }
        }
      }
    }
  }
  double* st = a.stats + (long)(blockIdx.x % MDS_STAT_SLOTS) * 2 * a.C + m.cbase;     // fp64 slots (include/mds.h)
  block_reduce_channels<2>(acc, m, red, [&](int v, int ch, acc_t s) { atomicAdd(st + (long)v * a.C + ch, (double)s); });
}
extern "C" int mds_bn_bwd_reduce(const mds_bn_bwd_reduce_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->M < 4294967295L, "bn_bwd_reduce: M must be below 2^32 rows");
  MDS_REQUIRE(a && a->M > 0 && a->C % 8 == 0 && a->C <= 2048, "bn_bwd_reduce: bad dims");
  MDS_REQUIRE(a->g.u && a->y && a->bn && a->stats, "bn_bwd_reduce: null pointer");
  MDS_REQUIRE(a->g.mode == MDS_G_PLAIN || a->g.mode == MDS_G_SILU || a->g.rows_per_group > 0, "bn_bwd_reduce: rows_per_group");
  const int ns = row_slices(a->C);
  long nb = (a->M + rows_per_pass(a->C, ns) - 1) / rows_per_pass(a->C, ns);
  const long rcap = mds_knob(MDS_KNOB_REDUCE_BLOCKS) > 0 ? mds_knob(MDS_KNOB_REDUCE_BLOCKS) : 512;
  if (nb > rcap / ns) nb = rcap / ns;
  if (nb < 1) nb = 1;                                     // (a developer knob below the slice count)
  MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH(bn_bwd_reduce_kernel<T>, dim3((unsigned)nb, ns), dim3(256), 0, stream, *a));
  return mds_check_launch("bn_bwd_reduce");
}

#ifndef MDS_APPLY_SLOTS
#define MDS_APPLY_SLOTS 2
#endif
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(mds_bn_bwd_apply_args a) {
  MDS_CHAIN_PRIO();
  // C = 1152: 144 eight-channel chunks per row would leave 112 of 256 threads idle - two column slices (blockIdx.y) of 72
  // chunks x 3 rows per pass instead, as in the reduce kernels (row_slices)
  const RowMap m = rowmap(a.C, gridDim.y, blockIdx.y);
  if (!m.valid) return;
  const int c0 = m.c0;
  float sc[8], sh[8], mu[8], rs[8], k0[8], k1[8], k2[8];
  load8f(a.bn + 0 * a.C + c0, sc);
  load8f(a.bn + 1 * a.C + c0, sh);
  load8f(a.bn + 2 * a.C + c0, mu);
  load8f(a.bn + 3 * a.C + c0, rs);
  load8f(a.coef + 0 * a.C + c0, k0);
  load8f(a.coef + 1 * a.C + c0, k1);
  load8f(a.coef + 2 * a.C + c0, k2);
  // fp32 plans: the three per-channel means in fp64 (mds_bn_bwd_finalize_args.coef64)
  constexpr bool F64 = std::is_same<T, float>::value;
  double d1[8], d2[8], dm[8];
  const bool use64 = F64 && a.coef64 != nullptr;
  if (use64) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { d1[j] = a.coef64[0 * a.C + c0 + j]; d2[j] = a.coef64[1 * a.C + c0 + j]; dm[j] = a.coef64[2 * a.C + c0 + j]; }
  }
  const T* y = (const T*)a.y;
  T* dy = (T*)a.dy;
  // two rolling row slots (see se_pool_kernel): the next rows of both operands are requested before this row's arithmetic
  const T* ug = (const T*)a.g.u;
  const long stride = (long)gridDim.x * m.rpb, rb = (long)blockIdx.x * m.rpb + m.rsub;
  constexpr int NSL = MDS_APPLY_SLOTS;
  RawV8<T> ry[NSL], ru[NSL];
#pragma unroll
  for (int k = 0; k < NSL; ++k) {
    const long rr = rb + k * stride < a.M ? rb + k * stride : 0;
    ry[k].ld(y + rr * a.C + c0);
    ru[k].ld(ug + rr * a.C + c0);
  }
  for (long row = rb; row < a.M; row += NSL * stride) {
#pragma unroll
    for (int k = 0; k < NSL; ++k) {
      const long rr = row + k * stride, rn = rr + NSL * stride < a.M ? rr + NSL * stride : 0;
      float v[8], z[8], u[8], g[8];
      ry[k].get(v);
      ru[k].get(u);
      ry[k].ld(y + rn * a.C + c0);
      ru[k].ld(ug + rn * a.C + c0);
      if (rr < a.M) {
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = v[j] * sc[j] + sh[j];
        eval_g_u(a.g, rr, c0, a.C, z, u, g);
        if (use64) {
#pragma unroll
          for (int j = 0; j < 8; ++j) g[j] = (float)((double)k0[j] * (((double)g[j] - d1[j]) - ((double)v[j] - dm[j]) * (double)rs[j] * d2[j]));
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) g[j] = k0[j] * (g[j] - k1[j] - (v[j] - mu[j]) * rs[j] * k2[j]);
        }
        store8(dy + rr * a.C + c0, g);
      }
    }
  }
}
extern "C" int mds_bn_bwd_apply(const mds_bn_bwd_apply_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->M < 4294967295L, "bn_bwd_apply: M must be below 2^32 rows");
  MDS_REQUIRE(a && a->M > 0 && a->C % 8 == 0 && a->C <= 2048, "bn_bwd_apply: bad dims");
  MDS_REQUIRE(a->g.u && a->y && a->bn && a->coef && a->dy, "bn_bwd_apply: null pointer");
  const int ns = row_slices(a->C);
  long nb = (a->M + rows_per_pass(a->C, ns) - 1) / rows_per_pass(a->C, ns);
  const long cap = (mds_knob(MDS_KNOB_STREAM_BLOCKS) ? mds_knob(MDS_KNOB_STREAM_BLOCKS) : 2048) / ns;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH(bn_bwd_apply_kernel<T>, dim3((unsigned)nb, ns), dim3(256), 0, stream, *a));
  return mds_check_launch("bn_bwd_apply");
}

// ------------------------------------------------------------------ GeM (fp32 math)
// one block per (b,t) group.
template <typename T>
__global__ __launch_bounds__(256) void gem_fwd_kernel(mds_gem_fwd_args a) {
  __shared__ float red[256 * 8];
  const RowMap m = rowmap(a.C);
  const int c0 = m.chunk * 8;
  const float p = a.p[0];
  float acc[1][8] = {{0, 0, 0, 0, 0, 0, 0, 0}};
  if (m.valid) {
    float sc[8], sh[8];
    if (a.pro.mode != MDS_PRO_NONE) { load8f(a.pro.scale + c0, sc); load8f(a.pro.shift + c0, sh); }
    const T* y = (const T*)a.y + (long)blockIdx.x * a.rows_per_group * a.C;
    for (long r = m.rsub; r < a.rows_per_group; r += m.rpb) {
      float v[8];
      load8(y + r * a.C + c0, v);
      apply_pro8(a.pro.mode, v, sc, sh);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[0][j] += expf(p * logf(fmaxf(v[j], a.eps)));
    }
  }
  block_reduce_rows<1>(acc, m, red);
  if (m.valid && m.rsub == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float mean = acc[0][j] / (float)a.rows_per_group;
      a.pooled[(long)blockIdx.x * a.C + c0 + j] = expf(logf(mean) / p);
    }
  }
}
// row-split variant: grid (groups, splits); partial sums by atomics into the zeroed accumulator
template <typename T, int BWD>
__global__ __launch_bounds__(256) void gem_partial_kernel(int groups, long rows_per_group, int C, const void* yv, mds_pro_t pro,
                                                          const float* pp, float eps, double* accum) {
  __shared__ float red[256 * 8];
  const RowMap m = rowmap(C);
  const int c0 = m.chunk * 8;
  const float p = pp[0];
  float acc[1][8] = {{0, 0, 0, 0, 0, 0, 0, 0}};
  if (m.valid) {
    float sc[8], sh[8];
    if (pro.mode != MDS_PRO_NONE) { load8f(pro.scale + c0, sc); load8f(pro.shift + c0, sh); }
    const T* y = (const T*)yv + (long)blockIdx.x * rows_per_group * C;
    for (long r = (long)blockIdx.y * m.rpb + m.rsub; r < rows_per_group; r += (long)gridDim.y * m.rpb) {
      float v[8];
      load8(y + r * C + c0, v);
      apply_pro8(pro.mode, v, sc, sh);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float lc = logf(fmaxf(v[j], eps));
        acc[0][j] += BWD ? expf(p * lc) * lc : expf(p * lc);
      }
    }
  }
  double* dst = accum + (long)blockIdx.x * C;
  block_reduce_channels<1>(acc, m, red, [&](int, int ch, float s) { atomicAdd(dst + ch, (double)s); });
}
__global__ void gem_finish_kernel(int n, float rows, const float* pp, const double* accum, float* pooled) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) pooled[e] = expf(logf((float)(accum[e] / rows)) / pp[0]);
}
static inline int gem_splits(int groups, long rows, int C) {
  long per = (rows + rows_per_pass(C) - 1) / rows_per_pass(C);   // block passes per group
  long want = 1024 / groups;
  if (want < 1) want = 1;
  long s = per / 4;                                              // >= 4 passes per block
  if (s > want) s = want;
  return (int)(s < 1 ? 1 : s);
}
extern "C" int mds_gem_fwd(const mds_gem_fwd_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->groups > 0 && a->rows_per_group > 0 && a->C % 8 == 0 && a->C <= 2048, "gem_fwd: bad dims");
  MDS_REQUIRE(a->pro.mode == MDS_PRO_NONE || a->pro.mode == MDS_PRO_BN_SILU || a->pro.mode == MDS_PRO_AFFINE, "gem_fwd: prologue");
  if (a->accum) {
    const int sp = gem_splits(a->groups, a->rows_per_group, a->C);
    MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH((gem_partial_kernel<T, 0>), dim3(a->groups, sp), dim3(256), 0, stream, a->groups,
                                               a->rows_per_group, a->C, a->y, a->pro, a->p, a->eps, a->accum));
    const int n = a->groups * a->C;
    MDS_LAUNCH(gem_finish_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, n, (float)a->rows_per_group, a->p, a->accum, a->pooled);
    return mds_check_launch("gem_fwd");
  }
  MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH(gem_fwd_kernel<T>, dim3(a->groups), dim3(256), 0, stream, *a));
  return mds_check_launch("gem_fwd");
}

template <typename T>
__global__ __launch_bounds__(256) void gem_bwd_kernel(mds_gem_bwd_args a) {
  __shared__ float red[256 * 8];
  const RowMap m = rowmap(a.C);
  const int c0 = m.chunk * 8;
  const float p = a.p[0];
  const float R = (float)a.rows_per_group;
  float sc[8], sh[8];
  const long gbase = (long)blockIdx.x * a.rows_per_group * a.C;
  const T* y = (const T*)a.y + gbase;
  T* u = (T*)a.u + gbase;
  // pass 1: S = sum_rows c^p * log c
  float acc[1][8] = {{0, 0, 0, 0, 0, 0, 0, 0}};
  if (m.valid) {
    if (a.pro.mode != MDS_PRO_NONE) { load8f(a.pro.scale + c0, sc); load8f(a.pro.shift + c0, sh); }
    for (long r = m.rsub; r < a.rows_per_group; r += m.rpb) {
      float v[8];
      load8(y + r * a.C + c0, v);
      apply_pro8(a.pro.mode, v, sc, sh);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float lc = logf(fmaxf(v[j], a.eps));
        acc[0][j] += expf(p * lc) * lc;
      }
    }
  }
  block_reduce_rows<1>(acc, m, red);
  float dp_part = 0.f;
  float coef[8];  // dpooled * out / (mean * R)
  if (m.valid) {
    if (m.rsub == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[c0 + j] = acc[0][j];
    }
  }
  __syncthreads();
  if (m.valid) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float out = a.pooled[(long)blockIdx.x * a.C + c0 + j];
      float dpo = a.dpooled[(long)blockIdx.x * a.C + c0 + j];
      float mean = expf(p * logf(out));  // out^p
      coef[j] = dpo * out / (mean * R);
      if (m.rsub == 0) {
        float S = red[c0 + j] / R;  // mean(c^p log c)
        dp_part += dpo * out * (-logf(mean) / (p * p) + S / (p * mean));
      }
    }
    // pass 2: u = coef * c^(p-1) * [a >= eps]
    for (long r = m.rsub; r < a.rows_per_group; r += m.rpb) {
      float v[8];
      load8(y + r * a.C + c0, v);
      apply_pro8(a.pro.mode, v, sc, sh);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (v[j] >= a.eps) ? coef[j] * expf((p - 1.0f) * logf(v[j])) : 0.0f;
      store8(u + r * a.C + c0, v);
    }
  }
  // block-sum dp_part -> one atomic
  __syncthreads();
  red[threadIdx.x] = dp_part;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < 256; ++i) s += red[i];
    atomicAdd(a.dp, s);
  }
}
// row-split backward, second launch: u = coef * c^(p-1) * [a >= eps] with coef recomputed per block from
// the finished accumulator; split 0 of each group also adds the exponent's gradient
template <typename T>
__global__ __launch_bounds__(256) void gem_apply_kernel(mds_gem_bwd_args a) {
  __shared__ float red[256];
  const RowMap m = rowmap(a.C);
  const int c0 = m.chunk * 8;
  const float p = a.p[0], R = (float)a.rows_per_group;
  float dp_part = 0.f;
  if (m.valid) {
    float sc[8], sh[8], coef[8];
    if (a.pro.mode != MDS_PRO_NONE) { load8f(a.pro.scale + c0, sc); load8f(a.pro.shift + c0, sh); }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float out = a.pooled[(long)blockIdx.x * a.C + c0 + j];
      const float dpo = a.dpooled[(long)blockIdx.x * a.C + c0 + j];
      const float mean = expf(p * logf(out));  // out^p
      coef[j] = dpo * out / (mean * R);
      if (m.rsub == 0 && blockIdx.y == 0) {
        const float S = a.accum[(long)blockIdx.x * a.C + c0 + j] / R;  // mean(c^p log c)
        dp_part += dpo * out * (-logf(mean) / (p * p) + S / (p * mean));
      }
    }
    const long gbase = (long)blockIdx.x * a.rows_per_group * a.C;
    const T* y = (const T*)a.y + gbase;
    T* u = (T*)a.u + gbase;
    for (long r = (long)blockIdx.y * m.rpb + m.rsub; r < a.rows_per_group; r += (long)gridDim.y * m.rpb) {
      float v[8];
      load8(y + r * a.C + c0, v);
      apply_pro8(a.pro.mode, v, sc, sh);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (v[j] >= a.eps) ? coef[j] * expf((p - 1.0f) * logf(v[j])) : 0.0f;
      store8(u + r * a.C + c0, v);
    }
  }
  if (blockIdx.y == 0) {
    red[threadIdx.x] = dp_part;
    __syncthreads();
    if (threadIdx.x == 0) {
      float s = 0.f;
      for (int i = 0; i < 256; ++i) s += red[i];
      atomicAdd(a.dp, s);
    }
  }
}
extern "C" int mds_gem_bwd(const mds_gem_bwd_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->groups > 0 && a->rows_per_group > 0 && a->C % 8 == 0 && a->C <= 2048, "gem_bwd: bad dims");
  if (a->accum) {
    const int sp = gem_splits(a->groups, a->rows_per_group, a->C);
    MDS_DISPATCH_DTYPE(a->dtype, T, {
      MDS_LAUNCH((gem_partial_kernel<T, 1>), dim3(a->groups, sp), dim3(256), 0, stream, a->groups, a->rows_per_group, a->C, a->y,
                 a->pro, a->p, a->eps, a->accum);
      MDS_LAUNCH(gem_apply_kernel<T>, dim3(a->groups, sp), dim3(256), 0, stream, *a);
    });
    return mds_check_launch("gem_bwd");
  }
  MDS_DISPATCH_DTYPE(a->dtype, T, MDS_LAUNCH(gem_bwd_kernel<T>, dim3(a->groups), dim3(256), 0, stream, *a));
  return mds_check_launch("gem_bwd");
}
