// elem.h — shared device helpers for the row-streaming (HBM-bound) kernels.
#pragma once
#include "platform.h"

#define MDS_DISPATCH_DTYPE(dtype, T, ...)                                         \
  do {                                                                            \
    if ((dtype) == MDS_F32) { typedef float T; __VA_ARGS__; }                     \
    else if ((dtype) == MDS_BF16) { typedef bf16_t T; __VA_ARGS__; }              \
    else { mds_set_error("unsupported dtype %d", (int)(dtype)); return MDS_ERR_UNSUPPORTED; } \
  } while (0)

// A 256-thread block walks a [rows][C] tensor 8 channels (16 B bf16 / 32 B fp32) per thread:
// cpr = C/8 threads cover one row, rpb = 256/cpr rows per pass; a thread keeps its channel
// chunk for the whole kernel so per-channel parameters live in registers.
// Wide rows (C/8 > 128 chunks, i.e. C = 1152) would leave 112 of 256 threads idle with one row per
// pass; the reduce kernels then split the channels into `nslices` column slices (blockIdx.z).
struct RowMap {
  int cpr, rpb, chunk, rsub;   // chunk: this thread's 8-channel chunk WITHIN its slice (LDS index)
  int c0;                      // first channel of the thread (global)
  int cbase;                   // first channel of the block's column slice
  bool valid;
};
MDS_DEV RowMap rowmap(int C, int nslices = 1, int slice = 0) {
  RowMap m;
  m.cpr = (C >> 3) / nslices;
  m.rpb = 256 / m.cpr;
  if (m.rpb < 1) m.rpb = 1;
  m.chunk = threadIdx.x % m.cpr;
  m.rsub = threadIdx.x / m.cpr;
  m.valid = m.rsub < m.rpb;
  m.c0 = (slice * m.cpr + m.chunk) * 8;
  m.cbase = slice * m.cpr * 8;
  return m;
}
// host: column slices of a reduce kernel, and rows per block pass
static inline int row_slices(int C) { return ((C / 8) > 128 && (C / 8) % 2 == 0) ? 2 : 1; }
static inline int rows_per_pass(int C, int nslices = 1) { int cpr = C / 8 / nslices; int r = 256 / cpr; return r < 1 ? 1 : r; }
// host: number of blocks for a row-streaming kernel over M rows (cap ~2048 blocks, grid-stride)
static inline int stream_blocks(long M, int C) {
  long b = (M + rows_per_pass(C) - 1) / rows_per_pass(C);
  const long cap = mds_knob(MDS_KNOB_STREAM_BLOCKS) ? mds_knob(MDS_KNOB_STREAM_BLOCKS) : 2048;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

MDS_DEV void load8f(const float* p, float (&v)[8]) { load8(p, v); }

// Reduce NV 8-vectors across the rsub dimension of the block; threads with rsub == 0 end up
// holding the block totals for their chunk.  `red` is LDS of >= 256*8*NV floats.
template <int NV>
MDS_DEV void block_reduce_rows(float (&acc)[NV][8], const RowMap& m, float* red) {
  __syncthreads();
  if (m.valid) {
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int j = 0; j < 8; ++j) red[((m.rsub * NV + v) * m.cpr + m.chunk) * 8 + j] = acc[v][j];
  }
  __syncthreads();
  if (m.valid && m.rsub == 0) {
    for (int r = 1; r < m.rpb; ++r)
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[v][j] += red[((r * NV + v) * m.cpr + m.chunk) * 8 + j];
  }
}

// The same reduction, finished per CHANNEL: thread e owns (vector v, channel ch of the block's column slice), sums the block's
// rpb partial rows from LDS and hands (v, ch, sum) to `emit` - consecutive lanes = consecutive channels, so an atomic add or a
// store issued from `emit` is coalesced.  (The per-thread form above leaves rsub == 0 threads with 8 consecutive channels each:
// one atomic instruction of a wave then touches 64 different 64-byte segments.  tools/probes/reduce_probe.hip: a 780-block
// column reduction over 99 MB takes 44 us with 8 fp64 atomics per thread issued that way, 19.5 us with one per lane and
// channel, 17.5 us without any epilogue.)  `red`: LDS of >= 256*8*NV floats.  ch is relative to the slice (rowmap().cbase).
template <int NV, typename A, typename F>
MDS_DEV void block_reduce_channels(const A (&acc)[NV][8], const RowMap& m, A* red, F&& emit) {
  const int W = m.cpr * 8;
  __syncthreads();
  if (m.valid) {
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int j = 0; j < 8; ++j) red[(m.rsub * NV + v) * W + m.chunk * 8 + j] = acc[v][j];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < NV * W; e += 256) {
    const int v = e / W, ch = e - v * W;
    A s = red[v * W + ch];
    for (int r = 1; r < m.rpb; ++r) s += red[(r * NV + v) * W + ch];
    emit(v, ch, s);
  }
}

// gradient-source evaluation (see mds_gsrc_t): g = f(u, z, gate, dpooled, mask); u already in registers
MDS_DEV void eval_g_u(const mds_gsrc_t& gs, long row, int c0, int C, const float (&z)[8], const float (&u)[8], float (&g)[8]) {
  if (gs.mode == MDS_G_PLAIN) {
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = u[j];
  } else if (gs.mode == MDS_G_SILU) {
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = u[j] * silu_gradf_(z[j]);
  } else if (gs.mode == MDS_G_SE_SILU) {
    const long grp = (long)((unsigned)row / (unsigned)gs.rows_per_group);   // rows < 2^32: a 32-bit divide is ~4x cheaper
    float ga[8], dp[8];
    load8f(gs.gate + grp * C + c0, ga);
    load8f(gs.dpooled + grp * C + c0, dp);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = (u[j] * ga[j] + dp[j]) * silu_gradf_(z[j]);
  } else {  // MDS_G_MASK
    float mk = gs.mask ? gs.mask[(unsigned)row / (unsigned)gs.rows_per_group] : 1.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = u[j] * mk;
  }
}
template <typename T>
MDS_DEV void eval_g(const mds_gsrc_t& gs, long row, int c0, int C, const float (&z)[8], float (&g)[8]) {
  float u[8];
  load8((const T*)gs.u + row * (long)C + c0, u);
  eval_g_u(gs, row, c0, C, z, u, g);
}

// prologue evaluation on 8 channels of one row (scale/shift already in registers)
MDS_DEV void apply_pro8(int mode, float (&v)[8], const float (&sc)[8], const float (&sh)[8]) {
  if (mode == MDS_PRO_NONE) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float z = v[j] * sc[j] + sh[j];
    v[j] = (mode == MDS_PRO_AFFINE) ? z : siluf_(z);
  }
}
