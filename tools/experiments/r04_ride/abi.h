/* ---- BatchNorm-backward apply pass with a 1x1 weight gradient riding on it (round 4).  In the backward of an
 * inverted-residual block (multidim_stacker.py:124-134 and the timm twin) two streaming passes already touch exactly the
 * wide operands of the block's two 1x1 weight gradients; this entry point is mds_bn_bwd_apply (same dy, bit for bit in
 * bf16) that ALSO accumulates, per (row slab, channel chunk) block,
 *   wide_act == 0:  P[c][k] = sum_m dy[m][c] * x[m][k]                  dy = this pass's result (BN1: the weight gradient
 *                                                                       of conv_pw, x = the block input [M][K])
 *   wide_act == 1:  P[c][k] = sum_m (silu(z[m][c]) * gate[grp][c]) * x[m][k]   z = y*scale + shift (BN2: the weight gradient
 *                                                                       of the gated projection conv_pwl, x = its dy [M][K];
 *                                                                       g.mode must be MDS_G_SE_SILU, whose gate it is)
 * on the matrix cores (the 64 x CW tile goes through LDS; the narrow operand's K <= 192 columns are the whole tile width,
 * so every wide element is loaded once).  Each slab's tile is STORED (no atomics) to part[slab][C][K]; mds_wg_finish adds
 * the slabs in slab order into the parameter gradient - bit-identical from run to run.
 * Replaces one aten::convolution_backward weight-gradient GEMM per call (no second read of the wide tensor).          */
typedef struct {
  int dtype;
  long M;
  int C;                /* wide channels: the BatchNorm's; a multiple of 64 or of 96 */
  mds_gsrc_t g;         /* MDS_G_PLAIN, or MDS_G_SE_SILU */
  const void* y;        /* [M][C] raw conv output (pre-BN) */
  const float* bn;      /* [4][C] scale, shift, mean, rstd */
  const float* lin;     /* [3][C] A, B, D of dy = A*g + B*y + D (mds_bn_bwd_finalize) */
  void* dy;             /* [M][C] out */
  int K;                /* narrow width: 48, 96, 112 or 192 */
  const void* x;        /* [M][K] narrow operand */
  int wide_act;
  long group_rows;      /* > 0: slabs never straddle a multiple of group_rows (required with MDS_G_SE_SILU: = g.rows_per_group) */
  int slabs;            /* mds_bn_bwd_apply_wg_slabs(M, C, K, group_rows, wide_act, dtype) */
  float* part;          /* fp32 [slabs][C][K] scratch, fully overwritten */
} mds_bn_bwd_apply_wg_args;
int mds_bn_bwd_apply_wg(const mds_bn_bwd_apply_wg_args* a, mds_stream_t stream);
int mds_bn_bwd_apply_wg_slabs(long M, int C, int K, long group_rows, int wide_act, int dtype);   /* number of row slabs (rows of part) the launch will use */

/* dw (+)= sum over slabs of part[s][C][K] in slab order; transpose: dw is [K][C] (the gated projection's [cout][mid]) */
typedef struct {
  int C, K, slabs;
  int transpose;
  const float* part;
  float* dw;
} mds_wg_finish_args;
int mds_wg_finish(const mds_wg_finish_args* a, mds_stream_t stream);
