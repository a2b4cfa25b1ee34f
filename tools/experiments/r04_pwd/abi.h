/* ---- data gradient of a 1x1 EXPANSION convolution (K = mid wide, N = cin <= 192) with the BatchNorm-backward apply pass
 * folded in (round 4, k_pwd.hip): dx[M][N] = dy[M][K] w[N][K]^T (+ residual), dy = A*g + B*y + D formed on load (dyp) and, if
 * dy_out is given, stored for the weight gradient; optionally the next BatchNorm backward's sums over dx (post: PLAIN / MASK).
 * A block owns 64 rows and all N columns and streams K: every wide element is read once.  Replaces mds_bn_bwd_apply +
 * mds_pw_fwd on the dependent chain (native_batch_norm_backward + the input half of convolution_backward behind
 * multidim_stacker.py:124-134 / timm InvertedResidual).                                                                 */
typedef struct {
  int dtype;
  long M;
  int K, N;             /* K: a multiple of 32 in 64 .. 2048; N: 48, 96, 112 or 192 */
  const void* x;        /* [M][K] dy, when dyp.mode == 0 */
  mds_dyp_t dyp;        /* mode 1: dy formed on load from g (PLAIN) and y */
  void* dy_out;         /* optional [M][K]: the formed dy is also stored */
  const void* w;        /* [N][K] (MDS_PACK_IO_FLIP of the conv weight) */
  void* y;              /* [M][N] out */
  const void* residual; /* optional [M][N], added */
  mds_poststat_t post;  /* NONE, PLAIN or MASK */
} mds_pw_dgrad_args;
int mds_pw_dgrad(const mds_pw_dgrad_args* a, mds_stream_t stream);
int mds_pw_dgrad_ok(long M, int K, int N);   /* 1 if mds_pw_dgrad takes the shape */
