/* mds.h — C ABI of libmds_hip.so: the MI355X (gfx950) kernels behind the drop-in
 * MultiDimStacker module (reference: /root/reference/src/models/multidim_stacker.py).
 *
 * The reference has no FFI of its own: its hot path is a Python nn.Module that dispatches to
 * ATen/cuDNN.  Each entry point below replaces the ATen call(s) issued by the cited reference
 * line(s); the Python host (ball-action-spotting_amd/mds) binds them with ctypes exactly as
 * INTEGRATION.md shows.
 *
 * Conventions
 *  - Plain C: pointers + sizes only.  Every buffer is caller-owned device memory (PyTorch
 *    allocations); the library never allocates, frees or retains pointers.
 *  - Every call is asynchronous on the caller's HIP stream (`stream` = hipStream_t as void*),
 *    never synchronises.  Process state: a thread-local error string, and (set once, idempotently) the
 *    per-kernel opt-in to more than 64 KiB of LDS; both are safe under concurrent callers.
 *  - Return 0 on success, negative MDS_ERR_* otherwise; mds_last_error() describes it.
 *  - Activations are channels-last "rows": a tensor [rows][C] with C contiguous, rows =
 *    N*H*W (2D) or B*T*H*W (3D).  dtype selects the storage type of activations and packed
 *    weights (MDS_F32 / MDS_BF16); statistics, gates, parameters' grads are always fp32.
 *  - "stats" buffers are fp64 [MDS_STAT_SLOTS][2][C], zeroed by the caller before the producing
 *    launch: slot s, row 0 = partial sum, row 1 = partial sum of squares (or of g*xhat).  Blocks reduce their rows
 *    in fp32 and add the partials to the slots with fp64 atomics: over 18 k - 4.7 M rows the sums behind a mean, a
 *    variance (E[y^2] - mean^2) and above all the backward sums (sum g cancels to ~1e-3 of its absolute mass on the
 *    residual stream) lose up to 1e-3 when thousands of partials are accumulated by fp32 atomics in a
 *    run-dependent order (round 2); in fp64 the cross-block accumulation is exact to 1e-16 and order-independent.
 *    The same holds for the other cross-block sums that feed activations or their gradients - the squeeze-excite pool
 *    (`pooled`), its backward (`dgate`) and the row-split GeM accumulators: all fp64.  What is left to fp32 atomics
 *    are the parameter-gradient accumulations (weight-gradient arena, GeM's dp), which feed nothing else in the step.
 */
#ifndef MDS_H
#define MDS_H
#ifdef __cplusplus
extern "C" {
#endif

#define MDS_VERSION 132
#define MDS_F32 0
#define MDS_BF16 1
#define MDS_STAT_SLOTS 32

#define MDS_ERR_BAD_ARG (-1)
#define MDS_ERR_UNSUPPORTED (-2)
#define MDS_ERR_LAUNCH (-3)

typedef void* mds_stream_t;

int mds_version(void);
const char* mds_last_error(void);
/* Developer knobs (process-wide, 0 = default).  MDS_KNOB_CONV_BLOCKS caps the grid of the persistent convolution
 * kernel so that the parity tests can drive its multi-tile software pipeline at small sizes. */
#define MDS_KNOB_CONV_BLOCKS 0
#define MDS_KNOB_DW_ORDER 1      /* depthwise grids: 0 = XCD-aware remap (default), 1 = channel chunk fastest, 2 = strip fastest (the former default) */
#define MDS_KNOB_PW_WRES 2     /* 1: mds_pw_fwd never takes the filter-resident kernel (A/B switch); 2: takes it at any M (tests); 3: lower row bar */
#define MDS_KNOB_WG_DBG 4      /* ablation bits of the 1x1 weight-gradient kernels (measurement only) */
#define MDS_KNOB_WG_BLOCKS 5   /* split-M block budget of mds_pw_wgrad (0 = default) */
#define MDS_KNOB_DW3_L 7       /* strip length of the 3x3x3 sliding-window kernels (0 = default) */
#define MDS_KNOB_STREAM_BLOCKS 8  /* block cap of the grid-stride elementwise kernels (0 = default) */
#define MDS_KNOB_DW2_L 9         /* strip length of the 3x3 stride-1 sliding-window kernels (0 = default rule) */
#define MDS_KNOB_PW_SPLIT 10       /* split-K of the small-M inference GEMMs: 0 = rule (mds_pw_fwd_split), 1 = never, n >= 2 = at most n */
#define MDS_KNOB_DW2_R 11          /* 1: the 3x3 stride-1 forward keeps six-row bands for small launches too (A/B) */
#define MDS_KNOB_PW_GY 12          /* block target of mds_pw_fwd when it spreads n-tiles over grid.y (0 = default 1536) */
#define MDS_KNOB_PW_BM64 13        /* row bar (in thousands) below which mds_pw_fwd takes 64-row tiles (0 = default 400) */
#define MDS_KNOB_REDUCE_PASSES 14  /* rows passes per block of the grouped reduce kernels (0 = default 32) */
#define MDS_KNOB_DW2_BLOCKS 15     /* block target of the 3x3 stride-1 strip rule (0 = default 640) */
#define MDS_KNOB_REDUCE_BLOCKS 16  /* block cap of mds_bn_bwd_reduce (0 = default) */
#define MDS_KNOB_PW_DEEP 17        /* 1: the fp32 inference launches of mds_pw_fwd keep ONE K chunk in flight (A/B; default: two) */
#define MDS_KNOB_PWK 18            /* K-streaming 1x1 GEMM (k_pwk8.hip): 0 = rule (forward launches), 1 = never, 2 = every legal shape (tests), 3 = rule + data gradients, 4 = data gradients only */
#define MDS_KNOB_PWK_BM 19         /* rows per tile of the K-streaming kernel: 0 = rule, 64 / 80 / 96 / 128 (A/B) */
#define MDS_KNOB_STEM_FWD 20       /* 1: the bf16 training stem forward takes the gather kernel instead of the LDS-tiled one (A/B) */
#define MDS_KNOB_DW3G 21           /* 1: the 3x3x3 depthwise forward at T != 5 takes the LDS-tiled kernel instead of the time-chunked sliding window (A/B) */
#define MDS_KNOB_C3 22             /* filter-in-registers 3x3 kernel (k_c3.hip): 0 = rule, 1 = never, 2 = every legal shape, any size (tests) */
#define MDS_KNOB_C3_DBG 23         /* ablation bits of k_c3.hip (measurement only): 1 no output stores, 2 no MFMAs, 4 no LDS-DMA, 8 no fragment reads, 16 the other helper-wave split, 32 no prologue form, 64 no stride-2 forward, 128 no row-streaming weight gradient, 256 not for the layer behind the prologue, 512 no stride-2 weight gradient */
#define MDS_KNOB_C3_BWD_BLOCKS 24   /* blocks of k_c3.hip's data-gradient launches (0 = rule): fewer than the CU count leaves CUs to the weight-gradient stream */
#define MDS_KNOB_COUNT 25
int mds_dev_set(int knob, int value);
/* Completion event of the NEXT launches of the calling thread (a hipEvent_t as void*; NULL disarms).  While armed, every kernel
 * this thread launches through the library is issued with the event as its STOP event (hipExtLaunchKernelGGL), i.e. the event is
 * bound to the last such kernel's own completion signal - no marker packet on the stream.  The planner uses it to let the
 * weight-gradient stream wait for a kernel of the dependent chain: an event RECORDED on that chain costs it 4.4 us (measured,
 * tools/probes/event_cost.py; 71 of them per training step), a stop event nothing.  Returns the number of launches that were
 * issued with the previously armed event (0 after an op that launched nothing: that event must not be waited on).            */
int mds_launch_event(void* event);

/* ---- output transform ("epilogue") for plans that KNOW the BatchNorm statistics before the producer runs (eval mode /
 * the predictor): the producer stores act(acc*scale[c] + shift[c]) instead of the raw convolution output, so no consumer
 * re-evaluates BN + SiLU while loading (the depthwise kernels do that 1.33x per element, halo included) and the
 * block-output pass (mds_bn_res) disappears into the projection's epilogue.  mode 0 = raw output (training). */
#define MDS_EPI_NONE 0
#define MDS_EPI_AFFINE 1      /* y = acc*scale + shift (+ residual)          */
#define MDS_EPI_BN_SILU 2     /* y = silu(acc*scale + shift) (+ residual)    */
typedef struct {
  int mode;
  const float* scale; /* [C] */
  const float* shift; /* [C] */
} mds_epi_t;

/* ---- operand transforms ("prologues"): how a consumer reads a producer's raw conv output.
 * Train-mode BatchNorm needs batch statistics before it can normalise, so producers store the
 * raw convolution output y (+ its per-channel sums) and every consumer applies
 *   a = act(y*scale[c] + shift[c]) [* gate[row/rows_per_group][c]]   while loading.          */
#define MDS_PRO_NONE 0
#define MDS_PRO_AFFINE 1       /* BN, no activation                                            */
#define MDS_PRO_BN_SILU 2      /* BN + SiLU   (timm BatchNormAct2d / BatchNormAct3d :53-69)     */
#define MDS_PRO_BN_SILU_GATE 3 /* BN + SiLU, then squeeze-excite gate (:72-90, timm SE)         */
#define MDS_PRO_GATE 4         /* squeeze-excite gate only: x is the materialised activation   */
typedef struct {
  int mode;
  const float* scale; /* [C] gamma*rstd                      */
  const float* shift; /* [C] beta - mean*gamma*rstd          */
  const float* gate;  /* [groups][C] sigmoid gate (mode 3)   */
  long rows_per_group;
} mds_pro_t;

/* ---- gradient sources: g (grad wrt a BatchNorm output z) is derived on the fly from an upstream tensor u:
 *   MDS_G_PLAIN      g = u
 *   MDS_G_SILU       g = u * silu'(z)
 *   MDS_G_SE_SILU    g = (u * gate[grp][c] + dpooled[grp][c]) * silu'(z)
 *   MDS_G_MASK       g = u * mask[grp]                                   (DropPath)           */
#define MDS_G_PLAIN 0
#define MDS_G_SILU 1
#define MDS_G_SE_SILU 2
#define MDS_G_MASK 3
typedef struct {
  int mode;
  const void* u;        /* [M][C] */
  const float* gate;    /* [groups][C] */
  const float* dpooled; /* [groups][C] */
  const float* mask;    /* [groups]    */
  long rows_per_group;
} mds_gsrc_t;

/* ---- "dy prologue": BatchNorm backward folded into the consumers of dy.  With the per-channel
 * coefficients lin = {A, B, D} written by mds_bn_bwd_finalize,
 *   dy = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)) = A*g + B*y + D          (y = raw conv output)
 * so a kernel that needs dy (the data-gradient GEMM, the weight-gradient GEMM) reads u and y and
 * forms dy while loading; the elementwise "apply" pass and the dy tensor disappear.
 * (replaces the second half of torch's native_batch_norm_backward behind BatchNormAct2d/3d.)   */
typedef struct {
  int mode;            /* 0: the operand pointer of the consumer IS dy;  1: dy formed on load      */
  mds_gsrc_t g;        /* u and how g derives from it: PLAIN or MASK (a SiLU factor is folded into u by the
                          producer, MDS_POST_SILU, or by mds_bn_bwd_apply)                           */
  const void* y;       /* [M][C] raw conv output of the BatchNorm's input                          */
  const float* bn;     /* [4][C] scale, shift, mean, rstd                                          */
  const float* lin;    /* [3][C] A, B, D                                                           */
} mds_dyp_t;

/* ---- "post statistics": the first half of BatchNorm backward (sum g, sum g*xhat) folded into the
 * epilogue of the kernel that PRODUCES u (a data-gradient GEMM): the output tile v is u of a BN layer
 * whose raw input is y.  MDS_POST_SILU also stores g = v*silu'(z) instead of v, so that every later
 * reader sees a PLAIN gradient source.                                                            */
#define MDS_POST_NONE 0
#define MDS_POST_PLAIN 1   /* g = v                         */
#define MDS_POST_MASK 2    /* g = v * mask[row / rows_per_group]; v itself is stored (it is also the shortcut gradient) */
#define MDS_POST_SILU 3    /* g = v * silu'(y*scale+shift); g is stored                                              */
typedef struct {
  int mode;
  const void* y;       /* [M][N] raw conv output */
  const float* bn;     /* [4][N] */
  const float* mask;   /* [groups] (MASK) */
  long rows_per_group;
  double* stats;       /* fp64 [SLOTS][2][N] caller-zeroed: sum g, sum g*xhat */
} mds_poststat_t;

/* ---- K4: 1x1 convolution = GEMM  y[M][N] = pro(x)[M][K] * w[N][K]^T  (+ residual)
 * replaces nn.Conv2d/Conv3d k=1 at multidim_stacker.py:106,120,179-183,199-203 and timm
 * conv_pw/conv_pwl; also used as its own data-gradient (w = transposed pack).                  */
typedef struct {
  int dtype;
  long M;
  int K, N;
  const void* x;        /* [M][K]            */
  const void* w;        /* [N][K] packed     */
  void* y;              /* [M][N]            */
  mds_pro_t pro;
  const void* residual; /* optional [M][N], added after the product                            */
  double* stats;         /* optional [SLOTS][2][N]                                               */
  mds_poststat_t post;  /* data-gradient use: BN-backward sums of the NEXT layer in the epilogue */
  mds_epi_t epi;        /* eval-mode output transform (no statistics, no post with it) */
  const void* w_frag;   /* optional second copy of w in MFMA-fragment order (MDS_PACK_FRAG_OI / _IO; bf16): with it the K-heavy
                           narrow-N launches (mds_pw_fwd_wants_frag) take the K-streaming kernel, whose filter operand goes
                           from memory straight to registers - one contiguous 1 KiB per (32-channel step, 16 columns)      */
  /* split-K for the small-M launches of inference plans (a 920-row layer is 15-30 blocks that each walk 36 K chunks
   * one memory round trip at a time): grid.z = split blocks share a tile, each stores its fp32 partial tile to
   * split_part[z][M][N]; the LAST block to finish a tile (split_ticket, self-resetting) adds the partials in z order -
   * deterministic - and runs the epilogue.  split <= 1: off.  mds_pw_fwd_split() gives the factor for a shape.   */
  int split;
  float* split_part;    /* fp32 [split][M][N] scratch */
  int* split_ticket;    /* [tiles * MDS_PW_SPLIT_TICKET_STRIDE], tiles = ceil(M / MDS_PW_SPLIT_TILE_ROWS) * ceil(N / 128); zero before the first launch */
  int form;             /* what the launch is, as the planner told mds_pw_fwd_wants_frag: 1 = forward, 2 = data gradient, 0 = not said
                           (then: data gradient if it has post statistics or a residual operand) - the K-streaming kernel's rule differs */
} mds_pw_fwd_args;
int mds_pw_fwd(const mds_pw_fwd_args* a, mds_stream_t stream);
int mds_pw_fwd_split(long M, int K, int N, int dtype);   /* recommended split-K factor (1 = none); <= MDS_PW_MAX_SPLIT */
int mds_pw_fwd_wants_frag(long M, int K, int N, int dtype, int data_gradient);   /* 1: a launch of this shape (forward / data-gradient form) uses w_frag when it is given one */
#define MDS_PW_MAX_SPLIT 16
#define MDS_PW_SPLIT_TILE_ROWS 64   /* tiles of a launch = ceil(M / 64) * ceil(N / 128) */
#define MDS_PW_SPLIT_TICKET_STRIDE 32   /* ints between two tiles' tickets: one 128-byte line each (atomics on one line serialise, ~0.13 us apiece) */

/* weight gradient of the 1x1 convolution: dw[N][K] += sum_m dy[m][n] * pro(x)[m][k] (fp32,
 * atomically accumulated into a caller-zeroed buffer laid out like the PyTorch parameter).     */
typedef struct {
  int dtype;
  long M;
  int K, N;
  const void* x;  /* [M][K] forward input (raw, pro re-applied) */
  const void* dy; /* [M][N]                                      */
  float* dw;      /* [N][K] fp32                                 */
  mds_pro_t pro;
} mds_pw_wgrad_args;
int mds_pw_wgrad(const mds_pw_wgrad_args* a, mds_stream_t stream);

/* ---- K2/K3: dense 3x3 convolution as an MFMA implicit GEMM over a tap list.
 * For output sub-grid point (a,b), a<A, b<B of image n:
 *   y[n][oy0 + a*os][ox0 + b*os][:] = sum_t pro(x)[n][a*is + dy[t]][b*is + dx[t]][:] * w[:][wi[t]][:]
 * (out-of-image input pixels contribute 0 AFTER the prologue = zero padding of the activation).
 * Covers: stride-1 'same' conv (is=1), TF-SAME stride-2 conv (is=2, timm Conv2dSame), the
 * stride-1 data gradient (flipped taps) and the stride-2 data gradient (os=2; one launch per
 * output parity class, or all four as tap groups of one launch).  replaces timm ConvBnAct.conv / EdgeResidual.conv_exp and their dgrads. */
#define MDS_MAX_TAPS 9
typedef struct {
  int dtype;
  int N, IH, IW, Cin;   /* input  [N][IH][IW][Cin]                      */
  int OH, OW, Cout;     /* output [N][OH][OW][Cout] (full tensor dims)  */
  int A, B;             /* sub-grid extent handled by this launch       */
  int oy0, ox0, os;     /* output position = (oy0 + a*os, ox0 + b*os)   */
  int is;               /* input stride                                  */
  int ntaps;
  int dy[MDS_MAX_TAPS], dx[MDS_MAX_TAPS], wi[MDS_MAX_TAPS];
  int wtaps;            /* taps in the packed weight: w[Cout][wtaps][Cin] */
  const void* x;
  const void* w;
  void* y;
  mds_pro_t pro;        /* modes NONE / AFFINE / BN_SILU                 */
  const void* residual; /* optional, same indexing as y                  */
  double* stats;         /* optional [SLOTS][2][Cout]                     */
  /* tap groups (ngroups = 2..4; 0/1 = none): the tap list is the concatenation of the groups' taps, every group is
   * evaluated from the SAME staged input patch and written to its own sub-grid (g_oy0 + a*os, g_ox0 + b*os), a < g_A,
   * b < g_B — the four output parities of the stride-2 data gradient in one launch (one read of dy instead of
   * four).  Needs is == 1, g_ntaps*Cin % 32 == 0, no residual / statistics; A, B = the largest g_A, g_B (oy0 = ox0 = 0). */
  int ngroups;
  int g_ntaps[4], g_oy0[4], g_ox0[4], g_A[4], g_B[4];
  mds_epi_t epi;        /* eval-mode output transform (no statistics with it; applied before `residual` is added) */
  mds_poststat_t post;  /* data-gradient use: the BatchNorm-backward sums of the layer BELOW in the epilogue (PLAIN / MASK with one mask value per
                           image); only launches for which mds_conv_dgrad_post_ok() says 1 may ask for it */
} mds_conv_fwd_args;
int mds_conv_fwd(const mds_conv_fwd_args* a, mds_stream_t stream);
/* 1: the bf16 data gradient of a 3x3 layer (forward shape N x IH x IW x Cin -> Cout, stride 1 / 2) takes a kernel that implements `post` */
int mds_conv_dgrad_post_ok(int dtype, int N, int IH, int IW, int Cin, int Cout, int stride, int has_residual);

/* weight gradient of the 3x3 convolution (forward geometry: is = stride, os = 1):
 * dw[co][ci][tap] += sum_{n,a,b} dy[n][a][b][co] * pro(x)[n][a*is+dy[t]][b*is+dx[t]][ci]
 * written in PyTorch OIHW order (dw[(co*Cin+ci)*wtaps + wi[t]]).                                */
typedef struct {
  int dtype;
  int N, IH, IW, Cin, OH, OW, Cout;
  int is, ntaps;
  int dy[MDS_MAX_TAPS], dx[MDS_MAX_TAPS], wi[MDS_MAX_TAPS];
  int wtaps;
  const void* x;
  const void* dyt; /* [N][OH][OW][Cout] */
  float* dw;
  mds_pro_t pro;
} mds_conv_wgrad_args;
int mds_conv_wgrad(const mds_conv_wgrad_args* a, mds_stream_t stream);

/* ---- stem: 3x3 stride-2 TF-SAME convolution of the fp32 frame triple (Cin = stack_size = 3
 * planes, NCHW as produced by x.view(b*S, 3, h, w), multidim_stacker.py:214) -> channels-last.  */
/* optional fused ingest (src/frames.py:7-31 + kornia hflip in src/predictors.py:63-64): the frame
 * triple is read from raw uint8 frames [nsrc][3][src_h][src_w]; constant-0 padding to H x W (pad_top / pad_left as
 * pad_to_frames computes them), /255 normalisation and — for images n >= nsrc, which re-read image n - nsrc —
 * the horizontal flip of test-time augmentation all happen in the gather of the im2col fragment.        */
typedef struct {
  const unsigned char* u8;  /* NULL: read x (fp32) as before */
  int nsrc, src_h, src_w;
  int pad_top, pad_left;
  float scale;              /* 1/255 */
} mds_ingest_t;
typedef struct {
  int dtype;
  int N, H, W, OH, OW, Cout; /* Cout <= 32 */
  int pad_t, pad_l;
  const float* x; /* [N][3][H][W] fp32   */
  const void* w;  /* [Cout][32] packed: k = plane*9 + ky*3 + kx, zero padded to 32 */
  void* y;        /* [N][OH][OW][Cout]   */
  double* stats;
  mds_ingest_t ingest;
  mds_epi_t epi;  /* eval-mode output transform */
} mds_stem_fwd_args;
int mds_stem_fwd(const mds_stem_fwd_args* a, mds_stream_t stream);

typedef struct {
  int dtype;
  int N, H, W, OH, OW, Cout;
  int pad_t, pad_l;
  const float* x;
  const void* dy; /* [N][OH][OW][Cout] */
  float* dw;      /* [Cout][3][3][3] fp32 (OIHW) */
  mds_dyp_t dyp;  /* dyp.mode == 1: dy is formed on load from the stem BatchNorm's backward inputs (`dy` ignored):
                     dy = A*g + B*y + D with g = u (MDS_G_PLAIN) or u*silu'(y*scale + shift) (MDS_G_SILU) - the stem has no
                     data gradient, so its BatchNorm-backward apply pass would only feed this kernel (bf16 path)          */
} mds_stem_wgrad_args;
int mds_stem_wgrad(const mds_stem_wgrad_args* a, mds_stream_t stream);

/* ---- K5/K6: depthwise 3x3 (2D, stride 1 or TF-SAME stride 2) and 3x3x3 (3D, pad 1).
 * Input is the raw output of the preceding 1x1 conv read through a BN+SiLU prologue.
 * replaces timm InvertedResidual.conv_dw and multidim_stacker.py:110-113.
 * 2D is the T == 1 case with kt restricted to the centre tap.                                   */
typedef struct {
  int dtype;
  int N, T, IH, IW, C; /* input [N][T][IH][IW][C] */
  int OH, OW;          /* output [N][T][OH][OW][C] */
  int stride, pad_t, pad_l;
  int kt;              /* 1 (2D) or 3 (3D, temporal pad 1) */
  const void* x;
  const float* w;      /* fp32 [C][kt*9] (PyTorch layout [C][1][kt][3][3]) */
  void* y;
  mds_pro_t pro;
  double* stats;
  mds_epi_t epi;       /* eval-mode output transform (sliding-window kernels: kt == 1, or kt == 3 with T == 5) */
  /* squeeze-excite pooling in the same pass (inference plans: the output IS the activation, so its per-image channel
   * means are the SE input - no mds_se_pool launch): pool[n][c] += pool_inv * sum over the image's stored outputs.
   * Needs an output transform, a sliding-window kernel, and kt == 3 or T == 1 (group = batch element n).        */
  double* pool;        /* optional fp64 [N][C], caller-zeroed */
  float pool_inv;      /* 1 / (T*OH*OW) */
} mds_dw_fwd_args;
int mds_dw_fwd(const mds_dw_fwd_args* a, mds_stream_t stream);

/* depthwise backward: given dy (grad of the raw dw output), produce
 *   g[n][..][c]  = (sum_taps dy * w) * silu'(z)      z = x*scale+shift   (grad wrt BN output of
 *                  the producing 1x1 conv, ready for its BN backward) + its stats (sum g, sum g*xhat)
 *   dw[c][tap]  += sum dy * silu(z)(shifted)                                                  */
typedef struct {
  int dtype;
  int N, T, IH, IW, C, OH, OW;
  int stride, pad_t, pad_l, kt;
  const void* x;      /* raw forward input (pre-BN)  */
  const void* dy;     /* [N][T][OH][OW][C]           */
  const float* w;
  void* g;            /* [N][T][IH][IW][C]           */
  float* dw;          /* fp32 [C][kt*9]              */
  mds_pro_t pro;      /* BN_SILU of the forward      */
  const float* mean;  /* [C] of x's BN (for xhat)    */
  const float* rstd;
  double* stats;      /* fp64 [SLOTS][2][C]: sum g, sum g*xhat */
} mds_dw_bwd_args;
int mds_dw_bwd(const mds_dw_bwd_args* a, mds_stream_t stream);

/* ---- K7: BatchNorm statistics -> per-channel affine (+ running-stat update, momentum, unbiased
 * running variance, num_batches_tracked += 1): torch.nn.functional.batch_norm semantics used by
 * timm BatchNormAct2d and BatchNormAct3d (:53-69).  out = fp32 [4][C]: scale, shift, mean, rstd. */
typedef struct {
  int C;
  long count;          /* elements per channel */
  const double* stats;  /* [SLOTS][2][C]; ignored when training == 0 */
  const float* gamma;
  const float* beta;
  float eps, momentum;
  int training;
  float* running_mean; /* updated in place when training */
  float* running_var;
  long long* num_batches_tracked;
  float* out;          /* [4][C] */
} mds_bn_finalize_args;
int mds_bn_finalize(const mds_bn_finalize_args* a, mds_stream_t stream);

/* eval-mode BatchNorm for a whole network in ONE launch: a device-resident table of layers, each turned into
 * out = [4][C] {scale, shift, mean, rstd} from its running statistics (what mds_bn_finalize does per layer with
 * training == 0; 72 launches per forward otherwise — the sliding-window predictor is launch-bound).          */
typedef struct {
  const float* gamma;
  const float* beta;
  const float* running_mean;
  const float* running_var;
  float* out;
  float eps;
  int C;
} mds_bn_eval_job;
int mds_bn_eval_table(const mds_bn_eval_job* jobs_dev, int njobs, int max_c, mds_stream_t stream);

/* y_out = act(bn(y)) * mask[row / rows_per_group] + shortcut     (block output materialisation:
 * BN3 + DropPath + residual, multidim_stacker.py:121-133 / timm blocks; or BN+SiLU of a projection) */
typedef struct {
  int dtype;
  long M;
  int C;
  const void* y;
  const float* scale;
  const float* shift;
  int act;                /* 0 none, 1 SiLU */
  const float* mask;      /* optional [groups] DropPath scale (0 or 1/keep) */
  long rows_per_group;
  const void* shortcut;   /* optional [M][C] */
  void* out;
} mds_bn_res_args;
int mds_bn_res(const mds_bn_res_args* a, mds_stream_t stream);

/* ---- K8: squeeze-excite.  pool: pooled[g][c] = mean_rows silu(bn(y))   (:82 x.mean((2,3,4))) */
typedef struct {
  int dtype;
  int groups;
  long rows_per_group;
  int C;
  const void* y;
  const float* scale;   /* NULL (with shift): y already IS the activation (producer with an mds_epi_t) */
  const float* shift;
  double* pooled; /* fp64 [groups][C], caller-zeroed (atomic accumulation of sums/rows; fp64: order-independent) */
  void* act;      /* optional [rows][C]: the activation silu(bn(y)) is also written out, so that the
                     three later consumers (gated 1x1 conv, its weight gradient, the SE backward
                     reduction) do not redo the exp/rcp work (they are VALU-bound otherwise) */
} mds_se_pool_args;
int mds_se_pool(const mds_se_pool_args* a, mds_stream_t stream);

/* gate = sigmoid(W2 * silu(W1 * pooled + b1) + b2)   (:83-86).  hidden pre-activations are kept. */
typedef struct {
  int groups, C, R;
  const double* pooled; /* fp64 [groups][C] (mds_se_pool) */
  const float* w1;      /* [R][C] */
  const float* b1;      /* [R]    */
  const float* w2;      /* [C][R] */
  const float* b2;      /* [C]    */
  float* hidden;        /* [groups][R] pre-activation */
  float* gate;          /* [groups][C] */
  const float* w2t;     /* optional [R][C] copy of w2 (MDS_PACK_IO_F32): lets the C-long dimension be
                           the coalesced one; without it the kernels walk w2 rows (slower)          */
} mds_se_fc_fwd_args;
int mds_se_fc_fwd(const mds_se_fc_fwd_args* a, mds_stream_t stream);

/* dgate_raw[g][c] = sum_rows u[row][c] * silu(bn(y))[row][c]   (u = grad wrt the gated tensor) */
typedef struct {
  int dtype;
  int groups;
  long rows_per_group;
  int C;
  const void* u;
  const void* y;      /* raw conv output, or the materialised activation when scale == NULL */
  const float* scale;
  const float* shift;
  double* dgate; /* fp64 [groups][C] caller-zeroed */
  /* optional fusion of the following BatchNorm-backward reduction (needs raw y + scale/shift):
   * with s' = silu'(z), xh = (y-mean)*rstd, per (group, channel) partial sums
   *   bnsums[g][blk][0..3][c] = sum u*s', sum u*s'*xh, sum s', sum s'*xh   over block blk's rows
   * (plain stores, no atomics: contended fp32 atomics were slower than the pass they replaced),
   * blk < mds_se_bwd_reduce_blocks(rows_per_group, C).  mds_se_fc_bwd forms sum g and sum g*xh of
   * g = (u*gate + dpooled)*s' from them without another pass over the two mid-width tensors.    */
  const float* mean;
  const float* rstd;
  float* bnsums; /* [groups][blocks][4][C] */
} mds_se_bwd_reduce_args;
int mds_se_bwd_reduce_blocks(long rows_per_group, int C);
int mds_se_bwd_reduce(const mds_se_bwd_reduce_args* a, mds_stream_t stream);

/* backward of the two SE FCs: in dgate_raw, gate, hidden, pooled; out dpooled[g][c] (grad wrt the
 * pooled mean, already divided by rows_per_group) and parameter grads (+=).                    */
typedef struct {
  int groups, C, R;
  long rows_per_group;
  const double* dgate;  /* fp64 (mds_se_bwd_reduce) */
  const float* gate;
  const float* hidden;
  const double* pooled; /* fp64 (mds_se_pool) */
  const float* w1;
  const float* w2;
  float* dpooled; /* [groups][C] */
  float* scratch; /* [groups][R] workspace (grad of the hidden pre-activations) */
  float* dw1;     /* [R][C] */
  float* db1;
  float* dw2;     /* [C][R] */
  float* db2;
  const float* bnsums; /* optional [groups][bn_nblk][4][C] from mds_se_bwd_reduce                  */
  int bn_nblk;         /* = mds_se_bwd_reduce_blocks(rows_per_group, C)                            */
  double* bn_stats;    /* optional fp64 [SLOTS][2][C] (zeroed): slot 0 receives sum g, sum g*xhat   */
  const float* w2t;    /* optional [R][C] copy of w2, see mds_se_fc_fwd_args                       */
} mds_se_fc_bwd_args;
int mds_se_fc_bwd(const mds_se_fc_bwd_args* a, mds_stream_t stream);
/* the same in two halves, for callers that overlap them: `_data` produces dpooled / the BatchNorm sums (on the
 * critical path of the backward pass), `_params` the four parameter gradients from what `_data` left in
 * `scratch` (a leaf of the dependency graph: the planner issues it on its second stream).            */
int mds_se_fc_bwd_data(const mds_se_fc_bwd_args* a, mds_stream_t stream);
int mds_se_fc_bwd_params(const mds_se_fc_bwd_args* a, mds_stream_t stream);
/* `_params` of several layers in ONE launch: a device-resident array of mds_se_fc_bwd_args (grid.y = layer).  The planner defers the
 * squeeze-excite parameter gradients of a gradient bucket and issues one table per bucket (20 launches per step -> <= 6).       */
typedef struct {
  const void* jobs;     /* device array of njobs mds_se_fc_bwd_args */
  int njobs;
  int max_rc;           /* max over the jobs of R * C */
} mds_se_fc_bwd_table_args;
int mds_se_fc_bwd_params_table(const mds_se_fc_bwd_table_args* a, mds_stream_t stream);

/* ---- BatchNorm backward, split in reduce / finalize / apply.  g (grad wrt the BN output z) is
 * derived on the fly from an upstream tensor u:
 *   (mds_gsrc_t, defined with the operand transforms at the top of this header)               */

typedef struct {
  int dtype;
  long M;
  int C;
  mds_gsrc_t g;
  const void* y;        /* raw conv output (pre-BN) */
  const float* bn;      /* [4][C] scale, shift, mean, rstd */
  double* stats;        /* fp64 [SLOTS][2][C]: sum g, sum g*xhat */
} mds_bn_bwd_reduce_args;
int mds_bn_bwd_reduce(const mds_bn_bwd_reduce_args* a, mds_stream_t stream);

/* sums -> dgamma, dbeta (+=) and coef[3][C] = { gamma*rstd, sum_g/M, sum_gx/M } */
typedef struct {
  int C;
  long count;
  const double* stats;  /* fp64 [SLOTS][2][C] */
  const float* gamma;
  const float* bn;  /* [4][C] */
  float* dgamma;    /* optional (NULL when the parameter is frozen) */
  float* dbeta;
  float* coef;      /* [3][C] */
  float* lin;       /* optional [3][C]: A, B, D of mds_dyp_t (dy = A*g + B*y + D)                          */
  int batch_stats;  /* 1: train-mode BN (batch statistics; the mean / xhat terms above).  0: eval-mode BN
                       (running statistics are constants): coef1 = coef2 = 0, i.e. dy = gamma*rstd*g        */
  const double* fwd_stats; /* optional, with coef64: the forward pass's fp64 [SLOTS][2][C] sums (mean in fp64)       */
  double* coef64;   /* optional [3][C]: sum_g/M, sum_gx/M and the batch mean in fp64, for mds_bn_bwd_apply's fp32
                       form (fp32 plans: the fp32-rounded means are the SAME for every row, so their rounding
                       error adds up M times in every sum over dy - the bias gradients upstream)            */
} mds_bn_bwd_finalize_args;
int mds_bn_bwd_finalize(const mds_bn_bwd_finalize_args* a, mds_stream_t stream);

/* dy = coef0 * (g - coef1 - xhat*coef2)  -> grad of the raw conv output */
typedef struct {
  int dtype;
  long M;
  int C;
  mds_gsrc_t g;
  const void* y;
  const float* bn;
  const float* coef;
  void* dy;
  const double* coef64;  /* optional (fp32 only): { sum_g/M, sum_gx/M, mean } in fp64 - the subtraction of the means is done
                            in fp64 and rounded once per element                                              */
} mds_bn_bwd_apply_args;
int mds_bn_bwd_apply(const mds_bn_bwd_apply_args* a, mds_stream_t stream);

/* ---- K11: GeM pooling over (H,W) of silu(bn(y)) for every (b, t, c):  multidim_stacker.py:35-45
 * pooled[b][t*C + c] = (mean_hw clamp(a, eps)^p)^(1/p);  fp32 math.                              */
typedef struct {
  int dtype;
  int groups;           /* b*t */
  long rows_per_group;  /* h*w */
  int C;
  const void* y;
  mds_pro_t pro;        /* NONE or BN_SILU */
  const float* p;       /* [1] learnable exponent */
  float eps;
  float* pooled;        /* [groups][C] == [b][t*C + c] */
  double* accum;        /* optional caller-zeroed fp64 [groups][C]: the rows of a group are then split over
                           several blocks (sum of clamp(a)^p by atomics + a finishing launch); without
                           it one block walks a whole group                                         */
} mds_gem_fwd_args;
int mds_gem_fwd(const mds_gem_fwd_args* a, mds_stream_t stream);

/* GeM backward: u[row][c] = grad wrt a = silu(bn(y)) (or wrt y when pro == NONE), dp += ...     */
typedef struct {
  int dtype;
  int groups;
  long rows_per_group;
  int C;
  const void* y;
  mds_pro_t pro;
  const float* p;
  float eps;
  const float* pooled;
  const float* dpooled; /* [groups][C] */
  void* u;              /* [rows][C] */
  float* dp;            /* [1] += */
  double* accum;        /* optional caller-zeroed fp64 [groups][C] (sum of c^p log c), as in mds_gem_fwd */
} mds_gem_bwd_args;
int mds_gem_bwd(const mds_gem_bwd_args* a, mds_stream_t stream);

/* ---- K12: dropout (host-supplied scaled mask) + Linear  (multidim_stacker.py:232-237)        */
typedef struct {
  int B, F, NC;
  const float* pooled; /* [B][F] */
  const float* mask;   /* optional [B][F] (0 or 1/(1-p)) */
  const float* w;      /* [NC][F] */
  const float* b;      /* [NC]    */
  float* logits;       /* [B][NC] */
  float* probs;        /* optional [B / tta][NC]: mean over each group of `tta` consecutive samples of sigmoid(logits) - the
                          predictor's nn.Sigmoid + torch.mean over the TTA pair (src/predictors.py:69-70) in the same launch */
  int tta;             /* samples per group (>= 1; used with probs only) */
} mds_head_fwd_args;
int mds_head_fwd(const mds_head_fwd_args* a, mds_stream_t stream);

typedef struct {
  int B, F, NC;
  const float* pooled;
  const float* mask;
  const float* w;
  const float* dlogits; /* [B][NC] */
  float* dpooled;       /* [B][F]  */
  float* dw;            /* [NC][F] += */
  float* db;            /* [NC]    += */
} mds_head_bwd_args;
int mds_head_bwd(const mds_head_bwd_args* a, mds_stream_t stream);

/* ---- what surrounds the hot path in a training step, as single launches.
 * Sigmoid focal loss forward + backward in one pass (src/losses.py:34-50): per element
 *   p = sigmoid(x), ce = BCEWithLogits(x, t), p_t = p*t + (1-p)*(1-t), l = [alpha_t] * ce * (1-p_t)^gamma
 * writes the reduced loss (mean / sum; or the per-element losses for reduction "none") and dl/dx (already
 * divided by n for "mean"), so autograd's backward is one scale by the incoming gradient.               */
#define MDS_REDUCE_NONE 0
#define MDS_REDUCE_MEAN 1
#define MDS_REDUCE_SUM 2
typedef struct {
  long n;
  const float* x;      /* [n] logits (fp32) */
  const float* t;      /* [n] targets       */
  float alpha, gamma;
  int reduction;
  float* loss;         /* [1] caller-zeroed accumulator (MEAN / SUM), or [n] (NONE) */
  float* dx;           /* [n] d loss / d x (for NONE: d l_i / d x_i)                 */
} mds_focal_args;
int mds_focal_fwd_bwd(const mds_focal_args* a, mds_stream_t stream);

/* Multi-tensor AdamW (torch.optim.AdamW semantics: decoupled weight decay, bias correction) over a
 * device-resident table of tensors in ONE launch; src/argus_models.py:62 `scaler.step(self.optimizer)`.
 * Gradients are addressed as gbase + goff (the engine hands all gradients over as views of one flat
 * buffer, so the table survives from step to step) and the moments live in two flat buffers.        */
typedef struct {
  float* p;            /* parameter                                   */
  long goff;           /* gradient = (const float*)gbase + goff        */
  long soff;           /* moments  = exp_avg + soff, exp_avg_sq + soff */
  long n;
} mds_opt_tensor;
typedef struct {
  const mds_opt_tensor* table;   /* device [ntensors] */
  const int* chunks;             /* device [nchunks][2]: tensor index, first element / MDS_OPT_CHUNK */
  int nchunks;
  const float* gbase;
  float* exp_avg;
  float* exp_avg_sq;
  float lr, beta1, beta2, eps, weight_decay;
  float bias1, bias2;            /* 1 - beta1^t, 1 - beta2^t for this step */
  const float* found_inf;        /* optional [1] (GradScaler): the whole update is skipped when != 0 */
  const float* grad_scale;       /* optional [1] (GradScaler): gradients are divided by it on load (no unscale pass) */
  /* optional device step counter (round 4): with step_in the bias corrections are 1 - beta^(t) for t = *step_in + 1 computed
   * in the kernel (bias1 / bias2 are ignored), and *step_out = t - or *step_in when the step is skipped on found_inf, as
   * torch's fused optimizers do not count a skipped step.  step_in != step_out (the host alternates two scalars).            */
  const float* step_in;
  float* step_out;
  double beta1_d, beta2_d;       /* the betas in double: the in-kernel bias corrections are 1 - pow(beta_d, t) in double, as the host
                                    arithmetic of torch.optim.AdamW (0 = take the fp32 betas)                                  */
} mds_adamw_args;
#define MDS_OPT_CHUNK 4096
int mds_multi_adamw(const mds_adamw_args* a, mds_stream_t stream);

/* Multi-tensor SGD with momentum / Nesterov (torch.optim.SGD semantics) - the optimizer of the long-sequence
 * fine-tune, configs/ball_action/ball_finetune_long_004.py:51-55 ("SGD", momentum 0.9, nesterov).  Per element
 *   g = grad / grad_scale + weight_decay * p;  buf = first ? g : momentum * buf + (1 - dampening) * g;
 *   p -= lr * (nesterov ? g + momentum * buf : buf)          (momentum == 0: p -= lr * g, no buffer access)
 * Same tables as mds_multi_adamw (soff addresses the momentum buffer).                                     */
typedef struct {
  const mds_opt_tensor* table;
  const int* chunks;
  int nchunks;
  const float* gbase;
  float* momentum_buf;           /* flat; may be NULL when momentum == 0 */
  float lr, momentum, dampening, weight_decay;
  int nesterov;
  int first;                     /* 1 on the first step of the group: buf = g (torch initialises the buffer with the gradient) */
  const float* found_inf;        /* optional [1]: skip the update when != 0 */
  const float* grad_scale;       /* optional [1]: gradients are divided by it on load */
  const float* step_in;          /* optional device step counter as in mds_adamw_args: `first` is then *step_in == 0 */
  float* step_out;
} mds_sgd_args;
int mds_multi_sgd(const mds_sgd_args* a, mds_stream_t stream);

/* Multi-tensor exponential moving average: ema = decay*ema + (1-decay)*model for every float entry of
 * the state_dict (src/ema.py:47-55); table entries: p = ema tensor, goff = offset of the model tensor
 * from `gbase` in ELEMENTS (gbase may be NULL with goff = absolute address / 4).                      */
typedef struct {
  const mds_opt_tensor* table;
  const int* chunks;
  int nchunks;
  const float* gbase;
  float decay;
} mds_ema_args;
int mds_multi_ema(const mds_ema_args* a, mds_stream_t stream);

/* ---- the training augmentations as fused passes over the (B, T, H, W) fp32 frame batch
 * (src/ball_action/augmentations.py:7-22 = src/action/augmentations.py:7-22, src/augmentations.py:42-78, call site
 * src/argus_models.py:49-53).  The geometric stages (camera move, rotation, resized crop, horizontal flip) are composed on
 * the host into ONE destination -> source affine map per frame; a pass is one launch over the whole batch in which every
 * sample does what its job entry says:
 *   COPY  dst = src                        WARP  dst(x, y) = bilinear src(map(x, y)), zeros outside (grid_sample,
 *   SHARP kornia sharpness (3x3 interior         align_corners=True)
 *         smoothing blended with the input)TAPS  dst = sum_k w_k src(x + dx_k, y + dy_k), zeros outside (motion blur)
 * followed - in the sample's LAST pass (`point`) - by the point operations in the reference's order: brightness (additive,
 * clamped), contrast (multiplicative, clamped), posterize (bit mask of the 8-bit value), Gaussian noise (from `noise`, or
 * generated in the kernel by Philox-4x32-10 + Box-Muller keyed by the job's seed and the element index).
 * A sample without spatial filters is read once and written once; sharpness / motion blur add one pass each through
 * scratch buffers for the samples that drew them (p = 0.2 each).                                                      */
#define MDS_AUG_COPY 0
#define MDS_AUG_WARP 1
#define MDS_AUG_SHARP 2
#define MDS_AUG_TAPS 3
#define MDS_AUG_MAX_TAPS 48
typedef struct {
  int active;                 /* 0: nothing to do for this sample in this pass */
  int src, dst;               /* indices into mds_aug_args.buf */
  int mode;                   /* MDS_AUG_* */
  int point;                  /* 1: point operations + final store (the sample's last pass) */
  float sharp_factor;         /* SHARP: out = blurred + (in - blurred) * factor */
  int ntaps;                  /* TAPS */
  int bright_on;   float bright_add;
  int contrast_on; float contrast_mul;
  int posterize_bits;         /* 0 = off; 1..7: keep that many high bits */
  int noise_on;    float noise_std, noise_mean;
  int noise_seed;
  int tap_dx[MDS_AUG_MAX_TAPS];
  int tap_dy[MDS_AUG_MAX_TAPS];
  float tap_w[MDS_AUG_MAX_TAPS];
} mds_aug_job;
typedef struct {
  int B, T, H, W;
  float* buf[4];              /* 0 = input batch (read only), 1 = output batch, 2 / 3 = scratch (only if a job names them) */
  const mds_aug_job* jobs;    /* device [B] */
  const float* maps;          /* device [B][T][6]: sx = m0 x + m1 y + m2, sy = m3 x + m4 y + m5 (WARP jobs) */
  const float* noise;         /* optional device (B, T, H, W) standard-normal draws; NULL = generate */
} mds_aug_args;
int mds_aug_pass(const mds_aug_args* a, mds_stream_t stream);

/* ---- SURVEY 8(f) N1 (predictor glue): row gather / scatter between the predictor's device rings and the plans' buffers -
 * raw frames ring -> the stem's uint8 input, 2D features -> the feature store, the store's five stacks -> the tail's
 * input.  The reference does this with a Python dict of tensors + torch.stack / torch.cat per frame
 * (src/predictors.py:58-68); index tensors on the device would cost a host-to-device copy per frame, so the slot
 * numbers travel in the kernel arguments.  dst row r = dst + dst_slot[r] * dst_pitch  <-  src + src_slot[r] * src_pitch. */
#define MDS_COPY_ROWS_MAX 320
typedef struct {
  void* dst;
  const void* src;
  long dst_pitch, src_pitch;   /* bytes between consecutive slots */
  long row_bytes;              /* bytes copied per row (16-byte vectors when bases, pitches and row_bytes allow) */
  int nrows;                   /* <= MDS_COPY_ROWS_MAX */
  int dst_slot[MDS_COPY_ROWS_MAX];
  int src_slot[MDS_COPY_ROWS_MAX];
} mds_copy_rows_args;
int mds_copy_rows(const mds_copy_rows_args* a, mds_stream_t stream);

/* ---- SURVEY 8(f) N4 (device side): the luma plane of a decoded NV12 surface -> a contiguous (H, W) uint8 frame.
 * What `NvDecFrameFetcher._convert` does with PySurfaceConverter(NV12 -> Y) + makefromDevicePtrUint8 + resize_
 * (src/frame_fetchers/nvdec.py:44-55): the decoder (rocDecode on AMD) hands over a pitched surface whose first `height`
 * rows are the Y plane; the frames the model sees are those bytes (MPEG range, no conversion).  `count` surfaces of one
 * shape go to consecutive frames of dst (a clip for fetch_frames, or slots of the predictor's frame ring).        */
typedef struct {
  int width, height;
  long pitch;                 /* bytes between rows of a surface */
  int count;
  const unsigned char* src;   /* first surface */
  long surface_stride;        /* bytes between consecutive surfaces (ignored for count == 1) */
  unsigned char* dst;         /* [count][height][width] */
} mds_frame_luma_args;
int mds_frame_luma(const mds_frame_luma_args* a, mds_stream_t stream);

/* ---- parameter packing: fp32 PyTorch parameters -> the layouts/dtypes the kernels read.
 * One launch handles a device-resident table of jobs.                                          */
#define MDS_PACK_OI 0      /* [O][I][taps] -> [O][taps][I]   (1x1: taps = 1; conv fwd)         */
#define MDS_PACK_IO_FLIP 1 /* [O][I][taps] -> [I][taps-1-t][O] (data-gradient pack)            */
#define MDS_PACK_STEM 2    /* [O][3][3][3] -> [O][32] zero padded                              */
#define MDS_PACK_IO_F32 3  /* [O][I] -> [I][O], kept in fp32 whatever `dtype` (squeeze-excite w2) */
/* MFMA-fragment order of a 1x1 filter w[N][K] (mds_pw_fwd_args.w_frag): element (n, k) at
 *   ((k / 32 * ceil(N / 16) + n / 16) * 64 + (n % 16) + 16 * (k % 32 / 8)) * 8 + k % 8, zero beyond N or K;
 * ceil(K / 32) * ceil(N / 16) * 512 elements.                                                                  */
#define MDS_PACK_FRAG_OI 4 /* [O][I] -> fragment order of w[N = O][K = I] (forward)                */
#define MDS_PACK_FRAG_IO 5 /* [O][I] -> fragment order of w[N = I][K = O] (data gradient)          */
typedef struct {
  const float* src;
  void* dst;
  int kind, O, I, taps;
} mds_pack_job;
int mds_pack_weights(const mds_pack_job* jobs_dev, int njobs, int max_elems, int dtype,
                     mds_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
