// k_pwd.hip — data gradient of the 1x1 EXPANSION convolutions with the BatchNorm-backward apply pass folded in
// (include/mds.h: mds_pw_dgrad).
//
//   dy[m][k] = A[k] g[m][k] + B[k] y[m][k] + D[k]          (BatchNorm backward, mds_dyp_t; stored to dy_out for the weight gradient)
//   dx[m][n] = sum_k dy[m][k] w[n][k] (+ residual[m][n])    K = mid (192 .. 1152, the wide side), N = cin <= 192
//   (+ the sums of the NEXT BatchNorm backward over dx, mds_poststat_t)
//
// Reference: the conv_pw half of timm's InvertedResidual / multidim_stacker.py:124-134 InvertedResidual3d backward -
// native_batch_norm_backward + convolution_backward (input gradient).  Before this kernel the dependent chain ran
// mds_bn_bwd_apply (g, y -> dy: three wide passes) and then mds_pw_fwd (dy -> dx), whose 128-column tiles read dy TWICE for
// N = 192 and whose 64-row tile count (288 at 18 400 rows) left it at 1.5 TB/s.  Here a block owns 64 rows and ALL N output
// columns and streams K in 64-channel chunks: every wide element is loaded once, dy is formed in registers, stored, staged in
// LDS and multiplied against the chunk's filter columns (L2-resident, 24 KB per chunk).  Roofline: HBM - the apply pass's
// three wide tensors + two narrow ones; MFMA work 24 per wave and chunk.
//
// Wave roles (see k_bwg.hip: vmcnt retires in order): waves 0-3 stream the wide pair with a ring of NS chunks requested
// ahead, form / store / stage dy; waves 4-7 fetch and stage the filter chunk two chunks ahead; all eight multiply
// (wave = (two of the four row fragments, every fourth column fragment)).
#include "gemm.h"

#define PWD_BM 64
#define PWD_KC 64

template <typename T> struct PwdCfg;
template <> struct PwdCfg<bf16_t> { static const int LD = 80; };    // 160 B = 32 B x 5
template <> struct PwdCfg<float> { static const int LD = 72; };     // 288 B = 32 B x 9

MDS_DEV void pwd_raw8(RawV8<bf16_t>& o, const float (&v)[8]) { o.v = pack8(v); }
MDS_DEV void pwd_raw8(RawV8<float>& o, const float (&v)[8]) { o.a = (f32x4){v[0], v[1], v[2], v[3]}; o.b = (f32x4){v[4], v[5], v[6], v[7]}; }

template <typename T, int N, bool DYP, bool POST>
__global__ __launch_bounds__(512, sizeof(T) == 2 ? 2 : 1) void pw_dgrad_kernel(mds_pw_dgrad_args a, int dbg) {
  constexpr int LD = PwdCfg<T>::LD, NB = (N + 31) / 32 * 32, NP = NB / 32, NFR = N / 16, KFW = (NFR + 3) / 4;
  constexpr int NS = sizeof(T) == 2 ? 3 : 2;         // wide ring depth
  constexpr int OP = N + 4;                          // fp32 output staging pitch
  constexpr int NV = N / 8, RW = 512 / NV, RP = (PWD_BM + RW - 1) / RW;   // epilogue: NV threads per row, RW rows per pass
  typedef typename Frag<T>::type frag_t;
  MDS_DYN_SMEM(smem);
  const int K = a.K, Kp = (K + PWD_KC - 1) / PWD_KC * PWD_KC, nkc = Kp / PWD_KC;
  T* ws = (T*)smem;                              // [64][LD]   dy chunk
  T* bs = ws + PWD_BM * LD;                      // [NB][LD]   filter chunk
  float* tab = (float*)(bs + NB * LD);           // [3][Kp]    A, B, D (DYP)
  float* os = (float*)smem;                      // [64][OP]   fp32 output tile (after the K loop, over everything above)
  const size_t loop_bytes = (size_t)(PWD_BM + NB) * LD * sizeof(T) + (DYP ? (size_t)3 * Kp * 4 : 0);
  const size_t out_bytes = (size_t)PWD_BM * OP * 4;
  float* psum = (float*)(smem + (loop_bytes > out_bytes ? loop_bytes : out_bytes));   // [2][N] (POST)
  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6), role = MDS_UNIFORM(tid >> 8), t = tid & 255;
  const int i = lane & 15, q = lane >> 4;
  const long m0 = (long)blockIdx.x * PWD_BM;
  const long M = a.M;

  if (DYP) {
    for (int e = tid; e < 3 * Kp; e += 512) {
      const int r = e / Kp, c = e - r * Kp;
      tab[e] = c < K ? a.dyp.lin[(long)r * K + c] : 0.f;
    }
  }
  if (POST) {
    for (int e = tid; e < 2 * N; e += 512) psum[e] = 0.f;
  }
  // epilogue operands of this thread (fixed 8-column group): requested here, ahead of the K loop
  const int ecg = tid % NV, erow = tid / NV;
  const bool eact = erow < RW;
  float pmu[8], prs[8];
  if (POST && eact) { load8f(a.post.bn + 2 * N + 8 * ecg, pmu); load8f(a.post.bn + 3 * N + 8 * ecg, prs); }
  __syncthreads();

  const int wm = wave & 1, wn = (wave >> 1) & 3;     // row fragments 2wm, 2wm + 1; column fragments wn + 4j
  f32x4 acc[2][KFW];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < KFW; ++v) acc[u][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto mfma_chunk = [&]() {
    if (dbg & 1) return;
#pragma unroll
    for (int ks = 0; ks < PWD_KC / 32; ++ks) {
      frag_t wf[KFW], df[2];
#pragma unroll
      for (int j = 0; j < KFW; ++j) {
        const int nf = wn + 4 * j;
        wf[j] = ld_frag(bs + (16 * ((NFR % 4 == 0 || nf < NFR) ? nf : 0) + i) * LD + 32 * ks + 8 * q);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) df[u] = ld_frag(ws + (16 * (2 * wm + u) + i) * LD + 32 * ks + 8 * q);
#pragma unroll
      for (int j = 0; j < KFW; ++j) {
        if (NFR % 4 == 0 || wn + 4 * j < NFR) {     // wave-uniform
#pragma unroll
          for (int u = 0; u < 2; ++u) mma16(wf[j], df[u], acc[u][j]);   // acc[r] = dx[row 16(2wm+u) + i][col 16 nf + 4q + r]
        }
      }
    }
  };

  const int vec = t & 7, r0 = t >> 3;                // staging: 16-byte vector of the chunk, first row (both roles)
  if (role == 0) {
    // ------------------------------------------------------------ WIDE waves: rows r0, r0 + 32 of the tile
    const T* src = (const T*)(DYP ? a.dyp.g.u : a.x);
    const T* ysrc = (const T*)a.dyp.y;
    T* dyo = (T*)a.dy_out;
    long rowoff[2];
    bool rok[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const long m = m0 + r0 + 32 * p;
      rok[p] = m < M;
      rowoff[p] = (rok[p] ? m : M - 1) * K;
    }
    struct Wide { RawV8<T> g[2], y[DYP ? 2 : 1]; };
    int req = 0;
    auto issue = [&](Wide& R) {
      const int kc = req < nkc ? req : nkc - 1;
      ++req;
      const int c = PWD_KC * kc + 8 * vec, cl = c < K ? c : K - 8;
      if (dbg & 4) return;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        R.g[p].ld(src + rowoff[p] + cl);
        if (DYP) R.y[p].ld(ysrc + rowoff[p] + cl);
      }
    };
    int kc_cur = 0;
    auto step = [&](Wide& R) {
      const int c = PWD_KC * kc_cur + 8 * vec;
      const bool cok = c < K;
      __syncthreads();     // the previous chunk's fragment reads are done
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        T* lp = ws + (r0 + 32 * p) * LD + 8 * vec;
        if (DYP) {
          float u[8], yv[8], dv[8], cA[8], cB[8], cD[8];
          R.g[p].get(u); R.y[p].get(yv);
          const float* tp = tab + (cok ? c : 0);
          load8f(tp, cA); load8f(tp + Kp, cB); load8f(tp + 2 * Kp, cD);
          const bool ok = cok && rok[p];
#pragma unroll
          for (int j = 0; j < 8; ++j) dv[j] = ok ? cA[j] * u[j] + cB[j] * yv[j] + cD[j] : 0.f;
          RawV8<T> o;
          pwd_raw8(o, dv);
          if (dyo && ok && !(dbg & 2)) o.st(dyo + rowoff[p] + c);
          o.st(lp);
        } else {
          if (!(cok && rok[p])) R.g[p].zero();
          R.g[p].st(lp);
        }
      }
      issue(R);            // this set's registers are free again: request chunk kc + NS
      ++kc_cur;
      __syncthreads();
      mfma_chunk();
    };
    Wide W[NS];
#pragma unroll
    for (int r = 0; r < NS; ++r) issue(W[r]);
    int s = 0;
    for (; s + NS <= nkc; s += NS) {
#pragma unroll
      for (int r = 0; r < NS; ++r) step(W[r]);
    }
#pragma unroll
    for (int r = 0; r < NS - 1; ++r)
      if (s + r < nkc) step(W[r]);
  } else {
    // ------------------------------------------------------------ FILTER waves: rows r0 + 32p of the [N][K] filter
    const T* wsrc = (const T*)a.w;
    struct Fil { RawV8<T> v[NP]; };
    int req = 0;
    auto issue = [&](Fil& R) {
      const int kc = req < nkc ? req : nkc - 1;
      ++req;
      const int c = PWD_KC * kc + 8 * vec, cl = c < K ? c : K - 8;
      if (dbg & 8) return;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int n = r0 + 32 * p;
        R.v[p].ld(wsrc + (long)(n < N ? n : N - 1) * K + cl);
      }
    };
    int kc_cur = 0;
    auto step = [&](Fil& R) {
      const bool cok = PWD_KC * kc_cur + 8 * vec < K;
      __syncthreads();
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int n = r0 + 32 * p;
        if (!(cok && n < N)) R.v[p].zero();
        R.v[p].st(bs + n * LD + 8 * vec);
      }
      issue(R);            // two chunks ahead
      ++kc_cur;
      __syncthreads();
      mfma_chunk();
    };
    Fil F[2];
    issue(F[0]); issue(F[1]);
    int s = 0;
    for (; s + 2 <= nkc; s += 2) { step(F[0]); step(F[1]); }
    if (s < nkc) step(F[0]);
  }

  // ---------------------------------------------------------------- epilogue: fp32 tile through LDS, then row-major
  if (dbg & 16) return;
  __syncthreads();       // the last chunk's fragment reads are done: the staging buffers are free
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int j = 0; j < KFW; ++j) {
      const int nf = wn + 4 * j;
      if (NFR % 4 == 0 || nf < NFR) *(f32x4*)(os + (16 * (2 * wm + u) + i) * OP + 16 * nf + 4 * q) = acc[u][j];
    }
  __syncthreads();
  if (eact) {
    float sg[8], sgx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sg[j] = 0.f; sgx[j] = 0.f; }
    T* yout = (T*)a.y;
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const int row = erow + RW * p;
      const long m = m0 + row;
      if (row < PWD_BM && m < M) {
        float v[8];
        load8f(os + row * OP + 8 * ecg, v);
        if (a.residual) {
          float rr[8];
          load8((const T*)a.residual + m * N + 8 * ecg, rr);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += rr[j];
        }
        if (POST) {
          float ys[8];
          load8((const T*)a.post.y + m * N + 8 * ecg, ys);
          const float mk = (a.post.mode == MDS_POST_MASK) ? a.post.mask[(unsigned)m / (unsigned)a.post.rows_per_group] : 1.0f;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float g = Elem<T>::rnd(v[j]) * mk;      // the sums see what later readers will read
            sg[j] += g;
            sgx[j] += g * ((ys[j] - pmu[j]) * prs[j]);
          }
        }
        store8(yout + m * N + 8 * ecg, v);
      }
    }
    if (POST) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { atomicAdd(psum + 8 * ecg + j, sg[j]); atomicAdd(psum + N + 8 * ecg + j, sgx[j]); }
    }
  }
  if (POST) {
    __syncthreads();
    if (tid < 2 * N) atomicAdd(a.post.stats + (long)(blockIdx.x % MDS_STAT_SLOTS) * 2 * N + tid, (double)psum[tid]);
  }
}

static bool pwd_n_ok(int N) { return N == 48 || N == 96 || N == 112 || N == 192; }
extern "C" int mds_pw_dgrad_ok(long M, int K, int N) { return (M > 0 && M < 2147483647L && K >= 64 && K % 32 == 0 && K <= 2048 && pwd_n_ok(N)) ? 1 : 0; }

extern "C" int mds_pw_dgrad(const mds_pw_dgrad_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && mds_pw_dgrad_ok(a->M, a->K, a->N), "pw_dgrad: bad dims (K a multiple of 32, 64 .. 2048; N in 48, 96, 112, 192)");
  const bool dyp = a->dyp.mode != 0;
  MDS_REQUIRE((dyp || a->x) && a->w && a->y, "pw_dgrad: null pointer");
  if (dyp) {
    MDS_REQUIRE(a->dyp.g.u && a->dyp.y && a->dyp.lin && a->dyp.g.mode == MDS_G_PLAIN, "pw_dgrad: the dy prologue needs u, y, lin and a PLAIN gradient source");
  } else {
    MDS_REQUIRE(!a->dy_out, "pw_dgrad: dy_out goes with the dy prologue");
  }
  const bool post = a->post.mode != MDS_POST_NONE;
  if (post) {
    MDS_REQUIRE(a->post.y && a->post.bn && a->post.stats && (a->post.mode == MDS_POST_PLAIN || a->post.mode == MDS_POST_MASK),
                "pw_dgrad: post statistics need y, bn, stats and the PLAIN or MASK form");
    MDS_REQUIRE(a->post.mode != MDS_POST_MASK || (a->post.mask && a->post.rows_per_group > 0), "pw_dgrad: post mask");
  }
  const int Kp = (a->K + PWD_KC - 1) / PWD_KC * PWD_KC;
  const dim3 grid((unsigned)cdiv(a->M, PWD_BM)), block(512);
#define PWD_GO(T, N_, DYP_, POST_)                                                                                         \
  do {                                                                                                                     \
    const size_t loop_ = (size_t)(PWD_BM + (N_ + 31) / 32 * 32) * PwdCfg<T>::LD * sizeof(T) + (DYP_ ? (size_t)3 * Kp * 4 : 0); \
    const size_t out_ = (size_t)PWD_BM * (N_ + 4) * 4;                                                                     \
    MDS_LAUNCH((pw_dgrad_kernel<T, N_, DYP_, POST_>), grid, block, (loop_ > out_ ? loop_ : out_) + 2 * N_ * 4, stream, *a, mds_knob(MDS_KNOB_WG_DBG)); \
  } while (0)
#define PWD_FLAGS(T, N_)                                                                  \
  do {                                                                                    \
    if (dyp) { if (post) PWD_GO(T, N_, true, true); else PWD_GO(T, N_, true, false); }    \
    else { if (post) PWD_GO(T, N_, false, true); else PWD_GO(T, N_, false, false); }      \
  } while (0)
  MDS_DISPATCH_DTYPE(a->dtype, T, do {
    switch (a->N) {
      case 48: PWD_FLAGS(T, 48); break;
      case 96: PWD_FLAGS(T, 96); break;
      case 112: PWD_FLAGS(T, 112); break;
      default: PWD_FLAGS(T, 192); break;
    }
  } while (0));
  return mds_check_launch("pw_dgrad");
}
