// k_c3.hip — stride-1 3x3 convolutions (forward and stride-1 data gradient) with the FILTER IN REGISTERS:
// a row-streaming implicit GEMM for gfx950 (round 6).
//
// The persistent kernel of k_conv.hip keeps the filter slab in LDS and reads one operand fragment per MFMA operand: at
// 0.5 ds_read_b128 per MFMA its four waves ask the LDS for half of what it can deliver, stage every patch through VGPRs
// (13-cycle ds_write_b128) behind two block barriers per tile, and reach 13-20 % of the MFMA peak on layers whose HBM floor
// is 2-3x below their time.  These layers have tiny filters (9 x Cin x Cout <= 74 KB) and millions of pixels, so here
//
//   * every CONSUMER wave (4 per block, one per SIMD) keeps its slice of the filter in VGPRs for the whole launch
//     (<= 180 registers) - weight fragments cost no LDS read at all;
//   * a wave owns a 16-pixel-wide column strip of a band and walks DOWN the image: for one input row it reads each
//     (dx, channel) fragment ONCE from LDS and issues the MFMAs of all three dy taps into three ROLLING accumulator sets
//     (output rows r-1, r, r+1) - one ds_read_b128 per 3 x NF MFMAs (NF = 16-channel output fragments per wave), i.e.
//     5-17 % of the LDS read bandwidth; an output row is finished, converted and stored when its third input row is done;
//   * the input rows (+ one halo pixel either side, + the residual operand's row) come through a ring of rows in LDS
//     filled by LDS-DMA (global_load_lds_dwordx4) from dedicated PRODUCER waves, RA rows ahead of the consumers and
//     straight across work-item boundaries: no staging registers, no ds_write, one bare s_barrier per row.  The producers'
//     vmcnt counts nothing but their own DMA, so the counted wait is exact; the consumers' vmcnt carries only their
//     output stores, which nothing ever waits for;
//   * the 16-byte parts of a pixel are rotated by a function of the pixel index on the way in (the DMA lane picks its
//     SOURCE address; the LDS image stays lane-linear), which makes the shifted fragment reads bank-conflict free
//     (tools/probes/c3_swizzle_check.py);
//   * out-of-image pixels and rows are DMA'd from a zero page, so every row of every item issues the same number of
//     pieces (exact vmcnt) and the inner loop has no masks.
//
// Work item = (image, band of WB columns, segment of rows); a block walks items blockIdx.x, + gridDim.x, ...
// bf16 only (fp32 plans keep k_conv.hip); prologue-free inputs only (activated / materialised tensors).
#include <stdlib.h>
#include <type_traits>
#include "gemm.h"

struct C3Args {
  const bf16_t* x;
  const bf16_t* w;     // [Cout][wtaps][Cin]
  bf16_t* y;
  const bf16_t* res;   // optional, indexed like y
  double* stats;       // optional [SLOTS][2][Cout]
  int N, H, W, wtaps;
  int OH, OW;          // output extents (stride-2 data gradient: 2 H x 2 W; otherwise H x W)
  int Ctot;            // channels of the output tensor: a launch computes COUT of them per grid.y slice ("channel pass")
  int tapw[9];         // weight slot of tap (dy, dx) at [3 * (dy + 1) + (dx + 1)]
  int nbands, nseg, rps, items;
  int RA, NR;
  int dbg;             // MDS_KNOB_C3_DBG bits
  // post statistics (mds_poststat_t; PLAIN, or MASK with one value per image): sum g, sum g * xhat of the BatchNorm below
  const bf16_t* py; const float* pbn; const float* pmask; double* pstats;
  const float* pro_scale; const float* pro_shift;      // NTW > 0: the input is read through BatchNorm + SiLU (mds_pro_t BN_SILU)
  int pmode;           // MDS_POST_*: SILU also replaces the stored u by g = u * silu'(y * scale + shift)
  void* trace;         // C3_TRACE builds: 160 x 8 x 4 cycle stamps of one block (passed in mds_conv_fwd_args.epi.scale, mode NONE)
};

__device__ __attribute__((aligned(256))) unsigned int c3_zero_page[64];

template <int CIN> struct C3Swz;     // part' = (part + ((A * pixel) >> SH)) % (CIN / 8): tools/probes/c3_swizzle_check.py search
template <> struct C3Swz<16> { static constexpr int A = 0, SH = 0; };
template <> struct C3Swz<32> { static constexpr int A = 1, SH = 1; };
template <> struct C3Swz<48> { static constexpr int A = 0, SH = 0; };
template <> struct C3Swz<64> { static constexpr int A = 1, SH = 0; };
template <> struct C3Swz<128> { static constexpr int A = 2, SH = 0; };
template <> struct C3Swz<192> { static constexpr int A = 1, SH = 0; };

template <int CIN, int NF, int NSPL, int SPW, bool RES, bool POST = false> struct C3Cfg {
  static constexpr int PP = CIN / 8;                 // 16-byte parts per input pixel
  static constexpr int KSR = (3 * PP + 3) / 4;       // k-steps (32 channels) of one input row: 3 dx x CIN, flattened
  static constexpr int NSG = 4 / NSPL;               // strip groups among the four consumer waves
  static constexpr int WB = 16 * NSG * SPW;          // band width (output columns)
  static constexpr int COUT = 16 * NF * NSPL;
  static constexpr int CP = COUT / 8;                // 16-byte parts per output pixel (residual operand)
  static constexpr int RPX = (WB + 2) * PP;          // 16-byte slots of one input row of the band (+ halo)
  static constexpr int RPXP = RES ? (RPX + 63) / 64 * 64 : RPX;   // the residual row starts on a DMA-piece boundary: a piece is all input or all residual
  static constexpr int RPYP = (RPXP + (RES ? WB * CP : 0) + (POST ? 63 : 0)) / (POST ? 64 : 1) * (POST ? 64 : 1);   // post.y's row of the band: on a piece boundary too
  static constexpr int RS = RPYP + (POST ? WB * CP : 0);
  static constexpr int PIECES = (RS + 63) / 64;      // LDS-DMA instructions per row
  static constexpr int ROWB = RS * 16;
};

struct C3Item {
  int n, x0, r0, r1;
};
MDS_DEV C3Item c3_item(const C3Args& g, int it, int WB) {
  const int seg = it % g.nseg, t = it / g.nseg;
  const int band = t % g.nbands, n = t / g.nbands;
  C3Item r;
  r.n = n; r.x0 = band * WB; r.r0 = seg * g.rps;
  r.r1 = r.r0 + g.rps < g.H ? r.r0 + g.rps : g.H;
  return r;
}

MDS_DEV C3Item c3_item_h(int nseg, int nbands, int rps, int H, int it, int WB) {
  const int seg = it % nseg, t = it / nseg;
  const int band = t % nbands, n = t / nbands;
  C3Item r;
  r.n = n; r.x0 = band * WB; r.r0 = seg * rps;
  r.r1 = r.r0 + rps < H ? r.r0 + rps : H;
  return r;
}

template <int NF> struct C3Out;
template <> struct C3Out<1> {
  static MDS_DEV void st(bf16_t* p, const float (&v)[4]) { store4(p, v); }
};
template <> struct C3Out<2> {
  static MDS_DEV void st(bf16_t* p, const float (&v)[8]) { store8(p, v); }
};
template <> struct C3Out<3> {
  static MDS_DEV void st(bf16_t* p, const float (&v)[12]) {
    float a[4] = {v[0], v[1], v[2], v[3]}, b[4] = {v[4], v[5], v[6], v[7]}, c[4] = {v[8], v[9], v[10], v[11]};
    store4(p, a); store4(p + 4, b); store4(p + 8, c);
  }
};

#ifndef C3_ABL
#define C3_ABL 0   /* experiment builds (make c3abl ABL=n): 1 no output stores, 2 no statistics, 4 fragment reads replaced by a register, 8 no MFMAs, 16 transform waves copy, 32 transform waves idle */
#endif
#ifdef C3_TRACE   /* experiment build (make c3trace): cycle stamps of block C3_TRACE, [batch][wave][phase] in LDS behind the ring, dumped through g.trace */
#define C3_STAMP(b_, ph_) do { if (trc && lane == 0 && (b_) < 160) { const unsigned long long t_ = __builtin_readcyclecounter(); \
    asm volatile("ds_write_b64 %0, %1" ::"v"((lds_t)(trc_base + (((b_) * 8 + wave) * 4 + (ph_)) * 8)), "v"(t_) : "memory"); } } while (0)
#else
#define C3_STAMP(b_, ph_) ((void)0)
#endif

template <int CIN, int NF, int NSPL, int SPW, int NPW, int NSW, bool RES, bool STATS, bool MASKED, bool POST = false, bool ONE = false, int NTW = 0>
__global__ __launch_bounds__(256 + 64 * (NPW + NSW + NTW)) void c3_kernel(C3Args g) {
  // NTW > 0: the input tensor is the RAW output of the layer below and is read through its BatchNorm + SiLU (the first 3x3 layer behind
  // the stem).  LDS-DMA cannot transform, so NTW TRANSFORM waves rewrite every landed row in place - silu(scale * v + shift), zero
  // outside the image (the padding is applied after the activation) - one batch ahead of the consumers, which therefore run one
  // barrier behind (LAG); the sigmoid is this layer's whole cost (150 M evaluations), spread here over NTW waves beside the MFMA waves.
  constexpr int LAG = NTW > 0 ? 1 : 0;
  // ONE: a 1x1 convolution through the same machinery (mds_pw_fwd's large prologue-free launches: the edge-residual projections'
  // data gradients, 0.3 GB of output each): only the centre tap exists - one MFMA in nine - and the rows of the 'image' are just
  // consecutive runs of W pixels.
  typedef C3Cfg<CIN, NF, NSPL, SPW, RES, POST> CF;
  constexpr int RPYP = CF::RPYP;
  static_assert(!(POST && STATS), "forward statistics and post statistics never meet");
  constexpr int PP = CF::PP, KSR = CF::KSR, WB = CF::WB, COUT = CF::COUT, CP = CF::CP, RPX = CF::RPX, RPXP = CF::RPXP, RS = CF::RS;
  constexpr int PIECES = CF::PIECES, ROWB = CF::ROWB;
  constexpr int CPP = COUT / 8;                      // 16-byte chunks of an output pixel
  constexpr int STGROW = WB * COUT * 2;              // bytes of one staged output row of the band
  constexpr int RSH = CPP == 4 ? 1 : 0;              // staged chunk c of pixel p sits at chunk (c + (p >> RSH)) % CPP: spreads the consumers' writes over the banks
  MDS_DYN_SMEM(smem);
  char* const stg = smem + g.NR * ROWB;               // [2 batch parities][3 rows][WB pixels][COUT] bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);
  const int G = gridDim.x;
  const int c0 = blockIdx.y * COUT;                  // first output channel of this channel pass
  // entries (rows to stage and to consume) of this block: every item has (rows + 2)
  int E = 0;
  for (int it = blockIdx.x; it < g.items; it += G) {
    const C3Item im = c3_item(g, it, WB);
    E += im.r1 - im.r0 + 2;
  }

#ifdef C3_TRACE
  const bool trc = blockIdx.x == C3_TRACE && g.trace != nullptr;
  const lds_t trc_base = lds_addr_of(smem) + (lds_t)(g.NR * ROWB + 6 * STGROW);
  int tb = 0;
#endif
  if (wave >= 4 && wave < 4 + NPW) {
    // ------------------------------------------------------------------ producers: LDS-DMA, RA rows ahead
    MDS_SETPRIO(3);       // the DMA stream is what every barrier waits for
    const int pw = wave - 4;
    constexpr int PCWMAX = (PIECES + NPW - 1) / NPW;
    const int pcw = (PIECES - pw + NPW - 1) / NPW;        // pieces this wave issues per row
    // what this lane moves in each of its pieces: constant over the launch (a piece = 64 consecutive 16-byte slots of a ring row)
    int dcol[PCWMAX], eoff[PCWMAX];          // column - x0 (INT_MIN/2: nothing: the gap in front of the residual row / past the end); element offset inside the pixel
#pragma unroll
    for (int j = 0; j < PCWMAX; ++j) {
      const int sg_ = 64 * (pw + NPW * j) + lane;
      if (sg_ < RPX) {
        const int p = sg_ / PP, psw = sg_ - p * PP;
        const int rot = ((C3Swz<CIN>::A * p) >> C3Swz<CIN>::SH) % PP;
        dcol[j] = p - 1; eoff[j] = 8 * ((psw - rot + PP) % PP);
      } else if (RES && sg_ >= RPXP && sg_ < RPXP + WB * CP) {
        const int t = sg_ - RPXP, p = t / CP;
        dcol[j] = p; eoff[j] = 8 * (t - p * CP);
      } else if (POST && sg_ >= RPYP && sg_ < RS) {
        const int t = sg_ - RPYP, p = t / CP;
        dcol[j] = p; eoff[j] = 8 * (t - p * CP);
      } else {
        dcol[j] = -(1 << 30); eoff[j] = 0;
      }
    }
    const lds_t ring = lds_addr_of(smem);
    int hit = blockIdx.x, hk = 0;                           // head of the DMA stream: (item, row index inside it)
    C3Item him = c3_item(g, hit < g.items ? hit : 0, WB);
    int C = 0, hslot = 0;
    // per item and piece: this lane's source address for the NEXT row and what it advances by per row (lanes whose column is
    // outside the image stay on the zero page with step 0) - a piece of a row inside the image is then one 64-bit add, the
    // M0 write and the DMA instruction
    const char* cur[PCWMAX];
    unsigned step[PCWMAX];
    auto open_item = [&]() {
      const char* xrow0 = (const char*)g.x + ((long)him.n * g.H + him.r0 - 1) * g.W * CIN * 2;                   // entry 0's rows:
      const char* rrow0 = RES ? (const char*)(g.res + c0) + ((long)him.n * g.H + him.r0 - 2) * g.W * g.Ctot * 2 : nullptr;   // r0 - 1 / r0 - 2
      const char* yrow0 = POST ? (const char*)(g.py + c0) + ((long)him.n * g.H + him.r0 - 2) * g.W * g.Ctot * 2 : nullptr;
#pragma unroll
      for (int j = 0; j < PCWMAX; ++j) {
        const bool isy = POST && 64 * (pw + NPW * j) >= RPYP;
        const bool isres = (RES || POST) && 64 * (pw + NPW * j) >= RPXP;           // an output-shaped row (residual or post.y)
        const int gx = him.x0 + dcol[j];
        const bool ok = gx >= 0 && gx < g.W;
        cur[j] = ok ? (isy ? yrow0 : (isres ? rrow0 : xrow0)) + (gx * (isres ? g.Ctot : CIN) + eoff[j]) * 2 : (const char*)c3_zero_page;
        step[j] = ok ? (unsigned)(g.W * (isres ? g.Ctot : CIN) * 2) : 0u;
      }
    };
    open_item();
    auto issue = [&]() {
      const int ri = him.r0 - 1 + hk, ro = ri - 1;
      const bool xok = ri >= 0 && ri < g.H, rok = !(RES || POST) || (ro >= 0 && ro < g.H);
      const lds_t dst = ring + (lds_t)(hslot * ROWB);
      hslot = hslot + 1 == g.NR ? 0 : hslot + 1;
      if (xok && rok) {
#pragma unroll
        for (int j = 0; j < PCWMAX; ++j) {
          const int pi = pw + NPW * j;
          if (pi < PIECES) {
            if (dcol[j] > -(1 << 29)) glds16(cur[j], dst + (lds_t)(pi * 1024));
            cur[j] += step[j];
          }
        }
      } else {          // a row above / below the image (the first and last rows of the items that touch its border): zeros
#pragma unroll
        for (int j = 0; j < PCWMAX; ++j) {
          const int pi = pw + NPW * j;
          if (pi < PIECES) {
            const bool isres = (RES || POST) && 64 * pi >= RPXP;              // wave-uniform
            const char* src = (isres ? rok : xok) ? cur[j] : (const char*)c3_zero_page;
            if (dcol[j] > -(1 << 29)) glds16(src, dst + (lds_t)(pi * 1024));
            cur[j] += step[j];
          }
        }
      }
      ++C;
      if (++hk == him.r1 - him.r0 + 2) {
        hk = 0; hit += G;
        if (hit < g.items) { him = c3_item(g, hit, WB); open_item(); }
      }
    };
    while (C < g.RA && C < E) issue();
    // the consumers take their rows in batches of three (the rolling accumulators' period) behind ONE barrier: per batch wait
    // for its last row, arrive, then top the stream up to RA rows ahead of the next batch (ring: RA + 3 rows)
    int e0 = 0;
    for (int it = blockIdx.x; it < g.items; it += G) {
      const C3Item im = c3_item(g, it, WB);
      const int K = im.r1 - im.r0 + 2;
      for (int k0 = 0; k0 < K; k0 += 3) {
        const int n = K - k0 < 3 ? K - k0 : 3;
        C3_STAMP(tb, 0);
        wait_vm_dyn(pcw * (C - e0 - n));      // rows e0 .. e0 + n - 1 have landed; the rows issued after them stay in flight
        C3_STAMP(tb, 1);
        raw_barrier();
        C3_STAMP(tb, 2);
        e0 += n;
        while (C < e0 + g.RA && C < E) issue();   // into slots of rows < e0 - n: every consumer is past them
        C3_STAMP(tb, 3);
#ifdef C3_TRACE
        ++tb;
#endif
      }
    }
    raw_barrier();        // the closing barrier: the last batch's outputs are staged
    if (LAG) raw_barrier();
#ifdef C3_TRACE
    raw_barrier();
#endif
    return;
  }
  if (wave >= 4 + NPW + NSW) {
    // ------------------------------------------------------------------ transform waves (NTW > 0)
    MDS_SETPRIO(1);       // below the consumers (prio 2): their short read - MFMA bursts must not queue behind two transform waves per SIMD
    const int tw = wave - 4 - NPW - NSW;
    // The unit of work is HALF a 16-byte slot (four channels): a row's 2 RPX half slots make NPC pieces of 64, dealt round the
    // transform waves row by row - finer pieces balance the four SIMDs' VALU queues (the stage is bound by v_exp / v_rcp issue:
    // 24.5 cycles per wave instruction each, profiles/r06_probe_valu_rate.txt).  Lane l of piece pc holds half (l & 1) of part
    // ((l >> 1) % PP) of pixel (32 pc + (l >> 1)) / PP; with PP | 32 and a rotation that repeats every 32 / PP pixels the
    // CHANNELS a lane transforms are the same in every piece: their scale / shift live in registers.
    static_assert(NTW == 0 || (32 % PP == 0 && (((C3Swz<CIN>::A * (32 / PP)) >> C3Swz<CIN>::SH) % PP) == 0), "lane-constant channels");
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int hl = lane >> 1, lp = hl / PP, lpart = ((hl % PP) - ((C3Swz<CIN>::A * lp) >> C3Swz<CIN>::SH) % PP + PP) % PP;
    f32x2 sc[2], sh[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ch = 8 * lpart + 4 * (lane & 1) + 2 * c;
      sc[c] = (f32x2){g.pro_scale[ch], g.pro_scale[ch + 1]};
      sh[c] = (f32x2){g.pro_shift[ch], g.pro_shift[ch + 1]};
    }
    constexpr int NPC = (2 * RPX + 63) / 64;              // pieces of one row
    int rslot = 0;
    for (int it = blockIdx.x; it < g.items; it += G) {
      const C3Item im = c3_item(g, it, WB);
      const int K = im.r1 - im.r0 + 2;
      for (int k0 = 0; k0 < K; k0 += 3) {
        const int n = K - k0 < 3 ? K - k0 : 3;
        asm volatile("" ::: "memory");
        raw_barrier();                                    // the batch has landed (the DMA waves waited for it before arriving)
        asm volatile("" ::: "memory");
        // (row, piece) pairs of the batch, dealt round the transform waves; a wave's reads of the batch all go out before its
        // first piece's arithmetic (one piece in flight per wave leaves the LDS round trip and the v_exp / v_rcp latencies bare)
        constexpr int MAXP = (3 * NPC + NTW - 1) / (NTW ? NTW : 1);
        u16x4 v[MAXP];
        char* ptr[MAXP];
        int edge[MAXP];                                   // 0 = inside the image, 1 = touches its edge, -1 = no piece
#pragma unroll
        for (int q = 0; q < MAXP; ++q) {
          const int u = tw + q * NTW;
          edge[q] = -1;
          if (u < n * NPC) {
            const int j = u / NPC, pc = u - j * NPC;
            const int ri = im.r0 - 1 + k0 + j, px0 = im.x0 - 1 + (32 / PP) * pc;
            int rs_ = rslot + j;
            if (rs_ >= g.NR) rs_ -= g.NR;
            const int h = 64 * pc + lane;
            ptr[q] = smem + rs_ * ROWB + (h < 2 * RPX ? h : 0) * 8;
            if (!(C3_ABL & 32)) v[q] = *(const u16x4*)ptr[q];
            const bool in = ri >= 0 && ri < g.H && px0 >= 0 && px0 + 32 / PP <= g.W;
            edge[q] = in ? 0 : 1;
          }
        }
#pragma unroll
        for (int q = 0; q < MAXP; ++q) {
          if (edge[q] < 0 || (C3_ABL & 32)) continue;     // wave-uniform
          if (C3_ABL & 16) { *(u16x4*)ptr[q] = v[q]; continue; }
          const int u = tw + q * NTW;
          const int j = u / NPC, pc = u - j * NPC;
          f32x2 z[2], e[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            z[c] = (f32x2){bf2f(v[q][2 * c]), bf2f(v[q][2 * c + 1])} * sc[c] + sh[c];
            const f32x2 t = z[c] * -1.4426950408889634f;
            e[c] = (f32x2){fast_exp2(t[0]), fast_exp2(t[1])} + 1.0f;
            z[c] *= (f32x2){fast_rcp(e[c][0]), fast_rcp(e[c][1])};
          }
          // zero padding AFTER the activation, as a factor and only in the pieces that touch the image's edge (a select per
          // element becomes eight masked branches in hipcc's hands; the zero page's raw 0 gives a finite silu(shift))
          if (edge[q]) {
            const int ri = im.r0 - 1 + k0 + j, gx = im.x0 - 1 + (32 / PP) * pc + lp;
            const float okf = (ri >= 0 && ri < g.H && gx >= 0 && gx < g.W) ? 1.f : 0.f;
            z[0] *= okf; z[1] *= okf;
          }
          typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
          if (64 * pc + lane < 2 * RPX) *(u32x2_*)ptr[q] = (u32x2_){pack2(z[0][0], z[0][1]), pack2(z[1][0], z[1][1])};
        }
        rslot += n;
        if (rslot >= g.NR) rslot -= g.NR;
        wait_lgkm0();                                     // the rewritten rows are in LDS before the next barrier lets the consumers at them
      }
    }
    raw_barrier();
    raw_barrier();
#ifdef C3_TRACE
    raw_barrier();
#endif
    return;
  }
  if (wave >= 4 + NPW) {
    // ------------------------------------------------------------------ store waves: staged output rows -> global memory
    // The consumers own channel slices, so a wave's own store would touch 64-byte pieces of sixteen pixels (measured: 40 of the
    // forward layer's 127 us); they stage the converted rows of a batch in LDS instead, and after the NEXT batch's barrier these
    // waves move them out as whole pixels: 1 KiB of consecutive bytes per wave instruction.  Their vmcnt holds stores only.
    MDS_SETPRIO(3);       // short bursts after each barrier; the consumers' next barrier waits for them
    const int sw = wave - 4 - NPW;
    // a piece = PPP whole pixels: lane l moves chunk l % CPP of pixel l / CPP
    constexpr int PPP = 64 / CPP, NP = (WB + PPP - 1) / PPP;
    static_assert(CPP <= 64, "an output pixel is at most one DMA piece wide");
    const int lpx = lane / CPP, chunk = lane - lpx * CPP;
    const bool lact = lpx < PPP;
    int bidx = 0;
    C3Item pim = c3_item(g, blockIdx.x, WB);
    int pk0 = 0, pn = 0;
    // POST (mds_poststat_t): the rows these waves move out are u of the BatchNorm below.  Its raw input y came through the ring
    // with the batch (with post statistics the ring is three rows longer: a batch's slots stay untouched until the barrier after
    // next), so sum g and sum g * xhat - of the
    // stored, rounded values - cost the consumers nothing.
    int rslot = 0, pslot = 0;                             // ring slot of the next row / of the previous batch's first row
    float ps[8], pss[8], pmu[8], prs[8], psc[8], psh[8];
    const bool psilu = POST && g.pmode == MDS_POST_SILU;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      ps[c] = 0.f; pss[c] = 0.f;
      psc[c] = psilu ? g.pbn[c0 + 8 * chunk + c] : 0.f;
      psh[c] = psilu ? g.pbn[g.Ctot + c0 + 8 * chunk + c] : 0.f;
      pmu[c] = POST ? g.pbn[2 * g.Ctot + c0 + 8 * chunk + c] : 0.f;
      prs[c] = POST ? g.pbn[3 * g.Ctot + c0 + 8 * chunk + c] : 0.f;
    }
    if (LAG) raw_barrier();                               // (the consumers run one barrier behind the ring)
    auto flush = [&](int par) {
      const float pmk = (POST && g.pmask) ? g.pmask[pim.n] : 1.f;      // MASK: DropPath's per-image factor
      for (int j = 0; j < pn; ++j) {
        const int k = pk0 + j, ro = pim.r0 + k - 2;
        if (k < 2 || ro >= pim.r1) continue;              // a row above / below the item: zeros or partial sums, never stored
        const char* srow = stg + (par * 3 + j) * STGROW;
        int rs_ = pslot + j;
        if (rs_ >= g.NR) rs_ -= g.NR;
        const bf16_t* yrow = (const bf16_t*)(smem + rs_ * ROWB + RPYP * 16);
        bf16_t* const orow = g.y + c0 + (((long)pim.n * g.H + ro) * g.W + pim.x0) * g.Ctot;
        for (int pc = sw; pc < NP; pc += NSW) {
          const int cl = pc * PPP + lpx;
          if (lact && cl < WB && (!MASKED || pim.x0 + cl < g.W)) {
            u16x8 v = *(const u16x8*)(srow + (cl * CPP + (chunk + (cl >> RSH)) % CPP) * 16);
            if (POST) {
              const u16x8 yv = *(const u16x8*)(yrow + cl * COUT + 8 * chunk);
              if (psilu) {      // the BatchNorm below sits under a SiLU: its gradient source g = u * silu'(z) replaces u in memory
                float gv[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) gv[c] = bf2f(v[c]) * silu_gradf_(bf2f(yv[c]) * psc[c] + psh[c]);
                v = pack8(gv);
              }
#pragma unroll
              for (int c = 0; c < 8; ++c) {
                const float gg = bf2f(v[c]) * pmk;
                ps[c] += gg;
                pss[c] += gg * ((bf2f(yv[c]) - pmu[c]) * prs[c]);
              }
            }
            *(u16x8*)(orow + cl * g.Ctot + 8 * chunk) = v;
          }
        }
      }
    };
    for (int it = blockIdx.x; it < g.items; it += G) {
      const C3Item im = c3_item(g, it, WB);
      const int K = im.r1 - im.r0 + 2;
      for (int k0 = 0; k0 < K; k0 += 3) {
        asm volatile("" ::: "memory");
        raw_barrier();                                    // batch bidx starts; batch bidx - 1 is staged
        asm volatile("" ::: "memory");
        if (bidx > 0) flush((bidx - 1) & 1);
        pim = im; pk0 = k0; pn = K - k0 < 3 ? K - k0 : 3;
        pslot = rslot; rslot += pn;
        if (rslot >= g.NR) rslot -= g.NR;
        ++bidx;
      }
    }
    asm volatile("" ::: "memory");
    raw_barrier();
    asm volatile("" ::: "memory");
    flush((bidx - 1) & 1);
    if (POST) {      // lanes chunk, chunk + CPP, ... hold the same channels
      double* st = g.pstats + (long)(blockIdx.x % MDS_STAT_SLOTS) * 2 * g.Ctot + c0;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float a = lact ? ps[c] : 0.f, b = lact ? pss[c] : 0.f;
        for (int o = CPP; o < 64; o += CPP) { a += __shfl_down(lact ? ps[c] : 0.f, o); b += __shfl_down(lact ? pss[c] : 0.f, o); }
        if (lane < CPP) {
          atomicAdd(st + 8 * chunk + c, (double)a);
          atomicAdd(st + g.Ctot + 8 * chunk + c, (double)b);
        }
      }
    }
#ifdef C3_TRACE
    raw_barrier();
#endif
    return;
  }

  // -------------------------------------------------------------------- consumers
  MDS_SETPRIO(2);
  const int i = lane & 15, q = lane >> 4;
  const int nsl = wave % NSPL, sg = wave / NSPL;
  const int cb = nsl * 16 * NF;                              // first output channel of this wave's slice
  // filter slice -> registers.  MFMA row (nf, i) is output channel cb + 4 NF (i >> 2) + 4 nf + (i & 3), so that accumulator
  // lane (i, q) ends up holding the 4 NF CONSECUTIVE channels cb + 4 NF q ... of its pixel (one 8 NF-byte store)
  u16x8 wr[3][KSR][NF];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int s = 0; s < KSR; ++s)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const int gr = 4 * s + q;
        u16x8 v = (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (gr < 3 * PP && (!ONE || (d == 1 && gr >= PP && gr < 2 * PP))) {
          const int dxi = gr / PP, part = gr - dxi * PP;
          const int slot = dxi == 0 ? g.tapw[3 * d] : (dxi == 1 ? g.tapw[3 * d + 1] : g.tapw[3 * d + 2]);
          const int ch = c0 + cb + 4 * NF * (i >> 2) + 4 * nf + (i & 3);
          v = *(const u16x8*)(g.w + ((long)ch * g.wtaps + slot) * CIN + 8 * part);
        }
        wr[d][s][nf] = v;
      }
  // LDS byte offset (inside a ring row) of this lane's fragment of k-step s of its first strip: pixel i + dxi, part rotated by the pixel
  int xoff[KSR];
#pragma unroll
  for (int s = 0; s < KSR; ++s) {
    const int gr = (4 * s + q) < 3 * PP ? 4 * s + q : 3 * PP - 1;     // zero-weight tail granules read a legal address
    const int dxi = gr / PP, part = gr - dxi * PP, p = i + dxi;
    xoff[s] = (p * PP + (part + ((C3Swz<CIN>::A * p) >> C3Swz<CIN>::SH)) % PP) * 16 + sg * SPW * 256 * PP;
  }
  f32x4 acc[3][SPW][NF];
  float ps[4 * NF], pss[4 * NF];          // BatchNorm sums of this lane's channels: forward statistics (sum, sum of squares) or post statistics (sum g, sum g * xhat)
#pragma unroll
  for (int c = 0; c < 4 * NF; ++c) { ps[c] = 0.f; pss[c] = 0.f; }
  int slot = 0;                                              // ring slot of the next row to consume
  constexpr int FR = SPW * KSR;                              // operand fragments of one row
  constexpr int CH = FR <= 6 ? FR : 6;                       // fragments per register set ("chunk"): two sets ping-pong
  constexpr int NCH = (FR + CH - 1) / CH;
  int bpar = 0;                                              // parity of the batch: which half of the staging area it fills
  if (LAG) raw_barrier();                                    // the transform waves work one batch ahead

  for (int it = blockIdx.x; it < g.items; it += G) {
    const C3Item im = c3_item(g, it, WB);
    const int K = im.r1 - im.r0 + 2;
#pragma unroll
    for (int sl = 0; sl < 3; ++sl)
#pragma unroll
      for (int st = 0; st < SPW; ++st)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[sl][st][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // One batch = up to three consecutive rows behind one barrier, as straight-line code: the fragment reads of chunk c + 1 are
    // issued before the MFMAs of chunk c (two register sets); the first MFMA into an accumulator set (tap row dy = -1, first
    // k-step) takes a zero C operand, so finished sets are never cleared; the first batch of an item (HEAD) leaves out the taps
    // that would feed rows above the item, so those sets stay zero: the "output rows" above the item are zeros, staged like
    // any other row and skipped by the store waves - no masks and no branch in the conversion / statistics / staging code.
    auto batch = [&](auto fullc, auto headc, int k0, int n) {
      constexpr bool FULL = decltype(fullc)::value, HEAD = decltype(headc)::value;
      C3_STAMP(tb, 0);
      wait_lgkm0();                       // this wave's staged output rows of the previous batch are written
      raw_barrier();                      // the batch's rows are in LDS (the producers waited for their DMA before arriving)
      asm volatile("" ::: "memory");
      C3_STAMP(tb, 1);
      char* const sbat = stg + bpar * 3 * STGROW;
      bpar ^= 1;
      const char* rows[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        rows[j] = smem + slot * ROWB;
        if (FULL || j < n) slot = slot + 1 == g.NR ? 0 : slot + 1;
      }
      u16x8 xs[2][CH];
      auto load_chunk = [&](int c) {       // (c is a compile-time constant after unrolling)
        const int j = c / NCH, cc = c - j * NCH;
        if (FULL || j < n) {
#pragma unroll
          for (int f = 0; f < CH; ++f) {
            const int fr = cc * CH + f;
            if (fr < FR) xs[c & 1][f] = (C3_ABL & 4) ? wr[0][0][0] : *(const u16x8*)(rows[j] + (fr / KSR) * 256 * PP + xoff[fr % KSR]);
          }
        }
      };
      // What a finished output row-strip still needs - conversion + store, and (forward layers) its share of the BatchNorm
      // statistics - is cut into NT small tasks that are placed IN FRONT of the MFMAs of the following fragments, one per
      // fragment, each fragment a scheduling region of its own: the vector instructions then issue while the previous
      // fragment's MFMAs are still in the pipe.  (Left to itself the scheduler sinks all of them behind the batch's last MFMA:
      // ~500 cycles per batch with the matrix pipe idle.)  The tasks of strip (jj, ss) must precede the first MFMA that
      // re-opens its accumulator set: row jj + 1's first k-step of the same strip, (SPW - 1) KSR fragments later.
      constexpr int NT = (STATS && SPW > 1) ? 3 : 1;
      auto task = [&](int jj, int ss, int tt) {
        if (!(FULL || jj < n)) return;
        const int SL = (jj + 2) % 3;        // output row r0 + k - 2 has seen its three input rows (phase jj = k % 3: batches start at multiples of three)
        const int k = k0 + jj, ro = im.r0 + k - 2;
        const int cl = 16 * (sg * SPW + ss) + i;
        const bool ok = !MASKED || im.x0 + cl < g.W;
        float v[4 * NF];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[4 * nf + r] = acc[SL][ss][nf][r];
        if (tt == 0) {
          if (RES) {
            const bf16_t* rp = (const bf16_t*)(rows[jj] + RPXP * 16) + cl * COUT + cb + 4 * NF * q;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
              const u16x4 rv = *(const u16x4*)(rp + 4 * nf);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[4 * nf + r] += bf2f(rv[r]);
            }
          }
          // staged at [row jj][pixel cl], 16-byte chunks rotated by the pixel (bank spread); rows the item does not own are staged
          // too (zeros / partial sums) and skipped by the store waves
          char* const sp = sbat + (jj * WB + cl) * (COUT * 2);
          const int rot = (cl >> RSH) % CPP;
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) {
            const int o = (cb + 4 * NF * q + 4 * nf) * 2;           // byte offset of these four channels inside the pixel
            float v4[4] = {v[4 * nf], v[4 * nf + 1], v[4 * nf + 2], v[4 * nf + 3]};
            if (!(C3_ABL & 1)) store4((bf16_t*)(sp + ((o / 16 + rot) % CPP) * 16 + (o & 8)), v4);
          }
        }
        if (STATS && !(C3_ABL & 2) && tt == (NT == 1 ? 0 : 1)) {
#pragma unroll
          for (int c2 = 0; c2 < 4 * NF; ++c2) ps[c2] += (MASKED && !ok) ? 0.f : v[c2];
        }
        if (STATS && !(C3_ABL & 2) && tt == (NT == 1 ? 0 : 2)) {
#pragma unroll
          for (int c2 = 0; c2 < 4 * NF; ++c2) pss[c2] += (MASKED && !ok) ? 0.f : v[c2] * v[c2];
        }
      };
      load_chunk(0);
#pragma unroll
      for (int L = 0; L < 3 * FR; ++L) {
        const int j = L / FR, fr = L - j * FR, st = fr / KSR, s = fr - st * KSR;
        const int c = j * NCH + fr / CH, f = fr % CH;
        if (f == 0 && c + 1 < 3 * NCH) load_chunk(c + 1);
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
#pragma unroll
          for (int ss = 0; ss < SPW; ++ss)
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
              if (jj * FR + ss * KSR + KSR + tt == L) task(jj, ss, tt);
        if (FULL || j < n) {
#pragma unroll
          for (int d = 2; d >= 0; --d) {
            if (HEAD && d > j) continue;          // input row r0 - 1 + j, tap row d - 1: output row r0 + j - d is above the item
            if (ONE && (d != 1 || 4 * s + 3 < PP || 4 * s >= 2 * PP)) continue;      // 1x1: the centre tap's k-steps only
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
              if (C3_ABL & 8) {
                acc[(j + 4 - d) % 3][st][nf][0] += bf2f(xs[c & 1][f][nf]);
              } else if (ONE ? (4 * s <= PP && PP < 4 * s + 4) : (d == 0 && s == 0)) {
                f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
                mma16(wr[d][s][nf], xs[c & 1][f], z);
                acc[(j + 4 - d) % 3][st][nf] = z;
              } else {
                mma16(wr[d][s][nf], xs[c & 1][f], acc[(j + 4 - d) % 3][st][nf]);
              }
            }
          }
        }
        if (FULL) {
        }
      }
      // the tasks whose fragment lies in the next batch
#pragma unroll
      for (int jj = 0; jj < 3; ++jj)
#pragma unroll
        for (int ss = 0; ss < SPW; ++ss)
#pragma unroll
          for (int tt = 0; tt < NT; ++tt)
            if (jj * FR + ss * KSR + KSR + tt >= 3 * FR) task(jj, ss, tt);
    };
    for (int k0 = 0; k0 < K; k0 += 3) {
      if (k0 + 3 <= K) {
        if (k0 == 0) batch(std::true_type(), std::true_type(), k0, 3);
        else batch(std::true_type(), std::false_type(), k0, 3);
      } else {
        if (k0 == 0) batch(std::false_type(), std::true_type(), k0, K - k0);
        else batch(std::false_type(), std::false_type(), k0, K - k0);
      }
      C3_STAMP(tb, 2);
#ifdef C3_TRACE
      ++tb;
#endif
    }
  }
  wait_lgkm0();
  raw_barrier();          // the closing barrier: the store waves take the last batch
  if (STATS) {
    double* st = g.stats + (long)(blockIdx.x % MDS_STAT_SLOTS) * 2 * g.Ctot + c0;
#pragma unroll
    for (int c = 0; c < 4 * NF; ++c) {
      const float a = sum_over_i16(ps[c]), b = sum_over_i16(pss[c]);
      if (i == 0) {
        atomicAdd(st + cb + 4 * NF * q + c, (double)a);
        atomicAdd(st + g.Ctot + cb + 4 * NF * q + c, (double)b);
      }
    }
  }
#ifdef C3_TRACE
  raw_barrier();
  if (trc) {
    const unsigned long long* tl = (const unsigned long long*)(smem + g.NR * ROWB + 6 * STGROW);
    for (int e = tid; e < 160 * 32; e += 256) ((unsigned long long*)g.trace)[e] = tl[e];
  }
#endif
}


// ------------------------------------------------------------------------------------------------------------------------
// Stride-2 data gradient (the transposed convolution of a TF-SAME stride-2 3x3 layer with even input extents: pads 0 / 1).
// Input = dy [N][H][W][CIN], output = dx [N][2H][2W][COUT].  Input pixel (a, b) feeds output (2a + ky, 2b + kx) through
// w[ky][kx], ky, kx in 0..2.  Same machinery as c3_kernel (filter slice in registers, DMA ring, store waves); what changes:
//   * one input row finishes TWO output rows: 2a (taps ky = 0 of row a + ky = 2 of row a - 1, carried in a rolling set) and
//     2a + 1 (ky = 1, this row only); one fragment (16 input pixels) serves the even AND the odd output columns (kx = 0 / 1)
//     and, read one pixel to the left, the even columns again (kx = 2): 9 KS NF MFMAs per 2 KS fragment reads;
//   * batches of two rows (the rolling set's period); an item's first row (a = r0 - 1) only contributes its ky = 2 taps.
// k_conv.hip evaluates this layer as four tap groups with a 16 x NF-wide LDS filter slab: one MFMA per two fragment reads.
template <int CIN, int NF, int NSPL, int NPW, int NSW, bool MASKED, bool POST>
__global__ __launch_bounds__(256 + 64 * (NPW + NSW)) void c3t_kernel(C3Args g) {
  constexpr int PP = CIN / 8, KS = CIN / 32, NSG = 4 / NSPL, WBI = 16 * NSG, WBO = 2 * WBI, COUT = 16 * NF * NSPL, CPP = COUT / 8;
  constexpr int RSX = (WBI + 1) * PP;                // the input row of the band (+ the left neighbour pixel)
  constexpr int RPYP = (RSX + 63) / 64 * 64;         // POST: post.y's two output rows of the band follow on a piece boundary
  constexpr int RS = POST ? RPYP + 2 * WBO * CPP : RSX, PIECES = (RS + 63) / 64, ROWB = RS * 16;
  constexpr int STGROW = WBO * COUT * 2;
  constexpr int RSH = CPP == 4 ? 1 : 0;
  static_assert(CIN % 32 == 0, "whole k-steps per tap");
  MDS_DYN_SMEM(smem);
  char* const stg = smem + g.NR * ROWB;               // [2 batch parities][2 input rows][2 output rows][WBO pixels][COUT] bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);
  const int G = gridDim.x;
  const int c0 = blockIdx.y * COUT;
  int E = 0;
  for (int it = blockIdx.x; it < g.items; it += G) {
    const C3Item im = c3_item(g, it, WBI);
    E += im.r1 - im.r0 + 1;
  }
  if (wave >= 4 && wave < 4 + NPW) {
    // ------------------------------------------------------------------ DMA waves
    MDS_SETPRIO(3);
    const int pw = wave - 4;
    constexpr int PCWMAX = (PIECES + NPW - 1) / NPW;
    const int pcw = (PIECES - pw + NPW - 1) / NPW;
    int dcol[PCWMAX], eoff[PCWMAX];
#pragma unroll
    for (int j = 0; j < PCWMAX; ++j) {
      const int sg_ = 64 * (pw + NPW * j) + lane;
      if (sg_ < RSX) {
        const int p = sg_ / PP, psw = sg_ - p * PP;
        const int rot = ((C3Swz<CIN>::A * p) >> C3Swz<CIN>::SH) % PP;
        dcol[j] = p - 1; eoff[j] = 8 * ((psw - rot + PP) % PP);
      } else if (POST && sg_ >= RPYP && sg_ < RS) {
        const int t = sg_ - RPYP, yr = t / (WBO * CPP), u = t - yr * (WBO * CPP), p = u / CPP;
        dcol[j] = p + (yr << 20); eoff[j] = 8 * (u - p * CPP);      // output column (relative to 2 x0) and, in bit 20, the output row's parity
      } else {
        dcol[j] = -(1 << 30); eoff[j] = 0;
      }
    }
    const lds_t ring = lds_addr_of(smem);
    int hit = blockIdx.x, hk = 0;
    C3Item him = c3_item(g, hit < g.items ? hit : 0, WBI);
    int C = 0, hslot = 0;
    const char* cur[PCWMAX];
    unsigned step[PCWMAX];
    auto open_item = [&]() {
      const char* xrow0 = (const char*)g.x + ((long)him.n * g.H + him.r0 - 1) * g.W * CIN * 2;
      const char* yrow0 = POST ? (const char*)(g.py + c0) + ((long)him.n * g.OH + 2 * (him.r0 - 1)) * g.OW * g.Ctot * 2 : nullptr;
#pragma unroll
      for (int j = 0; j < PCWMAX; ++j) {
        if (POST && 64 * (pw + NPW * j) >= RPYP) {
          const int yr = dcol[j] >> 20, gx = 2 * him.x0 + (dcol[j] & 0xfffff);
          const bool ok = dcol[j] >= 0 && gx < g.OW && 2 * (g.H - 1) + yr < g.OH;
          cur[j] = ok ? yrow0 + ((long)(yr * g.OW + gx) * g.Ctot + eoff[j]) * 2 : (const char*)c3_zero_page;
          step[j] = ok ? (unsigned)(2 * g.OW * g.Ctot * 2) : 0u;
        } else {
          const int gx = him.x0 + dcol[j];
          const bool ok = gx >= 0 && gx < g.W;
          cur[j] = ok ? xrow0 + (gx * CIN + eoff[j]) * 2 : (const char*)c3_zero_page;
          step[j] = ok ? (unsigned)(g.W * CIN * 2) : 0u;
        }
      }
    };
    open_item();
    auto issue = [&]() {
      const int ri = him.r0 - 1 + hk;
      const bool xok = ri >= 0 && ri < g.H;
      const lds_t dst = ring + (lds_t)(hslot * ROWB);
      hslot = hslot + 1 == g.NR ? 0 : hslot + 1;
#pragma unroll
      for (int j = 0; j < PCWMAX; ++j) {
        const int pi = pw + NPW * j;
        if (pi < PIECES) {
          const char* src = xok ? cur[j] : (const char*)c3_zero_page;
          if (dcol[j] > -(1 << 29)) glds16(src, dst + (lds_t)(pi * 1024));
          cur[j] += step[j];
        }
      }
      ++C;
      if (++hk == him.r1 - him.r0 + 1) {
        hk = 0; hit += G;
        if (hit < g.items) { him = c3_item(g, hit, WBI); open_item(); }
      }
    };
    while (C < g.RA && C < E) issue();
    int e0 = 0;
    for (int it = blockIdx.x; it < g.items; it += G) {
      const C3Item im = c3_item(g, it, WBI);
      const int K = im.r1 - im.r0 + 1;
      for (int k0 = 0; k0 < K; k0 += 2) {
        const int n = K - k0 < 2 ? K - k0 : 2;
        wait_vm_dyn(pcw * (C - e0 - n));
        raw_barrier();
        e0 += n;
        while (C < e0 + g.RA && C < E) issue();
      }
    }
    raw_barrier();
    return;
  }
  if (wave >= 4 + NPW) {
    // ------------------------------------------------------------------ store waves
    MDS_SETPRIO(3);
    const int sw = wave - 4 - NPW;
    constexpr int PPP = 64 / CPP, NP = (WBO + PPP - 1) / PPP;
    const int lpx = lane / CPP, chunk = lane - lpx * CPP;
    const bool lact = lpx < PPP;
    int bidx = 0;
    C3Item pim = c3_item(g, blockIdx.x, WBI);
    int pk0 = 0, pn = 0;
    int rslot = 0, pslot = 0;                             // POST (see c3_kernel's store waves): ring slot of the next row / of the previous batch's first row
    float ps[8], pss[8], pmu[8], prs[8], psc[8], psh[8];
    const bool psilu = POST && g.pmode == MDS_POST_SILU;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      ps[c] = 0.f; pss[c] = 0.f;
      psc[c] = psilu ? g.pbn[c0 + 8 * chunk + c] : 0.f;
      psh[c] = psilu ? g.pbn[g.Ctot + c0 + 8 * chunk + c] : 0.f;
      pmu[c] = POST ? g.pbn[2 * g.Ctot + c0 + 8 * chunk + c] : 0.f;
      prs[c] = POST ? g.pbn[3 * g.Ctot + c0 + 8 * chunk + c] : 0.f;
    }
    auto flush = [&](int par) {
      const float pmk = (POST && g.pmask) ? g.pmask[pim.n] : 1.f;
      for (int j = 0; j < pn; ++j) {
        const int k = pk0 + j, a = pim.r0 - 1 + k;
        if (k < 1) continue;                              // the item's first row only opens the carried set
        int rs_ = pslot + j;
        if (rs_ >= g.NR) rs_ -= g.NR;
        for (int yp = 0; yp < 2; ++yp) {
          const int Y = 2 * a + yp;
          if (Y >= g.OH) continue;
          const char* srow = stg + ((par * 2 + j) * 2 + yp) * STGROW;
          const bf16_t* yrow = (const bf16_t*)(smem + rs_ * ROWB + RPYP * 16) + yp * WBO * COUT;
          bf16_t* const orow = g.y + c0 + (((long)pim.n * g.OH + Y) * g.OW + 2 * pim.x0) * g.Ctot;
          for (int pc = sw; pc < NP; pc += NSW) {
            const int cl = pc * PPP + lpx;
            if (lact && cl < WBO && (!MASKED || 2 * pim.x0 + cl < g.OW)) {
              u16x8 v = *(const u16x8*)(srow + (cl * CPP + (chunk + (cl >> RSH)) % CPP) * 16);
              if (POST) {
                const u16x8 yv = *(const u16x8*)(yrow + cl * COUT + 8 * chunk);
                if (psilu) {
                  float gv[8];
#pragma unroll
                  for (int c = 0; c < 8; ++c) gv[c] = bf2f(v[c]) * silu_gradf_(bf2f(yv[c]) * psc[c] + psh[c]);
                  v = pack8(gv);
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                  const float gg = bf2f(v[c]) * pmk;
                  ps[c] += gg;
                  pss[c] += gg * ((bf2f(yv[c]) - pmu[c]) * prs[c]);
                }
              }
              *(u16x8*)(orow + cl * g.Ctot + 8 * chunk) = v;
            }
          }
        }
      }
    };
    for (int it = blockIdx.x; it < g.items; it += G) {
      const C3Item im = c3_item(g, it, WBI);
      const int K = im.r1 - im.r0 + 1;
      for (int k0 = 0; k0 < K; k0 += 2) {
        asm volatile("" ::: "memory");
        raw_barrier();
        asm volatile("" ::: "memory");
        if (bidx > 0) flush((bidx - 1) & 1);
        pim = im; pk0 = k0; pn = K - k0 < 2 ? K - k0 : 2;
        pslot = rslot; rslot += pn;
        if (rslot >= g.NR) rslot -= g.NR;
        ++bidx;
      }
    }
    asm volatile("" ::: "memory");
    raw_barrier();
    asm volatile("" ::: "memory");
    flush((bidx - 1) & 1);
    if (POST) {
      double* st = g.pstats + (long)(blockIdx.x % MDS_STAT_SLOTS) * 2 * g.Ctot + c0;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float a = lact ? ps[c] : 0.f, b = lact ? pss[c] : 0.f;
        for (int o = CPP; o < 64; o += CPP) { a += __shfl_down(lact ? ps[c] : 0.f, o); b += __shfl_down(lact ? pss[c] : 0.f, o); }
        if (lane < CPP) {
          atomicAdd(st + 8 * chunk + c, (double)a);
          atomicAdd(st + g.Ctot + 8 * chunk + c, (double)b);
        }
      }
    }
    return;
  }
  // -------------------------------------------------------------------- consumers
  MDS_SETPRIO(2);
  const int i = lane & 15, q = lane >> 4;
  const int nsl = wave % NSPL, sg = wave / NSPL;
  const int cb = nsl * 16 * NF;
  u16x8 wr[3][3][KS][NF];                                   // [ky][kx][k-step][output fragment]
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const int ch = c0 + cb + 4 * NF * (i >> 2) + 4 * nf + (i & 3);
          wr[ky][kx][s][nf] = *(const u16x8*)(g.w + ((long)ch * g.wtaps + g.tapw[3 * ky + kx]) * CIN + 8 * (4 * s + q));
        }
  int xoff[2][KS];                                          // [0]: the pixel itself (taps kx = 0, 1), [1]: its left neighbour (kx = 2)
#pragma unroll
  for (int dxi = 0; dxi < 2; ++dxi)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int p = i + 1 - dxi, part = 4 * s + q;
      xoff[dxi][s] = (p * PP + (part + ((C3Swz<CIN>::A * p) >> C3Swz<CIN>::SH)) % PP) * 16 + sg * 256 * PP;
    }
  f32x4 ev[2][2][NF], od[2][NF];                            // even output rows: [rolling set][x parity]; odd output row: [x parity]
  int slot = 0, bpar = 0;
  for (int it = blockIdx.x; it < g.items; it += G) {
    const C3Item im = c3_item(g, it, WBI);
    const int K = im.r1 - im.r0 + 1;
    auto batch = [&](auto fullc, auto headc, int k0, int n) {
      constexpr bool FULL = decltype(fullc)::value, HEAD = decltype(headc)::value;
      wait_lgkm0();
      raw_barrier();
      asm volatile("" ::: "memory");
      char* const sbat = stg + bpar * 4 * STGROW;
      bpar ^= 1;
      const char* rows[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        rows[j] = smem + slot * ROWB;
        if (FULL || j < n) slot = slot + 1 == g.NR ? 0 : slot + 1;
      }
      u16x8 xs[2][KS];
      auto load_chunk = [&](int c) {       // chunk c = (row c / 2, left-neighbour flag c % 2)
        const int j = c >> 1, dxi = c & 1;
        if (FULL || j < n) {
#pragma unroll
          for (int s = 0; s < KS; ++s) xs[c & 1][s] = *(const u16x8*)(rows[j] + xoff[dxi][s]);
        }
      };
      load_chunk(0);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int j = c >> 1, dxi = c & 1;
        if (c + 1 < 4) load_chunk(c + 1);
        if (FULL || j < n) {
          const bool head = HEAD && j == 0;                 // row r0 - 1: only the taps that reach down into the item (ky = 2)
#pragma unroll
          for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int ky = 2; ky >= 0; --ky) {
              if (head && ky != 2) continue;
#pragma unroll
              for (int nf = 0; nf < NF; ++nf) {
                // ky = 0 adds to the set row a - 1 opened (ev[j]), ky = 1 is the odd row (fresh), ky = 2 opens ev[j ^ 1] (fresh)
                if (dxi == 0) {
                  f32x4& te = ky == 0 ? ev[j][0][nf] : (ky == 1 ? od[0][nf] : ev[j ^ 1][0][nf]);
                  f32x4& to = ky == 0 ? ev[j][1][nf] : (ky == 1 ? od[1][nf] : ev[j ^ 1][1][nf]);
                  if (ky != 0 && s == 0) {
                    f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f}, z2 = (f32x4){0.f, 0.f, 0.f, 0.f};
                    mma16(wr[ky][0][s][nf], xs[c & 1][s], z);  te = z;
                    mma16(wr[ky][1][s][nf], xs[c & 1][s], z2); to = z2;
                  } else {
                    mma16(wr[ky][0][s][nf], xs[c & 1][s], te);
                    mma16(wr[ky][1][s][nf], xs[c & 1][s], to);
                  }
                } else {
                  f32x4& te = ky == 0 ? ev[j][0][nf] : (ky == 1 ? od[0][nf] : ev[j ^ 1][0][nf]);
                  mma16(wr[ky][2][s][nf], xs[c & 1][s], te);
                }
              }
            }
          if (dxi == 1) {
            // rows 2a (ev[j]) and 2a + 1 (od) of the band are complete: convert and stage (the store waves skip an item's row r0 - 1)
            const int clb = 2 * (16 * sg + i);
#pragma unroll
            for (int yp = 0; yp < 2; ++yp)
#pragma unroll
              for (int xp = 0; xp < 2; ++xp) {
                const int cl = clb + xp;
                char* const sp = sbat + ((j * 2 + yp) * WBO + cl) * (COUT * 2);
                const int rot = (cl >> RSH) % CPP;
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
                  const f32x4& t = yp == 0 ? ev[j][xp][nf] : od[xp][nf];
                  const int o = (cb + 4 * NF * q + 4 * nf) * 2;
                  float v4[4] = {t[0], t[1], t[2], t[3]};
                  store4((bf16_t*)(sp + ((o / 16 + rot) % CPP) * 16 + (o & 8)), v4);
                }
              }
          }
        }
      }
    };
    for (int k0 = 0; k0 < K; k0 += 2) {
      if (k0 + 2 <= K) {
        if (k0 == 0) batch(std::true_type(), std::true_type(), k0, 2);
        else batch(std::true_type(), std::false_type(), k0, 2);
      } else {
        if (k0 == 0) batch(std::false_type(), std::true_type(), k0, K - k0);
        else batch(std::false_type(), std::false_type(), k0, K - k0);
      }
    }
  }
  wait_lgkm0();
  raw_barrier();
}

// ------------------------------------------------------------------------------------------------------------------------
// Stride-2 FORWARD (TF-SAME with even input extents: pads 0 / 1): out(oy, ox) = sum in(2 oy + ky, 2 ox + kx) w[ky][kx], ky, kx in 0..2
// (blocks.1.0's and blocks.2.0's first convolutions: 16 -> 64 and 32 -> 128 channels, a quarter of the pixels out).  c3_kernel's
// machinery - filter slice in registers, DMA row ring, store waves, transform waves for an input read through BatchNorm + SiLU -
// around a different walk:
//   * an output row needs input rows 2 oy (ky = 0), 2 oy + 1 (ky = 1) and 2 oy + 2 (ky = 2, also ky = 0 of the row below): an EVEN
//     input row closes one accumulator set and opens the other, an odd row feeds the open one - two sets, batches of four input
//     rows (two output rows) behind one barrier;
//   * the three taps of a row are 3 CIN consecutive channels starting at pixel 2 ox, so the flattened k-steps of c3_kernel carry
//     over with fragment lane i at pixel 2 i (+ dx): pixel stride 2 costs a 2-way LDS bank conflict whatever the part rotation
//     (tools/probes/c3_swizzle_check.py) - 8 cycles per fragment read where the HBM stream leaves ~100;
//   * an input row of a band is 2 WB + 1 pixels; rows / columns H, W (the pad) come from the zero page.
template <int CIN> struct C3SwzS2;     // part' = (part + ((A * pixel) >> SH)) % (CIN / 8) under pixel stride 2
template <> struct C3SwzS2<16> { static constexpr int A = 1, SH = 3; };
template <> struct C3SwzS2<32> { static constexpr int A = 1, SH = 1; };

MDS_DEV C3Item c3s_item(const C3Args& g, int it, int WB) {      // output coordinates
  const int seg = it % g.nseg, t = it / g.nseg;
  const int band = t % g.nbands, n = t / g.nbands;
  C3Item r;
  r.n = n; r.x0 = band * WB; r.r0 = seg * g.rps;
  r.r1 = r.r0 + g.rps < g.OH ? r.r0 + g.rps : g.OH;
  return r;
}

template <int CIN, int NF, int NSPL, int SPW, int NPW, int NSW, bool STATS, bool MASKED, int NTW>
__global__ __launch_bounds__(256 + 64 * (NPW + NSW + NTW)) void c3s_kernel(C3Args g) {
  constexpr int LAG = NTW > 0 ? 1 : 0;
  constexpr int PP = CIN / 8, KSR = (3 * PP + 3) / 4, NSG = 4 / NSPL, WB = 16 * NSG * SPW, COUT = 16 * NF * NSPL, CPP = COUT / 8;
  constexpr int RPX = (2 * WB + 1) * PP, PIECES = (RPX + 63) / 64, ROWB = RPX * 16;
  constexpr int STGROW = WB * COUT * 2;
  constexpr int RSH = CPP == 4 ? 1 : 0;
  typedef C3SwzS2<CIN> SW;
  static_assert((((SW::A * 32) >> SW::SH) % PP) == 0, "the rotation repeats from strip to strip");
  MDS_DYN_SMEM(smem);
  char* const stg = smem + g.NR * ROWB;               // [2 batch parities][2 output rows][WB pixels][COUT] bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);
  const int G = gridDim.x;
  const int c0 = blockIdx.y * COUT;
  int E = 0;                                          // input rows this block stages and consumes: 2 rows + 1 per item
  for (int it = blockIdx.x; it < g.items; it += G) {
    const C3Item im = c3s_item(g, it, WB);
    E += 2 * (im.r1 - im.r0) + 1;
  }
  if (wave >= 4 && wave < 4 + NPW) {
    // ------------------------------------------------------------------ DMA waves
    MDS_SETPRIO(3);
    const int pw = wave - 4;
    constexpr int PCWMAX = (PIECES + NPW - 1) / NPW;
    const int pcw = (PIECES - pw + NPW - 1) / NPW;
    int dcol[PCWMAX], eoff[PCWMAX];
#pragma unroll
    for (int j = 0; j < PCWMAX; ++j) {
      const int sg_ = 64 * (pw + NPW * j) + lane;
      if (sg_ < RPX) {
        const int p = sg_ / PP, psw = sg_ - p * PP;
        const int rot = ((SW::A * p) >> SW::SH) % PP;
        dcol[j] = p; eoff[j] = 8 * ((psw - rot + PP) % PP);
      } else {
        dcol[j] = -(1 << 30); eoff[j] = 0;
      }
    }
    const lds_t ring = lds_addr_of(smem);
    int hit = blockIdx.x, hk = 0;
    C3Item him = c3s_item(g, hit < g.items ? hit : 0, WB);
    int C = 0, hslot = 0;
    const char* cur[PCWMAX];
    unsigned step[PCWMAX];
    auto open_item = [&]() {
      const char* xrow0 = (const char*)g.x + ((long)him.n * g.H + 2 * him.r0) * g.W * CIN * 2;
#pragma unroll
      for (int j = 0; j < PCWMAX; ++j) {
        const int gx = 2 * him.x0 + dcol[j];
        const bool ok = gx >= 0 && gx < g.W;
        cur[j] = ok ? xrow0 + (gx * CIN + eoff[j]) * 2 : (const char*)c3_zero_page;
        step[j] = ok ? (unsigned)(g.W * CIN * 2) : 0u;
      }
    };
    open_item();
    auto issue = [&]() {
      const bool xok = 2 * him.r0 + hk < g.H;             // (row H: the bottom pad)
      const lds_t dst = ring + (lds_t)(hslot * ROWB);
      hslot = hslot + 1 == g.NR ? 0 : hslot + 1;
#pragma unroll
      for (int j = 0; j < PCWMAX; ++j) {
        const int pi = pw + NPW * j;
        if (pi < PIECES) {
          const char* src = xok ? cur[j] : (const char*)c3_zero_page;
          if (dcol[j] > -(1 << 29)) glds16(src, dst + (lds_t)(pi * 1024));
          cur[j] += step[j];
        }
      }
      ++C;
      if (++hk == 2 * (him.r1 - him.r0) + 1) {
        hk = 0; hit += G;
        if (hit < g.items) { him = c3s_item(g, hit, WB); open_item(); }
      }
    };
    while (C < g.RA && C < E) issue();
    int e0 = 0;
    for (int it = blockIdx.x; it < g.items; it += G) {
      const C3Item im = c3s_item(g, it, WB);
      const int K = 2 * (im.r1 - im.r0) + 1;
      for (int k0 = 0; k0 < K; k0 += 4) {
        const int n = K - k0 < 4 ? K - k0 : 4;
        wait_vm_dyn(pcw * (C - e0 - n));
        raw_barrier();
        e0 += n;
        while (C < e0 + g.RA && C < E) issue();
      }
    }
    raw_barrier();
    if (LAG) raw_barrier();
    return;
  }
  if (wave >= 4 + NPW + NSW) {
    // ------------------------------------------------------------------ transform waves (NTW > 0): as in c3_kernel
    MDS_SETPRIO(3);
    const int tw = wave - 4 - NPW - NSW;
    static_assert(NTW == 0 || (32 % PP == 0 && (((SW::A * (32 / PP)) >> SW::SH) % PP) == 0), "lane-constant channels");
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int hl = lane >> 1, lp = hl / PP, lpart = ((hl % PP) - ((SW::A * lp) >> SW::SH) % PP + PP) % PP;
    f32x2 sc[2], sh[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ch = 8 * lpart + 4 * (lane & 1) + 2 * c;
      sc[c] = (f32x2){g.pro_scale[ch], g.pro_scale[ch + 1]};
      sh[c] = (f32x2){g.pro_shift[ch], g.pro_shift[ch + 1]};
    }
    constexpr int NPC = (2 * RPX + 63) / 64;              // half-slot pieces of one row
    int rslot = 0;
    for (int it = blockIdx.x; it < g.items; it += G) {
      const C3Item im = c3s_item(g, it, WB);
      const int K = 2 * (im.r1 - im.r0) + 1;
      for (int k0 = 0; k0 < K; k0 += 4) {
        const int n = K - k0 < 4 ? K - k0 : 4;
        asm volatile("" ::: "memory");
        raw_barrier();
        asm volatile("" ::: "memory");
        constexpr int MAXP = (4 * NPC + NTW - 1) / (NTW ? NTW : 1);
        u16x4 v[MAXP];
        char* ptr[MAXP];
        int edge[MAXP];
#pragma unroll
        for (int q = 0; q < MAXP; ++q) {
          const int u = tw + q * NTW;
          edge[q] = -1;
          if (u < n * NPC) {
            const int j = u / NPC, pc = u - j * NPC;
            const int ri = 2 * im.r0 + k0 + j, px0 = 2 * im.x0 + (32 / PP) * pc;
            int rs_ = rslot + j;
            if (rs_ >= g.NR) rs_ -= g.NR;
            const int h = 64 * pc + lane;
            ptr[q] = smem + rs_ * ROWB + (h < 2 * RPX ? h : 0) * 8;
            if (!(C3_ABL & 32)) v[q] = *(const u16x4*)ptr[q];
            edge[q] = (ri < g.H && px0 + 32 / PP <= g.W) ? 0 : 1;
          }
        }
#pragma unroll
        for (int q = 0; q < MAXP; ++q) {
          if (edge[q] < 0 || (C3_ABL & 32)) continue;
          if (C3_ABL & 16) { *(u16x4*)ptr[q] = v[q]; continue; }
          const int u = tw + q * NTW;
          const int j = u / NPC, pc = u - j * NPC;
          f32x2 z[2], e[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            z[c] = (f32x2){bf2f(v[q][2 * c]), bf2f(v[q][2 * c + 1])} * sc[c] + sh[c];
            const f32x2 t = z[c] * -1.4426950408889634f;
            e[c] = (f32x2){fast_exp2(t[0]), fast_exp2(t[1])} + 1.0f;
            z[c] *= (f32x2){fast_rcp(e[c][0]), fast_rcp(e[c][1])};
          }
          if (edge[q]) {      // the pad (row H, column W and beyond) is zero AFTER the activation
            const int ri = 2 * im.r0 + k0 + j, gx = 2 * im.x0 + (32 / PP) * pc + lp;
            const float okf = (ri < g.H && gx < g.W) ? 1.f : 0.f;
            z[0] *= okf; z[1] *= okf;
          }
          typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
          if (64 * pc + lane < 2 * RPX) *(u32x2_*)ptr[q] = (u32x2_){pack2(z[0][0], z[0][1]), pack2(z[1][0], z[1][1])};
        }
        rslot += n;
        if (rslot >= g.NR) rslot -= g.NR;
        wait_lgkm0();
      }
    }
    raw_barrier();
    raw_barrier();
    return;
  }
  if (wave >= 4 + NPW) {
    // ------------------------------------------------------------------ store waves
    MDS_SETPRIO(3);
    const int sw = wave - 4 - NPW;
    constexpr int PPP = 64 / CPP, NP = (WB + PPP - 1) / PPP;
    static_assert(CPP <= 64, "an output pixel is at most one piece wide");
    const int lpx = lane / CPP, chunk = lane - lpx * CPP;
    const bool lact = lpx < PPP;
    int bidx = 0;
    C3Item pim = c3s_item(g, blockIdx.x, WB);
    int pk0 = 0;
    if (LAG) raw_barrier();
    auto flush = [&](int par) {
      for (int j = 0; j < 2; ++j) {
        const int ro = pim.r0 + pk0 / 2 - 1 + j;          // batch k0 / 4 stages output rows r0 + k0 / 2 - 1 and r0 + k0 / 2
        if (ro < pim.r0 || ro >= pim.r1) continue;
        const char* srow = stg + (par * 2 + j) * STGROW;
        bf16_t* const orow = g.y + c0 + (((long)pim.n * g.OH + ro) * g.OW + pim.x0) * g.Ctot;
        for (int pc = sw; pc < NP; pc += NSW) {
          const int cl = pc * PPP + lpx;
          if (lact && cl < WB && (!MASKED || pim.x0 + cl < g.OW)) {
            const u16x8 v = *(const u16x8*)(srow + (cl * CPP + (chunk + (cl >> RSH)) % CPP) * 16);
            *(u16x8*)(orow + cl * g.Ctot + 8 * chunk) = v;
          }
        }
      }
    };
    for (int it = blockIdx.x; it < g.items; it += G) {
      const C3Item im = c3s_item(g, it, WB);
      const int K = 2 * (im.r1 - im.r0) + 1;
      for (int k0 = 0; k0 < K; k0 += 4) {
        asm volatile("" ::: "memory");
        raw_barrier();
        asm volatile("" ::: "memory");
        if (bidx > 0) flush((bidx - 1) & 1);
        pim = im; pk0 = k0;
        ++bidx;
      }
    }
    asm volatile("" ::: "memory");
    raw_barrier();
    asm volatile("" ::: "memory");
    flush((bidx - 1) & 1);
    return;
  }

  // -------------------------------------------------------------------- consumers
  MDS_SETPRIO(2);
  const int i = lane & 15, q = lane >> 4;
  const int nsl = wave % NSPL, sg = wave / NSPL;
  const int cb = nsl * 16 * NF;
  u16x8 wr[3][KSR][NF];      // [ky][k-step of (kx, channel)][output fragment]: c3_kernel's row order
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int s = 0; s < KSR; ++s)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const int gr = 4 * s + q;
        u16x8 v = (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (gr < 3 * PP) {
          const int dxi = gr / PP, part = gr - dxi * PP;
          const int slot = dxi == 0 ? g.tapw[3 * d] : (dxi == 1 ? g.tapw[3 * d + 1] : g.tapw[3 * d + 2]);
          const int ch = c0 + cb + 4 * NF * (i >> 2) + 4 * nf + (i & 3);
          v = *(const u16x8*)(g.w + ((long)ch * g.wtaps + slot) * CIN + 8 * part);
        }
        wr[d][s][nf] = v;
      }
  int xoff[KSR];
#pragma unroll
  for (int s = 0; s < KSR; ++s) {
    const int gr = (4 * s + q) < 3 * PP ? 4 * s + q : 3 * PP - 1;
    const int dxi = gr / PP, part = gr - dxi * PP, p = 2 * i + dxi;
    xoff[s] = (p * PP + (part + ((SW::A * p) >> SW::SH)) % PP) * 16 + sg * SPW * 512 * PP;
  }
  f32x4 acc[2][SPW][NF];
  float ps[4 * NF], pss[4 * NF];
#pragma unroll
  for (int c = 0; c < 4 * NF; ++c) { ps[c] = 0.f; pss[c] = 0.f; }
  int slot = 0;
  constexpr int FR = SPW * KSR;
  constexpr int CH = FR <= 6 ? FR : 6;
  constexpr int NCH = (FR + CH - 1) / CH;
  int bpar = 0;
  if (LAG) raw_barrier();

  for (int it = blockIdx.x; it < g.items; it += G) {
    const C3Item im = c3s_item(g, it, WB);
    const int K = 2 * (im.r1 - im.r0) + 1;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int st = 0; st < SPW; ++st)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[sl][st][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // One batch = input rows (e0, o0, e1, o1) of the item, e0 = 2 r0 + k0 even: e_jj closes set jj (output row r0 + k0 / 2 - 1 + jj:
    // its ky = 2 taps) and opens set 1 - jj (the row below: ky = 0, zero C operand on the first k-step); o_jj feeds set 1 - jj (ky = 1).
    // The first batch of an item (HEAD) leaves out e0's closing taps: set 0 stays zero, "output row r0 - 1" is staged as zeros
    // and skipped by the store waves.
    auto batch = [&](auto fullc, auto headc, int k0, int n) {
      constexpr bool FULL = decltype(fullc)::value, HEAD = decltype(headc)::value;
      wait_lgkm0();
      raw_barrier();
      asm volatile("" ::: "memory");
      char* const sbat = stg + bpar * 2 * STGROW;
      bpar ^= 1;
      const char* rows[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        rows[j] = smem + slot * ROWB;
        if (FULL || j < n) slot = slot + 1 == g.NR ? 0 : slot + 1;
      }
      u16x8 xs[2][CH];
      auto load_chunk = [&](int c) {
        const int j = c / NCH, cc = c - j * NCH;
        if (FULL || j < n) {
#pragma unroll
          for (int f = 0; f < CH; ++f) {
            const int fr = cc * CH + f;
            if (fr < FR) xs[c & 1][f] = *(const u16x8*)(rows[j] + (fr / KSR) * 512 * PP + xoff[fr % KSR]);
          }
        }
      };
      constexpr int NT = (STATS && SPW > 1) ? 3 : 1;
      auto task = [&](int jj, int ss, int tt) {
        if (!(FULL || 2 * jj < n)) return;
        const int cl = 16 * (sg * SPW + ss) + i;
        const bool ok = !MASKED || im.x0 + cl < g.OW;
        float v[4 * NF];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[4 * nf + r] = acc[jj][ss][nf][r];
        if (tt == 0) {
          char* const sp = sbat + (jj * WB + cl) * (COUT * 2);
          const int rot = (cl >> RSH) % CPP;
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) {
            const int o = (cb + 4 * NF * q + 4 * nf) * 2;
            float v4[4] = {v[4 * nf], v[4 * nf + 1], v[4 * nf + 2], v[4 * nf + 3]};
            store4((bf16_t*)(sp + ((o / 16 + rot) % CPP) * 16 + (o & 8)), v4);
          }
        }
        if (STATS && tt == (NT == 1 ? 0 : 1)) {
#pragma unroll
          for (int c2 = 0; c2 < 4 * NF; ++c2) ps[c2] += (MASKED && !ok) ? 0.f : v[c2];
        }
        if (STATS && tt == (NT == 1 ? 0 : 2)) {
#pragma unroll
          for (int c2 = 0; c2 < 4 * NF; ++c2) pss[c2] += (MASKED && !ok) ? 0.f : v[c2] * v[c2];
        }
      };
      load_chunk(0);
#pragma unroll
      for (int L = 0; L < 4 * FR; ++L) {
        const int j = L / FR, fr = L - j * FR, st = fr / KSR, s = fr - st * KSR;
        const int c = j * NCH + fr / CH, f = fr % CH;
        if (f == 0 && c + 1 < 4 * NCH) load_chunk(c + 1);
        // output row jj of the batch is closed by even row 2 jj: its strip ss is ready after that row's k-steps of the strip
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int ss = 0; ss < SPW; ++ss)
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
              if (2 * jj * FR + ss * KSR + KSR + tt == L) task(jj, ss, tt);
        if (FULL || j < n) {
          const int jj = j / 2;
          if ((j & 1) == 0) {
            if (!(HEAD && j == 0)) {
#pragma unroll
              for (int nf = 0; nf < NF; ++nf) mma16(wr[2][s][nf], xs[c & 1][f], acc[jj][st][nf]);
            }
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
              if (s == 0) {
                f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
                mma16(wr[0][s][nf], xs[c & 1][f], z);
                acc[1 - jj][st][nf] = z;
              } else {
                mma16(wr[0][s][nf], xs[c & 1][f], acc[1 - jj][st][nf]);
              }
            }
          } else {
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) mma16(wr[1][s][nf], xs[c & 1][f], acc[1 - jj][st][nf]);
          }
        }
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int ss = 0; ss < SPW; ++ss)
#pragma unroll
          for (int tt = 0; tt < NT; ++tt)
            if (2 * jj * FR + ss * KSR + KSR + tt >= 4 * FR) task(jj, ss, tt);
    };
    for (int k0 = 0; k0 < K; k0 += 4) {
      if (k0 + 4 <= K) {
        if (k0 == 0) batch(std::true_type(), std::true_type(), k0, 4);
        else batch(std::true_type(), std::false_type(), k0, 4);
      } else {
        if (k0 == 0) batch(std::false_type(), std::true_type(), k0, K - k0);
        else batch(std::false_type(), std::false_type(), k0, K - k0);
      }
    }
  }
  wait_lgkm0();
  raw_barrier();
  if (STATS) {
    double* st = g.stats + (long)(blockIdx.x % MDS_STAT_SLOTS) * 2 * g.Ctot + c0;
#pragma unroll
    for (int c = 0; c < 4 * NF; ++c) {
      const float a = sum_over_i16(ps[c]), b = sum_over_i16(pss[c]);
      if (i == 0) {
        atomicAdd(st + cb + 4 * NF * q + c, (double)a);
        atomicAdd(st + g.Ctot + cb + 4 * NF * q + c, (double)b);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the stride-1 3x3 layers (blocks.1.1: 32 -> 128, blocks.2.1: 48 -> 192), row streaming:
//   dw[co][ci][ky][kx] += sum over pixels of dy[n][r][x][co] * in[n][r + ky - 1][x + kx - 1][ci].
// The MFMA reduction index is the PIXEL: both operands are read from pixel-major LDS rows with the transposing read
// (ds_read_b64_tr_b16: 4 pixels x 16 channels per 16-lane group), 32 pixels of one band row per k-step.  A ring entry is image row j
// of the band: its input pixels (+ one halo pixel either side) and its dy pixels (zeros outside the item's rows, so that every
// (input row, dy row) pair is counted by exactly one item), landed by LDS-DMA waves as in c3_kernel.  Input row j meets dy rows
// j + 1, j, j - 1 (ky = 0, 1, 2): its three shifted fragments per 16 input channels are read ONCE for all three.  The four
// consumer waves split the OUTPUT channels (COW 16-channel fragments each, COP = 64 COW per channel pass = grid.y); a wave keeps
// its 9 x CIN x 16 COW accumulators in registers for the whole launch and adds them to dw once, through LDS, with coalesced atomics.
// k_conv.hip's kernel stages 8 x 16 patches through registers behind two block barriers per patch: 2.6 / 1.1 TB/s on these layers.
struct C3WArgs {
  const bf16_t* x;     // [N][H][W][CIN]
  const bf16_t* dy;    // [N][H][W][Ctot]
  float* dw;           // [Ctot][CIN][wtaps]
  int N, H, W, Ctot, wtaps;
  int tapw[9];         // weight slot of tap (ky, kx)
  int nbands, nseg, rps, items;
  int RA, NR;
  const float* pro_scale; const float* pro_shift;      // c3wp_kernel: the input is read through BatchNorm + SiLU
};

template <int NC> MDS_DEV int c3w_rot(int c, int p) {      // where 32-byte chunk c of pixel p sits inside the pixel: 8 consecutive pixels' chunk c tile all 64 banks
  return NC == 2 ? (c ^ ((p >> 2) & 1)) : (NC == 4 ? ((c + (p >> 1)) & 3) : (NC == 8 ? ((c + p) & 7) : (NC == 6 ? (c + ((p >> 2) & 1)) % 6 : c)));
}
template <int NC> MDS_DEV int c3w_unrot(int c, int p) {
  return NC == 2 ? (c ^ ((p >> 2) & 1)) : (NC == 4 ? ((c - (p >> 1)) & 3) : (NC == 8 ? ((c - p) & 7) : (NC == 6 ? (c - ((p >> 2) & 1) + 6) % 6 : c)));
}

template <int CIN, int COW, int NCW, int NPW>
__global__ __launch_bounds__(64 * (NCW + NPW)) void c3w_kernel(C3WArgs g) {
  constexpr int CI = CIN / 16, COP = 16 * COW * NCW, PPX = CIN / 8, PPY = COP / 8, WB = 32;
  constexpr int XS = (WB + 2) * PPX, XSP = (XS + 63) / 64 * 64, YS = WB * PPY, RS = XSP + YS, PIECES = RS / 64, ROWB = RS * 16;
  static_assert(YS % 64 == 0 && (CI == 2 || CI == 3) && (PPY == 8 || PPY == 12 || PPY == 16), "shapes of this kernel");
  MDS_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);
  const int G = gridDim.x;
  const int c0 = blockIdx.y * COP;
  int E = 0;
  for (int it = blockIdx.x; it < g.items; it += G) {
    const C3Item im = c3_item_h(g.nseg, g.nbands, g.rps, g.H, it, WB);
    E += im.r1 - im.r0 + 2;
  }
  if (wave >= NCW) {
    // ------------------------------------------------------------------ DMA waves
    MDS_SETPRIO(3);
    const int pw = wave - NCW;
    constexpr int PCWMAX = (PIECES + NPW - 1) / NPW;
    const int pcw = (PIECES - pw + NPW - 1) / NPW;
    int dcol[PCWMAX], eoff[PCWMAX];
#pragma unroll
    for (int j = 0; j < PCWMAX; ++j) {
      const int sg_ = 64 * (pw + NPW * j) + lane;
      if (sg_ < XS) {
        const int p = sg_ / PPX, t = sg_ - p * PPX;
        dcol[j] = p - 1; eoff[j] = 16 * c3w_unrot<CI>(t >> 1, p) + 8 * (t & 1);
      } else if (sg_ >= XSP && sg_ < RS) {
        const int u = sg_ - XSP, p = u / PPY, t = u - p * PPY;
        dcol[j] = p; eoff[j] = 16 * c3w_unrot<PPY / 2>(t >> 1, p) + 8 * (t & 1);
      } else {
        dcol[j] = -(1 << 30); eoff[j] = 0;
      }
    }
    const lds_t ring = lds_addr_of(smem);
    int hit = blockIdx.x, hk = 0;
    C3Item him = c3_item_h(g.nseg, g.nbands, g.rps, g.H, hit < g.items ? hit : 0, WB);
    int C = 0, hslot = 0;
    const char* cur[PCWMAX];
    unsigned step[PCWMAX];
    auto open_item = [&]() {
      const char* xrow0 = (const char*)g.x + ((long)him.n * g.H + him.r0 - 1) * g.W * CIN * 2;
      const char* yrow0 = (const char*)(g.dy + c0) + ((long)him.n * g.H + him.r0 - 1) * g.W * g.Ctot * 2;
#pragma unroll
      for (int j = 0; j < PCWMAX; ++j) {
        const bool isy = 64 * (pw + NPW * j) >= XSP;
        const int gx = him.x0 + dcol[j];
        const bool ok = gx >= 0 && gx < g.W;
        cur[j] = ok ? (isy ? yrow0 : xrow0) + (gx * (isy ? g.Ctot : CIN) + eoff[j]) * 2 : (const char*)c3_zero_page;
        step[j] = ok ? (unsigned)(g.W * (isy ? g.Ctot : CIN) * 2) : 0u;
      }
    };
    open_item();
    auto issue = [&]() {
      const int ri = him.r0 - 1 + hk;
      const bool xok = ri >= 0 && ri < g.H, yok = ri >= him.r0 && ri < him.r1;      // dy: the item's own rows only
      const lds_t dst = ring + (lds_t)(hslot * ROWB);
      hslot = hslot + 1 == g.NR ? 0 : hslot + 1;
#pragma unroll
      for (int j = 0; j < PCWMAX; ++j) {
        const int pi = pw + NPW * j;
        if (pi < PIECES) {
          const bool isy = 64 * pi >= XSP;                     // wave-uniform
          const char* src = (isy ? yok : xok) ? cur[j] : (const char*)c3_zero_page;
          if (dcol[j] > -(1 << 29)) glds16(src, dst + (lds_t)(pi * 1024));
          cur[j] += step[j];
        }
      }
      ++C;
      if (++hk == him.r1 - him.r0 + 2) {
        hk = 0; hit += G;
        if (hit < g.items) { him = c3_item_h(g.nseg, g.nbands, g.rps, g.H, hit, WB); open_item(); }
      }
    };
    while (C < g.RA && C < E) issue();
    int e0 = 0;
    for (int it = blockIdx.x; it < g.items; it += G) {
      const C3Item im = c3_item_h(g.nseg, g.nbands, g.rps, g.H, it, WB);
      const int K = im.r1 - im.r0 + 2;
      for (int k0 = 0; k0 < K; k0 += 3) {
        const int n = K - k0 < 3 ? K - k0 : 3;
        wait_vm_dyn(pcw * (C - e0 - n));
        raw_barrier();
        e0 += n;
        while (C < e0 + g.RA && C < E) issue();       // into slots of entries < e0 - n - 2: the consumers read back to there
      }
    }
    raw_barrier();          // the consumers are done with the ring (they reuse it to flush)
    return;
  }

  // -------------------------------------------------------------------- consumers
  MDS_SETPRIO(2);
  const int i = lane & 15, q = lane >> 4;
  // byte offsets (inside a ring entry) of this lane's 8-byte share of the transposing reads: 4 pixels x 16 channels per 16-lane group
  int xo[3][2][CI], yo[2][COW];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c = 0; c < CI; ++c) {
        const int p = kx + 4 * q + 16 * h + (i >> 2);           // band pixel 0 = column x0 - 1; k index 8 q + 4 h + e <-> pixel 4 q + 16 h + e (any
                                                                // pixel <-> k map serves, as long as both operands use it): a 32-lane LDS cycle touches 8 CONSECUTIVE pixels
        xo[kx][h][c] = p * PPX * 16 + c3w_rot<CI>(c, p) * 32 + 8 * (i & 3);
      }
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int f = 0; f < COW; ++f) {
      const int p = 4 * q + 16 * h + (i >> 2);
      yo[h][f] = XSP * 16 + p * PPY * 16 + c3w_rot<PPY / 2>(wave * COW + f, p) * 32 + 8 * (i & 3);
    }
  f32x4 acc[9][CI][COW];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < CI; ++c)
#pragma unroll
      for (int f = 0; f < COW; ++f) acc[t][c][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int sp = 0, sc = 0, sn = g.NR > 1 ? 1 : 0;           // ring slots of entries k - 1, k, k + 1
  for (int it = blockIdx.x; it < g.items; it += G) {
    const C3Item im = c3_item_h(g.nseg, g.nbands, g.rps, g.H, it, WB);
    const int K = im.r1 - im.r0 + 2;
    for (int k0 = 0; k0 < K; k0 += 3) {
      const int n = K - k0 < 3 ? K - k0 : 3;
      asm volatile("" ::: "memory");
      raw_barrier();                                  // entries k0 .. k0 + n - 1 have landed
      asm volatile("" ::: "memory");
      // input row k needs dy entries k - 1 .. k + 1: rows up to k0 + n - 2 now, the item's last row with its last batch
      const int klo = k0 == 0 ? 0 : k0 - 1, khi = k0 + n == K ? K - 1 : k0 + n - 2;
      for (int k = klo; k <= khi; ++k) {
        // all of the row's fragments go out before its first MFMA: one LDS round trip per row, which the SIMD's other consumer wave
        // fills with its MFMAs (NCW = 8: two per SIMD)
        const char* const er = smem + sc * ROWB;
        u16x8 af[3][CI], bf[3][COW];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int c = 0; c < CI; ++c) {
            u16x4 lo, hi;
            if (C3_ABL & 1024) { lo = (u16x4){(uint16_t)xo[kx][0][c], 1, 2, 3}; hi = lo; }
            else { lo = lds_tr4((const bf16_t*)(er + xo[kx][0][c])); hi = lds_tr4((const bf16_t*)(er + xo[kx][1][c])); }
            af[kx][c] = (u16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          if ((ky == 0 && k == K - 1) || (ky == 2 && k == 0)) continue;      // (wave-uniform) no entry below / above inside the item
          const char* const yr = smem + (ky == 0 ? sn : (ky == 1 ? sc : sp)) * ROWB;
#pragma unroll
          for (int f = 0; f < COW; ++f) {
            u16x4 lo, hi;
            if (C3_ABL & 1024) { lo = (u16x4){(uint16_t)yo[0][f], 1, 2, 3}; hi = lo; }
            else { lo = lds_tr4((const bf16_t*)(yr + yo[0][f])); hi = lds_tr4((const bf16_t*)(yr + yo[1][f])); }
            bf[ky][f] = (u16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          if ((ky == 0 && k == K - 1) || (ky == 2 && k == 0)) continue;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int c = 0; c < CI; ++c)
#pragma unroll
              for (int f = 0; f < COW; ++f) {
                if (C3_ABL & 512) acc[3 * ky + kx][c][f][0] += bf2f(bf[ky][f][0]) + bf2f(af[kx][c][1]);
                else mma16(bf[ky][f], af[kx][c], acc[3 * ky + kx][c][f]);      // acc[r] = dw[co = 4 q + r][ci = i]
              }
        }
        sp = sc; sc = sn; sn = sn + 1 == g.NR ? 0 : sn + 1;
      }
    }
  }
  asm volatile("" ::: "memory");
  raw_barrier();                                      // every wave is past its last ring read: the ring becomes the flush staging area
  asm volatile("" ::: "memory");
  // flush: a wave's 16-channel slab in the parameter's OIHW order through its own LDS region, then coalesced atomics
  constexpr int SLAB = CIN * 9;                        // floats per output channel (wtaps == 9)
  float* const fl = (float*)smem + (wave & 3) * 16 * SLAB;
  for (int rd = 0; rd < (NCW + 3) / 4; ++rd) {          // four waves' staging regions fit the LDS: the waves flush in rounds of four
    if ((wave >> 2) == rd) {
#pragma unroll
      for (int f = 0; f < COW; ++f) {
        wave_lds_sync();
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int c = 0; c < CI; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) fl[(4 * q + r) * SLAB + (16 * c + i) * 9 + g.tapw[t]] = acc[t][c][f][r];
        wave_lds_sync();
        float* const dst = g.dw + (long)(c0 + 16 * (wave * COW + f)) * SLAB;
        // (every block adds to the same 9 CIN COP addresses: each starts somewhere else)
        const int e0 = (int)((blockIdx.x * 37u) % (16 * SLAB / 64)) * 64;
        if (!(C3_ABL & 256)) for (int e = lane; e < 16 * SLAB; e += 64) { const int ee = e + e0 < 16 * SLAB ? e + e0 : e + e0 - 16 * SLAB; atomicAdd(dst + ee, fl[ee]); }
        else if (fl[lane] == 12345.678f) dst[lane] = 1.f;
      }
    }
    if (rd + 1 < (NCW + 3) / 4) { wait_lgkm0(); raw_barrier(); }
  }
}

// The same for blocks.0.0 (32 -> 16 channels at 368 x 640, input = the stem's RAW output read through BatchNorm + SiLU): a single
// 16-channel output fragment, so the four consumer waves split the 64-column band's two 32-pixel k-steps and the two 16-channel
// input fragments instead (9 accumulators each, summed through LDS at the end), and NTW transform waves rewrite the input part of
// every landed entry in place - silu(scale * v + shift), zero outside the image - one batch ahead of the consumers, which run one
// barrier behind (as c3_kernel's NTW form; every role executes batches + 2 barriers).
#ifndef C3WP_BATCH
#define C3WP_BATCH 3      /* (6 measured the same: 166 vs 168 us - the barrier rate is not what keeps the three stages from overlapping) */
#endif
template <int NPW, int NTW>
__global__ __launch_bounds__(64 * (4 + NPW + NTW)) void c3wp_kernel(C3WArgs g) {
  constexpr int CIN = 32, PPX = 4, PPY = 2, WB = 64, BT = C3WP_BATCH;      // BT ring entries per barrier
  constexpr int XS = (WB + 2) * PPX, XSP = (XS + 63) / 64 * 64, YS = WB * PPY, RS = XSP + YS, PIECES = RS / 64, ROWB = RS * 16;
  static_assert(YS % 64 == 0, "dy rows end on a piece boundary");
  MDS_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);
  const int G = gridDim.x;
  int E = 0;
  for (int it = blockIdx.x; it < g.items; it += G) {
    const C3Item im = c3_item_h(g.nseg, g.nbands, g.rps, g.H, it, WB);
    E += im.r1 - im.r0 + 2;
  }
  if (wave >= 4 && wave < 4 + NPW) {
    // ------------------------------------------------------------------ DMA waves (as c3w_kernel)
    MDS_SETPRIO(3);
    const int pw = wave - 4;
    constexpr int PCWMAX = (PIECES + NPW - 1) / NPW;
    const int pcw = (PIECES - pw + NPW - 1) / NPW;
    int dcol[PCWMAX], eoff[PCWMAX];
#pragma unroll
    for (int j = 0; j < PCWMAX; ++j) {
      const int sg_ = 64 * (pw + NPW * j) + lane;
      if (sg_ < XS) {
        const int p = sg_ / PPX, t = sg_ - p * PPX;
        dcol[j] = p - 1; eoff[j] = 16 * c3w_unrot<2>(t >> 1, p) + 8 * (t & 1);
      } else if (sg_ >= XSP && sg_ < RS) {
        const int u = sg_ - XSP, p = u / PPY, t = u - p * PPY;
        dcol[j] = p; eoff[j] = 8 * t;
      } else {
        dcol[j] = -(1 << 30); eoff[j] = 0;
      }
    }
    const lds_t ring = lds_addr_of(smem);
    int hit = blockIdx.x, hk = 0;
    C3Item him = c3_item_h(g.nseg, g.nbands, g.rps, g.H, hit < g.items ? hit : 0, WB);
    int C = 0, hslot = 0;
    const char* cur[PCWMAX];
    unsigned step[PCWMAX];
    auto open_item = [&]() {
      const char* xrow0 = (const char*)g.x + ((long)him.n * g.H + him.r0 - 1) * g.W * CIN * 2;
      const char* yrow0 = (const char*)g.dy + ((long)him.n * g.H + him.r0 - 1) * g.W * g.Ctot * 2;
#pragma unroll
      for (int j = 0; j < PCWMAX; ++j) {
        const bool isy = 64 * (pw + NPW * j) >= XSP;
        const int gx = him.x0 + dcol[j];
        const bool ok = gx >= 0 && gx < g.W;
        cur[j] = ok ? (isy ? yrow0 : xrow0) + (gx * (isy ? g.Ctot : CIN) + eoff[j]) * 2 : (const char*)c3_zero_page;
        step[j] = ok ? (unsigned)(g.W * (isy ? g.Ctot : CIN) * 2) : 0u;
      }
    };
    open_item();
    auto issue = [&]() {
      const int ri = him.r0 - 1 + hk;
      const bool xok = ri >= 0 && ri < g.H, yok = ri >= him.r0 && ri < him.r1;
      const lds_t dst = ring + (lds_t)(hslot * ROWB);
      hslot = hslot + 1 == g.NR ? 0 : hslot + 1;
#pragma unroll
      for (int j = 0; j < PCWMAX; ++j) {
        const int pi = pw + NPW * j;
        if (pi < PIECES) {
          const bool isy = 64 * pi >= XSP;
          const char* src = (isy ? yok : xok) ? cur[j] : (const char*)c3_zero_page;
          if (dcol[j] > -(1 << 29)) glds16(src, dst + (lds_t)(pi * 1024));
          cur[j] += step[j];
        }
      }
      ++C;
      if (++hk == him.r1 - him.r0 + 2) {
        hk = 0; hit += G;
        if (hit < g.items) { him = c3_item_h(g.nseg, g.nbands, g.rps, g.H, hit, WB); open_item(); }
      }
    };
    while (C < g.RA && C < E) issue();
    int e0 = 0;
    for (int it = blockIdx.x; it < g.items; it += G) {
      const C3Item im = c3_item_h(g.nseg, g.nbands, g.rps, g.H, it, WB);
      const int K = im.r1 - im.r0 + 2;
      for (int k0 = 0; k0 < K; k0 += BT) {
        const int n = K - k0 < BT ? K - k0 : BT;
        wait_vm_dyn(pcw * (C - e0 - n));
        raw_barrier();
        e0 += n;
        while (C < e0 + g.RA && C < E) issue();       // into slots of entries two batches back and older (ring: RA + 8)
      }
    }
    raw_barrier();
    raw_barrier();
    return;
  }
  if (wave >= 4 + NPW) {
    // ------------------------------------------------------------------ transform waves: half slots (four channels) of the input part
    MDS_SETPRIO(0);       // below the consumers: their short read - MFMA bursts must not queue behind two waves of v_exp / v_rcp per SIMD
    const int tw = wave - 4 - NPW;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // lane l of a piece: slot 32 pc + l / 2 = part (l / 2) % 4 of pixel 8 pc + l / 8; the 32-byte chunks of pixels 4 .. 7 of a piece
    // are swapped (c3w_rot<2>): the channels a lane transforms are the same in every piece
    const int tpart = (lane >> 1) & 3, ch0 = 16 * ((tpart >> 1) ^ ((lane >> 5) & 1)) + 8 * (tpart & 1) + 4 * (lane & 1);
    f32x2 sc[2], sh[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      sc[c] = (f32x2){g.pro_scale[ch0 + 2 * c], g.pro_scale[ch0 + 2 * c + 1]};
      sh[c] = (f32x2){g.pro_shift[ch0 + 2 * c], g.pro_shift[ch0 + 2 * c + 1]};
    }
    constexpr int NPC = (2 * XS + 63) / 64;
    int rslot = 0;
    for (int it = blockIdx.x; it < g.items; it += G) {
      const C3Item im = c3_item_h(g.nseg, g.nbands, g.rps, g.H, it, WB);
      const int K = im.r1 - im.r0 + 2;
      for (int k0 = 0; k0 < K; k0 += BT) {
        const int n = K - k0 < BT ? K - k0 : BT;
        asm volatile("" ::: "memory");
        raw_barrier();
        asm volatile("" ::: "memory");
        constexpr int MAXP = (BT * NPC + NTW - 1) / NTW;
        u16x4 v[MAXP];
        char* ptr[MAXP];
        int edge[MAXP];
#pragma unroll
        for (int qq = 0; qq < MAXP; ++qq) {
          const int u = tw + qq * NTW;
          edge[qq] = -1;
          if (u < n * NPC) {
            const int j = u / NPC, pc = u - j * NPC;
            const int ri = im.r0 - 1 + k0 + j, px0 = im.x0 - 1 + 8 * pc;
            int rs_ = rslot + j;
            if (rs_ >= g.NR) rs_ -= g.NR;
            const int h = 64 * pc + lane;
            ptr[qq] = smem + rs_ * ROWB + (h < 2 * XS ? h : 0) * 8;
            if (!(C3_ABL & 32)) v[qq] = *(const u16x4*)ptr[qq];
            edge[qq] = (ri >= 0 && ri < g.H && px0 >= 0 && px0 + 8 <= g.W) ? 0 : 1;
          }
        }
#pragma unroll
        for (int qq = 0; qq < MAXP; ++qq) {
          if (edge[qq] < 0 || (C3_ABL & 32)) continue;
          if (C3_ABL & 16) { *(u16x4*)ptr[qq] = v[qq]; continue; }
          const int u = tw + qq * NTW;
          const int j = u / NPC, pc = u - j * NPC;
          f32x2 z[2], e[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            z[c] = (f32x2){bf2f(v[qq][2 * c]), bf2f(v[qq][2 * c + 1])} * sc[c] + sh[c];
            const f32x2 t = z[c] * -1.4426950408889634f;
            e[c] = (f32x2){fast_exp2(t[0]), fast_exp2(t[1])} + 1.0f;
            z[c] *= (f32x2){fast_rcp(e[c][0]), fast_rcp(e[c][1])};
          }
          if (edge[qq]) {
            const int ri = im.r0 - 1 + k0 + j, gx = im.x0 - 1 + 8 * pc + (lane >> 3);
            const float okf = (ri >= 0 && ri < g.H && gx >= 0 && gx < g.W) ? 1.f : 0.f;
            z[0] *= okf; z[1] *= okf;
          }
          typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
          if (64 * pc + lane < 2 * XS) *(u32x2_*)ptr[qq] = (u32x2_){pack2(z[0][0], z[0][1]), pack2(z[1][0], z[1][1])};
        }
        rslot += n;
        if (rslot >= g.NR) rslot -= g.NR;
        wait_lgkm0();
      }
    }
    raw_barrier();
    raw_barrier();
    return;
  }

  // -------------------------------------------------------------------- consumers: wave = (32-pixel k-step of the band, 16-channel input fragment)
  MDS_SETPRIO(3);
  const int i = lane & 15, q = lane >> 4;
  const int chunk = wave >> 1, cf = wave & 1;
  int xo[3][2], yo[2];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int p = kx + 32 * chunk + 4 * q + 16 * h + (i >> 2);
      xo[kx][h] = p * PPX * 16 + c3w_rot<2>(cf, p) * 32 + 8 * (i & 3);
    }
#pragma unroll
  for (int h = 0; h < 2; ++h) yo[h] = XSP * 16 + (32 * chunk + 4 * q + 16 * h + (i >> 2)) * PPY * 16 + 8 * (i & 3);
  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int sp = 0, sc_ = 0, sn = g.NR > 1 ? 1 : 0;
  raw_barrier();                                      // one barrier behind the ring: batch j is consumed after barrier j + 1
  for (int it = blockIdx.x; it < g.items; it += G) {
    const C3Item im = c3_item_h(g.nseg, g.nbands, g.rps, g.H, it, WB);
    const int K = im.r1 - im.r0 + 2;
    for (int k0 = 0; k0 < K; k0 += BT) {
      const int n = K - k0 < BT ? K - k0 : BT;
      asm volatile("" ::: "memory");
      raw_barrier();                                  // batch k0's input rows are transformed; the entry after it has landed
      asm volatile("" ::: "memory");
      for (int k = k0; k < k0 + n; ++k) {
        if (C3_ABL & 1024) continue;
        const char* const er = smem + sc_ * ROWB;
        u16x8 af[3], bf[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const u16x4 lo = lds_tr4((const bf16_t*)(er + xo[kx][0])), hi = lds_tr4((const bf16_t*)(er + xo[kx][1]));
          af[kx] = (u16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          if ((ky == 0 && k == K - 1) || (ky == 2 && k == 0)) continue;
          const char* const yr = smem + (ky == 0 ? sn : (ky == 1 ? sc_ : sp)) * ROWB;
          const u16x4 lo = lds_tr4((const bf16_t*)(yr + yo[0])), hi = lds_tr4((const bf16_t*)(yr + yo[1]));
          bf[ky] = (u16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          if ((ky == 0 && k == K - 1) || (ky == 2 && k == 0)) continue;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            if (C3_ABL & 512) acc[3 * ky + kx][0] += bf2f(bf[ky][0]) + bf2f(af[kx][1]);
            else mma16(bf[ky], af[kx], acc[3 * ky + kx]);      // acc[r] = dw[co = 4 q + r][ci = 16 cf + i]
          }
        }
        sp = sc_; sc_ = sn; sn = sn + 1 == g.NR ? 0 : sn + 1;
      }
    }
  }
  asm volatile("" ::: "memory");
  raw_barrier();                                      // (the other roles' last barrier) the ring becomes the staging area
  asm volatile("" ::: "memory");
  constexpr int SLAB = CIN * 9;
  float* const fl = (float*)smem;                     // [16 output channels][CIN][9]: the four waves' sums meet here
  for (int e = tid; e < 16 * SLAB; e += 256) fl[e] = 0.f;
  wait_lgkm0();
  raw_barrier();
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(&fl[(4 * q + r) * SLAB + (16 * cf + i) * 9 + g.tapw[t]], acc[t][r]);
  wait_lgkm0();
  raw_barrier();
  const int e0 = (int)((blockIdx.x * 37u) % (16 * SLAB / 256)) * 256;
  for (int e = tid; e < 16 * SLAB; e += 256) { const int ee = e + e0 < 16 * SLAB ? e + e0 : e + e0 - 16 * SLAB; atomicAdd(g.dw + ee, fl[ee]); }
}

template <int NPW, int NTW>
static int c3wp_launch(const mds_conv_wgrad_args* a, const int (&tapw)[9], mds_stream_t stream) {
  constexpr int PPX = 4, PPY = 2, WB = 64;
  constexpr int XS = (WB + 2) * PPX, XSP = (XS + 63) / 64 * 64, RS = XSP + WB * PPY, PIECES = RS / 64, ROWB = RS * 16;
  C3WArgs g;
  g.x = (const bf16_t*)a->x; g.dy = (const bf16_t*)a->dyt; g.dw = a->dw;
  g.N = a->N; g.H = a->IH; g.W = a->IW; g.Ctot = a->Cout; g.wtaps = a->wtaps;
  g.pro_scale = a->pro.scale; g.pro_shift = a->pro.shift;
  for (int t = 0; t < 9; ++t) g.tapw[t] = tapw[t];
  const int pcw = (PIECES + NPW - 1) / NPW;
  int RA = (100 * 1024 + ROWB - 1) / ROWB;      // rows in flight: the ring is the whole LDS (a batch dips it by three rows)
  while (RA > 3 && pcw * (RA - 1) > 40) --RA;
  int NR = RA + 2 * C3WP_BATCH + 2;      // the batch being consumed + the one being transformed + the entry above / below
  const size_t lds_cap = 158 * 1024;
  while ((size_t)NR * ROWB > lds_cap && RA > 3) { --RA; NR = RA + 2 * C3WP_BATCH + 2; }
  if ((size_t)NR * ROWB > lds_cap || pcw * (RA - 1) > 40) return 0;
  g.RA = RA; g.NR = NR;
  int CUS = 256;
  if (mds_knob(MDS_KNOB_CONV_BLOCKS) > 0) CUS = mds_knob(MDS_KNOB_CONV_BLOCKS);
  g.nbands = cdiv(a->IW, WB);
  long best = -1;
  for (int ns = 1; ns <= 64 && ns <= a->IH; ++ns) {
    const int rps = cdiv(a->IH, ns), nsr = cdiv(a->IH, rps);
    const long items = (long)a->N * g.nbands * nsr;
    const long per = (items + CUS - 1) / CUS;
    const long cost = per * (rps + 2 + 2);
    if (best < 0 || cost < best) { best = cost; g.nseg = nsr; g.rps = rps; g.items = (int)items; }
  }
  const int grid = g.items < CUS ? g.items : CUS;
  size_t smem = (size_t)NR * ROWB;
  if (smem < (size_t)16 * 32 * 9 * 4) smem = (size_t)16 * 32 * 9 * 4;
  MDS_LAUNCH((c3wp_kernel<NPW, NTW>), dim3(grid), dim3(64 * (4 + NPW + NTW)), smem, stream, g);
  return 1;
}

template <int CIN, int COW, int NCW, int NPW>
static int c3w_launch(const mds_conv_wgrad_args* a, const int (&tapw)[9], mds_stream_t stream) {
  constexpr int COP = 16 * COW * NCW, PPX = CIN / 8, PPY = COP / 8, WB = 32;
  constexpr int XS = (WB + 2) * PPX, XSP = (XS + 63) / 64 * 64, RS = XSP + WB * PPY, PIECES = RS / 64, ROWB = RS * 16;
  if (a->Cout % COP) return 0;
  C3WArgs g;
  g.x = (const bf16_t*)a->x; g.dy = (const bf16_t*)a->dyt; g.dw = a->dw;
  g.N = a->N; g.H = a->IH; g.W = a->IW; g.Ctot = a->Cout; g.wtaps = a->wtaps; g.pro_scale = nullptr; g.pro_shift = nullptr;
  for (int t = 0; t < 9; ++t) g.tapw[t] = tapw[t];
  const int passes = a->Cout / COP;
  const int pcw = (PIECES + NPW - 1) / NPW;
  int RA = (80 * 1024 + ROWB - 1) / ROWB;
  if (RA < 3) RA = 3;
  while (RA > 3 && pcw * (RA - 1) > 40) --RA;
  if (pcw * (RA - 1) > 40) return 0;
  int NR = RA + 6;                                     // the consumers read back to entry k0 - 2 of the batch (+ one of slack)
  const size_t lds_cap = 158 * 1024;
  while ((size_t)NR * ROWB > lds_cap && RA > 3) { --RA; NR = RA + 6; }
  if ((size_t)NR * ROWB > lds_cap) return 0;
  g.RA = RA; g.NR = NR;
  int CUS = 256 / passes;
  if (mds_knob(MDS_KNOB_CONV_BLOCKS) > 0) CUS = mds_knob(MDS_KNOB_CONV_BLOCKS);
  g.nbands = cdiv(a->IW, WB);
  long best = -1;
  for (int ns = 1; ns <= 64 && ns <= a->IH; ++ns) {
    const int rps = cdiv(a->IH, ns), nsr = cdiv(a->IH, rps);
    const long items = (long)a->N * g.nbands * nsr;
    const long per = (items + CUS - 1) / CUS;
    const long cost = per * (rps + 2 + 2);
    if (best < 0 || cost < best) { best = cost; g.nseg = nsr; g.rps = rps; g.items = (int)items; }
  }
  const int grid = g.items < CUS ? g.items : CUS;
  size_t smem = (size_t)NR * ROWB;
  const size_t flush = (size_t)4 * 16 * CIN * 9 * 4;
  if (smem < flush) smem = flush;
  dim3 block(64 * (NCW + NPW));
  MDS_LAUNCH((c3w_kernel<CIN, COW, NCW, NPW>), dim3(grid, passes), block, smem, stream, g);
  return 1;
}

// The same for a STRIDE-2 layer (TF-SAME, even extents: blocks.2.0, 32 -> 128): dw[co][ci][ky][kx] += sum dy[r][x][co] in[2 r + ky][2 x + kx][ci].
// A ring entry is dy row r of a 32-column band with ITS two input rows 2 r and 2 r + 1 (65 pixels each): the even row meets dy row r
// (ky = 0) and dy row r - 1 (ky = 2, from the entry before - no look-ahead), the odd row meets dy row r (ky = 1); an item ends with one
// more entry that carries only the even row 2 r1.  Fragment lane (pixel) p reads input pixel 2 p + kx: a 2-way bank conflict
// whatever the chunk rotation (the pixel stride is half the bank period), on a kernel that moves 18 KB per 18 MFMAs per wave.
template <int NPW>
__global__ __launch_bounds__(64 * (8 + NPW)) void c3w2_kernel(C3WArgs g) {      // g.H, g.W: input extents; the dy tensor is H / 2 x W / 2
  constexpr int CIN = 32, CI = 2, COP = 128, NCW = 8, PPX = 4, PPY = 16, WB = 32;
  constexpr int XS = (2 * WB + 1) * PPX, XSP = (XS + 63) / 64 * 64, YS = WB * PPY, RS = 2 * XSP + YS, PIECES = RS / 64, ROWB = RS * 16;
  MDS_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = MDS_UNIFORM(tid >> 6);
  const int G = gridDim.x, OH = g.H / 2, OW = g.W / 2;
  const int c0 = blockIdx.y * COP;
  int E = 0;
  for (int it = blockIdx.x; it < g.items; it += G) {
    const C3Item im = c3_item_h(g.nseg, g.nbands, g.rps, OH, it, WB);
    E += im.r1 - im.r0 + 1;
  }
  if (wave >= NCW) {
    // ------------------------------------------------------------------ DMA waves
    MDS_SETPRIO(3);
    const int pw = wave - NCW;
    constexpr int PCWMAX = (PIECES + NPW - 1) / NPW;
    const int pcw = (PIECES - pw + NPW - 1) / NPW;
    int dcol[PCWMAX], eoff[PCWMAX];
#pragma unroll
    for (int j = 0; j < PCWMAX; ++j) {
      const int sg_ = 64 * (pw + NPW * j) + lane;
      const int reg = sg_ < XSP ? 0 : (sg_ < 2 * XSP ? 1 : 2), u = sg_ - reg * XSP;       // even input row | odd input row | dy row
      if (reg < 2 && u < XS) {
        const int p = u / PPX, t = u - p * PPX;
        dcol[j] = p; eoff[j] = 16 * c3w_unrot<CI>(t >> 1, p) + 8 * (t & 1);
      } else if (reg == 2) {
        const int p = u / PPY, t = u - p * PPY;
        dcol[j] = p; eoff[j] = 16 * c3w_unrot<PPY / 2>(t >> 1, p) + 8 * (t & 1);
      } else {
        dcol[j] = -(1 << 30); eoff[j] = 0;
      }
    }
    const lds_t ring = lds_addr_of(smem);
    int hit = blockIdx.x, hk = 0;
    C3Item him = c3_item_h(g.nseg, g.nbands, g.rps, OH, hit < g.items ? hit : 0, WB);
    int C = 0, hslot = 0;
    const char* cur[PCWMAX];
    unsigned step[PCWMAX];
    auto open_item = [&]() {
      const char* xrow0 = (const char*)g.x + ((long)him.n * g.H + 2 * him.r0) * g.W * CIN * 2;
      const char* yrow0 = (const char*)(g.dy + c0) + ((long)him.n * OH + him.r0) * OW * g.Ctot * 2;
#pragma unroll
      for (int j = 0; j < PCWMAX; ++j) {
        const int reg = 64 * (pw + NPW * j) < XSP ? 0 : (64 * (pw + NPW * j) < 2 * XSP ? 1 : 2);      // wave-uniform
        const int gx = reg == 2 ? him.x0 + dcol[j] : 2 * him.x0 + dcol[j];
        const bool ok = gx >= 0 && gx < (reg == 2 ? OW : g.W);
        cur[j] = !ok ? (const char*)c3_zero_page
                     : (reg == 2 ? yrow0 + (gx * g.Ctot + eoff[j]) * 2 : xrow0 + (long)reg * g.W * CIN * 2 + (gx * CIN + eoff[j]) * 2);
        step[j] = ok ? (unsigned)(reg == 2 ? OW * g.Ctot * 2 : 2 * g.W * CIN * 2) : 0u;
      }
    };
    open_item();
    auto issue = [&]() {
      const int R = him.r1 - him.r0, xe = 2 * (him.r0 + hk);
      const bool eok = xe < g.H, ook = hk < R && xe + 1 < g.H, yok = hk < R;      // (the closing entry: the even row only; row H: the pad)
      const lds_t dst = ring + (lds_t)(hslot * ROWB);
      hslot = hslot + 1 == g.NR ? 0 : hslot + 1;
#pragma unroll
      for (int j = 0; j < PCWMAX; ++j) {
        const int pi = pw + NPW * j;
        if (pi < PIECES) {
          const int reg = 64 * pi < XSP ? 0 : (64 * pi < 2 * XSP ? 1 : 2);
          const char* src = (reg == 0 ? eok : (reg == 1 ? ook : yok)) ? cur[j] : (const char*)c3_zero_page;
          if (dcol[j] > -(1 << 29)) glds16(src, dst + (lds_t)(pi * 1024));
          cur[j] += step[j];
        }
      }
      ++C;
      if (++hk == R + 1) {
        hk = 0; hit += G;
        if (hit < g.items) { him = c3_item_h(g.nseg, g.nbands, g.rps, OH, hit, WB); open_item(); }
      }
    };
    while (C < g.RA && C < E) issue();
    int e0 = 0;
    for (int it = blockIdx.x; it < g.items; it += G) {
      const C3Item im = c3_item_h(g.nseg, g.nbands, g.rps, OH, it, WB);
      const int K = im.r1 - im.r0 + 1;
      for (int k0 = 0; k0 < K; k0 += 2) {
        const int n = K - k0 < 2 ? K - k0 : 2;
        wait_vm_dyn(pcw * (C - e0 - n));
        raw_barrier();
        e0 += n;
        while (C < e0 + g.RA && C < E) issue();       // into slots of entries < e0 - n - 1 (ring: RA + 3)
      }
    }
    raw_barrier();
    return;
  }

  // -------------------------------------------------------------------- consumers: one 16-channel output fragment each
  MDS_SETPRIO(2);
  const int i = lane & 15, q = lane >> 4;
  int xo[2][3][2][CI], yo[2];
#pragma unroll
  for (int ro = 0; ro < 2; ++ro)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < CI; ++c) {
          const int p = 2 * (4 * q + 16 * h + (i >> 2)) + kx;      // output pixel 4 q + 16 h + e <-> k index 8 q + 4 h + e (as c3w_kernel)
          xo[ro][kx][h][c] = ro * XSP * 16 + p * PPX * 16 + c3w_rot<CI>(c, p) * 32 + 8 * (i & 3);
        }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int p = 4 * q + 16 * h + (i >> 2);
    yo[h] = 2 * XSP * 16 + p * PPY * 16 + c3w_rot<PPY / 2>(wave, p) * 32 + 8 * (i & 3);
  }
  f32x4 acc[9][CI];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < CI; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int sp = 0, sc = 0;
  for (int it = blockIdx.x; it < g.items; it += G) {
    const C3Item im = c3_item_h(g.nseg, g.nbands, g.rps, OH, it, WB);
    const int K = im.r1 - im.r0 + 1;
    for (int k0 = 0; k0 < K; k0 += 2) {
      const int n = K - k0 < 2 ? K - k0 : 2;
      asm volatile("" ::: "memory");
      raw_barrier();
      asm volatile("" ::: "memory");
      for (int k = k0; k < k0 + n; ++k) {
        const char* const er = smem + sc * ROWB;
        const char* const pr = smem + sp * ROWB;
        auto frag = [&](const char* base, int o0, int o1) {
          const u16x4 lo = lds_tr4((const bf16_t*)(base + o0)), hi = lds_tr4((const bf16_t*)(base + o1));
          return (u16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        u16x8 ae[3][CI], ao[3][CI], bc, bp;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int c = 0; c < CI; ++c) ae[kx][c] = frag(er, xo[0][kx][0][c], xo[0][kx][1][c]);
        if (k < K - 1) {                                   // (wave-uniform) a dy row of the item: its odd input row too
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int c = 0; c < CI; ++c) ao[kx][c] = frag(er, xo[1][kx][0][c], xo[1][kx][1][c]);
          bc = frag(er, yo[0], yo[1]);
        }
        if (k > 0) bp = frag(pr, yo[0], yo[1]);
        if (k < K - 1) {
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int c = 0; c < CI; ++c) { mma16(bc, ae[kx][c], acc[kx][c]); mma16(bc, ao[kx][c], acc[3 + kx][c]); }      // acc[r] = dw[co = 4 q + r][ci = i]
        }
        if (k > 0) {
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int c = 0; c < CI; ++c) mma16(bp, ae[kx][c], acc[6 + kx][c]);
        }
        sp = sc; sc = sc + 1 == g.NR ? 0 : sc + 1;
      }
    }
  }
  asm volatile("" ::: "memory");
  raw_barrier();
  asm volatile("" ::: "memory");
  constexpr int SLAB = CIN * 9;
  float* const fl = (float*)smem + (wave & 3) * 16 * SLAB;
  for (int rd = 0; rd < 2; ++rd) {
    if ((wave >> 2) == rd) {
      wave_lds_sync();
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < CI; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) fl[(4 * q + r) * SLAB + (16 * c + i) * 9 + g.tapw[t]] = acc[t][c][r];
      wave_lds_sync();
      float* const dst = g.dw + (long)(c0 + 16 * wave) * SLAB;
      const int e0 = (int)((blockIdx.x * 37u) % (16 * SLAB / 64)) * 64;
      for (int e = lane; e < 16 * SLAB; e += 64) { const int ee = e + e0 < 16 * SLAB ? e + e0 : e + e0 - 16 * SLAB; atomicAdd(dst + ee, fl[ee]); }
    }
    if (rd == 0) { wait_lgkm0(); raw_barrier(); }
  }
}

static int c3w2_try(const mds_conv_wgrad_args* a, mds_stream_t stream) {
  constexpr int NPW = 3, WB = 32, XSP = ((2 * WB + 1) * 4 + 63) / 64 * 64, RS = 2 * XSP + WB * 16, PIECES = RS / 64, ROWB = RS * 16;
  if (a->Cin != 32 || a->Cout != 128 || a->pro.mode != MDS_PRO_NONE || (mds_knob(MDS_KNOB_C3_DBG) & 512)) return 0;
  if (a->IH % 2 || a->IW % 2 || a->OH != a->IH / 2 || a->OW != a->IW / 2) return 0;
  if ((long)a->IH * a->IW * a->Cin >= (1L << 30) || (long)a->OH * a->OW * a->Cout >= (1L << 30)) return 0;
  int tapw[9];
  for (int t = 0; t < 9; ++t) tapw[t] = -1;
  for (int t = 0; t < 9; ++t) {
    if (a->dy[t] < 0 || a->dy[t] > 2 || a->dx[t] < 0 || a->dx[t] > 2) return 0;
    tapw[3 * a->dy[t] + a->dx[t]] = a->wi[t];
  }
  for (int t = 0; t < 9; ++t) if (tapw[t] < 0 || tapw[t] >= 9) return 0;
  if ((long)a->N * a->OH * a->OW < 16384 && mds_knob(MDS_KNOB_C3) != 2) return 0;
  C3WArgs g;
  g.x = (const bf16_t*)a->x; g.dy = (const bf16_t*)a->dyt; g.dw = a->dw;
  g.N = a->N; g.H = a->IH; g.W = a->IW; g.Ctot = a->Cout; g.wtaps = a->wtaps; g.pro_scale = nullptr; g.pro_shift = nullptr;
  for (int t = 0; t < 9; ++t) g.tapw[t] = tapw[t];
  const int pcw = (PIECES + NPW - 1) / NPW;
  int RA = 5;
  while (RA > 2 && ((size_t)(RA + 3) * ROWB > 158 * 1024 || pcw * (RA - 1) > 40)) --RA;
  if ((size_t)(RA + 3) * ROWB > 158 * 1024) return 0;
  g.RA = RA; g.NR = RA + 3;
  int CUS = 256;
  if (mds_knob(MDS_KNOB_CONV_BLOCKS) > 0) CUS = mds_knob(MDS_KNOB_CONV_BLOCKS);
  g.nbands = cdiv(a->OW, WB);
  long best = -1;
  for (int ns = 1; ns <= 64 && ns <= a->OH; ++ns) {
    const int rps = cdiv(a->OH, ns), nsr = cdiv(a->OH, rps);
    const long items = (long)a->N * g.nbands * nsr;
    const long per = (items + CUS - 1) / CUS;
    const long cost = per * (rps + 1 + 2);
    if (best < 0 || cost < best) { best = cost; g.nseg = nsr; g.rps = rps; g.items = (int)items; }
  }
  const int grid = g.items < CUS ? g.items : CUS;
  MDS_LAUNCH((c3w2_kernel<NPW>), dim3(grid, 1), dim3(64 * (8 + NPW)), (size_t)g.NR * ROWB, stream, g);
  return 1;
}

// mds_conv_wgrad's large prologue-free stride-1 bf16 launches; 1 = launched, 0 = not one of these (k_conv.hip's kernel)
int c3w_try(const mds_conv_wgrad_args* a, mds_stream_t stream) {
  if (mds_knob(MDS_KNOB_C3) == 1 || (mds_knob(MDS_KNOB_C3_DBG) & 128)) return 0;
  if (a->dtype == MDS_BF16 && a->is == 2 && a->ntaps == 9 && a->wtaps == 9) return c3w2_try(a, stream);
  const bool pro = a->pro.mode == MDS_PRO_BN_SILU;
  if (a->dtype != MDS_BF16 || a->is != 1 || a->ntaps != 9 || a->wtaps != 9 || (a->pro.mode != MDS_PRO_NONE && !pro)) return 0;
  if (pro && !(a->Cin == 32 && a->Cout == 16 && a->pro.scale && a->pro.shift && !(mds_knob(MDS_KNOB_C3_DBG) & 256))) return 0;
  if (a->OH != a->IH || a->OW != a->IW) return 0;
  if ((long)a->IH * a->IW * a->Cout >= (1L << 30)) return 0;
  int tapw[9];
  for (int t = 0; t < 9; ++t) tapw[t] = -1;
  for (int t = 0; t < 9; ++t) {
    if (a->dy[t] < -1 || a->dy[t] > 1 || a->dx[t] < -1 || a->dx[t] > 1) return 0;
    tapw[3 * (a->dy[t] + 1) + a->dx[t] + 1] = a->wi[t];
  }
  for (int t = 0; t < 9; ++t) if (tapw[t] < 0 || tapw[t] >= 9) return 0;
  if ((long)a->N * a->IH * a->IW < 16384 && mds_knob(MDS_KNOB_C3) != 2) return 0;
  if (pro) return c3wp_launch<3, 8>(a, tapw, stream);      // blocks.0.0: behind the stem's BatchNorm + SiLU
  if (a->Cin == 32 && a->Cout == 128) return c3w_launch<32, 1, 8, 3>(a, tapw, stream);      // blocks.1.1: eight consumer waves of one 16-channel fragment each
  if (a->Cin == 48 && a->Cout == 192) return c3w_launch<48, 1, 6, 2>(a, tapw, stream);      // blocks.2.1: two passes of 96 channels, six consumer waves
  return 0;
}

// host side ------------------------------------------------------------------------------------------------------------
template <int CIN, int NF, int NSPL, int SPW, int NPW, int NSW, bool ONE = false, int NTW = 0>
static int c3_launch(const mds_conv_fwd_args* a, const int (&tapw)[9], mds_stream_t stream) {
  typedef C3Cfg<CIN, NF, NSPL, SPW, true> CFR;
  typedef C3Cfg<CIN, NF, NSPL, SPW, false> CFN;
  typedef C3Cfg<CIN, NF, NSPL, SPW, true, true> CFP;
  typedef C3Cfg<CIN, NF, NSPL, SPW, false, true> CFQ;
  const bool res = a->residual != nullptr, stats = a->stats != nullptr, post = a->post.mode != MDS_POST_NONE;
  if (post && stats) return 0;
  const int WB = CFN::WB, rowb = post ? (res ? CFP::ROWB : CFQ::ROWB) : (res ? CFR::ROWB : CFN::ROWB);
  const int pieces = post ? (res ? CFP::PIECES : CFQ::PIECES) : (res ? CFR::PIECES : CFN::PIECES);
  C3Args g;
  g.x = (const bf16_t*)a->x; g.w = (const bf16_t*)a->w; g.y = (bf16_t*)a->y; g.res = (const bf16_t*)a->residual; g.stats = a->stats;
  g.N = a->N; g.H = a->IH; g.W = a->IW; g.OH = a->OH; g.OW = a->OW; g.wtaps = a->wtaps; g.Ctot = a->Cout;
  const int passes = a->Cout / CFN::COUT;
  for (int t = 0; t < 9; ++t) g.tapw[t] = tapw[t];
  // rows in flight: ~40 KB per CU ahead of the consumers (HBM latency x a CU's share of the bandwidth), at least 3 rows;
  // the producers' vmcnt (6 bits) carries pcw * (RA - 1) pieces
  const int pcw = (pieces + NPW - 1) / NPW;
  int RA = (64 * 1024 + rowb - 1) / rowb;
  if (RA < 3) RA = 3;
  while (RA > 3 && pcw * (RA - 1) > 40) --RA;
  if (pcw * (RA - 1) > 40) return 0;
  const int keep = (post ? 6 : 3) + (NTW ? 3 : 0);      // ring rows beyond the RA in flight: the batch being read (+ with post statistics the one before: its y rows serve the store waves)
  int NR = RA + keep;
  const size_t lds_cap = 155 * 1024 - 6 * (size_t)WB * CFN::COUT * 2 - 2 * CIN * 4;     // the staged output rows share the LDS
  while ((size_t)NR * rowb > lds_cap && RA > 3) { --RA; NR = RA + keep; }
  if ((size_t)NR * rowb > lds_cap) return 0;
  g.RA = RA; g.NR = NR; g.dbg = mds_knob(MDS_KNOB_C3_DBG); g.trace = (void*)a->epi.scale;
  g.py = (const bf16_t*)a->post.y; g.pbn = a->post.bn; g.pmask = a->post.mode == MDS_POST_MASK ? a->post.mask : nullptr; g.pstats = a->post.stats; g.pmode = a->post.mode;
  // items: bands x row segments, the segment count that minimises the longest block's rows (+2 halo rows, + a fill per item)
  int CUS = 256 / passes;
  if (!stats && mds_knob(MDS_KNOB_C3_BWD_BLOCKS) > 0) CUS = mds_knob(MDS_KNOB_C3_BWD_BLOCKS) / passes;
  if (mds_knob(MDS_KNOB_CONV_BLOCKS) > 0) CUS = mds_knob(MDS_KNOB_CONV_BLOCKS);      // tests: few blocks, many items each
  g.nbands = cdiv(a->IW, WB);
  long best = -1;
  for (int ns = 1; ns <= 64 && ns <= a->IH; ++ns) {
    const int rps = cdiv(a->IH, ns), nsr = cdiv(a->IH, rps);
    const long items = (long)a->N * g.nbands * nsr;
    const long per = (items + CUS - 1) / CUS;
    const long cost = per * (rps + 2 + 2);
    if (best < 0 || cost < best) { best = cost; g.nseg = nsr; g.rps = rps; g.items = (int)items; }
  }
  const int grid = g.items < CUS ? g.items : CUS;
  size_t smem = (size_t)NR * rowb + 6 * (size_t)WB * CFN::COUT * 2;
#ifdef C3_TRACE
  smem += 160 * 32 * 8;
#endif
  dim3 block(256 + 64 * (NPW + NSW + NTW));
  const bool masked = a->IW % WB != 0;
  g.pro_scale = a->pro.scale; g.pro_shift = a->pro.shift;
  if (NTW) {
    if (res || post || ONE) return 0;
#define C3_GOT(S, M) MDS_LAUNCH((c3_kernel<CIN, NF, NSPL, SPW, NPW, NSW, false, S, M, false, false, NTW>), dim3(grid, passes), block, smem, stream, g)
    if (stats) { if (masked) C3_GOT(true, true); else C3_GOT(true, false); }
    else { if (masked) C3_GOT(false, true); else C3_GOT(false, false); }
#undef C3_GOT
    return 1;
  }
#define C3_GO(R, S) do { if (masked) MDS_LAUNCH((c3_kernel<CIN, NF, NSPL, SPW, NPW, NSW, R, S, true, false, ONE>), dim3(grid, passes), block, smem, stream, g); \
                         else MDS_LAUNCH((c3_kernel<CIN, NF, NSPL, SPW, NPW, NSW, R, S, false, false, ONE>), dim3(grid, passes), block, smem, stream, g); } while (0)
  if (ONE) {
    if (post || stats || masked) return 0;
    if (res) MDS_LAUNCH((c3_kernel<CIN, NF, NSPL, SPW, NPW, NSW, true, false, false, false, true>), dim3(grid, passes), block, smem, stream, g);
    else MDS_LAUNCH((c3_kernel<CIN, NF, NSPL, SPW, NPW, NSW, false, false, false, false, true>), dim3(grid, passes), block, smem, stream, g);
  } else if (post) {
#define C3_GOP(R, M) MDS_LAUNCH((c3_kernel<CIN, NF, NSPL, SPW, NPW, NSW, R, false, M, true>), dim3(grid, passes), block, smem, stream, g)
    if (res) { if (masked) C3_GOP(true, true); else C3_GOP(true, false); }
    else { if (masked) C3_GOP(false, true); else C3_GOP(false, false); }
#undef C3_GOP
  } else if (res) { if (stats) return 0; C3_GO(true, false); }      // (a residual operand and statistics never meet in the network)
  else { if (stats) C3_GO(false, true); else C3_GO(false, false); }
#undef C3_GO
  return 1;
}


template <int CIN, int NF, int NSPL, int NPW, int NSW>
static int c3t_launch(const mds_conv_fwd_args* a, const int (&tapw)[9], mds_stream_t stream) {
  constexpr int PP = CIN / 8, NSG = 4 / NSPL, WBI = 16 * NSG, COUT = 16 * NF * NSPL;
  const bool post = a->post.mode != MDS_POST_NONE;
  constexpr int RSX = (WBI + 1) * PP, RPYP = (RSX + 63) / 64 * 64;
  const int RS = post ? RPYP + 2 * (2 * WBI) * (COUT / 8) : RSX, PIECES = (RS + 63) / 64, ROWB = RS * 16;
  C3Args g;
  g.x = (const bf16_t*)a->x; g.w = (const bf16_t*)a->w; g.y = (bf16_t*)a->y; g.res = nullptr; g.stats = nullptr;
  g.N = a->N; g.H = a->IH; g.W = a->IW; g.OH = a->OH; g.OW = a->OW; g.wtaps = a->wtaps; g.Ctot = a->Cout;
  for (int t = 0; t < 9; ++t) g.tapw[t] = tapw[t];
  const int passes = a->Cout / COUT;
  const int pcw = (PIECES + NPW - 1) / NPW;
  const size_t stage = 8 * (size_t)(2 * WBI) * COUT * 2;
  int RA = (64 * 1024 + ROWB - 1) / ROWB;
  if (RA < 2) RA = 2;
  while (RA > 2 && pcw * (RA - 1) > 40) --RA;
  if (pcw * (RA - 1) > 40) return 0;
  const int keep = post ? 4 : 2;
  int NR = RA + keep;
  while ((size_t)NR * ROWB + stage > 156 * 1024 && RA > 2) { --RA; NR = RA + keep; }
  if ((size_t)NR * ROWB + stage > 156 * 1024) return 0;
  g.RA = RA; g.NR = NR; g.dbg = 0; g.trace = nullptr;
  g.py = (const bf16_t*)a->post.y; g.pbn = a->post.bn; g.pmask = a->post.mode == MDS_POST_MASK ? a->post.mask : nullptr; g.pstats = a->post.stats; g.pmode = a->post.mode;
  int CUS = 256 / passes;
  if (mds_knob(MDS_KNOB_C3_BWD_BLOCKS) > 0) CUS = mds_knob(MDS_KNOB_C3_BWD_BLOCKS) / passes;
  if (mds_knob(MDS_KNOB_CONV_BLOCKS) > 0) CUS = mds_knob(MDS_KNOB_CONV_BLOCKS);
  g.nbands = cdiv(a->IW, WBI);
  long best = -1;
  for (int ns = 1; ns <= 64 && ns <= a->IH; ++ns) {
    const int rps = cdiv(a->IH, ns), nsr = cdiv(a->IH, rps);
    const long items = (long)a->N * g.nbands * nsr;
    const long per = (items + CUS - 1) / CUS;
    const long cost = per * (rps + 1 + 2);
    if (best < 0 || cost < best) { best = cost; g.nseg = nsr; g.rps = rps; g.items = (int)items; }
  }
  const int grid = g.items < CUS ? g.items : CUS;
  const size_t smem = (size_t)NR * ROWB + stage;
  dim3 block(256 + 64 * (NPW + NSW));
  const bool masked = 2 * a->IW != a->OW || a->IW % WBI != 0;
#define C3T_GO(M, P) MDS_LAUNCH((c3t_kernel<CIN, NF, NSPL, NPW, NSW, M, P>), dim3(grid, passes), block, smem, stream, g)
  if (post) { if (masked) C3T_GO(true, true); else C3T_GO(false, true); }
  else { if (masked) C3T_GO(true, false); else C3T_GO(false, false); }
#undef C3T_GO
  return 1;
}

// post statistics as these kernels take them: PLAIN, SILU, or MASK with one factor per image
static bool c3_post_ok(const mds_conv_fwd_args* a) {
  if (a->post.mode == MDS_POST_NONE) return true;
  if (a->post.mode != MDS_POST_PLAIN && a->post.mode != MDS_POST_MASK && a->post.mode != MDS_POST_SILU) return false;
  if (!a->post.y || !a->post.bn || !a->post.stats) return false;
  return a->post.mode != MDS_POST_MASK || (a->post.mask && a->post.rows_per_group == (long)a->OH * a->OW);
}

// the stride-2 data gradient as mds_conv_fwd receives it: four tap groups (one per output parity), is = 1, os = 2
static int c3t_try(const mds_conv_fwd_args* a, mds_stream_t stream) {
  if (a->dtype != MDS_BF16 || a->is != 1 || a->os != 2 || a->ngroups != 4 || a->ntaps != 9) return 0;
  if (a->pro.mode != MDS_PRO_NONE || a->epi.mode != MDS_EPI_NONE || a->residual || a->stats) return 0;
  if (!c3_post_ok(a)) return 0;
  if (a->OH != 2 * a->IH && a->OH != 2 * a->IH - 1) return 0;
  if (a->OW != 2 * a->IW && a->OW != 2 * a->IW - 1) return 0;
  if ((long)a->OH * a->OW * (a->Cin > a->Cout ? a->Cin : a->Cout) >= (1L << 30)) return 0;
  int tapw[9];
  for (int t = 0; t < 9; ++t) tapw[t] = -1;
  int t = 0;
  for (int gi = 0; gi < 4; ++gi) {
    const int py = a->g_oy0[gi], px = a->g_ox0[gi];
    if (py < 0 || py > 1 || px < 0 || px > 1) return 0;
    for (int u = 0; u < a->g_ntaps[gi]; ++u, ++t) {
      // pads 0: output 2a + py reads input a + dy with (py, dy) in {(0, 0): ky 0, (0, -1): ky 2, (1, 0): ky 1}; the same in x
      const int dy = a->dy[t], dx = a->dx[t];
      const int ky = py == 1 ? (dy == 0 ? 1 : -1) : (dy == 0 ? 0 : (dy == -1 ? 2 : -1));
      const int kx = px == 1 ? (dx == 0 ? 1 : -1) : (dx == 0 ? 0 : (dx == -1 ? 2 : -1));
      if (ky < 0 || kx < 0 || tapw[3 * ky + kx] >= 0) return 0;
      tapw[3 * ky + kx] = a->wi[t];
    }
  }
  for (int u = 0; u < 9; ++u) if (tapw[u] < 0 || tapw[u] >= a->wtaps) return 0;
  const long rows = (long)a->N * a->OH * a->OW;
  if (rows < 16384 && mds_knob(MDS_KNOB_C3) != 2) return 0;
  if (a->Cin == 128 && a->Cout == 32) return c3t_launch<128, 1, 2, 3, 1>(a, tapw, stream);
  if (a->Cin == 64 && a->Cout == 16) return c3t_launch<64, 1, 1, 2, 2>(a, tapw, stream);
  return 0;
}

template <int CIN, int NF, int NSPL, int SPW, int NPW, int NSW, int NTW>
static int c3s_launch(const mds_conv_fwd_args* a, const int (&tapw)[9], mds_stream_t stream) {
  constexpr int PP = CIN / 8, NSG = 4 / NSPL, WB = 16 * NSG * SPW, COUT = 16 * NF * NSPL;
  constexpr int RPX = (2 * WB + 1) * PP, PIECES = (RPX + 63) / 64, ROWB = RPX * 16, STGROW = WB * COUT * 2;
  if (a->Cout % COUT) return 0;
  C3Args g;
  g.x = (const bf16_t*)a->x; g.w = (const bf16_t*)a->w; g.y = (bf16_t*)a->y; g.res = nullptr; g.stats = a->stats;
  g.N = a->N; g.H = a->IH; g.W = a->IW; g.OH = a->OH; g.OW = a->OW; g.wtaps = a->wtaps; g.Ctot = a->Cout;
  const int passes = a->Cout / COUT;
  for (int t = 0; t < 9; ++t) g.tapw[t] = tapw[t];
  const int pcw = (PIECES + NPW - 1) / NPW;
  int RA = (64 * 1024 + ROWB - 1) / ROWB;
  if (RA < 4) RA = 4;
  while (RA > 4 && pcw * (RA - 1) > 40) --RA;
  if (pcw * (RA - 1) > 40) return 0;
  const int keep = 4 + (NTW ? 4 : 0);       // ring rows beyond the RA in flight: the batch being read (+ the one being transformed)
  int NR = RA + keep;
  const size_t lds_cap = 155 * 1024 - 4 * (size_t)STGROW;
  while ((size_t)NR * ROWB > lds_cap && RA > 4) { --RA; NR = RA + keep; }
  if ((size_t)NR * ROWB > lds_cap) return 0;
  g.RA = RA; g.NR = NR; g.dbg = 0; g.trace = nullptr;
  g.py = nullptr; g.pbn = nullptr; g.pmask = nullptr; g.pstats = nullptr; g.pmode = 0;
  g.pro_scale = a->pro.scale; g.pro_shift = a->pro.shift;
  int CUS = 256 / passes;
  if (mds_knob(MDS_KNOB_CONV_BLOCKS) > 0) CUS = mds_knob(MDS_KNOB_CONV_BLOCKS);
  g.nbands = cdiv(a->OW, WB);
  long best = -1;
  for (int ns = 1; ns <= 64 && ns <= a->OH; ++ns) {
    const int rps = cdiv(a->OH, ns), nsr = cdiv(a->OH, rps);
    const long items = (long)a->N * g.nbands * nsr;
    const long per = (items + CUS - 1) / CUS;
    const long cost = per * (2 * rps + 1 + 4);
    if (best < 0 || cost < best) { best = cost; g.nseg = nsr; g.rps = rps; g.items = (int)items; }
  }
  const int grid = g.items < CUS ? g.items : CUS;
  const size_t smem = (size_t)NR * ROWB + 4 * (size_t)STGROW;
  dim3 block(256 + 64 * (NPW + NSW + NTW));
  const bool masked = a->OW % WB != 0, stats = a->stats != nullptr;
#define C3S_GO(S, M) MDS_LAUNCH((c3s_kernel<CIN, NF, NSPL, SPW, NPW, NSW, S, M, NTW>), dim3(grid, passes), block, smem, stream, g)
  if (masked) {
    // (the transform-wave form runs three waves per SIMD - 168 VGPRs - and its column-masked statistics variant spills 34 of them: the
    //  network's widths are whole bands, ragged widths behind a prologue stay with k_conv.hip)
    if constexpr (NTW == 0) { if (stats) C3S_GO(true, true); else C3S_GO(false, true); }
    else return 0;
  } else { if (stats) C3S_GO(true, false); else C3S_GO(false, false); }
#undef C3S_GO
  return 1;
}

// the stride-2 forward layers (TF-SAME pads 0 / 1, even extents): taps (ky, kx) in 0..2 from the input pixel (2 oy, 2 ox)
static int c3s_try(const mds_conv_fwd_args* a, mds_stream_t stream) {
  const bool pro = a->pro.mode == MDS_PRO_BN_SILU;
  if (a->dtype != MDS_BF16 || a->os != 1 || a->ntaps != 9 || a->ngroups > 1 || a->residual || a->post.mode != MDS_POST_NONE) return 0;
  if ((a->pro.mode != MDS_PRO_NONE && !pro) || a->epi.mode != MDS_EPI_NONE || (pro && !(a->pro.scale && a->pro.shift))) return 0;
  if ((mds_knob(MDS_KNOB_C3_DBG) & 64)) return 0;      // A/B: the stride-2 forward layers through k_conv.hip
  if (a->IH % 2 || a->IW % 2 || a->OH != a->IH / 2 || a->OW != a->IW / 2 || a->A != a->OH || a->B != a->OW || a->oy0 || a->ox0) return 0;
  if ((long)a->IH * a->IW * a->Cin >= (1L << 30) || (long)a->OH * a->OW * a->Cout >= (1L << 30)) return 0;
  int tapw[9];
  for (int t = 0; t < 9; ++t) tapw[t] = -1;
  for (int t = 0; t < 9; ++t) {
    if (a->dy[t] < 0 || a->dy[t] > 2 || a->dx[t] < 0 || a->dx[t] > 2) return 0;
    tapw[3 * a->dy[t] + a->dx[t]] = a->wi[t];
  }
  for (int t = 0; t < 9; ++t) if (tapw[t] < 0 || tapw[t] >= a->wtaps) return 0;
  if ((long)a->N * a->OH * a->OW < 16384 && mds_knob(MDS_KNOB_C3) != 2) return 0;
  if (a->Cin == 32 && a->Cout == 128 && !pro) return c3s_launch<32, 2, 4, 2, 2, 2, 0>(a, tapw, stream);      // blocks.2.0
  if (a->Cin == 16 && a->Cout == 64 && !pro) return c3s_launch<16, 2, 2, 2, 2, 2, 0>(a, tapw, stream);
  if (a->Cin == 16 && a->Cout == 64 && pro) return c3s_launch<16, 1, 4, 4, 1, 1, 6>(a, tapw, stream);        // blocks.1.0: reads blocks.0.0's raw output
  return 0;
}

// 1 = launched, 0 = not a shape of this kernel (the caller goes on to k_conv.hip's kernels)
int c3_try(const mds_conv_fwd_args* a, mds_stream_t stream) {
  if (mds_knob(MDS_KNOB_C3) == 1) return 0;
  if (a->ngroups == 4) return c3t_try(a, stream);
  if (a->is == 2) return c3s_try(a, stream);
  if (a->dtype != MDS_BF16 || a->is != 1 || a->os != 1 || a->ntaps != 9 || a->ngroups > 1) return 0;
  const bool pro = a->pro.mode == MDS_PRO_BN_SILU;
  if ((a->pro.mode != MDS_PRO_NONE && !pro) || a->epi.mode != MDS_EPI_NONE || !c3_post_ok(a)) return 0;
  if (pro && (mds_knob(MDS_KNOB_C3_DBG) & 32)) return 0;      // A/B: the first 3x3 layer through k_conv.hip
  if (pro && !(a->Cin == 32 && a->Cout == 16 && !a->residual && a->post.mode == MDS_POST_NONE && a->pro.scale && a->pro.shift)) return 0;
  if (a->A != a->OH || a->B != a->OW || a->OH != a->IH || a->OW != a->IW || a->oy0 || a->ox0) return 0;
  if ((long)a->IH * a->IW * (a->Cin > a->Cout ? a->Cin : a->Cout) >= (1L << 30)) return 0;
  int tapw[9];
  for (int t = 0; t < 9; ++t) tapw[t] = -1;
  for (int t = 0; t < 9; ++t) {
    if (a->dy[t] < -1 || a->dy[t] > 1 || a->dx[t] < -1 || a->dx[t] > 1) return 0;
    tapw[3 * (a->dy[t] + 1) + a->dx[t] + 1] = a->wi[t];
  }
  for (int t = 0; t < 9; ++t) if (tapw[t] < 0 || tapw[t] >= a->wtaps) return 0;
  const long rows = (long)a->N * a->IH * a->IW;
  if (rows < 16384 && mds_knob(MDS_KNOB_C3) != 2) return 0;      // small launches (inference, tests of the old kernels): k_conv.hip
  // helper waves: the data gradients read wide rows and write narrow ones (three DMA waves, one store wave); the forward layers two and two
  const bool alt = (mds_knob(MDS_KNOB_C3_DBG) & 16) != 0;      // A/B: the other split
  if (a->Cin == 128 && a->Cout == 32) return alt ? c3_launch<128, 1, 2, 1, 2, 2>(a, tapw, stream) : c3_launch<128, 1, 2, 1, 3, 1>(a, tapw, stream);
  if (pro) return c3_launch<32, 1, 1, 1, 1, 1, false, 8>(a, tapw, stream);      // the first 3x3 layer: BN + SiLU of the stem's output on the way in
  if (a->Cin == 48 && a->Cout == 192) return c3_launch<48, 2, 2, 1, 2, 2>(a, tapw, stream);      // three passes of 64 channels: a 96-channel slice per wave pair does not fit the registers
  if (a->Cin == 16 && a->Cout == 32) return c3_launch<16, 2, 1, 1, 2, 2>(a, tapw, stream);      // (one DMA wave + three store waves for the POST_SILU form: 12.89 vs 12.15 ms per step - the ring starves)
  if (a->Cin == 32 && a->Cout == 128) return alt ? c3_launch<32, 2, 4, 2, 1, 3>(a, tapw, stream) : c3_launch<32, 2, 4, 2, 2, 2>(a, tapw, stream);
  return 0;
}

// what the planner asks before it folds a BatchNorm backward's sums into a 3x3 data gradient (engine._conv_dgrad): does the
// launch - forward shape N x IH x IW x Cin -> Cout at `stride` - go to one of the kernels above that implement `post`?
extern "C" int mds_conv_dgrad_post_ok(int dtype, int N, int IH, int IW, int Cin, int Cout, int stride, int has_residual) {
  if (dtype != MDS_BF16 || mds_knob(MDS_KNOB_C3) == 1) return 0;
  if ((long)N * IH * IW < 16384 && mds_knob(MDS_KNOB_C3) != 2) return 0;
  if (stride == 1) return (Cout == 128 && Cin == 32) || (!has_residual && Cout == 16 && Cin == 32);      // c3_kernel<128 -> 32> / <16 -> 32>
  if (stride == 2) return !has_residual && IH % 2 == 0 && IW % 2 == 0 && ((Cout == 128 && Cin == 32) || (Cout == 64 && Cin == 16));
  return 0;
}

// mds_pw_fwd's large prologue-free bf16 launches (the edge-residual projections' data gradients: 0.3 - 1.2 M rows, 32 / 48 -> 64 ... 192
// channels) as 1x1 "images" of W-pixel rows through c3_kernel<..., ONE>.  1 = launched, 0 = not one of these.
int c3_pw_try(const mds_pw_fwd_args* a, mds_stream_t stream) {
  if (mds_knob(MDS_KNOB_C3) == 1 || a->dtype != MDS_BF16) return 0;
  if (a->pro.mode != MDS_PRO_NONE || a->epi.mode != MDS_EPI_NONE || a->post.mode != MDS_POST_NONE || a->stats || a->split > 1) return 0;
  if (a->M < 262144 && mds_knob(MDS_KNOB_C3) != 2) return 0;
  const bool k32 = a->K == 32 && (a->N == 64 || a->N == 128), k48 = a->K == 48 && a->N % 64 == 0 && a->N <= 192;
  if (!k32 && !k48) return 0;
  int W = 0;
  for (int w = 640; w >= 32; w -= 32) if (a->M % w == 0 && a->M / w >= 3) { W = w; break; }
  if (!W || a->M * (a->N > a->K ? a->N : a->K) >= (1L << 30)) return 0;
  mds_conv_fwd_args c = {};
  c.dtype = a->dtype; c.N = 1; c.IH = c.OH = c.A = (int)(a->M / W); c.IW = c.OW = c.B = W; c.Cin = a->K; c.Cout = a->N;
  c.os = c.is = 1; c.ntaps = 9; c.wtaps = 1; c.x = a->x; c.w = a->w; c.y = a->y; c.residual = a->residual;
  const int tapw[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (a->K == 32 && a->N == 128) return c3_launch<32, 2, 4, 2, 1, 3, true>(&c, tapw, stream);
  if (a->K == 32) return c3_launch<32, 2, 2, 1, 1, 3, true>(&c, tapw, stream);
  return c3_launch<48, 2, 2, 1, 1, 3, true>(&c, tapw, stream);
}
