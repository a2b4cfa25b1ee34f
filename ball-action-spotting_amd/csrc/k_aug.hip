// k_aug.hip — SURVEY 8(f) N3: the reference's GPU augmentation pipeline (src/ball_action/augmentations.py:7-22,
// src/augmentations.py:42-78) as fused passes over the fp32 frame batch.  See include/mds.h (mds_aug_pass).
//
// Roofline: pure HBM streaming — a sample without spatial filter is read once (bilinear gathers of a near-identity affine
// map: neighbouring lanes touch neighbouring source pixels, the re-use is served by L1/L2) and written once, 4 pixels
// (16 bytes) per lane; algorithmic bytes = 2 x 4 B per pixel (+ the same again per extra filter pass of a sample).
#include "elem.h"

// Philox-4x32-10 (Salmon et al., SC'11): counter = element index / 4, key = the job's seed
MDS_DEV void philox4(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
  uint32_t c[4] = {c0, c1, 0u, 0u};
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) out[j] = c[j];
}
MDS_DEV void normal4(uint32_t seed, unsigned long long quad, float (&z)[4]) {
  uint32_t r[4];
  philox4((uint32_t)quad, (uint32_t)(quad >> 32), seed, 0x5bd1e995u, r);
  // Box-Muller on two pairs of uniforms in (0, 1]
  const float u0 = ((float)(r[0] >> 8) + 1.0f) * (1.0f / 16777216.0f), u1 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);
  const float u2 = ((float)(r[2] >> 8) + 1.0f) * (1.0f / 16777216.0f), u3 = (float)(r[3] >> 8) * (1.0f / 16777216.0f);
  const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
  z[0] = ra * cosf(6.28318530718f * u1); z[1] = ra * sinf(6.28318530718f * u1);
  z[2] = rb * cosf(6.28318530718f * u3); z[3] = rb * sinf(6.28318530718f * u3);
}

// One block = one output tile of AUG_TR rows x AUG_TC columns of one frame; one thread = 4 consecutive pixels (one 16-byte store).
// WARP / SHARP / TAPS first stage the SOURCE box the tile needs (the affine image of the tile, or the tile plus its halo) in
// LDS with coalesced row reads - zeros outside the frame, which is exactly the padding all three stages use - and gather
// from there: a direct gather costs 4...48 scalar global loads per pixel (1.1-1.7 TB/s for a rotation, 0.45 TB/s for the
// motion blur, measured), the staged form one 16-byte load per 4 source pixels.  A source box that does not fit (a map far
// from the identity) falls back to direct gathers.
#define AUG_TR 8
#define AUG_TC 128
#define AUG_LDS_FLOATS 7168      // 28 KiB: e.g. 40 rows x 176 columns

MDS_DEV float px(const float* plane, int H, int W, int x, int y) {   // zeros outside, branch-free (clamped address, masked value)
  const bool in = (unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H;
  const int xc = x < 0 ? 0 : (x >= W ? W - 1 : x), yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
  const float v = plane[(long)yc * W + xc];
  return in ? v : 0.0f;
}

__global__ __launch_bounds__(256) void aug_kernel(mds_aug_args a) {
  __shared__ float tile[AUG_LDS_FLOATS];
  const int b = blockIdx.z, t = blockIdx.y;
  const mds_aug_job& jb = a.jobs[b];
  if (!jb.active) return;
  const int H = a.H, W = a.W, tiles_x = (W + AUG_TC - 1) / AUG_TC;
  const int ty0 = (int)(blockIdx.x / tiles_x) * AUG_TR, tx0 = (int)(blockIdx.x % tiles_x) * AUG_TC;
  const long plane_off = ((long)b * a.T + t) * H * W;
  const float* src = a.buf[jb.src] + plane_off;
  float* dst = a.buf[jb.dst] + plane_off;
  const int mode = jb.mode, tid = threadIdx.x;
  const int y = ty0 + (tid >> 5), x0 = tx0 + 4 * (tid & 31);
  float m[6] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f};
  // ---- the source box of this tile: [bx0, bx0 + bw) x [by0, by0 + bh)
  int bx0 = tx0, by0 = ty0, bw = 0, bh = 0;
  if (mode == MDS_AUG_WARP) {
#pragma unroll
    for (int j = 0; j < 6; ++j) m[j] = a.maps[((long)b * a.T + t) * 6 + j];
    const float xa = (float)tx0, xb = (float)(tx0 + AUG_TC - 1 < W - 1 ? tx0 + AUG_TC - 1 : W - 1);
    const float ya = (float)ty0, yb = (float)(ty0 + AUG_TR - 1 < H - 1 ? ty0 + AUG_TR - 1 : H - 1);
    float sxl = 1e30f, sxh = -1e30f, syl = 1e30f, syh = -1e30f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float cx = (c & 1) ? xb : xa, cy = (c & 2) ? yb : ya;
      const float sx = m[0] * cx + m[1] * cy + m[2], sy = m[3] * cx + m[4] * cy + m[5];
      sxl = fminf(sxl, sx); sxh = fmaxf(sxh, sx); syl = fminf(syl, sy); syh = fmaxf(syh, sy);
    }
    // clamp to a band around the frame (everything beyond contributes zeros anyway) so the int conversions stay defined
    sxl = fmaxf(sxl, -2.0f); syl = fmaxf(syl, -2.0f); sxh = fminf(sxh, (float)W + 1.0f); syh = fminf(syh, (float)H + 1.0f);
    bx0 = ((int)floorf(sxl)) & ~3; by0 = (int)floorf(syl);
    bw = (int)floorf(sxh) + 2 - bx0; bh = (int)floorf(syh) + 2 - by0;
  } else if (mode == MDS_AUG_SHARP) {
    bx0 = tx0 - 4; by0 = ty0 - 1; bw = AUG_TC + 8; bh = AUG_TR + 2;
  } else if (mode == MDS_AUG_TAPS) {
    bx0 = tx0 - 8; by0 = ty0 - 5; bw = AUG_TC + 16; bh = AUG_TR + 10;      // motion kernel: |dx|, |dy| <= 5 (11 x 11)
  }
  bw = (bw + 3) & ~3;
  bool staged = mode != MDS_AUG_COPY && bw > 0 && bh > 0 && (long)bw * bh <= AUG_LDS_FLOATS;
  if (mode == MDS_AUG_TAPS) {      // taps beyond the staged halo (a larger kernel than 11 x 11): direct gathers
    for (int k = 0; k < jb.ntaps; ++k) staged = staged && jb.tap_dx[k] >= -8 && jb.tap_dx[k] <= 8 && jb.tap_dy[k] >= -5 && jb.tap_dy[k] <= 5;
  }
  __shared__ float tapw[MDS_AUG_MAX_TAPS];
  __shared__ int tapo[MDS_AUG_MAX_TAPS];
  const int ntaps = mode == MDS_AUG_TAPS ? jb.ntaps : 0;
  if (staged && tid < ntaps) { tapw[tid] = jb.tap_w[tid]; tapo[tid] = jb.tap_dy[tid] * bw + jb.tap_dx[tid]; }   // (read once, not per pixel and tap)
  if (staged) {
    const int nq = (bw >> 2) * bh;
    const bool vec = (W & 3) == 0;           // rows start 16-byte aligned and bx0 is a multiple of 4
    for (int e = tid; e < nq; e += 256) {
      const int r = e / (bw >> 2), c4 = (e - r * (bw >> 2)) << 2;
      const int gy = by0 + r, gx = bx0 + c4;
      f32x4 v4 = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)gy < (unsigned)H) {
        if (vec && gx >= 0 && gx + 3 < W) v4 = *(const f32x4*)(src + (long)gy * W + gx);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) v4[j] = (unsigned)(gx + j) < (unsigned)W ? src[(long)gy * W + gx + j] : 0.f;
        }
      }
      *(f32x4*)(tile + r * bw + c4) = v4;
    }
    __syncthreads();
  }
  if (y >= H || x0 >= W) return;
  auto tap = [&](int gx, int gy) -> float {    // source pixel (gx, gy), zeros outside the frame
    if (staged) {
      const int lx = gx - bx0, ly = gy - by0;
      return ((unsigned)lx < (unsigned)bw && (unsigned)ly < (unsigned)bh) ? tile[ly * bw + lx] : 0.f;
    }
    return px(src, H, W, gx, gy);
  };
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int x = x0 + j;
    float r = 0.0f;
    if (x < W) {
      if (mode == MDS_AUG_COPY) {
        r = src[(long)y * W + x];
      } else if (mode == MDS_AUG_WARP) {
        // torch grid_sample(bilinear, zeros, align_corners=True): pixel coordinates, each corner weighted if inside
        const float sx = m[0] * (float)x + m[1] * (float)y + m[2], sy = m[3] * (float)x + m[4] * (float)y + m[5];
        const float fx = floorf(sx), fy = floorf(sy);
        const float wx1 = sx - fx, wy1 = sy - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
        // (far outside the frame every corner is masked; the select only keeps the float -> int conversions defined)
        const bool near = sx > -2.0f && sx < (float)W + 1.0f && sy > -2.0f && sy < (float)H + 1.0f;
        const int jx = near ? (int)fx : -2, jy = near ? (int)fy : -2;
        r = tap(jx, jy) * (wx0 * wy0) + tap(jx + 1, jy) * (wx1 * wy0) + tap(jx, jy + 1) * (wx0 * wy1) + tap(jx + 1, jy + 1) * (wx1 * wy1);
      } else if (mode == MDS_AUG_SHARP) {
        const float c = tap(x, y);
        if (x >= 1 && x < W - 1 && y >= 1 && y < H - 1) {
          float s_ = tap(x - 1, y - 1) + tap(x, y - 1) + tap(x + 1, y - 1) + tap(x - 1, y) + 5.0f * c + tap(x + 1, y) +
                     tap(x - 1, y + 1) + tap(x, y + 1) + tap(x + 1, y + 1);
          s_ = fminf(fmaxf(s_ * (1.0f / 13.0f), 0.0f), 1.0f);
          r = s_ + (c - s_) * jb.sharp_factor;         // factor in (0, 1): no clamp (kornia _blend_one)
        } else {
          r = c;                                       // border pixels keep their value
        }
      } else if (staged) {   // MDS_AUG_TAPS from the staged box: every tap of every pixel of the tile lies inside it (checked above)
        const float* c0 = tile + (y - by0) * bw + (x - bx0);
        for (int k = 0; k < ntaps; ++k) r += tapw[k] * c0[tapo[k]];
      } else {
        for (int k = 0; k < jb.ntaps; ++k) r += jb.tap_w[k] * px(src, H, W, x + jb.tap_dx[k], y + jb.tap_dy[k]);
      }
    }
    v[j] = r;
  }
  if (jb.point) {
    if (jb.bright_on) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fminf(fmaxf(v[j] + jb.bright_add, 0.0f), 1.0f);
    }
    if (jb.contrast_on) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fminf(fmaxf(v[j] * jb.contrast_mul, 0.0f), 1.0f);
    }
    if (jb.posterize_bits > 0) {
      const int sh = 8 - jb.posterize_bits;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned u = (unsigned)(unsigned char)(int)(v[j] * 255.0f);     // torch .to(uint8): truncation
        v[j] = (float)((u >> sh) << sh) / 255.0f;
      }
    }
    if (jb.noise_on) {
      float z[4];
      if (a.noise) {
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] = (x0 + j < W) ? a.noise[plane_off + (long)y * W + x0 + j] : 0.0f;
      } else {     // counter = this quad's index in the (B, T, H, ceil(W / 4)) order: a function of (seed, element) only
        normal4((uint32_t)jb.noise_seed, (unsigned long long)((((long)b * a.T + t) * H + y) * (long)((W + 3) >> 2) + (x0 >> 2)), z);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = v[j] + z[j] * jb.noise_std + jb.noise_mean;
    }
  }
  float* o = dst + (long)y * W + x0;
  if (x0 + 3 < W && ((W & 3) == 0)) {
    *(f32x4*)o = (f32x4){v[0], v[1], v[2], v[3]};
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (x0 + j < W) o[j] = v[j];
  }
}

extern "C" int mds_aug_pass(const mds_aug_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->B > 0 && a->T > 0 && a->H > 0 && a->W > 0 && a->jobs && a->buf[0] && a->buf[1], "aug_pass: bad args");
  MDS_REQUIRE((long)a->H * a->W < 2147483647L, "aug_pass: frame too large");
  const int bx = cdiv(a->H, AUG_TR) * cdiv(a->W, AUG_TC);      // one block per 8 x 128 output tile
  MDS_REQUIRE(a->T <= 65535 && a->B <= 65535, "aug_pass: grid");
  MDS_LAUNCH(aug_kernel, dim3(bx, a->T, a->B), dim3(256), 0, stream, *a);
  return mds_check_launch("aug_pass");
}

// ------------------------------------------------------------------ SURVEY 8(f) N4, device side: pitched luma plane -> frames
// HBM streaming copy: 16 bytes per lane where source row and destination row are both 16-byte aligned, bytes otherwise.
__global__ __launch_bounds__(256) void frame_luma_kernel(mds_frame_luma_args a) {
  const int f = blockIdx.z, y = blockIdx.y;
  const unsigned char* s = a.src + (long)f * a.surface_stride + (long)y * a.pitch;
  unsigned char* d = a.dst + ((long)f * a.height + y) * a.width;
  const bool vec = ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0;
  const int nv = vec ? a.width >> 4 : 0;
  for (int v = blockIdx.x * 256 + threadIdx.x; v < nv; v += gridDim.x * 256) ((u32x4*)d)[v] = ((const u32x4*)s)[v];
  for (int x = 16 * nv + blockIdx.x * 256 + threadIdx.x; x < a.width; x += gridDim.x * 256) d[x] = s[x];
}
extern "C" int mds_frame_luma(const mds_frame_luma_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->src && a->dst && a->width > 0 && a->height > 0 && a->count > 0, "frame_luma: bad args");
  MDS_REQUIRE(a->pitch >= a->width && a->height <= 65535 && a->count <= 65535, "frame_luma: pitch must cover a row; at most 65535 rows / surfaces");
  MDS_LAUNCH(frame_luma_kernel, dim3(cdiv(a->width, 16 * 256) > 0 ? cdiv(a->width, 16 * 256) : 1, a->height, a->count), dim3(256), 0, stream, *a);
  return mds_check_launch("frame_luma");
}

// ------------------------------------------------------------------ predictor glue, slot-addressed row copies
// One launch moves the rows a predictor step needs between its device rings and a plan's buffers; the slot numbers are kernel
// arguments (an index tensor would be a host-to-device copy per frame).  grid = (pieces of a row, rows).
__global__ __launch_bounds__(256) void copy_rows_kernel(mds_copy_rows_args a) {
  const int r = blockIdx.y;
  const unsigned char* s = (const unsigned char*)a.src + (long)a.src_slot[r] * a.src_pitch;
  unsigned char* d = (unsigned char*)a.dst + (long)a.dst_slot[r] * a.dst_pitch;
  const bool vec = ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0;
  const long nv = vec ? a.row_bytes >> 4 : 0;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nv; v += (long)gridDim.x * 256) ((u32x4*)d)[v] = ((const u32x4*)s)[v];
  for (long x = 16 * nv + (long)blockIdx.x * 256 + threadIdx.x; x < a.row_bytes; x += (long)gridDim.x * 256) d[x] = s[x];
}
extern "C" int mds_copy_rows(const mds_copy_rows_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->src && a->dst && a->row_bytes > 0 && a->nrows > 0 && a->nrows <= MDS_COPY_ROWS_MAX, "copy_rows: bad args (at most %d rows)", MDS_COPY_ROWS_MAX);
  for (int r = 0; r < a->nrows; ++r) MDS_REQUIRE(a->dst_slot[r] >= 0 && a->src_slot[r] >= 0, "copy_rows: negative slot");
  long bx = cdiv(a->row_bytes, 16L * 256 * 4);      // four 16-byte vectors per lane
  bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
  MDS_LAUNCH(copy_rows_kernel, dim3((unsigned)bx, a->nrows), dim3(256), 0, stream, *a);
  return mds_check_launch("copy_rows");
}
