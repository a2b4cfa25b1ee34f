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

// zeros outside - branch-free (clamped address, masked value): a conditional load would serialise the four corners of a
// bilinear sample and the four pixels of a lane behind each other's memory latency
MDS_DEV float px(const float* plane, int H, int W, int x, int y) {
  const bool in = (unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H;
  const int xc = x < 0 ? 0 : (x >= W ? W - 1 : x), yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
  const float v = plane[(long)yc * W + xc];
  return in ? v : 0.0f;
}

__global__ __launch_bounds__(256) void aug_kernel(mds_aug_args a) {
  const int b = blockIdx.z, t = blockIdx.y;
  const mds_aug_job& jb = a.jobs[b];
  if (!jb.active) return;
  const int H = a.H, W = a.W, W4 = (W + 3) >> 2;
  const long plane_off = ((long)b * a.T + t) * H * W;
  const float* src = a.buf[jb.src] + plane_off;
  float* dst = a.buf[jb.dst] + plane_off;
  const int mode = jb.mode;
  float m[6];
  if (mode == MDS_AUG_WARP) {
#pragma unroll
    for (int j = 0; j < 6; ++j) m[j] = a.maps[((long)b * a.T + t) * 6 + j];
  }
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < (long)H * W4; q += (long)gridDim.x * 256) {
    const int y = (int)(q / W4), x0 = 4 * (int)(q - (long)y * W4);
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
          const int ix = (int)fx, iy = (int)fy;
          const float wx1 = sx - fx, wy1 = sy - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
          // (far outside the frame every corner is masked; the clamp below only keeps float -> int conversions defined)
          const bool near = sx > -2.0f && sx < (float)W + 1.0f && sy > -2.0f && sy < (float)H + 1.0f;
          const int jx = near ? ix : -2, jy = near ? iy : -2;
          r = px(src, H, W, jx, jy) * (wx0 * wy0) + px(src, H, W, jx + 1, jy) * (wx1 * wy0) +
              px(src, H, W, jx, jy + 1) * (wx0 * wy1) + px(src, H, W, jx + 1, jy + 1) * (wx1 * wy1);
        } else if (mode == MDS_AUG_SHARP) {
          const float c = src[(long)y * W + x];
          if (x >= 1 && x < W - 1 && y >= 1 && y < H - 1) {
            const float* p = src + (long)y * W + x;
            float s = p[-W - 1] + p[-W] + p[-W + 1] + p[-1] + 5.0f * c + p[1] + p[W - 1] + p[W] + p[W + 1];
            s = fminf(fmaxf(s * (1.0f / 13.0f), 0.0f), 1.0f);
            r = s + (c - s) * jb.sharp_factor;         // factor in (0, 1): no clamp (kornia _blend_one)
          } else {
            r = c;                                     // border pixels keep their value
          }
        } else {   // MDS_AUG_TAPS
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
        } else {
          normal4((uint32_t)jb.noise_seed, (unsigned long long)(((long)b * a.T + t) * (long)H * W4 + q), z);
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
}

extern "C" int mds_aug_pass(const mds_aug_args* a, mds_stream_t stream) {
  MDS_REQUIRE(a && a->B > 0 && a->T > 0 && a->H > 0 && a->W > 0 && a->jobs && a->buf[0] && a->buf[1], "aug_pass: bad args");
  MDS_REQUIRE((long)a->H * a->W < 2147483647L, "aug_pass: frame too large");
  const long quads = (long)a->H * ((a->W + 3) / 4);
  int bx = cdiv(quads, 256 * 4);          // four quads per thread
  if (bx < 1) bx = 1;
  MDS_REQUIRE(a->T <= 65535 && a->B <= 65535, "aug_pass: grid");
  MDS_LAUNCH(aug_kernel, dim3(bx, a->T, a->B), dim3(256), 0, stream, *a);
  return mds_check_launch("aug_pass");
}
