// gcr_cull.h -- which 4x4-pixel blocks of a 16x16 tile can a Gaussian contribute to?
//
// Used by the staging threads of K6 / K7 (gcr_blend.hip).  The blend loops evaluate
//     power = -0.5 (cx dx^2 + cz dy^2) - cy dx dy,   alpha = min(0.99, opacity * exp(power))
// per (pixel, Gaussian) and skip the pair when power > 0 or alpha < 1/255 (cr/forward.cu:309-319,
// cr/backward.cu:520-529).  With pmin = -ln(255 opacity) - 1e-3 (gcr_alpha_skip_bound) a pair
// with power < pmin is skipped, so a Gaussian matters for a block of pixels only if the ellipse
//     E = { q(u, v) <= tau },  q = 0.5 cx u^2 + cy u v + 0.5 cz v^2,  (u, v) = pixel - centre
// reaches one of the block's pixel centres, for any tau >= -pmin + (what fp32 rounding can move
// `power` by).  gcr_block_mask() returns a CONSERVATIVE 16-bit set of such blocks (bit by*4+bx <->
// pixel centres [x0+4bx, x0+4bx+3] x [y0+4by, y0+4by+3]): a clear bit proves that every pixel of
// the block skips the Gaussian, so dropping it from that block's list changes no output bit.
//
// Per block row (a strip v in [V0, V1]) the u-extent of E inside the strip is exact: E's right
// boundary u_hi(v) = (-cy v + sqrt(2 tau cx - det v^2)) / cx is concave with its maximum at
// v_r = -cy ex / cz (ex = E's half-width), so over the strip it peaks at clamp(v_r); likewise the
// left boundary at clamp(-v_r).  A block is kept iff its u-range meets [u_lo, u_hi].
// Everything it cannot bound (non-positive-definite conic, NaN) returns "all blocks".
//
// The header compiles on the host too (tests/test_cull_mask.py brute-forces the guarantee against
// the pixel-exact definition with gcc); on the device the square roots / reciprocals are the raw
// ~1 ulp instructions -- the margins below (0.1 % on tau, rounding slack, 0.01 px) absorb that.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GCR_CULL_FN __host__ __device__ inline __attribute__((always_inline))
#else
#include <math.h>
#define GCR_CULL_FN static inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define GCR_CULL_SQRT(x) __builtin_amdgcn_sqrtf(x)
#define GCR_CULL_RCP(x) __builtin_amdgcn_rcpf(x)
#else
#define GCR_CULL_SQRT(x) __builtin_sqrtf(x)
#define GCR_CULL_RCP(x) (1.0f / (x))
#endif

#define GCR_CULL_EPS 0.01f  // pixels

// BW = width of a block in pixels (4: sixteen 4x4 blocks, bit by*4+bx; 2: thirty-two 2x4 blocks, bit by*8+bx); a
// constant at every call site.
GCR_CULL_FN uint32_t gcr_block_mask_bw(float gx, float gy, float cx, float cy, float cz, float pmin,
                                       float tile_x0, float tile_y0, const int BW) {
// Nothing here is part of the numerics contract (the result only has to be conservative, and the margins cover
// a fused multiply-add's single rounding many times over): let the compiler contract, ~30 fewer VALU per entry.
#pragma clang fp contract(fast)
  const int NBX = 16 / BW;
  const uint32_t ALL = BW == 4 ? 0xFFFFu : 0xFFFFFFFFu;
  if (!(pmin < 0.0f)) return 0u;  // alpha < 1/255 everywhere (power <= 0 always)
  const float det = cx * cz - cy * cy;
  if (!(det > 0.0f) || !(cx > 0.0f) || !(cz > 0.0f)) return ALL;
  // centre relative to the tile origin; largest |dx|, |dy| any pixel of the tile sees
  const float rx = gx - tile_x0, ry = gy - tile_y0;
  const float ux = __builtin_fmaxf(__builtin_fabsf(rx), __builtin_fabsf(rx - 15.0f));
  const float uy = __builtin_fmaxf(__builtin_fabsf(ry), __builtin_fabsf(ry - 15.0f));
  // fp32 evaluation of `power` (7 roundings of terms bounded by cx dx^2 + cz dy^2) can sit above the
  // real value by < 1e-6 * (cx ux^2 + cz uy^2): widen tau by that, plus 0.1 %
  const float tau = -pmin * 1.001f + 1.0e-6f * (cx * ux * ux + cz * uy * uy) + 1.0e-6f;
  const float t2 = 2.0f * tau;
  const float idet = GCR_CULL_RCP(det);
  const float ex = GCR_CULL_SQRT(t2 * cz * idet) * 1.0001f;  // half-extents of E
  const float ey = GCR_CULL_SQRT(t2 * cx * idet) * 1.0001f;
  if (!(ex == ex) || !(ey == ey)) return ALL;
  const float icx = GCR_CULL_RCP(cx);
  const float vr = -cy * ex * GCR_CULL_RCP(cz);  // v at which u is extremal (right: vr, left: -vr)
  const float k0 = t2 * cx * 1.0002f;            // discriminant 2 tau cx - det v^2, a hair generous
  uint32_t mask = 0u;
#pragma unroll
  for (int by = 0; by < 4; by++) {
    // strip of pixel-centre rows [4by, 4by+3] relative to the centre, clipped to E's v-range
    const float a = __builtin_fmaxf((float)(4 * by) - ry - GCR_CULL_EPS, -ey);
    const float b = __builtin_fminf((float)(4 * by + 3) - ry + GCR_CULL_EPS, ey);
    // clamp(x, a, b) written so that a NaN never shrinks the interval
    const float vh = __builtin_fminf(__builtin_fmaxf(vr, a), b);
    const float vl = __builtin_fminf(__builtin_fmaxf(-vr, a), b);
    const float sh = GCR_CULL_SQRT(__builtin_fmaxf(k0 - det * vh * vh, 0.0f));
    const float sl = GCR_CULL_SQRT(__builtin_fmaxf(k0 - det * vl * vl, 0.0f));
    const float u_hi = (sh - cy * vh) * icx;
    const float u_lo = (-sl - cy * vl) * icx;
    // block bx covers centres [BW bx, BW bx + BW-1] (tile-relative): hit iff BW bx + BW-1 + eps >= lo and BW bx - eps <= hi
    const float lo = u_lo + rx - ((float)(BW - 1) + GCR_CULL_EPS) - 1.0e-4f * __builtin_fabsf(u_lo);
    const float hi = u_hi + rx + GCR_CULL_EPS + 1.0e-4f * __builtin_fabsf(u_hi);
    // fmax/fmin return the non-NaN operand: an unbounded row keeps all its blocks
    const float inv = 1.0f / (float)BW;
    const float flo = __builtin_fminf(__builtin_fmaxf(__builtin_ceilf(lo * inv), 0.0f), (float)NBX);
    const float fhi = __builtin_fmaxf(__builtin_fminf(__builtin_floorf(hi * inv), (float)(NBX - 1)), -1.0f);
    const int ilo = (int)flo, ihi = (int)fhi;  // 0..NBX, -1..NBX-1
    const uint32_t row = (a <= b && ilo <= ihi) ? ((2u << (ihi & (NBX - 1))) - (1u << (ilo & (NBX - 1)))) : 0u;
    mask |= row << (NBX * by);
  }
  return mask;
}
// the sixteen 4x4 blocks (K7's rows)
GCR_CULL_FN uint32_t gcr_block_mask(float gx, float gy, float cx, float cy, float cz, float pmin,
                                    float tile_x0, float tile_y0) {
  return gcr_block_mask_bw(gx, gy, cx, cy, cz, pmin, tile_x0, tile_y0, 4);
}
// the thirty-two blocks of 2 (wide) x 4 (high) pixels (K6's sub-rows): bit by*8 + bx2
GCR_CULL_FN uint32_t gcr_block_mask_2x4(float gx, float gy, float cx, float cy, float cz, float pmin,
                                        float tile_x0, float tile_y0) {
  return gcr_block_mask_bw(gx, gy, cx, cy, cz, pmin, tile_x0, tile_y0, 2);
}
// the 4x4 mask a 2x4 mask implies (a 4x4 block = two neighbouring 2x4 blocks): conservative like its argument
GCR_CULL_FN uint32_t gcr_block_mask_4x4_of_2x4(uint32_t m) {
  uint32_t t = (m | (m >> 1)) & 0x55555555u;  // bit 2k of every byte: column pair k
  t = (t | (t >> 1)) & 0x33333333u;
  t = (t | (t >> 2)) & 0x0F0F0F0Fu;           // one nibble per band, in the low half of its byte
  return (t & 0xFu) | ((t >> 4) & 0xF0u) | ((t >> 8) & 0xF00u) | ((t >> 12) & 0xF000u);
}
