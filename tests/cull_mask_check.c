// Brute-force check of gcr_block_mask (gaussiancity_amd/csrc/gcr_cull.h) on the host: for random ellipses, every
// 4x4 block that contains a pixel the blend loop would evaluate (power in [pmin, 0], computed in fp32 exactly as
// gcr_power does) must have its bit set.  Prints: cases, violations, set bits, needed bits.  Second argument 2: the
// thirty-two 2x4 blocks of K6's sub-rows (gcr_block_mask_2x4) instead, and the 4x4 mask derived from them.
//   gcc -O2 -ffp-contract=off -I gaussiancity_amd/csrc tests/cull_mask_check.c -lm
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "gcr_cull.h"

static unsigned long long rng_s = 88172645463325252ull;
static double urand(void) {
  rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17;
  return (double)(rng_s >> 11) / 9007199254740992.0;
}
static float power_fp32(float cx, float cy, float cz, float dx, float dy) {
  return fmaf(-(cy * dx), dy, -0.5f * fmaf(cz * dy, dy, (cx * dx) * dx));
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 200000;
  const int bw = argc > 2 ? atoi(argv[2]) : 4;
  long violations = 0, set_bits = 0, needed_bits = 0, allmask = 0;
  for (long it = 0; it < n; it++) {
    const int mode = it % 4;
    // covariance from two axis sigmas and an angle; the rasterizer adds 0.3 to the diagonal (cr/forward.cu:102-103)
    double smax = mode == 1 ? 3000.0 : (mode == 2 ? 40.0 : 300.0);
    double s1 = exp(log(0.05) + urand() * (log(smax) - log(0.05)));
    double s2 = mode == 3 ? s1 * (0.9 + 0.2 * urand()) : exp(log(0.05) + urand() * (log(smax) - log(0.05)));
    double th = urand() * 6.283185307179586;
    double c = cos(th), s = sin(th);
    double a = c * c * s1 * s1 + s * s * s2 * s2 + 0.3, b = c * s * (s1 * s1 - s2 * s2), d = s * s * s1 * s1 + c * c * s2 * s2 + 0.3;
    float fa = (float)a, fb = (float)b, fd = (float)d;
    float det = fa * fd - fb * fb;
    if (!(det > 0.0f)) continue;
    float det_inv = 1.f / det;
    float cx = fd * det_inv, cy = -fb * det_inv, cz = fa * det_inv;   // conic (cr/forward.cu:199-201)
    float opacity = (float)(mode == 2 ? 1.0 : exp(log(0.004) + urand() * (log(1.0) - log(0.004))));
    float pmin = fmaxf(-87.0f, -logf(255.0f * opacity) - 1.0e-3f);
    float tile_x0 = (float)(16 * (int)(urand() * 120)), tile_y0 = (float)(16 * (int)(urand() * 68));
    double reach = 3.4 * sqrt(a > d ? a : d) + 20.0;
    float gx = tile_x0 + 8.0f + (float)((urand() * 2 - 1) * reach), gy = tile_y0 + 8.0f + (float)((urand() * 2 - 1) * reach);
    if (it % 17 == 0) { gx = floorf(gx); gy = floorf(gy) + 0.5f; }
    uint32_t m = bw == 4 ? gcr_block_mask(gx, gy, cx, cy, cz, pmin, tile_x0, tile_y0)
                         : gcr_block_mask_2x4(gx, gy, cx, cy, cz, pmin, tile_x0, tile_y0);
    if (m == (bw == 4 ? 0xFFFFu : 0xFFFFFFFFu)) allmask++;
    uint32_t need16 = 0;
    uint32_t need = 0;
    for (int y = 0; y < 16; y++)
      for (int x = 0; x < 16; x++) {
        float dx = gx - (tile_x0 + x), dy = gy - (tile_y0 + y);
        float pw = power_fp32(cx, cy, cz, dx, dy);
        if (!(pw > 0.0f) && !(pw < pmin)) {
          need |= bw == 4 ? 1u << ((y >> 2) * 4 + (x >> 2)) : 1u << ((y >> 2) * 8 + (x >> 1));
          need16 |= 1u << ((y >> 2) * 4 + (x >> 2));
        }
      }
    if (bw != 4) {
      // the derived 4x4 mask: conservative, and never more than the direct one
      const uint32_t d = gcr_block_mask_4x4_of_2x4(m);
      if ((need16 & ~d) || (d & ~gcr_block_mask(gx, gy, cx, cy, cz, pmin, tile_x0, tile_y0))) {
        violations++;
        if (violations <= 5) fprintf(stderr, "VIOLATION (derived 4x4) mask2x4=%08x derived=%04x need=%04x\n", m, d, need16);
      }
    }
    if (need & ~m) {
      violations++;
      if (violations <= 5)
        fprintf(stderr, "VIOLATION g=(%g,%g) conic=(%g,%g,%g) pmin=%g tile=(%g,%g) mask=%08x need=%08x\n", gx, gy, cx, cy, cz,
                pmin, tile_x0, tile_y0, m, need);
    }
    set_bits += __builtin_popcount(m);
    needed_bits += __builtin_popcount(need);
  }
  printf("{\"cases\": %ld, \"violations\": %ld, \"set_bits\": %ld, \"needed_bits\": %ld, \"all_blocks_masks\": %ld}\n", n, violations,
         set_bits, needed_bits, allmask);
  return violations != 0;
}
