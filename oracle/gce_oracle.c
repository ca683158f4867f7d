/*
 * gce_oracle.c -- CPU restatement of GaussianCity's multi-resolution hash-grid encoder
 * (reference extensions/grid_encoder/grid_encoder_ext.cu, "ge/").
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file; the product (gaussiancity_amd/) never imports, links or calls it.
 *
 * PARITY STATUS: "parity unpinned" at the kernel level -- the reference is CUDA only (no CPU path, no
 * tests, no fixtures).  Its Python half (extensions/grid_encoder/__init__.py) IS importable and pins
 * the level offsets, per-level scale, argument order and output permutation (tests/golden/
 * grid_encoder_golden.json).  The kernels are restated statement by statement and checked against an
 * independent float64 PyTorch-autograd formulation (tests/test_grid_encoder_oracle.py).
 *
 * NUMERICS "gce-fp32-v1" (shared with the HIP kernels): float32 only (upstream also dispatches half /
 * double; GaussianCity's embeddings are float32 parameters); IEEE binary32 in the reference's
 * association order, no contraction; the per-level scale exp2f(level*S)*H - 1 (ge/:132,277) is evaluated
 * ONCE ON THE HOST with libm for both implementations and handed to the kernels -- the GPU's v_exp_f32
 * and libm's exp2f are not reproducible on each other's side.  Forward and dy_dx are then bit-exact;
 * grad_embeddings is a sum of atomics on the GPU (order-dependent) and is tolerance-checked; here it
 * is accumulated in (level, point, corner) order.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ORG_MAX_D 7

/* ge/:132 (kernel_grid), :277 (kernel_grid_backward): scale of level l */
void org_level_scales(int L, float S, uint32_t H, float *out) {
  for (int l = 0; l < L; l++) out[l] = exp2f((float)l * S) * (float)H - 1.0f;
}

/* ge/:52-69 */
static uint32_t fast_hash(int D, const uint32_t *pos_grid) {
  static const uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
  uint32_t r = 0;
  for (int i = 0; i < D; i++) r ^= pos_grid[i] * primes[i];
  return r;
}

/* ge/:71-95 */
static uint32_t grid_index(int D, uint32_t C, uint32_t gridtype, int align_corners, uint32_t ch, uint32_t hashmap_size,
                           uint32_t resolution, const uint32_t *pos_grid) {
  uint32_t stride = 1, index = 0;
  for (int d = 0; d < D && stride <= hashmap_size; d++) {
    index += pos_grid[d] * stride;
    stride *= align_corners ? resolution : (resolution + 1);
  }
  if (gridtype == 0 && stride > hashmap_size) index = fast_hash(D, pos_grid);
  return (index % hashmap_size) * C + ch;
}

static int locate(const float *in, int D, float scale, int align_corners, float *pos, uint32_t *pos_grid) {
  for (int d = 0; d < D; d++)
    if (in[d] < 0 || in[d] > 1) return 0; /* out of [0,1]: ge/:113-120, :281-285 */
  for (int d = 0; d < D; d++) { /* ge/:140-145 */
    pos[d] = in[d] * scale + (align_corners ? 0.0f : 0.5f);
    pos_grid[d] = (uint32_t)floorf(pos[d]);
    pos[d] -= (float)pos_grid[d];
  }
  return 1;
}

/* ge/:97-256 (kernel_grid).  outputs [L][B][C]; dy_dx [B][L][D][C] when calc_grad_inputs. */
void org_forward(const float *inputs, const float *grid, const int32_t *offsets, float *outputs, uint32_t B, int D,
                 uint32_t C, uint32_t L, const float *scales, int calc_grad_inputs, float *dy_dx, uint32_t gridtype,
                 int align_corners) {
#pragma omp parallel for collapse(2) schedule(static)
  for (uint32_t level = 0; level < L; level++) {
    for (uint32_t b = 0; b < B; b++) {
      const float *g = grid + (size_t)(uint32_t)offsets[level] * C;
      float *out = outputs + ((size_t)level * B + b) * C;
      float *dd = calc_grad_inputs ? dy_dx + ((size_t)b * L + level) * D * C : NULL;
      float pos[ORG_MAX_D];
      uint32_t pos_grid[ORG_MAX_D];
      const float scale = scales[level];
      if (!locate(inputs + (size_t)b * D, D, scale, align_corners, pos, pos_grid)) {
        for (uint32_t ch = 0; ch < C; ch++) out[ch] = 0;
        if (dd) memset(dd, 0, sizeof(float) * D * C);
        continue;
      }
      const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
      const uint32_t resolution = (uint32_t)ceil(scale) + 1;
      float results[8] = {0};
      for (uint32_t idx = 0; idx < (1u << D); idx++) { /* ge/:152-180 */
        float w = 1;
        uint32_t pl[ORG_MAX_D];
        for (int d = 0; d < D; d++) {
          if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pos_grid[d]; }
          else { w *= pos[d]; pl[d] = pos_grid[d] + 1; }
        }
        const uint32_t index = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pl);
        for (uint32_t ch = 0; ch < C; ch++) results[ch] += w * g[index + ch];
      }
      for (uint32_t ch = 0; ch < C; ch++) out[ch] = results[ch];
      if (dd) { /* ge/:192-254 */
        for (int gd = 0; gd < D; gd++) {
          float rg[8] = {0};
          for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
            float w = scale;
            uint32_t pl[ORG_MAX_D];
            for (int nd = 0; nd < D - 1; nd++) {
              const int d = (nd >= gd) ? (nd + 1) : nd;
              if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; pl[d] = pos_grid[d]; }
              else { w *= pos[d]; pl[d] = pos_grid[d] + 1; }
            }
            pl[gd] = pos_grid[gd];
            const uint32_t il = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pl);
            pl[gd] = pos_grid[gd] + 1;
            const uint32_t ir = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pl);
            for (uint32_t ch = 0; ch < C; ch++) rg[ch] += w * (g[ir + ch] - g[il + ch]);
          }
          for (uint32_t ch = 0; ch < C; ch++) dd[gd * C + ch] = rg[ch];
        }
      }
    }
  }
}

/* ge/:258-337 (kernel_grid_backward): grad [L][B][C] -> grad_grid (zeroed by the caller, as upstream's
 * torch.zeros_like).  Sequential (level, point, corner, channel) order. */
void org_backward_embeddings(const float *grad, const float *inputs, const int32_t *offsets, float *grad_grid, uint32_t B,
                             int D, uint32_t C, uint32_t L, const float *scales, uint32_t gridtype, int align_corners) {
#pragma omp parallel for schedule(static) /* levels write disjoint slices of grad_grid */
  for (uint32_t level = 0; level < L; level++) {
    float *gg = grad_grid + (size_t)(uint32_t)offsets[level] * C;
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const float scale = scales[level];
    const uint32_t resolution = (uint32_t)ceil(scale) + 1;
    for (uint32_t b = 0; b < B; b++) {
      float pos[ORG_MAX_D];
      uint32_t pos_grid[ORG_MAX_D];
      if (!locate(inputs + (size_t)b * D, D, scale, align_corners, pos, pos_grid)) continue;
      const float *gc = grad + ((size_t)level * B + b) * C;
      for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1;
        uint32_t pl[ORG_MAX_D];
        for (int d = 0; d < D; d++) {
          if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pos_grid[d]; }
          else { w *= pos[d]; pl[d] = pos_grid[d] + 1; }
        }
        const uint32_t index = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pl);
        for (uint32_t ch = 0; ch < C; ch++) gg[index + ch] += w * gc[ch];
      }
    }
  }
}

/* ge/:339-366 (kernel_input_backward): grad [L][B][C], dy_dx [B][L][D][C] -> grad_inputs [B][D] */
void org_backward_inputs(const float *grad, const float *dy_dx, float *grad_inputs, uint32_t B, int D, uint32_t C,
                         uint32_t L) {
#pragma omp parallel for schedule(static)
  for (uint32_t t = 0; t < B * (uint32_t)D; t++) {
    const uint32_t b = t / D, d = t - b * D;
    const float *dd = dy_dx + (size_t)b * L * D * C;
    float r = 0;
    for (uint32_t l = 0; l < L; l++)
      for (uint32_t ch = 0; ch < C; ch++) r += grad[((size_t)l * B + b) * C + ch] * dd[(l * D + d) * C + ch];
    grad_inputs[t] = r;
  }
}
