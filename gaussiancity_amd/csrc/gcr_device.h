// gcr_device.h -- device-side helpers shared by every gfx950 kernel of the rasterizer.
//
// Numerics contract "gcr-fp32-v1" (DESIGN.md section 4): IEEE binary32, source association
// order of the reference, no implicit contraction (the library is built with
// -ffp-contract=off), correctly rounded division / sqrt
// (-fhip-fp32-correctly-rounded-divide-sqrt), denormals kept, fused multiply-adds only where
// __builtin_fmaf is written, exp() = gcr_expf() below.  With this contract the whole forward
// pass is bit-reproducible against the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GCR_DEV __device__ __forceinline__
#define GCR_TILE_X 16  // cr/config.h:16
#define GCR_TILE_Y 16  // cr/config.h:17
#define GCR_TILE_PIXELS 256
#define GCR_WAVE 64

GCR_DEV float gcr_min(float a, float b) { return a < b ? a : b; }
GCR_DEV float gcr_max(float a, float b) { return a > b ? a : b; }

// float -> int, round toward zero, saturating, NaN -> 0 (cvt.rzi.s32.f32 semantics that the
// reference relies on in getRect / radii, cr/auxiliary.h:36-46, cr/forward.cu:228).
GCR_DEV int gcr_f2i_sat(float v) {
  if (!(v == v)) return 0;
  if (v >= 2147483648.0f) return 2147483647;
  if (v <= -2147483648.0f) return (-2147483647 - 1);
  return (int)v;
}

// gcr-fp32-v1 exponential (two-term Cody-Waite reduction, degree-6 polynomial, < 1 ulp).
// Replaces exp() at cr/forward.cu:317 and cr/backward.cu:525.  Every step is an exactly
// specified IEEE operation, so CPU and GPU agree bit for bit (v_exp_f32 would not).
GCR_DEV float gcr_expf(float x) {
  if (x < -87.0f) return 0.0f;
  if (x > 88.0f) return __builtin_inff();
  const float n = __builtin_rintf(x * 1.44269504088896341f);
  float r = __builtin_fmaf(n, -0.693145751953125f, x);
  r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);
  float p = 0.0013933652080595493f;
  p = __builtin_fmaf(p, r, 0.008363181725144386f);
  p = __builtin_fmaf(p, r, 0.04166646674275398f);
  p = __builtin_fmaf(p, r, 0.16666576266288757f);
  p = __builtin_fmaf(p, r, 0.5f);
  p = __builtin_fmaf(p, r, 1.0f);
  p = __builtin_fmaf(p, r, 1.0f);
  return p * __int_as_float(((int)n + 127) << 23);
}

// Same function without the range guards, for callers that guarantee -87 <= x <= 0 (the blend
// kernels clamp their skip bound to >= -87, which is exactly what the x < -87 guard does:
// exp -> 0 -> alpha = 0 < 1/255 -> skipped).  Bit-identical to gcr_expf on that domain.
GCR_DEV float gcr_expf_noguard(float x) {
  const float n = __builtin_rintf(x * 1.44269504088896341f);
  float r = __builtin_fmaf(n, -0.693145751953125f, x);
  r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);
  float p = 0.0013933652080595493f;
  p = __builtin_fmaf(p, r, 0.008363181725144386f);
  p = __builtin_fmaf(p, r, 0.04166646674275398f);
  p = __builtin_fmaf(p, r, 0.16666576266288757f);
  p = __builtin_fmaf(p, r, 0.5f);
  p = __builtin_fmaf(p, r, 1.0f);
  p = __builtin_fmaf(p, r, 1.0f);
  return p * __int_as_float(((int)n + 127) << 23);
}

// Non-parity variant (option "fast_exp"): hardware v_exp_f32, ~1 ulp, NOT bit-reproducible.
GCR_DEV float gcr_expf_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

// power = -0.5f*(a*dx*dx + c*dy*dy) - b*dx*dy   (cr/forward.cu:309-310, cr/backward.cu:520-521)
// with the explicit-FMA sites of gcr-fp32-v1.
GCR_DEV float gcr_power(float cx, float cy, float cz, float dx, float dy) {
  return __builtin_fmaf(-(cy * dx), dy, -0.5f * __builtin_fmaf(cz * dy, dy, (cx * dx) * dx));
}

// cr/auxiliary.h:32-34 -- double arithmetic on purpose.
GCR_DEV float gcr_ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// ---- wave64 primitives -------------------------------------------------------------------
// DPP butterfly: after the four in-row steps every lane of a 16-lane row holds its row sum;
// row_bcast:15 / row_bcast:31 then chain the rows so that lanes 48..63 hold the wave sum.
template <int CTRL, int ROW_MASK>
GCR_DEV float gcr_dpp_f(float v) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// Sum over the 64 lanes of a wave; the result is valid in lane 63 (and lanes 48..62).
GCR_DEV float gcr_wave_sum_to_lane63(float v) {
  v += gcr_dpp_f<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += gcr_dpp_f<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v += gcr_dpp_f<0x141, 0xf>(v);  // row_half_mirror
  v += gcr_dpp_f<0x140, 0xf>(v);  // row_mirror
  v += gcr_dpp_f<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
  v += gcr_dpp_f<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
  return v;
}

GCR_DEV uint32_t gcr_wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
GCR_DEV uint32_t gcr_wave_max_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}
// inclusive prefix sum over the 64 lanes
GCR_DEV uint32_t gcr_wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// Exclusive prefix sum over a 256-thread block; *total receives the block sum.
GCR_DEV uint32_t gcr_block_excl_scan_256(uint32_t v, uint32_t* lds4, uint32_t* total) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t incl = gcr_wave_incl_scan_u32(v, lane);
  if (lane == 63) lds4[w] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t s = lds4[i];
    if (i < w) base += s;
    tot += s;
  }
  *total = tot;
  __syncthreads();
  return base + incl - v;
}
