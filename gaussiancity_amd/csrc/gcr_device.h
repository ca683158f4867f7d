// gcr_device.h -- device-side helpers shared by every gfx950 kernel of the rasterizer.
//
// Numerics contract "gcr-fp32-v2" (DESIGN.md section 4): IEEE binary32, source association
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

// gcr-fp32-v2 exponential: two-term Cody-Waite reduction, degree-6 polynomial, < 1 ulp on [-87, 0].
// Replaces exp() at cr/forward.cu:317 and cr/backward.cu:525.  Every step is an exactly specified
// IEEE operation, so CPU (oracle/gcr_oracle.c: gcr_expf) and GPU agree bit for bit (v_exp_f32 would
// not).  11 VALU instructions:
//   t = fma(x, log2(e), 1.5*2^23)   -> the integer n = round(x*log2 e) sits in t's low mantissa bits
//   n = t - 1.5*2^23                 (exact)
//   r = x - n*ln2                    (two FMAs, hi/lo split of ln 2)
//   p = poly6(r)                     (six FMAs)
//   result = bits(p) + (bits(t) << 23)   = p * 2^n: ONE integer shift-add (0x4B400000 << 23 == 0 mod 2^32,
//            so only n reaches the exponent field); exact while p * 2^n is a normal number.
#define GCR_EXP_MAGIC 12582912.0f  // 1.5 * 2^23
GCR_DEV float gcr_expf_core(float x) {
  const float t = __builtin_fmaf(x, 1.44269504088896341f, GCR_EXP_MAGIC);
  const float n = t - GCR_EXP_MAGIC;
  float r = __builtin_fmaf(n, -0.693145751953125f, x);
  r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);
  float p = 0.0013933652080595493f;
  p = __builtin_fmaf(p, r, 0.008363181725144386f);
  p = __builtin_fmaf(p, r, 0.04166646674275398f);
  p = __builtin_fmaf(p, r, 0.16666576266288757f);
  p = __builtin_fmaf(p, r, 0.5f);
  p = __builtin_fmaf(p, r, 1.0f);
  p = __builtin_fmaf(p, r, 1.0f);
  return __uint_as_float(__float_as_uint(p) + (__float_as_uint(t) << 23));
}
GCR_DEV float gcr_expf(float x) {
  if (x < -87.0f) return 0.0f;
  if (x > 88.0f) return __builtin_inff();
  return gcr_expf_core(x);
}

// Same function without the range guards, for callers that guarantee -87 <= x <= 0 (the blend
// kernels clamp their skip bound to >= -87, which is exactly what the x < -87 guard does:
// exp -> 0 -> alpha = 0 < 1/255 -> skipped).  Bit-identical to gcr_expf on that domain.
GCR_DEV float gcr_expf_noguard(float x) { return gcr_expf_core(x); }

// power = -0.5f*(a*dx*dx + c*dy*dy) - b*dx*dy   (cr/forward.cu:309-310, cr/backward.cu:520-521)
// with the explicit-FMA sites of gcr-fp32-v1.
GCR_DEV float gcr_power(float cx, float cy, float cz, float dx, float dy) {
  return __builtin_fmaf(-(cy * dx), dy, -0.5f * __builtin_fmaf(cz * dy, dy, (cx * dx) * dx));
}

// cr/auxiliary.h:32-34 -- double arithmetic on purpose.
GCR_DEV float gcr_ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// One 8-byte word to host-visible (pinned, coherent) memory, written through so that a polling host thread
// sees it while later kernels of the stream are still running.  RELAXED on purpose: the word is the whole
// message, and a system-scope RELEASE would first write back the XCD's L2 (everything the kernel just wrote).
GCR_DEV void gcr_store_to_host(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Workgroup `b` of `nb` (BS threads each) writes its share of zeros over n floats at p: a scalar head up to the
// first 16-byte boundary, dwordx4 stores, a scalar tail.  Non-temporal: nothing reads these lines soon.
template <int BS = 256>
GCR_DEV void gcr_fill_zero_segment(float* __restrict__ p, unsigned long long n, int b, int nb, int tid) {
  const unsigned long long mis = ((unsigned long long)(uintptr_t)p >> 2) & 3ull;
  unsigned long long head = (4ull - mis) & 3ull;
  if (head > n) head = n;
  const unsigned long long n4 = (n - head) >> 2;
  const unsigned long long tail0 = head + (n4 << 2);
  if (b == 0) {
    if ((unsigned long long)tid < head) p[tid] = 0.0f;
    if (tail0 + (unsigned long long)tid < n && tid < 3) p[tail0 + tid] = 0.0f;
  }
  typedef float gcr_f4 __attribute__((ext_vector_type(4)));
  gcr_f4* __restrict__ q = (gcr_f4*)(p + head);
  const gcr_f4 z = {0.0f, 0.0f, 0.0f, 0.0f};
  for (unsigned long long i = (unsigned long long)b * (unsigned long long)BS + (unsigned long long)tid; i < n4;
       i += (unsigned long long)nb * (unsigned long long)BS)
    __builtin_nontemporal_store(z, q + i);  // measured: plain stores make the C2 backward 5 % slower
}

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

// ---- reduce-scatter of nine per-lane values over each 16-lane DPP row ----------------------
// Lane-xor exchanges: xor 1 / 2 are quad permutes; xor 4 / 8 take two bank-masked row shifts
// (lanes with the bit clear read from lane+n via row_shl:n, lanes with it set from lane-n via
// row_shr:n; bank_mask picks which 4-lane banks each instruction writes).
template <int CTRL, int BANK_MASK>
GCR_DEV float gcr_dpp_keep(float old, float src) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, BANK_MASK, false));
}
GCR_DEV float gcr_lane_xor1(float v) { return gcr_dpp_keep<0xB1, 0xf>(0.0f, v); }  // quad_perm [1,0,3,2]
GCR_DEV float gcr_lane_xor2(float v) { return gcr_dpp_keep<0x4E, 0xf>(0.0f, v); }  // quad_perm [2,3,0,1]
GCR_DEV float gcr_lane_xor4(float v) {
  const float t = gcr_dpp_keep<0x114, 0xA>(0.0f, v);  // row_shr:4 -> banks 1,3
  return gcr_dpp_keep<0x104, 0x5>(t, v);              // row_shl:4 -> banks 0,2
}
GCR_DEV float gcr_lane_xor8(float v) {
  const float t = gcr_dpp_keep<0x118, 0xC>(0.0f, v);  // row_shr:8 -> banks 2,3
  return gcr_dpp_keep<0x108, 0x3>(t, v);              // row_shl:8 -> banks 0,1
}
// One butterfly level on a pair: the lane whose `bit` is clear ends up with a's partial sum,
// its partner with b's -- one register leaves the level instead of two.
#define GCR_RS_MERGE(a, b, bit, XORFN) (((bit) ? (b) : (a)) + XORFN((bit) ? (a) : (b)))
// In: v[0..8] per lane.  Out (return value): within every 16-lane row, lane r holds the row sum
// of v[r] for r = 0..7 and lanes 8..15 hold the row sum of v[8].  31 VALU ops (9 plain
// row all-reduces would be 36, a full wave all-reduce 54).
GCR_DEV float gcr_row_reduce_scatter9(const float (&v)[9], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
  const float m01 = GCR_RS_MERGE(v[0], v[1], b0, gcr_lane_xor1);
  const float m23 = GCR_RS_MERGE(v[2], v[3], b0, gcr_lane_xor1);
  const float m45 = GCR_RS_MERGE(v[4], v[5], b0, gcr_lane_xor1);
  const float m67 = GCR_RS_MERGE(v[6], v[7], b0, gcr_lane_xor1);
  float s8 = v[8] + gcr_lane_xor1(v[8]);
  const float m03 = GCR_RS_MERGE(m01, m23, b1, gcr_lane_xor2);
  const float m47 = GCR_RS_MERGE(m45, m67, b1, gcr_lane_xor2);
  s8 += gcr_lane_xor2(s8);
  const float m07 = GCR_RS_MERGE(m03, m47, b2, gcr_lane_xor4);
  s8 += gcr_dpp_keep<0x141, 0xf>(0.0f, s8);  // row_half_mirror: the other quad of the 8-lane half
  return GCR_RS_MERGE(m07, s8, b3, gcr_lane_xor8);
}

// The same reduce-scatter in 20 VALU instructions (round 4; the version above compiles to 35: 16 selects, 13 DPP
// operations, 4 moves / adds and two s_nop).  The lane bits are consumed in the order 3, 2, 1, 0 instead of 0, 1, 2, 3:
// on bits 3 and 2 a DPP row rotate / row shift with a BANK MASK adds and selects in one instruction (a bank is four
// lanes: bits 3 and 2 of the lane number), so only bit 1 needs selects.  Written as one asm block because the
// bank-masked form needs the destination as the "old" value of the lanes it leaves alone, and because a DPP
// instruction must not read a VGPR a VALU instruction wrote in the two slots before it (the hardware does not
// interlock: the independent chains are interleaved by hand, s_nop where nothing is left to interleave).
// Out: every lane of a 16-lane row holds in `r` the row sum of term t = bit3 + 2 bit2 + 4 bit1 of its lane number
// (both values of bit 0), and in the return value the row sum of v[8].
GCR_DEV float gcr_row_reduce_scatter9_b(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7,
                                        float v8, float& r) {
  float keep, send;
  const unsigned long long m_bit1 = 0xCCCCCCCCCCCCCCCCull;
  asm volatile(
      "s_nop 1\n\t"
      // bit 3: lanes 0-7 <- (v_even[i] + v_even[i^8]), lanes 8-15 <- (v_odd[i] + v_odd[i^8])
      "v_add_f32_dpp %[a0], %[a0], %[a0] row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %[a2], %[a2], %[a2] row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %[a4], %[a4], %[a4] row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %[a6], %[a6], %[a6] row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %[a0], %[b1], %[b1] row_ror:8 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
      "v_add_f32_dpp %[a2], %[b3], %[b3] row_ror:8 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
      "v_add_f32_dpp %[a4], %[b5], %[b5] row_ror:8 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
      "v_add_f32_dpp %[a6], %[b7], %[b7] row_ror:8 row_mask:0xf bank_mask:0xc bound_ctrl:1\n\t"
      "v_add_f32_dpp %[s8], %[s8], %[s8] row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      // bit 2: lanes with bit 2 clear (banks 0, 2) keep the first register's terms, the others take the second's
      "v_add_f32_dpp %[a0], %[a0], %[a0] row_shl:4 row_mask:0xf bank_mask:0x5 bound_ctrl:1\n\t"
      "v_add_f32_dpp %[a0], %[a2], %[a2] row_shr:4 row_mask:0xf bank_mask:0xa bound_ctrl:1\n\t"
      "v_add_f32_dpp %[a4], %[a4], %[a4] row_shl:4 row_mask:0xf bank_mask:0x5 bound_ctrl:1\n\t"
      "v_add_f32_dpp %[a4], %[a6], %[a6] row_shr:4 row_mask:0xf bank_mask:0xa bound_ctrl:1\n\t"
      "v_add_f32_dpp %[s8], %[s8], %[s8] row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      // bit 1 (inside a bank: selects)
      "v_cndmask_b32_e64 %[keep], %[a0], %[a4], %[m1]\n\t"
      "v_cndmask_b32_e64 %[send], %[a4], %[a0], %[m1]\n\t"
      "v_add_f32_dpp %[s8], %[s8], %[s8] row_ror:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %[keep], %[send], %[keep] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %[s8], %[s8], %[s8] row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 0\n\t"
      // bit 0: both lanes of a pair end up with the complete sum of their term
      "v_add_f32_dpp %[keep], %[keep], %[keep] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      : [a0] "+v"(v0), [a2] "+v"(v2), [a4] "+v"(v4), [a6] "+v"(v6), [s8] "+v"(v8), [keep] "=&v"(keep), [send] "=&v"(send)
      : [b1] "v"(v1), [b3] "v"(v3), [b5] "v"(v5), [b7] "v"(v7), [m1] "s"(m_bit1));
  r = keep;
  return v8;
}
// which of the nine terms lane (lane & 15) of a row holds after gcr_row_reduce_scatter9_b, or -1: the even lanes hold
// term bit3 + 2 bit2 + 4 bit1, lane 1 speaks for term 8
GCR_DEV int gcr_row_reduce_scatter9_b_slot(int lane) {
  const int l = lane & 15;
  if (l == 1) return 8;
  if (l & 1) return -1;
  return ((l >> 3) & 1) + 2 * ((l >> 2) & 1) + 4 * ((l >> 1) & 1);
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
