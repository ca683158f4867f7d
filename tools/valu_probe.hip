// valu_probe.hip -- issue cost of the VALU instructions the blend kernels are made of, on the box it runs on.
// Every wave runs a long loop of 8 INDEPENDENT chains of one instruction (so the 4-cycle dependent latency is
// hidden and what is timed is issue), 8 waves per SIMD.  Reported: cycles per wave64 instruction per SIMD,
// assuming the clock given on the command line (default 2400 MHz).
//   build: hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o tools/_build/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(X) X X X X X X X X
template <int KIND>
__global__ __launch_bounds__(256) void k_probe(float* out, int iters, float seed, unsigned long long* ticks) {
  const unsigned long long t_begin = __builtin_readcyclecounter();
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * 0.5f, b3 = a3 * 0.5f, b4 = a4 * .5f, b5 = a5 * .5f, b6 = a6 * .5f, b7 = a7 * .5f;
  const float m = 0.999f, c = 0.001f;
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3}, p4 = {a4, b4}, p5 = {a5, b5}, p6 = {a6, b6}, p7 = {a7, b7};
  const v2 pm = {m, m}, pc = {c, c};
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) {  // v_fma_f32
#define S(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(m), "v"(c));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 1) {  // v_pk_fma_f32 (two fp32 FMAs per lane)
#define S(r) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(pm), "v"(pc));
      REP8(S(p0) S(p1) S(p2) S(p3) S(p4) S(p5) S(p6) S(p7))
#undef S
    } else if (KIND == 2) {  // v_pk_mul_f32
#define S(r) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r) : "v"(pm));
      REP8(S(p0) S(p1) S(p2) S(p3) S(p4) S(p5) S(p6) S(p7))
#undef S
    } else if (KIND == 3) {  // v_exp_f32 (transcendental)
#define S(r) asm volatile("v_exp_f32 %0, %0" : "+v"(r));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 4) {  // v_cndmask_b32 (select on vcc)
#define S(r) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(c));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 5) {  // v_cmp_lt_f32 writing an SGPR pair
#define S(r) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(r), "v"(c) : "s20", "s21");
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 6) {  // v_rcp_f32
#define S(r) asm volatile("v_rcp_f32 %0, %0" : "+v"(r));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 7) {  // v_mul_f32 with an SGPR operand
#define S(r) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r) : "s"(m));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 8) {  // v_pk_add_f32
#define S(r) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r) : "v"(pc));
      REP8(S(p0) S(p1) S(p2) S(p3) S(p4) S(p5) S(p6) S(p7))
#undef S
    } else if (KIND == 9) {  // v_ldexp_f32
#define S(r) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(r) : "v"(0));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 10) {  // v_cndmask_b32 with an SGPR-pair mask (VOP3)
#define S(r) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(r) : "v"(c));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 11) {  // half v_fma_f32, half v_cndmask_b32 (vcc), interleaved
#define S(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(m), "v"(c));
#define Q(r) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(c));
      REP8(S(a0) Q(a1) S(a2) Q(a3) S(a4) Q(a5) S(a6) Q(a7))
#undef S
#undef Q
    } else if (KIND == 12) {  // v_mov_b32 (what an exec-masked update costs)
#define S(r) asm volatile("v_mov_b32 %0, %1" : "+v"(r) : "v"(c));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 13) {  // v_cndmask_b32 whose two sources differ from the destination
#define S(r, q) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r) : "v"(q), "v"(c));
      REP8(S(a0, b0) S(a1, b1) S(a2, b2) S(a3, b3) S(a4, b4) S(a5, b5) S(a6, b6) S(a7, b7))
#undef S
    } else if (KIND == 14) {  // v_mul_f32, all VGPR operands
#define S(r) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r) : "v"(m));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 15) {  // v_cmp_lt_f32 writing vcc (VOPC)
#define S(r) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r), "v"(c) : "vcc");
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 16) {  // v_min_f32 / v_max_f32 pair (select-free clamps)
#define S(r) asm volatile("v_min_f32 %0, %0, %1" : "+v"(r) : "v"(m));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 17) {  // v_fmac_f32 (VOP2 two-operand FMA)
#define S(r) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r) : "v"(m), "v"(c));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 19) {  // v_cmp_lt_u64 -> vcc (the rank loops of the tile sort compare 64-bit keys)
#define S(r) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(r), "v"(pm) : "vcc");
      REP8(S(p0) S(p1) S(p2) S(p3) S(p4) S(p5) S(p6) S(p7))
#undef S
    } else if (KIND == 20) {  // v_cmp_lt_u32 -> vcc
#define S(r) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(r), "v"(c) : "vcc");
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 21) {  // v_addc_co_u32 (count += carry)
#define S(r) asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(r) : : "vcc");
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 22) {  // v_cmp_lt_u64 -> vcc followed by v_addc_co_u32: one rank step
#define S(r, q) asm volatile("v_cmp_lt_u64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(r) : "v"(q), "v"(pm) : "vcc");
      REP8(S(a0, p0) S(a1, p1) S(a2, p2) S(a3, p3) S(a4, p4) S(a5, p5) S(a6, p6) S(a7, p7))
#undef S
    } else if (KIND == 23) {  // v_add_f32 with a DPP quad permute on src0 (K7's reduce-scatter, xor-1 / xor-2 levels)
#define S(r) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 24) {  // v_add_f32 with a DPP row shift and a bank mask (xor-4 / xor-8 levels)
#define S(r) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xa" : "+v"(r));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 25) {  // v_mov_b32 with DPP
#define S(r, q) asm volatile("v_mov_b32_dpp %0, %1 row_shl:4 row_mask:0xf bank_mask:0x5" : "+v"(r) : "v"(q));
      REP8(S(a0, b0) S(a1, b1) S(a2, b2) S(a3, b3) S(a4, b4) S(a5, b5) S(a6, b6) S(a7, b7))
#undef S
    } else if (KIND == 26) {  // v_fmaak_f32: a 32-bit LITERAL addend (the exp polynomial's coefficients)
#define S(r) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3d2aaa75" : "+v"(r) : "v"(m));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 27) {  // v_fmamk_f32: literal multiplier
#define S(r) asm volatile("v_fmamk_f32 %0, %0, 0x3f7fbe77, %1" : "+v"(r) : "v"(c));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 28) {  // v_fmac_f32 with a literal source
#define S(r) asm volatile("v_fmac_f32 %0, 0x3a83126f, %1" : "+v"(r) : "v"(m));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 29) {  // v_min_f32 with a literal
#define S(r) asm volatile("v_min_f32 %0, 0x3f7d70a4, %0" : "+v"(r));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 30) {  // v_lshl_add_u32 (VOP3, integer)
#define S(r) asm volatile("v_lshl_add_u32 %0, %1, 23, %0" : "+v"(r) : "v"(c));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 31) {  // v_mul_u32_u24 with an inline constant
#define S(r) asm volatile("v_mul_u32_u24 %0, 48, %0" : "+v"(r));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 32) {  // v_sub_f32 (VOP2, two VGPRs)
#define S(r) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r) : "v"(c));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 33) {  // v_cndmask_b32_e64 with the abs/neg modifiers K6's Tw select uses
#define S(r) asm volatile("v_cndmask_b32_e64 %0, -|%0|, %1, s[20:21]" : "+v"(r) : "v"(c));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 34) {  // v_add_u32
#define S(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(c));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 35) {  // v_lshlrev_b32
#define S(r) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 36) {  // v_and_b32
#define S(r) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r) : "v"(c));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 37) {  // v_cvt_f32_ubyte0
#define S(r) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(r));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 38) {  // v_med3_f32
#define S(r) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(m));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 39) {  // v_max_f32
#define S(r) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r) : "v"(c));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 40) {  // v_cndmask_b32_e32 on vcc, destination distinct from the sources' producer chain
#define S(r, q) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r) : "v"(q), "v"(c));
      REP8(S(a0, b0) S(a1, b1) S(a2, b2) S(a3, b3) S(a4, b4) S(a5, b5) S(a6, b6) S(a7, b7))
#undef S
    } else if (KIND == 41) {  // v_mul_f32 with the clamp output modifier (VOP3)
#define S(r) asm volatile("v_mul_f32_e64 %0, %0, %1 clamp" : "+v"(r) : "v"(m));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 42) {  // v_fma_f32 with an SGPR operand
#define S(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "s"(m), "v"(c));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 43) {  // pair: compare into VCC, select on VCC (VOP2 form)
#define S(r, q) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(r) : "v"(q), "v"(c) : "vcc");
      REP8(S(a0, b0) S(a1, b1) S(a2, b2) S(a3, b3) S(a4, b4) S(a5, b5) S(a6, b6) S(a7, b7))
#undef S
    } else if (KIND == 44) {  // pair: compare into an SGPR pair, select on it (VOP3 form)
#define S(r, q) asm volatile("v_cmp_lt_f32 s[20:21], %1, %2\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]" : "+v"(r) : "v"(q), "v"(c) : "s20", "s21");
      REP8(S(a0, b0) S(a1, b1) S(a2, b2) S(a3, b3) S(a4, b4) S(a5, b5) S(a6, b6) S(a7, b7))
#undef S
    } else if (KIND == 45) {  // pair: SALU writes VCC, select on VCC (what the compiler emits behind `a && b`)
#define S(r) asm volatile("s_and_b64 vcc, s[20:21], s[22:23]\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r) : "v"(c) : "vcc");
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 46) {  // pair: SALU writes an SGPR pair, select on it
#define S(r) asm volatile("s_and_b64 s[24:25], s[20:21], s[22:23]\n\tv_cndmask_b32_e64 %0, %0, %1, s[24:25]" : "+v"(r) : "v"(c) : "s24", "s25");
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    } else if (KIND == 18) {  // v_fma_f32 with two literal-free inline constants (VOP3, 3 VGPR reads vs 2)
#define S(r) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(r) : "v"(m));
      REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
    }
  }
  const unsigned long long t_end = __builtin_readcyclecounter();
  if (ticks != nullptr && (threadIdx.x & 63) == 0) ticks[(blockIdx.x * 256 + threadIdx.x) >> 6] = t_end - t_begin;
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int KIND>
double run(const char* name, float* out, int blocks, int iters, double mhz) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  static unsigned long long* ticks = nullptr;
  if (!ticks) hipMalloc(&ticks, sizeof(unsigned long long) * blocks * 4);
  k_probe<KIND><<<blocks, 256>>>(out, 64, 1.0f, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_probe<KIND><<<blocks, 256>>>(out, iters, 1.0f, ticks);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // waves per SIMD = blocks * 4 waves / (256 CUs * 4 SIMDs); instructions per wave = iters * 64
  const double waves_per_simd = blocks * 4.0 / 1024.0;
  const double cyc = ms * 1e-3 * mhz * 1e6 / (waves_per_simd * iters * 64.0);
  // the same from the waves' own s_memtime deltas (shader-clock ticks, no clock assumption): a wave shares its SIMD
  // with waves_per_simd - 1 others, so ticks per instruction per SIMD = mean delta / (instructions * waves_per_simd)
  static unsigned long long host_ticks[1 << 16];
  hipMemcpy(host_ticks, ticks, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < blocks * 4; i++) mean += (double)host_ticks[i];
  mean /= blocks * 4;
  const double tick_cyc = mean / (iters * 64.0 * waves_per_simd);
  printf("{\"instr\": \"%s\", \"ms\": %.4f, \"cycles_per_wave_instr_per_simd_at_assumed_clock\": %.3f, "
         "\"memtime_ticks_per_wave_instr_per_simd\": %.3f, \"implied_clock_MHz_if_tick_is_a_cycle\": %.0f}\n",
         name, ms, cyc, tick_cyc, mean / (ms * 1e-3) / 1e6);
  return cyc;
}

int main(int argc, char** argv) {
  const double mhz = argc > 1 ? atof(argv[1]) : 2400.0;
  const int blocks = 256 * 8, iters = 4096;  // 8 blocks per CU = 8 waves per SIMD
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  printf("{\"assumed_clock_MHz\": %.0f, \"blocks\": %d, \"waves_per_simd\": 8, \"independent_chains\": 8}\n", mhz, blocks);
  run<0>("v_fma_f32", out, blocks, iters, mhz);
  run<1>("v_pk_fma_f32", out, blocks, iters, mhz);
  run<2>("v_pk_mul_f32", out, blocks, iters, mhz);
  run<8>("v_pk_add_f32", out, blocks, iters, mhz);
  run<3>("v_exp_f32", out, blocks, iters, mhz);
  run<6>("v_rcp_f32", out, blocks, iters, mhz);
  run<4>("v_cndmask_b32", out, blocks, iters, mhz);
  run<5>("v_cmp_lt_f32 -> sgpr", out, blocks, iters, mhz);
  run<7>("v_mul_f32 (sgpr operand)", out, blocks, iters, mhz);
  run<9>("v_ldexp_f32", out, blocks, iters, mhz);
  run<14>("v_mul_f32 (vgpr operands)", out, blocks, iters, mhz);
  run<17>("v_fmac_f32", out, blocks, iters, mhz);
  run<18>("v_fma_f32 (inline constant addend)", out, blocks, iters, mhz);
  run<16>("v_min_f32", out, blocks, iters, mhz);
  run<12>("v_mov_b32", out, blocks, iters, mhz);
  run<10>("v_cndmask_b32_e64 (sgpr-pair mask)", out, blocks, iters, mhz);
  run<13>("v_cndmask_b32 (dst != src)", out, blocks, iters, mhz);
  run<11>("v_fma_f32 / v_cndmask_b32 interleaved", out, blocks, iters, mhz);
  run<15>("v_cmp_lt_f32 -> vcc", out, blocks, iters, mhz);
  run<23>("v_add_f32_dpp quad_perm", out, blocks, iters, mhz);
  run<24>("v_add_f32_dpp row_shr:4 bank_mask", out, blocks, iters, mhz);
  run<25>("v_mov_b32_dpp row_shl:4 bank_mask", out, blocks, iters, mhz);
  run<26>("v_fmaak_f32 (literal addend)", out, blocks, iters, mhz);
  run<27>("v_fmamk_f32 (literal multiplier)", out, blocks, iters, mhz);
  run<28>("v_fmac_f32 (literal source)", out, blocks, iters, mhz);
  run<29>("v_min_f32 (literal)", out, blocks, iters, mhz);
  run<30>("v_lshl_add_u32", out, blocks, iters, mhz);
  run<31>("v_mul_u32_u24 (inline constant)", out, blocks, iters, mhz);
  run<32>("v_sub_f32", out, blocks, iters, mhz);
  run<33>("v_cndmask_b32_e64 (sgpr-pair mask, -|x| modifier)", out, blocks, iters, mhz);
  run<34>("v_add_u32", out, blocks, iters, mhz);
  run<35>("v_lshlrev_b32", out, blocks, iters, mhz);
  run<36>("v_and_b32", out, blocks, iters, mhz);
  run<37>("v_cvt_f32_ubyte0", out, blocks, iters, mhz);
  run<38>("v_med3_f32", out, blocks, iters, mhz);
  run<39>("v_max_f32", out, blocks, iters, mhz);
  run<40>("v_cndmask_b32_e32 (vcc, dst != src)", out, blocks, iters, mhz);
  run<41>("v_mul_f32_e64 clamp", out, blocks, iters, mhz);
  run<42>("v_fma_f32 (sgpr operand)", out, blocks, iters, mhz);
  run<43>("PAIR v_cmp -> vcc ; v_cndmask_e32 vcc", out, blocks, iters, mhz);
  run<44>("PAIR v_cmp -> s[20:21] ; v_cndmask_e64 s[20:21]", out, blocks, iters, mhz);
  // (kinds 45 / 46 -- SALU-written masks -- did not finish within the run's timeout on the box they were tried on:
  //  they read SGPRs the compiler may own; left out of the default list until they are rewritten with operands)
  run<19>("v_cmp_lt_u64 -> vcc", out, blocks, iters, mhz);
  run<20>("v_cmp_lt_u32 -> vcc", out, blocks, iters, mhz);
  run<21>("v_addc_co_u32", out, blocks, iters, mhz);
  run<22>("v_cmp_lt_u64 + v_addc_co_u32 (two instructions)", out, blocks, iters, mhz);
  hipFree(out);
  return 0;
}
