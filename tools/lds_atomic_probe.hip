// lds_atomic_probe.hip -- what K7's per-step accumulator update costs the CU's LDS pipeline (round 5: the knock-out of
// that one ds_add_f32 shortens the backward blend by a quarter, profiles/r05_k7_knockouts_item.jsonl).  8 waves per SIMD,
// every wave a long loop of LDS operations whose addresses change from trip to trip like a walk's do; reported: cycles per
// wave64 instruction per CU at the assumed clock.
//   build: hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_probe.hip -o tools/_build/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// OP 0: ds_add_f32, 1: ds_write_b32, 2: ds_add_u32, 3: ds_add_rtn_f32, 4: ds_add_f64, 5: ds_add_u64, 6: ds_max_f32, 7: ds_pk_add_f16, 8: ds_write_b64
// PATTERN 0: 64 lanes, consecutive floats (conflict-free)
//         1: K7 -- 9 lanes of each 16-lane row add to nine consecutive floats of their row's entry (entry-major, stride 9)
//         2: as 1, entry-major with stride 8 for terms 0..7 and term 8 in an array of its own
//         3: as 1, but the four rows' entries are in four separate 32-bank-aligned regions, term-major (each row's nine lanes
//            hit banks slot .. slot + 8 of its own copy: no better than 1 in banks, a control)
//         4: as 1 with all 64 lanes active (the idle 28 on a dummy line each)
template <int OP, int PATTERN>
__global__ __launch_bounds__(256) void k_lds(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float buf[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) buf[i] = 0.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = lane >> 4, c = lane & 15;
  const uint32_t base = (uint32_t)(uintptr_t)(buf) + (uint32_t)w * 16384u;  // 4096 floats per wave (the 8-byte ops use addr * 2 - base)
  uint32_t rng = 12345u + (uint32_t)row * 977u + (uint32_t)w * 31u + blockIdx.x;
  float acc = 0.0f;
  const float val = 1.0f;
  const bool active = PATTERN == 0 || PATTERN == 4 || c < 9;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      rng = rng * 1664525u + 1013904223u;
      const uint32_t slot = (rng >> 16) % 200u;  // this row's entry in this step
      uint32_t addr;
      if (PATTERN == 0) addr = base + (uint32_t)lane * 4u + (slot & 7u) * 256u;
      else if (PATTERN == 1) addr = base + (slot * 9u + (uint32_t)c) * 4u;
      else if (PATTERN == 2) addr = base + (c < 8 ? slot * 8u + (uint32_t)c : 1600u + slot) * 4u;
      else if (PATTERN == 3) addr = base + ((uint32_t)c * 208u + slot) * 4u;
      else addr = base + (c < 9 ? slot * 9u + (uint32_t)c : 1808u + (uint32_t)lane) * 4u;
      if (active) {
        if (OP == 0) asm volatile("ds_add_f32 %0, %1" ::"v"(addr), "v"(val) : "memory");
        else if (OP == 1) asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(val) : "memory");
        else if (OP == 2) asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(1u) : "memory");
        else if (OP == 4) asm volatile("ds_add_f64 %0, %1" ::"v"(addr * 2u - base), "v"((double)val) : "memory");
        else if (OP == 5) asm volatile("ds_add_u64 %0, %1" ::"v"(addr * 2u - base), "v"(1ull) : "memory");
        else if (OP == 6) asm volatile("ds_max_f32 %0, %1" ::"v"(addr), "v"(val) : "memory");
        else if (OP == 7) asm volatile("ds_pk_add_f16 %0, %1" ::"v"(addr), "v"(0x3c003c00u) : "memory");
        else if (OP == 8) asm volatile("ds_write_b64 %0, %1" ::"v"(addr * 2u - base), "v"(1ull) : "memory");
        else {
          float r;
          asm volatile("ds_add_rtn_f32 %0, %1, %2" : "=v"(r) : "v"(addr), "v"(val) : "memory");
          asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
          acc += r;
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[blockIdx.x * 256 + threadIdx.x] = acc + buf[threadIdx.x];
}

template <int OP, int PATTERN>
void run(const char* name, float* out, int blocks, int iters, double mhz) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k_lds<OP, PATTERN><<<blocks, 256>>>(out, 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_lds<OP, PATTERN><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_cu = blocks * 4.0 / 256.0;
  const double cyc = ms * 1e-3 * mhz * 1e6 / (waves_per_cu * iters * 8.0);
  printf("{\"op\": \"%s\", \"ms\": %.4f, \"cycles_per_wave_instruction_per_CU_at_assumed_clock\": %.2f}\n", name, ms, cyc);
}

int main(int argc, char** argv) {
  const double mhz = argc > 1 ? atof(argv[1]) : 2400.0;
  const int blocks = 256 * 8, iters = 2048;
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  printf("{\"assumed_clock_MHz\": %.0f, \"waves_per_cu\": 32, \"note\": \"includes ~6 VALU/SALU of address arithmetic per instruction; compare rows with each other\"}\n", mhz);
  run<1, 0>("ds_write_b32, 64 lanes consecutive", out, blocks, iters, mhz);
  run<0, 0>("ds_add_f32, 64 lanes consecutive", out, blocks, iters, mhz);
  run<2, 0>("ds_add_u32, 64 lanes consecutive", out, blocks, iters, mhz);
  run<3, 0>("ds_add_rtn_f32, 64 lanes consecutive", out, blocks, iters, mhz);
  run<1, 1>("ds_write_b32, K7 pattern (4 rows x 9 lanes, stride 9)", out, blocks, iters, mhz);
  run<0, 1>("ds_add_f32, K7 pattern (4 rows x 9 lanes, stride 9)", out, blocks, iters, mhz);
  run<2, 1>("ds_add_u32, K7 pattern", out, blocks, iters, mhz);
  run<0, 2>("ds_add_f32, K7 lanes, stride 8 + term 8 apart", out, blocks, iters, mhz);
  run<0, 3>("ds_add_f32, K7 lanes, term-major", out, blocks, iters, mhz);
  run<0, 4>("ds_add_f32, K7 pattern, idle lanes on dummy words", out, blocks, iters, mhz);
  run<4, 0>("ds_add_f64, 64 lanes consecutive", out, blocks, iters, mhz);
  run<4, 1>("ds_add_f64, K7 pattern", out, blocks, iters, mhz);
  run<5, 0>("ds_add_u64, 64 lanes consecutive", out, blocks, iters, mhz);
  run<5, 1>("ds_add_u64, K7 pattern", out, blocks, iters, mhz);
  run<8, 1>("ds_write_b64, K7 pattern", out, blocks, iters, mhz);
  run<6, 0>("ds_max_f32, 64 lanes consecutive", out, blocks, iters, mhz);
  run<7, 0>("ds_pk_add_f16, 64 lanes consecutive", out, blocks, iters, mhz);
  hipFree(out);
  return 0;
}
