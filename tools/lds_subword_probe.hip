// lds_subword_probe.hip -- do the sub-dword LDS operations of the blend kernels' list handling (ds_read_u8 per walk step in
// K6 / K7, ds_write_b8 while the lists are built) run at the rate of their dword forms?  Same harness as lds_atomic_probe.hip.
//   build: hipcc --offload-arch=gfx950 -O3 tools/lds_subword_probe.hip -o tools/_build/lds_subword_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// OP 0 ds_read_b32, 1 ds_read_u16, 2 ds_read_u8, 3 ds_write_b32, 4 ds_write_b16, 5 ds_write_b8, 6 ds_read_b64, 7 ds_read_b96
// PATTERN 0: lane-distinct consecutive elements, 1: one element per 8-lane sub-row (K6's list reads), 2: scattered bytes (list build)
template <int OP, int PATTERN>
__global__ __launch_bounds__(256) void k_lds(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char buf[32768];
  for (int i = threadIdx.x; i < 32768 / 4; i += 256) reinterpret_cast<uint32_t*>(buf)[i] = (uint32_t)i;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t esz = (OP == 2 || OP == 5) ? 1u : ((OP == 1 || OP == 4) ? 2u : (OP == 6 ? 8u : (OP == 7 ? 16u : 4u)));
  uint32_t sel = PATTERN == 0 ? (uint32_t)lane : (PATTERN == 1 ? (uint32_t)(lane >> 3) * 232u / (esz > 1 ? 1u : 1u) : (uint32_t)((lane * 37) & 255));
  const uint32_t base = (uint32_t)(uintptr_t)(buf) + (uint32_t)w * 8192u;
  uint32_t acc = 0;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const uint32_t addr = base + ((sel * esz + (uint32_t)k * 16u * esz) & 4095u & ~(esz > 4 ? 7u : esz - 1u));
      uint32_t v = 1u;
      if (OP == 0) asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
      else if (OP == 1) asm volatile("ds_read_u16 %0, %1" : "=v"(v) : "v"(addr));
      else if (OP == 2) asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(addr));
      else if (OP == 3) asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
      else if (OP == 4) asm volatile("ds_write_b16 %0, %1" ::"v"(addr), "v"(v) : "memory");
      else if (OP == 5) asm volatile("ds_write_b8 %0, %1" ::"v"(addr), "v"(v) : "memory");
      else if (OP == 6) { unsigned long long q; asm volatile("ds_read_b64 %0, %1" : "=v"(q) : "v"(addr)); v = (uint32_t)q; }
      else { typedef uint32_t u3 __attribute__((ext_vector_type(3))); u3 q; asm volatile("ds_read_b96 %0, %1" : "=v"(q) : "v"(addr)); v = q.x; }
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      acc += v;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[blockIdx.x * 256 + threadIdx.x] = (float)acc;
}

template <int OP, int PATTERN>
void run(const char* name, float* out, int blocks, int iters, double mhz) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k_lds<OP, PATTERN><<<blocks, 256>>>(out, 16);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k_lds<OP, PATTERN><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_cu = blocks * 4.0 / 256.0;
  const double cyc = ms * 1e-3 * mhz * 1e6 / (waves_per_cu * iters * 16.0);
  printf("{\"op\": \"%s\", \"ms\": %.4f, \"cycles_per_wave_instruction_per_CU_at_assumed_clock\": %.2f}\n", name, ms, cyc);
}

int main(int argc, char** argv) {
  const double mhz = argc > 1 ? atof(argv[1]) : 2400.0;
  const int blocks = 256 * 8, iters = 2048;
  float* out;
  (void)hipMalloc(&out, sizeof(float) * blocks * 256);
  printf("{\"assumed_clock_MHz\": %.0f, \"waves_per_cu\": 32}\n", mhz);
  run<0, 0>("ds_read_b32, lane-distinct", out, blocks, iters, mhz);
  run<1, 0>("ds_read_u16, lane-distinct", out, blocks, iters, mhz);
  run<2, 0>("ds_read_u8, lane-distinct", out, blocks, iters, mhz);
  run<2, 1>("ds_read_u8, one byte per 8-lane sub-row (K6 list read)", out, blocks, iters, mhz);
  run<1, 1>("ds_read_u16, one element per 8-lane sub-row", out, blocks, iters, mhz);
  run<6, 0>("ds_read_b64, lane-distinct", out, blocks, iters, mhz);
  run<7, 0>("ds_read_b96, lane-distinct", out, blocks, iters, mhz);
  run<3, 0>("ds_write_b32, lane-distinct", out, blocks, iters, mhz);
  run<4, 0>("ds_write_b16, lane-distinct", out, blocks, iters, mhz);
  run<5, 0>("ds_write_b8, lane-distinct", out, blocks, iters, mhz);
  run<5, 2>("ds_write_b8, scattered bytes (list build)", out, blocks, iters, mhz);
  (void)hipFree(out);
  return 0;
}
