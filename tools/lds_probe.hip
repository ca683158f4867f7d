// lds_probe.hip -- what a ds_read costs the CU's LDS pipeline, by width and by address pattern (is a broadcast read,
// all 16 lanes of a DPP row on the same address as in the blend kernels' staged entries, cheaper than distinct
// addresses?).  8 waves per SIMD, every wave a long loop of independent reads; reported: cycles per wave64 read
// instruction per CU at the assumed clock.
//   build: hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o tools/_build/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// PATTERN 0: lane-distinct consecutive elements (conflict-free), 1: one address per 16-lane row, 2: one address per wave
template <int WIDTH, int PATTERN>
__global__ __launch_bounds__(256) void k_lds(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float buf[4096 + 64];
  for (int i = threadIdx.x; i < 4096 + 64; i += 256) buf[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int sel = PATTERN == 0 ? lane : (PATTERN == 1 ? (lane >> 4) * 5 : 0);
  uint32_t addr = (uint32_t)(uintptr_t)(buf) + (uint32_t)sel * (WIDTH * 4u) + (threadIdx.x >> 6) * 4096u;
  float acc = 0.0f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (WIDTH == 4) {
        float4 v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(0));
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        acc += v.x;
      } else if (WIDTH == 2) {
        float2 v;
        asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        acc += v.x;
      } else {
        float v;
        asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        acc += v;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int WIDTH, int PATTERN>
void run(const char* name, float* out, int blocks, int iters, double mhz) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k_lds<WIDTH, PATTERN><<<blocks, 256>>>(out, 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_lds<WIDTH, PATTERN><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_cu = blocks * 4.0 / 256.0;
  const double cyc = ms * 1e-3 * mhz * 1e6 / (waves_per_cu * iters * 16.0);
  printf("{\"read\": \"%s\", \"ms\": %.4f, \"cycles_per_wave_read_per_CU_at_assumed_clock\": %.2f}\n", name, ms, cyc);
}

int main(int argc, char** argv) {
  const double mhz = argc > 1 ? atof(argv[1]) : 2400.0;
  const int blocks = 256 * 8, iters = 2048;
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  printf("{\"assumed_clock_MHz\": %.0f, \"waves_per_cu\": 32}\n", mhz);
  run<1, 0>("ds_read_b32, lane-distinct", out, blocks, iters, mhz);
  run<1, 1>("ds_read_b32, one address per 16-lane row", out, blocks, iters, mhz);
  run<1, 2>("ds_read_b32, one address per wave", out, blocks, iters, mhz);
  run<2, 0>("ds_read_b64, lane-distinct", out, blocks, iters, mhz);
  run<2, 1>("ds_read_b64, one address per 16-lane row", out, blocks, iters, mhz);
  run<4, 0>("ds_read_b128, lane-distinct", out, blocks, iters, mhz);
  run<4, 1>("ds_read_b128, one address per 16-lane row", out, blocks, iters, mhz);
  run<4, 2>("ds_read_b128, one address per wave", out, blocks, iters, mhz);
  hipFree(out);
  return 0;
}
