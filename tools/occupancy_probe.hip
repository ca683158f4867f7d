// tools/occupancy_probe.hip -- how many 256-thread workgroups with S bytes of static + D bytes of dynamic LDS does a CU of this
// device admit (hipOccupancyMaxActiveBlocksPerMultiprocessor)?  Answers the LDS allocation granularity question behind the
// forward blend's "blend_lds_pad" (gcr_api.hip FRAME PIPELINE): 19 456 B static -> 8 per CU; which pad gives 7, 6, 5?
//   hipcc --offload-arch=gfx950 -O2 tools/occupancy_probe.hip -o tools/_build/occupancy_probe && tools/_build/occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_lds(float* out) {
  __shared__ float s[19456 / 4];
  extern __shared__ float dyn[];
  s[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = s[(threadIdx.x * 7) % 4864] + (dyn != nullptr ? 1.0f : 0.0f);
}
__global__ __launch_bounds__(256) void k_small(float* out) {
  extern __shared__ float dyn[];
  out[blockIdx.x * 256 + threadIdx.x] = dyn != nullptr ? 1.0f : 0.0f;
}
int main() {
  int last = -1;
  for (int pad = 0; pad <= 16384; pad += 128) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_lds, 256, (size_t)pad) != hipSuccess) { printf("query failed\n"); return 1; }
    if (n != last) printf("{\"static\": 19456, \"dynamic_pad\": %d, \"total\": %d, \"blocks_per_cu\": %d}\n", pad, 19456 + pad, n);
    last = n;
  }
  last = -1;
  for (int dynb = 0; dynb <= 65536; dynb += 256) {
    int n = 0;
    (void)hipFuncSetAttribute((const void*)k_small, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_small, 256, (size_t)dynb) != hipSuccess) { printf("query failed\n"); return 1; }
    if (n != last) printf("{\"static\": 0, \"dynamic\": %d, \"blocks_per_cu\": %d}\n", dynb, n);
    last = n;
  }
  return 0;
}
