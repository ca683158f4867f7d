// tools/queue_probe.hip -- what a BLOCKED cross-stream event wait on one HIP stream costs the kernels of another.
// Stream A runs N back-to-back kernels of `us` microseconds each (one wave per CU: they cannot be short of resources).
// Meanwhile stream B holds  (0) nothing  (1) hipStreamWaitEvent on an event that stream C records behind a 3 ms kernel,
// i.e. a barrier packet that stays blocked for the whole measurement  (2) a 3 ms kernel of its own, no barrier
// (3..5) as (1) with four / two / three such blocked streams  (6) four streams each holding a one-wave kernel that sleeps
// until a device flag is set (what a barrier packet does, done by a wave instead of the command processor).  Prints the time of A's sequence per kernel.
//   hipcc --offload-arch=gfx950 -O2 tools/queue_probe.hip -o tools/_build/queue_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void k_wait_flag(const int* flag, long long max_cycles) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && wall_clock64() - t0 < max_cycles) __builtin_amdgcn_s_sleep(64);
}
__global__ void k_set_flag(int* flag, int v) { __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_spin(long long cycles, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (sink != nullptr && threadIdx.x == 9999) *sink = 1;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main() {
  int rate_khz = 100000;  // wall_clock64 ticks at 100 MHz on gfx9
  hipStream_t A, B[4], C;
  int* flag = nullptr;
  CK(hipMalloc(&flag, 4));
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  for (auto& b : B) CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking));
  hipEvent_t ev, e0, e1;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int N = 40;
  for (int us : {5}) {
    const long long cyc = (long long)us * rate_khz / 1000;
    for (int mode = 0; mode < 7; mode++) {
      for (int rep = 0; rep < 3; rep++) {
        CK(hipDeviceSynchronize());
        if (mode == 6) {
          k_set_flag<<<1, 1, 0, C>>>(flag, 0);
          CK(hipDeviceSynchronize());
          for (int b = 0; b < 4; b++) k_wait_flag<<<1, 64, 0, B[b]>>>(flag, 5000ll * rate_khz / 1000);
          k_spin<<<1, 64, 0, C>>>(3000ll * rate_khz / 1000, nullptr);
          k_set_flag<<<1, 1, 0, C>>>(flag, 1);
        } else if (mode == 1 || mode >= 3) {
          k_spin<<<1, 64, 0, C>>>(3000ll * rate_khz / 1000, nullptr);
          CK(hipEventRecord(ev, C));
          for (int b = 0; b < (mode == 3 ? 4 : mode == 4 ? 2 : mode == 5 ? 3 : 1); b++) {
            CK(hipStreamWaitEvent(B[b], ev, 0));
            k_spin<<<1, 64, 0, B[b]>>>(1, nullptr);
          }
        } else if (mode == 2) {
          k_spin<<<1, 64, 0, B[0]>>>(3000ll * rate_khz / 1000, nullptr);
        }
        CK(hipEventRecord(e0, A));
        for (int i = 0; i < N; i++) k_spin<<<256, 64, 0, A>>>(cyc, nullptr);
        CK(hipEventRecord(e1, A));
        const auto h0 = std::chrono::steady_clock::now();
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipDeviceSynchronize());
        (void)h0;
        printf("{\"kernel_us\": %d, \"other_stream\": \"%s\", \"rep\": %d, \"us_per_kernel\": %.2f}\n", us,
               mode == 0 ? "idle" : mode == 1 ? "one blocked event wait" : mode == 2 ? "a running kernel" : mode == 3 ? "four blocked event waits" : mode == 4 ? "two blocked event waits" : mode == 5 ? "three blocked event waits" : "four one-wave kernels sleeping on a flag",
               rep, 1000.0f * ms / N);
      }
    }
  }
  return 0;
}
