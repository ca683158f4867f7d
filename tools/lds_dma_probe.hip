#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const uint32_t* __restrict__ src, uint32_t* out, int n) {
  __shared__ uint32_t s[256];
  __shared__ uint32_t h[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t* p = src + min(tid, n - 1);
  __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(s + w * 64), 4, 0, 0);
  if (w == 0 && lane >= 1 && lane < 4) __builtin_amdgcn_global_load_lds((gptr_t)(src + 100 + lane), (lptr_t)h, 4, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[tid] = s[tid] + h[tid & 3];
}
int main() {
  uint32_t *d, *o; hipMalloc(&d, 4096); hipMalloc(&o, 1024);
  uint32_t hsrc[1024]; for (int i = 0; i < 1024; i++) hsrc[i] = i * 3 + 1;
  hipMemcpy(d, hsrc, 4096, hipMemcpyHostToDevice);
  hipMemset(o, 0, 1024);
  k<<<1, 256>>>(d, o, 200);
  uint32_t ho[256]; hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; i++) {
    uint32_t exp_s = hsrc[i < 200 ? i : 199];
    uint32_t hh[4] = {0, hsrc[101], hsrc[102], hsrc[103]};
    if ((i & 3) != 0 && ho[i] != exp_s + hh[i & 3]) { bad++; if (bad < 5) printf("i=%d got %u want %u\n", i, ho[i], exp_s + hh[i&3]); }
  }
  printf("bad=%d (h[0] lane inactive: out[0]-s[0]=%u)\n", bad, ho[0] - hsrc[0]);
  return bad != 0;
}
