// gce_grid.hip -- MI355X (gfx950) kernels + C ABI (include/gce.h) of the multi-resolution hash-grid
// encoder.  Reference behaviour: extensions/grid_encoder/grid_encoder_ext.cu ("ge/").
//
// The op is a gather (forward) / scatter-add (backward) over a table of L levels x <= 2^19 rows x C
// floats (268 MB for GaussianCity's 16 x 2^19 x 8): HBM/L2-latency bound, no contraction to speak of.
// Layout choices for gfx950:
//   * grid = (points / 256, levels), level-major like upstream, so concurrently resident workgroups
//     mostly work on ONE level: its 16 MB slice stays in the 32 MB of L2 while it is being gathered;
//   * a row of C = 8 floats is 32 B = two dwordx4 loads; the 2^D corner indices are computed once and
//     kept in registers, so the input-gradient pass (D * 2^(D-1) left/right pairs) re-reads rows that
//     are L1/L2 hits instead of re-hashing (upstream hashes them again);
//   * backward uses hardware global_atomic_add_f32 (the library is built with -munsafe-fp-atomics).
// Arithmetic follows the reference's association order without contraction (gce-fp32-v1), so forward
// and dy_dx are bit-comparable with oracle/gce_oracle.c.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/gce.h"

namespace {

thread_local std::string g_err;
std::atomic<int> g_timing{0};

int fail(int code, const char* msg) {
  g_err = msg;
  return code;
}
int fail_hip(hipError_t e, const char* where) {
  g_err = std::string(where) + ": " + hipGetErrorString(e);
  return GCE_ERR_HIP;
}
#define HIP_TRY(expr, where)                          \
  do {                                                \
    hipError_t e_ = (expr);                           \
    if (e_ != hipSuccess) return fail_hip(e_, where); \
  } while (0)

enum Stage { ST_FWD = 0, ST_BWD_EMB, ST_BWD_IN, ST_N };
struct StageSlot {
  hipEvent_t a = nullptr, b = nullptr;
  bool pending = false;
  double ms = 0.0;
  int n = 0;
};
StageSlot g_slots[ST_N];
void stage_resolve(StageSlot& s) {
  if (!s.pending) return;
  if (hipEventSynchronize(s.b) == hipSuccess) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
      s.ms += ms;
      s.n++;
    }
  }
  s.pending = false;
}
struct StageTimer {
  hipStream_t s;
  StageSlot* sl = nullptr;
  StageTimer(hipStream_t s_, int stage) : s(s_) {
    if (g_timing.load() == 0) return;
    sl = &g_slots[stage];
    if (!sl->a) {
      (void)hipEventCreate(&sl->a);
      (void)hipEventCreate(&sl->b);
    }
    stage_resolve(*sl);
    (void)hipEventRecord(sl->a, s);
  }
  ~StageTimer() {
    if (!sl) return;
    (void)hipEventRecord(sl->b, s);
    sl->pending = true;
  }
};

struct LevelScales {
  float v[GCE_MAX_LEVELS];
};

template <int D>
__device__ __forceinline__ uint32_t fast_hash(const uint32_t (&pos_grid)[D]) {  // ge/:52-69
  constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < D; i++) r ^= pos_grid[i] * primes[i];
  return r;
}

// ge/:71-95 without the channel term: first float of the row
template <int D, int C>
__device__ __forceinline__ uint32_t grid_index(uint32_t gridtype, bool align_corners, uint32_t hashmap_size,
                                               uint32_t resolution, const uint32_t (&pos_grid)[D]) {
  uint32_t stride = 1, index = 0;
#pragma unroll
  for (int d = 0; d < D; d++) {
    if (stride <= hashmap_size) {
      index += pos_grid[d] * stride;
      stride *= align_corners ? resolution : (resolution + 1);
    }
  }
  if (gridtype == 0 && stride > hashmap_size) index = fast_hash<D>(pos_grid);
  return (index % hashmap_size) * C;
}

template <int C>
struct Row {
  float v[C];
};
template <int C>
__device__ __forceinline__ Row<C> load_row(const float* __restrict__ p) {
  Row<C> r;
  if constexpr (C == 8) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  } else if constexpr (C == 4) {
    const float4 a = reinterpret_cast<const float4*>(p)[0];
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  } else if constexpr (C == 2) {
    const float2 a = reinterpret_cast<const float2*>(p)[0];
    r.v[0] = a.x; r.v[1] = a.y;
  } else {
    r.v[0] = p[0];
  }
  return r;
}

// in-range test + cell / fraction (ge/:113-145, :281-296)
template <int D>
__device__ __forceinline__ bool locate(const float* __restrict__ in, float scale, bool align_corners, float (&pos)[D],
                                       uint32_t (&pos_grid)[D]) {
  bool oob = false;
#pragma unroll
  for (int d = 0; d < D; d++) {
    pos[d] = in[d];
    if (pos[d] < 0 || pos[d] > 1) oob = true;
  }
  if (oob) return false;
#pragma unroll
  for (int d = 0; d < D; d++) {
    pos[d] = pos[d] * scale + (align_corners ? 0.0f : 0.5f);
    pos_grid[d] = (uint32_t)__builtin_floorf(pos[d]);
    pos[d] -= (float)pos_grid[d];
  }
  return true;
}

// ------------------------------------------------------------------------------------------- K9
// ge/:97-256 (kernel_grid): thread = (point, level).  Every corner row is gathered ONCE: the 2^D rows
// of a channel group (CG = min(C, 4) channels = one dwordx4 / dwordx2 / dword per row) sit in registers
// while both the interpolation and the D input-gradient sums (left/right pairs, ge/:192-254) are formed
// from them; C = 8 takes two such passes.  Upstream (and the first version here: 0.89 ms for 16k points)
// re-fetches 2 * D * 2^(D-1) rows for the gradient pass.  Channels are independent, so splitting them
// changes no operation order: outputs and dy_dx stay bit-identical to the oracle.
template <int CG>
struct RowG {
  float v[CG];
};
template <int CG>
__device__ __forceinline__ RowG<CG> load_rowg(const float* __restrict__ p) {
  RowG<CG> r;
  if constexpr (CG == 4) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  } else if constexpr (CG == 2) {
    const float2 a = *reinterpret_cast<const float2*>(p);
    r.v[0] = a.x; r.v[1] = a.y;
  } else {
    r.v[0] = p[0];
  }
  return r;
}

template <int D, int C>
__global__ __launch_bounds__(256) void k_grid_fwd(const float* __restrict__ inputs, const float* __restrict__ grid,
                                                  const int32_t* __restrict__ offsets, float* __restrict__ outputs,
                                                  uint32_t B, uint32_t L, const LevelScales scales, bool calc_grad_inputs,
                                                  float* __restrict__ dy_dx, uint32_t gridtype, bool align_corners) {
  constexpr int CG = C < 4 ? C : 4;
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const uint32_t level = blockIdx.y;
  const uint32_t off0 = (uint32_t)offsets[level], off1 = (uint32_t)offsets[level + 1];
  const float* __restrict__ g = grid + (size_t)off0 * C;
  float* __restrict__ out = outputs + ((size_t)level * B + b) * C;
  float* __restrict__ dd = dy_dx + ((size_t)b * L + level) * D * C;  // B L D C
  const float scale = scales.v[level];

  float pos[D];
  uint32_t pos_grid[D];
  if (!locate<D>(inputs + (size_t)b * D, scale, align_corners, pos, pos_grid)) {
#pragma unroll
    for (int ch = 0; ch < C; ch++) out[ch] = 0;
    if (calc_grad_inputs) {
#pragma unroll
      for (int i = 0; i < D * C; i++) dd[i] = 0;
    }
    return;
  }
  const uint32_t hashmap_size = off1 - off0;
  const uint32_t resolution = (uint32_t)ceil(scale) + 1;

  uint32_t corner[1 << D];  // row offsets of the 2^D corners (bit d of the corner number = +1 along d)
#pragma unroll
  for (uint32_t idx = 0; idx < (1u << D); idx++) {
    uint32_t pl[D];
#pragma unroll
    for (int d = 0; d < D; d++) pl[d] = pos_grid[d] + ((idx >> d) & 1u);
    corner[idx] = grid_index<D, C>(gridtype, align_corners, hashmap_size, resolution, pl);
  }

#pragma unroll
  for (int c0 = 0; c0 < C; c0 += CG) {
    RowG<CG> row[1 << D];
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) row[idx] = load_rowg<CG>(g + corner[idx] + c0);

    float results[CG];
#pragma unroll
    for (int ch = 0; ch < CG; ch++) results[ch] = 0;
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {  // ge/:152-180
      float w = 1;
#pragma unroll
      for (int d = 0; d < D; d++) w *= ((idx & (1u << d)) == 0) ? 1 - pos[d] : pos[d];
#pragma unroll
      for (int ch = 0; ch < CG; ch++) results[ch] += w * row[idx].v[ch];
    }
#pragma unroll
    for (int ch = 0; ch < CG; ch++) out[c0 + ch] = results[ch];

    if (calc_grad_inputs) {  // ge/:192-254
#pragma unroll
      for (int gd = 0; gd < D; gd++) {
        float rg[CG];
#pragma unroll
        for (int ch = 0; ch < CG; ch++) rg[ch] = 0;
#pragma unroll
        for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
          float w = scale;
          uint32_t cidx = 0;  // corner number with bit gd clear
#pragma unroll
          for (int nd = 0; nd < D - 1; nd++) {
            const int d = (nd >= gd) ? (nd + 1) : nd;
            if ((idx & (1u << nd)) == 0) {
              w *= 1 - pos[d];
            } else {
              w *= pos[d];
              cidx |= 1u << d;
            }
          }
#pragma unroll
          for (int ch = 0; ch < CG; ch++) rg[ch] += w * (row[cidx | (1u << gd)].v[ch] - row[cidx].v[ch]);
        }
#pragma unroll
        for (int ch = 0; ch < CG; ch++) dd[gd * C + c0 + ch] = rg[ch];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------- K10
// ge/:258-337 (kernel_grid_backward): thread = (point, channel) with the channel fastest, so the C
// lanes of one point add into the C consecutive floats of a row: a wave's atomic instruction touches
// 64/C rows x 4C bytes instead of 64 scattered dwords (first version, one thread per point: 3.4 ms for
// 16k points; upstream uses two channels per thread for the same reason).  Each lane redoes the hash --
// integer ALU is free next to 2^D atomics.
template <int D, int C>
__global__ __launch_bounds__(256) void k_grid_bwd(const float* __restrict__ grad, const float* __restrict__ inputs,
                                                  const int32_t* __restrict__ offsets, float* __restrict__ grad_grid,
                                                  uint32_t B, uint32_t L, const LevelScales scales, uint32_t gridtype,
                                                  bool align_corners) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t b = (uint32_t)(t / C), ch = (uint32_t)(t % C);
  if (b >= B) return;
  const uint32_t level = blockIdx.y;
  const uint32_t off0 = (uint32_t)offsets[level], off1 = (uint32_t)offsets[level + 1];
  float* __restrict__ gg = grad_grid + (size_t)off0 * C + ch;
  const float scale = scales.v[level];
  float pos[D];
  uint32_t pos_grid[D];
  if (!locate<D>(inputs + (size_t)b * D, scale, align_corners, pos, pos_grid)) return;  // grad stays 0
  const uint32_t hashmap_size = off1 - off0;
  const uint32_t resolution = (uint32_t)ceil(scale) + 1;
  const float gc = grad[((size_t)level * B + b) * C + ch];
#pragma unroll
  for (uint32_t idx = 0; idx < (1u << D); idx++) {
    float w = 1;
    uint32_t pl[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
      if ((idx & (1u << d)) == 0) {
        w *= 1 - pos[d];
        pl[d] = pos_grid[d];
      } else {
        w *= pos[d];
        pl[d] = pos_grid[d] + 1;
      }
    }
    atomicAdd(&gg[grid_index<D, C>(gridtype, align_corners, hashmap_size, resolution, pl)], w * gc);
  }
}

// ------------------------------------------------------------------------------------------- K11
// ge/:339-366 (kernel_input_backward): thread = (point, input dim), sequential over levels and channels
template <int D, int C>
__global__ __launch_bounds__(256) void k_input_bwd(const float* __restrict__ grad, const float* __restrict__ dy_dx,
                                                   float* __restrict__ grad_inputs, uint32_t B, uint32_t L) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * D) return;
  const uint32_t b = t / D, d = t - b * D;
  const float* __restrict__ dd = dy_dx + (size_t)b * L * D * C;
  float r = 0;
  for (uint32_t l = 0; l < L; l++) {
#pragma unroll
    for (int ch = 0; ch < C; ch++) r += grad[((size_t)l * B + b) * C + ch] * dd[(l * D + d) * C + ch];
  }
  grad_inputs[t] = r;
}

// ------------------------------------------------------------------------------ K9t / K10t / K11t: half and double
// Upstream dispatches its three kernels over the embeddings' dtype (AT_DISPATCH_FLOATING_TYPES_AND_HALF, ge/:555,597):
// inputs and the interpolation weights stay float, embeddings / outputs / dy_dx / grad / grad_embeddings / grad_inputs are
// scalar_t, and every accumulator is a scalar_t -- for at::Half that means a rounding to binary16 after every product and
// every sum (c10::Half: float * Half -> float, Half += float converts the float first, Half + Half rounds once).  GaussianCity
// itself never leaves float32, so these are plain one-thread-per-(point, level) kernels, not the register-resident float
// version above; what they keep is the reference's arithmetic, operation for operation (gce-f16-v1 / gce-f64-v1,
// oracle/grid_oracle_typed.py), so outputs and dy_dx are bit-comparable here too.
template <typename T>
struct GceOps;
template <>
struct GceOps<double> {
  static __device__ __forceinline__ double mulw(float w, double g) { return (double)w * g; }  // float * double -> double
  static __device__ __forceinline__ double add(double a, double b) { return a + b; }
  static __device__ __forceinline__ double sub(double a, double b) { return a - b; }
  static __device__ __forceinline__ double mul(double a, double b) { return a * b; }
  static __device__ __forceinline__ void atomic_add(double* base, size_t i, double v) { atomicAdd(base + i, v); }
};
template <>
struct GceOps<_Float16> {
  // float * Half -> float (rounded to binary32), THEN Half(float): two roundings, as c10 does it.  The opaque barrier keeps
  // the compiler from folding the pair into one v_fma_mixlo_f16 (exact product, one rounding to binary16): 1 result in 7 000
  // differs by an ulp, found by the bit-exact test.  (Half +/- Half and Half * Half are exact in binary32: no such case.)
  static __device__ __forceinline__ _Float16 mulw(float w, _Float16 g) {
    float p = w * (float)g;
    asm volatile("" : "+v"(p));
    return (_Float16)p;
  }
  static __device__ __forceinline__ _Float16 add(_Float16 a, _Float16 b) { return (_Float16)((float)a + (float)b); }
  static __device__ __forceinline__ _Float16 sub(_Float16 a, _Float16 b) { return (_Float16)((float)a - (float)b); }
  static __device__ __forceinline__ _Float16 mul(_Float16 a, _Float16 b) { return (_Float16)((float)a * (float)b); }
  // one binary16 add through the packed atomic (global_atomic_pk_add_f16): the element's pair partner gets +0
  static __device__ __forceinline__ void atomic_add(_Float16* base, size_t i, _Float16 v) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 pv;
    pv[0] = (i & 1) ? (_Float16)0.0f : v;
    pv[1] = (i & 1) ? v : (_Float16)0.0f;
    __builtin_amdgcn_global_atomic_fadd_v2f16(reinterpret_cast<h2*>(base + (i & ~(size_t)1)), pv);
  }
};

template <typename T, int D, int C>
__global__ __launch_bounds__(256) void k_grid_fwd_t(const float* __restrict__ inputs, const T* __restrict__ grid,
                                                    const int32_t* __restrict__ offsets, T* __restrict__ outputs, uint32_t B,
                                                    uint32_t L, const LevelScales scales, bool calc_grad_inputs,
                                                    T* __restrict__ dy_dx, uint32_t gridtype, bool align_corners) {
  using O = GceOps<T>;
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const uint32_t level = blockIdx.y;
  const uint32_t off0 = (uint32_t)offsets[level], off1 = (uint32_t)offsets[level + 1];
  const T* __restrict__ g = grid + (size_t)off0 * C;
  T* __restrict__ out = outputs + ((size_t)level * B + b) * C;
  T* __restrict__ dd = dy_dx + ((size_t)b * L + level) * D * C;
  const float scale = scales.v[level];
  float pos[D];
  uint32_t pos_grid[D];
  if (!locate<D>(inputs + (size_t)b * D, scale, align_corners, pos, pos_grid)) {
#pragma unroll
    for (int ch = 0; ch < C; ch++) out[ch] = (T)0.0f;
    if (calc_grad_inputs) {
#pragma unroll
      for (int i = 0; i < D * C; i++) dd[i] = (T)0.0f;
    }
    return;
  }
  const uint32_t hashmap_size = off1 - off0;
  const uint32_t resolution = (uint32_t)ceil(scale) + 1;
  T results[C];
#pragma unroll
  for (int ch = 0; ch < C; ch++) results[ch] = (T)0.0f;
#pragma unroll
  for (uint32_t idx = 0; idx < (1u << D); idx++) {  // ge/:152-180
    float w = 1;
    uint32_t pl[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
      w *= ((idx & (1u << d)) == 0) ? 1 - pos[d] : pos[d];
      pl[d] = pos_grid[d] + ((idx >> d) & 1u);
    }
    const uint32_t index = grid_index<D, C>(gridtype, align_corners, hashmap_size, resolution, pl);
#pragma unroll
    for (int ch = 0; ch < C; ch++) results[ch] = O::add(results[ch], O::mulw(w, g[index + ch]));
  }
#pragma unroll
  for (int ch = 0; ch < C; ch++) out[ch] = results[ch];
  if (calc_grad_inputs) {  // ge/:192-254
#pragma unroll
    for (int gd = 0; gd < D; gd++) {
      T rg[C];
#pragma unroll
      for (int ch = 0; ch < C; ch++) rg[ch] = (T)0.0f;
#pragma unroll
      for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
        float w = scale;
        uint32_t pl[D];
#pragma unroll
        for (int nd = 0; nd < D - 1; nd++) {
          const int d = (nd >= gd) ? (nd + 1) : nd;
          if ((idx & (1u << nd)) == 0) {
            w *= 1 - pos[d];
            pl[d] = pos_grid[d];
          } else {
            w *= pos[d];
            pl[d] = pos_grid[d] + 1;
          }
        }
        pl[gd] = pos_grid[gd];
        const uint32_t il = grid_index<D, C>(gridtype, align_corners, hashmap_size, resolution, pl);
        pl[gd] = pos_grid[gd] + 1;
        const uint32_t ir = grid_index<D, C>(gridtype, align_corners, hashmap_size, resolution, pl);
#pragma unroll
        for (int ch = 0; ch < C; ch++) rg[ch] = O::add(rg[ch], O::mulw(w, O::sub(g[ir + ch], g[il + ch])));
      }
#pragma unroll
      for (int ch = 0; ch < C; ch++) dd[gd * C + ch] = rg[ch];
    }
  }
}

template <typename T, int D, int C>
__global__ __launch_bounds__(256) void k_grid_bwd_t(const T* __restrict__ grad, const float* __restrict__ inputs,
                                                    const int32_t* __restrict__ offsets, T* __restrict__ grad_grid, uint32_t B,
                                                    uint32_t L, const LevelScales scales, uint32_t gridtype,
                                                    bool align_corners) {
  using O = GceOps<T>;
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t b = (uint32_t)(t / C), ch = (uint32_t)(t % C);
  if (b >= B) return;
  const uint32_t level = blockIdx.y;
  const uint32_t off0 = (uint32_t)offsets[level], off1 = (uint32_t)offsets[level + 1];
  const float scale = scales.v[level];
  float pos[D];
  uint32_t pos_grid[D];
  if (!locate<D>(inputs + (size_t)b * D, scale, align_corners, pos, pos_grid)) return;  // grad stays 0
  const uint32_t hashmap_size = off1 - off0;
  const uint32_t resolution = (uint32_t)ceil(scale) + 1;
  const T gc = grad[((size_t)level * B + b) * C + ch];
#pragma unroll
  for (uint32_t idx = 0; idx < (1u << D); idx++) {
    float w = 1;
    uint32_t pl[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
      if ((idx & (1u << d)) == 0) {
        w *= 1 - pos[d];
        pl[d] = pos_grid[d];
      } else {
        w *= pos[d];
        pl[d] = pos_grid[d] + 1;
      }
    }
    const size_t i = (size_t)off0 * C + grid_index<D, C>(gridtype, align_corners, hashmap_size, resolution, pl) + ch;
    O::atomic_add(grad_grid, i, O::mulw(w, gc));  // ge/:313-335: w * grad rounded to scalar_t, then one scalar_t add
  }
}

template <typename T, int D, int C>
__global__ __launch_bounds__(256) void k_input_bwd_t(const T* __restrict__ grad, const T* __restrict__ dy_dx,
                                                     T* __restrict__ grad_inputs, uint32_t B, uint32_t L) {
  using O = GceOps<T>;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * D) return;
  const uint32_t b = t / D, d = t - b * D;
  const T* __restrict__ dd = dy_dx + (size_t)b * L * D * C;
  T r = (T)0.0f;
  for (uint32_t l = 0; l < L; l++) {
#pragma unroll
    for (int ch = 0; ch < C; ch++) r = O::add(r, O::mul(grad[((size_t)l * B + b) * C + ch], dd[(l * D + d) * C + ch]));
  }
  grad_inputs[t] = r;
}

int check_dims(uint32_t B, uint32_t D, uint32_t C, uint32_t L) {
  (void)B;
  if (D < 2 || D > 5) return fail(GCE_ERR_UNSUPPORTED, "GridEncoding: D must be 2, 3, 4, or 5.");  // ge/:441-456
  if (C != 1 && C != 2 && C != 4 && C != 8) return fail(GCE_ERR_UNSUPPORTED, "GridEncoding: C must be 1, 2, 4, or 8.");  // ge/:399
  if (L < 1 || L > GCE_MAX_LEVELS) return fail(GCE_ERR_UNSUPPORTED, "GridEncoding: L must be in [1, 32]");
  return 0;
}

#define GCE_DISPATCH_DC(D_, C_, CALL)                         \
  switch ((D_)*16 + (C_)) {                                   \
    case 2 * 16 + 1: { constexpr int D = 2, C = 1; CALL; } break; \
    case 2 * 16 + 2: { constexpr int D = 2, C = 2; CALL; } break; \
    case 2 * 16 + 4: { constexpr int D = 2, C = 4; CALL; } break; \
    case 2 * 16 + 8: { constexpr int D = 2, C = 8; CALL; } break; \
    case 3 * 16 + 1: { constexpr int D = 3, C = 1; CALL; } break; \
    case 3 * 16 + 2: { constexpr int D = 3, C = 2; CALL; } break; \
    case 3 * 16 + 4: { constexpr int D = 3, C = 4; CALL; } break; \
    case 3 * 16 + 8: { constexpr int D = 3, C = 8; CALL; } break; \
    case 4 * 16 + 1: { constexpr int D = 4, C = 1; CALL; } break; \
    case 4 * 16 + 2: { constexpr int D = 4, C = 2; CALL; } break; \
    case 4 * 16 + 4: { constexpr int D = 4, C = 4; CALL; } break; \
    case 4 * 16 + 8: { constexpr int D = 4, C = 8; CALL; } break; \
    case 5 * 16 + 1: { constexpr int D = 5, C = 1; CALL; } break; \
    case 5 * 16 + 2: { constexpr int D = 5, C = 2; CALL; } break; \
    case 5 * 16 + 4: { constexpr int D = 5, C = 4; CALL; } break; \
    case 5 * 16 + 8: { constexpr int D = 5, C = 8; CALL; } break; \
    default: return fail(GCE_ERR_UNSUPPORTED, "GridEncoding: unsupported (D, C)"); \
  }

}  // namespace

// ---- typed entry points (ABI v2): dtype of embeddings / outputs / dy_dx / grad / grad_embeddings / grad_inputs
template <typename T>
static int forward_typed(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs, uint32_t B,
                         uint32_t D, uint32_t C, uint32_t L, const LevelScales& sc, int calc_grad_inputs, void* dy_dx,
                         uint32_t gridtype, int align_corners, hipStream_t s) {
  const dim3 grid((B + 255) / 256, L, 1);
  StageTimer t(s, ST_FWD);
  GCE_DISPATCH_DC(D, C, (k_grid_fwd_t<T, D, C><<<grid, 256, 0, s>>>(inputs, (const T*)embeddings, offsets, (T*)outputs, B, L, sc,
                                                                  calc_grad_inputs != 0, (T*)dy_dx, gridtype, align_corners != 0)))
  return 0;
}
template <typename T>
static int backward_typed(const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings, uint32_t B,
                          uint32_t D, uint32_t C, uint32_t L, const LevelScales& sc, int calc_grad_inputs, const void* dy_dx,
                          void* grad_inputs, uint32_t gridtype, int align_corners, hipStream_t s) {
  const dim3 grid((unsigned)(((uint64_t)B * C + 255) / 256), L, 1);
  {
    StageTimer t(s, ST_BWD_EMB);
    GCE_DISPATCH_DC(D, C, (k_grid_bwd_t<T, D, C><<<grid, 256, 0, s>>>((const T*)grad, inputs, offsets, (T*)grad_embeddings, B, L, sc,
                                                                    gridtype, align_corners != 0)))
  }
  if (calc_grad_inputs) {
    StageTimer t(s, ST_BWD_IN);
    GCE_DISPATCH_DC(D, C, (k_input_bwd_t<T, D, C><<<(B * D + 255) / 256, 256, 0, s>>>((const T*)grad, (const T*)dy_dx,
                                                                                    (T*)grad_inputs, B, L)))
  }
  return 0;
}

extern "C" {

int gce_abi_version(void) { return GCE_ABI_VERSION; }
const char* gce_last_error(void) { return g_err.c_str(); }

int gce_set_option(const char* name, int value) {
  if (!name) return -1;
  if (!strcmp(name, "timing")) return g_timing.exchange(value);
  return -1;
}
int gce_get_stage_ms(float* out, int n) {
  if (!out) return 0;
  int k = 0;
  for (; k < n && k < ST_N; k++) {
    stage_resolve(g_slots[k]);
    out[k] = g_slots[k].n ? (float)(g_slots[k].ms / g_slots[k].n) : 0.0f;
    g_slots[k].ms = 0.0;
    g_slots[k].n = 0;
  }
  return k;
}

int gce_level_scales(uint32_t L, float S, uint32_t H, float* scales_host) {
  if (!scales_host || L > GCE_MAX_LEVELS) return fail(GCE_ERR_INVALID_ARGUMENT, "gce_level_scales: bad arguments");
  for (uint32_t l = 0; l < L; l++) scales_host[l] = exp2f((float)l * S) * (float)H - 1.0f;
  return 0;
}

int gce_forward(const float* inputs, const float* embeddings, const int32_t* offsets, float* outputs, uint32_t B,
                uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, float* dy_dx,
                uint32_t gridtype, int align_corners, void* hip_stream) {
  if (int rc = check_dims(B, D, C, L)) return rc;
  if (B == 0) return 0;
  if (!inputs || !embeddings || !offsets || !outputs) return fail(GCE_ERR_INVALID_ARGUMENT, "gce_forward: null tensor");
  if (calc_grad_inputs && !dy_dx) return fail(GCE_ERR_INVALID_ARGUMENT, "gce_forward: calc_grad_inputs needs dy_dx");
  LevelScales sc;
  gce_level_scales(L, S, H, sc.v);
  hipStream_t s = (hipStream_t)hip_stream;
  const dim3 grid((B + 255) / 256, L, 1);
  {
    StageTimer t(s, ST_FWD);
    GCE_DISPATCH_DC(D, C, (k_grid_fwd<D, C><<<grid, 256, 0, s>>>(inputs, embeddings, offsets, outputs, B, L, sc,
                                                               calc_grad_inputs != 0, dy_dx, gridtype, align_corners != 0)))
  }
  HIP_TRY(hipGetLastError(), "grid forward launch");
  return 0;
}

int gce_backward(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets,
                 float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                 int calc_grad_inputs, const float* dy_dx, float* grad_inputs, uint32_t gridtype, int align_corners,
                 void* hip_stream) {
  (void)embeddings;  // upstream passes it but the kernels never read it (ge/:258-337)
  if (int rc = check_dims(B, D, C, L)) return rc;
  if (B == 0) return 0;
  if (!grad || !inputs || !offsets || !grad_embeddings) return fail(GCE_ERR_INVALID_ARGUMENT, "gce_backward: null tensor");
  if (calc_grad_inputs && (!dy_dx || !grad_inputs))
    return fail(GCE_ERR_INVALID_ARGUMENT, "gce_backward: calc_grad_inputs needs dy_dx and grad_inputs");
  LevelScales sc;
  gce_level_scales(L, S, H, sc.v);
  hipStream_t s = (hipStream_t)hip_stream;
  const dim3 grid((unsigned)(((uint64_t)B * C + 255) / 256), L, 1);
  {
    StageTimer t(s, ST_BWD_EMB);
    GCE_DISPATCH_DC(D, C, (k_grid_bwd<D, C><<<grid, 256, 0, s>>>(grad, inputs, offsets, grad_embeddings, B, L, sc, gridtype,
                                                               align_corners != 0)))
  }
  HIP_TRY(hipGetLastError(), "grid backward launch");
  if (calc_grad_inputs) {
    StageTimer t(s, ST_BWD_IN);
    GCE_DISPATCH_DC(D, C, (k_input_bwd<D, C><<<(B * D + 255) / 256, 256, 0, s>>>(grad, dy_dx, grad_inputs, B, L)))
    HIP_TRY(hipGetLastError(), "input backward launch");
  }
  return 0;
}


int gce_forward_t(int dtype, const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs, uint32_t B,
                  uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, void* dy_dx,
                  uint32_t gridtype, int align_corners, void* hip_stream) {
  if (dtype == GCE_F32)
    return gce_forward(inputs, (const float*)embeddings, offsets, (float*)outputs, B, D, C, L, S, H, calc_grad_inputs,
                       (float*)dy_dx, gridtype, align_corners, hip_stream);
  if (dtype != GCE_F16 && dtype != GCE_F64) return fail(GCE_ERR_UNSUPPORTED, "gce_forward_t: dtype must be GCE_F32, GCE_F16 or GCE_F64");
  if (int rc = check_dims(B, D, C, L)) return rc;
  if (B == 0) return 0;
  if (!inputs || !embeddings || !offsets || !outputs) return fail(GCE_ERR_INVALID_ARGUMENT, "gce_forward_t: null tensor");
  if (calc_grad_inputs && !dy_dx) return fail(GCE_ERR_INVALID_ARGUMENT, "gce_forward_t: calc_grad_inputs needs dy_dx");
  LevelScales sc;
  gce_level_scales(L, S, H, sc.v);
  hipStream_t s = (hipStream_t)hip_stream;
  const int rc = dtype == GCE_F16
                     ? forward_typed<_Float16>(inputs, embeddings, offsets, outputs, B, D, C, L, sc, calc_grad_inputs, dy_dx, gridtype, align_corners, s)
                     : forward_typed<double>(inputs, embeddings, offsets, outputs, B, D, C, L, sc, calc_grad_inputs, dy_dx, gridtype, align_corners, s);
  if (rc) return rc;
  HIP_TRY(hipGetLastError(), "grid forward launch");
  return 0;
}

int gce_backward_t(int dtype, const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                   void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                   int calc_grad_inputs, const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners,
                   void* hip_stream) {
  if (dtype == GCE_F32)
    return gce_backward((const float*)grad, inputs, (const float*)embeddings, offsets, (float*)grad_embeddings, B, D, C, L, S, H,
                        calc_grad_inputs, (const float*)dy_dx, (float*)grad_inputs, gridtype, align_corners, hip_stream);
  if (dtype != GCE_F16 && dtype != GCE_F64) return fail(GCE_ERR_UNSUPPORTED, "gce_backward_t: dtype must be GCE_F32, GCE_F16 or GCE_F64");
  if (int rc = check_dims(B, D, C, L)) return rc;
  if (B == 0) return 0;
  if (!grad || !inputs || !offsets || !grad_embeddings) return fail(GCE_ERR_INVALID_ARGUMENT, "gce_backward_t: null tensor");
  if (calc_grad_inputs && (!dy_dx || !grad_inputs))
    return fail(GCE_ERR_INVALID_ARGUMENT, "gce_backward_t: calc_grad_inputs needs dy_dx and grad_inputs");
  if (dtype == GCE_F16 && ((uintptr_t)grad_embeddings & 3u))
    return fail(GCE_ERR_INVALID_ARGUMENT, "gce_backward_t: half grad_embeddings must be 4-byte aligned (packed atomics)");
  LevelScales sc;
  gce_level_scales(L, S, H, sc.v);
  hipStream_t s = (hipStream_t)hip_stream;
  const int rc = dtype == GCE_F16
                     ? backward_typed<_Float16>(grad, inputs, offsets, grad_embeddings, B, D, C, L, sc, calc_grad_inputs, dy_dx, grad_inputs, gridtype, align_corners, s)
                     : backward_typed<double>(grad, inputs, offsets, grad_embeddings, B, D, C, L, sc, calc_grad_inputs, dy_dx, grad_inputs, gridtype, align_corners, s);
  if (rc) return rc;
  HIP_TRY(hipGetLastError(), "grid backward launch");
  return 0;
}

}  // extern "C"
