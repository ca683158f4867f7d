// gcv_points.hip -- MI355X (gfx950) kernels + C ABI (include/gcv.h) of the point-generation /
// visibility path: footprint extruder (K15), points -> volume (K14), ray/voxel traversal (K12).
// Reference behaviour: extensions/footprint_extruder/footprint_extruder.cpp ("fe/"),
// extensions/voxlib/{points_to_volume,ray_voxel_intersection}.cu, voxlib_common.h ("vox/").
//
// Everything here is integer / index work except K12's ray setup and crossing times, which follow
// the reference's fp32 expression order with IEEE division and sqrt (the library is built with
// -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt), so all outputs are bit-comparable
// with oracle/gcv_oracle.c.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/gcv.h"

namespace {

thread_local std::string g_err;
std::atomic<int> g_timing{0};
std::atomic<int> g_entry_jump{1};  // option "entry_jump": closed-form walk from the camera to the grid (A/B knob)

int fail(int code, const char* msg) {
  g_err = msg;
  return code;
}
int fail_hip(hipError_t e, const char* where) {
  g_err = std::string(where) + ": " + hipGetErrorString(e);
  return GCV_ERR_HIP;
}
#define HIP_TRY(expr, where)                          \
  do {                                                \
    hipError_t e_ = (expr);                           \
    if (e_ != hipSuccess) return fail_hip(e_, where); \
  } while (0)

// ---- stage timers (non-blocking; resolved lazily) ---------------------------------------------
enum Stage { ST_COUNT = 0, ST_EMIT, ST_CLEAR, ST_SCATTER, ST_OCC, ST_TRAVERSE, ST_N };
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

// =============================================================================================== K15
// fe/:143-213 is a serial triple loop on the host.  Here:
//   * whether a column is a border column does not depend on z (fe/:128-141: only the first test
//     does), so the number of points a pixel emits is closed-form: every k for a border column,
//     else the top k (z > td - scale) plus the bottom k (z == bu) when bottom points are included;
//   * count pass (one thread per pixel, block sums) -> one-block scan of the block sums -> emit pass
//     that recomputes the count, scans inside the block and writes the pixel's points at its global
//     offset: upstream's output order (row-major pixels, z ascending) without any sort.
constexpr int EX_BLOCK = 256;

struct ExtrudeArgs {
  int inc_btm, H, W;
  const int16_t* lut;
  gcv_seg_ins m;
  const int16_t *seg, *td, *bu;
  const uint8_t* pts;
};

__device__ __forceinline__ int ex_semantic(int ins, const gcv_seg_ins& m) {  // fe/:90-100
  if (ins < m.bldg_ins_min_id) return ins;
  if (ins >= m.car_ins_min_id) return m.car_semantic_id;
  return m.bldg_facade_semantic_id;
}

__device__ __forceinline__ bool ex_nbr_same(const int16_t* __restrict__ map, int x, int y, int W, int s) {  // fe/:102-126
  const int16_t c = map[(size_t)y * W + x];
  const int16_t* up = map + (size_t)(y - s) * W + x;
  const int16_t* mid = map + (size_t)y * W + x;
  const int16_t* dn = map + (size_t)(y + s) * W + x;
  return c == up[-s] && c == up[0] && c == up[s] && c == mid[-s] && c == mid[s] && c == dn[-s] && c == dn[0] &&
         c == dn[s];
}

// What pixel (i, j) emits: nk = number of k values in [bu, td] step scale; full = border column.
struct ExPixel {
  int count, nk, scale, sem, ins, bu;
  bool full;
};

__device__ __forceinline__ ExPixel ex_pixel(const ExtrudeArgs& a, int i, int j, unsigned long long* err) {
  ExPixel p;
  p.count = 0; p.nk = 0; p.scale = 1; p.sem = 0; p.ins = 0; p.bu = 0; p.full = false;
  const size_t idx = (size_t)i * a.W + j;
  if (!a.pts[idx]) return p;
  p.ins = a.seg[idx];
  p.sem = ex_semantic(p.ins, a.m);
  const int scale = p.sem >= 0 ? a.lut[p.sem] : 0;
  if (scale <= 0) {  // upstream: scale 0 from std::map::operator[] and an endless loop (fe/:186-189)
    if (err != nullptr) atomicMin(err, (unsigned long long)idx);
    return p;
  }
  p.scale = scale;
  const int td = a.td[idx], bu = a.bu[idx];
  p.bu = bu;
  if (bu > td) return p;
  p.nk = (td - bu) / scale + 1;
  p.full = j < scale || j >= a.W - scale - 1 || i < scale || i >= a.H - scale - 1 ||  // fe/:135-137
           !ex_nbr_same(a.seg, j, i, a.W, scale) || !ex_nbr_same(a.td, j, i, a.W, scale);
  p.count = p.full ? p.nk : (p.nk == 1 ? 1 : 1 + (a.inc_btm ? 1 : 0));
  return p;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}
// exclusive scan over a 256-thread block; *total = block sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* lds4, uint32_t* total) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t incl = wave_incl_scan(v, lane);
  if (lane == 63) lds4[w] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < EX_BLOCK / 64; k++) {
    const uint32_t s = lds4[k];
    if (k < w) base += s;
    tot += s;
  }
  *total = tot;
  __syncthreads();
  return base + incl - v;
}

// scratch layout: [0] u64 total, [1] u64 first bad pixel (~0 = none), then u64 block offsets
__global__ __launch_bounds__(EX_BLOCK) void k_extrude_count(const ExtrudeArgs a, unsigned long long* scratch) {
  __shared__ uint32_t lds4[4];
  const long long pix = (long long)blockIdx.x * EX_BLOCK + threadIdx.x;
  uint32_t c = 0;
  if (pix < (long long)a.H * a.W) c = (uint32_t)ex_pixel(a, (int)(pix / a.W), (int)(pix % a.W), scratch + 1).count;
  uint32_t total;
  block_excl_scan(c, lds4, &total);
  if (threadIdx.x == 0) scratch[2 + blockIdx.x] = total;
}

// in-place exclusive scan of the n block sums (one 1024-thread block; n = H*W/256 is small)
__global__ __launch_bounds__(1024) void k_extrude_scan(unsigned long long* scratch, int n) {
  __shared__ unsigned long long part[1024];
  unsigned long long* sums = scratch + 2;
  const int tid = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int beg = min(n, tid * per), end = min(n, beg + per);
  unsigned long long s = 0;
  for (int i = beg; i < end; i++) s += sums[i];
  part[tid] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const unsigned long long t = tid >= o ? part[tid - o] : 0ull;
    __syncthreads();
    part[tid] += t;
    __syncthreads();
  }
  unsigned long long run = part[tid] - s;
  for (int i = beg; i < end; i++) {
    const unsigned long long t = sums[i];
    sums[i] = run;
    run += t;
  }
  if (tid == 1023) scratch[0] = part[1023];
}

__global__ __launch_bounds__(EX_BLOCK) void k_extrude_emit(const ExtrudeArgs a, const unsigned long long* scratch,
                                                           int16_t* __restrict__ out, long long n_points) {
  __shared__ uint32_t lds4[4];
  const long long pix = (long long)blockIdx.x * EX_BLOCK + threadIdx.x;
  ExPixel p;
  p.count = 0;
  int i = 0, j = 0;
  if (pix < (long long)a.H * a.W) {
    i = (int)(pix / a.W);
    j = (int)(pix % a.W);
    p = ex_pixel(a, i, j, nullptr);
  }
  uint32_t total;
  const uint32_t local = block_excl_scan((uint32_t)p.count, lds4, &total);
  if (p.count == 0) return;
  long long o = (long long)scratch[2 + blockIdx.x] + local;
  if (o + p.count > n_points) return;  // caller passed a stale count; never write out of bounds
  const bool roof = p.sem == a.m.bldg_facade_semantic_id;
  const int16_t roof_ins = (int16_t)(p.ins + a.m.roof_ins_offset);  // fe/:199-203
  int16_t* w = out + 5 * o;
  if (p.full) {
    for (int m = 0; m < p.nk; m++, w += 5) {
      const bool top = m == p.nk - 1;
      w[0] = (int16_t)j; w[1] = (int16_t)i; w[2] = (int16_t)(p.bu + m * p.scale); w[3] = (int16_t)p.scale;
      w[4] = top && roof ? roof_ins : (int16_t)p.ins;
    }
  } else {
    if (p.count == 2) {  // bottom point
      w[0] = (int16_t)j; w[1] = (int16_t)i; w[2] = (int16_t)p.bu; w[3] = (int16_t)p.scale; w[4] = (int16_t)p.ins;
      w += 5;
    }
    w[0] = (int16_t)j; w[1] = (int16_t)i; w[2] = (int16_t)(p.bu + (p.nk - 1) * p.scale); w[3] = (int16_t)p.scale;
    w[4] = roof ? roof_ins : (int16_t)p.ins;
  }
}

// =============================================================================================== K13
// vox/maps_to_volume.cu:21-101: BEV maps straight into an int16 instance volume [H][W][depth] (one voxel per k
// of a border column).  Same closed-form border test as the extruder; thread per pixel.
__global__ __launch_bounds__(256) void k_maps_to_volume(int H, int W, int depth, const int8_t* __restrict__ scales,
                                                        int n_scales, const int16_t* __restrict__ inst_map,
                                                        const int16_t* __restrict__ td, const int16_t* __restrict__ bu,
                                                        const uint8_t* __restrict__ pts, int16_t* __restrict__ volume,
                                                        unsigned long long* __restrict__ err) {
  const long long px = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (px >= (long long)H * W) return;
  if (!pts[px]) return;
  const int j = (int)(px / W), i = (int)(px % W);
  const int hgt_up = td[px], hgt_lw = bu[px];
  const int inst = inst_map[px];
  const int sem = inst < 10 ? inst : 2;  // vox/maps_to_volume.cu:17-18,44
  if (sem < 0 || sem >= n_scales || scales[sem] <= 0) {
    atomicMin(err, (unsigned long long)px);
    return;
  }
  const int scale = scales[sem];
  bool border = i < scale || i >= W - scale - 1 || j < scale || j >= H - scale - 1;
  if (!border) border = !ex_nbr_same(td, i, j, W, scale) || !ex_nbr_same(inst_map, i, j, W, scale);
  int16_t* col = volume + px * depth;
  for (int k = hgt_lw; k <= hgt_up; k += scale) {
    const bool top = k > hgt_up - scale;
    if (!top && !border) continue;
    if (k < 0 || k >= depth) continue;  // upstream writes out of bounds here
    col[k] = (int16_t)((top && sem == 2) ? inst + 1 : inst);
  }
}

// =============================================================================================== K14
// vox/points_to_volume.cu:21-50.  One thread per point; atomicMax makes the overlap rule
// deterministic (highest id wins = sequential order).
//
// Occupancy = 1 bit per 16x16x16 macro cell, for the traversal's empty-space jumps.  Millions of
// points share a few thousand bitmask words and device-scope atomics to one address serialise at the
// memory side (a per-point atomicOr took 2.4 ms of a 2.9 ms scatter; testing the bit first does not
// help because the per-XCD L2s are not coherent).  Consecutive points are spatial neighbours, so each
// wave first merges the bits of lanes that target the same word and issues ONE atomicOr per distinct
// word; only cubes straddling a macro-cell face add individual atomics.
constexpr int MC_SHIFT = 4;  // macro cell = 16^3 voxels


__device__ __forceinline__ long long mc_linear(int kb, int jb, int lb, int wb, int db) {
  return ((long long)kb * wb + jb) * db + lb;
}

__device__ __forceinline__ void occ_set_wave(uint32_t* occ, bool valid, long long lin) {
  uint32_t word = valid ? (uint32_t)(lin >> 5) : 0xffffffffu;
  const uint32_t bit = valid ? 1u << (lin & 31) : 0u;
  while (true) {
    const unsigned long long todo = __ballot(word != 0xffffffffu);
    if (todo == 0ull) break;
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t w0 = (uint32_t)__shfl((int)word, leader, 64);
    const bool mine = word == w0;
    uint32_t bits = mine ? bit : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bits |= (uint32_t)__shfl_xor((int)bits, o, 64);
    if ((int)(threadIdx.x & 63) == leader) atomicOr(&occ[w0], bits);
    if (mine) word = 0xffffffffu;
  }
}

__global__ __launch_bounds__(256) void k_points_to_volume(long long n, int h, int w, int d,
                                                          const int16_t* __restrict__ points,
                                                          const int32_t* __restrict__ pt_ids,
                                                          const int16_t* __restrict__ scales,
                                                          int32_t* __restrict__ volume, uint32_t* __restrict__ occ) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool valid = idx < n;
  int pid = 0, x = 0, y = 0, z = 0, xe = 0, ye = 0, ze = 0;
  if (valid) {
    pid = pt_ids[idx];
    x = points[3 * idx]; y = points[3 * idx + 1]; z = points[3 * idx + 2];
    const int sx = scales[3 * idx], sy = scales[3 * idx + 1], sz = scales[3 * idx + 2];
    valid = !(x >= w || y >= h || z >= d || x < 0 || y < 0 || z < 0);
    xe = min(x + sx, w); ye = min(y + sy, h); ze = min(z + sz, d);
    valid = valid && xe > x && ye > y && ze > z;
  }
  if (valid) {
    for (int j = x; j < xe; ++j)
      for (int k = y; k < ye; ++k)
        for (int l = z; l < ze; ++l) atomicMax(&volume[((long long)k * w + j) * d + l], pid);
  }
  if (occ != nullptr) {  // wave-uniform
    const int wb = (w + 15) >> MC_SHIFT, db = (d + 15) >> MC_SHIFT;
    occ_set_wave(occ, valid, mc_linear(y >> MC_SHIFT, x >> MC_SHIFT, z >> MC_SHIFT, wb, db));
    if (valid && (((ye - 1) >> MC_SHIFT) != (y >> MC_SHIFT) || ((xe - 1) >> MC_SHIFT) != (x >> MC_SHIFT) ||
                  ((ze - 1) >> MC_SHIFT) != (z >> MC_SHIFT))) {  // cube straddles a macro-cell face
      for (int kb = y >> MC_SHIFT; kb <= (ye - 1) >> MC_SHIFT; kb++)
        for (int jb = x >> MC_SHIFT; jb <= (xe - 1) >> MC_SHIFT; jb++)
          for (int lb = z >> MC_SHIFT; lb <= (ze - 1) >> MC_SHIFT; lb++) {
            const long long lin = mc_linear(kb, jb, lb, wb, db);
            atomicOr(&occ[lin >> 5], 1u << (lin & 31));
          }
    }
  }
}

// Fused-pipeline variants (scripts/dataset_generator.py:1366-1388, _get_volume, without the host
// round trips): bounds of the extruded rows, then rows [n][5] = (x, y, z, scale, instance) straight
// into the volume with the localisation offsets applied in flight, id = row index + 1 and a cube of
// `scale` voxels per side (utils/helpers.get_point_scales with no special classes).
// Two stages, no atomics: six hot addresses would serialise every wave's result at the memory side
// (measured 1.1 ms for 17 M rows with per-wave atomicMin/Max; 0.05 ms like this).
constexpr int BOUNDS_BLOCKS = 1024;
__global__ __launch_bounds__(256) void k_rows_bounds(long long n, const int16_t* __restrict__ rows, int stride,
                                                     int* __restrict__ partial) {
  __shared__ int red[4][6];
  int mn[3] = {32767, 32767, 32767}, mx[3] = {-32768, -32768, -32768};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const int v = rows[i * stride + a];
      mn[a] = min(mn[a], v);
      mx[a] = max(mx[a], v);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[a] = min(mn[a], __shfl_xor(mn[a], o, 64));
      mx[a] = max(mx[a], __shfl_xor(mx[a], o, 64));
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      red[threadIdx.x >> 6][a] = mn[a];
      red[threadIdx.x >> 6][3 + a] = mx[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    int v = red[0][a];
    for (int w = 1; w < 4; w++) v = a < 3 ? min(v, red[w][a]) : max(v, red[w][a]);
    partial[6 * blockIdx.x + a] = v;
  }
}
// partial[6*b + a] for b < nb  ->  partial[a]  (one block)
__global__ __launch_bounds__(256) void k_bounds_final(int* __restrict__ partial, int nb) {
  __shared__ int red[256][6];
  int v[6] = {32767, 32767, 32767, -32768, -32768, -32768};
  for (int b = threadIdx.x; b < nb; b += 256)
    for (int a = 0; a < 6; a++) v[a] = a < 3 ? min(v[a], partial[6 * b + a]) : max(v[a], partial[6 * b + a]);
  for (int a = 0; a < 6; a++) red[threadIdx.x][a] = v[a];
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    int r = red[0][a];
    for (int t = 1; t < 256; t++) r = a < 3 ? min(r, red[t][a]) : max(r, red[t][a]);
    partial[a] = r;
  }
}

// ERASE: write zeros over the same cubes instead -- restores a resident volume to all-zero after the
// traversal (gcv_rows_erase_volume), which costs a pass over the ~32 M written voxels instead of a clear
// of the whole 5 GB box.
template <bool ERASE>
__global__ __launch_bounds__(256) void k_rows_to_volume(long long n, int h, int w, int d, const int16_t* __restrict__ rows,
                                                        int ox, int oy, int oz, int32_t* __restrict__ volume,
                                                        uint32_t* __restrict__ occ) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool valid = idx < n;
  int x = 0, y = 0, z = 0, xe = 0, ye = 0, ze = 0;
  if (valid) {
    const int16_t* r = rows + 5 * idx;
    // int16 arithmetic as the reference's tensor ops (points[:, k] -= offset wraps in int16)
    x = (int16_t)(r[0] - ox); y = (int16_t)(r[1] - oy); z = (int16_t)(r[2] - oz);
    const int s = r[3];
    valid = !(x >= w || y >= h || z >= d || x < 0 || y < 0 || z < 0);
    xe = min(x + s, w); ye = min(y + s, h); ze = min(z + s, d);
    valid = valid && xe > x && ye > y && ze > z;
  }
  if (valid) {
    const int pid = (int)(idx + 1);
    for (int j = x; j < xe; ++j)
      for (int k = y; k < ye; ++k)
        for (int l = z; l < ze; ++l) {
          if (ERASE)
            volume[((long long)k * w + j) * d + l] = 0;
          else
            atomicMax(&volume[((long long)k * w + j) * d + l], pid);
        }
  }
  if (!ERASE && occ != nullptr) {
    const int wb = (w + 15) >> MC_SHIFT, db = (d + 15) >> MC_SHIFT;
    occ_set_wave(occ, valid, mc_linear(y >> MC_SHIFT, x >> MC_SHIFT, z >> MC_SHIFT, wb, db));
    if (valid && (((ye - 1) >> MC_SHIFT) != (y >> MC_SHIFT) || ((xe - 1) >> MC_SHIFT) != (x >> MC_SHIFT) ||
                  ((ze - 1) >> MC_SHIFT) != (z >> MC_SHIFT))) {
      for (int kb = y >> MC_SHIFT; kb <= (ye - 1) >> MC_SHIFT; kb++)
        for (int jb = x >> MC_SHIFT; jb <= (xe - 1) >> MC_SHIFT; jb++)
          for (int lb = z >> MC_SHIFT; lb <= (ze - 1) >> MC_SHIFT; lb++) {
            const long long lin = mc_linear(kb, jb, lb, wb, db);
            atomicOr(&occ[lin >> 5], 1u << (lin & 31));
          }
    }
  }
}

// occupancy of an arbitrary dense volume: one thread per (k, j, 16-voxel run along d); the wave merges
// its bits (64 consecutive runs = at most a few words) before touching memory
__global__ __launch_bounds__(256) void k_build_occ(const int32_t* __restrict__ volume, int h, int w, int d,
                                                   uint32_t* __restrict__ occ) {
  const int db = (d + 15) >> MC_SHIFT, wb = (w + 15) >> MC_SHIFT;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool any = false;
  long long lin = 0;
  if (t < (long long)h * w * db) {
    const int lb = (int)(t % db);
    const long long kj = t / db;
    const int j = (int)(kj % w), k = (int)(kj / w);
    const int32_t* p = volume + ((long long)k * w + j) * d + 16 * lb;
    const int nz = min(16, d - 16 * lb);
    int32_t acc = 0;
    for (int l = 0; l < nz; l++) acc |= p[l];
    any = acc != 0;
    lin = mc_linear(k >> MC_SHIFT, j >> MC_SHIFT, lb, wb, db);
  }
  occ_set_wave(occ, any, lin);
}

// =============================================================================================== K12
// vox/ray_voxel_intersection.cu:54-216.  One wave = one 8x8 pixel tile (upstream's block shape, which
// is exactly a wave64).  Every crossing time is recomputed from the integer cell (upstream's
// expression, IEEE division), so the walk is a pure function of the cell sequence.
//
// MI355X change (HAS_OCC): upstream tests one voxel per step -- a dependent, cache-missing load per
// cell, ~700 per ray.  With the macro-cell bitmask a ray standing in an EMPTY 16^3 cell jumps to the
// cell it leaves through, in one move that reproduces the reference walk exactly:
//   * upstream's loop is a 3-way merge of the per-axis crossing sequences, ordered by (time, axis)
//     with its "<=" chain as the tie rule; crossing times along one axis never decrease;
//   * the exit event E is the first, in that order, of the three "leave the macro cell" crossings;
//   * every other axis has by then performed exactly the crossings that precede E in that order --
//     counted with the same fp32 expression (estimate from ori + T*dir, then corrected with the exact
//     predicate, so the count is the reference's whatever the estimate was);
//   * cells, times and the quit flag after the jump are what stepping would have left.
// Voxels are only read inside occupied macro cells.  Outputs are bit-identical with and without the
// bitmask (tests/test_points_gpu.py).
struct RvipArgs {
  int dims[3];
  long long strides[3];
  int max_samples;
  int img[2];
  float ori[3], fwd[3], side[3], up[3];
  float c[2], f;
  int wb, db;    // macro cells along w and d (occupancy)
  int contig32;  // contiguous [h][w][d] volume with fewer than 2^32 voxels
};

__device__ __forceinline__ void dev_normalize3(float* a) {  // vox/voxlib_common.h:56-68
  float len = 0.0f;
#pragma unroll
  for (int i = 0; i < 3; i++) len += a[i] * a[i];
  len = __builtin_sqrtf(len);
#pragma unroll
  for (int i = 0; i < 3; i++) a[i] /= len;
}

// time of the crossing through integer boundary N on axis AX: upstream's axis_t expression
#define GCV_TCROSS(AX, N) (((float)(N)-p.ori[AX]) / raydir[AX])

#define GCV_STEP(AX)                                  \
  {                                                   \
    tnow = axis_t[AX];                                \
    if (raydir[AX] > 0) {                             \
      axis_int[AX] += 1;                              \
      if (axis_int[AX] >= p.dims[AX]) quit = true;    \
      axis_t[AX] = GCV_TCROSS(AX, axis_int[AX] + 1);  \
    } else {                                          \
      axis_int[AX] -= 1;                              \
      if (axis_int[AX] < 0) quit = true;              \
      axis_t[AX] = GCV_TCROSS(AX, axis_int[AX]);      \
    }                                                 \
  }

// Cell of axis AX after the crossings it performs before event (TE, E).  Its boundaries are cur+1,
// cur+2, ... for a positive direction and cur, cur-1, ... for a negative one; crossing N precedes the
// event iff (t(N), AX) < (TE, E).  MAXN = crossings that keep the axis inside the macro cell.
#define GCV_PRED(AX, N, TE, E) ((GCV_TCROSS(AX, N) < (TE)) || (GCV_TCROSS(AX, N) == (TE) && (AX) < (E)))
#define GCV_COUNT(AX, TE, E, MAXN, OUT)                                              \
  {                                                                                  \
    int cnt = 0;                                                                     \
    const int maxn = (MAXN);                                                         \
    if (raydir[AX] > 0) {                                                            \
      const float pos = p.ori[AX] + (TE)*raydir[AX];                                 \
      cnt = (int)__builtin_floorf(pos) - axis_int[AX];                               \
      cnt = max(0, min(cnt, maxn));                                                  \
      while (cnt > 0 && !GCV_PRED(AX, axis_int[AX] + cnt, TE, E)) cnt--;             \
      while (cnt < maxn && GCV_PRED(AX, axis_int[AX] + cnt + 1, TE, E)) cnt++;       \
      OUT = axis_int[AX] + cnt;                                                      \
    } else if (raydir[AX] < 0) {                                                     \
      const float pos = p.ori[AX] + (TE)*raydir[AX];                                 \
      cnt = axis_int[AX] - (int)__builtin_ceilf(pos) + 1;                            \
      cnt = max(0, min(cnt, maxn));                                                  \
      while (cnt > 0 && !GCV_PRED(AX, axis_int[AX] - cnt + 1, TE, E)) cnt--;         \
      while (cnt < maxn && GCV_PRED(AX, axis_int[AX] - cnt, TE, E)) cnt++;           \
      OUT = axis_int[AX] - cnt;                                                      \
    } else {                                                                         \
      OUT = axis_int[AX];                                                            \
    }                                                                                \
  }

template <bool HAS_OCC, bool ENTRY_JUMP>
__global__ __launch_bounds__(64) void k_rvip(int32_t* __restrict__ out_voxel_id, float* __restrict__ out_depth,
                                             float* __restrict__ out_raydirs, const int32_t* __restrict__ in_voxel,
                                             const uint32_t* __restrict__ occ, const RvipArgs p) {
  const int col = blockIdx.x * 8 + (threadIdx.x & 7);
  const int row = blockIdx.y * 8 + (threadIdx.x >> 3);
  if (row >= p.img[0] || col >= p.img[1]) return;
  const long long pix = (long long)row * p.img[1] + col;
  const long long npix = (long long)p.img[0] * p.img[1];

  float raydir[3];
  const float n0 = p.c[0] - (float)row;  // flip height
  const float n1 = (float)col - p.c[1];
#pragma unroll
  for (int i = 0; i < 3; i++) raydir[i] = p.up[i] * n0 + p.side[i] * n1 + p.fwd[i] * p.f;
  dev_normalize3(raydir);
  out_raydirs[pix * 3] = raydir[0];
  out_raydirs[pix * 3 + 1] = raydir[1];
  out_raydirs[pix * 3 + 2] = raydir[2];

  float axis_t[3];
  int axis_int[3];
#pragma unroll
  for (int i = 0; i < 3; i++) axis_int[i] = (int)__builtin_floorf(p.ori[i]);
#pragma unroll
  for (int i = 0; i < 3; i++) {
    if (raydir[i] > 0)
      axis_t[i] = GCV_TCROSS(i, axis_int[i] + 1);
    else if (raydir[i] < 0)
      axis_t[i] = GCV_TCROSS(i, axis_int[i]);
    else
      axis_t[i] = HUGE_VALF;
  }
  const float qnan = __int_as_float(0x7fc00000);
  bool quit = false;
  long long mc_cached = -1;  // macro cell whose bit is in mc_bit
  bool mc_bit = true;

  // Entry jump.  A camera above the city starts outside the grid and upstream walks ~500 cells before the
  // first voxel test.  Nothing is tested on the way, so only the state at the moment the ray is inside for
  // the first time matters -- and, the walk being a (time, axis)-ordered merge of per-axis crossings, that
  // state is closed-form: the entering event E* is the LAST of the "enter the valid range" crossings of the
  // axes that start out of range; every other axis has then performed the crossings that precede E*
  // (GCV_COUNT, exact predicate); if one of them has thereby passed its far bound, or an out-of-range axis
  // moves away / does not move, upstream ends in `quit` without ever testing a voxel.
  bool pending_test = false;  // the cell reached by the entry jump is tested before any further step
  float tnow_entry = 0.0f;
  if (ENTRY_JUMP && !(((unsigned)axis_int[0] < (unsigned)p.dims[0]) & ((unsigned)axis_int[1] < (unsigned)p.dims[1]) &
                      ((unsigned)axis_int[2] < (unsigned)p.dims[2]))) {
    bool never = false;
    int e = -1;
    float TE = 0.0f;
    int enter_cell[3] = {axis_int[0], axis_int[1], axis_int[2]};
#pragma unroll
    for (int a = 0; a < 3; a++) {
      if ((unsigned)axis_int[a] < (unsigned)p.dims[a]) continue;
      int boundary;
      if (raydir[a] > 0 && axis_int[a] < 0) {
        boundary = 0;
        enter_cell[a] = 0;
      } else if (raydir[a] < 0 && axis_int[a] >= p.dims[a]) {
        boundary = p.dims[a];
        enter_cell[a] = p.dims[a] - 1;
      } else {
        never = true;
        boundary = 0;
      }
      const float Ta = GCV_TCROSS(a, boundary);
      if (e < 0 || Ta >= TE) {  // ascending a: on equal times the higher axis comes later in upstream's order
        TE = Ta;
        e = a;
      }
    }
    if (never) {
      quit = true;
    } else {
      constexpr int FAR = 1 << 30;
      int n0c = axis_int[0], n1c = axis_int[1], n2c = axis_int[2];
      if (e == 0) {
        GCV_COUNT(1, TE, 0, FAR, n1c)
        GCV_COUNT(2, TE, 0, FAR, n2c)
        n0c = enter_cell[0];
      } else if (e == 1) {
        GCV_COUNT(0, TE, 1, FAR, n0c)
        GCV_COUNT(2, TE, 1, FAR, n2c)
        n1c = enter_cell[1];
      } else {
        GCV_COUNT(0, TE, 2, FAR, n0c)
        GCV_COUNT(1, TE, 2, FAR, n1c)
        n2c = enter_cell[2];
      }
      axis_int[0] = n0c;
      axis_int[1] = n1c;
      axis_int[2] = n2c;
      // an axis that walked past its far bound (or has not reached its range: impossible before E*, kept as
      // a guard) means upstream hit `quit` on the way
      if (!(((unsigned)n0c < (unsigned)p.dims[0]) & ((unsigned)n1c < (unsigned)p.dims[1]) &
            ((unsigned)n2c < (unsigned)p.dims[2]))) {
        quit = true;
      } else {
#pragma unroll
        for (int a = 0; a < 3; a++) {
          if (raydir[a] > 0)
            axis_t[a] = GCV_TCROSS(a, axis_int[a] + 1);
          else if (raydir[a] < 0)
            axis_t[a] = GCV_TCROSS(a, axis_int[a]);
        }
        pending_test = true;
        tnow_entry = TE;
      }
    }
  }

  for (int plane = 0; plane < p.max_samples; plane++) {
    float t = qnan, t2 = qnan;
    int32_t blk_id = 0;
    while (!quit) {
      float tnow;
      bool jump = false;
      if (pending_test) {
        pending_test = false;
        tnow = tnow_entry;
        goto test_cell;
      }
      if (HAS_OCC && axis_int[0] >= 0 && axis_int[0] < p.dims[0] && axis_int[1] >= 0 && axis_int[1] < p.dims[1] &&
          axis_int[2] >= 0 && axis_int[2] < p.dims[2]) {
        const long long lin =
            mc_linear(axis_int[0] >> MC_SHIFT, axis_int[1] >> MC_SHIFT, axis_int[2] >> MC_SHIFT, p.wb, p.db);
        if (lin != mc_cached) {
          mc_cached = lin;
          mc_bit = (occ[lin >> 5] >> (lin & 31)) & 1u;
        }
        jump = !mc_bit;
      }
      if (jump) {
        // leave-the-macro-cell crossing of every axis (boundary coordinate and time)
        int lo[3], hi[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
          lo[a] = (axis_int[a] >> MC_SHIFT) << MC_SHIFT;
          hi[a] = min(lo[a] + 15, p.dims[a] - 1);
        }
        const float T0 = raydir[0] > 0 ? GCV_TCROSS(0, hi[0] + 1) : (raydir[0] < 0 ? GCV_TCROSS(0, lo[0]) : HUGE_VALF);
        const float T1 = raydir[1] > 0 ? GCV_TCROSS(1, hi[1] + 1) : (raydir[1] < 0 ? GCV_TCROSS(1, lo[1]) : HUGE_VALF);
        const float T2 = raydir[2] > 0 ? GCV_TCROSS(2, hi[2] + 1) : (raydir[2] < 0 ? GCV_TCROSS(2, lo[2]) : HUGE_VALF);
        int n0c = axis_int[0], n1c = axis_int[1], n2c = axis_int[2];
        if (T0 <= T1 && T0 <= T2) {  // same "<=" chain as the step selection
          tnow = T0;
          GCV_COUNT(1, tnow, 0, raydir[1] > 0 ? hi[1] - axis_int[1] : axis_int[1] - lo[1], n1c)
          GCV_COUNT(2, tnow, 0, raydir[2] > 0 ? hi[2] - axis_int[2] : axis_int[2] - lo[2], n2c)
          n0c = raydir[0] > 0 ? hi[0] + 1 : lo[0] - 1;
          if (n0c >= p.dims[0] || n0c < 0) quit = true;
        } else if (T1 <= T2) {
          tnow = T1;
          GCV_COUNT(0, tnow, 1, raydir[0] > 0 ? hi[0] - axis_int[0] : axis_int[0] - lo[0], n0c)
          GCV_COUNT(2, tnow, 1, raydir[2] > 0 ? hi[2] - axis_int[2] : axis_int[2] - lo[2], n2c)
          n1c = raydir[1] > 0 ? hi[1] + 1 : lo[1] - 1;
          if (n1c >= p.dims[1] || n1c < 0) quit = true;
        } else {
          tnow = T2;
          GCV_COUNT(0, tnow, 2, raydir[0] > 0 ? hi[0] - axis_int[0] : axis_int[0] - lo[0], n0c)
          GCV_COUNT(1, tnow, 2, raydir[1] > 0 ? hi[1] - axis_int[1] : axis_int[1] - lo[1], n1c)
          n2c = raydir[2] > 0 ? hi[2] + 1 : lo[2] - 1;
          if (n2c >= p.dims[2] || n2c < 0) quit = true;
        }
        axis_int[0] = n0c;
        axis_int[1] = n1c;
        axis_int[2] = n2c;
#pragma unroll
        for (int a = 0; a < 3; a++) {
          if (raydir[a] > 0)
            axis_t[a] = GCV_TCROSS(a, axis_int[a] + 1);
          else if (raydir[a] < 0)
            axis_t[a] = GCV_TCROSS(a, axis_int[a]);
        }
      } else {
        // One cell step, branch-free: the lanes of a wave step along different axes, and a 3-way branch
        // runs up to three bodies -- each with its own IEEE division -- per iteration (measured 1.69 ms
        // for the city workload).  Selecting the axis' operands first leaves ONE division for all lanes.
        const bool s0 = axis_t[0] <= axis_t[1] && axis_t[0] <= axis_t[2];  // upstream's "<=" chain
        const bool s1 = !s0 && axis_t[1] <= axis_t[2];
        tnow = s0 ? axis_t[0] : (s1 ? axis_t[1] : axis_t[2]);
        const float dir_a = s0 ? raydir[0] : (s1 ? raydir[1] : raydir[2]);
        const float ori_a = s0 ? p.ori[0] : (s1 ? p.ori[1] : p.ori[2]);
        const int dim_a = s0 ? p.dims[0] : (s1 ? p.dims[1] : p.dims[2]);
        int cell_a = s0 ? axis_int[0] : (s1 ? axis_int[1] : axis_int[2]);
        const bool fwd = dir_a > 0;
        cell_a += fwd ? 1 : -1;
        quit = fwd ? cell_a >= dim_a : cell_a < 0;  // (quit was false: the loop condition)
        const float t_a = ((float)(fwd ? cell_a + 1 : cell_a) - ori_a) / dir_a;  // GCV_TCROSS of the chosen axis
        axis_int[0] = s0 ? cell_a : axis_int[0];
        axis_int[1] = s1 ? cell_a : axis_int[1];
        axis_int[2] = (s0 || s1) ? axis_int[2] : cell_a;
        axis_t[0] = s0 ? t_a : axis_t[0];
        axis_t[1] = s1 ? t_a : axis_t[1];
        axis_t[2] = (s0 || s1) ? axis_t[2] : t_a;
      }
      if (quit) break;
    test_cell:
      // one unsigned compare per axis, combined without short-circuit branches (the scalar unit is the
      // scarce resource in this loop, as in the blend kernels)
      const bool in_grid = ((unsigned)axis_int[0] < (unsigned)p.dims[0]) & ((unsigned)axis_int[1] < (unsigned)p.dims[1]) &
                           ((unsigned)axis_int[2] < (unsigned)p.dims[2]);
      if (!in_grid) continue;  // still outside the grid
      if (HAS_OCC) {
        const long long lin =
            mc_linear(axis_int[0] >> MC_SHIFT, axis_int[1] >> MC_SHIFT, axis_int[2] >> MC_SHIFT, p.wb, p.db);
        if (lin != mc_cached) {
          mc_cached = lin;
          mc_bit = (occ[lin >> 5] >> (lin & 31)) & 1u;
        }
        if (!mc_bit) continue;  // empty macro cell: the voxel is 0 without reading it
      }
      if (p.contig32)  // wave-uniform: contiguous [h][w][d] with < 2^32 voxels -> 32-bit index arithmetic
        blk_id = in_voxel[((uint32_t)axis_int[0] * (uint32_t)p.dims[1] + (uint32_t)axis_int[1]) * (uint32_t)p.dims[2] +
                          (uint32_t)axis_int[2]];
      else
        blk_id = in_voxel[(long long)axis_int[0] * p.strides[0] + (long long)axis_int[1] * p.strides[1] +
                          (long long)axis_int[2] * p.strides[2]];
      if (blk_id == 0) continue;
      t = tnow;
      if (axis_t[0] <= axis_t[1] && axis_t[0] <= axis_t[2])
        t2 = axis_t[0];
      else if (axis_t[1] <= axis_t[2])
        t2 = axis_t[1];
      else
        t2 = axis_t[2];
      break;
    }
    out_depth[pix * p.max_samples + plane] = t;
    out_depth[npix * p.max_samples + pix * p.max_samples + plane] = t2;
    out_voxel_id[pix * p.max_samples + plane] = blk_id;
  }
}

// host half of vox/ray_voxel_intersection.cu:256-266 (same float operations, -ffp-contract=off)
void host_normalize3(float* a) {
  float len = 0.0f;
  for (int i = 0; i < 3; i++) len += a[i] * a[i];
  len = sqrtf(len);
  for (int i = 0; i < 3; i++) a[i] /= len;
}
void host_cross3(float* r, const float* a, const float* b) {
  r[0] = a[1] * b[2] - a[2] * b[1];
  r[1] = a[2] * b[0] - a[0] * b[2];
  r[2] = a[0] * b[1] - a[1] * b[0];
}

int make_extrude_args(int inc_btm, const int16_t* lut, const gcv_seg_ins* m, int H, int W, const int16_t* seg,
                      const int16_t* td, const int16_t* bu, const uint8_t* pts, size_t scratch_bytes, ExtrudeArgs* a) {
  if (!lut || !m || !seg || !td || !bu || !pts) return fail(GCV_ERR_INVALID_ARGUMENT, "null map / table pointer");
  if (H <= 0 || W <= 0 || H > 32767 || W > 32767)
    return fail(GCV_ERR_INVALID_ARGUMENT, "map size must be in [1, 32767] (coordinates are int16, fe/:163-166)");
  if (scratch_bytes < gcv_extrude_scratch_bytes(H, W)) return fail(GCV_ERR_BUFFER_TOO_SMALL, "extrude scratch too small");
  a->inc_btm = inc_btm ? 1 : 0;
  a->H = H; a->W = W; a->lut = lut; a->m = *m; a->seg = seg; a->td = td; a->bu = bu; a->pts = pts;
  return 0;
}

}  // namespace

extern "C" {

int gcv_abi_version(void) { return GCV_ABI_VERSION; }
const char* gcv_last_error(void) { return g_err.c_str(); }

int gcv_set_option(const char* name, int value) {
  if (!name) return -1;
  if (!strcmp(name, "timing")) return g_timing.exchange(value);
  if (!strcmp(name, "entry_jump")) return g_entry_jump.exchange(value);
  return -1;
}

int gcv_get_stage_ms(float* out, int n) {
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

size_t gcv_extrude_scratch_bytes(int32_t height, int32_t width) {
  const size_t npix = (size_t)(height > 0 ? height : 0) * (size_t)(width > 0 ? width : 0);
  return sizeof(unsigned long long) * (2 + (npix + EX_BLOCK - 1) / EX_BLOCK + 1);
}

int gcv_extrude_count(int32_t inc_btm, const int16_t* lut, const gcv_seg_ins* m, int32_t H, int32_t W,
                      const int16_t* seg, const int16_t* td, const int16_t* bu, const uint8_t* pts, void* scratch,
                      size_t scratch_bytes, int64_t* n_points_host, void* hip_stream) {
  if (!n_points_host || !scratch) return fail(GCV_ERR_INVALID_ARGUMENT, "null scratch / n_points_host");
  ExtrudeArgs a;
  if (int rc = make_extrude_args(inc_btm, lut, m, H, W, seg, td, bu, pts, scratch_bytes, &a)) return rc;
  hipStream_t s = (hipStream_t)hip_stream;
  unsigned long long* sc = (unsigned long long*)scratch;
  const int nblocks = (int)(((size_t)H * W + EX_BLOCK - 1) / EX_BLOCK);
  const unsigned long long init[2] = {0ull, ~0ull};
  HIP_TRY(hipMemcpyAsync(sc, init, sizeof(init), hipMemcpyHostToDevice, s), "extrude scratch init");
  {
    StageTimer t(s, ST_COUNT);
    k_extrude_count<<<nblocks, EX_BLOCK, 0, s>>>(a, sc);
    k_extrude_scan<<<1, 1024, 0, s>>>(sc, nblocks);
  }
  HIP_TRY(hipGetLastError(), "extrude count launch");
  unsigned long long head[2];
  HIP_TRY(hipMemcpyAsync(head, sc, sizeof(head), hipMemcpyDeviceToHost, s), "extrude count read-back");
  HIP_TRY(hipStreamSynchronize(s), "extrude count sync");
  if (head[1] != ~0ull) {
    char msg[160];
    snprintf(msg, sizeof(msg), "pixel %llu: semantic id without a positive scale (upstream would not terminate)", head[1]);
    return fail(GCV_ERR_UNKNOWN_CLASS, msg);
  }
  *n_points_host = (int64_t)head[0];
  return 0;
}

int gcv_extrude_emit(int32_t inc_btm, const int16_t* lut, const gcv_seg_ins* m, int32_t H, int32_t W,
                     const int16_t* seg, const int16_t* td, const int16_t* bu, const uint8_t* pts, const void* scratch,
                     size_t scratch_bytes, int16_t* points_out, int64_t n_points, void* hip_stream) {
  if (!scratch) return fail(GCV_ERR_INVALID_ARGUMENT, "null scratch");
  if (n_points < 0) return fail(GCV_ERR_INVALID_ARGUMENT, "n_points < 0");
  if (n_points == 0) return 0;
  if (!points_out) return fail(GCV_ERR_INVALID_ARGUMENT, "null points_out");
  ExtrudeArgs a;
  if (int rc = make_extrude_args(inc_btm, lut, m, H, W, seg, td, bu, pts, scratch_bytes, &a)) return rc;
  hipStream_t s = (hipStream_t)hip_stream;
  const int nblocks = (int)(((size_t)H * W + EX_BLOCK - 1) / EX_BLOCK);
  {
    StageTimer t(s, ST_EMIT);
    k_extrude_emit<<<nblocks, EX_BLOCK, 0, s>>>(a, (const unsigned long long*)scratch, points_out, (long long)n_points);
  }
  HIP_TRY(hipGetLastError(), "extrude emit launch");
  return 0;
}

int gcv_maps_to_volume(const int16_t* inst_map, const int16_t* td_hf, const int16_t* bu_hf, const uint8_t* pts_map,
                       const int8_t* scales, int32_t n_scales, int32_t height, int32_t width, int32_t depth,
                       int16_t* volume, void* scratch8, void* hip_stream) {
  if (!inst_map || !td_hf || !bu_hf || !pts_map || !scales || !volume || !scratch8)
    return fail(GCV_ERR_INVALID_ARGUMENT, "gcv_maps_to_volume: null argument");
  if (height <= 0 || width <= 0 || depth <= 0 || n_scales <= 0)
    return fail(GCV_ERR_INVALID_ARGUMENT, "gcv_maps_to_volume: sizes must be positive");
  hipStream_t s = (hipStream_t)hip_stream;
  const unsigned long long none = ~0ull;
  HIP_TRY(hipMemcpyAsync(scratch8, &none, 8, hipMemcpyHostToDevice, s), "maps_to_volume scratch init");
  HIP_TRY(hipMemsetAsync(volume, 0, sizeof(int16_t) * (size_t)height * width * depth, s), "maps_to_volume clear");
  const long long npx = (long long)height * width;
  k_maps_to_volume<<<(unsigned)((npx + 255) / 256), 256, 0, s>>>(height, width, depth, scales, n_scales, inst_map, td_hf,
                                                                bu_hf, pts_map, volume, (unsigned long long*)scratch8);
  HIP_TRY(hipGetLastError(), "maps_to_volume launch");
  unsigned long long bad = none;
  HIP_TRY(hipMemcpyAsync(&bad, scratch8, 8, hipMemcpyDeviceToHost, s), "maps_to_volume read-back");
  HIP_TRY(hipStreamSynchronize(s), "maps_to_volume sync");
  if (bad != none) {
    char msg[160];
    snprintf(msg, sizeof(msg), "pixel %llu: class without a positive scale (upstream would not terminate)", bad);
    return fail(GCV_ERR_UNKNOWN_CLASS, msg);
  }
  return 0;
}

size_t gcv_occupancy_bytes(int32_t h, int32_t w, int32_t d) {
  if (h <= 0 || w <= 0 || d <= 0) return 0;
  const size_t cells = (size_t)((h + 15) >> 4) * (size_t)((w + 15) >> 4) * (size_t)((d + 15) >> 4);
  return 4 * ((cells + 31) / 32);
}

int gcv_points_to_volume(int64_t n, const int16_t* points, const int32_t* pt_ids, const int16_t* scales, int32_t h,
                         int32_t w, int32_t d, int32_t* volume, uint32_t* occupancy, void* hip_stream) {
  if (h <= 0 || w <= 0 || d <= 0) return fail(GCV_ERR_INVALID_ARGUMENT, "volume dimensions must be positive");
  if (!volume) return fail(GCV_ERR_INVALID_ARGUMENT, "null volume");
  if (n < 0 || (n > 0 && (!points || !pt_ids || !scales))) return fail(GCV_ERR_INVALID_ARGUMENT, "null point arrays");
  hipStream_t s = (hipStream_t)hip_stream;
  {
    StageTimer t(s, ST_CLEAR);
    HIP_TRY(hipMemsetAsync(volume, 0, sizeof(int32_t) * (size_t)h * w * d, s), "volume clear");
    if (occupancy) HIP_TRY(hipMemsetAsync(occupancy, 0, gcv_occupancy_bytes(h, w, d), s), "occupancy clear");
  }
  if (n > 0) {
    StageTimer t(s, ST_SCATTER);
    k_points_to_volume<<<(unsigned)((n + 255) / 256), 256, 0, s>>>((long long)n, h, w, d, points, pt_ids, scales, volume,
                                                                  occupancy);
    HIP_TRY(hipGetLastError(), "points_to_volume launch");
  }
  return 0;
}

size_t gcv_bounds_scratch_bytes(void) { return sizeof(int) * 6 * BOUNDS_BLOCKS; }

int gcv_points_bounds(int64_t n, const int16_t* rows, int32_t row_stride, void* scratch, int32_t min_host[3],
                      int32_t max_host[3], void* hip_stream) {
  if (n <= 0 || !rows || !scratch || !min_host || !max_host || row_stride < 3)
    return fail(GCV_ERR_INVALID_ARGUMENT, "gcv_points_bounds: need n > 0, rows, scratch, outputs, stride >= 3");
  hipStream_t s = (hipStream_t)hip_stream;
  const int blocks = (int)std::min<long long>((n + 255) / 256, BOUNDS_BLOCKS);
  k_rows_bounds<<<blocks, 256, 0, s>>>((long long)n, rows, row_stride, (int*)scratch);
  k_bounds_final<<<1, 256, 0, s>>>((int*)scratch, blocks);
  HIP_TRY(hipGetLastError(), "bounds launch");
  int out[6];
  HIP_TRY(hipMemcpyAsync(out, scratch, sizeof(out), hipMemcpyDeviceToHost, s), "bounds read-back");
  HIP_TRY(hipStreamSynchronize(s), "bounds sync");
  for (int a = 0; a < 3; a++) {
    min_host[a] = out[a];
    max_host[a] = out[3 + a];
  }
  return 0;
}

int gcv_rows_erase_volume(int64_t n, const int16_t* rows, const int32_t offset[3], int32_t h, int32_t w, int32_t d,
                          int32_t* volume, void* hip_stream) {
  if (h <= 0 || w <= 0 || d <= 0) return fail(GCV_ERR_INVALID_ARGUMENT, "volume dimensions must be positive");
  if (!volume || !offset) return fail(GCV_ERR_INVALID_ARGUMENT, "null volume / offset");
  if (n < 0 || (n > 0 && !rows)) return fail(GCV_ERR_INVALID_ARGUMENT, "null rows");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)hip_stream;
  StageTimer t(s, ST_CLEAR);
  k_rows_to_volume<true><<<(unsigned)((n + 255) / 256), 256, 0, s>>>((long long)n, h, w, d, rows, offset[0], offset[1],
                                                                    offset[2], volume, nullptr);
  HIP_TRY(hipGetLastError(), "rows_erase_volume launch");
  return 0;
}

int gcv_rows_to_volume(int64_t n, const int16_t* rows, const int32_t offset[3], int32_t h, int32_t w, int32_t d,
                       int32_t* volume, uint32_t* occupancy, int32_t volume_is_zero, void* hip_stream) {
  if (h <= 0 || w <= 0 || d <= 0) return fail(GCV_ERR_INVALID_ARGUMENT, "volume dimensions must be positive");
  if (!volume || !offset) return fail(GCV_ERR_INVALID_ARGUMENT, "null volume / offset");
  if (n < 0 || (n > 0 && !rows)) return fail(GCV_ERR_INVALID_ARGUMENT, "null rows");
  if (n >= 2147483647ll) return fail(GCV_ERR_INVALID_ARGUMENT, "more than 2^31-2 points (ids are int32, dataset_generator.py:1381)");
  hipStream_t s = (hipStream_t)hip_stream;
  if (!volume_is_zero) {
    StageTimer t(s, ST_CLEAR);
    HIP_TRY(hipMemsetAsync(volume, 0, sizeof(int32_t) * (size_t)h * w * d, s), "volume clear");
  }
  if (occupancy) HIP_TRY(hipMemsetAsync(occupancy, 0, gcv_occupancy_bytes(h, w, d), s), "occupancy clear");
  if (n > 0) {
    StageTimer t(s, ST_SCATTER);
    k_rows_to_volume<false><<<(unsigned)((n + 255) / 256), 256, 0, s>>>((long long)n, h, w, d, rows, offset[0], offset[1],
                                                                offset[2], volume, occupancy);
    HIP_TRY(hipGetLastError(), "rows_to_volume launch");
  }
  return 0;
}

int gcv_build_occupancy(const int32_t* volume, int32_t h, int32_t w, int32_t d, uint32_t* occupancy, void* hip_stream) {
  if (h <= 0 || w <= 0 || d <= 0) return fail(GCV_ERR_INVALID_ARGUMENT, "volume dimensions must be positive");
  if (!volume || !occupancy) return fail(GCV_ERR_INVALID_ARGUMENT, "null volume / occupancy");
  hipStream_t s = (hipStream_t)hip_stream;
  StageTimer t(s, ST_OCC);
  HIP_TRY(hipMemsetAsync(occupancy, 0, gcv_occupancy_bytes(h, w, d), s), "occupancy clear");
  const long long threads = (long long)h * w * ((d + 15) >> 4);
  k_build_occ<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(volume, h, w, d, occupancy);
  HIP_TRY(hipGetLastError(), "build occupancy launch");
  return 0;
}

int gcv_ray_voxel_intersection(const int32_t* volume, const int32_t dims[3], const int64_t strides[3],
                               const uint32_t* occupancy, const float cam_ori[3], const float cam_dir[3],
                               const float cam_up[3], float cam_f, const float cam_c[2], const int32_t img_dims[2],
                               int32_t max_samples, int32_t* out_voxel_id, float* out_depth, float* out_raydirs,
                               void* hip_stream) {
  if (!volume || !dims || !strides || !cam_ori || !cam_dir || !cam_up || !cam_c || !img_dims)
    return fail(GCV_ERR_INVALID_ARGUMENT, "null argument");
  if (!out_voxel_id || !out_depth || !out_raydirs) return fail(GCV_ERR_INVALID_ARGUMENT, "null output");
  if (dims[0] <= 0 || dims[1] <= 0 || dims[2] <= 0) return fail(GCV_ERR_INVALID_ARGUMENT, "volume dims must be positive");
  if (img_dims[0] <= 0 || img_dims[1] <= 0 || max_samples <= 0)
    return fail(GCV_ERR_INVALID_ARGUMENT, "image dims / max_samples must be positive");
  if (occupancy && !(strides[2] == 1 && strides[1] == dims[2] && strides[0] == (int64_t)dims[1] * dims[2]))
    return fail(GCV_ERR_INVALID_ARGUMENT, "occupancy requires a contiguous [h][w][d] volume");
  RvipArgs p;
  for (int i = 0; i < 3; i++) {
    p.dims[i] = dims[i];
    p.strides[i] = strides[i];
    p.ori[i] = cam_ori[i];
    p.fwd[i] = cam_dir[i];
  }
  host_normalize3(p.fwd);
  host_cross3(p.side, p.fwd, cam_up);
  host_normalize3(p.side);
  host_cross3(p.up, p.side, p.fwd);
  host_normalize3(p.up);
  p.f = cam_f;
  p.c[0] = cam_c[0]; p.c[1] = cam_c[1];
  p.max_samples = max_samples;
  p.img[0] = img_dims[0]; p.img[1] = img_dims[1];
  p.wb = (dims[1] + 15) >> 4;
  p.db = (dims[2] + 15) >> 4;
  p.contig32 = strides[2] == 1 && strides[1] == dims[2] && strides[0] == (int64_t)dims[1] * dims[2] &&
               (uint64_t)dims[0] * (uint64_t)dims[1] * (uint64_t)dims[2] < (1ull << 32);
  hipStream_t s = (hipStream_t)hip_stream;
  const dim3 grid((img_dims[1] + 7) / 8, (img_dims[0] + 7) / 8, 1);
  {
    StageTimer t(s, ST_TRAVERSE);
    const bool entry = g_entry_jump.load() != 0;
    if (occupancy && entry)
      k_rvip<true, true><<<grid, 64, 0, s>>>(out_voxel_id, out_depth, out_raydirs, volume, occupancy, p);
    else if (occupancy)
      k_rvip<true, false><<<grid, 64, 0, s>>>(out_voxel_id, out_depth, out_raydirs, volume, occupancy, p);
    else if (entry)
      k_rvip<false, true><<<grid, 64, 0, s>>>(out_voxel_id, out_depth, out_raydirs, volume, nullptr, p);
    else
      k_rvip<false, false><<<grid, 64, 0, s>>>(out_voxel_id, out_depth, out_raydirs, volume, nullptr, p);
  }
  HIP_TRY(hipGetLastError(), "ray_voxel_intersection launch");
  return 0;
}

}  // extern "C"
