// gcr_binning.hip -- tile binning for gfx950: K3 key emit, K4 stable LSD radix sort, K5 ranges.
// Integer / byte work, HBM-bound; no MFMA.  wave64 ballots do the intra-wave digit matching.
#include "gcr_device.h"
#include "gcr_internal.h"

namespace {

constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 16;  // keys per thread
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;
constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;

// ------------------------------------------------------------------------------------- K3
// cr/rasterizer_impl.cu:66-99 (duplicateWithKeys).  The per-Gaussian offset
// (offsets[idx-1] upstream) is rebuilt as block_offsets[block] + block-local exclusive scan,
// which is the same number.  key = tile << 32 | float_bits(depth), value = Gaussian index,
// emitted in row-major tile order.
__global__ __launch_bounds__(256) void k_emit(int P, const uint32_t* __restrict__ tiles_touched,
                                              const uint32_t* __restrict__ block_offsets,
                                              const float4* __restrict__ rec, int gx,
                                              uint64_t* __restrict__ keys,
                                              uint32_t* __restrict__ vals) {
  __shared__ uint32_t lds4[4];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const uint32_t t = idx < P ? tiles_touched[idx] : 0u;
  uint32_t total;
  uint32_t off = gcr_block_excl_scan_256(t, lds4, &total);
  if (t == 0) return;
  off += block_offsets[blockIdx.x];
  const float4 q2 = rec[(size_t)idx * GCR_REC_QUADS + 2];
  const uint32_t dbits = __float_as_uint(q2.y);
  const uint32_t rx = __float_as_uint(q2.z), ry = __float_as_uint(q2.w);
  const uint32_t minx = rx & 0xffffu, maxx = rx >> 16, miny = ry & 0xffffu, maxy = ry >> 16;
  for (uint32_t y = miny; y < maxy; y++)
    for (uint32_t x = minx; x < maxx; x++) {
      const uint64_t key = ((uint64_t)(y * (uint32_t)gx + x) << 32) | dbits;
      keys[off] = key;
      vals[off] = (uint32_t)idx;
      off++;
    }
}

// ------------------------------------------------------------------------------------- K4
// Stable least-significant-digit radix sort, 8-bit digits (replaces
// cub::DeviceRadixSort::SortPairs at cr/rasterizer_impl.cu:255-260; stability is what makes
// equal-depth ties resolve in ascending Gaussian index, which the blend order depends on).
// Per pass: (a) per-block digit histogram -> table[digit][block] and global digit totals,
// (b) 256 independent exclusive scans along blocks, (c) stable scatter.

__global__ __launch_bounds__(SORT_THREADS) void k_radix_hist(const uint64_t* __restrict__ keys,
                                                             int64_t R, int shift, int nb,
                                                             uint32_t* __restrict__ table,
                                                             uint32_t* __restrict__ ghist) {
  __shared__ uint32_t h[RADIX];
  const int tid = threadIdx.x;
  h[tid] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
#pragma unroll 4
  for (int i = 0; i < SORT_ITEMS; i++) {
    const int64_t idx = base + (int64_t)i * SORT_THREADS + tid;
    if (idx < R) atomicAdd(&h[(uint32_t)(keys[idx] >> shift) & (RADIX - 1)], 1u);
  }
  __syncthreads();
  const uint32_t c = h[tid];
  table[(size_t)tid * nb + blockIdx.x] = c;
  if (c) atomicAdd(&ghist[tid], c);
}

// one block per digit: exclusive scan of table[digit][0..nb) in place
__global__ __launch_bounds__(256) void k_radix_scan(uint32_t* __restrict__ table, int nb) {
  __shared__ uint32_t lds4[4];
  uint32_t* row = table + (size_t)blockIdx.x * nb;
  uint32_t running = 0;
  for (int base = 0; base < nb; base += 256) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < nb ? row[i] : 0u;
    uint32_t total;
    const uint32_t ex = gcr_block_excl_scan_256(v, lds4, &total);
    if (i < nb) row[i] = running + ex;
    running += total;
  }
}

__global__ __launch_bounds__(SORT_THREADS) void k_radix_scatter(
    const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin, uint64_t* __restrict__ kout,
    uint32_t* __restrict__ vout, int64_t R, int shift, int nb, const uint32_t* __restrict__ table,
    const uint32_t* __restrict__ ghist) {
  __shared__ uint32_t cnt[4][RADIX];  // per-wave running digit counts
  __shared__ uint32_t dbase[RADIX];   // global output base of (digit, this block)
  __shared__ uint32_t lds4[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#pragma unroll
  for (int i = 0; i < 4; i++) cnt[i][tid] = 0;
  {
    uint32_t total;
    const uint32_t gb = gcr_block_excl_scan_256(ghist[tid], lds4, &total);
    dbase[tid] = gb + table[(size_t)tid * nb + blockIdx.x];
  }
  __syncthreads();

  // Wave w owns the contiguous slice [base + w*1024, +1024); element (i, lane) is
  // base + w*1024 + i*64 + lane, so (wave, i, lane) order == input order (stability).
  const int64_t wbase = (int64_t)blockIdx.x * SORT_TILE + (int64_t)w * (64 * SORT_ITEMS);
  uint64_t key[SORT_ITEMS];
  uint32_t lrank[SORT_ITEMS];
  const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int i = 0; i < SORT_ITEMS; i++) {
    const int64_t idx = wbase + i * 64 + lane;
    const bool valid = idx < R;
    key[i] = valid ? kin[idx] : ~0ull;
    const uint32_t d = (uint32_t)(key[i] >> shift) & (RADIX - 1);
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RADIX_BITS; b++) {
      const bool bit = (d >> b) & 1u;
      const uint64_t bb = __ballot(bit);
      m &= bit ? bb : ~bb;
    }
    const uint32_t rank_in = (uint32_t)__popcll(m & lt_mask);
    const uint32_t old = cnt[w][d];
    __builtin_amdgcn_wave_barrier();
    if (valid && rank_in == 0) cnt[w][d] = old + (uint32_t)__popcll(m);
    __builtin_amdgcn_wave_barrier();
    lrank[i] = old + rank_in;
  }
  __syncthreads();
  {  // exclusive prefix of the per-wave counts across the four waves, per digit
    uint32_t run = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t t = cnt[i][tid];
      cnt[i][tid] = run;
      run += t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < SORT_ITEMS; i++) {
    const int64_t idx = wbase + i * 64 + lane;
    if (idx < R) {
      const uint32_t d = (uint32_t)(key[i] >> shift) & (RADIX - 1);
      const uint32_t pos = dbase[d] + cnt[w][d] + lrank[i];
      kout[pos] = key[i];
      vout[pos] = vin[idx];
    }
  }
}

// ------------------------------------------------------------------------------------- K5
// cr/rasterizer_impl.cu:104-124 (identifyTileRanges); ranges must be zeroed beforehand
// (:262-264) so untouched tiles stay (0,0).
__global__ __launch_bounds__(256) void k_tile_ranges(const uint64_t* __restrict__ keys, int64_t R,
                                                     uint32_t* __restrict__ ranges) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= R) return;
  const uint32_t currtile = (uint32_t)(keys[idx] >> 32);
  if (idx == 0)
    ranges[2 * currtile + 0] = 0;
  else {
    const uint32_t prevtile = (uint32_t)(keys[idx - 1] >> 32);
    if (currtile != prevtile) {
      ranges[2 * prevtile + 1] = (uint32_t)idx;
      ranges[2 * currtile + 0] = (uint32_t)idx;
    }
  }
  if (idx == R - 1) ranges[2 * currtile + 1] = (uint32_t)R;
}

}  // namespace

hipError_t gcr_launch_emit(int P, const uint32_t* tiles_touched, const uint32_t* block_offsets,
                           const float4* rec, int gx, uint64_t* keys, uint32_t* vals,
                           hipStream_t s) {
  if (P <= 0) return hipSuccess;
  k_emit<<<(P + 255) / 256, 256, 0, s>>>(P, tiles_touched, block_offsets, rec, gx, keys, vals);
  return hipGetLastError();
}

int gcr_sort_passes(int end_bit) { return (end_bit + RADIX_BITS - 1) / RADIX_BITS; }

static inline int sort_num_blocks(int64_t R) { return (int)((R + SORT_TILE - 1) / SORT_TILE); }

size_t gcr_sort_hist_bytes(int64_t R, int end_bit) {
  const size_t nb = (size_t)(R > 0 ? sort_num_blocks(R) : 1);
  // table[RADIX][nb] + one global histogram per pass
  return sizeof(uint32_t) * (RADIX * nb + (size_t)RADIX * gcr_sort_passes(end_bit));
}

hipError_t gcr_launch_sort(uint64_t* k0, uint32_t* v0, uint64_t* k1, uint32_t* v1, int64_t R,
                           int end_bit, uint32_t* hist, int* sorted_half, hipStream_t s) {
  const int passes = gcr_sort_passes(end_bit);
  *sorted_half = passes & 1;
  if (R <= 0) return hipSuccess;
  const int nb = sort_num_blocks(R);
  uint32_t* table = hist;
  uint32_t* ghist = hist + (size_t)RADIX * nb;
  hipError_t e = hipMemsetAsync(ghist, 0, sizeof(uint32_t) * RADIX * passes, s);
  if (e != hipSuccess) return e;
  uint64_t* kin = k0;
  uint32_t* vin = v0;
  uint64_t* kout = k1;
  uint32_t* vout = v1;
  for (int p = 0; p < passes; p++) {
    const int shift = p * RADIX_BITS;
    uint32_t* gh = ghist + (size_t)p * RADIX;
    k_radix_hist<<<nb, SORT_THREADS, 0, s>>>(kin, R, shift, nb, table, gh);
    k_radix_scan<<<RADIX, 256, 0, s>>>(table, nb);
    k_radix_scatter<<<nb, SORT_THREADS, 0, s>>>(kin, vin, kout, vout, R, shift, nb, table, gh);
    uint64_t* tk = kin; kin = kout; kout = tk;
    uint32_t* tv = vin; vin = vout; vout = tv;
  }
  return hipGetLastError();
}

hipError_t gcr_launch_tile_ranges(const uint64_t* keys, int64_t R, uint32_t* ranges, int T,
                                  hipStream_t s) {
  hipError_t e = hipMemsetAsync(ranges, 0, sizeof(uint32_t) * 2 * (size_t)T, s);
  if (e != hipSuccess) return e;
  if (R <= 0) return hipSuccess;
  k_tile_ranges<<<(unsigned)((R + 255) / 256), 256, 0, s>>>(keys, R, ranges);
  return hipGetLastError();
}
