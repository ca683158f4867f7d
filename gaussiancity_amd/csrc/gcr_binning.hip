// gcr_binning.hip -- tile binning for gfx950: K3 key emit, K4 stable LSD radix sort, K5 ranges.
// Integer / byte work, HBM-bound; no MFMA.  wave64 ballots do the intra-wave digit matching.
#include <stdlib.h>
#include "gcr_device.h"
#include "gcr_internal.h"
#include "gcr_sort.h"

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

// ------------------------------------------------------------------- K3'/K4' (fast path)
// Counting-sort binning without global atomics.  Device-scope atomics on MI355X resolve at the
// memory side (the eight XCD L2s are not coherent): fire-and-forget adds are cheap, but RETURNING
// adds -- what a scatter needs to claim a slot -- measured ~13 G/s, which made a global-cursor
// scatter as slow as the radix sort it was meant to replace.  So slots are claimed in LDS:
//   A'  k_tile_table<false>: K1's per-block survivor lists are grouped into NG <= 512 groups;
//       workgroup g counts its instances per tile in an LDS table (ds_add) and stores the row
//       to table[g][0..T).
//   B'  k_table_colscan: exclusive prefix down every tile column (base[g][t] = instances of
//       tile t owned by groups < g), the tile totals, their prefix inside each 64-tile block and
//       the block totals; R and the longest list are accumulated into the frame summary.
//   C'  k_tile_table<true>: every workgroup rebuilds the tile starts from the block totals
//       (a <= 640-element scan), writes its slice of `ranges`, loads start + base[g][t] into LDS cursors and
//       claims slots with LDS returning atomics; the (depth<<32|index) key goes straight to its
//       final tile segment.
//   D'  k_tile_sort: bitonic sort of every tile segment in LDS.
// The slot order inside a tile is arbitrary but the key is unique, so the sorted list is
// deterministic: positive-float depth bits ascending, ties in ascending Gaussian index -- the
// order the reference gets from its stable radix sort on tile<<32|depth over instances emitted
// in index order (cr/rasterizer_impl.cu:66-99,255-260).
constexpr int RANK_MERGE_MAX = 1024;  // chunked rank sort + merge below, bitonic network above
// Threads per workgroup of the tile-table kernels (template parameter TT_THREADS): 1024, one workgroup per CU, 256
// groups (gcr_tile_table_groups: fewer, fatter groups = fewer table rows through HBM; rounds 1-2 ran 512 groups of 512
// threads while two tables fit a CU's LDS).  At 4K a table (130 KiB) only leaves room for one workgroup per CU anyway;
// the scatter there is bound by its 8-byte stores landing in 32-byte sectors, not by occupancy.
constexpr int TT_MAX_TBLOCKS = 640;  // 64-tile blocks: T <= 40960 > the LDS limit of 148 KiB / 4 B

// The K1 blocks' survivor lists laid end to end: pre[k] = survivors of blocks < k, pre[nblocks] = all of them.
// Thread t owns the PER consecutive blocks [t * PER, ...): `vc` are their counts (tt_load_counts, requested at the
// top of the kernel so that the round trip hides behind the caller's set-up); one wave scan + one pass over the wave
// totals.  `wtot`: THREADS / 64 words nobody else is using (the function starts and ends with a workgroup barrier).
template <int THREADS>
struct TtPrefix {
  static constexpr int PER = (GCR_K1_MAX_BLOCKS + THREADS) / THREADS;  // (+ the slot of the total)
  uint32_t vc[PER];
  GCR_DEV void load(const uint32_t* __restrict__ vis_count, int nblocks, int tid) {
#pragma unroll
    for (int u = 0; u < PER; u++) {
      const int bi = tid * PER + u;
      vc[u] = bi < nblocks ? vis_count[bi] : 0u;
    }
  }
  GCR_DEV void scan(uint32_t* pre, uint32_t* wtot, int nblocks, int tid) const {
    const int lane = tid & 63, w = tid >> 6;
    uint32_t mine = 0;
#pragma unroll
    for (int u = 0; u < PER; u++) mine += vc[u];
    const uint32_t incl = gcr_wave_incl_scan_u32(mine, lane);
    __syncthreads();
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    uint32_t run = incl - mine;
#pragma unroll
    for (int k = 0; k < THREADS / 64; k++)
      if (k < w) run += wtot[k];
#pragma unroll
    for (int u = 0; u < PER; u++) {
      const int bi = tid * PER + u;
      if (bi <= nblocks) pre[bi] = run;  // (bi == nblocks: the total; counts beyond the last block are 0)
      run += vc[u];
    }
    __syncthreads();
  }
};
// Where position f of the lists laid end to end lies in vis_list / vis_rec (binary search in the LDS prefix:
// pre[k] <= f < pre[k + 1]).
GCR_DEV uint32_t tt_position(const uint32_t* pre, int nblocks, int chunk, uint32_t f) {
  int lo = 0, hi = nblocks;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (pre[mid] <= f)
      lo = mid;
    else
      hi = mid;
  }
  return (uint32_t)lo * (uint32_t)chunk + (f - pre[lo]);
}
// Survivor record {index, depth bits, rect_x, rect_y} at position f of the lists laid end to end (binary search in the LDS prefix: pre[k] <= f < pre[k + 1]).
GCR_DEV uint4 tt_survivor(const uint32_t* pre, int nblocks, const uint4* __restrict__ vis_rec, int chunk, uint32_t f) {
  return vis_rec[tt_position(pre, nblocks, chunk, f)];
}

template <bool SCATTER, int TT_THREADS>
__global__ __launch_bounds__(TT_THREADS) void k_tile_table(int T, int gx, int G, int nblocks_k1, int chunk,
                                                           const uint4* __restrict__ vis_rec,
                                                           const uint32_t* __restrict__ vis_count,
                                                           uint32_t* __restrict__ table,
                                                           const uint32_t* __restrict__ tile_total,
                                                           const uint32_t* __restrict__ tile_local,
                                                           const uint32_t* __restrict__ blk_total,
                                                           uint32_t* __restrict__ ranges,
                                                           uint64_t* __restrict__ pairs,
                                                           unsigned long long* __restrict__ frame,
                                                           unsigned long long cap_instances,
                                                           unsigned long long cap_list,
                                                           unsigned long long* __restrict__ host_word,
                                                           unsigned int seq,
                                                           const unsigned long long* __restrict__ block_tiles,
                                                           const uint4* __restrict__ banded, int banded_capacity) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gcr_smem[];
  uint32_t* cnt = reinterpret_cast<uint32_t*>(gcr_smem);  // [T]
  __shared__ uint32_t pre[GCR_K1_MAX_BLOCKS + 1];         // prefix of ALL K1 blocks' list lengths
  __shared__ uint32_t blk_base[TT_MAX_TBLOCKS];           // first instance of every 64-tile block
  __shared__ uint32_t wtot[TT_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  uint32_t* __restrict__ row = table + (size_t)blockIdx.x * T;
  TtPrefix<TT_THREADS> blocks;
  blocks.load(vis_count, nblocks_k1, tid);
  if (!SCATTER) {
    // K1 is complete when this kernel starts, so its first workgroup sums the K1 blocks' shares of num_rendered
    // (what the reference gets from its inclusive scan, cr/rasterizer_impl.cu:228-238), starts the frame summary
    // {R, longest list (added by the column scan), go} and publishes R to the host at once: (frame tag << 32 | R)
    // in ONE 8-byte store to pinned memory that the calling thread polls -- no copy, no event, and the host is
    // released while the tile tables, the scatter, the sort and the blend still run.
    if (blockIdx.x == 0) {
      unsigned long long part = 0ull, pf = 0ull;
      for (int b = tid; b < nblocks_k1; b += TT_THREADS) {
        const unsigned long long v = block_tiles[b];
        part += v & ~GCR_PREFILTER_FLAG;
        pf |= v & GCR_PREFILTER_FLAG;  // gcr_camera.prefiltered and a Gaussian behind the near plane (gcr_internal.h)
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
      __shared__ unsigned long long wpart[TT_THREADS / 64];
      if (lane == 0) wpart[w] = part;
      const int prefilter_violation = __syncthreads_or(pf != 0ull ? 1 : 0);
      if (tid == 0) {
        unsigned long long total = 0ull;
#pragma unroll
        for (int k = 0; k < TT_THREADS / 64; k++) total += wpart[k];
        if (prefilter_violation) total = GCR_PREFILTER_MARK;  // (beyond every capacity: the rest of the frame is vetoed)
        frame[0] = total;
        frame[1] = 0ull;
        frame[2] = 0ull;
        frame[GCR_FRAME_PIECE] = 0ull;  // no backward state yet (set by a forward blend that writes it)
        frame[GCR_FRAME_BANDED] = banded != nullptr ? 1ull : 0ull;  // which numbering the table rows were counted in
        if (host_word != nullptr)
          gcr_store_to_host(host_word, ((unsigned long long)seq << 32) | (total > 0xffffffffull ? 0xffffffffull : total));
      }
    }
    for (int t = tid; t < T; t += TT_THREADS) cnt[t] = 0u;
  } else {
    // Speculative launch: the host may not know R yet.  Every workgroup takes the same decision
    // from the device-side frame summary; workgroup 0 publishes it for the later kernels.
    const bool go = frame[0] <= cap_instances && frame[1] <= cap_list;
    if (blockIdx.x == 0 && tid == 0) {
      frame[2] = go ? 1ull : 0ull;
      // the frame's longest list, for the host's next length hint (pinned memory, nobody waits for it)
      if (host_word != nullptr) gcr_store_to_host(host_word, frame[1]);
    }
    if (!go) return;
    // tile starts = exclusive prefix of the 64-tile block totals (<= TT_MAX_TBLOCKS values, so
    // every workgroup redoes this tiny scan instead of paying a single-block kernel for it)
    const int ntb = (T + 63) / 64;
    uint32_t running = 0;
    for (int b0 = 0; b0 < ntb; b0 += TT_THREADS) {
      const int bi = b0 + tid;
      const uint32_t v = bi < ntb ? blk_total[bi] : 0u;
      const uint32_t incl = gcr_wave_incl_scan_u32(v, lane);
      if (lane == 63) wtot[w] = incl;
      __syncthreads();
      uint32_t before = 0, all = 0;
#pragma unroll
      for (int k = 0; k < TT_THREADS / 64; k++) {
        const uint32_t x = wtot[k];
        if (k < w) before += x;
        all += x;
      }
      if (bi < ntb) blk_base[bi] = running + before + incl - v;
      running += all;
      __syncthreads();
    }
    // this workgroup's slice of the tile ranges (identifyTileRanges upstream,
    // cr/rasterizer_impl.cu:104-124) and all LDS cursors
    const int slice = (T + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_lo = blockIdx.x * slice, t_hi = min(T, t_lo + slice);
    for (int t = tid; t < T; t += TT_THREADS) {
      const uint32_t start = blk_base[t >> 6] + tile_local[t];
      cnt[t] = start + row[t];
      if (t >= t_lo && t < t_hi) {
        ranges[2 * t] = start;
        ranges[2 * t + 1] = start + tile_total[t];
      }
    }
  }
  // The workgroup's share of the frame's survivors: an EQUAL cut of the K1 blocks' lists laid end to end, not a run of
  // whole K1 blocks (rounds 1-5) -- a scene whose index order is spatially coherent (real GaussianCity points are a raster
  // scan of the BEV maps) leaves most K1 blocks without a survivor and a few with thousands, and whole blocks made the
  // few groups that own them do the frame's counting and scattering alone (round 6 A/B, C5 renumbered along a Morton
  // curve: count + scan 54 -> 352 us, scatter 195 -> 668 us alone).  Both instantiations cut the same way, from
  // vis_count alone.  (G, the blocks per group of the old cut, is unused.)
  (void)G;
  blocks.scan(pre, wtot, nblocks_k1, tid);
  const uint32_t survivors = pre[nblocks_k1];
  const uint32_t f_lo = (uint32_t)(((uint64_t)survivors * blockIdx.x) / gridDim.x);
  const uint32_t f_hi = (uint32_t)(((uint64_t)survivors * (blockIdx.x + 1u)) / gridDim.x);
  // `banded` (k_band_permute ran on this frame): the same survivors, renumbered by the tile band their rectangle starts
  // in -- the workgroup's share then lands in a few hundred neighbouring tiles instead of all over the image
  // (`banded` holds records while they fit -- banded_capacity of them -- and 4-byte positions in vis_rec beyond that:
  // k_band decides from the same count every kernel sees)
  const bool use_band = banded != nullptr && (SCATTER ? frame[GCR_FRAME_BANDED] != 0ull : true);
  const bool band_recs = survivors <= (uint32_t)banded_capacity;
  const uint32_t* __restrict__ banded_pos = reinterpret_cast<const uint32_t*>(banded);
  for (uint32_t f = f_lo + (uint32_t)tid; f < f_hi; f += TT_THREADS) {
    const uint4 sv = !use_band ? tt_survivor(pre, nblocks_k1, vis_rec, chunk, f)
                               : (band_recs ? banded[f] : vis_rec[banded_pos[f]]);
    const uint32_t rx = sv.z, ry = sv.w;
    const uint32_t minx = rx & 0xffffu, maxx = rx >> 16, miny = ry & 0xffffu, maxy = ry >> 16;
    const uint64_t key = ((uint64_t)sv.y << 32) | sv.x;
    for (uint32_t y = miny; y < maxy; y++)
      for (uint32_t x = minx; x < maxx; x++) {
        const uint32_t t = y * (uint32_t)gx + x;
        if (SCATTER) {
          const uint32_t pos = atomicAdd(&cnt[t], 1u);  // ds_add_rtn_u32
          pairs[pos] = key;
        } else {
          atomicAdd(&cnt[t], 1u);  // ds_add_u32
        }
      }
  }
  if (!SCATTER) {
    __syncthreads();
    for (int t = tid; t < T; t += TT_THREADS) row[t] = cnt[t];
  }
}

// ------------------------------------------------------------------------------------------- band sort (round 6)
// A scatter workgroup's instances go wherever its Gaussians happen to lie: with 256 groups a (group, tile) pair holds
// 0.5 (C3) to 1.4 (C5) instances, so every 8-byte key is a 32-byte sector written alone -- 3.9x the bytes at C5, the
// scatter's time (round 6 A/B: the same frame with its Gaussians renumbered along a Morton curve scatters in 82 us
// instead of 195).  For frames with many instances the survivors are therefore renumbered first, by the BAND of
// GCR_BANDS equal runs of tiles (row-major) that holds the first tile of their rectangle: two small kernels, a counting
// sort over BAND_CUTS equal cuts of the survivor lists.  The tile-table kernels then take equal cuts of THAT order: a group's
// instances fall into a few hundred neighbouring tiles, tens of them per tile, next to each other in `pairs`.
//   k_band_hist:    hist[g][b] = survivors of cut g whose rectangle starts in band b
//   k_band_permute: every workgroup sums the BAND_CUTS x GCR_BANDS table (128 KiB, out of L2) into band starts + its own
//                   offsets, then drops its survivors' indices into banded[] with LDS cursors
// The order inside a band is whatever the LDS atomics make it -- like the slot order inside a tile, it is not
// observable: the tile sort orders every list by its unique (depth, index) keys.
constexpr int GCR_BANDS = 256;
constexpr int BAND_CUTS = 128;  // workgroups of the two kernels (= rows of the histogram)
GCR_DEV uint32_t band_of(uint32_t rx, uint32_t ry, uint32_t gx, uint32_t band_mul) {
  const uint32_t t0 = (ry & 0xffffu) * gx + (rx & 0xffffu);
  const uint32_t b = (uint32_t)(((uint64_t)t0 * band_mul) >> 32);
  return b < (uint32_t)GCR_BANDS ? b : (uint32_t)GCR_BANDS - 1u;
}

constexpr int BAND_THREADS = 1024;  // one workgroup per CU: the loop is two dependent gathers per survivor, so width is what hides them
constexpr int BAND_ILP = 4;         // survivors per thread and trip, their gathers in flight together
template <bool PERMUTE>
__global__ __launch_bounds__(BAND_THREADS) void k_band(int gx, uint32_t band_mul, int nblocks_k1, int chunk,
                                                       const uint4* __restrict__ vis_rec,
                                                       const uint32_t* __restrict__ vis_count,
                                                       uint32_t* __restrict__ hist, uint4* __restrict__ banded,
                                                       int banded_capacity) {
  __shared__ uint32_t pre[GCR_K1_MAX_BLOCKS + 1];
  __shared__ uint32_t wtot[BAND_THREADS / 64];
  __shared__ uint32_t cur[GCR_BANDS];
  __shared__ uint32_t part[2][BAND_THREADS / GCR_BANDS][GCR_BANDS];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  TtPrefix<BAND_THREADS> blocks;
  blocks.load(vis_count, nblocks_k1, tid);
  if (PERMUTE) {
    // band b = tid mod GCR_BANDS, a quarter of the cuts per thread: the band's total over all cuts and its share in
    // front of this cut (rows are 1 KiB, read coalesced; the table comes out of L2)
    constexpr int Q = BAND_THREADS / GCR_BANDS, ROWS = BAND_CUTS / Q;
    const int b = tid & (GCR_BANDS - 1), q = tid / GCR_BANDS;
    uint32_t total = 0, before = 0;
#pragma unroll 16
    for (int r = 0; r < ROWS; r++) {
      const int g = q * ROWS + r;
      const uint32_t v = hist[(size_t)g * GCR_BANDS + b];
      total += v;
      before += g < (int)blockIdx.x ? v : 0u;
    }
    part[0][q][b] = total;
    part[1][q][b] = before;
    __syncthreads();
    if (tid < GCR_BANDS) {  // waves 0..3
      total = 0;
      before = 0;
#pragma unroll
      for (int k = 0; k < Q; k++) {
        total += part[0][k][tid];
        before += part[1][k][tid];
      }
      const uint32_t incl = gcr_wave_incl_scan_u32(total, lane);
      if (lane == 63) wtot[w] = incl;
      part[0][0][tid] = incl - total + before;  // (own slot: read above by this thread only)
    }
    __syncthreads();
    if (tid < GCR_BANDS) {
      uint32_t start = part[0][0][tid];
#pragma unroll
      for (int k = 0; k < GCR_BANDS / 64; k++)
        if (k < w) start += wtot[k];
      cur[tid] = start;
    }
  } else {
    if (tid < GCR_BANDS) cur[tid] = 0u;
  }
  blocks.scan(pre, wtot, nblocks_k1, tid);
  const uint32_t survivors = pre[nblocks_k1];
  const uint32_t f_lo = (uint32_t)(((uint64_t)survivors * blockIdx.x) / gridDim.x);
  const uint32_t f_hi = (uint32_t)(((uint64_t)survivors * (blockIdx.x + 1u)) / gridDim.x);
  const bool band_recs = survivors <= (uint32_t)banded_capacity;  // else: 4-byte positions (P of them always fit)
  uint32_t* __restrict__ banded_pos = reinterpret_cast<uint32_t*>(banded);
  for (uint32_t f0 = f_lo + (uint32_t)tid; f0 < f_hi; f0 += BAND_THREADS * BAND_ILP) {
    uint32_t pos[BAND_ILP];
    uint4 sv[BAND_ILP];
#pragma unroll
    for (int u = 0; u < BAND_ILP; u++) {
      const uint32_t f = f0 + (uint32_t)u * BAND_THREADS;
      pos[u] = f < f_hi ? tt_position(pre, nblocks_k1, chunk, f) : 0xffffffffu;
    }
#pragma unroll
    for (int u = 0; u < BAND_ILP; u++)
      if (pos[u] != 0xffffffffu) sv[u] = vis_rec[pos[u]];
#pragma unroll
    for (int u = 0; u < BAND_ILP; u++)
      if (pos[u] != 0xffffffffu) {
        const uint32_t b = band_of(sv[u].z, sv[u].w, (uint32_t)gx, band_mul);
        if (PERMUTE) {
          const uint32_t slot = atomicAdd(&cur[b], 1u);
          if (band_recs)
            banded[slot] = sv[u];
          else
            banded_pos[slot] = pos[u];
        } else {
          atomicAdd(&cur[b], 1u);
        }
      }
  }
  if (!PERMUTE) {
    __syncthreads();
    if (tid < GCR_BANDS) hist[(size_t)blockIdx.x * GCR_BANDS + tid] = cur[tid];
  }
}

// table[g][t] (g < NG <= 512) -> exclusive prefix over g in place; tile_total[t] = column sum.
// Workgroup = 64 tiles x 16 row groups; every thread keeps its <= 32 rows in registers, so the
// column is read exactly once with all loads in flight.  Wave 0 then scans the workgroup's 64 tile
// totals (tile_local = exclusive prefix inside the 64-tile block, blk_total = their sum) and folds
// the block's longest list into the frame summary with one device-scope atomicMax per 64 tiles
// (fire-and-forget; R itself was accumulated by K1b so that the host can read it earlier).
__global__ __launch_bounds__(1024) void k_table_colscan(uint32_t* __restrict__ table, int NG, int T,
                                                        uint32_t* __restrict__ tile_total,
                                                        uint32_t* __restrict__ tile_local,
                                                        uint32_t* __restrict__ blk_total,
                                                        unsigned long long* __restrict__ frame) {
  __shared__ uint32_t part[16][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane;
  const int rpg = (NG + 15) / 16;  // rows per row-group, <= 32
  const int r0 = grp * rpg;
  uint32_t v[32];
  uint32_t sum = 0;
#pragma unroll
  for (int r = 0; r < 32; r++) {
    const int g = r0 + r;
    v[r] = (r < rpg && g < NG && t < T) ? table[(size_t)g * T + t] : 0u;
  }
#pragma unroll
  for (int r = 0; r < 32; r++) sum += v[r];
  part[grp][lane] = sum;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const uint32_t p = part[k][lane];
    if (k < grp) before += p;
    all += p;
  }
  if (t < T) {
    uint32_t run = before;
#pragma unroll
    for (int r = 0; r < 32; r++) {
      const int g = r0 + r;
      if (r < rpg && g < NG) {
        table[(size_t)g * T + t] = run;
        run += v[r];
      }
    }
  }
  if (grp == 0) {  // wave 0: `all` is this lane's tile total (0 beyond T)
    const uint32_t mine = t < T ? all : 0u;
    const uint32_t incl = gcr_wave_incl_scan_u32(mine, lane);
    const uint32_t mx = gcr_wave_max_u32(mine);
    if (t < T) {
      tile_total[t] = mine;
      tile_local[t] = incl - mine;
    }
    if (lane == 63) {
      blk_total[blockIdx.x] = incl;
      if (mx) atomicMax(&frame[1], (unsigned long long)mx);
    }
  }
}

// Global-cursor variant of the scatter, used only when the tile table does not fit in LDS
// (T * 4 B > 160 KiB, i.e. images beyond ~8K x 5K): one returning device-scope atomic per instance.
// `frame` (optional) is the speculative-launch guard: a vetoed frame (capacity guess too short)
// must neither write past the capacity-sized `pairs` buffer nor advance the cursors the retry
// will scatter from.
__global__ __launch_bounds__(256) void k_scatter_instances(int chunk, const uint32_t* __restrict__ vis_list,
                                                           const uint32_t* __restrict__ vis_count,
                                                           const float4* __restrict__ rec, int gx,
                                                           uint32_t* __restrict__ cursor,
                                                           uint64_t* __restrict__ pairs,
                                                           const unsigned long long* __restrict__ frame) {
  if (frame != nullptr && frame[2] == 0ull) return;  // speculative launch vetoed
  const uint32_t nvis = vis_count[blockIdx.x];
  const uint32_t* __restrict__ my_list = vis_list + (size_t)blockIdx.x * chunk;
  for (uint32_t it = threadIdx.x; it < nvis; it += 256) {
    const int idx = (int)my_list[it];
    const float4 q2 = rec[(size_t)idx * GCR_REC_QUADS + 2];
    const uint64_t key = ((uint64_t)__float_as_uint(q2.y) << 32) | (uint32_t)idx;
    const uint32_t rx = __float_as_uint(q2.z), ry = __float_as_uint(q2.w);
    const uint32_t minx = rx & 0xffffu, maxx = rx >> 16, miny = ry & 0xffffu, maxy = ry >> 16;
    const uint32_t w = maxx - minx, n = w * (maxy - miny);
    for (uint32_t k0 = 0; k0 < n; k0 += 8) {
      uint32_t pos[8];
#pragma unroll
      for (uint32_t u = 0; u < 8; u++) {
        const uint32_t k = k0 + u;
        if (k < n) {
          const uint32_t t = (miny + k / w) * (uint32_t)gx + minx + k % w;
          pos[u] = atomicAdd(&cursor[(size_t)t * GCR_CURSOR_STRIDE], 1u);
        }
      }
#pragma unroll
      for (uint32_t u = 0; u < 8; u++)
        if (k0 + u < n) pairs[pos[u]] = key;
    }
  }
}

// One workgroup per tile sorts the tile's n <= capacity keys in LDS and writes the Gaussian
// indices (low 32 bits) to the sorted list: chunked rank sort + merge for n <= RANK_MERGE_MAX, bitonic network (padded to a
// power of two with ~0) above.
__global__ __launch_bounds__(256) void k_tile_sort(const uint32_t* __restrict__ ranges,
                                                   uint64_t* __restrict__ pairs, uint64_t* __restrict__ pairs_spare,
                                                   uint32_t* __restrict__ list,
                                                   const unsigned long long* __restrict__ frame, int lds_capacity,
                                                   uint4* __restrict__ lazy) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gcr_smem[];
  uint64_t* s = reinterpret_cast<uint64_t*>(gcr_smem);
  if (frame != nullptr && frame[2] == 0ull) return;  // speculative launch vetoed
  const int tid = threadIdx.x;
  const uint32_t r0 = ranges[2 * blockIdx.x], r1 = ranges[2 * blockIdx.x + 1];
  const int n = (int)(r1 - r0);
  if (lazy != nullptr) {
    // gcr_sort.h "lazy tile sort": a long list gets its first segment here, the forward blend sorts on as far as it walks
    if (n > GCR_LAZY_MIN) {
      uint32_t n_sorted = 0;
      uint64_t L = 0;
      gcr_lazy_extend<GCR_LAZY_CAP_K4>(s, pairs + r0, (uint32_t)n, n_sorted, L, list + r0, tid);
      if (tid == 0) lazy[blockIdx.x] = make_uint4(n_sorted, 0u, (uint32_t)L, (uint32_t)(L >> 32));
      return;
    }
    if (tid == 0) lazy[blockIdx.x] = make_uint4((uint32_t)max(n, 0), 0u, ~0u, ~0u);  // sorted whole below
  }
  if (n <= 0) return;
  if (n > lds_capacity) {  // longer than the LDS of this launch (a power of two >= 64): sorted runs + merge passes
    gcr_tile_sort_long(s, (uint32_t)lds_capacity, pairs + r0, pairs_spare + r0, list + r0, (uint32_t)n, tid);
    return;
  }
  if (n == 1) {
    if (tid == 0) list[r0] = (uint32_t)pairs[r0];
    return;
  }
  if (n <= RANK_MERGE_MAX) {
    // Chunked rank sort + binary-search merge (one barrier).  Keys are unique, so
    //   final position(i) = #{keys < key_i} = rank inside its own 64-key chunk
    //                                         + sum over the other chunks of lower_bound(key_i).
    // Phase 1: a wave's 64 keys belong to one chunk, so it counts with wave-uniform (broadcast)
    // 16-byte LDS reads and drops each key at its rank into the sorted copy `sb`.
    // Phase 2: six-step binary searches in the other sorted chunks.
    const int nchunks = (n + 63) >> 6, npad = nchunks << 6;
    uint64_t* sb = s + npad;
    for (int i = tid; i < npad; i += 256) s[i] = i < n ? pairs[r0 + i] : ~0ull;
    __syncthreads();
    const ulonglong2* s2 = reinterpret_cast<const ulonglong2*>(s);
    uint64_t mine[RANK_MERGE_MAX / 256];
    uint32_t rank[RANK_MERGE_MAX / 256];
#pragma unroll
    for (int e = 0; e < RANK_MERGE_MAX / 256; e++) {
      const int i = tid + e * 256;
      if (i < npad) {  // wave-uniform: npad is a multiple of 64
        const int c = i >> 6;
        const uint64_t key = s[i];
        uint32_t r = 0;
#pragma unroll 8
        for (int j = 0; j < 32; j++) {
          const ulonglong2 kk = s2[c * 32 + j];
          r += (kk.x < key ? 1u : 0u) + (kk.y < key ? 1u : 0u);
        }
        mine[e] = key;
        rank[e] = r;
        if (i < n) sb[c * 64 + r] = key;  // pads (~0) rank behind every real key of the chunk
      }
    }
    __syncthreads();
    const int last_real = n - (nchunks - 1) * 64;  // real keys in the last chunk (1..64)
#pragma unroll
    for (int e = 0; e < RANK_MERGE_MAX / 256; e++) {
      const int i = tid + e * 256;
      if (i < n) {
        const int c = i >> 6;
        const uint64_t key = mine[e];
        uint32_t pos = rank[e];
        for (int c2 = 0; c2 < nchunks; c2++) {
          if (c2 == c) continue;  // wave-uniform
          const uint64_t* chunk = sb + c2 * 64;
          const int len = c2 == nchunks - 1 ? last_real : 64;
          int lo = 0, hi = len;  // lower_bound: first index whose key is not < key
#pragma unroll
          for (int step = 0; step < 7; step++) {
            if (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (chunk[mid] < key)
                lo = mid + 1;
              else
                hi = mid;
            }
          }
          pos += (uint32_t)lo;
        }
        list[r0 + pos] = (uint32_t)key;
      }
    }
    return;
  }
  int N2 = 2;
  while (N2 < n) N2 <<= 1;
  for (int i = tid; i < N2; i += 256) s[i] = i < n ? pairs[r0 + i] : ~0ull;
  __syncthreads();
  gcr_bitonic_sort_lds(s, N2, tid);
  for (int i = tid; i < n; i += 256) list[r0 + i] = (uint32_t)s[i];
}

// ------------------------------------------------------------------------------------- K4
// (fallback path, used when a tile list exceeds the LDS sort capacity or "force_radix" is set)
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

// The radix path leaves an untouched tile at (0, 0) (cr/rasterizer_impl.cu:262-264 zeroes `ranges` first), which is all
// the reference's blend needs.  The backward pieces of this build address their slots as floor(start / P) + tile
// (gcr_internal.h), which needs CONTIGUOUS ranges: an empty tile must sit at (e, e) with e = the end of the last
// non-empty tile before it -- what the tile-table path produces by construction.  One workgroup, running prefix
// maximum of the range ends (they are non-decreasing over the non-empty tiles).
__global__ __launch_bounds__(256) void k_ranges_make_contiguous(uint32_t* __restrict__ ranges, int T) {
  __shared__ uint32_t wmax[4];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry_s = 0u;
  __syncthreads();
  for (int t0 = 0; t0 < T; t0 += 256) {
    const int t = t0 + tid;
    const uint32_t r0 = t < T ? ranges[2 * t] : 0u, r1 = t < T ? ranges[2 * t + 1] : 0u;
    uint32_t v = r1;  // inclusive prefix maximum of the ends over the 256 tiles of this round
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(v, o, 64);
      if (lane >= o) v = u > v ? u : v;
    }
    if (lane == 63) wmax[w] = v;
    __syncthreads();
    uint32_t before = carry_s;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (k < w) before = wmax[k] > before ? wmax[k] : before;
    const uint32_t incl = v > before ? v : before;
    // end of the last non-empty tile strictly before t
    uint32_t prev = __shfl_up(incl, 1, 64);
    if (lane == 0) prev = before;
    if (t < T && r0 == r1) {
      ranges[2 * t] = prev;
      ranges[2 * t + 1] = prev;
    }
    __syncthreads();
    if (tid == 255) carry_s = incl;
    __syncthreads();
  }
}

}  // namespace

hipError_t gcr_launch_emit(int P, const uint32_t* tiles_touched, const uint32_t* block_offsets,
                           const float4* rec, int gx, uint64_t* keys, uint32_t* vals,
                           hipStream_t s) {
  if (P <= 0) return hipSuccess;
  k_emit<<<(P + 255) / 256, 256, 0, s>>>(P, tiles_touched, block_offsets, rec, gx, keys, vals);
  return hipGetLastError();
}

// Geometry of the tile-table path: NG groups of G consecutive K1 blocks; 0 if the table does
// not fit in LDS.
bool gcr_band_sort_possible(int T) { return T >= GCR_BANDS; }

int gcr_tile_table_groups(int T, int nblocks_k1, int* G_out) {
  const size_t lds = (size_t)T * sizeof(uint32_t);
  // (160 KiB of LDS per workgroup: the table + 11 KiB of static arrays, the K1 blocks' prefix among them)
  if (lds > 148 * 1024 || (T + 63) / 64 > TT_MAX_TBLOCKS) return 0;
  // 1024-thread workgroups, one per CU -- on HALF of the CUs.  Round 3 A/B at C3 (same box, experiment build): 512 groups
  // of 512 threads (two tables per CU) 4 407-4 432 frames/s, 256 groups of 1024 threads 4 470-4 550 -- half the table rows to
  // write, scan and read back.  Round 6, with the survivor records and the equal cut (profiles/r06_tile_table_groups_ab.txt):
  // alone the count and the scatter take the same time with 128 groups as with 256 (C3 17.9 + 25.9 against 17.7 + 25.3 us)
  // -- they are set-up and table traffic, not survivors -- and with three frames in flight the CUs they leave alone are
  // worth more to the other frames' kernels: C3 5 476-5 510 -> 5 689-5 701 frames/s (96 groups 5 726-5 747), C5 1 039 ->
  // 1 072-1 099, the dense stress scene D1 1 232 -> 1 217 (96: 1 182, which is why it is not 96).
  int ng = 128;
#ifdef GCR_EXPERIMENTS
  if (const char* e = getenv("GCR_TT_GROUPS")) ng = atoi(e);
#endif
  if (ng > nblocks_k1) ng = nblocks_k1;
  if (ng < 1) ng = 1;
  int G = (nblocks_k1 + ng - 1) / ng;
  *G_out = G;
  return (nblocks_k1 + G - 1) / G;
}

static hipError_t tile_table_attr() {
  static hipError_t done = [] {
    const void* fns[4] = {(const void*)k_tile_table<false, 512>, (const void*)k_tile_table<true, 512>,
                          (const void*)k_tile_table<false, 1024>, (const void*)k_tile_table<true, 1024>};
    for (const void* f : fns) {
      const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 148 * 1024);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }();
  return done;
}
static inline bool tile_table_wide(int) { return true; }  // (the 512-thread instantiations remain for A/B builds)

hipError_t gcr_launch_tile_count(int T, int gx, int NG, int G, int nblocks_k1, int chunk, const uint4* vis_rec,
                                 const uint32_t* vis_count, uint32_t* table,
                                 uint32_t* tile_total, uint32_t* tile_local, uint32_t* blk_total,
                                 unsigned long long* frame, const unsigned long long* block_tiles,
                                 unsigned long long* host_R, unsigned int seq, uint4* banded, int banded_capacity,
                                 hipStream_t s) {
  hipError_t e = tile_table_attr();
  if (e != hipSuccess) return e;
  if (banded != nullptr) {
    // (the histogram borrows the head of the tile table, which the count kernel overwrites afterwards: T >= GCR_BANDS)
    const uint32_t band_mul = (uint32_t)((((uint64_t)GCR_BANDS << 32) + (uint64_t)T - 1) / (uint64_t)T);
    k_band<false><<<BAND_CUTS, BAND_THREADS, 0, s>>>(gx, band_mul, nblocks_k1, chunk, vis_rec, vis_count, table, nullptr,
                                                      banded_capacity);
    k_band<true><<<BAND_CUTS, BAND_THREADS, 0, s>>>(gx, band_mul, nblocks_k1, chunk, vis_rec, vis_count, table, banded,
                                                     banded_capacity);
  }
  if (tile_table_wide(T))
    k_tile_table<false, 1024><<<NG, 1024, (size_t)T * sizeof(uint32_t), s>>>(
        T, gx, G, nblocks_k1, chunk, vis_rec, vis_count, table, nullptr, nullptr, nullptr, nullptr, nullptr,
        frame, 0ull, 0ull, host_R, seq, block_tiles, banded, banded_capacity);
  else
    k_tile_table<false, 512><<<NG, 512, (size_t)T * sizeof(uint32_t), s>>>(
        T, gx, G, nblocks_k1, chunk, vis_rec, vis_count, table, nullptr, nullptr, nullptr, nullptr, nullptr,
        frame, 0ull, 0ull, host_R, seq, block_tiles, banded, banded_capacity);
  k_table_colscan<<<(T + 63) / 64, 1024, 0, s>>>(table, NG, T, tile_total, tile_local, blk_total, frame);
  return hipGetLastError();
}

hipError_t gcr_launch_tile_scatter(int T, int gx, int NG, int G, int nblocks_k1, int chunk, const uint4* vis_rec,
                                   const uint32_t* vis_count, uint32_t* table,
                                   const uint32_t* tile_total, const uint32_t* tile_local,
                                   const uint32_t* blk_total, uint32_t* ranges, uint64_t* pairs,
                                   unsigned long long* frame, unsigned long long cap_instances,
                                   unsigned long long cap_list, unsigned long long* host_longest,
                                   const uint4* banded, int banded_capacity, hipStream_t s) {
  hipError_t e = tile_table_attr();
  if (e != hipSuccess) return e;
  if (tile_table_wide(T))
    k_tile_table<true, 1024><<<NG, 1024, (size_t)T * sizeof(uint32_t), s>>>(
        T, gx, G, nblocks_k1, chunk, vis_rec, vis_count, table, tile_total, tile_local, blk_total, ranges, pairs,
        frame, cap_instances, cap_list, host_longest, 0u, nullptr, banded, banded_capacity);
  else
    k_tile_table<true, 512><<<NG, 512, (size_t)T * sizeof(uint32_t), s>>>(
        T, gx, G, nblocks_k1, chunk, vis_rec, vis_count, table, tile_total, tile_local, blk_total, ranges, pairs,
        frame, cap_instances, cap_list, host_longest, 0u, nullptr, banded, banded_capacity);
  return hipGetLastError();
}

// cursor[t] = ranges[t].start: makes the global-cursor scatter idempotent (a frame may be
// scattered again from the same scan, e.g. gcr_forward_render called twice on one preprocess).
__global__ __launch_bounds__(256) void k_restore_cursors(const uint32_t* __restrict__ ranges,
                                                         uint32_t* __restrict__ cursor, int T) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < T) cursor[(size_t)t * GCR_CURSOR_STRIDE] = ranges[2 * t];
}

hipError_t gcr_launch_scatter_instances(int nblocks, int chunk, const uint32_t* vis_list,
                                        const uint32_t* vis_count, const float4* rec, int gx,
                                        uint32_t* tile_cursor, uint64_t* pairs, const uint32_t* ranges, int T,
                                        const unsigned long long* frame, hipStream_t s) {
  if (nblocks <= 0) return hipSuccess;
  if (frame == nullptr)  // staged (retry) path: the cursors may have been consumed by an earlier scatter
    k_restore_cursors<<<(T + 255) / 256, 256, 0, s>>>(ranges, tile_cursor, T);
  k_scatter_instances<<<nblocks, 256, 0, s>>>(chunk, vis_list, vis_count, rec, gx, tile_cursor, pairs, frame);
  return hipGetLastError();
}

// fallback only: tiles touched per Gaussian (0 if culled; array zeroed first) from K1's visible
// lists, then per-256 block sums in index order for k_emit.
__global__ __launch_bounds__(256) void k_fill_tiles_touched(int chunk, const uint32_t* __restrict__ vis_list,
                                                            const uint32_t* __restrict__ vis_count,
                                                            const float4* __restrict__ rec,
                                                            uint32_t* __restrict__ tiles_touched) {
  const uint32_t nvis = vis_count[blockIdx.x];
  const uint32_t* __restrict__ my_list = vis_list + (size_t)blockIdx.x * chunk;
  for (uint32_t it = threadIdx.x; it < nvis; it += 256) {
    const uint32_t idx = my_list[it];
    const float4 q2 = rec[(size_t)idx * GCR_REC_QUADS + 2];
    const uint32_t rx = __float_as_uint(q2.z), ry = __float_as_uint(q2.w);
    tiles_touched[idx] = ((rx >> 16) - (rx & 0xffffu)) * ((ry >> 16) - (ry & 0xffffu));
  }
}

__global__ __launch_bounds__(256) void k_block_sums(int P, const uint32_t* __restrict__ tiles_touched,
                                                    uint32_t* __restrict__ block_sums) {
  __shared__ uint32_t wsum[4];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const uint32_t t = idx < P ? tiles_touched[idx] : 0u;
  const uint32_t ws = gcr_wave_sum_u32(t);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = ws;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

hipError_t gcr_launch_tiles_touched(int P, int nblocks, int chunk, const uint32_t* vis_list,
                                    const uint32_t* vis_count, const float4* rec, uint32_t* tiles_touched,
                                    uint32_t* block_sums, hipStream_t s) {
  if (P <= 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(tiles_touched, 0, sizeof(uint32_t) * (size_t)P, s);
  if (e != hipSuccess) return e;
  k_fill_tiles_touched<<<nblocks, 256, 0, s>>>(chunk, vis_list, vis_count, rec, tiles_touched);
  k_block_sums<<<(P + 255) / 256, 256, 0, s>>>(P, tiles_touched, block_sums);
  return hipGetLastError();
}

int gcr_tile_sort_capacity(void) { return 4096; }  // 32 KiB of LDS per workgroup at most

// `list_length_hint`: the longest list the caller expects.  It only sizes the LDS of the launch (a power of two
// between 64 and the capacity): lists within it take the in-LDS paths, longer ones the run + merge path through
// the spare key buffer `pairs_spare` (same size as `pairs`) -- any length is sorted correctly.
//
// `lazy` ([T] uint4 states, or null = sort everything): lists longer than GCR_LAZY_MIN only get their first segment
// (gcr_sort.h); the LDS of the launch is then at most 16 KiB instead of 32.
hipError_t gcr_launch_tile_sort(const uint32_t* ranges, int T, uint64_t* pairs, uint64_t* pairs_spare,
                                uint32_t* list, int64_t list_length_hint, const unsigned long long* frame,
                                uint4* lazy, hipStream_t s) {
  if (T <= 0) return hipSuccess;
  static_assert(GCR_LAZY_MIN == RANK_MERGE_MAX, "lists the lazy path leaves to this kernel take the rank + merge path");
  const int cap = lazy != nullptr ? GCR_LAZY_MIN : gcr_tile_sort_capacity();
  size_t n2 = 64;
  while ((int64_t)n2 < list_length_hint && n2 < (size_t)cap) n2 <<= 1;
  // bitonic path: n2 keys; rank/merge path (<= RANK_MERGE_MAX keys): two buffers of n rounded up to 64
  size_t lds = n2 <= (size_t)RANK_MERGE_MAX ? 2 * n2 * sizeof(uint64_t) : n2 * sizeof(uint64_t);
  if (lazy != nullptr && lds < gcr_lazy_lds_keys(GCR_LAZY_CAP_K4) * sizeof(uint64_t))
    lds = gcr_lazy_lds_keys(GCR_LAZY_CAP_K4) * sizeof(uint64_t);
  k_tile_sort<<<T, 256, lds, s>>>(ranges, pairs, pairs_spare, list, frame, (int)n2, lazy);
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
  k_ranges_make_contiguous<<<1, 256, 0, s>>>(ranges, T);
  return hipGetLastError();
}
