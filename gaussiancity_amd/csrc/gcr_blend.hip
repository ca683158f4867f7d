// gcr_blend.hip -- K6 forward alpha compositing and K7 reverse-walk gradient for gfx950.
//
// One 256-thread workgroup (4 x wave64) per 16x16 tile.  Wave w owns the 8x8-pixel quadrant
// (w&1, w>>1) of the tile and each of its four 16-lane DPP ROWS owns one 4x4-pixel BLOCK of that
// quadrant (lane = row*16 + y*4 + x).  A tile's depth-sorted list is consumed in chunks of 256
// entries: the 256 threads gather one 48-byte Gaussian record each (3 x dwordx4 from <= 2 cache
// lines) and stage it in LDS, so the blend loop never touches global memory.  Differences from
// cr/forward.cu:238-346 that do not change results:
//   * colour is staged in LDS too (the reference gathers it per contributing pixel, :328);
//   * a per-Gaussian conservative bound pmin = -ln(255*opacity) - 1e-3 is staged; pixels with
//     power < pmin are skipped before exp() -- exactly pixels the alpha < 1/255 test (:318)
//     would skip anyway;
//   * BLOCK CULLING: the staging thread computes which of the tile's sixteen 4x4 blocks the
//     ellipse {power >= pmin} can reach (gcr_cull.h: exact per block row, conservative, checked
//     by brute force on the host).  Every wave compacts the chunk into FOUR lists, one per row,
//     and each row walks only its own block's entries: the four rows of a wave work on different
//     Gaussians in the same instruction (per-lane LDS addresses, identical inside a row).  An
//     entry missing from a block's list has power < pmin for its 16 pixels, i.e. it is skipped
//     by those pixels upstream as well, and skipped entries never change a pixel's state -- so
//     n_contrib / final_T / colour are unchanged.  Measured on the BASELINE scenes this walks
//     32 % (C3) to 45 % (C2) fewer wave-steps than 8x8 bounding-box culling;
//   * `contributor` is recovered from the list position instead of a per-lane counter;
//   * a wave stops as soon as its own 64 pixels are done, on top of the block vote (:284-286).
// Arithmetic: gcr-fp32-v2 (gcr_device.h) -> out_color / final_T / n_contrib are bit-identical
// to the oracle.
#include "gcr_cull.h"
#include "gcr_device.h"
#include "gcr_internal.h"

namespace {

constexpr int CHUNK = 256;
constexpr int LIST_STRIDE = CHUNK + 8;       // u16 slots per row list: entries + pipeline pads
constexpr uint32_t ENTRY_BYTES = 48;
constexpr uint32_t SENT_OFF = CHUNK * ENTRY_BYTES;  // byte offset of the sentinel entry sE[CHUNK]
constexpr uint32_t NO_ENTRY = 0xFFFFFFFFu;

// One staged list entry: 48 bytes, read by the blend loops as three 16-byte quads at one address.
//   K6: a = (x, y, -0.5*conic.x, -conic.y)   b = (-0.5*conic.z, opacity, r, g)   c = (b, pmin, -, -)
//   K7: a = (x, y, conic.x, conic.y)         b = (conic.z, opacity, r, g)
//       c = (b, pmin, bits(list entry number), bits(byte offset of the entry's accumulator column))
// The sentinel (slot CHUNK) has pmin = +inf: no pixel is ever in range of it; list pads point at it.
struct __attribute__((aligned(16))) StagedEntry {
  float4 a, b, c;
};
static_assert(sizeof(StagedEntry) == ENTRY_BYTES, "StagedEntry must be 48 bytes");

// pmin such that power < pmin  =>  opacity*exp(power) < 1/255 with a 1e-3 safety margin.
// Clamped to >= -87 so the blend loops may use the guard-free exponential (see gcr_device.h).
GCR_DEV float gcr_alpha_skip_bound(float opacity) {
  if (!(opacity > 0.0f)) return __builtin_inff();  // alpha <= 0 < 1/255: always skipped
  return gcr_max(-87.0f, -__builtin_logf(255.0f * opacity) - 1.0e-3f);
}

template <bool FAST_EXP>
GCR_DEV float blend_exp(float x) {
  return FAST_EXP ? gcr_expf_fast(x) : gcr_expf_noguard(x);
}

// Lane geometry shared by K6 and K7: which pixel a lane owns and which mask bits its wave's rows use.
struct LaneGeom {
  int lane, w, row, pxi, pyi;
  int bit[4];  // mask bit of the 4x4 block owned by row rr of this wave
};
GCR_DEV LaneGeom lane_geom(int tid, int tx, int ty) {
  LaneGeom g;
  g.lane = tid & 63;
  g.w = tid >> 6;
  g.row = g.lane >> 4;
  const int li = g.lane & 15;
  const int bx = (g.w & 1) * 2 + (g.row & 1), by = (g.w >> 1) * 2 + (g.row >> 1);
  g.pxi = tx * GCR_TILE_X + bx * 4 + (li & 3);
  g.pyi = ty * GCR_TILE_Y + by * 4 + (li >> 2);
#pragma unroll
  for (int rr = 0; rr < 4; rr++) g.bit[rr] = ((g.w >> 1) * 2 + (rr >> 1)) * 4 + (g.w & 1) * 2 + (rr & 1);
  return g;
}

// byte offset -> chunk slot (offsets are multiples of 48 below 2^14: 1366/65536 is 1/48 rounded up,
// exact for slots < 2048)
GCR_DEV uint32_t slot_of_offset(uint32_t off) { return (off * 1366u) >> 16; }

// ------------------------------------------------------------------------------------- K6
// What bounds this loop is VALU issue (tools/valu_probe.hip: a wave64 v_fma_f32 occupies its SIMD
// for two cycles; v_cmp -> SGPR, SGPR operands and v_cndmask with an SGPR-pair mask cost ~1.7x that;
// v_pk_*_f32 cost 1.9x for two results, so packing buys nothing).  Hence
//   * per-row lists (above) -- fewer steps;
//   * each step is branch-free with ONE wave-uniform skip and 38 VALU instructions:
//       - conic pre-scaled by the staging thread (-0.5*cx, -cy, -0.5*cz): power in 6 instead of 7
//         (a scaling by a power of two commutes with every rounding, so `power` has the bits of
//         fma(-(cy*dx), dy, -0.5*fma(cz*dy, dy, (cx*dx)*dx)) whenever no product is denormal, and a
//         denormal term is absorbed by exp(): the oracle's expression is kept there);
//       - gcr-fp32-v2 exp in 11;
//       - "done" folded into the SIGN of the working transmittance Tw (T*(1-a) < 1e-4 upstream
//         flips it; |Tw| stays the T upstream leaves behind), and the contribution weight forced to
//         0 instead of selecting the three colour accumulators:
//             a_eff = valid ? alpha : 0        test_T = Tw*(1 - a_eff)      keep = !(test_T < 1e-4)
//             C    += (colour*w)*Tw with w = keep ? a_eff : 0               Tw = keep ? test_T : -|Tw|
//         A pixel that skips the entry has a_eff = 0 -> test_T = Tw (>= 1e-4 while live) -> keep, w = 0;
//         a finished pixel has Tw < 0 -> !keep -> Tw unchanged.  (colour*0)*Tw adds an exact zero.
//         Two corner cases are knowingly different from upstream, both on garbage input only: a
//         NON-FINITE colour poisons every pixel of the 4x4 blocks that evaluate the Gaussian
//         (upstream: only pixels it contributes to), and an accumulator that is exactly -0.0 (needs a
//         colour below 1e-40) may become +0.0.
//   * SORT (template): the workgroup sorts its own tile first.  GaussianCity's scenes have ~150 entries per tile, and
//     sorting 150 keys is a microsecond of work for the workgroup that is about to gather them anyway -- as a
//     separate kernel (K4) it is a latency-bound launch of 18 us on the frame's critical path.  Lists of up to
//     CHUNK keys are rank-sorted in LDS (every thread counts the keys below its own through wave-uniform 16-byte
//     reads; unique keys => rank = position); a longer list -- only possible when the caller's length hint was
//     stale, the host then switches to the K4 path -- is ranked the same way through global loads: slow, correct.
//     The sorted indices are also written to `list_out` for the backward.
template <bool FAST_EXP, bool SORT>
__global__ __launch_bounds__(256) void k_blend_fwd(const GcrBlendArgs a) {
  __shared__ StagedEntry sE[CHUNK + 1];
  __shared__ uint32_t sMask[CHUNK];
  __shared__ uint16_t sList[4][4][LIST_STRIDE];  // [wave][row]: byte offsets into sE, list order

  if (a.frame != nullptr && a.frame[2] == 0ull) return;  // speculative launch vetoed
  const int tile = blockIdx.x;
  const int tx = tile % a.gx, ty = tile / a.gx;
  const int tid = threadIdx.x;
  const LaneGeom g = lane_geom(tid, tx, ty);
  const int lane = g.lane, w = g.w;
  const bool inside = g.pxi < a.W && g.pyi < a.H;
  const float pixx = (float)g.pxi, pixy = (float)g.pyi;
  const float tile_x0 = (float)(tx * GCR_TILE_X), tile_y0 = (float)(ty * GCR_TILE_Y);
  const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
  const int total = (int)(r1 - r0);
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const char* const sEb = reinterpret_cast<const char*>(sE);
  if (tid == 0) {
    sE[CHUNK].a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    sE[CHUNK].b = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    sE[CHUNK].c = make_float4(0.0f, __builtin_inff(), 0.0f, 0.0f);
  }

  float Tw = inside ? 1.0f : -1.0f;
  float C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
  uint32_t last_contributor = 0;
  // pieces (gcr_internal.h "backward pieces"): the list is walked in equal pieces of <= a.piece entries
  const uint32_t piece_P = (uint32_t)a.piece;
  const int cs = (int)gcr_piece_size((uint32_t)total, piece_P);
  const uint32_t sbase = r0 / piece_P + (uint32_t)tile;
  uint32_t entered = 0;  // pieces this workgroup walked into (block-uniform)

  uint32_t sorted_id = 0;  // SORT, total <= CHUNK: the Gaussian of list position `tid`
  if (SORT && total > 0) {
    const uint64_t* __restrict__ seg = a.pairs + r0;
    uint32_t* __restrict__ out = a.list_out + r0;
    if (total <= CHUNK) {
      // keys into LDS (aliasing the record buffer, which is not live yet), padded with ~0 to a multiple of 2
      uint64_t* sKey = reinterpret_cast<uint64_t*>(sE);
      const uint64_t mine = tid < total ? seg[tid] : ~0ull;
      sKey[tid] = mine;
      __syncthreads();
      const ulonglong2* k2 = reinterpret_cast<const ulonglong2*>(sKey);
      uint32_t rank = 0;
      for (int j = 0; j < (total + 1) / 2; j++) {  // wave-uniform (broadcast) reads
        const ulonglong2 kk = k2[j];
        rank += (kk.x < mine ? 1u : 0u) + (kk.y < mine ? 1u : 0u);
      }
      __syncthreads();  // every thread has read the keys: sMask may be written (it does not alias, sE does)
      if (tid < total) {
        sMask[rank] = (uint32_t)mine;  // sMask doubles as the sorted index list until the staging overwrites it
        out[rank] = (uint32_t)mine;
      }
      __syncthreads();
      sorted_id = tid < total ? sMask[tid] : 0u;
      __syncthreads();  // sMask / sE are free for the staging below
    } else {
      // stale length hint: rank through global memory (n^2 / 256 compares per thread), then blend from list_out
      for (int i = tid; i < total; i += 256) {
        const uint64_t mine = seg[i];
        uint32_t rank = 0;
        for (int j = 0; j < total; j++) rank += seg[j] < mine ? 1u : 0u;
        out[rank] = (uint32_t)mine;
      }
      __syncthreads();  // workgroup-scope visibility of list_out for the staging loads below
    }
  }

// One list entry (record QA/QB/QC at byte offset OFF) against this lane's pixel.
#define GCR_BLEND_STEP(QA, QB, QC, OFF)                                                      \
  {                                                                                          \
    const float dx = QA.x - pixx, dy = QA.y - pixy;                                          \
    const float power = __builtin_fmaf(QA.w * dx, dy, __builtin_fmaf(QB.x * dy, dy, (QA.z * dx) * dx)); \
    const bool in_range = !(power > 0.0f) && !(power < QC.y);                                \
    if (__ballot(in_range) != 0ull) { /* wave-uniform */                                     \
      /* lanes outside [pmin, 0] may produce garbage; every use below is behind a select */  \
      const float araw = __builtin_fminf(0.99f, QB.y * blend_exp<FAST_EXP>(power));          \
      const bool valid = in_range && !(araw < 1.0f / 255.0f);                                \
      const float a_eff = valid ? araw : 0.0f;                                               \
      const float test_T = Tw * (1 - a_eff);                                                 \
      const bool keep = !(test_T < 0.0001f);                                                 \
      const float wgt = keep ? a_eff : 0.0f;                                                 \
      C0 = __builtin_fmaf(QB.z * wgt, Tw, C0);                                               \
      C1 = __builtin_fmaf(QB.w * wgt, Tw, C1);                                               \
      C2 = __builtin_fmaf(QC.x * wgt, Tw, C2);                                               \
      last_off = (keep && valid) ? (OFF) : last_off;                                         \
      Tw = keep ? test_T : -__builtin_fabsf(Tw);                                             \
    }                                                                                        \
  }
#define GCR_BLEND_LOAD(QA, QB, QC, OFF)                            \
  QA = *reinterpret_cast<const float4*>(sEb + (OFF));             \
  QB = *reinterpret_cast<const float4*>(sEb + (OFF) + 16);         \
  QC = *reinterpret_cast<const float2*>(sEb + (OFF) + 32);

  for (int base = 0; base < total; base += cs) {
    // block-wide vote (cr/forward.cu:284-286); also fences the previous chunk's LDS reads
    if (__syncthreads_count(!(Tw > 0.0f)) == 256) break;
    // crossing a piece boundary: checkpoint of the per-pixel state for the backward (|Tw| = T; a finished pixel's
    // checkpoint is never read).  256 x 16 bytes, one coalesced 4 KB store per boundary.
    if (entered > 0u && a.ckpt != nullptr)
      a.ckpt[(size_t)(sbase + entered - 1u) * 256u + tid] = make_float4(__builtin_fabsf(Tw), C0, C1, C2);
    entered++;
    const int n = min(cs, total - base);
    uint32_t my_mask = 0;
    if (tid < n) {
      const uint32_t id = !SORT ? a.list[r0 + base + tid]
                                : ((total <= CHUNK && cs == total) ? sorted_id : a.list_out[r0 + base + tid]);
      const float4* __restrict__ rec = a.rec + (size_t)id * GCR_REC_QUADS;
      const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
      const float pmin = gcr_alpha_skip_bound(q1.y);
      my_mask = gcr_block_mask(q0.x, q0.y, q0.z, q0.w, q1.x, pmin, tile_x0, tile_y0);
      sE[tid].a = make_float4(q0.x, q0.y, -0.5f * q0.z, -q0.w);
      sE[tid].b = make_float4(-0.5f * q1.x, q1.y, q1.z, q1.w);
      sE[tid].c = make_float4(q2.x, pmin, 0.0f, 0.0f);
    }
    sMask[tid] = my_mask;
    __syncthreads();
    if (__ballot(Tw > 0.0f) == 0ull) continue;  // this wave's quadrant is finished; keep voting
    // compact the chunk into this wave's four row lists (ascending list order is preserved)
    int cnt[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (k * 64 < n) {  // wave-uniform
        const int jj = k * 64 + lane;
        const uint32_t m = sMask[jj];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const bool rel = (m >> g.bit[rr]) & 1u;
          const uint64_t bal = __ballot(rel);
          if (rel) sList[w][rr][cnt[rr] + __popcll(bal & lt_mask)] = (uint16_t)(jj * (int)ENTRY_BYTES);
          cnt[rr] += __popcll(bal);
        }
      }
    }
    const int maxcnt = max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3]));
    // shorter lists are padded with the sentinel up to the longest one, plus three slots the
    // software pipeline below reads ahead (never consumed)
#pragma unroll
    for (int rr = 0; rr < 4; rr++)
      for (int s = cnt[rr] + lane; s < maxcnt + 3; s += 64) sList[w][rr][s] = (uint16_t)SENT_OFF;
    __builtin_amdgcn_wave_barrier();
    // Software pipeline, two entries per trip with ping-pong registers: while entry i blends,
    // the record of entry i+1 and the list slot of entry i+2 are already in flight.
    const uint16_t* lp = &sList[w][g.row][0];
    float4 qa0, qb0, qa1, qb1;
    float2 qc0, qc1;
    uint32_t e0 = lp[0], e1 = lp[1];
    uint32_t last_off = NO_ENTRY;
    GCR_BLEND_LOAD(qa0, qb0, qc0, e0)
    for (int i = 0; i < maxcnt; i += 2, lp += 2) {
      GCR_BLEND_LOAD(qa1, qb1, qc1, e1)
      const uint32_t cur0 = e0;
      e0 = lp[2];
      GCR_BLEND_STEP(qa0, qb0, qc0, cur0)
      if (i + 1 >= maxcnt) break;
      GCR_BLEND_LOAD(qa0, qb0, qc0, e0)
      const uint32_t cur1 = e1;
      e1 = lp[3];
      GCR_BLEND_STEP(qa1, qb1, qc1, cur1)
      if (__ballot(Tw > 0.0f) == 0ull) break;  // every pixel of the wave is done
    }
    if (last_off != NO_ENTRY) last_contributor = (uint32_t)base + slot_of_offset(last_off) + 1u;
  }
#undef GCR_BLEND_STEP
#undef GCR_BLEND_LOAD
  if (inside) {
    const float Tout = __builtin_fabsf(Tw);
    const size_t pix_id = (size_t)a.W * g.pyi + g.pxi;
    const size_t plane = (size_t)a.H * a.W;
    a.final_T[pix_id] = Tout;
    a.n_contrib[pix_id] = last_contributor;
    a.out_color[pix_id] = C0 + Tout * a.bg[0];
    a.out_color[plane + pix_id] = C1 + Tout * a.bg[1];
    a.out_color[2 * plane + pix_id] = C2 + Tout * a.bg[2];
  }
  if (a.work != nullptr && entered > 0u) {
    // a tile that crossed a boundary also leaves its final colour (without background) in its last slot: a
    // backward piece that starts from a checkpoint needs C_final - C_prefix
    if (entered >= 2u)
      a.ckpt[(size_t)(sbase + gcr_piece_count((uint32_t)total, piece_P) - 1u) * 256u + tid] =
          make_float4(__builtin_fabsf(Tw), C0, C1, C2);
    // work items of the backward blend: one per piece this workgroup walked into (the pieces behind a saturated
    // tile's last one are nobody's business); `entered` consecutive places of the dense list, claimed with one atomic
    __shared__ uint32_t sWorkPos;
    if (tid == 0)
      sWorkPos = atomicAdd(reinterpret_cast<unsigned int*>(a.frame_out + GCR_FRAME_NWORK), entered);
    __syncthreads();
    for (uint32_t k = tid; k < entered; k += 256u)
      a.work[sWorkPos + k] = make_uint4((uint32_t)tile, r0, (uint32_t)total, k);
  }
  if (a.work != nullptr && tile == 0 && tid == 0) {
    a.frame_out[GCR_FRAME_PIECE] = piece_P;
    a.frame_out[GCR_FRAME_CKPT_OFF] = a.ckpt_off;
    a.frame_out[GCR_FRAME_WORK_OFF] = a.work_off;
  }
}

// ------------------------------------------------------------------------------------- K7
// cr/backward.cu:428-581.  Per pixel the reverse walk is the reference's.  What changes:
//
// (a) How the nine per-(pixel, Gaussian) gradient terms reach memory.  The reference issues nine global float
//     atomics per pixel per Gaussian; here
//   1. each 16-lane DPP row (= one 4x4 block, all lanes on the SAME Gaussian) reduce-scatters the nine terms
//      (31 VALU ops, gcr_row_reduce_scatter9; no LDS traffic),
//   2. lanes 0..8 of each row add the row sums into a per-piece LDS accumulator (one ds_add_f32 to nine
//      consecutive floats of the entry's accumulator row),
//   3. after the piece the nine sums of an entry are flushed by nine ADJACENT lanes into the Gaussian's 64-byte
//      gradient record (gcr_internal.h): one wave instruction = 4 entries = 4 cache lines
//     so global atomics drop from 9 per (pixel, Gaussian) to one cache-line transaction per (piece, Gaussian).
//
// (b) The unit of work is a (tile, PIECE) item of the list the forward left behind (gcr_internal.h "backward
//     pieces"), not a tile: a pixel whose walk goes on behind the piece starts from the forward's checkpoint.
//     Only entries below the row's max n_contrib are visited (skipped by every pixel upstream too,
//     contributor >= last_contributor, :511-512), and each row visits only the entries whose block mask includes its
//     4x4 block (see K6).
//
// (c) PERSISTENT, SOFTWARE-PIPELINED workgroups.  Per-wave phase clocks (tools/k7_clocks.py, round 3) showed why one
//     workgroup per tile ran at half the VALU rate it sustains while walking: a workgroup's life is a chain of
//     dependent memory round trips (ranges -> pixel state -> list -> records), then the VALU-bound walk, then the
//     flush -- and the four workgroups of a CU, started together, stay in step: all wait, all walk (VALU saturated
//     by four walkers), all flush.  Walking waves per SIMD swung between 0 and 3.5 with a mean of 1.7.  Here a
//     workgroup stays resident, strides over the work list, and has the NEXT item's loads in flight while it walks
//     the current one: the item's descriptor (scalar) at the top of the iteration, its list entries before the walk,
//     the record gather from inside the walk loop (two steps in, when the indices have arrived), pixel state and
//     checkpoints right after the walk, behind the flush.  Between two walks remain the flush, three barriers, the
//     LDS staging with the block masks and the list compaction.
//     The dense zero fill of the backward's outputs rides along in slices, one per item (non-temporal stores).
// Experiment builds (make -C csrc experiments): per-wave phase clocks.  With gcr_debug_set_clock_buffer() armed, lane 0
// of every wave stores {hw id, xcc id, s_memtime at the phase boundaries} of the first items it walks
// (tools/k7_clocks.py turns them into phase statistics).  The shipping build contains none of it.
#ifdef GCR_EXPERIMENTS
#define GCR_K7_NCLK 8
#define GCR_K7_ITEMS 4
#define GCR_K7_CLK(i) \
  if (clk_on) clk[i] = __builtin_readcyclecounter();
#else
#define GCR_K7_CLK(i)
#endif

// One work item, block-uniform (scalar registers): piece `k` of tile `tile`, list entries [lo, lo + n).
struct BwdItem {
  int tile;
  uint32_t r0, lo, n, slot_ck, slot_final;
  bool has_ckpt;  // the list goes on behind this piece: some pixel may start from the checkpoint at lo + n
};
GCR_DEV BwdItem bwd_item(const uint4 d, uint32_t P) {
  BwdItem it;
  it.tile = (int)d.x;
  it.r0 = d.y;
  const uint32_t len = d.z, k = d.w;
  const uint32_t npieces = gcr_piece_count(len, P), cs = gcr_piece_size(len, P);
  it.lo = k * cs;
  it.n = min(cs, len - it.lo);
  const uint32_t sbase = d.y / P + d.x;
  it.slot_ck = sbase + k;
  it.slot_final = sbase + npieces - 1u;
  it.has_ckpt = k + 1u < npieces;
  return it;
}
struct BwdEntry {
  float4 q0, q1, q2;
  uint32_t id;
};
struct BwdPixel {
  float T_final;
  uint32_t last_contributor;
  float d0, d1, d2;
};
// chunk slot `tid` of an item holds list entry lo + n - 1 - tid (back to front)
GCR_DEV uint32_t bwd_load_id(const GcrBlendArgs& a, const BwdItem& it, int tid) {
  return tid < (int)it.n ? a.list[it.r0 + it.lo + it.n - 1u - (uint32_t)tid] : 0u;
}
GCR_DEV BwdEntry bwd_gather(const GcrBlendArgs& a, const BwdItem& it, uint32_t id, int tid) {
  BwdEntry e;
  e.id = id;
  e.q0 = e.q1 = e.q2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (tid < (int)it.n) {
    const float4* __restrict__ rec = a.rec + (size_t)id * GCR_REC_QUADS;
    e.q0 = rec[0];
    e.q1 = rec[1];
    e.q2 = rec[2];
  }
  return e;
}
GCR_DEV BwdPixel bwd_load_pixel(const GcrBlendArgs& a, const BwdItem& it, int tid) {
  const LaneGeom g = lane_geom(tid, it.tile % a.gx, it.tile / a.gx);
  BwdPixel p = {0.0f, 0u, 0.0f, 0.0f, 0.0f};
  if (g.pxi < a.W && g.pyi < a.H) {
    const size_t pix_id = (size_t)a.W * g.pyi + g.pxi;
    const size_t plane = (size_t)a.H * a.W;
    p.T_final = a.final_T[pix_id];
    p.last_contributor = a.n_contrib[pix_id];
    p.d0 = a.dL_dpix[pix_id];
    p.d1 = a.dL_dpix[plane + pix_id];
    p.d2 = a.dL_dpix[2 * plane + pix_id];
  }
  return p;
}

template <bool FAST_EXP>
__global__ __launch_bounds__(256) void k_blend_bwd(const GcrBlendArgs a) {
  __shared__ StagedEntry sE[CHUNK + 1];
  __shared__ uint32_t sMask[CHUNK];
  __shared__ uint32_t sId[CHUNK];
  __shared__ uint16_t sList[4][4][LIST_STRIDE];
  __shared__ float sAcc[CHUNK * 9];  // entry-major: the nine sums of chunk slot j at [9j, 9j+9)

  const int tid = threadIdx.x;
  const uint32_t G = gridDim.x, wg = blockIdx.x;
  const uint32_t piece_P = (uint32_t)a.frame_in[GCR_FRAME_PIECE];
  const uint32_t nwork = piece_P != 0u ? (uint32_t)a.frame_in[GCR_FRAME_NWORK] : 0u;
  // zero fill of the dense outputs: every workgroup streams S slices, one per work item it walks and the rest at
  // the end (virtual block sidx * G + wg of G * S)
  const uint32_t S = max(1u, (nwork + G - 1u) / G);
  uint32_t slices_done = 0;
#ifdef GCR_EXPERIMENTS
#define GCR_FILL_ON !(a.debug_flags & 2)
#else
#define GCR_FILL_ON true
#endif
#ifdef GCR_EXPERIMENTS
#define GCR_LDS_ADD_ON !(a.debug_flags & 4)
#else
#define GCR_LDS_ADD_ON true
#endif
#define GCR_FILL_SLICE()                                                                                     \
  {                                                                                                          \
    if (GCR_FILL_ON)                                                                                         \
    for (int sgi = 0; sgi < a.fill.nseg; sgi++)                                                              \
      gcr_fill_zero_segment(a.fill.ptr[sgi], a.fill.n[sgi], (int)(slices_done * G + wg), (int)(G * S), tid); \
    slices_done++;                                                                                           \
  }
#ifdef GCR_EXPERIMENTS
  bool clk_on = a.clock_buf != nullptr;
  unsigned long long clk[GCR_K7_NCLK] = {};
  uint32_t clk_item = 0;
#endif

  if (wg < nwork) {
    const float4* __restrict__ ckpt = reinterpret_cast<const float4*>(a.binning_base + a.frame_in[GCR_FRAME_CKPT_OFF]);
    const uint4* __restrict__ work = reinterpret_cast<const uint4*>(a.binning_base + a.frame_in[GCR_FRAME_WORK_OFF]);
    const int lane = tid & 63, w = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const int acc_slot = (tid & 15) <= 8 ? (tid & 15) : -1;  // which of the 9 terms this lane flushes
    const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
    const float ddelx_dx = (float)(0.5 * a.W), ddely_dy = (float)(0.5 * a.H);
    const char* const sEb = reinterpret_cast<const char*>(sE);
    char* const acc_base = reinterpret_cast<char*>(&sAcc[acc_slot >= 0 ? acc_slot : 0]);
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (tid == 0) {
      sE[CHUNK].a = zero4;
      sE[CHUNK].b = zero4;
      sE[CHUNK].c = make_float4(0.0f, __builtin_inff(), __uint_as_float(NO_ENTRY), 0.0f);
    }

    // first item: nothing to hide its loads behind
    GCR_K7_CLK(0)
    BwdItem cur = bwd_item(work[wg], piece_P);
    BwdEntry ent = bwd_gather(a, cur, bwd_load_id(a, cur, tid), tid);
    BwdPixel px = bwd_load_pixel(a, cur, tid);
    float4 ck = zero4, cf = zero4;
    if (cur.has_ckpt) {
      ck = ckpt[(size_t)cur.slot_ck * 256u + tid];
      cf = ckpt[(size_t)cur.slot_final * 256u + tid];
    }

    for (uint32_t it = wg;;) {
      const bool has_next = it + G < nwork;
      const BwdItem nxt = bwd_item(work[has_next ? it + G : it], piece_P);  // scalar load, used before the walk
      GCR_K7_CLK(1)
      const int tx = cur.tile % a.gx, ty = cur.tile / a.gx;
      const LaneGeom g = lane_geom(tid, tx, ty);
      const float pixx = (float)g.pxi, pixy = (float)g.pyi;
      const float tile_x0 = (float)(tx * GCR_TILE_X), tile_y0 = (float)(ty * GCR_TILE_Y);
      const int n = (int)cur.n;
      const uint32_t hi = cur.lo + cur.n;  // this item walks list entries [lo, hi), back to front

      // ---- per-pixel state
      const float T_final = px.T_final;
      const uint32_t last_contributor = px.last_contributor;
      const float dLp0 = px.d0, dLp1 = px.d1, dLp2 = px.d2;
      float bg_dot_dpixel = 0;
      bg_dot_dpixel += bg0 * dLp0;
      bg_dot_dpixel += bg1 * dLp1;
      bg_dot_dpixel += bg2 * dLp2;
      // entries any pixel of this row (4x4 block) consumed
      uint32_t row_max = last_contributor;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        const uint32_t t = __shfl_xor(row_max, o, 64);
        row_max = t > row_max ? t : row_max;
      }
      uint32_t rmax[4];
#pragma unroll
      for (int rr = 0; rr < 4; rr++) rmax[rr] = __shfl(row_max, rr * 16, 64);
      // Start state of the reverse walk at entry hi - 1.  A pixel whose last contributor lies in this piece (or before
      // it) starts as upstream: T_final, accum_rec = 0.  A pixel that goes on behind the piece starts from the
      // forward's checkpoint at boundary hi: T after entry hi - 1, and accum_rec = what is blended behind it,
      // normalised: (C_final - C_prefix(hi)) / T(hi) -- which the first update
      // `a = last_alpha * lc + (1 - last_alpha) * acc` with last_alpha = 0 hands to the first contributing entry
      // unchanged.
      float T = T_final;
      float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;  // accum_rec
      if (cur.has_ckpt && last_contributor > hi) {
        T = ck.x;
        acc0 = (cf.y - ck.y) / ck.x;
        acc1 = (cf.z - ck.z) / ck.x;
        acc2 = (cf.w - ck.w) / ck.x;
      }
      const float neg_T_final = -T_final;
      float last_alpha = 0.0f, lc0 = 0.0f, lc1 = 0.0f, lc2 = 0.0f;

      // ---- stage the prefetched entry
      uint32_t my_mask = 0;
      if (tid < n) {
        const uint32_t entry_l = hi - 1u - (uint32_t)tid;  // == `contributor` upstream
        const float pmin = gcr_alpha_skip_bound(ent.q1.y);
        sE[tid].a = ent.q0;
        sE[tid].b = ent.q1;
        sE[tid].c = make_float4(ent.q2.x, pmin, __uint_as_float(entry_l), __uint_as_float((uint32_t)tid * 36u));
        sId[tid] = ent.id;
        my_mask = gcr_block_mask(ent.q0.x, ent.q0.y, ent.q0.z, ent.q0.w, ent.q1.x, pmin, tile_x0, tile_y0);
      }
      sMask[tid] = my_mask;
#pragma unroll
      for (int k = 0; k < 9; k++) sAcc[k * CHUNK + tid] = 0.0f;
      GCR_K7_CLK(2)  // own entry staged
      __syncthreads();
      GCR_K7_CLK(3)  // everybody's entries staged

      // compact the slots each row still consumes: block bit set and list entry < the row's max n_contrib
      int cnt[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (k * 64 < n) {  // wave-uniform
          const int jj = k * 64 + lane;
          const uint32_t m = sMask[jj];  // 0 for jj >= n
          const uint32_t entry_l = hi - 1u - (uint32_t)jj;
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const bool rel = ((m >> g.bit[rr]) & 1u) && entry_l < rmax[rr];
            const uint64_t bal = __ballot(rel);
            if (rel) sList[w][rr][cnt[rr] + __popcll(bal & lt_mask)] = (uint16_t)(jj * (int)ENTRY_BYTES);
            cnt[rr] += __popcll(bal);
          }
        }
      }
      const int maxcnt = max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3]));
#pragma unroll
      for (int rr = 0; rr < 4; rr++)
        for (int s = cnt[rr] + lane; s < maxcnt + 3; s += 64) sList[w][rr][s] = (uint16_t)SENT_OFF;
      __builtin_amdgcn_wave_barrier();

      // next item, level 1: its list entries (the records are gathered from inside the walk, once these have arrived)
      const uint32_t next_id = has_next ? bwd_load_id(a, nxt, tid) : 0u;
      BwdEntry next_ent;
      next_ent.q0 = next_ent.q1 = next_ent.q2 = zero4;
      next_ent.id = 0u;

// One list entry against this lane's pixel (cr/backward.cu:505-580).  Branch-free body: lanes
// that upstream would `continue` keep their state via selects and contribute exact zeros to the
// row reduction (see K6: scalar-unit pressure).  `power` is forced to 0 on those lanes so that
// every intermediate stays finite and the three masked factors (dchannel_dcolor, dL_dalpha) zero
// all nine terms.  The two quotients share the divisor 1-alpha: one v_rcp_f32 + one Newton step
// (<= 1 ulp) instead of two IEEE division expansions -- the only place the HIP path leaves
// gcr-fp32-v2; K7's sums are order-dependent (atomics) and tolerance-checked anyway.
#define GCR_BWD_STEP(QA, QB, QC)                                                               \
  {                                                                                            \
    const float dx = QA.x - pixx, dy = QA.y - pixy;                                            \
    const float power_raw = gcr_power(QA.z, QA.w, QB.x, dx, dy);                               \
    const bool in_range = __float_as_uint(QC.z) < last_contributor && !(power_raw > 0.0f) &&   \
                          !(power_raw < QC.y);                                                 \
    if (__ballot(in_range) != 0ull) { /* else the whole wave skips this step */                \
      const float power = in_range ? power_raw : 0.0f;                                         \
      const float G = blend_exp<FAST_EXP>(power);                                              \
      const float alpha = __builtin_fminf(0.99f, QB.y * G);                                    \
      const bool use = in_range && !(alpha < 1.0f / 255.0f);                                   \
      const float om = 1.f - alpha;                                                            \
      const float rc0 = __builtin_amdgcn_rcpf(om);                                             \
      const float rcp = __builtin_fmaf(rc0, __builtin_fmaf(-om, rc0, 1.0f), rc0);              \
      const float Tn = T * rcp;                                                                \
      const float dchannel_dcolor = use ? alpha * Tn : 0.0f;                                   \
      const float a0 = __builtin_fmaf(last_alpha, lc0, (1.f - last_alpha) * acc0);            \
      const float a1 = __builtin_fmaf(last_alpha, lc1, (1.f - last_alpha) * acc1);            \
      const float a2 = __builtin_fmaf(last_alpha, lc2, (1.f - last_alpha) * acc2);            \
      float dL_dalpha = 0.0f;                                                                  \
      dL_dalpha = __builtin_fmaf(QB.z - a0, dLp0, dL_dalpha);                                  \
      dL_dalpha = __builtin_fmaf(QB.w - a1, dLp1, dL_dalpha);                                  \
      dL_dalpha = __builtin_fmaf(QC.x - a2, dLp2, dL_dalpha);                                  \
      dL_dalpha *= Tn;                                                                         \
      dL_dalpha += (neg_T_final * rcp) * bg_dot_dpixel;                                        \
      dL_dalpha = use ? dL_dalpha : 0.0f;                                                      \
      const float dL_dG = QB.y * dL_dalpha;                                                    \
      const float gdx = G * dx, gdy = G * dy;                                                  \
      const float dG_ddelx = -gdx * QA.z - gdy * QA.w;                                         \
      const float dG_ddely = -gdy * QB.x - gdx * QA.w;                                         \
      float v[9];                                                                              \
      v[0] = dchannel_dcolor * dLp0;                                                           \
      v[1] = dchannel_dcolor * dLp1;                                                           \
      v[2] = dchannel_dcolor * dLp2;                                                           \
      v[3] = dL_dG * dG_ddelx * ddelx_dx;                                                      \
      v[4] = dL_dG * dG_ddely * ddely_dy;                                                      \
      v[5] = -0.5f * gdx * dx * dL_dG;                                                         \
      v[6] = -0.5f * gdx * dy * dL_dG;                                                         \
      v[7] = -0.5f * gdy * dy * dL_dG;                                                         \
      v[8] = G * dL_dalpha;                                                                    \
      T = use ? Tn : T;                                                                        \
      acc0 = use ? a0 : acc0;                                                                  \
      acc1 = use ? a1 : acc1;                                                                  \
      acc2 = use ? a2 : acc2;                                                                  \
      lc0 = use ? QB.z : lc0;                                                                  \
      lc1 = use ? QB.w : lc1;                                                                  \
      lc2 = use ? QC.x : lc2;                                                                  \
      last_alpha = use ? alpha : last_alpha;                                                   \
      /* reduce-scatter over each 16-lane row, then ONE ds_add_f32: lanes 0..8 of every row */ \
      /* add their row's sum of term (lane & 15) into the column of the row's entry */         \
      const float rsum = gcr_row_reduce_scatter9(v, lane);                                     \
      if (acc_slot >= 0 && GCR_LDS_ADD_ON) atomicAdd(reinterpret_cast<float*>(acc_base + __float_as_uint(QC.w)), rsum); \
    }                                                                                          \
  }
#define GCR_BWD_LOAD(QA, QB, QC, OFF)                              \
  QA = *reinterpret_cast<const float4*>(sEb + (OFF));             \
  QB = *reinterpret_cast<const float4*>(sEb + (OFF) + 16);         \
  QC = *reinterpret_cast<const float4*>(sEb + (OFF) + 32);


      GCR_K7_CLK(4)  // lists compacted
      // software pipeline, two entries per trip (see K6)
      const uint16_t* lp = &sList[w][g.row][0];
      float4 qa0, qb0, qc0, qa1, qb1, qc1;
      uint32_t e0 = lp[0], e1 = lp[1];
      GCR_BWD_LOAD(qa0, qb0, qc0, e0)
      for (int i = 0; i < maxcnt; i += 2, lp += 2) {
        if (i == 2 && has_next) next_ent = bwd_gather(a, nxt, next_id, tid);  // next item, level 2
        GCR_BWD_LOAD(qa1, qb1, qc1, e1)
        e0 = lp[2];
        GCR_BWD_STEP(qa0, qb0, qc0)
        if (i + 1 >= maxcnt) break;
        GCR_BWD_LOAD(qa0, qb0, qc0, e0)
        e1 = lp[3];
        GCR_BWD_STEP(qa1, qb1, qc1)
      }
      if (maxcnt <= 2 && has_next) next_ent = bwd_gather(a, nxt, next_id, tid);
#undef GCR_BWD_STEP
#undef GCR_BWD_LOAD
      GCR_K7_CLK(5)  // walk done
      __syncthreads();
      GCR_K7_CLK(6)  // every wave's walk done
      // next item, level 3: pixel state and checkpoints, in flight behind the flush
      BwdPixel next_px = {0.0f, 0u, 0.0f, 0.0f, 0.0f};
      float4 next_ck = zero4, next_cf = zero4;
      if (has_next) {
        next_px = bwd_load_pixel(a, nxt, tid);
        if (nxt.has_ckpt) {
          next_ck = ckpt[(size_t)nxt.slot_ck * 256u + tid];
          next_cf = ckpt[(size_t)nxt.slot_final * 256u + tid];
        }
      }
#ifdef GCR_EXPERIMENTS
      if (!(a.debug_flags & 1))
#endif
      {
        // lane = (entry within a group of 16) * 16 + component: the nine sums of an entry go out from nine
        // adjacent lanes into one 64-byte record
        const int comp = tid & 15;
        const int rec_idx = comp < 3 ? comp : (comp == 8 ? 3 : comp + 1);  // record layout: gcr_internal.h
        if (comp < 9) {
          for (int e = tid >> 4; e < n; e += 16) {
            const float v = sAcc[e * 9 + comp];
            if (v != 0.0f) atomicAdd(&a.grad_rec[(size_t)sId[e] * GCR_GRAD_REC_FLOATS + rec_idx], v);
          }
        }
      }
      GCR_FILL_SLICE()
#ifdef GCR_EXPERIMENTS
      if (clk_on) {
        clk[7] = __builtin_readcyclecounter();  // flush and fill slice issued
        if (lane == 0) {
          unsigned long long* o =
              a.clock_buf + (((size_t)wg * 4u + (size_t)w) * GCR_K7_ITEMS + clk_item) * (GCR_K7_NCLK + 2);
          o[0] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) |         // HW_REG_HW_ID
                 ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);  // HW_REG_XCC_ID
          o[1] = ((unsigned long long)maxcnt << 32) | (unsigned long long)n;
          for (int i = 0; i < GCR_K7_NCLK; i++) o[2 + i] = clk[i];
        }
        clk[0] = clk[7];
        if (++clk_item >= GCR_K7_ITEMS) clk_on = false;
      }
#endif
      if (!has_next) break;
      __syncthreads();  // the flush has read sAcc / sId: the next item may be staged
      it += G;
      cur = nxt;
      ent = next_ent;
      px = next_px;
      ck = next_ck;
      cf = next_cf;
    }
  }
  while (slices_done < S) GCR_FILL_SLICE()
#undef GCR_FILL_SLICE
}

}  // namespace

hipError_t gcr_launch_blend_fwd(const GcrBlendArgs& a, bool fast_exp, bool sort_in_kernel, hipStream_t s) {
  const int T = a.gx * a.gy;
  if (T <= 0) return hipSuccess;
  if (sort_in_kernel) {
    if (fast_exp)
      k_blend_fwd<true, true><<<T, 256, 0, s>>>(a);
    else
      k_blend_fwd<false, true><<<T, 256, 0, s>>>(a);
  } else {
    if (fast_exp)
      k_blend_fwd<true, false><<<T, 256, 0, s>>>(a);
    else
      k_blend_fwd<false, false><<<T, 256, 0, s>>>(a);
  }
  return hipGetLastError();
}

// Persistent grid of the backward blend: as many workgroups as the device keeps resident (queried once).
static int gcr_blend_bwd_resident(bool fast_exp) {
  static int cached[2] = {0, 0};
  int& c = cached[fast_exp ? 1 : 0];
  if (c == 0) {
    int dev = 0, cus = 256, per_cu = 4;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    }
    const hipError_t e = fast_exp ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_blend_bwd<true>, 256, 0)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_blend_bwd<false>, 256, 0);
    if (e != hipSuccess || per_cu < 1) per_cu = 4;
    c = per_cu * cus;
    if (const char* env = getenv("GCR_K7_BLOCKS")) c = atoi(env) > 0 ? atoi(env) : c;  // experiments only
  }
  return c;
}

hipError_t gcr_launch_blend_bwd(const GcrBlendArgs& a, bool fast_exp, hipStream_t s) {
  const int T = a.gx * a.gy;
  if (T <= 0) return hipSuccess;
  const unsigned int grid = (unsigned int)gcr_blend_bwd_resident(fast_exp);
  if (fast_exp)
    k_blend_bwd<true><<<grid, 256, 0, s>>>(a);
  else
    k_blend_bwd<false><<<grid, 256, 0, s>>>(a);
  return hipGetLastError();
}
