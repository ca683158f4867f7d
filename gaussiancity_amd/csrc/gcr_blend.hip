// gcr_blend.hip -- K6 forward alpha compositing and K7 reverse-walk gradient for gfx950.
//
// K6 (k_blend_fwd): one 256-thread workgroup (4 x wave64) per 16x16 tile.  Wave w owns the 8x8-pixel quadrant
// (w&1, w>>1) of the tile, cut into EIGHT sub-rows of 8 lanes, each a block of 2 (wide) x 4 (high) pixels with a list
// of its own (lane_geom6 below).  A tile's depth-sorted list is consumed in chunks of <= 223 entries: one thread per
// entry gathers the 48-byte Gaussian record (3 x dwordx4 from <= 2 cache lines) and stages it in LDS, so the blend loop
// never touches global memory.  Differences from cr/forward.cu:238-346 that do not change results:
//   * colour is staged in LDS too (the reference gathers it per contributing pixel, :328);
//   * a per-Gaussian conservative bound pmin = -ln(255*opacity) - 1e-3 is staged; pixels with power < pmin are skipped
//     before exp() -- exactly pixels the alpha < 1/255 test (:318) would skip anyway;
//   * BLOCK CULLING: the staging thread computes which of the tile's thirty-two 2x4 blocks the ellipse {power >= pmin}
//     can reach (gcr_cull.h, gcr_block_mask_2x4: exact per 4-pixel band, conservative, checked by brute force on the
//     host).  Every wave compacts the chunk into EIGHT lists of byte slots, one per sub-row, and each sub-row walks only
//     its own block's entries: the eight sub-rows of a wave work on different Gaussians in the same instruction (per-lane
//     LDS addresses, identical inside a sub-row).  An entry missing from a block's list has power < pmin for its 8
//     pixels, i.e. it is skipped by those pixels upstream as well, and skipped entries never change a pixel's state -- so
//     n_contrib / final_T / colour are unchanged;
//   * `contributor` is recovered from the list position instead of a per-lane counter;
//   * a wave stops as soon as its own 64 pixels are done, on top of the block vote (:284-286).
// K7: two kernels walk the same per-(tile, piece) work items the training-type K6 leaves behind -- k_blend_bwd_item (one
// workgroup per item: the records of a piece are gathered ONCE for the tile's four quadrant waves and every entry leaves
// as ONE record update) and k_blend_bwd (one independent wave per (item, quadrant); the deterministic mode's kernel).
// Its rows are 16-lane DPP rows = 4x4-pixel blocks (lane_geom below).
// Arithmetic: gcr-fp32-v2 (gcr_device.h) -> out_color / final_T / n_contrib are bit-identical to the oracle.
#include "gcr_cull.h"
#include "gcr_device.h"
#include "gcr_internal.h"
#include "gcr_sort.h"

namespace {

constexpr int CHUNK = 256;
constexpr uint32_t ENTRY_BYTES = 48;
// K6 (round 4): eight sub-row lists of BYTES per wave -- a list element is the slot of a staged entry, 0..222, and
// slot 223 is the sentinel, so a chunk holds at most 223 entries (GCR_PIECE_MAX).  223, not 255: with 224 record slots
// and 232-byte lists the workgroup's LDS is 19 456 bytes, EIGHT workgroups per CU instead of seven -- and the kernel
// (which needs 64 VGPRs for that, and gets them: the walk was never what held 72) is as long as its occupancy lets it be
// short: 129 / 108 / 103 us at 5 / 6 / 7 workgroups per CU (profiles/r04_k6_occupancy_cap_experiment.jsonl).
constexpr int K6_SENT_SLOT = 223;
constexpr int K6_LIST_STRIDE = 232;          // bytes per sub-row list: 223 entries + 3 pipeline pads, 8-aligned
static_assert(GCR_PIECE_MAX <= K6_SENT_SLOT, "a K6 chunk must leave slot 255 to the sentinel");
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

// the blend kernels' exponential: gcr-fp32-v2 (gcr_device.h), reproduced bit for bit by the oracle.  (A v_exp_f32
// "fast_exp" mode existed until round 4; it was slower than this one on the eight-workgroup kernel and is gone.)
GCR_DEV float blend_exp(float x) { return gcr_expf_noguard(x); }

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

// K6's lanes (round 4).  Wave w owns quadrant w like K7's, but cut into EIGHT sub-rows of 8 lanes, each a block of 2
// (wide) x 4 (high) pixels: sub-row s = columns 2(s&3), 2(s&3)+1 and rows 4(s>>2) .. 4(s>>2)+3 of the quadrant, and its
// bit in gcr_block_mask_2x4 is (2(w>>1) + (s>>2))*8 + 4(w&1) + (s&3).  A sub-row walks the entries that reach ITS block:
// with eight lists instead of four the wave's longest list is 7 % (C3) to 12 % (C2) shorter
// (profiles/r04_row_balance_sim.jsonl: the same lists dealt to the waves by length instead gain 0.2 %).
struct LaneGeom6 {
  int lane, w, sub, pxi, pyi;
};
GCR_DEV LaneGeom6 lane_geom6(int tid, int tx, int ty) {
  LaneGeom6 g;
  g.lane = tid & 63;
  g.w = tid >> 6;
  g.sub = g.lane >> 3;
  const int li = g.lane & 7;
  g.pxi = tx * GCR_TILE_X + (g.w & 1) * 8 + (g.sub & 3) * 2 + (li & 1);
  g.pyi = ty * GCR_TILE_Y + (g.w >> 1) * 8 + (g.sub >> 2) * 4 + (li >> 1);
  return g;
}
// the thread of K7's geometry (lane_geom) that owns the same pixel: the checkpoints are stored for the backward
GCR_DEV uint32_t lane6_to_k7_thread(uint32_t tid) {
  const uint32_t lane = tid & 63u, sub = lane >> 3, li = lane & 7u;
  const uint32_t row4 = (sub >> 2) * 2u + ((sub & 3u) >> 1);
  const uint32_t li4 = (li >> 1) * 4u + (sub & 1u) * 2u + (li & 1u);
  return (tid & ~63u) + row4 * 16u + li4;
}

// Output window (gcr_camera.win_*): where pixel (px, py) of the frame lands in out_color / dL_dpix, or -1.  The
// window is given in image coordinates AFTER the optional mirroring.
GCR_DEV long long gcr_out_index(const GcrBlendArgs& a, int px, int py, size_t* plane) {
  const int ox = a.flip_x ? a.W - 1 - px : px, oy = a.flip_y ? a.H - 1 - py : py;
  if (a.win_w == 0) {
    *plane = (size_t)a.H * a.W;
    return (long long)a.W * oy + ox;
  }
  *plane = (size_t)a.win_h * a.win_w;
  const int wx = ox - a.win_x, wy = oy - a.win_y;
  return (wx >= 0 && wx < a.win_w && wy >= 0 && wy < a.win_h) ? (long long)a.win_w * wy + wx : -1ll;
}
// does tile (tx, ty) own a pixel of the window?
GCR_DEV bool gcr_tile_in_window(const GcrBlendArgs& a, int tx, int ty) {
  if (a.win_w == 0) return true;
  int x0 = tx * GCR_TILE_X, x1 = min(a.W, x0 + GCR_TILE_X) - 1, y0 = ty * GCR_TILE_Y, y1 = min(a.H, y0 + GCR_TILE_Y) - 1;
  if (a.flip_x) { const int t = a.W - 1 - x1; x1 = a.W - 1 - x0; x0 = t; }
  if (a.flip_y) { const int t = a.H - 1 - y1; y1 = a.H - 1 - y0; y0 = t; }
  return x1 >= a.win_x && x0 < a.win_x + a.win_w && y1 >= a.win_y && y0 < a.win_y + a.win_h;
}

// scripts/inference.py:655-667 / utils/helpers.tensor_to_image: ((clamp(c, -1, 1) / 2 + 0.5) * 255).to(uint8), each step a
// float32 operation of its own (torch runs them as separate kernels; the library is built without contraction)
GCR_DEV uint8_t gcr_video_byte(float c) {
  float t = c < -1.0f ? -1.0f : (c > 1.0f ? 1.0f : c);  // torch.clamp: a NaN stays a NaN (and converts to 0 below)
  t = t / 2.0f;
  t = t + 0.5f;
  t = t * 255.0f;
  return (uint8_t)(t == t ? (int)t : 0);
}

// byte offset -> chunk slot (offsets are multiples of 48 below 2^14: 1366/65536 is 1/48 rounded up,
// exact for slots < 2048)
GCR_DEV uint32_t slot_of_offset(uint32_t off) { return (off * 1366u) >> 16; }

// ------------------------------------------------------------------------------------- K6
// What bounds this loop is VALU issue, priced per kind (tools/valu_probe.hip, profiles/r04_valu_probe.jsonl): fp32 add /
// mul / fma and integer add / and issue in ~2.5 cycles per wave64; compares, selects, min / max, shifts, v_lshl_add_u32,
// v_mul_u32_u24, DPP operations and SGPR operands in ~4.2; v_pk_*_f32 cost as much as the two plain ones they replace.
// Hence
//   * per-sub-row lists (above) -- fewer steps;
//   * each step is branch-free, 40 VALU instructions (27 full-rate, 13 half-rate), NO wave-uniform skip:
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
//         NON-FINITE colour poisons every pixel of the 2x4 blocks whose list holds the Gaussian (the conservative
//         block mask; upstream: only pixels it contributes to), and an accumulator that is exactly -0.0 (needs a
//         colour below 1e-40) may become +0.0.
//   * SORT (template): the workgroup sorts its own tile first.  GaussianCity's scenes have ~150 entries per tile, and
//     sorting 150 keys is a microsecond of work for the workgroup that is about to gather them anyway -- as a
//     separate kernel (K4) it is a latency-bound launch of 18 us on the frame's critical path.  Lists of up to one
//     chunk (223 keys) are rank-sorted in LDS (every thread counts the keys below its own through wave-uniform 16-byte
//     reads; unique keys => rank = position); a longer list -- only possible when the caller's length hint was
//     stale, the host then switches to the K4 path -- is ranked the same way through global loads: slow, correct.
//     The sorted indices are also written to `list_out` for the backward.
// The step's wave-uniform skip ("no lane of the wave is in range of its sub-row's entry") is compiled OUT: with eight
// sub-rows on eight different entries it almost never fires, the body is exact for a wave without a lane in range (every use
// is behind a select), and without the branch the two steps of a trip are one basic block: round 4 A/B at C3, same box, blend
// alone 106.4 -> 104.0 us, 5 125-5 134 -> 5 161-5 192 frames/s (profiles/r04_micro_ab.jsonl).  1 = the branch (A/B builds).
#ifndef GCR_K6_SKIP_BRANCH
#define GCR_K6_SKIP_BRANCH 0
#endif
#if GCR_K6_SKIP_BRANCH
#define GCR_K6_WAVE_SKIP(x) (__ballot(x) != 0ull)
#else
#define GCR_K6_WAVE_SKIP(x) true
#endif
// K7: a row standing on the sentinel entry (its list is shorter than the wave's longest) has nothing to add: without its
// nine lanes in the LDS add the launch was 89.2 instead of 94.0 us at C2 (same A/B file).  1 = add anyway (A/B builds).
#ifndef GCR_K7_ADD_SENTINEL
#define GCR_K7_ADD_SENTINEL 0
#endif
// (lanes whose row sum is exactly zero left out of the add as well: 90.3 vs 89.4 us, no gain, not kept)
#ifdef GCR_EXPERIMENTS  /* knock-outs for timing: bit 0 = no step arithmetic, bit 1 = no chunk loop at all */
#define GCR_K6_STEP_ON !(a.debug_flags & 1)
#define GCR_K6_LOOP_ON !(a.debug_flags & 2)
#else
#define GCR_K6_STEP_ON true
#define GCR_K6_LOOP_ON true
#endif
// K6's call of gcr_lazy_extend(): a real call, so that the rare sort takes no part in the register allocation of the
// blend loop.  The tile's state comes from and goes back to its uint4 in global memory (no pointers to locals: no stack);
// it is read and written with device-scope atomics, because a second call of the same workgroup must see what the
// first one stored, whatever the vector L1 still holds.  Returns the new n_sorted (the same value in every lane).
__device__ __noinline__ uint32_t k6_lazy_extend(uint64_t* s, const uint64_t* keys, uint32_t n, uint32_t* out, uint4* state) {
  const int tid = threadIdx.x;
  uint32_t* w = reinterpret_cast<uint32_t*>(state);
  uint32_t ns = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint64_t l = (uint64_t)__hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) |
               ((uint64_t)__hip_atomic_load(w + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 32);
  gcr_lazy_extend<GCR_LAZY_CAP_K6>(s, keys, n, ns, l, out, tid);
  if (tid == 0) {
    __hip_atomic_store(w, ns, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 2, (uint32_t)l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 3, (uint32_t)(l >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return ns;
}

// THE FRAME GATE of an asynchronous frame (gcr_forward_async, include/gcr.h).  A frame that fitted its binning buffer
// never gets here.  A frame the scatter kernel vetoed (num_rendered larger than the caller's capacity guess) renders
// nothing on the caller's stream: the first thread of the forward blend -- the frame's LAST kernel -- tells the host
// (words[3] = seq) and keeps the kernel, and with it the stream, from finishing until the library's rescue thread has
// rendered the frame with an exactly sized buffer on a stream of its own (words[2] = seq): everything the caller
// enqueued behind the frame sees the finished image.
// The wait for a rescue to START is bounded (about two seconds); a gate that gives up says so in words[4] and the ticket
// resolves to an error instead of a hung device.  Once the rescue has started (words[7] = seq: from then on it writes
// the frame's buffers) the gate goes on waiting for it, sixteen times as long -- letting the stream go on would hand
// buffers that are still being written to whatever the caller enqueued behind the frame.  The two sides decide who was
// first the Dekker way: the gate stores words[4] and THEN reads words[7]; the rescue stores words[7] and THEN reads
// words[4] (a PCIe read does not pass the posted write in front of it, the host side fences): at least one of them
// sees the other.  A rescue that sees words[4] touches nothing and releases the gate with the failure word set.
__device__ __noinline__ void k6_frame_gate(unsigned long long* words, unsigned int seq, unsigned int max_polls) {
  gcr_store_to_host(words + 3, (unsigned long long)seq);
  for (unsigned int i = 0; i < max_polls; i++) {
    const unsigned long long v = __hip_atomic_load(words + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((unsigned int)v == seq) return;
    __builtin_amdgcn_s_sleep(127);
    __builtin_amdgcn_s_sleep(127);
  }
  gcr_store_to_host(words + 4, (unsigned long long)seq);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the store is acknowledged before the load below is issued
  const unsigned long long st = __hip_atomic_load(words + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if ((unsigned int)st != seq) return;  // nobody started: the stream goes on with an unrendered frame, the ticket says so
  // a rescue is writing the frame's buffers: hold the stream until it says it is done (or has failed) -- sixteen times
  // longer than the wait for its start (about half a minute by default); only a rescue that is stuck for that long (its
  // own stream starved behind this very kernel, a dead host thread) lets the stream go on: a device that hangs for good
  // would be the worse failure, and the ticket then never resolves to a success
  for (unsigned long long i = 0; i < 16ull * max_polls; i++) {
    const unsigned long long v = __hip_atomic_load(words + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((unsigned int)v == seq) return;
    __builtin_amdgcn_s_sleep(127);
    __builtin_amdgcn_s_sleep(127);
  }
}

// amdgpu_waves_per_eu(8, 8): the call above constrains the register assignment (what lives across it must sit in
// callee-saved registers) and the allocator, left alone, ends at 78 VGPRs = six waves per SIMD; told to fit eight (which
// the workgroup's LDS allows since the chunk is 223 entries), it finds a 64-register assignment without a spill and with
// the blend steps unchanged instruction for instruction (the walk uses 61; staging and list building held the rest).
//
// STATE (template): the backward's per-piece state -- checkpoints at the piece boundaries, work items, the block mask
// of every staged entry -- is written only by frames rendered with gcr_camera.backward == 1.  Every other frame (the
// headline inference workload) runs the instantiation without a single instruction of it; gcr_backward on such a
// frame regenerates the state with one more pass of this kernel (a.out_color == nullptr: no pixel is stored).
template <bool SORT, bool STATE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_blend_fwd(const GcrBlendArgs a) {
  __shared__ StagedEntry sE[K6_SENT_SLOT + 1];
  __shared__ uint32_t sMask[CHUNK];
  __shared__ __attribute__((aligned(16))) uint8_t sList[4][8][K6_LIST_STRIDE];  // [wave][sub-row]: slots of sE, list order

  if (a.frame != nullptr && a.frame[2] == 0ull) {  // speculative launch vetoed
    if (a.gate_words != nullptr && blockIdx.x == 0 && threadIdx.x == 0) k6_frame_gate(a.gate_words, a.gate_seq, a.gate_polls);
    return;
  }
  const int tile = blockIdx.x;
  const int tx = tile % a.gx, ty = tile / a.gx;
  const int tid = threadIdx.x;
  const float tile_x0 = (float)(tx * GCR_TILE_X), tile_y0 = (float)(ty * GCR_TILE_Y);
  const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
  if (!gcr_tile_in_window(a, tx, ty)) {
    // none of this tile's pixels is wanted (gcr_camera.win_*): nothing to blend, and no work for the backward either
    if (STATE && a.work != nullptr) {
      const uint32_t P0 = (uint32_t)a.piece, sb = r0 / P0 + (uint32_t)tile, ns = r1 / P0 + 1u - r0 / P0;
      for (uint32_t k = tid; k < ns; k += 256u) a.work[sb + k] = make_uint4(GCR_NO_TILE, 0u, 0u, 0u);
      if (tile == 0 && tid == 0) {
        a.frame_out[GCR_FRAME_PIECE] = P0;
        a.frame_out[GCR_FRAME_CKPT_OFF] = a.ckpt_off;
        a.frame_out[GCR_FRAME_WORK_OFF] = a.work_off;
        a.frame_out[GCR_FRAME_MASK_OFF] = a.mask_off;
        a.frame_out[GCR_FRAME_STAGED_OFF] = a.staged_off;
        a.frame_out[GCR_FRAME_CARVE] = a.carve_bytes;
      }
    }
    return;
  }
  const int total = (int)(r1 - r0);
  const char* const sEb = reinterpret_cast<const char*>(sE);
#ifdef GCR_EXPERIMENTS  // tools/k6_clocks.py: when every workgroup ran, and where
  if (a.clock_buf != nullptr && tid == 0) {
    unsigned long long* o = a.clock_buf + (size_t)tile * 16;
    o[0] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
    o[1] = __builtin_readcyclecounter();
    o[3] = (unsigned long long)total;  // (needs the ranges: the store waits for them)
    o[10] = __builtin_readcyclecounter();
  }
#endif
  if (tid == 0) {  // the sentinel: behind everything that borrows the record buffer (sort keys: 2 KB, lazy sort: 10 KB)
    sE[K6_SENT_SLOT].a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    sE[K6_SENT_SLOT].b = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    sE[K6_SENT_SLOT].c = make_float4(0.0f, __builtin_inff(), 0.0f, 0.0f);
  }
  static_assert(gcr_lazy_lds_keys(GCR_LAZY_CAP_K6) * 8 <= K6_SENT_SLOT * (int)ENTRY_BYTES, "the lazy sort's scratch must end below the sentinel");

  float Tw;
  {
    const LaneGeom6 g0 = lane_geom6(tid, tx, ty);
    Tw = (g0.pxi < a.W && g0.pyi < a.H) ? 1.0f : -1.0f;
  }
  float C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
  uint32_t last_contributor = 0;
  // pieces (gcr_internal.h "backward pieces"): the list is walked in equal pieces of <= a.piece entries
  const uint32_t piece_P = (uint32_t)a.piece;
  const int cs = (int)gcr_piece_size((uint32_t)total, piece_P);
  const uint32_t sbase = r0 / piece_P + (uint32_t)tile;
  uint32_t entered = 0;  // pieces this workgroup walked into (block-uniform)

  // lazily sorted list (gcr_sort.h): list[r0, r0 + lz_sorted) is in final order; the walk sorts on when it gets there
  uint32_t lz_sorted = (uint32_t)max(total, 0);
  if (!SORT && a.lazy != nullptr) lz_sorted = a.lazy[tile].x;

  uint32_t sorted_id = 0;  // SORT, total <= CHUNK: the Gaussian of list position `tid`
  if (SORT && total > 0) {
    const uint64_t* __restrict__ seg = a.pairs + r0;
    uint32_t* __restrict__ out = a.list_out + r0;
    if (total <= K6_SENT_SLOT) {
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
    if (GCR_K6_WAVE_SKIP(in_range) && GCR_K6_STEP_ON) { /* wave-uniform */                    \
      /* lanes outside [pmin, 0] may produce garbage; every use below is behind a select */  \
      const float araw = __builtin_fminf(0.99f, QB.y * blend_exp(power));          \
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

  // The walk.  OUTER loop: one trip per sorted segment of the list (gcr_sort.h "lazy tile sort": exactly one trip unless
  // the tile's list is longer than 1024 entries AND the pixels are still alive where the sorted part ends).  The
  // lane geometry is derived behind an opaque barrier INSIDE it, so that nothing but the per-pixel state (Tw, C,
  // last_contributor) is live in registers while the rare sort runs: with the geometry computed once above the loop the
  // sort's registers came on top of everything the blend loop keeps (70 -> 86 VGPRs, five waves per SIMD instead of seven).
  // The sort itself is a real call (k6_lazy_extend) for the same reason.
  int base = 0;
  bool finished = false;  // block-uniform
  while (base < total && !finished && GCR_K6_LOOP_ON) {
  if (!SORT) {
    const uint32_t need = (uint32_t)(base + min(cs, total - base));
    while (lz_sorted < need) {  // sE is not live here
      const uint32_t ns = k6_lazy_extend(reinterpret_cast<uint64_t*>(sE), a.pairs + r0, (uint32_t)total,
                                         a.list_out + r0, a.lazy + tile);  // (list_out == list on this path)
      lz_sorted = (uint32_t)__builtin_amdgcn_readfirstlane((int)ns);
    }
  }
  int tid_op = threadIdx.x;
  asm volatile("" : "+v"(tid_op));
  const LaneGeom6 g = lane_geom6(tid_op, tx, ty);
  const int lane = g.lane;
  const float pixx = (float)g.pxi, pixy = (float)g.pyi;
  // wave-uniform by construction: this wave's lists and the place of its eight bits in a 2x4 block mask
  const int w_u = __builtin_amdgcn_readfirstlane(g.w);
  const int mask_sh = (w_u >> 1) * 16 + (w_u & 1) * 4;
  uint8_t* const lw = &sList[w_u][0][0];
  for (; base < total; base += cs) {
    // block-wide vote (cr/forward.cu:284-286); also fences the previous chunk's LDS reads
    if (__syncthreads_count(!(Tw > 0.0f)) == 256) {
      finished = true;
      break;
    }
    if (!SORT && lz_sorted < (uint32_t)(base + min(cs, total - base))) break;  // sort on, then come back
    // crossing a piece boundary: checkpoint of the per-pixel state for the backward (|Tw| = T; a finished pixel's
    // checkpoint is never read).  256 x 16 bytes, one coalesced 4 KB store per boundary.
    if (STATE && entered > 0u && a.ckpt != nullptr) {
      // (the thread index is re-read behind an opaque barrier so that the store's address arithmetic is done here and
      // does not sit in registers across the walk: 76 -> 71 VGPRs, seven waves per SIMD again)
      uint32_t t_op = threadIdx.x;
      asm volatile("" : "+v"(t_op));
      a.ckpt[(size_t)(sbase + entered - 1u) * 256u + lane6_to_k7_thread(t_op)] = make_float4(__builtin_fabsf(Tw), C0, C1, C2);
    }
    entered++;
    const int n = min(cs, total - base);
    uint32_t my_mask = 0;
    if (tid < n) {
      const uint32_t id = !SORT ? a.list[r0 + base + tid]
                                : ((total <= K6_SENT_SLOT && cs == total) ? sorted_id : a.list_out[r0 + base + tid]);
      const float4* __restrict__ rec = a.rec + (size_t)id * GCR_REC_QUADS;
#ifdef GCR_EXPERIMENTS  // hop clocks of thread 0's first entry: ids arrived (the address below needs them) / records arrived
      if (a.clock_buf != nullptr && tid == 0 && base == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        a.clock_buf[(size_t)tile * 16 + 8] = __builtin_readcyclecounter();
      }
#endif
      const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
#ifdef GCR_EXPERIMENTS
      if (a.clock_buf != nullptr && tid == 0 && base == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        a.clock_buf[(size_t)tile * 16 + 9] = __builtin_readcyclecounter();
      }
#endif
      const float pmin = gcr_alpha_skip_bound(q1.y);
      my_mask = gcr_block_mask_2x4(q0.x, q0.y, q0.z, q0.w, q1.x, pmin, tile_x0, tile_y0);
      const float4 sa = make_float4(q0.x, q0.y, -0.5f * q0.z, -q0.w), sb = make_float4(-0.5f * q1.x, q1.y, q1.z, q1.w);
      const float4 sc = make_float4(q2.x, pmin, 0.0f, 0.0f);
      sE[tid].a = sa;
      sE[tid].b = sb;
      sE[tid].c = sc;
      if (STATE && a.staged_out != nullptr) {  // the backward's records, in list order: one coalesced block per piece there
        float4* __restrict__ so = a.staged_out + (size_t)(r0 + (uint32_t)base + (uint32_t)tid) * 3;
        so[0] = sa;
        so[1] = sb;
        so[2] = sc;
      }
    }
    sMask[tid] = my_mask;
    if (STATE && tid < n && a.mask_out != nullptr)  // reused by the backward, whose rows are 4x4 blocks
      a.mask_out[r0 + base + tid] = (uint16_t)gcr_block_mask_4x4_of_2x4(my_mask);
    __syncthreads();
#ifdef GCR_EXPERIMENTS  // first chunk, wave 0: records staged / lists built / walk done
    if (a.clock_buf != nullptr && tid == 0 && base == 0) a.clock_buf[(size_t)tile * 16 + 4] = __builtin_readcyclecounter();
#endif
    if (__ballot(Tw > 0.0f) == 0ull) continue;  // this wave's quadrant is finished; keep voting
    // Every list starts as sentinels (shorter lists run on them up to the longest one, and the software pipeline below
    // reads three slots ahead): 8 x 232 bytes = 116 quads, stored before the entries below (the LDS keeps a wave's order)
    {
      constexpr int NQ = 8 * K6_LIST_STRIDE / 16;  // quads of this wave's eight lists
      static_assert(8 * K6_LIST_STRIDE % 16 == 0 && NQ > 64 && NQ <= 128, "list fill: two stores per lane");
      uint4* const f = reinterpret_cast<uint4*>(lw);
      const uint32_t sb = 0x01010101u * (uint32_t)K6_SENT_SLOT;
      const uint4 ff = make_uint4(sb, sb, sb, sb);
      f[lane] = ff;
      if (64 + lane < NQ) f[64 + lane] = ff;
    }
    // compact the chunk into this wave's eight sub-row lists (ascending list order is preserved)
    int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (k * 64 < n) {  // wave-uniform
        const int jj = k * 64 + lane;
        const uint32_t m = sMask[jj] >> mask_sh;  // bits 0..3: the upper band's four blocks, bits 8..11: the lower band's
#pragma unroll
        for (int s = 0; s < 8; s++) {
          uint32_t bit = m & (1u << (s < 4 ? s : s + 4));
          asm volatile("" : "+v"(bit));  // (one compare for the ballot and the store's predicate, not two)
          const bool rel = bit != 0u;
          const uint64_t bal = __ballot(rel);
          const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
          if (rel) lw[s * K6_LIST_STRIDE + cnt[s] + (int)before] = (uint8_t)jj;
          cnt[s] += __popcll(bal);
        }
      }
    }
    const int maxcnt = __builtin_amdgcn_readfirstlane(
        max(max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3])), max(max(cnt[4], cnt[5]), max(cnt[6], cnt[7]))));
    __builtin_amdgcn_wave_barrier();
#ifdef GCR_EXPERIMENTS
    if (a.clock_buf != nullptr && tid == 0 && base == 0) {
      a.clock_buf[(size_t)tile * 16 + 5] = __builtin_readcyclecounter();
      a.clock_buf[(size_t)tile * 16 + 7] = (unsigned long long)maxcnt;
    }
#endif
    // Software pipeline, two entries per trip with ping-pong registers: while entry i blends,
    // the record of entry i+1 and the list slot of entry i+2 are already in flight.
    const uint8_t* lp = &sList[w_u][g.sub][0];
    float4 qa0, qb0, qa1, qb1;
    float2 qc0, qc1;
    uint32_t e0 = (uint32_t)lp[0] * ENTRY_BYTES, e1 = (uint32_t)lp[1] * ENTRY_BYTES;
    uint32_t last_off = NO_ENTRY;
    GCR_BLEND_LOAD(qa0, qb0, qc0, e0)
    for (int i = 0; i < maxcnt; i += 2, lp += 2) {
      GCR_BLEND_LOAD(qa1, qb1, qc1, e1)
      const uint32_t cur0 = e0;
      e0 = (uint32_t)lp[2] * ENTRY_BYTES;
      GCR_BLEND_STEP(qa0, qb0, qc0, cur0)
      if (i + 1 >= maxcnt) break;
      GCR_BLEND_LOAD(qa0, qb0, qc0, e0)
      const uint32_t cur1 = e1;
      e1 = (uint32_t)lp[3] * ENTRY_BYTES;
      GCR_BLEND_STEP(qa1, qb1, qc1, cur1)
      if (__ballot(Tw > 0.0f) == 0ull) break;  // every pixel of the wave is done
    }
#ifdef GCR_EXPERIMENTS
    if (a.clock_buf != nullptr && tid == 0 && base == 0) a.clock_buf[(size_t)tile * 16 + 6] = __builtin_readcyclecounter();
#endif
    if (last_off != NO_ENTRY) last_contributor = (uint32_t)base + slot_of_offset(last_off) + 1u;
  }
  }
#undef GCR_BLEND_STEP
#undef GCR_BLEND_LOAD
  int tid_end = threadIdx.x;
  asm volatile("" : "+v"(tid_end));
  const LaneGeom6 g = lane_geom6(tid_end, tx, ty);
  if (g.pxi < a.W && g.pyi < a.H && (!STATE || a.out_color != nullptr)) {  // (state-only pass: pixels already stored)
    const float Tout = __builtin_fabsf(Tw);
    const size_t pix_id = (size_t)a.W * g.pyi + g.pxi;
    // a.nt_out (block-uniform; round 6): an inference frame's per-pixel outputs are not read again on the device -- stored
    // with the non-temporal policy they do not push the next frame's inputs out of the caches.  Set for callers that wait
    // for every frame's num_rendered (one frame alone 0.262 -> 0.253 ms with it); beside two other frames' kernels it
    // buys nothing (profiles/r06_cache_policy_ab.jsonl), and a training frame's backward reads all of it.
    const bool nt_out = !STATE && a.nt_out != 0;
    if (nt_out) {
      __builtin_nontemporal_store(Tout, a.final_T + pix_id);
      __builtin_nontemporal_store(last_contributor, a.n_contrib + pix_id);
    } else {
      a.final_T[pix_id] = Tout;
      a.n_contrib[pix_id] = last_contributor;
    }
    // the image may be stored mirrored (the wrapper's flip_lr / flip_ud without a copy kernel) and / or as a window of
    // the frame (the helpers' crop without slice kernels); the per-pixel state is neither
    size_t oplane;
    int px_op = g.pxi, py_op = g.pyi;  // opaque copies: the index arithmetic below must not be hoisted above the walk
    asm volatile("" : "+v"(px_op), "+v"(py_op));
    const long long out_id = gcr_out_index(a, px_op, py_op, &oplane);
    if (out_id >= 0) {
      const float c0 = C0 + Tout * GCR_CAM(a, bg, a.bg, 0);
      const float c1 = C1 + Tout * GCR_CAM(a, bg, a.bg, 1);
      const float c2 = C2 + Tout * GCR_CAM(a, bg, a.bg, 2);
      if (!STATE && a.out_u8) {  // the video frame directly (gcr_camera.out_u8): uint8 [H,W,3]
        uint8_t* o = reinterpret_cast<uint8_t*>(a.out_color) + 3 * (size_t)out_id;
        o[0] = gcr_video_byte(c0);
        o[1] = gcr_video_byte(c1);
        o[2] = gcr_video_byte(c2);
      } else if (nt_out) {
        __builtin_nontemporal_store(c0, a.out_color + out_id);
        __builtin_nontemporal_store(c1, a.out_color + oplane + out_id);
        __builtin_nontemporal_store(c2, a.out_color + 2 * oplane + out_id);
      } else {
        a.out_color[out_id] = c0;
        a.out_color[oplane + out_id] = c1;
        a.out_color[2 * oplane + out_id] = c2;
      }
    }
  }
#ifdef GCR_EXPERIMENTS
  if (a.clock_buf != nullptr && tid == 0) a.clock_buf[(size_t)tile * 16 + 2] = __builtin_readcyclecounter();
#endif
  if (STATE && a.work != nullptr) {
    // a tile that crossed a boundary also leaves its final colour (without background) in its last slot: a
    // backward piece that starts from a checkpoint needs C_final - C_prefix
    if (entered >= 2u)
      a.ckpt[(size_t)(sbase + gcr_piece_count((uint32_t)total, piece_P) - 1u) * 256u + lane6_to_k7_thread((uint32_t)tid)] =
          make_float4(__builtin_fabsf(Tw), C0, C1, C2);
    // work items of the backward blend, one per slot of [slot_base(tile), slot_base(tile + 1)): the pieces this
    // workgroup walked into carry {tile, list start, list length, piece}; the others (behind a saturated tile's last
    // piece; the gap slot) carry nothing.  Fixed places: no atomic, no barrier at the end of the forward.
    const uint32_t nslots = r1 / piece_P + 1u - r0 / piece_P;
    for (uint32_t k = tid; k < nslots; k += 256u)
      a.work[sbase + k] = k < entered ? make_uint4((uint32_t)tile, r0, (uint32_t)total, k) : make_uint4(GCR_NO_TILE, 0u, 0u, 0u);
  }
  if (STATE && a.work != nullptr && tile == 0 && tid == 0) {
    a.frame_out[GCR_FRAME_PIECE] = piece_P;
    a.frame_out[GCR_FRAME_CKPT_OFF] = a.ckpt_off;
    a.frame_out[GCR_FRAME_WORK_OFF] = a.work_off;
    a.frame_out[GCR_FRAME_MASK_OFF] = a.mask_off;
    a.frame_out[GCR_FRAME_STAGED_OFF] = a.staged_off;
    a.frame_out[GCR_FRAME_CARVE] = a.carve_bytes;
  }
}

// ------------------------------------------------------------------------------------- K7
// cr/backward.cu:428-581.  Per pixel the reverse walk is the reference's.  What changes, in both kernels below:
//
// (a) How the nine per-(pixel, Gaussian) gradient terms reach memory.  The reference issues nine global float
//     atomics per pixel per Gaussian; here
//   1. each 16-lane DPP row (= one 4x4 block, all lanes on the SAME Gaussian) reduce-scatters the nine terms
//      (20 VALU instructions, gcr_row_reduce_scatter9_b; no LDS traffic),
//   2. nine lanes of each row add the row sums into an LDS accumulator of DOUBLES (one ds_add_f64: "LDS FLOAT ATOMICS"
//      below says why not floats),
//   3. the nine sums of an entry are flushed by nine ADJACENT lanes into the Gaussian's 64-byte gradient record
//      (gcr_internal.h): one wave instruction = 4 entries = 4 cache lines.
//
// (b) The unit of work is a (tile, PIECE) item of the list the forward left behind (gcr_internal.h "backward
//     pieces"), not a tile: a pixel whose walk goes on behind the piece starts from the forward's checkpoint.  An
//     item's records come from the forward's staged copy in list order (gcr_layout.bin_staged), its block masks from the
//     forward as well: no mask arithmetic (222 VALU per entry) and no gather by Gaussian index in the backward.
//
// (c) k_blend_bwd (round 3): ONE WAVE PER (item, 8x8 QUADRANT), NO WORKGROUP BARRIERS -- sixteen independent pipelines per
//     CU, all state in wave-private LDS; an entry that reaches several quadrants of a tile is staged and flushed by each
//     of them.  Selected by option "bwd_wave_units" and always by the deterministic mode (its LDS sums have a fixed
//     order).  k_blend_bwd_item (round 5, the default, further down): one workgroup per item.
// Only entries below the row's max n_contrib are visited (skipped by every pixel upstream too,
// contributor >= last_contributor, :511-512), and each row visits only the entries whose block mask includes its
// 4x4 block (see K6).
// Experiment builds (make -C csrc experiments): per-wave phase clocks.  With gcr_debug_set_clock_buffer() armed, lane 0
// of every wave stores {hw id, xcc id, s_memtime at its phase boundaries} (tools/k7_clocks.py).  The shipping build
// contains none of it.
#ifdef GCR_EXPERIMENTS
#define GCR_K7_NCLK 8
#define GCR_K7_CLK(i) \
  if (clk_on) clk[i] = __builtin_readcyclecounter();
#define GCR_FLUSH_ON !(a.debug_flags & 1)
#define GCR_LDS_ADD_ON !(a.debug_flags & 4)
#define GCR_STEP_ON !(a.debug_flags & 16)   /* knock-out: the step's arithmetic (loads and loop stay) */
#define GCR_K7_IEEE_DIV (a.debug_flags & 64) /* attribution (tools/fuzz_f64.py): the step's two quotients as IEEE divisions, like the oracle */
#define GCR_PASSES_ON !(a.debug_flags & 32) /* knock-out: everything after the unit's prologue */
#else
#define GCR_K7_IEEE_DIV false
#define GCR_STEP_ON true
#define GCR_PASSES_ON true
#define GCR_K7_CLK(i)
#define GCR_FLUSH_ON true
#define GCR_LDS_ADD_ON true
#endif

constexpr int WPASS = 64;                          // list entries a wave stages per pass
constexpr int WLIST_STRIDE = WPASS + 8;            // u16 slots per row list: entries + pipeline pads
constexpr uint32_t WSENT_OFF = WPASS * ENTRY_BYTES;  // byte offset of the wave's sentinel entry

// One list entry against this lane's pixel (cr/backward.cu:505-580).  Branch-free body with THREE selects: what
// bounds this kernel is VALU issue at ~4 cycles per instruction for this mix (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU,
// profiles/r03_k7_pmc.txt: v_cndmask with an SGPR-pair mask, v_cmp and the DPP moves cost about twice an FMA), so
// the step avoids selects on the loop state altogether:
//   * a lane that upstream would `continue` gets power = 0 (every intermediate stays finite) and alpha_eff = 0;
//   * with alpha_eff = 0 the transmittance update is exact without a select: rcp(1) = 1, T * 1 = T;
//   * accum_rec is updated EAGERLY: (last_alpha, last_color, accum_rec) := (alpha_eff, colour, a) where
//     a = last_alpha * last_color + (1 - last_alpha) * accum_rec is what upstream computes at its next contributing
//     entry.  After a skipped entry the state reads (0, *, a), whose next blend 0 * c + 1 * a = a is exactly the value
//     the untouched state would have produced -- same bits, no selects;
//   * dchannel_dcolor = alpha_eff * T is zero by itself; only dL_dalpha needs masking (one select).
// The two quotients share the divisor 1 - alpha: one v_rcp_f32 + one Newton step (<= 1 ulp) instead of two IEEE
// division expansions -- the only place the HIP path leaves gcr-fp32-v2; K7's sums are order-dependent (atomics)
// and tolerance-checked anyway.
// GCR_K7_RS_ASM=0 builds the compiler-scheduled reduce-scatter (35 VALU instructions) for A/B runs
#ifndef GCR_K7_RS_ASM
#define GCR_K7_RS_ASM 1
#endif
#if GCR_K7_RS_ASM
#define GCR_K7_ROW_SLOT gcr_row_reduce_scatter9_b_slot(lane)
#define GCR_K7_ROW_SUMS                                                                                        \
  float rs_r;                                                                                                  \
  const float rs_8 = gcr_row_reduce_scatter9_b(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], rs_r);    \
  const float rsum = acc_slot == 8 ? rs_8 : rs_r;
#else
#define GCR_K7_ROW_SLOT ((lane & 15) <= 8 ? (lane & 15) : -1)
#define GCR_K7_ROW_SUMS const float rsum = gcr_row_reduce_scatter9(v, lane);
#endif
// ROUND 5 -- MOMENTS.  cr/backward.cu:540-575 turns dL/dalpha of a (pixel, Gaussian) pair into nine addends.  Six of them
// are the same number u = G * dL/dalpha times a polynomial in the pixel offset d = (dx, dy):
//     dL_dopacity += u                          dL_dconic.x += -0.5 o u dx dx        dL_dmean2D.x += -o (cx u dx + cy u dy) W/2
//                                               dL_dconic.y += -0.5 o u dx dy        dL_dmean2D.y += -o (cz u dy + cy u dx) H/2
//                                               dL_dconic.w += -0.5 o u dy dy        (o = opacity, c = conic, dL_dG = o dL/dalpha)
// so the walk accumulates the six MOMENTS  S = sum u, Sx = sum u dx, Sy = sum u dy, Sxx, Sxy, Syy  (6 multiplications per
// step instead of 20) and the per-Gaussian factors are applied ONCE per Gaussian by the preprocess gradient kernel,
// which reads the moments from the accumulation record (gcr_internal.h): [0..2] colour sums, [3] S, [4] Sx, [5] Sy,
// [6] Sxx, [7] Sxy, [8] Syy.  Sums of products are re-associated (within the 1e-4 gradient bar; tools/fuzz_parity.py).
// The background term's pixel constant -T_final * (bg . dL/dpixel) is formed once per pixel (`nbg`).  The conic the step
// reads is K6's pre-scaled one (-0.5 cx, -cy, -0.5 cz): `power` in 6 instructions and with the bits -- hence the alpha
// and skip decisions -- of the forward blend that rendered the frame.
#define GCR_BWD_STEP(QA, QB, QC)                                                               \
  {                                                                                            \
    const float dx = QA.x - pixx, dy = QA.y - pixy;                                            \
    const float power_raw = __builtin_fmaf(QA.w * dx, dy, __builtin_fmaf(QB.x * dy, dy, (QA.z * dx) * dx)); \
    const bool in_range = __float_as_uint(QC.z) < last_contributor && !(power_raw > 0.0f) &&   \
                          !(power_raw < QC.y);                                                 \
    if (GCR_K7_WAVE_SKIP(in_range) && GCR_STEP_ON) { /* else the whole wave skips this step */   \
      const float power = in_range ? power_raw : 0.0f;                                         \
      const float G = blend_exp(power);                                                        \
      const float alpha = __builtin_fminf(0.99f, QB.y * G);                                    \
      const bool use = in_range && !(alpha < 1.0f / 255.0f);                                   \
      const float a_eff = use ? alpha : 0.0f;                                                  \
      const float om = 1.f - a_eff;                                                            \
      const float rc0 = __builtin_amdgcn_rcpf(om);                                             \
      const float rcp = __builtin_fmaf(rc0, __builtin_fmaf(-om, rc0, 1.0f), rc0);              \
      T = GCR_K7_IEEE_DIV ? T / om : T * rcp;                                                  \
      const float dchannel_dcolor = a_eff * T;                                                 \
      const float oml = 1.f - last_alpha;                                                      \
      acc0 = __builtin_fmaf(last_alpha, lc0, oml * acc0);                                      \
      acc1 = __builtin_fmaf(last_alpha, lc1, oml * acc1);                                      \
      acc2 = __builtin_fmaf(last_alpha, lc2, oml * acc2);                                      \
      lc0 = QB.z;                                                                              \
      lc1 = QB.w;                                                                              \
      lc2 = QC.x;                                                                              \
      last_alpha = a_eff;                                                                      \
      float dL_dalpha = (lc0 - acc0) * dLp0;                                                   \
      dL_dalpha = __builtin_fmaf(lc1 - acc1, dLp1, dL_dalpha);                                 \
      dL_dalpha = __builtin_fmaf(lc2 - acc2, dLp2, dL_dalpha);                                 \
      dL_dalpha = GCR_K7_IEEE_DIV ? dL_dalpha * T + nbg / om : __builtin_fmaf(nbg, rcp, dL_dalpha * T); \
      dL_dalpha = use ? dL_dalpha : 0.0f;                                                      \
      const float u = G * dL_dalpha;                                                           \
      const float ux = u * dx, uy = u * dy;                                                    \
      float v[9];                                                                              \
      v[0] = dchannel_dcolor * dLp0;                                                           \
      v[1] = dchannel_dcolor * dLp1;                                                           \
      v[2] = dchannel_dcolor * dLp2;                                                           \
      v[3] = ux;                                                                               \
      v[4] = uy;                                                                               \
      v[5] = ux * dx;                                                                          \
      v[6] = ux * dy;                                                                          \
      v[7] = uy * dy;                                                                          \
      v[8] = u;                                                                                \
      /* reduce-scatter over each 16-lane row, then ONE ds_add_f64: nine lanes of every row    */ \
      /* add their row's sum of one term each into the column of the row's entry               */ \
      GCR_K7_ROW_SUMS                                                                          \
      if (acc_slot >= 0 && GCR_LDS_ADD_ON && (GCR_K7_ADD_SENTINEL || __float_as_uint(QC.z) != NO_ENTRY))                 \
        atomicAdd(reinterpret_cast<double*>(acc_base + __float_as_uint(QC.w)), (double)rsum);                             \
    }                                                                                          \
  }
// LDS FLOAT ATOMICS (round 5).  The row sums meet in DOUBLE accumulators: on gfx950 ds_add_f32 (and ds_pk_add_f16) is
// serialised lane by lane -- 193 cycles of the CU's LDS pipeline for a full wave64, 109 for this step's 36 lanes -- while
// ds_add_f64, ds_add_u32 / u64 and ds_max_f32 run at the rate of a plain store (13 cycles for the same 36 lanes):
// tools/lds_atomic_probe.hip, profiles/r05_lds_atomic_probe.jsonl.  The one ds_add_f32 per step was 65 us of LDS pipeline
// time per launch at C2 -- the knock-out that removed it shortened the launch by a quarter in rounds 3, 4 and 5, and nothing
// else ever had.  One v_cvt_f64_f32 per step and 4.6 KB more LDS per wave buy it back (and sums that are exact to the
// last float bit per item).
// The wave-uniform skip of a step no lane is in range of: K6 keeps it; here a step of a row list almost always has a lane
// in range (the list holds the entries whose block mask reaches the row's block, below the row's max n_contrib), the body
// is correct for a wave without one (alpha_eff = 0 everywhere), and without the branch the two steps of a trip are ONE basic
// block the scheduler can interleave -- the walk is a ~50-deep dependent chain per step with four waves per SIMD to hide it.
#ifndef GCR_K7_SKIP_BRANCH
#define GCR_K7_SKIP_BRANCH 0
#endif
#ifndef GCR_K7_XCD_MAP  /* 0 = round 4's unit order (A/B builds) */
#define GCR_K7_XCD_MAP 1
#endif
#if GCR_K7_SKIP_BRANCH
#define GCR_K7_WAVE_SKIP(x) (__ballot(x) != 0ull)
#else
#define GCR_K7_WAVE_SKIP(x) true
#endif
#ifdef GCR_EXPERIMENTS  /* knock-out bit 3: one LDS read per step instead of three (results wrong) */
#define GCR_BWD_LOAD(QA, QB, QC, OFF)                              \
  QA = *reinterpret_cast<const float4*>(sEb + (OFF));             \
  if (!(a.debug_flags & 8)) {                                      \
    QB = *reinterpret_cast<const float4*>(sEb + (OFF) + 16);       \
    QC = *reinterpret_cast<const float4*>(sEb + (OFF) + 32);       \
  } else {                                                         \
    QB = make_float4(QA.z, 0.5f, QA.x, QA.y);                      \
    QC = make_float4(QA.w, -5.0f, __uint_as_float(0u), __uint_as_float(0u)); \
  }
#else
#define GCR_BWD_LOAD(QA, QB, QC, OFF)                              \
  QA = *reinterpret_cast<const float4*>(sEb + (OFF));             \
  QB = *reinterpret_cast<const float4*>(sEb + (OFF) + 16);         \
  QC = *reinterpret_cast<const float4*>(sEb + (OFF) + 32);
#endif



#ifdef GCR_K7_WAVES_PER_EU  /* A/B builds: ask for that many waves per SIMD (4 = what 118-120 VGPRs give) */
#define GCR_K7_OCC __attribute__((amdgpu_waves_per_eu(GCR_K7_WAVES_PER_EU, 8)))
#else
#define GCR_K7_OCC
#endif
__global__ __launch_bounds__(64) GCR_K7_OCC void k_blend_bwd(const GcrBlendArgs a) {
  __shared__ StagedEntry sE[WPASS + 1];
  __shared__ uint32_t sId[WPASS];
  __shared__ uint16_t sList[4][WLIST_STRIDE];
  __shared__ uint16_t sRel[WPASS];   // pass slots with something to flush, compacted
  __shared__ double sAcc[WPASS * 9];  // entry-major: the nine sums of pass slot j at [9j, 9j+9); DOUBLES -- see "LDS float atomics" below

  const int lane = threadIdx.x;
  if ((int)blockIdx.x < a.fill.blocks) {  // the launch's leading waves: zero fill of the dense outputs
    for (int sgi = 0; sgi < a.fill.nseg; sgi++)
      gcr_fill_zero_segment<64>(a.fill.ptr[sgi], a.fill.n[sgi], (int)blockIdx.x, a.fill.blocks, lane);
    return;
  }
#ifdef GCR_EXPERIMENTS
  bool clk_on = a.clock_buf != nullptr;
  unsigned long long clk[GCR_K7_NCLK] = {};
#endif
  GCR_K7_CLK(0)
  // a frame rendered without the backward's state (gcr_camera.backward == 0), or whose carve does not fit the buffer
  // this call was handed: nothing to walk -- the preprocess gradient kernel turns the survivors' gradients into NaN
  if (!gcr_frame_has_state(a.frame_in, a.binning_bytes)) return;
  const uint32_t piece_P = (uint32_t)a.frame_in[GCR_FRAME_PIECE];
  // (slot, quadrant) units: every slot of the forward's piece grid holds a work item or GCR_NO_TILE
  const unsigned long long nunits = 4ull * gcr_piece_slots(a.R, (unsigned long long)(a.gx * a.gy), piece_P);
  const float4* __restrict__ ckpt = reinterpret_cast<const float4*>(a.binning_base + a.frame_in[GCR_FRAME_CKPT_OFF]);
  const uint4* __restrict__ work = reinterpret_cast<const uint4*>(a.binning_base + a.frame_in[GCR_FRAME_WORK_OFF]);
  const uint16_t* __restrict__ masks = reinterpret_cast<const uint16_t*>(a.binning_base + a.frame_in[GCR_FRAME_MASK_OFF]);
  const float4* __restrict__ staged = reinterpret_cast<const float4*>(a.binning_base + a.frame_in[GCR_FRAME_STAGED_OFF]);
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const int acc_slot = GCR_K7_ROW_SLOT;  // which of the nine terms this lane adds to the accumulators
  const float bg0 = GCR_CAM(a, bg, a.bg, 0), bg1 = GCR_CAM(a, bg, a.bg, 1), bg2 = GCR_CAM(a, bg, a.bg, 2);
  const char* const sEb = reinterpret_cast<const char*>(sE);
  char* const acc_base = reinterpret_cast<char*>(&sAcc[acc_slot >= 0 ? acc_slot : 0]);
  if (lane == 0) {
    sE[WPASS].a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    sE[WPASS].b = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    sE[WPASS].c = make_float4(0.0f, __builtin_inff(), __uint_as_float(NO_ENTRY), 0.0f);
  }
#pragma unroll
  for (int k = 0; k < 9; k++) sAcc[k * WPASS + lane] = 0.0;  // kept zero by the flush from here on

  // the grid covers the units the host expects; should the forward have used a smaller piece, the waves stride on
  // Which unit a workgroup takes: 32 consecutive workgroups share 8 items x 4 quadrants so that the four quadrants of an
  // item are 8 workgroups apart -- workgroup b runs on XCD b % 8 (observed, for speed only), so an item's records,
  // masks, list entries and checkpoints are fetched into ONE of the eight L2s instead of up to four (round 5).
  const unsigned long long nvirt = (nunits + 31ull) & ~31ull;
  for (unsigned long long v = (unsigned long long)blockIdx.x - (unsigned long long)a.fill.blocks; v < nvirt;
       v += (unsigned long long)gridDim.x - (unsigned long long)a.fill.blocks) {
#if GCR_K7_XCD_MAP
    const unsigned long long item = (v >> 5) * 8ull + (v & 7ull);
    const int q = (int)((v >> 3) & 3ull);  // quadrant of the tile: the `wave` of K6's lane geometry
#else
    const unsigned long long item = v >> 2;
    const int q = (int)(v & 3ull);
#endif
    if (4ull * item >= nunits) continue;
    uint4 d = work[item];
    // (one 16-byte load: left alone the compiler fetches d.x, tests it, and only then fetches the rest -- a second
    // dependent round trip at the head of every unit's chain)
    asm volatile("" : "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));
    if (d.x == GCR_NO_TILE) continue;  // wave-uniform
    const int tile = (int)d.x;
    const uint32_t r0 = d.y, len = d.z, kpiece = d.w;
    const uint32_t npieces = gcr_piece_count(len, piece_P), cs = gcr_piece_size(len, piece_P);
    const uint32_t lo = kpiece * cs, hi = min(len, lo + cs);  // the piece: list entries [lo, hi), walked back to front
    const uint32_t sbase = r0 / piece_P + (uint32_t)tile;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const LaneGeom g = lane_geom(q * 64 + lane, tx, ty);
    const uint32_t quad_bits = (1u << g.bit[0]) | (1u << g.bit[1]) | (1u << g.bit[2]) | (1u << g.bit[3]);
    const bool inside = g.pxi < a.W && g.pyi < a.H;
    const float pixx = (float)g.pxi, pixy = (float)g.pyi;
    const size_t pix_id = (size_t)a.W * g.pyi + g.pxi;

    // ---- per-pixel state (all loads of the prologue are independent of each other)
    const float T_final = inside ? a.final_T[pix_id] : 0.0f;
    const uint32_t last_contributor = inside ? a.n_contrib[pix_id] : 0u;
    float dLp0 = 0.0f, dLp1 = 0.0f, dLp2 = 0.0f;
    if (inside) {  // (dL_dpix is the gradient of the image as it was handed out: mirrored / windowed as the store was)
      size_t iplane;
      const long long in_id = gcr_out_index(a, g.pxi, g.pyi, &iplane);
      if (in_id >= 0) {
        dLp0 = a.dL_dpix[in_id];
        dLp1 = a.dL_dpix[iplane + in_id];
        dLp2 = a.dL_dpix[2 * iplane + in_id];
      }
    }
    float4 ck = make_float4(0.0f, 0.0f, 0.0f, 0.0f), cf = ck;
    if (kpiece + 1u < npieces) {  // the list goes on behind this piece
      ck = ckpt[(size_t)(sbase + kpiece) * 256u + (uint32_t)(q * 64 + lane)];
      cf = ckpt[(size_t)(sbase + npieces - 1u) * 256u + (uint32_t)(q * 64 + lane)];
    }
    float bg_dot_dpixel = 0;
    bg_dot_dpixel += bg0 * dLp0;
    bg_dot_dpixel += bg1 * dLp1;
    bg_dot_dpixel += bg2 * dLp2;
    // entries any pixel of this row (4x4 block) / of the quadrant consumed
    uint32_t row_max = last_contributor;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const uint32_t t = __shfl_xor(row_max, o, 64);
      row_max = t > row_max ? t : row_max;
    }
    uint32_t rmax[4];
#pragma unroll
    for (int rr = 0; rr < 4; rr++) rmax[rr] = __shfl(row_max, rr * 16, 64);
    const uint32_t wave_max = max(max(rmax[0], rmax[1]), max(rmax[2], rmax[3]));
    GCR_K7_CLK(1)  // pixel state known
    if (wave_max <= lo || !GCR_PASSES_ON) continue;  // wave-uniform: this quadrant consumed nothing of the piece
    const int top = (int)min(hi, wave_max);

    // Start state of the reverse walk.  A pixel whose last contributor lies in this piece (or before it) starts as
    // upstream: T_final, accum_rec = 0.  A pixel that goes on behind the piece starts from the forward's checkpoint
    // at boundary hi: T after entry hi - 1, and accum_rec = what is blended behind it, normalised:
    // (C_final - C_prefix(hi)) / T(hi) -- which the first update `a = last_alpha * lc + (1 - last_alpha) * acc`
    // with last_alpha = 0 hands to the first contributing entry unchanged.
    float T = T_final;
    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;  // accum_rec
    if (last_contributor > hi) {
      T = ck.x;
      acc0 = (cf.y - ck.y) / ck.x;
      acc1 = (cf.z - ck.z) / ck.x;
      acc2 = (cf.w - ck.w) / ck.x;
    }
    const float nbg = -T_final * bg_dot_dpixel;  // the background term's pixel constant (cr/backward.cu:556-560)
    float last_alpha = 0.0f, lc0 = 0.0f, lc1 = 0.0f, lc2 = 0.0f;

    // ---- passes of 64 list entries, back to front: pass slot `lane` of pass k holds list entry top - 1 - 64 k - lane.
    // A unit's life is a chain of dependent memory round trips, and round 3's knock-outs showed that they ADD (a wave per
    // unit, 16 waves per CU: nothing covers another wave's wait).  Per pass the chain was masks -> list entry -> records;
    // here (round 4) the block masks and list entries of ALL the unit's passes are requested at once (a piece is <= 256
    // entries = <= 4 passes) and the records of pass k + 1 are loaded while pass k is walked (round 5: from the forward's
    // staged copy in list order, so they do not wait for the ids either): work item -> pixel state -> masks + list +
    // records of the FIRST pass is the whole chain, whatever the number of passes.
    const int npass = (top - (int)lo + WPASS - 1) / WPASS;
    uint32_t wm0, wm1, wm2, wm3, wi0, wi1, wi2, wi3;
#define GCR_K7_WINDOW(K, M, I)                                         \
  {                                                                    \
    const int e = top - 1 - (K) * WPASS - lane;                        \
    const bool hv = (K) < npass && e >= (int)lo;                       \
    M = hv ? (uint32_t)masks[r0 + (uint32_t)(hv ? e : 0)] : 0u;        \
    I = hv ? a.list[r0 + (uint32_t)(hv ? e : 0)] : 0u;                 \
  }
    GCR_K7_WINDOW(0, wm0, wi0)
    GCR_K7_WINDOW(1, wm1, wi1)
    GCR_K7_WINDOW(2, wm2, wi2)
    GCR_K7_WINDOW(3, wm3, wi3)
#undef GCR_K7_WINDOW
    bool rel_n = (wm0 & quad_bits) != 0u;  // reaches one of this quadrant's 4x4 blocks (entries below wave_max only)
    float4 n0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), n1 = n0, n2 = n0;
    if (rel_n) {  // (the forward's staged copy, addressed by list position: no dependence on the id)
      const float4* __restrict__ rec = staged + (size_t)(r0 + (uint32_t)(top - 1 - lane)) * 3;
      n0 = rec[0];
      n1 = rec[1];
      n2 = rec[2];
    }
    for (int k = 0; k < npass; k++) {
      const int e_l = top - 1 - k * WPASS - lane;
      const uint32_t m16 = wm0, id = wi0;
      const bool any_rel = rel_n;
      const float4 q0 = n0, q1 = n1, q2 = n2;
      // the next pass's records: in flight while this pass is staged, walked and flushed
      wm0 = wm1; wm1 = wm2; wm2 = wm3; wm3 = 0u;
      wi0 = wi1; wi1 = wi2; wi2 = wi3; wi3 = 0u;
      rel_n = k + 1 < npass && (wm0 & quad_bits) != 0u;
      if (rel_n) {
        const float4* __restrict__ rec = staged + (size_t)(r0 + (uint32_t)(top - 1 - (k + 1) * WPASS - lane)) * 3;
        n0 = rec[0];
        n1 = rec[1];
        n2 = rec[2];
      }
      const uint64_t rel_bal = __ballot(any_rel);
      if (rel_bal == 0ull) continue;  // wave-uniform: nothing of these 64 entries concerns the quadrant
      if (any_rel) {
        sE[lane].a = q0;  // (as the forward staged it: K6's pre-scaled conic, its skip bound)
        sE[lane].b = q1;
        sE[lane].c = make_float4(q2.x, q2.y, __uint_as_float((uint32_t)e_l), __uint_as_float((uint32_t)lane * 72u));
        sId[lane] = id;
        sRel[__popcll(rel_bal & lt_mask)] = (uint16_t)lane;
      }
      const int nrel = __popcll(rel_bal);
      // the four row lists: block bit set and list entry below the row's max n_contrib (lane order = walk order)
      int maxcnt = 0;
      int cnt[4];
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const bool rel = ((m16 >> g.bit[rr]) & 1u) && (uint32_t)e_l < rmax[rr];
        const uint64_t bal = __ballot(rel);
        if (rel) sList[rr][__popcll(bal & lt_mask)] = (uint16_t)(lane * (int)ENTRY_BYTES);
        cnt[rr] = __popcll(bal);
        maxcnt = max(maxcnt, cnt[rr]);
      }
#pragma unroll
      for (int rr = 0; rr < 4; rr++)
        for (int s = cnt[rr] + lane; s < maxcnt + 3; s += 64) sList[rr][s] = (uint16_t)WSENT_OFF;
      __builtin_amdgcn_wave_barrier();
      GCR_K7_CLK(2)  // first pass staged (overwritten by later passes: the last one counts)

      // software pipeline, two entries per trip (see K6)
      const uint16_t* lp = &sList[g.row][0];
      float4 qa0, qb0, qc0, qa1, qb1, qc1;
      uint32_t e0 = lp[0], e1 = lp[1];
      GCR_BWD_LOAD(qa0, qb0, qc0, e0)
      for (int i = 0; i < maxcnt; i += 2, lp += 2) {
        GCR_BWD_LOAD(qa1, qb1, qc1, e1)
        e0 = lp[2];
        GCR_BWD_STEP(qa0, qb0, qc0)
        if (i + 1 >= maxcnt) break;
        GCR_BWD_LOAD(qa0, qb0, qc0, e0)
        e1 = lp[3];
        GCR_BWD_STEP(qa1, qb1, qc1)
      }
      __builtin_amdgcn_wave_barrier();
      GCR_K7_CLK(3)  // walk of the pass done
      // flush: lane = (entry within a group of 4) * 16 + component: the nine sums of an entry go out from nine adjacent
      // lanes into one 64-byte record; the accumulator rows are left zeroed for the next pass
      {
        const int comp = lane & 15;
        const int rec_idx = comp < 3 ? comp : (comp == 8 ? 3 : comp + 1);  // record layout: gcr_internal.h
        if (comp < 9) {
          for (int i = lane >> 4; i < nrel; i += 4) {
            const int slot = (int)sRel[i];
            const float v = (float)sAcc[slot * 9 + comp];
            if (v != 0.0f) {
              sAcc[slot * 9 + comp] = 0.0;
              if (!GCR_FLUSH_ON) continue;
              if (!a.deterministic) {
                atomicAdd(&a.grad_rec[(size_t)sId[slot] * GCR_GRAD_REC_FLOATS + rec_idx], v);
              } else {  // fixed point with the Gaussian's own binary point, order-independent (gcr_internal.h)
                const uint32_t gid = sId[slot];
                const float4 q2 = a.rec[(size_t)gid * GCR_REC_QUADS + 2];
                const int kc = gcr_det_frac_bits(sE[slot].a.x, sE[slot].a.y, __float_as_uint(q2.z), __float_as_uint(q2.w), a.W, a.H, true);
                const int ko = gcr_det_frac_bits(sE[slot].a.x, sE[slot].a.y, __float_as_uint(q2.z), __float_as_uint(q2.w), a.W, a.H, false);
                const int kf = (rec_idx >= 4) ? kc : ko;  // record slots 4..8 = the five moments with a pixel offset in them
                const float sc = __builtin_ldexpf(v, kf);  // exact: a power of two (or +-inf: saturates below)
                const long long q = sc >= 9.2e18f ? 0x7fffffffffffffffll : (sc <= -9.2e18f ? -0x7fffffffffffffffll : __float2ll_rn(sc));
                unsigned long long* const r64 = reinterpret_cast<unsigned long long*>(a.grad_rec) + (size_t)gid * (GCR_GRAD_REC_FLOATS_DET / 2);
                atomicAdd(r64 + rec_idx, (unsigned long long)q);
                r64[GCR_DET_K_SLOT] = (unsigned long long)(kc + 64) | ((unsigned long long)(ko + 64) << 8);  // (the same from every flush)
              }
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
#ifdef GCR_EXPERIMENTS
    if (clk_on) {
      clk[4] = __builtin_readcyclecounter();
      if (lane == 0) {
        unsigned long long* o = a.clock_buf + (size_t)blockIdx.x * (GCR_K7_NCLK + 2);
        o[0] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) |         // HW_REG_HW_ID
               ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);  // HW_REG_XCC_ID
        o[1] = (unsigned long long)(top - (int)lo);
        for (int i = 0; i < GCR_K7_NCLK; i++) o[2 + i] = clk[i];
      }
      clk_on = false;  // first unit only
    }
#endif
  }
}

// ------------------------------------------------------------------------------------- K7, one workgroup per item
// Round 5.  The wave-per-(item, quadrant) kernel above gathers every list entry of a piece once per
// quadrant that can see it and flushes it once per quadrant (counter traffic 6.9x the algorithmic bytes at C2, 2.5
// record updates per consumed entry).  Here ONE 256-thread workgroup owns a (tile, piece) item:
//   1. prologue, all loads of a thread independent of each other: the work item, then the thread's pixel state (final T,
//      n_contrib, dL/dpixel, the two checkpoints) and -- thread t < n -- block mask, Gaussian id and staged record of list
//      entry hi - 1 - t (the piece back to front: a piece is <= 223 entries, one per thread): two round trips;
//   2. thread t loads that entry's record ONCE -- from the forward's staged copy in list order (48 B per instance, one
//      coalesced block per piece, no dependence on the id) -- and stages it if its mask reaches the tile; barrier;
//   3. every wave builds the four row lists of its quadrant from the shared masks (its own rows' max n_contrib bounds
//      them, no cross-wave value is needed) and walks them exactly as above -- no barrier inside the walk; the row
//      sums of all four waves meet in ONE LDS accumulator column per entry (ds_add_f64 is atomic across waves);
//   4. barrier; the nine sums of an entry leave as ONE record update per (entry, piece), nine adjacent lanes.
// Two barriers in a workgroup's life, none in the walk.  The float sums of different waves meet in LDS in arrival order, so
// the deterministic mode (whose promise is bit-identical reruns) stays on the kernel above.
#ifndef GCR_K7_IPASS  /* A/B builds: a smaller staging capacity (bwd_piece must not exceed it) */
#define GCR_K7_IPASS GCR_PIECE_MAX
#endif
#ifdef GCR_K7_ITEM_WAVES  /* A/B builds: ask for that many waves per SIMD */
#define GCR_K7_ITEM_OCC __attribute__((amdgpu_waves_per_eu(GCR_K7_ITEM_WAVES, GCR_K7_ITEM_WAVES)))
#else
#define GCR_K7_ITEM_OCC
#endif
constexpr int IPASS = GCR_K7_IPASS;                  // entries a workgroup stages: one whole piece
constexpr int ILIST_STRIDE = IPASS + 9;              // byte slots per row list: entries + pipeline pads (232)
static_assert(IPASS < 256, "one thread per piece entry, byte list slots (slot IPASS = the sentinel)");

__global__ __launch_bounds__(256) GCR_K7_ITEM_OCC void k_blend_bwd_item(const GcrBlendArgs a) {
  __shared__ StagedEntry sE[IPASS + 1];
  __shared__ uint32_t sId[IPASS + 1];
  __shared__ uint16_t sMask[256];
  __shared__ uint8_t sList[4][4][ILIST_STRIDE];
  __shared__ double sAcc[IPASS * 9];  // entry-major: the nine sums of slot j at [9j, 9j+9), as doubles

  const int tid = threadIdx.x;
  if ((int)blockIdx.x < a.fill.blocks) {  // the launch's leading workgroups: zero fill of the dense outputs
    for (int sgi = 0; sgi < a.fill.nseg; sgi++)
      gcr_fill_zero_segment<256>(a.fill.ptr[sgi], a.fill.n[sgi], (int)blockIdx.x, a.fill.blocks, tid);
    return;
  }
  if (!gcr_frame_has_state(a.frame_in, a.binning_bytes)) return;  // (see k_blend_bwd)
  const uint32_t piece_P = (uint32_t)a.frame_in[GCR_FRAME_PIECE];
  const unsigned long long nitems = gcr_piece_slots(a.R, (unsigned long long)(a.gx * a.gy), piece_P);
  const float4* __restrict__ ckpt = reinterpret_cast<const float4*>(a.binning_base + a.frame_in[GCR_FRAME_CKPT_OFF]);
  const uint4* __restrict__ work = reinterpret_cast<const uint4*>(a.binning_base + a.frame_in[GCR_FRAME_WORK_OFF]);
  const uint16_t* __restrict__ masks = reinterpret_cast<const uint16_t*>(a.binning_base + a.frame_in[GCR_FRAME_MASK_OFF]);
  const float4* __restrict__ staged = reinterpret_cast<const float4*>(a.binning_base + a.frame_in[GCR_FRAME_STAGED_OFF]);
  const int lane = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave's quadrant of the tile
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const int acc_slot = GCR_K7_ROW_SLOT;
  const float bg0 = GCR_CAM(a, bg, a.bg, 0), bg1 = GCR_CAM(a, bg, a.bg, 1), bg2 = GCR_CAM(a, bg, a.bg, 2);
  const char* const sEb = reinterpret_cast<const char*>(sE);
  char* const acc_base = reinterpret_cast<char*>(&sAcc[acc_slot >= 0 ? acc_slot : 0]);
  if (tid == 0) {
    sE[IPASS].a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    sE[IPASS].b = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    sE[IPASS].c = make_float4(0.0f, __builtin_inff(), __uint_as_float(NO_ENTRY), 0.0f);
  }

  bool first = true;
  for (unsigned long long item = (unsigned long long)blockIdx.x - (unsigned long long)a.fill.blocks; item < nitems;
       item += (unsigned long long)gridDim.x - (unsigned long long)a.fill.blocks) {
    uint4 d = work[item];
    asm volatile("" : "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));  // one 16-byte load (see k_blend_bwd)
    if (d.x == GCR_NO_TILE) continue;  // workgroup-uniform
    if (!first) __syncthreads();       // (a second item only when the forward used a smaller piece than the host expects)
    first = false;
    const int tile = (int)d.x;
    const uint32_t r0 = d.y, len = d.z, kpiece = d.w;
    const uint32_t npieces = gcr_piece_count(len, piece_P), cs = gcr_piece_size(len, piece_P);
    const uint32_t lo = kpiece * cs, hi = min(len, lo + cs);  // the piece: list entries [lo, hi), walked back to front
    const int n = __builtin_amdgcn_readfirstlane((int)min(hi - lo, (uint32_t)IPASS));  // (cs <= piece_P <= 223)
    const uint32_t sbase = r0 / piece_P + (uint32_t)tile;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const LaneGeom g = lane_geom(tid, tx, ty);
    const bool inside = g.pxi < a.W && g.pyi < a.H;
    const float pixx = (float)g.pxi, pixy = (float)g.pyi;
    const size_t pix_id = (size_t)a.W * g.pyi + g.pxi;

    // ---- prologue: every load below is independent of the others
    const bool hv = tid < n;
    const uint32_t e_t = hi - 1u - (uint32_t)(hv ? tid : 0);  // staging slot `tid` holds list entry hi - 1 - tid
    const uint32_t m16 = hv ? (uint32_t)masks[r0 + e_t] : 0u;
    const uint32_t id = hv ? a.list[r0 + e_t] : 0u;
    const float T_final = inside ? a.final_T[pix_id] : 0.0f;
    const uint32_t last_contributor = inside ? a.n_contrib[pix_id] : 0u;
    float dLp0 = 0.0f, dLp1 = 0.0f, dLp2 = 0.0f;
    if (inside) {
      size_t iplane;
      const long long in_id = gcr_out_index(a, g.pxi, g.pyi, &iplane);
      if (in_id >= 0) {
        dLp0 = a.dL_dpix[in_id];
        dLp1 = a.dL_dpix[iplane + in_id];
        dLp2 = a.dL_dpix[2 * iplane + in_id];
      }
    }
    float4 ck = make_float4(0.0f, 0.0f, 0.0f, 0.0f), cf = ck;
    if (kpiece + 1u < npieces) {  // the list goes on behind this piece
      ck = ckpt[(size_t)(sbase + kpiece) * 256u + (uint32_t)tid];
      cf = ckpt[(size_t)(sbase + npieces - 1u) * 256u + (uint32_t)tid];
    }
    // the accumulators of this item's entries start at zero (stores, no wait)
    for (int k = tid; k < n * 9; k += 256) sAcc[k] = 0.0;

    // ---- the entry's record, staged once for the whole tile: the forward's own staged copy (centre, pre-scaled conic,
    // opacity, colour, skip bound), addressed by list position -- a piece is ONE coalesced block, and the load does not
    // wait for the id (counter traffic without the zero fill 2.26x -> see DESIGN.md section 7)
    if (hv) {
      const float4* __restrict__ rec = staged + (size_t)(r0 + e_t) * 3;
      const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
      if (m16 != 0u) {
        sE[tid].a = q0;
        sE[tid].b = q1;
        sE[tid].c = make_float4(q2.x, q2.y, __uint_as_float(e_t), __uint_as_float((uint32_t)tid * 72u));
        sId[tid] = id;
      }
    }
    sMask[tid] = (uint16_t)m16;

    float bg_dot_dpixel = 0;
    bg_dot_dpixel += bg0 * dLp0;
    bg_dot_dpixel += bg1 * dLp1;
    bg_dot_dpixel += bg2 * dLp2;
    uint32_t row_max = last_contributor;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const uint32_t t = __shfl_xor(row_max, o, 64);
      row_max = t > row_max ? t : row_max;
    }
    uint32_t rmax[4];
#pragma unroll
    for (int rr = 0; rr < 4; rr++) rmax[rr] = __shfl(row_max, rr * 16, 64);
    const uint32_t wave_max = max(max(rmax[0], rmax[1]), max(rmax[2], rmax[3]));

    float T = T_final;
    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;  // accum_rec; start state as in k_blend_bwd
    if (last_contributor > hi) {
      T = ck.x;
      acc0 = (cf.y - ck.y) / ck.x;
      acc1 = (cf.z - ck.z) / ck.x;
      acc2 = (cf.w - ck.w) / ck.x;
    }
    const float nbg = -T_final * bg_dot_dpixel;
    float last_alpha = 0.0f, lc0 = 0.0f, lc1 = 0.0f, lc2 = 0.0f;
    __syncthreads();  // records, masks, zeroed accumulators

    if (wave_max > lo && GCR_PASSES_ON) {  // wave-uniform: else this quadrant consumed nothing of the piece
      // the four row lists: block bit set and list entry below the row's max n_contrib (slot order = walk order)
      uint8_t* const lw = &sList[q][0][0];
      int cnt[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (k * 64 < n) {  // wave-uniform
          const int j = k * 64 + lane;
          const uint32_t m = (uint32_t)sMask[j];  // (0 for slots >= n)
          const uint32_t e_j = hi - 1u - (uint32_t)j;
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const bool rel = ((m >> g.bit[rr]) & 1u) && e_j < rmax[rr];
            const uint64_t bal = __ballot(rel);
            if (rel) lw[rr * ILIST_STRIDE + cnt[rr] + __popcll(bal & lt_mask)] = (uint8_t)j;
            cnt[rr] += __popcll(bal);
          }
        }
      }
      const int maxcnt = __builtin_amdgcn_readfirstlane(max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3])));
#pragma unroll
      for (int rr = 0; rr < 4; rr++)
        for (int s = cnt[rr] + lane; s < maxcnt + 3; s += 64) lw[rr * ILIST_STRIDE + s] = (uint8_t)IPASS;
      __builtin_amdgcn_wave_barrier();

      // software pipeline, two entries per trip (see K6)
      const uint8_t* lp = &sList[q][g.row][0];
      float4 qa0, qb0, qc0, qa1, qb1, qc1;
      uint32_t e0 = (uint32_t)lp[0] * ENTRY_BYTES, e1 = (uint32_t)lp[1] * ENTRY_BYTES;
      GCR_BWD_LOAD(qa0, qb0, qc0, e0)
      for (int i = 0; i < maxcnt; i += 2, lp += 2) {
        GCR_BWD_LOAD(qa1, qb1, qc1, e1)
        e0 = (uint32_t)lp[2] * ENTRY_BYTES;
        GCR_BWD_STEP(qa0, qb0, qc0)
        if (i + 1 >= maxcnt) break;
        GCR_BWD_LOAD(qa0, qb0, qc0, e0)
        e1 = (uint32_t)lp[3] * ENTRY_BYTES;
        GCR_BWD_STEP(qa1, qb1, qc1)
      }
    }
    __syncthreads();  // every wave's row sums are in the accumulators
    // flush: 16 entries per trip, nine adjacent lanes each -> one 64-byte record update per (entry, piece)
    {
      const int comp = tid & 15;
      const int rec_idx = comp < 3 ? comp : (comp == 8 ? 3 : comp + 1);  // record layout: gcr_internal.h
      if (comp < 9 && GCR_FLUSH_ON) {
        for (int i = tid >> 4; i < n; i += 16) {
          const float v = (float)sAcc[i * 9 + comp];
          if (v != 0.0f) atomicAdd(&a.grad_rec[(size_t)sId[i] * GCR_GRAD_REC_FLOATS + rec_idx], v);
        }
      }
    }
  }
}

#undef GCR_BWD_STEP
#undef GCR_BWD_LOAD

}  // namespace

template <bool SORT>
static void launch_blend_fwd_state(const GcrBlendArgs& a, int T, hipStream_t s) {
  size_t pad = 0;
#ifdef GCR_EXPERIMENTS  // unused dynamic LDS: fewer blend workgroups per CU (8 by default), room for another frame's kernels
  if (const char* e = getenv("GCR_K6_LDS_PAD")) pad = (size_t)atoll(e);
#endif
  if (a.work != nullptr)  // the backward's state is wanted (gcr_camera.backward == 1, or gcr_backward's regeneration pass)
    k_blend_fwd<SORT, true><<<T, 256, pad, s>>>(a);
  else
    k_blend_fwd<SORT, false><<<T, 256, pad, s>>>(a);
}

hipError_t gcr_launch_blend_fwd(const GcrBlendArgs& a, bool sort_in_kernel, hipStream_t s) {
  const int T = a.gx * a.gy;
  if (T <= 0) return hipSuccess;
  if (sort_in_kernel)
    launch_blend_fwd_state<true>(a, T, s);
  else
    launch_blend_fwd_state<false>(a, T, s);
  return hipGetLastError();
}

hipError_t gcr_launch_blend_bwd(const GcrBlendArgs& a, bool wave_units, hipStream_t s) {
  const int T = a.gx * a.gy;
  if (T <= 0) return hipSuccess;
  // one workgroup per work item (or one wave per (item, quadrant)) for the piece size the host expects the forward to
  // have used; the number of items is only known on the device, the slot count bounds it
  unsigned long long items = gcr_piece_slots(a.R, (unsigned long long)T, (unsigned long long)a.piece);
  if (wave_units || a.deterministic) {
    unsigned long long units = (4ull * items + 31ull) & ~31ull;
#ifdef GCR_EXPERIMENTS
    if (const char* env = getenv("GCR_K7_BLOCKS")) units = (unsigned long long)atoll(env);
#endif
    const unsigned int grid = (unsigned int)(units > 0x3fffffe0ull ? 0x3fffffe0ull : units) + (unsigned int)a.fill.blocks;
    k_blend_bwd<<<grid, 64, 0, s>>>(a);
  } else {
    GcrBlendArgs b = a;
    b.fill.blocks = (a.fill.blocks + 3) / 4;  // the same number of filling threads in 256-thread workgroups
    const unsigned int grid = (unsigned int)(items > 0x3fffffffull ? 0x3fffffffull : items) + (unsigned int)b.fill.blocks;
    k_blend_bwd_item<<<grid, 256, 0, s>>>(b);
  }
  return hipGetLastError();
}
