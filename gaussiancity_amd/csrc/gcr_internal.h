// gcr_internal.h -- host-side declarations shared by the translation units of libgcr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gcr.h"

// One Gaussian's projected state, written by K1 and gathered by K3/K6/K7/K8.  ONE 64-byte block, four 16-byte quads:
//   q0 = (x, y, conic.x, conic.y)   q1 = (conic.z, opacity, r, g)
//   q2 = (b, depth, rect_x, rect_y) with rect_* = min | max << 16 (tile units, as uint bits)
//   q3 = (clamp mask as uint bits -- bit ch: colour channel ch was clamped at 0, cr/forward.cu:68-73 --, 0, 0, 0)
// Until round 5 the record was 48 bytes at a 48-byte stride with the clamp mask in a byte array of its own: K1 is bound
// by its bytes, and a survivor's 48-byte store covered one 32-byte sector and half of another, its mask one byte of a
// third -- partial sectors are read, merged and written back.  64-byte records are whole sectors for the writer and one
// 64-byte block instead of one and a half for every gather.
#define GCR_REC_QUADS 4
// ... and its world-space covariance (K8 reads it back): six floats in a 32-byte slot, written as one whole sector
#define GCR_COV3D_FLOATS 8

// Per-tile atomic counters are padded to one per 128-byte line: device-scope atomics to the
// same line serialise (measured: C2's 1120 unpadded counters made K1 4.6x slower).
#define GCR_CURSOR_STRIDE 32

// Camera constants by value (gcr_camera.host_camera): when `by_value` is set the kernels take bg / view / proj /
// campos from their own argument block instead of loading them through the pointers.
struct GcrCamVals {
  int by_value;
  float view[16], proj[16], campos[3], bg[3];
};
#define GCR_CAM(A, FIELD, PTR, i) ((A).cam.by_value ? (A).cam.FIELD[i] : (PTR)[i])

struct GcrPreprocessArgs {
  int P, D, M, W, H, gx, gy;
  float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
  const float *means3D, *scales, *rotations, *opacities, *shs, *cov3D_precomp, *colors_precomp;
  const float *view, *proj, *campos;
  int32_t* radii;
  float4* rec;
  float* cov3D;
  uint32_t* tile_count;  // [T * GCR_CURSOR_STRIDE] per-tile instance counts (zeroed before K1)
  uint32_t* vis_list;    // [P] block b's survivors at [b*chunk, b*chunk + vis_count[b])
  uint32_t* vis_count;   // [nblocks]
  uint4* vis_rec;        // [P] {index, depth bits, rect_x, rect_y} of every survivor, at its position in vis_list
  uint32_t* cand_list;   // [P] K1a's candidates of block b at [b*chunk, b*chunk + cand_count[b])
  uint32_t* cand_count;  // [nblocks]
  unsigned long long* block_tiles;  // [nblocks] every K1 block's share of num_rendered
  int nblocks, chunk;    // persistent grid: block b owns Gaussians [b*chunk, (b+1)*chunk)
  int s_mean, s_opac, s_col, s_scale, s_rot;  // row strides in floats (3 / 1 / 3 / 3 / 4 when dense)
  int prefiltered;  // gcr_camera.prefiltered: a Gaussian behind the near plane is an error of the caller (GCR_PREFILTER_*)
  const float4* cull_cache;  // gcr_gaussians.cull_cache, part A: [P] (mean, rho) records of gcr_build_cull_cache, or null (stateless)
  const float4* cull_shape;  // ... part B: [P] x 32 bytes (scales, opacity, rotation) or (covariance, opacity, 0)
  int nt_stream;  // the stateless fused kernel streams its inputs with the non-temporal policy (gcr_preprocess.hip "NT"): results identical
  GcrCamVals cam;
};

// gcr_camera.prefiltered (cr/auxiliary.h:135-156: "Point is filtered although prefiltered is set. This shouldn't happen!" +
// __trap() upstream).  Here a K1 block that meets such a Gaussian sets bit 63 of its share of num_rendered (block_tiles),
// the kernel that sums the shares publishes GCR_PREFILTER_MARK instead of num_rendered -- which also vetoes every later
// kernel of the frame, like any num_rendered beyond the capacity -- and the host turns it into GCR_ERR_INVALID_ARGUMENT
// with the reference's text instead of a dead context.
#define GCR_PREFILTER_FLAG (1ull << 63)
#define GCR_PREFILTER_MARK 0xFFFFFFFEull

// Persistent-grid geometry of K1 (also used by the kernels that walk its visible lists):
// 8 workgroups per CU (a measured constant, see gcr_preprocess_resident_blocks), at most GCR_K1_MAX_BLOCKS.
#define GCR_K1_MAX_BLOCKS 2048
int gcr_preprocess_resident_blocks(bool split);  // 8 x CUs of the current device (cached)
static inline void gcr_preprocess_grid(int P, int max_blocks, int* nblocks, int* chunk) {
  int nb = (P + 255) / 256;
  if (nb > max_blocks) nb = max_blocks;
  if (nb < 1) nb = 1;
  long long c = ((long long)P + nb - 1) / nb;
  c = (c + 255) / 256 * 256;
  if (c < 256) c = 256;
  // two streaming iterations per block once that still leaves >= 512 blocks: a block's one processing pass then
  // finds twice the candidates (C2, 17 % visible: 41 -> 82 of 256 lanes busy; forward 75.8 -> 70.7 us, the
  // backward's walk over fewer, longer lists 117.8 -> 119.0 us)
  if (c < 512 && (long long)P >= 512ll * 512ll) c = 512;
  *chunk = (int)c;
  *nblocks = (int)(((long long)P + c - 1) / c);
  if (*nblocks < 1) *nblocks = 1;
}

struct GcrPreprocessBwdArgs {
  int P, D, M, W, H;
  float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
  const float *means3D, *scales, *rotations, *shs, *cov3D;  // cov3D: precomp or geometry state
  const float *view, *proj, *campos;
  const int32_t* radii;
  int s_cov3d;  // floats between two Gaussians' covariances: 6 for the caller's cov3D_precomp, GCR_COV3D_FLOATS for the state
  const uint32_t *vis_list, *vis_count;  // K1's per-block survivor lists
  int nblocks, chunk;
  const float4* grad_rec;  // K7's per-Gaussian accumulation records (GCR_GRAD_REC_FLOATS each)
  const float4* rec;       // K1's projected records (conic, opacity: the factors of the records' moments)
  const unsigned long long* frame;  // device frame words (null when nothing was rendered): a frame without backward
  unsigned long long binning_bytes; //   state, or one whose carve exceeds the buffer handed in, gets NaN gradients
  float *dL_dmean2D, *dL_dcolor, *dL_dopacity;  // written here from the records (API outputs)
  float *dL_dmean3D, *dL_dcov3D, *dL_dsh, *dL_dscale, *dL_drot;
  int deterministic;                            // grad_rec holds fixed-point records (GCR_GRAD_REC_FLOATS_DET)
  int s_mean, s_scale, s_rot;                   // input row strides in floats
  int g_mean, g_opac, g_col, g_scale, g_rot;    // output row strides in floats
  GcrCamVals cam;
};

// K7 accumulates its nine per-(tile, Gaussian) sums into ONE 64-byte record per Gaussian
//   [0..2] = dL_dcolor.rgb   [3] = S   [4] = Sx  [5] = Sy  [6] = Sxx  [7] = Sxy  [8] = Syy
// so that the nine global atomics of a flush land in one cache line and leave the CU as one
// memory-side transaction (scattered over four arrays they were 44 % of K7: DESIGN.md section 5).
// S.. are MOMENTS of u = G * dL/dalpha over the pixel offset (dx, dy) from the Gaussian's centre (round 5, gcr_blend.hip
// "MOMENTS"): dL_dopacity = S, dL_dconic = -0.5 o (Sxx, Sxy, Syy), dL_dmean2D = -o ((cx Sx + cy Sy) W/2, (cz Sy + cy Sx) H/2).
// K8 reads the record with three dwordx4 loads, applies those factors and writes the API's dL_dmeans2D / dL_dcolors /
// dL_dopacity.
#define GCR_GRAD_REC_FLOATS 16
// Option "deterministic_backward": the record is nine 64-bit FIXED-POINT sums (same order) in a 128-byte slot instead
// of nine floats in a 64-byte one.  Integer addition is associative, so the per-Gaussian totals no longer depend on the
// order in which the tiles' waves reach the memory-side atomic units -- two runs give the same bits (SURVEY.md section
// 5 asks for such a debug mode; the float atomics of the default path differ in the last bits from run to run, like the
// reference's).
// The binary point is PER GAUSSIAN (round 4; a fixed Q31.32 before): the sums of a Gaussian that covers many pixels far
// from its centre are large -- dL_dconic adds 0.5 d.x^2 dL_dG per pixel, 3.5e9 for a 300-pixel footprint, beyond Q31.32
// (found by tools/fuzz_parity.py, seed 41 case 635: gradients wrong by 15 x max, silently) -- while a small Gaussian's
// need the fine end.  k = fractional bits, a function of the Gaussian's projected record and the image size alone, one
// for the three conic sums and one for the other six (a conic-sized range would cost the dL_dmean2D sums, which the
// projection multiplies by focal / depth afterwards, their precision: 0.27 on a dL_dmean3D of 12 in the same case):
//     B_conic = pixels of its tile rectangle * max(squared distance centre -> farthest rectangle corner, 8 max(W, H)) * 64
//               (the five moments with a pixel offset in them: record slots 4..8)
//     B_other = pixels of its tile rectangle * 8 max(W, H) * 64   (colour sums and S: slots 0..3)
//     k = clamp(61 - (floor(log2 B) + 1), -16, 32)
// (the sums are bounded by pixels * {d^2 | W, 1} * |dL_dG|: B leaves |dL_dG| up to 64 per pixel before a sum can wrap;
// an addend beyond the range still saturates).  The blend gradient kernel computes both when it flushes and leaves
// (k_conic + 64) | (k_other + 64) << 8 in the record's last slot (every flush of a Gaussian stores the same value); the
// preprocess gradient kernel reads them.
#define GCR_GRAD_REC_FLOATS_DET 32
#define GCR_DET_K_SLOT 15   // 64-bit slot of the record that holds k + 64 (0 = never flushed: all sums are zero)
static inline __host__ __device__ int gcr_det_frac_bits(float x, float y, uint32_t rect_x, uint32_t rect_y, int W, int H,
                                                        bool conic) {
  const float minx = 16.0f * (float)(rect_x & 0xffffu), maxx = 16.0f * (float)(rect_x >> 16);
  const float miny = 16.0f * (float)(rect_y & 0xffffu), maxy = 16.0f * (float)(rect_y >> 16);
  const float npix = (maxx - minx) * (maxy - miny);
  const float ax = x - minx, bx = x - maxx, ay = y - miny, by = y - maxy;
  const float dx2 = ax * ax > bx * bx ? ax * ax : bx * bx, dy2 = ay * ay > by * by ? ay * ay : by * by;
  const float wh = 8.0f * (float)(W > H ? W : H);
  float d2 = conic ? dx2 + dy2 : wh;
  d2 = d2 > wh ? d2 : wh;  // (a NaN centre compares false: wh)
  float B = npix * d2 * 64.0f;
  B = B > 1.0f ? B : 1.0f;
  union { float f; uint32_t u; } c;
  c.f = B;
  const int e = (int)((c.u >> 23) & 0xffu) - 127 + 1;  // > log2 B (255 - 126 for inf / NaN: the coarse end)
  const int k = 61 - e;
  return k > 32 ? 32 : (k < -16 ? -16 : k);
}

// The dense zero fill of the backward's outputs (every Gaussian K8 does not visit keeps gradient 0): up to eight
// float arrays, streamed by the first `blocks` workgroups of the K7 launch while its tile workgroups -- which
// are VALU-bound and leave HBM idle -- walk their lists.
#define GCR_FILL_SEGMENTS 8
struct GcrFillArgs {
  float* ptr[GCR_FILL_SEGMENTS];
  unsigned long long n[GCR_FILL_SEGMENTS];  // floats
  int nseg;
  int blocks;  // workgroups of 256 threads that share the fill
};

// ---- backward pieces -------------------------------------------------------------------------------------
// The forward blend walks a tile's list in PIECES of equal size (<= P entries, P = option "bwd_piece") and leaves,
// at every piece boundary it crosses, a CHECKPOINT of the per-pixel state (T, prefix colour) -- 16 bytes x 256
// pixels -- plus the final state once it has crossed one.  The backward blend then works on (tile, piece) ITEMS
// instead of whole tiles: a piece that is not the pixel's last starts from the checkpoint behind it (T from the
// forward, accum_rec = (C_final - C_prefix) / T) instead of walking everything behind it first
// (cr/backward.cu:495-580 walks the whole list per tile).
// Checkpoint slots: tile t's pieces own the slots from
//   slot_base(t) = floor(ranges[t].start / P) + t,   slot_base(t) + ceil(len / P) <= slot_base(t + 1),
// so neither a prefix sum over tiles nor a host-side count is needed; total slots = floor(R / P) + T.  Slot
// slot_base(t) + k holds the checkpoint at boundary k + 1 (k < pieces - 1); the tile's LAST slot holds the final state.
// Work items: slot s of tile t's range also holds a 16-byte item {tile, list start, list length, piece} if the
// forward walked into that piece, {GCR_NO_TILE} otherwise (a saturated tile's remaining pieces, the gap slot between two
// tiles); the backward launches one wave per (slot, quadrant).
#define GCR_PIECE_MIN 64
#define GCR_PIECE_MAX 223  // (K6 keeps list slots in bytes, slot 223 is its sentinel: 8 workgroups per CU, gcr_blend.hip)
#define GCR_NO_TILE 0xFFFFFFFFu
#define GCR_CKPT_BYTES 4096  // 256 pixels x float4
static inline __host__ __device__ uint32_t gcr_piece_count(uint32_t len, uint32_t P) { return (len + P - 1u) / P; }
static inline __host__ __device__ uint32_t gcr_piece_size(uint32_t len, uint32_t P) {
  const uint32_t n = gcr_piece_count(len, P);
  return n ? (len + n - 1u) / n : 0u;  // equal pieces: 273 entries at P = 128 -> 3 x 91, not 128 + 128 + 17
}
static inline __host__ __device__ unsigned long long gcr_piece_slots(unsigned long long R, unsigned long long T,
                                                                     unsigned long long P) {
  return R / P + T;
}
// device frame words (geometry buffer, gcr_layout.geom_num_rendered): [0] R, [1] longest list, [2] go flag,
// [3] piece size the forward used, [4] / [5] byte offsets of the checkpoints / the work list in the binning buffer,
// [7] byte offset of the block masks the forward blend computed for every list entry it staged
#define GCR_FRAME_PIECE 3
#define GCR_FRAME_CKPT_OFF 4
#define GCR_FRAME_WORK_OFF 5
#define GCR_FRAME_MASK_OFF 7  // byte offset of the per-instance block masks (uint16, sorted-list order)
#define GCR_FRAME_CARVE 6     // bytes of the binning buffer as the forward carved it (for ITS capacity): a backward
                              // that is handed a smaller buffer must not follow the offsets above
#define GCR_FRAME_STAGED_OFF 8  // byte offset of the staged records (48 B per instance, sorted-list order; round 5): the
                                // frame words' slot in the geometry buffer is 128 bytes, words 10..15 are free
// default of the process-wide option "band_sort_min" (instances of the caller's capacity guess; tuned in round 6)
#define GCR_BAND_SORT_MIN_DEFAULT 6000000
#define GCR_FRAME_BANDED 9      // != 0: the tile table of this frame was counted over the band-sorted survivors
                                // (gcr_binning.hip "band sort"; set by the count kernel, read by the scatter kernel)
// Has the forward left the backward's state in a buffer of `binning_bytes`?  (frame word 3 is zeroed by the count
// kernel of every frame and set by a forward blend that writes the state.)
static inline __host__ __device__ bool gcr_frame_has_state(const unsigned long long* frame, unsigned long long binning_bytes) {
  return frame[GCR_FRAME_PIECE] != 0ull && frame[GCR_FRAME_CARVE] <= binning_bytes;
}

struct GcrBlendArgs {
  const uint32_t* ranges;  // [T][2]
  const uint32_t* list;    // sorted instance -> Gaussian
  const uint64_t* pairs;   // fwd, sort-in-kernel variant: the tile segments of unsorted (depth << 32 | index) keys
  uint32_t* list_out;      // ... and where the sorted indices go (the backward walks them)
  const float4* rec;
  int W, H, gx, gy;
  const float* bg;
  float* final_T;
  uint32_t* n_contrib;
  float* out_color;         // fwd
  const unsigned long long* frame;  // optional device guard: frame[2]==0 -> kernel does nothing
  const float* dL_dpix;     // bwd
  float* grad_rec;          // bwd: [P][GCR_GRAD_REC_FLOATS] accumulation records (zeroed for K1's survivors)
  GcrFillArgs fill;         // bwd: zero fill of the dense outputs, streamed in slices between the work items
  int deterministic;        // bwd: fixed-point gradient records (option "deterministic_backward")
  int win_x, win_y, win_w, win_h;  // output window in image coordinates AFTER mirroring (win_w == 0: whole image)
  int flip_x, flip_y;       // fwd: out_color stored mirrored; bwd: dL_dpix loaded mirrored (gcr_camera.flip_x / flip_y)
  int out_u8;               // fwd: out_color is uint8 [H,W,3] video frames (gcr_camera.out_u8)
  int nt_out;               // fwd, frames without backward state: final_T / n_contrib / float image stored with the non-temporal policy
  GcrCamVals cam;           // bg by value when cam.by_value
  int debug_flags;  // experiment builds only (GCR_EXPERIMENTS, "k7_skip_flush"): bit 0 = K7 drops its global atomics
  // pieces / checkpoints (above)
  int piece;                       // fwd: piece size P
  float4* ckpt;                    // fwd: [slots][256] checkpoints
  uint4* work;                     // fwd: [slots] work items {tile, list start, list length, piece}
  uint16_t* mask_out;              // fwd: [R] block mask of every staged list entry (the backward reuses them)
  unsigned long long mask_off;     // fwd: its byte offset in the binning buffer (published in the frame words)
  float4* staged_out;              // fwd: [R][3] every list entry as staged (the backward's records, in list order)
  unsigned long long staged_off;   // fwd: its byte offset in the binning buffer (published in the frame words)
  unsigned long long* frame_out;   // fwd: device frame words; [3..7] published by the first tile
  unsigned long long ckpt_off, work_off;  // fwd: what it publishes (byte offsets in the binning buffer)
  unsigned long long carve_bytes;         // fwd: ... and the size of the carve they belong to
  unsigned long long* gate_words;         // fwd, asynchronous frames: the ticket's host words (the FRAME GATE, gcr_blend.hip)
  unsigned int gate_seq, gate_polls;
  unsigned long long binning_bytes;       // bwd: size of the buffer behind binning_base
  const char* binning_base;        // bwd: checkpoints / work list are found through the frame words
  const unsigned long long* frame_in;  // bwd
  uint4* lazy;                     // fwd: [T] lazy-sort states (gcr_sort.h) of lists the blend may have to sort on, or null
  unsigned long long R;            // bwd: num_rendered (with `piece`: bounds the number of work items for the grid)
#ifdef GCR_EXPERIMENTS
  unsigned long long* clock_buf;   // bwd: [grid][4 waves][10] phase clocks (gcr_debug_set_clock_buffer), or null
#endif
};

// launchers (each enqueues on `s`, returns hipGetLastError())
hipError_t gcr_launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present,
                                   hipStream_t s);
hipError_t gcr_launch_preprocess(const GcrPreprocessArgs& a, bool split, hipStream_t s);
hipError_t gcr_launch_build_cull_cache(const GcrPreprocessArgs& a, float4* outA, float4* outB, hipStream_t s);
// the cache buffer: part A at offset 0, part B at gcr_cull_cache_offset_b(P) (128-byte aligned)
static inline size_t gcr_cull_cache_offset_b(int P) { return (((size_t)P * 16) + 127) & ~(size_t)127; }
// zero the K7 accumulation records of K1's survivors / stream zeros over the backward's outputs (R == 0 frames)
hipError_t gcr_launch_zero_grad_records(int nblocks, int chunk, const uint32_t* vis_list, const uint32_t* vis_count,
                                        float4* grad_rec, int rec_quads, hipStream_t s);
hipError_t gcr_launch_fill(const GcrFillArgs& f, hipStream_t s);
hipError_t gcr_launch_scan_block_sums(uint32_t* block_sums, int n, unsigned long long* total,
                                      hipStream_t s);
// fallback binning: per-Gaussian tile counts from radii + record rect, then emit in index order
hipError_t gcr_launch_tiles_touched(int P, int nblocks, int chunk, const uint32_t* vis_list,
                                    const uint32_t* vis_count, const float4* rec, uint32_t* tiles_touched,
                                    uint32_t* block_sums, hipStream_t s);
hipError_t gcr_launch_emit(int P, const uint32_t* tiles_touched, const uint32_t* block_offsets,
                           const float4* rec, int gx, uint64_t* keys, uint32_t* vals,
                           hipStream_t s);
// Fast binning path: tile counts -> ranges/cursors (+ total, max), scatter, per-tile LDS sort.
hipError_t gcr_launch_scan_tiles(uint32_t* tile_cursor, int stride, uint32_t* ranges, int T,
                                 unsigned long long* frame, unsigned long long cap_instances,
                                 unsigned long long cap_list, unsigned long long* host_R, unsigned int seq,
                                 const unsigned long long* block_tiles, int nblocks_k1, hipStream_t s);
int gcr_tile_table_groups(int T, int nblocks_k1, int* G_out);
hipError_t gcr_launch_tile_count(int T, int gx, int NG, int G, int nblocks_k1, int chunk, const uint4* vis_rec,
                                 const uint32_t* vis_count, uint32_t* table,
                                 uint32_t* tile_total, uint32_t* tile_local, uint32_t* blk_total,
                                 unsigned long long* frame, const unsigned long long* block_tiles,
                                 unsigned long long* host_R, unsigned int seq, uint4* banded, int banded_capacity,
                                 hipStream_t s);
bool gcr_band_sort_possible(int T);  // band sort (gcr_binning.hip): `banded` = 16 B per survivor, null = off
hipError_t gcr_launch_tile_scatter(int T, int gx, int NG, int G, int nblocks_k1, int chunk, const uint4* vis_rec,
                                   const uint32_t* vis_count, uint32_t* table,
                                   const uint32_t* tile_total, const uint32_t* tile_local,
                                   const uint32_t* blk_total, uint32_t* ranges, uint64_t* pairs,
                                   unsigned long long* frame, unsigned long long cap_instances,
                                   unsigned long long cap_list, unsigned long long* host_longest,
                                   const uint4* banded, int banded_capacity, hipStream_t s);
hipError_t gcr_launch_scatter_instances(int nblocks, int chunk, const uint32_t* vis_list,
                                        const uint32_t* vis_count, const float4* rec, int gx,
                                        uint32_t* tile_cursor, uint64_t* pairs, const uint32_t* ranges, int T,
                                        const unsigned long long* frame, hipStream_t s);
int gcr_tile_sort_capacity(void);  // longest per-tile list the LDS sort accepts
hipError_t gcr_launch_tile_sort(const uint32_t* ranges, int T, uint64_t* pairs, uint64_t* pairs_spare,
                                uint32_t* list, int64_t list_length_hint, const unsigned long long* frame,
                                uint4* lazy, hipStream_t s);
// Stable LSD radix sort of R (u64 key, u32 value) pairs on bits [0, end_bit).  Ping-pongs
// between (k0,v0) and (k1,v1); returns in *sorted_half which half holds the result.
size_t gcr_sort_hist_bytes(int64_t R, int end_bit);
hipError_t gcr_launch_sort(uint64_t* k0, uint32_t* v0, uint64_t* k1, uint32_t* v1, int64_t R,
                           int end_bit, uint32_t* hist, int* sorted_half, hipStream_t s);
int gcr_sort_passes(int end_bit);
hipError_t gcr_launch_tile_ranges(const uint64_t* keys, int64_t R, uint32_t* ranges, int T,
                                  hipStream_t s);
hipError_t gcr_launch_blend_fwd(const GcrBlendArgs& a, bool sort_in_kernel, hipStream_t s);
// wave_units: one wave per (work item, quadrant) [round 4; always in the deterministic mode] instead of one workgroup per item
hipError_t gcr_launch_blend_bwd(const GcrBlendArgs& a, bool wave_units, hipStream_t s);
hipError_t gcr_launch_preprocess_bwd(const GcrPreprocessBwdArgs& a, hipStream_t s);

// cr/rasterizer_impl.cu:35-48
static inline uint32_t gcr_higher_msb(uint32_t n) {
  uint32_t msb = sizeof(n) * 4;
  uint32_t step = msb;
  while (step > 1) {
    step /= 2;
    if (n >> msb)
      msb += step;
    else
      msb -= step;
  }
  if (n >> msb) msb++;
  return msb;
}
