/*
 * gcr.h -- C ABI of the MI355X-native differentiable Gaussian rasterizer (libgcr_hip.so).
 *
 * This is the drop-in boundary for the hot path of hzxie/GaussianCity's
 * extensions/diff_gaussian_rasterization ("dgr/", CUDA sources under "cr/").  Plain pointers
 * and sizes only -- no torch types.  Every pointer is a DEVICE pointer (gfx950 HBM) unless
 * the name ends in _host.  All memory is owned by the caller (the reference lets torch own
 * it: dgr/rasterize_points.cu:57-68,118-126); nothing is retained across calls (the optional cull
 * cache is the caller's buffer, too).
 *
 * Entry point                      replaces (reference interface)
 * -------------------------------  ---------------------------------------------------------
 * gcr_rasterize_forward            CudaRasterizer::Rasterizer::forward   cr/rasterizer.h:25-37
 *                                  (== cr/rasterizer_impl.cu:178-283, std::function resize
 *                                  callbacks become C callbacks)
 * gcr_forward                      the same forward with every launch enqueued before the host
 *                                  learns num_rendered (no GPU idle gap at the sync)
 * gcr_forward_async                the same forward WITHOUT the host wait: num_rendered arrives later in a
 *                                  pinned word of the caller's (callers that never look at it -- the Python
 *                                  API discards it, dgr/__init__.py:404-420 -- enqueue frame after frame)
 * gcr_forward_preprocess +         the same forward split at its one host sync
 *   gcr_forward_render             (cr/rasterizer_impl.cu:236-238) so the caller allocates
 *                                  the binning buffer itself (dgr/rasterize_points.cu:27-33)
 * gcr_backward                     CudaRasterizer::Rasterizer::backward  cr/rasterizer.h:39-48
 *                                  (== cr/rasterizer_impl.cu:287-338)
 * gcr_mark_visible                 CudaRasterizer::Rasterizer::markVisible cr/rasterizer.h:22-23
 * gcr_geometry_bytes/_image_bytes/ required<GeometryState|ImageState|BinningState>(n)
 *   _binning_bytes                 cr/rasterizer_impl.h:65-69
 * gcr_last_error                   the std::runtime_error text (cr/auxiliary.h:158-167)
 * gcr_build_cull_cache +           (no counterpart: optional) what a caller who renders ONE mostly off-screen set of
 *   gcr_gaussians.cull_cache       Gaussians from many poses may keep so that the per-Gaussian pass of the forward
 *                                  (cr/forward.cu:147-233) streams 16 instead of 40 bytes per Gaussian; same outputs
 *
 * Conventions shared with the reference: matrices are 16 floats in the row-vector layout the
 * kernels index as m[0],m[4],m[8],m[12] (cr/auxiliary.h:48-56); quaternions are (r,x,y,z) and
 * are NOT normalised; absent optional inputs are NULL (the reference passes empty tensors,
 * dgr/__init__.py:250-259); all arithmetic fp32, radii int32.
 *
 * Return codes: 0 (or a non-negative count) on success, negative gcr_status on failure with
 * a message retrievable through gcr_last_error() (thread-local).
 */
#ifndef GCR_H_INCLUDED
#define GCR_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 9 since round 6.  The number moves when a struct changes size or an entry point its signature; two late changes of round 6 did
 * neither and kept it: the process-wide option "stream_policy" (gcr_set_option), and gcr_layout.geom_cov3D, which is still carved
 * and no longer written. */
#define GCR_ABI_VERSION 9
#define GCR_BLOCK_X 16 /* cr/config.h:16 */
#define GCR_BLOCK_Y 16 /* cr/config.h:17 */
#define GCR_NUM_CHANNELS 3 /* cr/config.h:15 */

typedef enum gcr_status {
  GCR_OK = 0,
  GCR_ERR_INVALID_ARGUMENT = -1, /* bad shape / null pointer / exactly-one-of violated */
  GCR_ERR_BUFFER_TOO_SMALL = -2, /* a scratch buffer is smaller than gcr_*_bytes() */
  GCR_ERR_OVERFLOW = -3,         /* num_rendered does not fit the 32-bit instance index */
  GCR_ERR_DEVICE = -4,           /* HIP runtime error (text in gcr_last_error) */
  GCR_ERR_ALLOC = -5             /* a resize callback returned NULL */
} gcr_status;

/* Per-call options (ABI v6; v7: "fast_exp" is gone, "bwd_wave_units" added at the end).  Every field: -1 = the process-wide default (gcr_set_option), >= 0 = this call's value.
 * The reference's entry points are re-entrant and carry no global state (dgr/rasterize_points.cu:37-93); with this
 * record two host threads can render with different options at the same time.  The forward and the backward of one
 * frame must be given the same values (as with the process-wide knobs).  Meanings: see gcr_set_option below. */
typedef struct gcr_options {
  int32_t lazy_sort;
  int32_t sort_in_blend;
  int32_t bwd_piece;
  int32_t deterministic_backward;
  int32_t split_preprocess;
  int32_t force_radix;
  int32_t force_global_cursor;
  int32_t bwd_wave_units;
} gcr_options;
#define GCR_OPTIONS_DEFAULT {-1, -1, -1, -1, -1, -1, -1, -1}

/* GaussianRasterizationSettings (dgr/__init__.py:203-215) */
typedef struct gcr_camera {
  int32_t img_h, img_w;
  float tanfovx, tanfovy;
  float scale_modifier;
  int32_t sh_degree;   /* active degree D, 0..3 */
  int32_t prefiltered; /* !=0: the caller declares that no Gaussian lies behind the near plane (view-space z <= 0.2).
                          Upstream a Gaussian that does then hits a device printf + __trap() (cr/auxiliary.h:148-152: "Point
                          is filtered although prefiltered is set. This shouldn't happen!").  Here (ABI v7) the frame is
                          not rendered and the call -- gcr_forward, gcr_forward_preprocess, gcr_rasterize_forward, or the
                          ticket of gcr_forward_async -- fails with GCR_ERR_INVALID_ARGUMENT and that text; the device and
                          the stream stay usable.  0 (what GaussianCity always passes, dgr/__init__.py:399): such
                          Gaussians are culled silently, as upstream */
  int32_t debug;       /* !=0: synchronise + check after every stage (cr/auxiliary.h:158) */
  const float *bg;          /* [3] */
  const float *view_matrix; /* [16] */
  const float *proj_matrix; /* [16] */
  const float *campos;      /* [3] */
  int32_t host_camera; /* !=0: bg / view_matrix / proj_matrix / campos are HOST pointers; the 38 floats are copied
                          into the kernels' arguments at call time (no device copy of the camera is needed, and the
                          kernels start without a dependent load).  0: device pointers as in the reference */
  int32_t flip_x, flip_y; /* !=0: out_color is written mirrored in x / y and dL_dpix is read mirrored -- what
                          GaussianRasterizerWrapper's flip_lr / flip_ud do with torch.flip after the render
                          (dgr/__init__.py:421-424), without the copy kernels; the pixel values are the same bits */
  int32_t win_x, win_y, win_w, win_h; /* win_w > 0: only this window of the (mirrored, if flip_*) image is wanted --
                          out_color and dL_dpix are [3, win_h, win_w]; tiles that do not touch the window are neither
                          blended nor walked by the backward (their pixels would be cropped away / have zero gradient:
                          GaussianCity renders 960x540 and keeps a 640x448 crop, utils/helpers.py:255-260).  radii,
                          num_rendered and the binning state are those of the full frame.  0: the whole image */
  int32_t backward;    /* never changes the image.  Forward calls: 1 = gcr_backward WILL be called on this frame: the forward
                          blend cuts its tile lists into the pieces the backward balances best with (option "bwd_piece")
                          and leaves the backward's per-piece state (checkpoints, work items, block masks) in the binning
                          buffer.  0 = inference: no such state is written (the headline workload does not pay for a
                          backward it never runs) and the binning buffer may be the smaller gcr_binning_bytes_lean().
                          gcr_backward needs a frame rendered with 1: on a frame rendered with 0 the gradients of every
                          rendered Gaussian come out as NaN (never as plausible zeros) -- render its state first with
                          gcr_forward_render(out_color = NULL, backward = 1) into a gcr_binning_bytes() buffer */
  const gcr_options *options; /* HOST pointer or NULL (= all defaults); read during the call only */
  int32_t out_u8;      /* !=0 (forward of an inference frame only, backward == 0): out_color is NOT float [3,H,W] but the
                          video frame the reference's render loop makes of it with five elementwise kernels
                          (scripts/inference.py:655-667: utils/helpers.tensor_to_image(img) * 255 -> uint8):
                          uint8 [H,W,3] (or [win_h,win_w,3]), value = (uint8)(((clamp(c, -1, 1) / 2 + 0.5) * 255)) in
                          float32 arithmetic, the same roundings in the same order -- the same bytes */
} gcr_camera;

/* Per-Gaussian inputs (argument list of cr/rasterizer.h:25-37) */
typedef struct gcr_gaussians {
  int32_t P;                   /* number of Gaussians, <= 700 000 000 (32-bit index arithmetic in the kernels) */
  int32_t M;                   /* SH coefficients per Gaussian (sh.size(1)); 0 without SH */
  const float *means3D;        /* [P,3] */
  const float *opacities;      /* [P] */
  const float *shs;            /* [P,M,3] or NULL */
  const float *colors_precomp; /* [P,3]   or NULL  (exactly one of shs/colors_precomp) */
  const float *scales;         /* [P,3]   or NULL */
  const float *rotations;      /* [P,4]   or NULL */
  const float *cov3D_precomp;  /* [P,6]   or NULL  (exactly one of scales+rotations/cov3D) */
  /* Row strides in floats (0 = dense: 3 / 1 / 3 / 3 / 4).  GaussianCity hands the rasterizer column slices of one
   * [N,14] tensor (dgr/__init__.py:404-409): with strides = 14 and the five pointers aimed at columns 0 / 3 / 4 / 7 /
   * 11 the kernels read it in place, where the reference's binding copies every slice (.contiguous(),
   * dgr/rasterize_points.cu:37-93 through torch).  shs and cov3D_precomp are always dense. */
  int32_t stride_means3D, stride_opacities, stride_colors, stride_scales, stride_rotations;
  /* ABI v8, optional (NULL = none; forward calls only, the backward never reads it): the buffer gcr_build_cull_cache
   * filled from THESE means3D / scales / rotations (or cov3D_precomp) / opacities at the camera's scale_modifier;
   * gcr_cull_cache_bytes(P) bytes, 128-byte aligned.  A set of Gaussians that does not change between frames and is
   * mostly off screen in each of them (a whole city flown through) is then culled from one 16-byte record per Gaussian instead of 40
   * bytes out of three arrays (56 when the rows are [N,14]), and the few per cent that survive the cull fetch their
   * scales, rotation and opacity as one 32-byte record instead of a 128-byte line of each array.  Every output is the
   * same bits as without it: the cull takes the same decisions from the same numbers, it only ever skips Gaussians
   * whose exact projection has radius 0 (cr/forward.cu:147-233 decides everything else), and the records are copies.
   * The library cannot see whether the arrays still hold what the cache was built from -- a stale cache renders the
   * OLD positions, sizes and opacities of the culled / surviving Gaussians; keeping it current is the caller's job (the
   * Python layer keys it on the tensors' versions).  It pays when most of the set is culled; a set that is mostly on
   * screen renders faster without it (the candidates' records are extra bytes then).  Ignored under option "split_preprocess". */
  const void *cull_cache;
} gcr_gaussians;

/* The cull cache of gcr_gaussians.cull_cache for g's means3D / scales / rotations (or cov3D_precomp) / opacities (strides
 * as in g) at `scale_modifier`.  Part A, P x 16 bytes: (mean_i, rho_i), rho = a bound on the spectral radius of Gaussian
 * i's world-space covariance (the camera-independent factor of the cull's screen bound, cr/forward.cu:126-144 in
 * interval form).  Part B, P x 32 bytes at the next 128-byte boundary: (scales, opacity, rotation) or (covariance,
 * opacity, 0).  One streaming kernel on `hip_stream` (44 -> 48 bytes per Gaussian); g->cull_cache itself is not read. */
size_t gcr_cull_cache_bytes(int32_t P);
int gcr_build_cull_cache(const gcr_gaussians *g, float scale_modifier, void *cull_cache_out, void *hip_stream);

/* Gradient outputs (cr/rasterizer.h:39-48).  The arrays may be UNINITIALISED memory: gcr_backward writes every
 * element of every output (zeros for Gaussians that were not rendered), where the reference asks its caller for
 * zero-filled tensors (dgr/rasterize_points.cu:118-126, torch::zeros) -- handing in zeroed arrays is harmless.
 * dL_dmeans2D / dL_dcolors / dL_dopacity are written by the preprocess-backward kernel from the accumulation
 * records. */
typedef struct gcr_grads {
  float *dL_dmeans2D;   /* [P,3] (x,y used) */
  float *dL_dconic;     /* [P, gcr_grad_record_floats()] (16 floats; 32 in the deterministic mode) scratch (contents undefined on return), 64-byte
                           aligned: the role of the
                           reference's dL_dconic [P,2,2] (dgr/rasterize_points.cu:121), widened to one
                           64-byte accumulation record per Gaussian so that K7's nine atomics per (tile piece,
                           Gaussian) share a cache line.  Since ABI v7 the nine sums are the colour gradient and six
                           MOMENTS of G * dL/dalpha over the pixel offset (S, Sx, Sy, Sxx, Sxy, Syy); the preprocess
                           gradient kernel turns them into dL_dopacity / dL_dmeans2D / the conic gradient with the
                           Gaussian's opacity and conic -- the record itself is no longer the reference's dL_dconic */
  float *dL_dopacity;   /* [P] */
  float *dL_dcolors;    /* [P,3] */
  float *dL_dmeans3D;   /* [P,3] */
  float *dL_dcov3D;     /* [P,6] */
  float *dL_dsh;        /* [P,M,3] (unused when shs==NULL) */
  float *dL_dscales;    /* [P,3] (unused when scales==NULL) */
  float *dL_drotations; /* [P,4] (unused when scales==NULL); 16-byte aligned (stored as float4) --
                           gcr_backward rejects a misaligned dL_dconic / dL_drotations with
                           GCR_ERR_INVALID_ARGUMENT */
  /* Row strides in floats of dL_dmeans3D / dL_dopacity / dL_dcolors / dL_dscales / dL_drotations (0 = dense) and,
   * when any is set, the ONE block all strided outputs live in (e.g. a [N,14] gradient tensor with strides 14):
   * gcr_backward zero-fills `packed` as a whole and writes the survivors' values into its columns -- the reference's
   * caller assembles that tensor with five slice-backward kernels and four adds. */
  int32_t stride_means3D, stride_opacity, stride_colors, stride_scales, stride_rotations;
  float *packed;         /* NULL when every output is dense */
  int64_t packed_floats;
} gcr_grads;

/* Byte offsets of the sub-arrays carved from the three opaque scratch buffers.  Exposed so
 * that tests can compare intermediate state with the oracle; not needed by normal callers. */
typedef struct gcr_layout {
  /* geometry buffer (per Gaussian) */
  size_t geom_rec;           /* float[16] per Gaussian, one 64-byte block (ABI v8; 12 floats at a 48-byte stride before):
                                x,y,conic.x,conic.y | conic.z,opacity,r,g | b,depth,rect_x(min|max<<16),rect_y(min|max<<16) |
                                clamp mask as uint32 (bit ch set = colour channel ch clamped), 0, 0, 0 */
  size_t geom_cov3D;         /* unused since round 6 (late): 32 bytes per Gaussian are still carved, nothing is written -- the backward
                                derives a Gaussian's covariance from its scales and rotation again (same function, same bits) instead of
                                every forward storing it (float[6] in a 32-byte slot in ABI v8 / v9 builds before that) */
  size_t geom_clamped;       /* unused since ABI v8 (the mask lives in the record's fourth quad); P bytes are still carved */
  size_t geom_tiles_touched; /* uint32 per Gaussian   (radix fallback path only) */
  size_t geom_block_sums;    /* uint32 per 256-Gaussian block (radix fallback path only) */
  size_t geom_vis_list;      /* uint32 per Gaussian: K1 block b's survivors, packed at b*chunk */
  size_t geom_vis_count;     /* uint32 per K1 block */
  size_t geom_num_rendered;  /* uint64 {num_rendered, longest tile list, go flag, backward piece size,
                                byte offsets of bin_ckpt / bin_work, carve size, offsets of bin_mask / bin_staged as the
                                forward carved them} (nine of the 32 words reserved here) */
  size_t geom_block_tiles;   /* uint64 per K1 block: its share of num_rendered */
  size_t geom_total;
  /* image buffer */
  size_t img_final_T;   /* float per pixel   (ImageState::accum_alpha) */
  size_t img_n_contrib; /* uint32 per pixel */
  size_t img_ranges;    /* uint32[2] per tile */
  size_t img_tile_cursor; /* one uint32 per tile, 128 B apart: instance count (after K1), then
                             scatter write cursor */
  size_t img_tile_table;  /* uint32 [groups][T]: per-group tile counts, then exclusive prefixes */
  size_t img_tile_lazy;   /* uint32[4] per tile {n_sorted, 0, L lo, L hi}: the first n_sorted entries of the tile's
                             list are in final order, every key >= L is not among them (option "lazy_sort") */
  size_t img_total;
  /* binning buffer (per instance) */
  size_t bin_keys[2]; /* uint64 per instance, ping/pong */
  size_t bin_vals[2]; /* uint32 per instance, ping/pong */
  size_t bin_hist;    /* radix-sort histogram table */
  size_t bin_sorted;  /* 0 or 1: which ping/pong half holds the sorted list */
  size_t bin_work;      /* 16 B per (tile, piece) slot: work items of the backward blend {tile, list start, list
                           length, piece} written by the forward blend for every piece it walked into */
  size_t bin_mask;      /* uint16 per instance, sorted-list order: which of the tile's sixteen 4x4 blocks the
                           entry can reach (computed by the forward blend while staging, reused by the backward) */
  size_t bin_ckpt;      /* 4096 B per slot: per-pixel (T, prefix colour) at the piece boundaries the forward
                           blend crossed -- what lets the backward blend start in the middle of a tile list */
  size_t bin_total;
  size_t bin_lean_total; /* everything in front of bin_work: all a frame with gcr_camera.backward == 0 uses */
  size_t bin_staged;    /* (ABI v7) 48 B per instance, sorted-list order: every entry as the forward blend staged it
                           (centre, pre-scaled conic, opacity, colour, skip bound) -- the backward blend reads a piece's
                           records as one coalesced block instead of gathering them by Gaussian index */
  size_t geom_vis_rec;  /* (ABI v9) uint32[4] per Gaussian, K1 block b's survivors packed at b*chunk like geom_vis_list:
                           {Gaussian index, depth bits, rect_x, rect_y} -- what the binning kernels need of a survivor, so
                           that they read K1's lists front to back instead of gathering a 64-byte record per survivor */
} gcr_layout;

/* Host-side summary of K1+K2, produced by gcr_forward_preprocess and consumed by
 * gcr_forward_render (the library keeps no state between calls). */
typedef struct gcr_frame_info {
  int64_t num_rendered;       /* R: total (Gaussian,tile) instances == the reference's return value */
  int64_t max_tile_instances; /* longest per-tile list: sizes the per-tile sort (lists beyond the LDS
                                 capacity of 4096 are sorted per tile by a merge sort through HBM) */
} gcr_frame_info;

int gcr_abi_version(void);
const char *gcr_last_error(void);

size_t gcr_geometry_bytes(int32_t P);
size_t gcr_image_bytes(int32_t W, int32_t H);
size_t gcr_binning_bytes(int64_t R, int32_t W, int32_t H);
/* binning buffer of a frame that will never see gcr_backward (gcr_camera.backward == 0): sorted list, keys and the
 * radix fallback's scratch only -- 24 B per instance instead of ~90 */
size_t gcr_binning_bytes_lean(int64_t R, int32_t W, int32_t H);
int gcr_get_layout(int32_t P, int32_t W, int32_t H, int64_t R, gcr_layout *out);

/* K1 (project, cov2D, SH colour, tile rect, per-tile instance counts) + K2 (scan of the tile
 * counts -> tile ranges).  Writes radii[P], fills geom and the tile tables of img, and returns
 * the frame summary through *info_host after ONE 16-byte D2H copy (the sync the reference has
 * at cr/rasterizer_impl.cu:236-238). */
int gcr_forward_preprocess(const gcr_camera *cam, const gcr_gaussians *g, void *geom,
                           size_t geom_bytes, void *img, size_t img_bytes, int32_t *radii,
                           gcr_frame_info *info_host, void *hip_stream);

/* Whole forward in ONE call without a mid-frame host stall.  `binning` must hold
 * gcr_binning_bytes(binning_capacity) bytes; binning_capacity is the caller's guess of
 * num_rendered (e.g. 1.5 x the previous frame's value; 0 = no guess) and tile_list_capacity the
 * longest per-tile list it expects (e.g. the previous frame's; 0 = 4096).  The list expectation only
 * sizes the LDS of the per-tile sort (1.5 x the expectation; with option "sort_in_blend" an
 * expectation <= 384 moves the sort into the forward blend) and never affects the result: a list
 * longer than expected is sorted by a slower path of the same kernel (any length; the sort kernel
 * merges sorted runs through the spare key buffer).  All kernels of the frame are enqueued at once --
 * they take the tile ranges from device memory and a device-side flag vetoes them if
 * binning_capacity was too small -- and the host waits only for num_rendered, which the exact
 * projection pass accumulates and the next kernel's first workgroup stores into a pinned host word the calling
 * thread polls: no copy, no event, and the wait ends as soon as that pass is done (the tile-table
 * kernels, the scatter, the sort and the blend run meanwhile).
 * Returns 0: frame complete; info_host->num_rendered is exact, info_host->max_tile_instances is the
 *   longest list of the most recent FINISHED frame of this host thread (a hint for the next guess).
 * Returns 1 (GCR_RETRY_RENDER): *info_host exact, nothing rendered yet (binning_capacity too small, or
 *   "force_radix"): allocate gcr_binning_bytes(info_host->num_rendered) and call gcr_forward_render.
 * Returns <0: error. */
#define GCR_RETRY_RENDER 1
int gcr_forward(const gcr_camera *cam, const gcr_gaussians *g, void *geom, size_t geom_bytes,
                void *binning, size_t binning_bytes, int64_t binning_capacity,
                int64_t tile_list_capacity, void *img, size_t img_bytes, int32_t *radii,
                float *out_color, gcr_frame_info *info_host, void *hip_stream);

/* gcr_forward without the host wait (ABI v6).  The reference blocks its caller in every frame until it has read
 * num_rendered back (cr/rasterizer_impl.cu:236-238), although its Python callers never look at the number
 * (dgr/__init__.py:404-420 keeps it for the backward only).  Here the whole frame is enqueued and the call returns;
 * the host thread goes on to enqueue the next frames while this one runs.
 *   words_host  eight 64-bit words of pinned host memory from gcr_host_words_alloc(), owned by the caller and not
 *               reused for another frame before this one's ticket is resolved (gcr_ticket_wait / _poll):
 *                 [0] <- (seq << 32 | min(num_rendered, 2^32 - 1)), stored by the device as soon as K1 is done
 *                 [1] <- the frame's longest tile list (a hint for the next frame's tile_list_capacity)
 *                 [2..7] protocol words of the overflow rescue below
 *   seq         a non-zero tag of the caller's choice that tells this frame's store from an earlier one's
 *   binning_capacity  > 0: the caller's guess of num_rendered; `binning` holds gcr_binning_bytes(binning_capacity)
 *               (gcr_binning_bytes_lean(binning_capacity) suffices when cam->backward == 0)
 * When num_rendered turns out larger than binning_capacity the frame is still rendered correctly, IN STREAM ORDER:
 * the first thread of the frame's last kernel (the forward blend) is a gate that a frame that fitted never reaches
 * and that otherwise holds the stream until a rescue thread of the library (started by the first asynchronous call)
 * has rendered the frame on a high-priority stream of its own with a temporary, exactly sized binning buffer
 * (hipMalloc / hipFree, the one place the library allocates device memory) -- so whatever the caller enqueued behind
 * the frame sees the right image.  The gate waits about two seconds (option "gate_polls") for a rescue to START; a
 * gate that gives up lets the stream go on with an unrendered image and the ticket resolves to GCR_ERR_DEVICE; a rescue
 * that comes later than that touches nothing.  Once a rescue has started the gate waits for it sixteen times as long.
 * The state of a rescued frame is NOT in the caller's `binning` buffer: before gcr_backward, call gcr_forward_render with
 * out_color == NULL and a buffer of gcr_binning_bytes(num_rendered).  Until the ticket is resolved the caller keeps
 * geom / img / radii / out_color and the Gaussians' arrays alive or releases them stream-ordered on `hip_stream`
 * (torch's caching allocator does): the rescue reads and writes them only while the gate holds that stream.
 * fork(): the child starts without a rescue thread and without outstanding frames (pthread_atfork handler); it must
 * initialise HIP itself, as with any HIP library.
 * Returns 0, or < 0 on an argument / launch error (nothing useful was enqueued). */
unsigned long long *gcr_host_words_alloc(size_t n_words); /* pinned, coherent, zero-filled; NULL on failure */
void gcr_host_words_free(unsigned long long *words);
int gcr_forward_async(const gcr_camera *cam, const gcr_gaussians *g, void *geom, size_t geom_bytes,
                      void *binning, size_t binning_bytes, int64_t binning_capacity,
                      int64_t tile_list_capacity, void *img, size_t img_bytes, int32_t *radii,
                      float *out_color, unsigned long long *words_host, uint32_t seq, void *hip_stream);
/* Resolve an asynchronous frame's ticket.  _poll never blocks: 0 = resolved (*info_host filled: num_rendered exact,
 * max_tile_instances = the longest list seen so far or 0), 1 = not yet.  _wait blocks (spinning, with a liveness
 * check of `hip_stream` when non-NULL) until the frame has published num_rendered and -- for a frame that needed the
 * rescue -- until the rescue is complete.  Both return GCR_ERR_OVERFLOW / GCR_ERR_DEVICE when the frame failed. */
int gcr_ticket_poll(const unsigned long long *words_host, uint32_t seq, int64_t binning_capacity,
                    gcr_frame_info *info_host);
int gcr_ticket_wait(const unsigned long long *words_host, uint32_t seq, int64_t binning_capacity,
                    void *hip_stream, gcr_frame_info *info_host);
long gcr_rescue_count(void); /* diagnostics: asynchronous frames of this process that needed the rescue so far */
long gcr_rescue_dropped_count(void); /* ... and calls for help that were not answered before their gate gave up */

/* K3 (instance emit) + K4 (depth sort inside every tile, ties in ascending Gaussian index ==
 * the reference's stable radix sort by tile|depth) + K5 (tile ranges) + K6 (blend).
 * out_color is [3,H,W].  *info must be what gcr_forward_preprocess returned.
 * out_color == NULL (ABI v6): state only -- binning, sort and the forward blend's walk with cam->backward == 1, no
 * pixel is stored: what gcr_backward needs after an asynchronous frame overflowed its binning buffer. */
int gcr_forward_render(const gcr_camera *cam, const gcr_gaussians *g, void *geom,
                       size_t geom_bytes, void *binning, size_t binning_bytes, void *img,
                       size_t img_bytes, const gcr_frame_info *info, float *out_color,
                       void *hip_stream);

/* K7 (reverse-walk blend gradient) + K8 (preprocess gradient). dL_dpix is [3,H,W]. */
/* floats per Gaussian of gcr_grads.dL_dconic: 16, or 32 under option "deterministic_backward" (the process-wide
 * value; _opt: with a call's gcr_options applied, NULL = defaults) */
int gcr_grad_record_floats(void);
int gcr_grad_record_floats_opt(const gcr_options *options);

int gcr_backward(const gcr_camera *cam, const gcr_gaussians *g, const int32_t *radii,
                 const void *geom, size_t geom_bytes, const void *binning,
                 size_t binning_bytes, const void *img, size_t img_bytes, int64_t R,
                 const float *dL_dpix, const gcr_grads *grads, void *hip_stream);

/* K0: present[i] = (view-space z > 0.2).  present is uint8[P]. */
int gcr_mark_visible(int32_t P, const float *means3D, const float *view_matrix,
                     const float *proj_matrix, uint8_t *present, void *hip_stream);

/* One-call forward with the reference's resize-callback contract
 * (cr/rasterizer.h:25-27: std::function<char*(size_t)>).  Each callback must return a device
 * buffer of at least `bytes` bytes.  Returns num_rendered (>= 0) or a negative gcr_status. */
typedef void *(*gcr_resize_fn)(void *user, size_t bytes);
int64_t gcr_rasterize_forward(gcr_resize_fn geometry_buffer, void *geometry_user,
                              gcr_resize_fn binning_buffer, void *binning_user,
                              gcr_resize_fn image_buffer, void *image_user,
                              const gcr_camera *cam, const gcr_gaussians *g, float *out_color,
                              int32_t *radii, void *hip_stream);

/* Process-wide DEFAULTS of the per-call gcr_options (the shipping/parity configuration unless changed; a call that
 * carries gcr_camera.options overrides them for itself only):
 *   "force_radix"  1: always use the global LSD radix sort path for binning          default 0
 *   "force_global_cursor" 1: count/scatter with device-scope atomics instead of LDS  default 0
 *                     tile tables (the variant used when T*4 B does not fit in LDS)
 *   "sort_in_blend" 1: the forward blend sorts its own tile when the expected longest list  default 0
 *                     is <= 384 (one launch less: lower frame latency, lower throughput)
 *   "split_preprocess" 1: K1 as two kernels (streaming cull, then exact pass) instead of   default 0
 *                     the fused one (A/B)
 *   "deterministic_backward" 1: gcr_backward accumulates the per-Gaussian blend gradients as 64-bit       default 0
 *                     fixed-point sums (binary point per Gaussian) instead of fp32 atomics: integer addition is associative, so two runs
 *                     give bit-identical gradients whatever order the tiles' waves arrive in (a debug mode: fp32
 *                     atomics -- here and in the reference -- differ in the last bits from run to run).
 *                     gcr_grad_record_floats() then returns 32: size gcr_grads.dL_dconic AFTER setting the option.
 *   "bwd_piece"    entries per backward piece (64..223) of frames rendered with             default 160
 *                     gcr_camera.backward != 0 (include/gcr.h; gcr_internal.h "backward pieces")
 *   "lazy_sort"    1: tile lists longer than 1024 entries are sorted segment by segment, only as far   default 1
 *                     as the forward blend walks them (saturating scenes never read most of a long list); the entries
 *                     behind the last one consumed stay unsorted (gcr_layout.img_tile_lazy says how far the order is
 *                     final).  0: every list is sorted whole, as the reference does.
 *   "bwd_wave_units" 1: the backward blend runs as one wave per (work item, 8x8 quadrant) -- round 4's    default 0
 *                     kernel, and always the deterministic mode's -- instead of one workgroup per (tile, piece) work
 *                     item that gathers a piece's records once for the tile's four quadrant waves and leaves one
 *                     record update per (entry, piece) (A/B)
 *   "band_sort_min" (ABI v9, process-wide only) frames whose capacity guess (binning_capacity) is at least this   default 6 000 000
 *                     many instances renumber K1's survivors by the band of tiles their rectangle starts in before the
 *                     tile-table kernels count and scatter them (two more small launches; a workgroup's instances then
 *                     land next to each other: C5 941-957 -> 1 025-1 071 frames/s, dense stress scene 916 -> 1 229-1 243;
 *                     C3's 1.2 M instances stay below it).  0: every frame; -1: none.  The tile lists are the same
 *                     either way -- the order inside a tile segment before the tile sort is not observable.
 *   "stream_policy" (process-wide only) cache policy of a frame's one-pass streams -- the streaming cull's input loads,   default -1
 *                     the zeros it stores for culled Gaussians, an inference frame's per-pixel outputs.  -1: non-temporal
 *                     for frames of the entry points that wait for num_rendered (gcr_forward, gcr_forward_preprocess)
 *                     while no other such frame of the process is waiting -- the caller's next cull cannot overlap this
 *                     one: C3 through the int-returning binding 5 150 -> 5 330 frames/s, one frame alone 0.262 ->
 *                     0.248 ms -- and the default policy for asynchronous frames, whose culls run side by side and
 *                     live on each other's lines in L2 / Infinity Cache (non-temporal there: 5 740 -> 5 580).
 *                     0: never; 1: always.  The results are the same bits either way.
 *   "timing"       1: record per-stage HIP events (see gcr_get_stage_ms)             default 0
 *   "gate_polls"   polls (about 5 us each) a frame gate waits for an overflow rescue to START before it       default 400000
 *                     gives up and the ticket resolves to GCR_ERR_DEVICE (gcr_forward_async)
 *   "rescue_hold"  1: the rescue thread answers no call for help (test hook for the gate's timeout path)     default 0
 * Returns the previous value or <0 if the name is unknown.  (The v_exp_f32 mode "fast_exp" of ABI <= 6 is gone: on the
 * round-4 forward blend it was slower than the bit-exact exponential it replaced.) */
int gcr_set_option(const char *name, int value);
/* The current process-wide value of an option, without changing it (ABI v7); INT32_MIN if the name is unknown. */
int gcr_get_option(const char *name);

/* Average per-stage device time (ms) on this thread since the previous call,
 * measured with hipEvents on the caller's stream (non-blocking) when gcr_set_option("timing",1).
 * stage ids: 0 preprocess, 1 scan, 2 emit, 3 sort, 4 ranges, 5 blend_fwd, 6 blend_bwd,
 * 7 preprocess_bwd.  Returns the number of stages written. */
int gcr_get_stage_ms(float *ms_out, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* GCR_H_INCLUDED */
