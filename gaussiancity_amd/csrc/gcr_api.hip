// gcr_api.hip -- C ABI of libgcr_hip.so (declared in include/gcr.h): scratch-buffer layout,
// stage sequencing on the caller's HIP stream, error reporting, optional per-stage timing.
// Stage order follows cr/rasterizer_impl.cu:178-283 (forward) and :287-338 (backward).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>

#include <pthread.h>

#include "gcr_internal.h"

namespace {

thread_local std::string g_err;
std::atomic<int> g_timing{0};
std::atomic<int> g_force_radix{0};
std::atomic<int> g_force_global_cursor{0};
std::atomic<int> g_lazy_sort{1};         // 1: long tile lists are sorted segment by segment, as far as the blend walks (gcr_sort.h)
std::atomic<int> g_sort_in_blend{0};     // 1: the forward blend sorts short tile lists itself (lower frame latency, lower throughput)
std::atomic<int> g_split_preprocess{0};  // 1: K1 as two kernels (streaming cull, then exact pass) instead of the fused one
std::atomic<int> g_deterministic{0};  // 1: fixed-point gradient records (order-independent sums), gcr_internal.h
std::atomic<int> g_bwd_wave_units{0};  // 1: the backward blend as one wave per (work item, quadrant) (round 4) instead of one workgroup per item
std::atomic<int> g_rescue_hold{0};  // test hook ("rescue_hold"): 1 = the rescue thread answers no call for help
std::atomic<int> g_gate_polls{400000};  // polls before a frame gate gives up waiting for a rescue to START: about two seconds
                                        // (~5 us per poll: a PCIe round trip + two s_sleep 127); tests shorten it
// Frames whose instance-capacity guess is at least this many instances band-sort their survivors before the tile tables
// (gcr_binning.hip "band sort": two launches more, a scatter that writes whole sectors).  0: every frame; -1: none.
std::atomic<int> g_band_sort_min{GCR_BAND_SORT_MIN_DEFAULT};
std::atomic<int> g_bwd_piece{160};  // entries per backward piece of frames rendered for a backward (gcr_camera.backward)
// Cache policy of a frame's one-pass streams (option "stream_policy"; gcr_preprocess.hip "NT", gcr_blend.hip `nt_out`;
// results are identical either way).  -1 = automatic: non-temporal for the frames of the SYNCHRONOUS entry points
// (gcr_forward, gcr_forward_preprocess: the caller waits for num_rendered, so its next cull cannot start before this one
// has ended) while no other such frame of the process is between its enqueue and that answer; the default policy for
// asynchronous frames, whose culls run side by side and share their lines.  0 = never, 1 = always.
std::atomic<int> g_stream_policy{-1};
std::atomic<int> g_sync_culls{0};  // synchronous frames whose num_rendered has not reached the host yet
struct SyncCullGuard {             // (counts one such frame for the lifetime of the object)
  const int others;
  SyncCullGuard() : others(g_sync_culls.fetch_add(1)) {}
  ~SyncCullGuard() { g_sync_culls.fetch_sub(1); }
  SyncCullGuard(const SyncCullGuard&) = delete;
  SyncCullGuard& operator=(const SyncCullGuard&) = delete;
};
#ifdef GCR_EXPERIMENTS  // make EXTRA=-DGCR_EXPERIMENTS: timing experiments, never in the shipping library
std::atomic<int> g_k7_skip_flush{0};  // K7 drops its global atomics: results are wrong when set
std::atomic<int> g_k6_debug{0};  // forward blend knock-outs (gcr_blend.hip GCR_K6_*)
std::atomic<unsigned long long*> g_clock_buf{nullptr};  // K7 per-wave phase clocks (gcr_debug_set_clock_buffer)
#endif

// One call's options: the process-wide defaults above, overridden field by field by gcr_camera.options (ABI v6).
// Resolved once at the top of every entry point and handed down by value -- nothing below reads the globals.
struct Opts {
  int lazy_sort, sort_in_blend, bwd_piece, deterministic, split_preprocess, force_radix, force_global_cursor, bwd_wave_units;
  int band_sort_min;  // process-wide only (no field in gcr_options: the record keeps its size)
  int stream_policy;  // process-wide only
};
Opts resolve_options(const gcr_options* o) {
  Opts r;
  r.lazy_sort = g_lazy_sort.load();
  r.sort_in_blend = g_sort_in_blend.load();
  r.bwd_piece = g_bwd_piece.load();
  r.deterministic = g_deterministic.load();
  r.split_preprocess = g_split_preprocess.load();
  r.force_radix = g_force_radix.load();
  r.force_global_cursor = g_force_global_cursor.load();
  r.bwd_wave_units = g_bwd_wave_units.load();
  r.band_sort_min = g_band_sort_min.load();
  r.stream_policy = g_stream_policy.load();
  if (o != nullptr) {
    if (o->bwd_wave_units >= 0) r.bwd_wave_units = o->bwd_wave_units != 0;
    if (o->lazy_sort >= 0) r.lazy_sort = o->lazy_sort != 0;
    if (o->sort_in_blend >= 0) r.sort_in_blend = o->sort_in_blend != 0;
    if (o->bwd_piece >= 0)
      r.bwd_piece = o->bwd_piece < GCR_PIECE_MIN ? GCR_PIECE_MIN : (o->bwd_piece > GCR_PIECE_MAX ? GCR_PIECE_MAX : o->bwd_piece);
    if (o->deterministic_backward >= 0) r.deterministic = o->deterministic_backward != 0;
    if (o->split_preprocess >= 0) r.split_preprocess = o->split_preprocess != 0;
    if (o->force_radix >= 0) r.force_radix = o->force_radix != 0;
    if (o->force_global_cursor >= 0) r.force_global_cursor = o->force_global_cursor != 0;
  }
  return r;
}

enum Stage { ST_PRE = 0, ST_SCAN, ST_EMIT, ST_SORT, ST_RANGES, ST_BLEND_FWD, ST_BLEND_BWD, ST_PRE_BWD, ST_COUNT };

int fail(gcr_status code, const std::string& msg) {
  g_err = msg;
  return (int)code;
}
// gcr_camera.prefiltered and a Gaussian behind the near plane (gcr_internal.h GCR_PREFILTER_MARK): the reference's words
int fail_prefiltered() {
  return fail(GCR_ERR_INVALID_ARGUMENT, "Point is filtered although prefiltered is set. This shouldn't happen!");
}
int fail_hip(hipError_t e, const char* where) {
  g_err = std::string(where) + ": " + hipGetErrorString(e);
  return (int)GCR_ERR_DEVICE;
}
#define HIP_TRY(expr, where)                      \
  do {                                            \
    hipError_t _e = (expr);                       \
    if (_e != hipSuccess) return fail_hip(_e, where); \
  } while (0)

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// spin-wait hint for the polling loop of gcr_forward
inline void gcr_cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  __asm__ __volatile__("yield");
#endif
}

// Longest tile list (the caller's expectation, or the exact maximum on the staged path) up to which the forward
// blend sorts its tiles itself.  Its fast path covers lists of one blend chunk (256); a somewhat longer list is
// still ranked correctly by the same workgroup through global loads (n^2/256 compares per thread), so the
// threshold leaves room for the longest list to grow from one frame to the next.
constexpr int64_t GCR_SORT_IN_BLEND_MAX = 384;

// Stage timer: pairs of hipEvents per stage recorded on the caller's stream, a ring of STAGE_RING pairs per stage.
// Non-blocking in practice: a pair is resolved when its ring slot comes round again, STAGE_RING frames later, and no
// caller keeps that many frames in flight (one pair per stage made the host wait for the PREVIOUS frame's stage
// before it could enqueue this frame's -- with three frames in flight that wait thinned out the overlap and the
// timed kernels looked a third shorter than a rocprofv3 trace of the uninstrumented loop shows them).
// gcr_get_stage_ms() resolves what is pending and reports the average per stage since the last call.
constexpr int STAGE_RING = 8;
struct StageSlot {
  hipEvent_t a[STAGE_RING] = {}, b[STAGE_RING] = {};
  bool pending[STAGE_RING] = {};
  int next = 0;
  double sum_ms = 0.0;
  long count = 0;
};
thread_local StageSlot g_slots[ST_COUNT];

void stage_resolve(StageSlot& sl, int i) {
  if (!sl.pending[i]) return;
  float ms = 0;
  if (hipEventSynchronize(sl.b[i]) == hipSuccess && hipEventElapsedTime(&ms, sl.a[i], sl.b[i]) == hipSuccess) {
    sl.sum_ms += ms;
    sl.count += 1;
  }
  sl.pending[i] = false;
}

struct StageTimer {
  hipStream_t s;
  StageSlot* sl = nullptr;
  int i = 0;
  StageTimer(hipStream_t s_, int stage) : s(s_) {
    if (g_timing.load() == 0) return;
    sl = &g_slots[stage];
    i = sl->next;
    sl->next = (i + 1) % STAGE_RING;
    if (!sl->a[i]) {
      (void)hipEventCreate(&sl->a[i]);
      (void)hipEventCreate(&sl->b[i]);
    }
    stage_resolve(*sl, i);
    (void)hipEventRecord(sl->a[i], s);
  }
  ~StageTimer() {
    if (!sl) return;
    (void)hipEventRecord(sl->b[i], s);
    sl->pending[i] = true;
  }
};

int debug_sync(const gcr_camera* cam, hipStream_t s, const char* where) {
  if (cam->debug) {  // cr/auxiliary.h:158-167
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail_hip(e, where);
  }
  return 0;
}

void compute_layout(int32_t P, int32_t W, int32_t H, int64_t R, gcr_layout* L) {
  const size_t p = (size_t)(P > 0 ? P : 0);
  const size_t nblk = (p + 255) / 256;
  size_t o = 0;
  L->geom_rec = o;            o = align_up(o + p * sizeof(float4) * GCR_REC_QUADS);
  L->geom_cov3D = o;          o = align_up(o + p * GCR_COV3D_FLOATS * sizeof(float));
  L->geom_clamped = o;        o = align_up(o + p);
  L->geom_tiles_touched = o;  o = align_up(o + p * sizeof(uint32_t));
  L->geom_block_sums = o;     o = align_up(o + (nblk + 1) * sizeof(uint32_t));
  L->geom_vis_list = o;       o = align_up(o + p * sizeof(uint32_t));
  L->geom_vis_count = o;      o = align_up(o + (nblk + 1) * sizeof(uint32_t));
  L->geom_num_rendered = o;   o = align_up(o + 16 * sizeof(uint64_t));  // frame words (gcr_internal.h: GCR_FRAME_*, nine in use)
  L->geom_block_tiles = o;    o = align_up(o + GCR_K1_MAX_BLOCKS * sizeof(uint64_t));  // K1 blocks' shares of R
  L->geom_vis_rec = o;        o = align_up(o + p * sizeof(uint4));  // the survivors' binning records (ABI v9)
  L->geom_total = o;

  const size_t npix = (size_t)(W > 0 ? W : 0) * (size_t)(H > 0 ? H : 0);
  const size_t gx = (size_t)(W + GCR_BLOCK_X - 1) / GCR_BLOCK_X, gy = (size_t)(H + GCR_BLOCK_Y - 1) / GCR_BLOCK_Y;
  const size_t T = gx * gy;
  o = 0;
  L->img_final_T = o;    o = align_up(o + npix * sizeof(float));
  L->img_n_contrib = o;  o = align_up(o + npix * sizeof(uint32_t));
  L->img_ranges = o;       o = align_up(o + T * 2 * sizeof(uint32_t));
  L->img_tile_cursor = o;  o = align_up(o + T * GCR_CURSOR_STRIDE * sizeof(uint32_t));
  {
    int G = 1;
    const int ng = gcr_tile_table_groups((int)T, GCR_K1_MAX_BLOCKS, &G);  // upper bound on groups
    L->img_tile_table = o;   o = align_up(o + (size_t)ng * T * sizeof(uint32_t));
  }
  L->img_tile_lazy = o;    o = align_up(o + T * 4 * sizeof(uint32_t));
  L->img_total = o;

  const size_t r = (size_t)(R > 0 ? R : 0);
  const int end_bit = 32 + (int)gcr_higher_msb((uint32_t)T);
  // The sorted instance list sits at offset 0 whatever R is, so a forward that carved the
  // buffer by *capacity* and a backward that carves it by the actual R find it in the same place.
  L->bin_sorted = (size_t)(gcr_sort_passes(end_bit) & 1);
  o = 0;
  L->bin_vals[L->bin_sorted] = o;      o = align_up(o + r * sizeof(uint32_t));
  L->bin_vals[1 - L->bin_sorted] = o;  o = align_up(o + r * sizeof(uint32_t));
  L->bin_keys[0] = o;  o = align_up(o + r * sizeof(uint64_t));
  L->bin_keys[1] = o;  o = align_up(o + r * sizeof(uint64_t));
  L->bin_hist = o;     o = align_up(o + gcr_sort_hist_bytes((int64_t)r, end_bit));
  // (tile, piece) slots of the backward blend, sized for the smallest piece the option "bwd_piece" admits
  const size_t slots = r ? (size_t)gcr_piece_slots(r, T, GCR_PIECE_MIN) : 0;
  L->bin_lean_total = o;  // a frame that never sees gcr_backward needs nothing behind this point
  L->bin_work = o;       o = align_up(o + slots * 16);
  L->bin_mask = o;       o = align_up(o + r * sizeof(uint16_t));
  L->bin_ckpt = o;       o = align_up(o + slots * (size_t)GCR_CKPT_BYTES);
  L->bin_staged = o;     o = align_up(o + r * 48);
  L->bin_total = o;
}

int check_inputs(const gcr_camera* cam, const gcr_gaussians* g, bool need_opacity = true) {
  if (!cam || !g) return fail(GCR_ERR_INVALID_ARGUMENT, "null camera/gaussians record");
  if (g->P < 0) return fail(GCR_ERR_INVALID_ARGUMENT, "P must be >= 0");
  if (g->P > 700000000)  // the kernels index [3 * i + 2] in 32-bit arithmetic (7e8 Gaussians = 165 GB of inputs at SH0)
    return fail(GCR_ERR_INVALID_ARGUMENT, "P above 700 000 000 is not supported");
  if (cam->img_w <= 0 || cam->img_h <= 0) return fail(GCR_ERR_INVALID_ARGUMENT, "image size must be positive");
  if (cam->img_w > 16 * 65535 || cam->img_h > 16 * 65535)
    return fail(GCR_ERR_INVALID_ARGUMENT, "image too large for 16-bit tile coordinates");
  if (!cam->bg || !cam->view_matrix || !cam->proj_matrix || !cam->campos)
    return fail(GCR_ERR_INVALID_ARGUMENT, "bg/view_matrix/proj_matrix/campos must be non-null");
  if (cam->out_u8 != 0 && cam->backward == 1)
    return fail(GCR_ERR_INVALID_ARGUMENT, "out_u8 (uint8 video frames) is for inference frames: gcr_camera.backward must be 0");
  if (cam->win_w != 0 || cam->win_h != 0) {
    if (cam->win_w <= 0 || cam->win_h <= 0 || cam->win_x < 0 || cam->win_y < 0 || cam->win_x + cam->win_w > cam->img_w ||
        cam->win_y + cam->win_h > cam->img_h)
      return fail(GCR_ERR_INVALID_ARGUMENT, "the output window must lie inside the image");
  }
  if (g->P == 0) return 0;
  if (!g->means3D) return fail(GCR_ERR_INVALID_ARGUMENT, "means3D must have dimensions (num_points, 3)");
  if (need_opacity && !g->opacities) return fail(GCR_ERR_INVALID_ARGUMENT, "opacities must be non-null");
  if ((g->shs == nullptr) == (g->colors_precomp == nullptr))
    return fail(GCR_ERR_INVALID_ARGUMENT, "provide exactly one of SHs or precomputed colors");
  const bool has_sr = g->scales != nullptr && g->rotations != nullptr;
  if (has_sr == (g->cov3D_precomp != nullptr) || ((g->scales != nullptr) != (g->rotations != nullptr)))
    return fail(GCR_ERR_INVALID_ARGUMENT,
                "provide exactly one of scale/rotation pair or precomputed 3D covariance");
  if (g->cull_cache && ((uintptr_t)g->cull_cache & 127u))
    return fail(GCR_ERR_INVALID_ARGUMENT, "cull_cache must be 128-byte aligned (gcr_cull_cache_bytes(P) bytes filled by gcr_build_cull_cache)");
  if (g->shs) {
    if (cam->sh_degree < 0 || cam->sh_degree > 3) return fail(GCR_ERR_INVALID_ARGUMENT, "sh_degree must be 0..3");
    if (g->M < (cam->sh_degree + 1) * (cam->sh_degree + 1))
      return fail(GCR_ERR_INVALID_ARGUMENT, "sh has fewer coefficients than sh_degree needs");
  }
  return 0;
}

}  // namespace

extern "C" {

#ifdef GCR_EXPERIMENTS
// experiment builds only, not declared in include/gcr.h: device buffer of (grid x 4 x 10) u64 for K7's phase clocks
void gcr_debug_set_clock_buffer(void* dev_ptr) { g_clock_buf.store((unsigned long long*)dev_ptr); }
#endif

int gcr_abi_version(void) { return GCR_ABI_VERSION; }
int gcr_grad_record_floats(void) { return g_deterministic.load() ? GCR_GRAD_REC_FLOATS_DET : GCR_GRAD_REC_FLOATS; }
int gcr_grad_record_floats_opt(const gcr_options* options) {
  return resolve_options(options).deterministic ? GCR_GRAD_REC_FLOATS_DET : GCR_GRAD_REC_FLOATS;
}
const char* gcr_last_error(void) { return g_err.c_str(); }

size_t gcr_geometry_bytes(int32_t P) {
  gcr_layout L;
  compute_layout(P, 16, 16, 0, &L);
  return L.geom_total;
}
size_t gcr_image_bytes(int32_t W, int32_t H) {
  gcr_layout L;
  compute_layout(0, W, H, 0, &L);
  return L.img_total;
}
size_t gcr_binning_bytes(int64_t R, int32_t W, int32_t H) {
  gcr_layout L;
  compute_layout(0, W, H, R, &L);
  return L.bin_total;
}
size_t gcr_binning_bytes_lean(int64_t R, int32_t W, int32_t H) {
  gcr_layout L;
  compute_layout(0, W, H, R, &L);
  return L.bin_lean_total;
}
int gcr_get_layout(int32_t P, int32_t W, int32_t H, int64_t R, gcr_layout* out) {
  if (!out) return fail(GCR_ERR_INVALID_ARGUMENT, "null layout");
  compute_layout(P, W, H, R, out);
  return 0;
}

int gcr_set_option(const char* name, int value) {
  if (!name) return -1;
  if (!strcmp(name, "bwd_wave_units")) return g_bwd_wave_units.exchange(value != 0);
  if (!strcmp(name, "rescue_hold")) return g_rescue_hold.exchange(value != 0);
  if (!strcmp(name, "gate_polls")) return g_gate_polls.exchange(value < 1 ? 1 : value);
  if (!strcmp(name, "timing")) return g_timing.exchange(value);
  if (!strcmp(name, "band_sort_min")) return g_band_sort_min.exchange(value < 0 ? -1 : value);
  if (!strcmp(name, "stream_policy")) return g_stream_policy.exchange(value < 0 ? -1 : (value != 0));
  if (!strcmp(name, "force_radix")) return g_force_radix.exchange(value);
  if (!strcmp(name, "force_global_cursor")) return g_force_global_cursor.exchange(value);
#ifdef GCR_EXPERIMENTS
  if (!strcmp(name, "k7_skip_flush")) return g_k7_skip_flush.exchange(value);
  if (!strcmp(name, "k6_debug")) return g_k6_debug.exchange(value);
#endif
  if (!strcmp(name, "split_preprocess")) return g_split_preprocess.exchange(value);
  if (!strcmp(name, "sort_in_blend")) return g_sort_in_blend.exchange(value);
  if (!strcmp(name, "lazy_sort")) return g_lazy_sort.exchange(value != 0);
  if (!strcmp(name, "deterministic_backward")) return g_deterministic.exchange(value != 0);
  if (!strcmp(name, "bwd_piece")) {
    const int v = value < GCR_PIECE_MIN ? GCR_PIECE_MIN : (value > GCR_PIECE_MAX ? GCR_PIECE_MAX : value);
    return g_bwd_piece.exchange(v);
  }
  return -1;
}

int gcr_get_option(const char* name) {
  if (!name) return INT32_MIN;
  if (!strcmp(name, "bwd_wave_units")) return g_bwd_wave_units.load();
  if (!strcmp(name, "rescue_hold")) return g_rescue_hold.load();
  if (!strcmp(name, "gate_polls")) return g_gate_polls.load();
  if (!strcmp(name, "timing")) return g_timing.load();
  if (!strcmp(name, "band_sort_min")) return g_band_sort_min.load();
  if (!strcmp(name, "stream_policy")) return g_stream_policy.load();
  if (!strcmp(name, "force_radix")) return g_force_radix.load();
  if (!strcmp(name, "force_global_cursor")) return g_force_global_cursor.load();
  if (!strcmp(name, "split_preprocess")) return g_split_preprocess.load();
  if (!strcmp(name, "sort_in_blend")) return g_sort_in_blend.load();
  if (!strcmp(name, "lazy_sort")) return g_lazy_sort.load();
  if (!strcmp(name, "deterministic_backward")) return g_deterministic.load();
  if (!strcmp(name, "bwd_piece")) return g_bwd_piece.load();
#ifdef GCR_EXPERIMENTS
  if (!strcmp(name, "k7_skip_flush")) return g_k7_skip_flush.load();
  if (!strcmp(name, "k6_debug")) return g_k6_debug.load();
#endif
  return INT32_MIN;
}

int gcr_get_stage_ms(float* ms_out, int capacity) {
  int n = capacity < (int)ST_COUNT ? capacity : (int)ST_COUNT;
  for (int i = 0; i < n; i++) {
    StageSlot& sl = g_slots[i];
    for (int k = 0; k < STAGE_RING; k++) stage_resolve(sl, k);
    ms_out[i] = sl.count ? (float)(sl.sum_ms / (double)sl.count) : 0.0f;
    sl.sum_ms = 0.0;
    sl.count = 0;
  }
  return n;
}

// gcr_camera.host_camera: the camera constants travel in the kernels' argument blocks
static void fill_cam(GcrCamVals& c, const gcr_camera* cam) {
  memset(&c, 0, sizeof(c));
  if (!cam->host_camera) return;
  c.by_value = 1;
  memcpy(c.view, cam->view_matrix, sizeof(c.view));
  memcpy(c.proj, cam->proj_matrix, sizeof(c.proj));
  memcpy(c.campos, cam->campos, sizeof(c.campos));
  memcpy(c.bg, cam->bg, sizeof(c.bg));
}
static inline int stride_or(int32_t s, int dense) { return s > 0 ? (int)s : dense; }

// Enqueues K1 + tile counting + tile scan; leaves {R, longest list, go flag} in the geometry
// buffer (*frame_dev_out).  cap_* only influence the go flag used by speculative launches.
// `host_R` (optional): pinned word that receives (seq << 32 | num_rendered) as soon as K1 is done.
// May the band-sorted survivor numbering use gcr_layout.geom_tiles_touched on this frame?
static inline bool band_sort_slot_free(const Opts& op, int T) {
  return !op.split_preprocess && !op.force_radix && !op.force_global_cursor && gcr_band_sort_possible(T);
}

static int enqueue_preprocess(const Opts& op, const gcr_camera* cam, const gcr_gaussians* g, void* geom, size_t geom_bytes,
                              void* img, size_t img_bytes, int32_t* radii, unsigned long long cap_instances,
                              unsigned long long cap_list, unsigned long long** frame_dev_out, hipStream_t s,
                              unsigned long long* host_R = nullptr, unsigned int seq = 0, bool nt_stream = false) {
  if (!geom || !radii || !img) return fail(GCR_ERR_INVALID_ARGUMENT, "geom/img/radii must be non-null");
  gcr_layout L;
  compute_layout(g->P, cam->img_w, cam->img_h, 0, &L);
  if (geom_bytes < L.geom_total) return fail(GCR_ERR_BUFFER_TOO_SMALL, "geometry buffer too small");
  if (img_bytes < L.img_total) return fail(GCR_ERR_BUFFER_TOO_SMALL, "image buffer too small");
  char *gb = (char*)geom, *ib = (char*)img;

  GcrPreprocessArgs a;
  a.P = g->P; a.D = cam->sh_degree; a.M = g->M; a.W = cam->img_w; a.H = cam->img_h;
  a.gx = (cam->img_w + GCR_BLOCK_X - 1) / GCR_BLOCK_X;
  a.gy = (cam->img_h + GCR_BLOCK_Y - 1) / GCR_BLOCK_Y;
  const int T = a.gx * a.gy;
  a.tanfovx = cam->tanfovx; a.tanfovy = cam->tanfovy;
  a.focal_y = cam->img_h / (2.0f * cam->tanfovy);  // cr/rasterizer_impl.cu:189-190
  a.focal_x = cam->img_w / (2.0f * cam->tanfovx);
  a.scale_modifier = cam->scale_modifier;
  a.means3D = g->means3D; a.scales = g->scales; a.rotations = g->rotations;
  a.opacities = g->opacities; a.shs = g->shs; a.cov3D_precomp = g->cov3D_precomp;
  a.colors_precomp = g->colors_precomp;
  a.view = cam->view_matrix; a.proj = cam->proj_matrix; a.campos = cam->campos;
  a.s_mean = stride_or(g->stride_means3D, 3);
  a.s_opac = stride_or(g->stride_opacities, 1);
  a.s_col = stride_or(g->stride_colors, 3);
  a.s_scale = stride_or(g->stride_scales, 3);
  a.s_rot = stride_or(g->stride_rotations, 4);
  a.prefiltered = cam->prefiltered != 0;
  a.cull_cache = op.split_preprocess ? nullptr : reinterpret_cast<const float4*>(g->cull_cache);
  a.cull_shape = a.cull_cache ? reinterpret_cast<const float4*>((const char*)g->cull_cache + gcr_cull_cache_offset_b(g->P)) : nullptr;
  a.nt_stream = nt_stream ? 1 : 0;
  fill_cam(a.cam, cam);
  a.radii = radii;
  a.rec = (float4*)(gb + L.geom_rec);
  a.cov3D = (float*)(gb + L.geom_cov3D);
  a.tile_count = (uint32_t*)(ib + L.img_tile_cursor);
  a.vis_list = (uint32_t*)(gb + L.geom_vis_list);
  a.vis_count = (uint32_t*)(gb + L.geom_vis_count);
  a.vis_rec = (uint4*)(gb + L.geom_vis_rec);
  // candidate list / counts of K1a live in the arrays only the radix fallback needs later
  a.cand_list = (uint32_t*)(gb + L.geom_tiles_touched);
  a.cand_count = (uint32_t*)(gb + L.geom_block_sums);
  gcr_preprocess_grid(g->P, gcr_preprocess_resident_blocks(op.split_preprocess != 0), &a.nblocks, &a.chunk);
  unsigned long long* frame = (unsigned long long*)(gb + L.geom_num_rendered);
  a.block_tiles = (unsigned long long*)(gb + L.geom_block_tiles);
  *frame_dev_out = frame;
  int G = 1;
  const int NG = op.force_global_cursor ? 0 : gcr_tile_table_groups(T, a.nblocks, &G);
  uint32_t* cursor = (uint32_t*)(ib + L.img_tile_cursor);
  uint32_t* ranges = (uint32_t*)(ib + L.img_ranges);
  if (NG > 0) {
    // default: per-tile counts are built in LDS tables after K1 (no global atomics)
    a.tile_count = nullptr;
    {
      StageTimer t(s, ST_PRE);
      HIP_TRY(gcr_launch_preprocess(a, op.split_preprocess != 0, s), "preprocess");
    }
    if (int rc = debug_sync(cam, s, "preprocess")) return rc;
    StageTimer t(s, ST_SCAN);
    // Band sort for frames expected to hold many instances (the caller's capacity guess: ~0 = no guess, the staged entry
    // point): the renumbered survivors go where only K1a's candidates (split mode) and the radix fallback keep anything
    const bool band = band_sort_slot_free(op, T) && op.band_sort_min >= 0 &&
                      (op.band_sort_min == 0 || (cap_instances != ~0ull && cap_instances >= (unsigned long long)op.band_sort_min));
    // tile_total | tile_local | blk_total share the (T x 128 B) cursor region, unused on this path
    HIP_TRY(gcr_launch_tile_count(T, a.gx, NG, G, a.nblocks, a.chunk, a.vis_rec, a.vis_count,
                                  (uint32_t*)(ib + L.img_tile_table), cursor, cursor + (size_t)T,
                                  cursor + 2 * (size_t)T, frame, a.block_tiles, host_R, seq,
                                  band ? (uint4*)(gb + L.geom_tiles_touched) : nullptr, g->P / 4, s),
            "tile count");
  } else {
    {
      StageTimer t(s, ST_PRE);
      HIP_TRY(hipMemsetAsync(a.tile_count, 0, sizeof(uint32_t) * GCR_CURSOR_STRIDE * (size_t)T, s), "tile count memset");
      HIP_TRY(gcr_launch_preprocess(a, op.split_preprocess != 0, s), "preprocess");
    }
    if (int rc = debug_sync(cam, s, "preprocess")) return rc;
    StageTimer t(s, ST_SCAN);
    HIP_TRY(gcr_launch_scan_tiles(cursor, GCR_CURSOR_STRIDE, ranges, T, frame, cap_instances, cap_list, host_R, seq,
                                  a.block_tiles, a.nblocks, s),
            "tile scan");
  }
  return 0;
}

// Where the forward blend leaves its checkpoints (gcr_internal.h "backward pieces").  `binning` may be null when
// nothing can be rendered (R_layout == 0): the tiles are empty then and the backward never launches its blend.
static void set_piece_args(const Opts& op, GcrBlendArgs& b, bool want_state, const gcr_layout& L, void* binning,
                           void* geom) {
  char* bb = want_state ? (char*)binning : nullptr;  // no state: the blend runs its instantiation without it
  // Measured (tools/k7_ab.py, round 5: one workgroup per item): 160-entry pieces, C2 backward blend 60 us against 63 at
  // 128 and 67 at 223, the forward blend 31.4 against 32.5 / 31.0 (profiles/r05_k7_ab.jsonl).  Pieces cost the
  // forward blend a few per cent (more staging rounds, more sentinel steps, the checkpoint stores), so only frames
  // announced as training frames pay for them.
  b.piece = want_state ? op.bwd_piece : GCR_PIECE_MAX;
#ifdef GCR_EXPERIMENTS
  if (!want_state)  // A/B: the chunk of inference frames (a saturating tile stages a whole chunk and consumes part of it)
    if (const char* e = getenv("GCR_INFER_PIECE")) b.piece = atoi(e) < GCR_PIECE_MIN ? GCR_PIECE_MIN : (atoi(e) > GCR_PIECE_MAX ? GCR_PIECE_MAX : atoi(e));
#endif
  b.ckpt = bb ? (float4*)(bb + L.bin_ckpt) : nullptr;
  b.work = bb ? (uint4*)(bb + L.bin_work) : nullptr;
  b.mask_out = bb ? (uint16_t*)(bb + L.bin_mask) : nullptr;
  b.staged_out = bb ? (float4*)(bb + L.bin_staged) : nullptr;
  b.staged_off = L.bin_staged;
  b.ckpt_off = L.bin_ckpt;
  b.work_off = L.bin_work;
  b.mask_off = L.bin_mask;
  b.carve_bytes = L.bin_total;
  b.frame_out = (unsigned long long*)((char*)geom + L.geom_num_rendered);
#ifdef GCR_EXPERIMENTS
  if (const char* e = getenv("GCR_K6_NOEXTRAS")) {  // A/B only: what the backward's state costs the forward blend
    if (strchr(e, 'w')) { b.ckpt = nullptr; b.work = nullptr; }
    if (strchr(e, 'm')) { b.mask_out = nullptr; b.staged_out = nullptr; }
  }
#endif
}

// Enqueues scatter + per-tile LDS sort + forward blend (the default binning path).
// `frame_guard` (device {R, max, go}) makes the three kernels no-ops when go == 0.
static int enqueue_render_lds(const Opts& op, const gcr_camera* cam, const gcr_gaussians* g, void* geom, void* binning, void* img,
                              int64_t R_layout, int64_t list_length_hint, bool speculative,
                              unsigned long long cap_instances, unsigned long long cap_list, float* out_color,
                              hipStream_t s, unsigned long long* host_longest = nullptr,
                              unsigned long long* gate_words = nullptr, unsigned int gate_seq = 0, bool nt_out = false) {
  gcr_layout L;
  compute_layout(g->P, cam->img_w, cam->img_h, R_layout, &L);
  char *gb = (char*)geom, *bb = (char*)binning, *ib = (char*)img;
  const int gx = (cam->img_w + GCR_BLOCK_X - 1) / GCR_BLOCK_X;
  const int gy = (cam->img_h + GCR_BLOCK_Y - 1) / GCR_BLOCK_Y;
  const int T = gx * gy;
  const float4* rec = (const float4*)(gb + L.geom_rec);
  const uint32_t* vis_list = (const uint32_t*)(gb + L.geom_vis_list);
  const uint32_t* vis_count = (const uint32_t*)(gb + L.geom_vis_count);
  uint32_t* ranges = (uint32_t*)(ib + L.img_ranges);
  uint64_t* pairs = (uint64_t*)(bb + L.bin_keys[0]);
  uint32_t* list = (uint32_t*)(bb + L.bin_vals[L.bin_sorted]);
  unsigned long long* frame_dev = (unsigned long long*)(gb + L.geom_num_rendered);
  const unsigned long long* frame_guard = speculative ? frame_dev : nullptr;
  int nblocks, chunk;
  gcr_preprocess_grid(g->P, gcr_preprocess_resident_blocks(op.split_preprocess != 0), &nblocks, &chunk);
  int G = 1;
  const int NG = op.force_global_cursor ? 0 : gcr_tile_table_groups(T, nblocks, &G);
  {
    StageTimer t(s, ST_EMIT);
    if (NG > 0) {
      // also rebuilds `ranges` from the block totals, so it runs even when nothing is rendered
      uint32_t* cursor = (uint32_t*)(ib + L.img_tile_cursor);
      HIP_TRY(gcr_launch_tile_scatter(T, gx, NG, G, nblocks, chunk, (const uint4*)(gb + L.geom_vis_rec), vis_count,
                                      (uint32_t*)(ib + L.img_tile_table), cursor, cursor + (size_t)T,
                                      cursor + 2 * (size_t)T, ranges, pairs, frame_dev, cap_instances, cap_list,
                                      host_longest,
                                      band_sort_slot_free(op, T) ? (const uint4*)(gb + L.geom_tiles_touched) : nullptr,
                                      g->P / 4, s),
              "tile scatter");
    } else if (R_layout > 0) {
      HIP_TRY(gcr_launch_scatter_instances(nblocks, chunk, vis_list, vis_count, rec, gx,
                                           (uint32_t*)(ib + L.img_tile_cursor), pairs, ranges, T, frame_guard, s),
              "scatter instances");
    }
  }
  if (int rc = debug_sync(cam, s, "scatter instances")) return rc;
  // Option "sort_in_blend": short lists (the caller's expectation, or the exact maximum on the staged path) are
  // sorted by the blend's own workgroups and K4 does not run as a kernel.  Measured at C3: one frame alone 0.287 ->
  // 0.273 ms (one launch less on the critical path), but 4 870 -> 4 580 frames/s with two frames in flight (the
  // VALU-bound blend grows by 14 us, while the separate latency-bound sort kernel hides behind the other frame's
  // work) -- so it is off by default.
  const bool sort_in_blend = R_layout > 0 && list_length_hint <= GCR_SORT_IN_BLEND_MAX && op.sort_in_blend != 0;
  uint4* lazy = (R_layout > 0 && !sort_in_blend && op.lazy_sort != 0) ? (uint4*)(ib + L.img_tile_lazy) : nullptr;
  if (R_layout > 0 && !sort_in_blend) {
    StageTimer t(s, ST_SORT);
    // LDS of the sort sized for 1.5x the expected longest list (longer ones take its run + merge path)
    HIP_TRY(gcr_launch_tile_sort(ranges, T, pairs, (uint64_t*)(bb + L.bin_keys[1]), list,
                                 list_length_hint + list_length_hint / 2 + 64, frame_guard, lazy, s),
            "tile sort");
  }
  if (int rc = debug_sync(cam, s, "tile sort")) return rc;
  GcrBlendArgs b;
  memset(&b, 0, sizeof(b));
  b.ranges = ranges;
  b.list = list;
  b.rec = rec;
  b.W = cam->img_w; b.H = cam->img_h; b.gx = gx; b.gy = gy;
  b.bg = cam->bg;
  b.flip_x = cam->flip_x != 0; b.flip_y = cam->flip_y != 0;
  b.win_x = cam->win_x; b.win_y = cam->win_y; b.win_w = cam->win_w; b.win_h = cam->win_h;
  b.out_u8 = cam->out_u8 != 0 && cam->backward != 1;
  b.nt_out = nt_out && cam->backward != 1;
  fill_cam(b.cam, cam);
  b.final_T = (float*)(ib + L.img_final_T);
  b.n_contrib = (uint32_t*)(ib + L.img_n_contrib);
  b.out_color = out_color;
  b.frame = frame_guard;
  b.pairs = pairs;
  b.list_out = list;
  b.lazy = lazy;
  set_piece_args(op, b, cam->backward == 1, L, R_layout > 0 ? binning : nullptr, geom);
  b.gate_words = gate_words;  // asynchronous frames: the forward blend is the frame's last kernel and carries the gate
  b.gate_seq = gate_seq;
  b.gate_polls = (unsigned int)g_gate_polls.load();
#ifdef GCR_EXPERIMENTS
  b.debug_flags = g_k6_debug.load();
  b.clock_buf = g_clock_buf.load();  // K6: [tile][8] = {hw id | xcc id << 32, clock at entry, at exit, list length, staged, lists built, walk done, steps}
#endif
#ifdef GCR_EXPERIMENTS  // GCR_CHAIN_BLEND=1: the blends of consecutive frames (whatever their streams) run one after the other --
  // they are VALU-bound and gain nothing from sharing the chip -- so that another frame's K1 / binning kernels fall beside
  // a blend instead of beside their own kind (tools/r05_chain.sh; with GCR_K6_LDS_PAD=2048 there is room for them)
  static const bool chain = getenv("GCR_CHAIN_BLEND") != nullptr && atoi(getenv("GCR_CHAIN_BLEND")) != 0;
  static std::mutex chain_mu;
  static hipEvent_t chain_ev[64] = {};
  static unsigned chain_n = 0;
  if (chain) {
    std::lock_guard<std::mutex> lk(chain_mu);
    if (chain_n > 0) (void)hipStreamWaitEvent(s, chain_ev[(chain_n - 1) % 64], 0);
  }
#endif
  {
    StageTimer t(s, ST_BLEND_FWD);
    HIP_TRY(gcr_launch_blend_fwd(b, sort_in_blend, s), "blend forward");
  }
#ifdef GCR_EXPERIMENTS
  if (chain) {
    std::lock_guard<std::mutex> lk(chain_mu);
    hipEvent_t& e = chain_ev[chain_n % 64];
    if (!e) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    (void)hipEventRecord(e, s);
    chain_n++;
  }
#endif
  return debug_sync(cam, s, "blend forward");
}

// pinned landing zone of the {R, longest list} read-back (per host thread; the synchronous entry point's own words)
// num_rendered reaches the host WITHOUT a copy: the exact projection pass accumulates it and the first workgroup
// of the kernel that follows stores (frame tag << 32 | R) into a pinned, coherent host word the host thread polls.  Compared with
// the reference's blocking cudaMemcpy (cr/rasterizer_impl.cu:236) -- and with round 1's 24-byte async copy +
// event after the column scan -- the host is released as soon as K1 is done (the tile-table kernels, the
// scatter, the sort and the blend of the frame are enqueued but still to run), and a frame costs no
// hipMemcpyAsync / hipEventRecord / hipEventSynchronize calls.  pinned[1] receives the frame's longest tile list
// from the scatter kernel; nobody waits for it, the next call uses it as its length hint.
static unsigned long long* host_words_alloc(size_t n) {
  unsigned long long* p = nullptr;
  // Portable: the same host thread may render on another device later (ext._on_device) and that device's
  // kernels store into the same words
  if (n == 0 || hipHostMalloc((void**)&p, n * sizeof(unsigned long long),
                              hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable) != hipSuccess)
    return nullptr;
  for (size_t i = 0; i < n; i++) p[i] = 0ull;
  return p;
}
struct FrameReadback {
  unsigned long long* pinned = nullptr;  // [0] = seq << 32 | R, [1] = longest list of the most recent scatter
  unsigned int seq = 0;
  int ensure() {
    if (!pinned) {
      pinned = host_words_alloc(8);
      if (!pinned) return fail(GCR_ERR_DEVICE, "hipHostMalloc for the frame read-back failed");
    }
    return 0;
  }
};
thread_local FrameReadback g_readback;

// allocations handed out by gcr_host_words_alloc (pointer -> words): gcr_host_words_free must take the rescue thread's
// eyes off a block before it unmaps it
static std::mutex g_words_mu;
static std::map<unsigned long long*, size_t> g_words_blocks;
static void rescue_forget_words(const unsigned long long* base, size_t n);  // (RescueService, below)

unsigned long long* gcr_host_words_alloc(size_t n_words) {
  unsigned long long* p = host_words_alloc(n_words);
  if (p) {
    std::lock_guard<std::mutex> lk(g_words_mu);
    g_words_blocks[p] = n_words;
  }
  return p;
}
void gcr_host_words_free(unsigned long long* words) {
  if (!words) return;
  size_t n = 0;
  {
    std::lock_guard<std::mutex> lk(g_words_mu);
    auto it = g_words_blocks.find(words);
    if (it != g_words_blocks.end()) {
      n = it->second;
      g_words_blocks.erase(it);
    }
  }
  if (n) rescue_forget_words(words, n);
  (void)hipHostFree(words);
}

int gcr_forward_preprocess(const gcr_camera* cam, const gcr_gaussians* g, void* geom,
                           size_t geom_bytes, void* img, size_t img_bytes, int32_t* radii,
                           gcr_frame_info* info_host, void* hip_stream) {
  if (int rc = check_inputs(cam, g)) return rc;
  if (!info_host) return fail(GCR_ERR_INVALID_ARGUMENT, "info_host is null");
  info_host->num_rendered = 0;
  info_host->max_tile_instances = 0;
  if (g->P == 0) return 0;  // dgr/rasterize_points.cu:71
  const Opts op = resolve_options(cam->options);
  hipStream_t s = (hipStream_t)hip_stream;
  unsigned long long* frame = nullptr;
  const SyncCullGuard sync_cull;  // (until this call returns: the copy below is the frame's host wait)
  const bool nt = op.stream_policy > 0 || (op.stream_policy < 0 && sync_cull.others == 0);
  if (int rc = enqueue_preprocess(op, cam, g, geom, geom_bytes, img, img_bytes, radii, ~0ull, ~0ull, &frame, s, nullptr, 0, nt))
    return rc;
  unsigned long long r[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(r, frame, sizeof(r), hipMemcpyDeviceToHost, s), "num_rendered copy");
  HIP_TRY(hipStreamSynchronize(s), "num_rendered sync");  // cr/rasterizer_impl.cu:236-238
  if (r[0] == GCR_PREFILTER_MARK) return fail_prefiltered();
  if (r[0] > 0x7fffffffull)
    return fail(GCR_ERR_OVERFLOW, "num_rendered exceeds 2^31-1 (32-bit instance index, as in the reference)");
  info_host->num_rendered = (int64_t)r[0];
  info_host->max_tile_instances = (int64_t)r[1];
  return 0;
}

// bytes a forward needs in `binning` for `capacity` instances: the lean carve for frames that never see a backward
static size_t forward_binning_need(const gcr_camera* cam, int64_t capacity) {
  return cam->backward != 1 ? gcr_binning_bytes_lean(capacity, cam->img_w, cam->img_h)
                             : gcr_binning_bytes(capacity, cam->img_w, cam->img_h);
}

int gcr_forward(const gcr_camera* cam, const gcr_gaussians* g, void* geom, size_t geom_bytes, void* binning,
                size_t binning_bytes, int64_t binning_capacity, int64_t tile_list_capacity, void* img,
                size_t img_bytes, int32_t* radii, float* out_color, gcr_frame_info* info_host, void* hip_stream) {
  if (int rc = check_inputs(cam, g)) return rc;
  if (!info_host) return fail(GCR_ERR_INVALID_ARGUMENT, "info_host is null");
  info_host->num_rendered = 0;
  info_host->max_tile_instances = 0;
  if (g->P == 0) return 0;
  if (!out_color) return fail(GCR_ERR_INVALID_ARGUMENT, "out_color is null");
  if (binning_capacity < 0 || binning_capacity > 0x7fffffffll)
    return fail(GCR_ERR_INVALID_ARGUMENT, "binning_capacity out of range");
  if (binning_capacity > 0 && (!binning || binning_bytes < forward_binning_need(cam, binning_capacity)))
    return fail(GCR_ERR_BUFFER_TOO_SMALL, "binning buffer smaller than gcr_binning_bytes(binning_capacity)");
  if (int rc = g_readback.ensure()) return rc;
  const Opts op = resolve_options(cam->options);
  hipStream_t s = (hipStream_t)hip_stream;
  const bool speculate = binning_capacity > 0 && !op.force_radix && !cam->debug;
  // The longest-list guess only sizes the LDS of the tile sort (any length is sorted correctly), so the
  // speculation can only be vetoed by num_rendered exceeding the binning capacity.
  const int64_t list_hint = tile_list_capacity > 0 ? tile_list_capacity : (int64_t)gcr_tile_sort_capacity();
  FrameReadback& rb = g_readback;
  const unsigned int seq = ++rb.seq ? rb.seq : ++rb.seq;  // never 0: the word starts out as 0
  unsigned long long* frame = nullptr;
  // "stream_policy": this caller's next cull cannot start before the wait below has seen this one end
  const SyncCullGuard sync_cull;
  const bool nt = op.stream_policy > 0 || (op.stream_policy < 0 && sync_cull.others == 0);
  if (int rc = enqueue_preprocess(op, cam, g, geom, geom_bytes, img, img_bytes, radii,
                                  speculate ? (unsigned long long)binning_capacity : 0ull, ~0ull, &frame, s,
                                  rb.pinned, seq, nt))
    return rc;
  if (speculate) {
    // Everything else of the frame is enqueued before the host knows R: the kernels read the
    // tile ranges from device memory and are vetoed by frame[2] if the capacity guess was short.
    if (int rc = enqueue_render_lds(op, cam, g, geom, binning, img, binning_capacity, list_hint, true,
                                    (unsigned long long)binning_capacity, ~0ull, out_color, s, rb.pinned + 1,
                                    nullptr, 0, nt))
      return rc;
  }
  // the one host wait of the frame: poll the pinned word until the kernel after K1 has tagged it with this frame
  volatile unsigned long long* word = rb.pinned;
  unsigned long long v = *word;
  for (unsigned long spins = 0; (unsigned int)(v >> 32) != seq; v = *word) {
    gcr_cpu_relax();
    if ((++spins & 0xffffu) == 0) {  // every ~65k polls: make sure the stream is still alive
      const hipError_t q = hipStreamQuery(s);
      if (q != hipSuccess && q != hipErrorNotReady) return fail_hip(q, "frame info wait");
      if (q == hipSuccess && (unsigned int)(*word >> 32) != seq)
        return fail(GCR_ERR_DEVICE, "the stream finished without publishing num_rendered");
    }
  }
  const unsigned long long R = v & 0xffffffffull;
  // the same decision the scatter kernel takes on the device from the same number
  const bool go = R <= (unsigned long long)binning_capacity;
  if (R == GCR_PREFILTER_MARK) return fail_prefiltered();
  if (R > 0x7fffffffull)
    return fail(GCR_ERR_OVERFLOW, "num_rendered exceeds 2^31-1 (32-bit instance index, as in the reference)");
  info_host->num_rendered = (int64_t)R;
  if (speculate && go) {
    // longest list: what the most recent scatter kernel of this thread published (a hint for the next guess);
    // before any has (the first speculative frame returns while its own scatter is still queued) the caller's
    // own expectation is echoed, so that a hint loop does not fall back to "unknown"
    const unsigned long long seen = ((volatile unsigned long long*)rb.pinned)[1];
    info_host->max_tile_instances = seen ? (int64_t)seen : tile_list_capacity;
    return 0;
  }
  // retry path: the caller needs the exact longest list of THIS frame to size the sort
  unsigned long long mx = 0;
  HIP_TRY(hipMemcpyAsync(&mx, frame + 1, sizeof(mx), hipMemcpyDeviceToHost, s), "longest list copy");
  HIP_TRY(hipStreamSynchronize(s), "longest list sync");
  info_host->max_tile_instances = (int64_t)mx;
  return 1;  // GCR_RETRY_RENDER: call gcr_forward_render with a binning buffer sized for info_host
}

// ------------------------------------------------------------------------------------ asynchronous frames + rescue
// gcr_forward_async (include/gcr.h): the frame is enqueued and the call returns.  What the synchronous entry point
// does on the host when the capacity guess was short -- allocate an exactly sized buffer, render again -- is done here
// by a RESCUE THREAD, while the frame's last kernel (the gate in k_blend_fwd's first thread, gcr_blend.hip) holds the
// caller's stream.  Protocol words of a frame (pinned host memory, `seq` = the frame's tag in each):
//   [3] gate -> host: the frame overflowed, every kernel of it has run, the stream is held
//   [7] host -> gate: a rescue has STARTED (from here on it writes geom / img / out_color)
//   [2] host -> gate: release (the rescue is complete, or has failed)
//   [4] gate -> host: gave up waiting for a rescue to start (about two seconds)
//   [5] the rescue failed          [6] the frame was handled by the rescue (its state is not in the caller's buffer)
// [4] and [7] are decided the Dekker way (k6_frame_gate): a rescue that finds [4] set after announcing itself touches
// nothing -- the stream may long have gone on and the buffers may belong to somebody else -- and a gate
// that finds [7] set after giving up goes on waiting for [2] (sixteen times as long: never a device hung for good).
// The thread is started by the first asynchronous call, sleeps while no asynchronous frame is outstanding and
// otherwise looks at the outstanding frames' words every 100 us; a frame that fitted (the overwhelming majority: the
// caller's guess carries a margin) leaves the list as soon as its num_rendered is known; any entry older than 120 s
// (a stream that died, a gate that never launched) is dropped.
struct AsyncFrame {
  unsigned long long* words;
  unsigned int seq;
  int device;
  gcr_camera cam;
  gcr_options opt;
  bool has_opt;
  float camvals[38];  // view 16 | proj 16 | campos 3 | bg 3 when cam.host_camera (the caller's host arrays may be gone)
  gcr_gaussians g;
  void* geom;
  size_t geom_bytes;
  void* img;
  size_t img_bytes;
  float* out_color;
  int64_t capacity;
  std::chrono::steady_clock::time_point born;
};

class RescueService {
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<AsyncFrame> frames_;
  bool started_ = false;
  hipStream_t streams_[64] = {};
  void* scratch_[64] = {};        // per device: the rescue's binning buffer, kept from one rescue to the next (grow only)
  size_t scratch_bytes_[64] = {};
  std::atomic<long> rescued_{0};
  std::atomic<long> dropped_{0};  // calls for help that came too late (the gate had given up)
  const unsigned long long* rescuing_ = nullptr;  // (mu_) words of the frame rescue() is working on: already off frames_,
  std::condition_variable idle_cv_;               //   still written to -- forget() waits for it

  static void publish(unsigned long long* w, int idx, unsigned int seq) {
    std::atomic_thread_fence(std::memory_order_seq_cst);
    ((volatile unsigned long long*)w)[idx] = (unsigned long long)seq;
    std::atomic_thread_fence(std::memory_order_seq_cst);
  }

  void rescue(AsyncFrame& f) {
    volatile unsigned long long* w = f.words;
    // announce, THEN look whether the gate has given up (k6_frame_gate does the same the other way round)
    publish(f.words, 7, f.seq);
    if ((unsigned int)w[4] == f.seq) {  // too late: the stream has gone on (or is about to) -- nothing may be written
      publish(f.words, 5, f.seq);
      publish(f.words, 2, f.seq);  // (a gate that saw word 7 after giving up is waiting for this)
      dropped_.fetch_add(1);
      return;
    }
    const unsigned long long R = w[0] & 0xffffffffull;
    bool ok = false;
    void* stale = nullptr;  // a scratch buffer that was too small: freed AFTER the gate is released (hipFree waits for
    //                         every stream of the device, the caller's -- which the gate is holding -- included)
    do {
      if (R > 0x7fffffffull) break;
      if (hipSetDevice(f.device) != hipSuccess) break;
      const int d = f.device & 63;
      if (!streams_[d]) {
        // highest priority: a queue of its own where the runtime has one -- the rescue's kernels must never sit in a
        // hardware queue BEHIND the gate that is waiting for them (streams are multiplexed onto a few queues)
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = hi = 0;
        if (hipStreamCreateWithPriority(&streams_[d], hipStreamNonBlocking, hi) != hipSuccess &&
            hipStreamCreateWithFlags(&streams_[d], hipStreamNonBlocking) != hipSuccess)
          break;
      }
      hipStream_t s = streams_[d];
      gcr_camera cam = f.cam;
      cam.backward = 0;  // the temporary buffer dies with the rescue: no backward state (gcr_forward_render with
      //                    out_color == NULL rebuilds it in a buffer of the caller's when a backward follows)
      cam.options = f.has_opt ? &f.opt : nullptr;
      if (cam.host_camera) {
        cam.view_matrix = f.camvals;
        cam.proj_matrix = f.camvals + 16;
        cam.campos = f.camvals + 32;
        cam.bg = f.camvals + 35;
      }
      gcr_layout L;
      compute_layout(f.g.P, cam.img_w, cam.img_h, 0, &L);
      gcr_frame_info info;
      info.num_rendered = (int64_t)R;
      unsigned long long longest = 0;
      if (hipMemcpyAsync(&longest, (char*)f.geom + L.geom_num_rendered + 8, 8, hipMemcpyDeviceToHost, s) != hipSuccess) break;
      if (hipStreamSynchronize(s) != hipSuccess) break;
      info.max_tile_instances = (int64_t)longest;
      const size_t bytes = gcr_binning_bytes_lean((int64_t)R, cam.img_w, cam.img_h);
      if (scratch_bytes_[d] < bytes) {
        stale = scratch_[d];
        scratch_[d] = nullptr;
        scratch_bytes_[d] = 0;
        if (hipMalloc(&scratch_[d], bytes + bytes / 2) != hipSuccess) {
          scratch_[d] = nullptr;
          break;
        }
        scratch_bytes_[d] = bytes + bytes / 2;
      }
      if (gcr_forward_render(&cam, &f.g, f.geom, f.geom_bytes, scratch_[d], scratch_bytes_[d], f.img, f.img_bytes, &info,
                             f.out_color, s) < 0)
        break;
      if (hipStreamSynchronize(s) != hipSuccess) break;
      ok = true;
    } while (false);
    if (!ok) publish(f.words, 5, f.seq);
    publish(f.words, 6, f.seq);
    publish(f.words, 2, f.seq);  // releases the gate
    rescued_.fetch_add(1);
    if (stale) (void)hipFree(stale);
  }

  void run() {
    for (;;) {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return !frames_.empty(); });
      const auto now = std::chrono::steady_clock::now();
      const bool hold = g_rescue_hold.load() != 0;
      for (auto it = frames_.begin(); it != frames_.end();) {
        volatile unsigned long long* w = it->words;
        const unsigned long long v = w[0];
        if ((unsigned int)(v >> 32) == it->seq) {
          if ((v & 0xffffffffull) <= (unsigned long long)it->capacity) {  // fitted: its gate lets the stream pass
            it = frames_.erase(it);
            continue;
          }
          if ((unsigned int)w[4] == it->seq) {  // its gate has given up: the frame is lost, the ticket says so
            dropped_.fetch_add(1);
            it = frames_.erase(it);
            continue;
          }
          if (!hold && (unsigned int)w[3] == it->seq) {  // the gate is holding the stream: every kernel of the frame is done
            AsyncFrame f = *it;
            frames_.erase(it);
            rescuing_ = f.words;
            lk.unlock();
            rescue(f);
            lk.lock();
            rescuing_ = nullptr;
            idle_cv_.notify_all();
            break;  // iterators are gone: rescan on the next round
          }
        }
        if (now - it->born > std::chrono::seconds(120)) {  // a stream that died, a gate that never launched: stop looking
          it = frames_.erase(it);
          continue;
        }
        ++it;
      }
      lk.unlock();
      std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
  }

 public:
  void add(const AsyncFrame& f) {
    std::lock_guard<std::mutex> lk(mu_);
    // the caller reuses a word set only for a frame whose ticket it has resolved: an older entry on the same words is
    // finished (or abandoned)
    for (auto it = frames_.begin(); it != frames_.end();) it = it->words == f.words ? frames_.erase(it) : it + 1;
    const bool was_empty = frames_.empty();
    frames_.push_back(f);
    if (!started_) {
      started_ = true;
      std::thread([this] { run(); }).detach();
    }
    if (was_empty) cv_.notify_one();  // (with frames outstanding the thread is polling, not waiting)
  }
  void forget(const unsigned long long* base, size_t n) {  // gcr_host_words_free: these words are about to be unmapped
    std::unique_lock<std::mutex> lk(mu_);
    for (auto it = frames_.begin(); it != frames_.end();) it = (it->words >= base && it->words < base + n) ? frames_.erase(it) : it + 1;
    // a frame that was handed to rescue() before this call is off the list but its words are still being written
    // (words 7 / 5 / 6 / 2): the caller may not unmap them under the rescue thread
    idle_cv_.wait(lk, [&] { return !(rescuing_ != nullptr && rescuing_ >= base && rescuing_ < base + n); });
  }
  void remove(const unsigned long long* words, unsigned int seq) {  // a frame whose enqueue failed after add()
    std::lock_guard<std::mutex> lk(mu_);
    for (auto it = frames_.begin(); it != frames_.end();) it = (it->words == words && it->seq == seq) ? frames_.erase(it) : it + 1;
  }
  long rescued() const { return rescued_.load(); }
  long dropped() const { return dropped_.load(); }
  // fork(): the child has this object's memory but not its thread (and no usable HIP context).  The handler below runs in
  // the child: a fresh service state, so that a child that initialises HIP itself and renders asynchronously starts a
  // thread of its own instead of trusting `started_`.  (The mutex may have been held by the parent's thread at the
  // moment of the fork: it is re-constructed in place, never unlocked.)
  void reset_after_fork() {
    new (&mu_) std::mutex();
    new (&cv_) std::condition_variable();
    new (&idle_cv_) std::condition_variable();
    rescuing_ = nullptr;
    new (&frames_) std::deque<AsyncFrame>();
    started_ = false;
    for (int i = 0; i < 64; i++) {
      streams_[i] = nullptr;
      scratch_[i] = nullptr;
      scratch_bytes_[i] = 0;
    }
  }
};
// never destroyed: the detached thread may outlive static destruction at process exit
static RescueService& rescue_service() {
  static RescueService* r = [] {
    RescueService* p = new RescueService();
    (void)pthread_atfork(nullptr, nullptr, [] { rescue_service().reset_after_fork(); });
    return p;
  }();
  return *r;
}

static void rescue_forget_words(const unsigned long long* base, size_t n) { rescue_service().forget(base, n); }

long gcr_rescue_count(void) { return rescue_service().rescued(); }  // diagnostics: frames that needed the rescue so far
long gcr_rescue_dropped_count(void) { return rescue_service().dropped(); }  // ... and calls for help that came after the gate had given up

int gcr_forward_async(const gcr_camera* cam, const gcr_gaussians* g, void* geom, size_t geom_bytes, void* binning,
                      size_t binning_bytes, int64_t binning_capacity, int64_t tile_list_capacity, void* img,
                      size_t img_bytes, int32_t* radii, float* out_color, unsigned long long* words_host, uint32_t seq,
                      void* hip_stream) {
  if (int rc = check_inputs(cam, g)) return rc;
  if (!words_host || seq == 0u) return fail(GCR_ERR_INVALID_ARGUMENT, "words_host must be non-null and seq non-zero");
  if (g->P == 0) {  // dgr/rasterize_points.cu:71: nothing to do; the ticket resolves to 0 at once
    ((volatile unsigned long long*)words_host)[1] = 0ull;
    ((volatile unsigned long long*)words_host)[0] = (unsigned long long)seq << 32;
    return 0;
  }
  if (!out_color) return fail(GCR_ERR_INVALID_ARGUMENT, "out_color is null");
  const Opts op = resolve_options(cam->options);
  if (binning_capacity <= 0 || binning_capacity > 0x7fffffffll || op.force_radix || cam->debug)
    return fail(GCR_ERR_INVALID_ARGUMENT,
                "gcr_forward_async needs a capacity guess > 0 and neither force_radix nor debug: use gcr_forward");
  if (!binning || binning_bytes < forward_binning_need(cam, binning_capacity))
    return fail(GCR_ERR_BUFFER_TOO_SMALL, "binning buffer smaller than gcr_binning_bytes(binning_capacity)");
  hipStream_t s = (hipStream_t)hip_stream;
  const int64_t list_hint = tile_list_capacity > 0 ? tile_list_capacity : (int64_t)gcr_tile_sort_capacity();
  unsigned long long* frame = nullptr;
  // (asynchronous frames are what runs side by side: their streams keep the default cache policy unless "stream_policy" is 1)
  if (int rc = enqueue_preprocess(op, cam, g, geom, geom_bytes, img, img_bytes, radii, (unsigned long long)binning_capacity,
                                  ~0ull, &frame, s, words_host, seq, op.stream_policy > 0))
    return rc;
  AsyncFrame f;
  memset((void*)&f, 0, sizeof(f));
  f.words = words_host;
  f.seq = seq;
  if (hipGetDevice(&f.device) != hipSuccess) f.device = 0;
  f.cam = *cam;
  f.has_opt = cam->options != nullptr;
  if (f.has_opt) f.opt = *cam->options;
  f.cam.options = nullptr;
  if (cam->host_camera) {
    memcpy(f.camvals, cam->view_matrix, 16 * sizeof(float));
    memcpy(f.camvals + 16, cam->proj_matrix, 16 * sizeof(float));
    memcpy(f.camvals + 32, cam->campos, 3 * sizeof(float));
    memcpy(f.camvals + 35, cam->bg, 3 * sizeof(float));
  }
  f.g = *g;
  f.geom = geom;
  f.geom_bytes = geom_bytes;
  f.img = img;
  f.img_bytes = img_bytes;
  f.out_color = out_color;
  f.capacity = binning_capacity;
  f.born = std::chrono::steady_clock::now();
  rescue_service().add(f);  // (before the kernel that may call for it is enqueued)
  const int rc = enqueue_render_lds(op, cam, g, geom, binning, img, binning_capacity, list_hint, true,
                                    (unsigned long long)binning_capacity, ~0ull, out_color, s, words_host + 1, words_host, seq,
                                    op.stream_policy > 0);
  if (rc < 0) rescue_service().remove(words_host, seq);  // no gate was enqueued: nobody will ever call for this frame
  return rc;
}

static int ticket_state(const unsigned long long* words_host, uint32_t seq, int64_t capacity, gcr_frame_info* info) {
  volatile const unsigned long long* w = words_host;
  const unsigned long long v = w[0];
  if ((unsigned int)(v >> 32) != seq) return 1;
  const unsigned long long R = v & 0xffffffffull;
  if (R == GCR_PREFILTER_MARK) return fail_prefiltered();
  if (R > 0x7fffffffull)
    return fail(GCR_ERR_OVERFLOW, "num_rendered exceeds 2^31-1 (32-bit instance index, as in the reference)");
  if (R > (unsigned long long)capacity) {  // the frame needed the rescue: resolved when that is complete
    if ((unsigned int)w[5] == seq) return fail(GCR_ERR_DEVICE, "the overflow rescue of an asynchronous frame failed");
    if ((unsigned int)w[2] != seq) {
      if ((unsigned int)w[4] == seq && (unsigned int)w[7] != seq)  // (with word 7 set the gate is waiting for word 2)
        return fail(GCR_ERR_DEVICE, "an asynchronous frame overflowed its binning buffer and was not rescued in time");
      return 1;
    }
  }
  if (info) {
    info->num_rendered = (int64_t)R;
    info->max_tile_instances = (int64_t)w[1];
  }
  return 0;
}

int gcr_ticket_poll(const unsigned long long* words_host, uint32_t seq, int64_t binning_capacity,
                    gcr_frame_info* info_host) {
  if (!words_host || seq == 0u) return fail(GCR_ERR_INVALID_ARGUMENT, "words_host must be non-null and seq non-zero");
  return ticket_state(words_host, seq, binning_capacity, info_host);
}

int gcr_ticket_wait(const unsigned long long* words_host, uint32_t seq, int64_t binning_capacity, void* hip_stream,
                    gcr_frame_info* info_host) {
  if (!words_host || seq == 0u) return fail(GCR_ERR_INVALID_ARGUMENT, "words_host must be non-null and seq non-zero");
  hipStream_t s = (hipStream_t)hip_stream;
  for (unsigned long spins = 0;; ) {
    const int st = ticket_state(words_host, seq, binning_capacity, info_host);
    if (st != 1) return st;
    gcr_cpu_relax();
    if ((++spins & 0xfffffu) == 0 && hip_stream != nullptr) {  // every ~1M polls: is the stream still alive?
      const hipError_t q = hipStreamQuery(s);
      if (q != hipSuccess && q != hipErrorNotReady) return fail_hip(q, "ticket wait");
      if (q == hipSuccess && ticket_state(words_host, seq, binning_capacity, info_host) == 1)
        return fail(GCR_ERR_DEVICE, "the stream finished without publishing num_rendered");
    }
  }
}

int gcr_forward_render(const gcr_camera* cam, const gcr_gaussians* g, void* geom, size_t geom_bytes,
                       void* binning, size_t binning_bytes, void* img, size_t img_bytes,
                       const gcr_frame_info* info, float* out_color, void* hip_stream) {
  // (binning, sort and blend read the geometry state K1 left, none of the Gaussians' input arrays: a backward that
  // re-renders a frame's state has no opacities to hand in, cr/rasterizer.h:39-48)
  if (int rc = check_inputs(cam, g, false)) return rc;
  if (g->P == 0) return 0;
  if (!geom || !img || !info) return fail(GCR_ERR_INVALID_ARGUMENT, "geom/img/info must be non-null");
  if (!out_color && cam->backward != 1)
    return fail(GCR_ERR_INVALID_ARGUMENT, "out_color may only be NULL for a state-only pass (gcr_camera.backward == 1)");
  const int64_t R = info->num_rendered;
  if (R < 0 || R > 0x7fffffffll) return fail(GCR_ERR_INVALID_ARGUMENT, "R out of range");
  if (R > 0 && !binning) return fail(GCR_ERR_INVALID_ARGUMENT, "binning buffer is null");
  gcr_layout L;
  compute_layout(g->P, cam->img_w, cam->img_h, R, &L);
  if (geom_bytes < L.geom_total) return fail(GCR_ERR_BUFFER_TOO_SMALL, "geometry buffer too small");
  if (img_bytes < L.img_total) return fail(GCR_ERR_BUFFER_TOO_SMALL, "image buffer too small");
  if (R > 0 && binning_bytes < (cam->backward == 1 ? L.bin_total : L.bin_lean_total))
    return fail(GCR_ERR_BUFFER_TOO_SMALL, "binning buffer too small");
  const Opts op = resolve_options(cam->options);
  hipStream_t s = (hipStream_t)hip_stream;
  // Tile lists beyond the LDS sort capacity are sorted by the same workgroup with runs + merge passes; every
  // other tile stays on the LDS path (no whole-frame fallback for a long list).
  if (R == 0 || !op.force_radix)
    return enqueue_render_lds(op, cam, g, geom, binning, img, R, info->max_tile_instances, false, ~0ull, ~0ull, out_color, s,
                              nullptr, nullptr, 0, op.stream_policy > 0);

  // "force_radix" (A/B and test option): the reference's own scheme -- emit tile|depth keys in index
  // order, stable global radix sort, boundary scan.
  char *gb = (char*)geom, *bb = (char*)binning, *ib = (char*)img;
  const int gx = (cam->img_w + GCR_BLOCK_X - 1) / GCR_BLOCK_X;
  const int gy = (cam->img_h + GCR_BLOCK_Y - 1) / GCR_BLOCK_Y;
  const int T = gx * gy;
  const float4* rec = (const float4*)(gb + L.geom_rec);
  uint32_t* tiles_touched = (uint32_t*)(gb + L.geom_tiles_touched);
  uint32_t* block_sums = (uint32_t*)(gb + L.geom_block_sums);
  const uint32_t* vis_list = (const uint32_t*)(gb + L.geom_vis_list);
  const uint32_t* vis_count = (const uint32_t*)(gb + L.geom_vis_count);
  int nblocks, chunk;
  gcr_preprocess_grid(g->P, gcr_preprocess_resident_blocks(op.split_preprocess != 0), &nblocks, &chunk);
  uint64_t* k0 = (uint64_t*)(bb + L.bin_keys[0]);
  uint64_t* k1 = (uint64_t*)(bb + L.bin_keys[1]);
  uint32_t* v0 = (uint32_t*)(bb + L.bin_vals[0]);
  uint32_t* v1 = (uint32_t*)(bb + L.bin_vals[1]);
  uint32_t* ranges = (uint32_t*)(ib + L.img_ranges);
  int half = (int)L.bin_sorted;
  {
    StageTimer t(s, ST_EMIT);
    unsigned long long* scratch_total = (unsigned long long*)(bb + L.bin_hist);
    HIP_TRY(gcr_launch_tiles_touched(g->P, nblocks, chunk, vis_list, vis_count, rec, tiles_touched, block_sums, s),
            "tiles touched");
    HIP_TRY(gcr_launch_scan_block_sums(block_sums, (g->P + 255) / 256, scratch_total, s), "scan");
    HIP_TRY(gcr_launch_emit(g->P, tiles_touched, block_sums, rec, gx, k0, v0, s), "emit");
  }
  if (int rc = debug_sync(cam, s, "emit")) return rc;
  {
    StageTimer t(s, ST_SORT);
    const int end_bit = 32 + (int)gcr_higher_msb((uint32_t)T);  // cr/rasterizer_impl.cu:252
    HIP_TRY(gcr_launch_sort(k0, v0, k1, v1, R, end_bit, (uint32_t*)(bb + L.bin_hist), &half, s), "sort");
  }
  if (int rc = debug_sync(cam, s, "sort")) return rc;
  {
    StageTimer t(s, ST_RANGES);
    HIP_TRY(gcr_launch_tile_ranges(half ? k1 : k0, R, ranges, T, s), "tile ranges");
  }
  if (int rc = debug_sync(cam, s, "tile ranges")) return rc;
  GcrBlendArgs b;
  memset(&b, 0, sizeof(b));
  b.ranges = ranges;
  b.list = half ? v1 : v0;
  b.rec = rec;
  b.W = cam->img_w; b.H = cam->img_h; b.gx = gx; b.gy = gy;
  b.bg = cam->bg;
  b.flip_x = cam->flip_x != 0; b.flip_y = cam->flip_y != 0;
  b.win_x = cam->win_x; b.win_y = cam->win_y; b.win_w = cam->win_w; b.win_h = cam->win_h;
  fill_cam(b.cam, cam);
  b.final_T = (float*)(ib + L.img_final_T);
  b.n_contrib = (uint32_t*)(ib + L.img_n_contrib);
  b.out_color = out_color;
  b.out_u8 = cam->out_u8 != 0 && cam->backward != 1;
  set_piece_args(op, b, cam->backward == 1, L, binning, geom);
  {
    StageTimer t(s, ST_BLEND_FWD);
    HIP_TRY(gcr_launch_blend_fwd(b, false, s), "blend forward");
  }
  return debug_sync(cam, s, "blend forward");
}

int gcr_backward(const gcr_camera* cam, const gcr_gaussians* g, const int32_t* radii, const void* geom,
                 size_t geom_bytes, const void* binning, size_t binning_bytes, const void* img,
                 size_t img_bytes, int64_t R, const float* dL_dpix, const gcr_grads* gr,
                 void* hip_stream) {
  // opacities are not an input of the backward (cr/rasterizer.h:39-48): read from geom state
  if (int rc = check_inputs(cam, g, false)) return rc;
  if (g->P == 0) return 0;
  if (!gr || !dL_dpix || !radii || !geom || !img)
    return fail(GCR_ERR_INVALID_ARGUMENT, "grads/dL_dpix/radii/geom/img must be non-null");
  if (!gr->dL_dmeans2D || !gr->dL_dconic || !gr->dL_dopacity || !gr->dL_dcolors || !gr->dL_dmeans3D ||
      !gr->dL_dcov3D || (g->shs && !gr->dL_dsh) || (g->scales && (!gr->dL_dscales || !gr->dL_drotations)))
    return fail(GCR_ERR_INVALID_ARGUMENT, "a required gradient output is null");
  // the accumulation records are read and written as 16-byte quads of 64-byte records, dL_drotations as float4
  if ((uintptr_t)gr->dL_dconic & 63u)
    return fail(GCR_ERR_INVALID_ARGUMENT, "dL_dconic (the gradient records) must be 64-byte aligned");
  if (g->scales && stride_or(gr->stride_rotations, 4) == 4 && ((uintptr_t)gr->dL_drotations & 15u))
    return fail(GCR_ERR_INVALID_ARGUMENT, "dL_drotations must be 16-byte aligned");
  if (g->rotations && stride_or(g->stride_rotations, 4) == 4 && ((uintptr_t)g->rotations & 15u))
    return fail(GCR_ERR_INVALID_ARGUMENT, "rotations must be 16-byte aligned when dense");
  const bool any_grad_stride = gr->stride_means3D > 0 || gr->stride_opacity > 0 || gr->stride_colors > 0 ||
                               gr->stride_scales > 0 || gr->stride_rotations > 0;
  if (any_grad_stride && (!gr->packed || gr->packed_floats <= 0))
    return fail(GCR_ERR_INVALID_ARGUMENT, "strided gradient outputs need the block they live in (gcr_grads.packed)");
  if (R < 0 || R > 0x7fffffffll) return fail(GCR_ERR_INVALID_ARGUMENT, "R out of range");
  if (cam->backward != 1)
    return fail(GCR_ERR_INVALID_ARGUMENT,
                "gcr_backward needs a frame rendered with gcr_camera.backward == 1 (pass the same value here); for a frame "
                "rendered without it call gcr_forward_render(out_color = NULL, backward = 1) first");
  gcr_layout L;
  compute_layout(g->P, cam->img_w, cam->img_h, R, &L);
  if (geom_bytes < L.geom_total) return fail(GCR_ERR_BUFFER_TOO_SMALL, "geometry buffer too small");
  if (img_bytes < L.img_total) return fail(GCR_ERR_BUFFER_TOO_SMALL, "image buffer too small");
  // (the forward may have carved the buffer for a larger capacity: the blend kernel checks the carve the forward
  // published against binning_bytes on the device and poisons the gradients instead of reading out of bounds)
  if (R > 0 && (!binning || binning_bytes < L.bin_total))
    return fail(GCR_ERR_BUFFER_TOO_SMALL, "binning buffer too small");
  const Opts op = resolve_options(cam->options);
  hipStream_t s = (hipStream_t)hip_stream;
  const char *gb = (const char*)geom, *bb = (const char*)binning, *ib = (const char*)img;
  const int gx = (cam->img_w + GCR_BLOCK_X - 1) / GCR_BLOCK_X;
  const int gy = (cam->img_h + GCR_BLOCK_Y - 1) / GCR_BLOCK_Y;
  int nblocks_k1 = 0, chunk_k1 = 0;
  gcr_preprocess_grid(g->P, gcr_preprocess_resident_blocks(op.split_preprocess != 0), &nblocks_k1, &chunk_k1);

  const int det = op.deterministic;  // the caller sized dL_dconic with gcr_grad_record_floats_opt() under the same options

  // Every output is written here (the reference asks its caller for nine zero-filled tensors,
  // dgr/rasterize_points.cu:118-126): zeros are streamed over the dense arrays by extra workgroups of the K7
  // launch, K8 then writes the survivors' values.
  GcrFillArgs fill;
  memset(&fill, 0, sizeof(fill));
  {
    const unsigned long long P = (unsigned long long)g->P;
    auto seg = [&](float* p, unsigned long long n) {
      if (p != nullptr && n != 0) {
        fill.ptr[fill.nseg] = p;
        fill.n[fill.nseg++] = n;
      }
    };
    // strided outputs live in gr->packed, which is filled as one block; dense ones are filled one by one
    seg(gr->dL_dmeans2D, 3 * P);
    seg(gr->stride_colors > 0 ? nullptr : gr->dL_dcolors, (unsigned long long)GCR_NUM_CHANNELS * P);
    seg(gr->stride_opacity > 0 ? nullptr : gr->dL_dopacity, P);
    seg(gr->stride_means3D > 0 ? nullptr : gr->dL_dmeans3D, 3 * P);
    seg(gr->dL_dcov3D, 6 * P);
    seg(gr->dL_dsh, g->shs ? 3ull * (unsigned long long)g->M * P : 0ull);
    seg(gr->stride_scales > 0 ? nullptr : gr->dL_dscales, 3 * P);
    seg(gr->stride_rotations > 0 ? nullptr : gr->dL_drotations, 4 * P);
    if (any_grad_stride) seg(gr->packed, (unsigned long long)gr->packed_floats);
    unsigned long long total = 0;
    for (int i = 0; i < fill.nseg; i++) total += fill.n[i];
    const unsigned long long want = (total + 1023ull) / 1024ull;  // >= 16 floats per thread of a 64-thread wave
    // Few waves on purpose: the fill rides in the backward blend's launch as extra one-wave workgroups at the front
    // of its grid.  Measured (round 2, 256-thread workgroups) with 64 / 128 / 256 / 512 / 1024 of them at C2 (142 MB
    // of zeros): 118 / 117 / 121 / 128 / 136 us backward wall time -- more of them only take issue slots and memory
    // queues away from the walking waves.  256 waves = the same 16k threads.
    unsigned long long cap = 256ull;
#ifdef GCR_EXPERIMENTS
    if (const char* e = getenv("GCR_FILL_BLOCKS")) cap = (unsigned long long)atoi(e);
#endif
    fill.blocks = (int)(want < cap ? (want ? want : 1ull) : cap);
  }

  if (R > 0) {
    StageTimer t(s, ST_BLEND_BWD);  // one slot per stage: a second timer of the same stage would halve the average
    HIP_TRY(gcr_launch_zero_grad_records(nblocks_k1, chunk_k1, (const uint32_t*)(gb + L.geom_vis_list),
                                         (const uint32_t*)(gb + L.geom_vis_count), (float4*)gr->dL_dconic,
                                         (det ? GCR_GRAD_REC_FLOATS_DET : GCR_GRAD_REC_FLOATS) / 4, s),
            "zero gradient records");
    GcrBlendArgs b;
    memset(&b, 0, sizeof(b));
    b.fill = fill;
    b.ranges = (const uint32_t*)(ib + L.img_ranges);
    b.list = (const uint32_t*)(bb + L.bin_vals[L.bin_sorted]);
    b.rec = (const float4*)(gb + L.geom_rec);
    b.W = cam->img_w; b.H = cam->img_h; b.gx = gx; b.gy = gy;
    b.bg = cam->bg;
    b.flip_x = cam->flip_x != 0; b.flip_y = cam->flip_y != 0;
    b.win_x = cam->win_x; b.win_y = cam->win_y; b.win_w = cam->win_w; b.win_h = cam->win_h;
    fill_cam(b.cam, cam);
    b.final_T = (float*)(ib + L.img_final_T);
    b.n_contrib = (uint32_t*)(ib + L.img_n_contrib);
    b.dL_dpix = dL_dpix;
    b.grad_rec = gr->dL_dconic;  // [P][gcr_grad_record_floats()] accumulation records (include/gcr.h)
    b.deterministic = det;
    b.binning_base = bb;
    b.frame_in = (const unsigned long long*)(gb + L.geom_num_rendered);
    b.R = (unsigned long long)R;
    b.piece = op.bwd_piece;  // sizes the grid only: the kernel takes the forward's piece size from frame_in
    b.binning_bytes = (unsigned long long)binning_bytes;
#ifdef GCR_EXPERIMENTS
    b.debug_flags = g_k7_skip_flush.load();  // bit 0: no global flush, bit 1: no zero fill, bit 2: no LDS adds
    b.clock_buf = g_clock_buf.load();
#endif
    HIP_TRY(gcr_launch_blend_bwd(b, op.bwd_wave_units != 0, s), "blend backward");
  } else {
    HIP_TRY(gcr_launch_fill(fill, s), "gradient zero fill");
  }
  if (int rc = debug_sync(cam, s, "blend backward")) return rc;

  GcrPreprocessBwdArgs a;
  a.P = g->P; a.D = cam->sh_degree; a.M = g->M; a.W = cam->img_w; a.H = cam->img_h;
  a.tanfovx = cam->tanfovx; a.tanfovy = cam->tanfovy;
  a.focal_y = cam->img_h / (2.0f * cam->tanfovy);
  a.focal_x = cam->img_w / (2.0f * cam->tanfovx);
  a.scale_modifier = cam->scale_modifier;
  a.means3D = g->means3D; a.scales = g->scales; a.rotations = g->rotations; a.shs = g->shs;
  // cr/rasterizer_impl.cu:329-330 reads the forward's stored covariance when none was supplied; K8 derives it from scales and
  // rotation again instead (gcr_preprocess.hip phase_a1_exact: the same function, the same bits) -- the pointer below is only
  // read for a supplied covariance
  a.cov3D = g->cov3D_precomp ? g->cov3D_precomp : (const float*)(gb + L.geom_cov3D);
  a.s_cov3d = g->cov3D_precomp ? 6 : GCR_COV3D_FLOATS;
  a.view = cam->view_matrix; a.proj = cam->proj_matrix; a.campos = cam->campos;
  a.radii = radii;
  a.vis_list = (const uint32_t*)(gb + L.geom_vis_list);
  a.vis_count = (const uint32_t*)(gb + L.geom_vis_count);
  a.nblocks = nblocks_k1; a.chunk = chunk_k1;
  a.frame = R > 0 ? (const unsigned long long*)(gb + L.geom_num_rendered) : nullptr;
  a.binning_bytes = (unsigned long long)binning_bytes;
  a.grad_rec = (const float4*)gr->dL_dconic;
  a.rec = (const float4*)(gb + L.geom_rec);
  a.deterministic = det;
  a.dL_dmean2D = gr->dL_dmeans2D; a.dL_dcolor = gr->dL_dcolors; a.dL_dopacity = gr->dL_dopacity;
  a.dL_dmean3D = gr->dL_dmeans3D; a.dL_dcov3D = gr->dL_dcov3D; a.dL_dsh = gr->dL_dsh;
  a.dL_dscale = gr->dL_dscales; a.dL_drot = gr->dL_drotations;
  a.s_mean = stride_or(g->stride_means3D, 3);
  a.s_scale = stride_or(g->stride_scales, 3);
  a.s_rot = stride_or(g->stride_rotations, 4);
  a.g_mean = stride_or(gr->stride_means3D, 3);
  a.g_opac = stride_or(gr->stride_opacity, 1);
  a.g_col = stride_or(gr->stride_colors, 3);
  a.g_scale = stride_or(gr->stride_scales, 3);
  a.g_rot = stride_or(gr->stride_rotations, 4);
  fill_cam(a.cam, cam);
  {
    StageTimer t(s, ST_PRE_BWD);
    HIP_TRY(gcr_launch_preprocess_bwd(a, s), "preprocess backward");
  }
  return debug_sync(cam, s, "preprocess backward");
}

size_t gcr_cull_cache_bytes(int32_t P) { return P > 0 ? gcr_cull_cache_offset_b(P) + (size_t)P * 32 : 0; }

int gcr_build_cull_cache(const gcr_gaussians* g, float scale_modifier, void* cull_cache_out, void* hip_stream) {
  if (!g) return fail(GCR_ERR_INVALID_ARGUMENT, "null gaussians record");
  if (g->P < 0 || g->P > 700000000) return fail(GCR_ERR_INVALID_ARGUMENT, "P must be 0 .. 700 000 000");
  if (g->P == 0) return 0;
  if (!g->means3D) return fail(GCR_ERR_INVALID_ARGUMENT, "means3D must have dimensions (num_points, 3)");
  if (!g->opacities) return fail(GCR_ERR_INVALID_ARGUMENT, "opacities must be non-null");
  const bool has_sr = g->scales != nullptr && g->rotations != nullptr;
  if (has_sr == (g->cov3D_precomp != nullptr) || ((g->scales != nullptr) != (g->rotations != nullptr)))
    return fail(GCR_ERR_INVALID_ARGUMENT, "provide exactly one of scale/rotation pair or precomputed 3D covariance");
  if (!cull_cache_out || ((uintptr_t)cull_cache_out & 127u))
    return fail(GCR_ERR_INVALID_ARGUMENT, "cull_cache_out must be a 128-byte aligned buffer of gcr_cull_cache_bytes(P) bytes");
  GcrPreprocessArgs a;
  memset(&a, 0, sizeof(a));
  a.P = g->P;
  a.scale_modifier = scale_modifier;
  a.means3D = g->means3D; a.scales = g->scales; a.rotations = g->rotations; a.cov3D_precomp = g->cov3D_precomp;
  a.opacities = g->opacities;
  a.s_mean = stride_or(g->stride_means3D, 3);
  a.s_scale = stride_or(g->stride_scales, 3);
  a.s_rot = stride_or(g->stride_rotations, 4);
  a.s_opac = stride_or(g->stride_opacities, 1);
  HIP_TRY(gcr_launch_build_cull_cache(a, reinterpret_cast<float4*>(cull_cache_out),
                                      reinterpret_cast<float4*>((char*)cull_cache_out + gcr_cull_cache_offset_b(g->P)),
                                      (hipStream_t)hip_stream),
          "build cull cache");
  return 0;
}

int gcr_mark_visible(int32_t P, const float* means3D, const float* view_matrix,
                     const float* proj_matrix, uint8_t* present, void* hip_stream) {
  (void)proj_matrix;  // the reference's in_frustum only uses it for the commented-out side test
  if (P < 0) return fail(GCR_ERR_INVALID_ARGUMENT, "P must be >= 0");
  if (P == 0) return 0;
  if (!means3D || !view_matrix || !present) return fail(GCR_ERR_INVALID_ARGUMENT, "null pointer");
  HIP_TRY(gcr_launch_mark_visible(P, means3D, view_matrix, present, (hipStream_t)hip_stream), "mark_visible");
  return 0;
}

int64_t gcr_rasterize_forward(gcr_resize_fn geometry_buffer, void* geometry_user,
                              gcr_resize_fn binning_buffer, void* binning_user,
                              gcr_resize_fn image_buffer, void* image_user, const gcr_camera* cam,
                              const gcr_gaussians* g, float* out_color, int32_t* radii,
                              void* hip_stream) {
  if (int rc = check_inputs(cam, g)) return rc;
  if (!geometry_buffer || !binning_buffer || !image_buffer)
    return fail(GCR_ERR_INVALID_ARGUMENT, "resize callbacks must be non-null");
  if (g->P == 0) return 0;
  const size_t gbytes = gcr_geometry_bytes(g->P);
  void* geom = geometry_buffer(geometry_user, gbytes);
  if (!geom) return fail(GCR_ERR_ALLOC, "geometry resize callback returned null");
  const size_t ibytes = gcr_image_bytes(cam->img_w, cam->img_h);
  void* img = image_buffer(image_user, ibytes);
  if (!img) return fail(GCR_ERR_ALLOC, "image resize callback returned null");
  // the reference's forward always leaves what its backward needs (cr/rasterizer.h:25-48 has no inference mode): so
  // does this entry point, whatever gcr_camera.backward says
  gcr_camera cam1 = *cam;
  cam1.backward = 1;
  gcr_frame_info info;
  if (int rc = gcr_forward_preprocess(&cam1, g, geom, gbytes, img, ibytes, radii, &info, hip_stream)) return rc;
  const size_t bbytes = gcr_binning_bytes(info.num_rendered, cam->img_w, cam->img_h);
  void* bin = binning_buffer(binning_user, bbytes);
  if (!bin && info.num_rendered > 0) return fail(GCR_ERR_ALLOC, "binning resize callback returned null");
  if (int rc = gcr_forward_render(&cam1, g, geom, gbytes, bin, bbytes, img, ibytes, &info, out_color, hip_stream))
    return rc;
  return info.num_rendered;
}

}  // extern "C"
