/*
 * gcv_oracle.c -- CPU restatement of GaussianCity's point-generation / visibility path
 * (SURVEY.md section 8 row f2): footprint extruder -> points_to_volume -> ray/voxel traversal.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this file; the product (gaussiancity_amd/) never imports, links or calls it.
 *
 * PARITY STATUS
 *   K15 orv_extrude            PINNED: the reference's own extruder is CPU C++ and is compiled in
 *                              place from /root/reference/extensions/footprint_extruder/
 *                              footprint_extruder.cpp into oracle/_ref/ (oracle/Makefile, target
 *                              `ref`); tests/test_points_oracle.py checks this restatement against it
 *                              and tests/golden/points_*.npz hold vectors it produced.
 *   K14 orv_points_to_volume   "parity unpinned": CUDA only (extensions/voxlib/points_to_volume.cu),
 *   K12 orv_rvip               no CPU path, no tests, no fixtures upstream
 *                              (extensions/voxlib/ray_voxel_intersection.cu).  Restated statement by
 *                              statement; K12 additionally checked against an independent float64
 *                              slab-intersection formulation (tests/test_points_oracle.py).
 *
 * NUMERICS (K12): IEEE binary32 in the reference's association order, no contraction
 * (-ffp-contract=off), correctly rounded division and sqrt; "no hit" is the quiet NaN 0x7fc00000
 * (upstream nanf("0") -- the payload CUDA returns is not reproducible on the host, so it is fixed
 * here for both sides).  K14: upstream's overlapping writes race; here the HIGHEST point id wins,
 * which is what a sequential loop in point order produces and what the HIP path implements with
 * atomicMax -- one of the outcomes upstream can produce, made deterministic.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ K15 */
/* segInsMap of footprint_extruder.cpp:90-100,201 (dataset_generator.py:984-1004) */
typedef struct {
  int16_t bldg_ins_min_id, car_ins_min_id, car_semantic_id, bldg_facade_semantic_id, roof_ins_offset;
} orv_seg_ins;

/* footprint_extruder.cpp:90-100 */
static int16_t get_semantic_id(int16_t ins, const orv_seg_ins *m) {
  if (ins < m->bldg_ins_min_id) return ins;
  if (ins >= m->car_ins_min_id) return m->car_semantic_id;
  return m->bldg_facade_semantic_id; /* building labels merge into the facade class */
}

/* footprint_extruder.cpp:102-126.  Index arithmetic as upstream: short operands promote to int,
 * the product/sum is converted to size_t (getArrayIndex, :86-88). */
static int nbr_same(const int16_t *map, int x, int y, int width, int scale) {
  const int16_t c = map[(size_t)(y * width + x)];
  const int dx[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, dy[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
  for (int i = 0; i < 8; i++)
    if (c != map[(size_t)((y + dy[i] * scale) * width + (x + dx[i] * scale))]) return 0;
  return 1;
}

/* footprint_extruder.cpp:128-141 */
static int is_border(int x, int y, int z, int height, int width, int scale, int inc_btm,
                     const int16_t *seg, const int16_t *td, const int16_t *bu) {
  const size_t idx = (size_t)(y * width + x);
  if (z > td[idx] - scale || (z == bu[idx] && inc_btm)) return 1;
  if (x < scale || x >= width - scale - 1 || y < scale || y >= height - scale - 1) return 1;
  return !nbr_same(seg, x, y, width, scale) || !nbr_same(td, x, y, width, scale);
}

/* footprint_extruder.cpp:143-213 (getPointsFromProjection).
 * scale_of_semantic[s] = scales[classes[s]] for s in [0, 32768); upstream's std::map::operator[]
 * yields scale 0 for an unknown semantic id and then never terminates (k += 0) -- here a scale <= 0
 * is an error (return -1 - pixel index).  Points are written in upstream's order (row-major pixels,
 * k ascending) as (x=j, y=i, z=k, scale, instanceID), 5 x int16.  Returns the number of points;
 * only the first `cap` are stored. */
int64_t orv_extrude(int inc_btm, const int16_t *scale_of_semantic, const orv_seg_ins *m, int height,
                    int width, const int16_t *seg, const int16_t *td, const int16_t *bu,
                    const uint8_t *pts, int16_t *out, int64_t cap) {
  int64_t n = 0;
  for (int i = 0; i < height; i++) {
    for (int j = 0; j < width; j++) {
      const size_t idx = (size_t)(i * width + j);
      if (!pts[idx]) continue;
      int16_t ins = seg[idx];
      const int16_t sem = get_semantic_id(ins, m);
      const int16_t scale = (sem >= 0) ? scale_of_semantic[sem] : 0;
      if (scale <= 0) return -1 - (int64_t)idx;
      for (int16_t k = bu[idx]; k <= td[idx]; k = (int16_t)(k + scale)) {
        if (!is_border(j, i, k, height, width, scale, inc_btm, seg, td, bu)) continue; /* hollow */
        if (k > td[idx] - scale && sem == m->bldg_facade_semantic_id)
          ins = (int16_t)(ins + m->roof_ins_offset); /* roof instance id, :200-203 */
        if (n < cap) {
          int16_t *o = out + 5 * n;
          o[0] = (int16_t)j; o[1] = (int16_t)i; o[2] = k; o[3] = scale; o[4] = ins;
        }
        n++;
        if ((int)k + (int)scale > 32767) break; /* upstream would overflow the short loop counter */
      }
    }
  }
  return n;
}

/* ------------------------------------------------------------------------------------ K14 */
/* extensions/voxlib/points_to_volume.cu:21-50.  volume[h][w][d] (d fastest) must be zeroed by the
 * caller (upstream: torch::zeros, :66-67).  Sequential in point order => the highest index wins
 * where cubes overlap. */
void orv_points_to_volume(int64_t n_pts, int h, int w, int d, const int16_t *points,
                          const int32_t *pt_ids, const int16_t *scales, int32_t *volume) {
  for (int64_t idx = 0; idx < n_pts; idx++) {
    const int32_t pid = pt_ids[idx];
    const int16_t x = points[3 * idx], y = points[3 * idx + 1], z = points[3 * idx + 2];
    const int16_t sx = scales[3 * idx], sy = scales[3 * idx + 1], sz = scales[3 * idx + 2];
    if (x >= w || y >= h || z >= d || x < 0 || y < 0 || z < 0) continue;
    for (int j = x; j < x + sx && j < w; ++j)
      for (int k = y; k < y + sy && k < h; ++k)
        for (int l = z; l < z + sz && l < d; ++l)
          volume[(int64_t)k * w * d + (int64_t)j * d + l] = pid;
  }
}

/* ------------------------------------------------------------------------------------ K13 */
/* extensions/voxlib/maps_to_volume.cu:21-101 ("parity unpinned": CUDA only, and no caller in the reference's
 * own Python).  Constants as upstream (:16-19): buildings start at instance 10 here (not 100), merge into
 * semantic class 2, roof = instance + 1; depth is 504 upstream and a parameter here.  One int16 per voxel:
 * the instance id at (y, x, k) for every k of a border column -- no cube, unlike points_to_volume.
 * Upstream writes volume[.. + k] without looking at depth (out of bounds for k >= 504 or k < 0); here such k
 * are skipped.  scales[sem] <= 0 or sem >= n_scales: error -2 - pixel (upstream: endless loop / OOB read). */
int64_t orv_maps_to_volume(int height, int width, int depth, const int8_t *scales, int n_scales, const int16_t *inst_map,
                           const int16_t *td_hf, const int16_t *bu_hf, const uint8_t *pts_map, int16_t *volume) {
  memset(volume, 0, sizeof(int16_t) * (size_t)height * width * depth);
  for (int j = 0; j < height; j++)
    for (int i = 0; i < width; i++) {
      const size_t px = (size_t)j * width + i;
      if (!pts_map[px]) continue;
      const int hgt_up = td_hf[px], hgt_lw = bu_hf[px];
      const int16_t inst = inst_map[px];
      const int sem = inst < 10 ? inst : 2;
      if (sem < 0 || sem >= n_scales || scales[sem] <= 0) return -2 - (int64_t)px;
      const int scale = scales[sem];
      int16_t *col = volume + px * depth;
      const int edge = (i < scale) || (i >= width - scale - 1) || (j < scale) || (j >= height - scale - 1);
      int border = edge;
      if (!edge) border = !nbr_same(td_hf, i, j, width, scale) || !nbr_same(inst_map, i, j, width, scale);
      for (int k = hgt_lw; k <= hgt_up; k += scale) {
        const int top = k > hgt_up - scale;
        if (!top && !border) continue; /* hollow */
        if (k < 0 || k >= depth) continue;
        col[k] = (int16_t)((top && sem == 2) ? inst + 1 : inst);
      }
    }
  return 0;
}

/* ------------------------------------------------------------------------------------ K12 */
/* voxlib_common.h:56-82 */
static void normalize3(float *a) {
  float len = 0.0f;
  for (int i = 0; i < 3; i++) len += a[i] * a[i];
  len = sqrtf(len);
  for (int i = 0; i < 3; i++) a[i] /= len;
}
/* voxlib_common.h:31-36 */
static void cross3(float *r, const float *a, const float *b) {
  r[0] = a[1] * b[2] - a[2] * b[1];
  r[1] = a[2] * b[0] - a[0] * b[2];
  r[2] = a[0] * b[1] - a[1] * b[0];
}

static float quiet_nan(void) {
  const uint32_t bits = 0x7fc00000u;
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

/* Host part of ray_voxel_intersection_perspective_cuda (ray_voxel_intersection.cu:256-266):
 * camera frame in world space.  frame = fwd[3] side[3] up[3]. */
void orv_camera_frame(const float *cam_dir, const float *cam_up, float *frame) {
  float *fwd = frame, *side = frame + 3, *up = frame + 6;
  for (int i = 0; i < 3; i++) fwd[i] = cam_dir[i];
  normalize3(fwd);
  cross3(side, fwd, cam_up);
  normalize3(side);
  cross3(up, side, fwd);
  normalize3(up);
}

/* Device part (ray_voxel_intersection.cu:54-216), one call per pixel.  One DDA step of axis a. */
#define ORV_STEP(a)                                                                  \
  do {                                                                               \
    tnow = axis_t[a];                                                                \
    if (raydir[a] > 0) {                                                             \
      axis_int[a] += 1;                                                              \
      if (axis_int[a] >= dims[a]) quit = 1;                                          \
      axis_t[a] = ((float)(axis_int[a] + 1) - rayori[a]) / raydir[a];                \
    } else {                                                                         \
      axis_int[a] -= 1;                                                              \
      if (axis_int[a] < 0) quit = 1;                                                 \
      axis_t[a] = ((float)axis_int[a] - rayori[a]) / raydir[a];                      \
    }                                                                                \
  } while (0)

void orv_rvip(const int32_t *in_voxel, const int *dims, const int64_t *strides, const float *cam_ori,
              const float *cam_dir, const float *cam_up, float cam_f, const float *cam_c,
              const int *img_dims, int max_samples, int32_t *out_voxel_id, float *out_depth,
              float *out_raydirs) {
  float frame[9];
  orv_camera_frame(cam_dir, cam_up, frame);
  const float *fwd = frame, *side = frame + 3, *up = frame + 6;
  const int64_t npix = (int64_t)img_dims[0] * img_dims[1];
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t pix = 0; pix < npix; pix++) {
    const int r = (int)(pix / img_dims[1]), c = (int)(pix % img_dims[1]);
    float rayori[3] = {cam_ori[0], cam_ori[1], cam_ori[2]}, raydir[3];
    const float n0 = cam_c[0] - (float)r; /* flip height */
    const float n1 = (float)c - cam_c[1];
    for (int i = 0; i < 3; i++) raydir[i] = up[i] * n0 + side[i] * n1 + fwd[i] * cam_f;
    normalize3(raydir);
    out_raydirs[pix * 3] = raydir[0];
    out_raydirs[pix * 3 + 1] = raydir[1];
    out_raydirs[pix * 3 + 2] = raydir[2];
    float axis_t[3];
    int axis_int[3];
    for (int i = 0; i < 3; i++) axis_int[i] = (int)floorf(rayori[i]);
    for (int i = 0; i < 3; i++) {
      if (raydir[i] > 0)
        axis_t[i] = ((float)(axis_int[i] + 1) - rayori[i]) / raydir[i];
      else if (raydir[i] < 0)
        axis_t[i] = ((float)axis_int[i] - rayori[i]) / raydir[i];
      else
        axis_t[i] = HUGE_VALF;
    }
    int quit = 0;
    for (int plane = 0; plane < max_samples; plane++) {
      float t = quiet_nan(), t2 = quiet_nan();
      int32_t blk_id = 0;
      while (!quit) {
        float tnow;
        if (axis_t[0] <= axis_t[1] && axis_t[0] <= axis_t[2])
          ORV_STEP(0);
        else if (axis_t[1] <= axis_t[2])
          ORV_STEP(1);
        else
          ORV_STEP(2);
        if (quit) break;
        if (axis_int[0] < 0 || axis_int[0] >= dims[0] || axis_int[1] < 0 || axis_int[1] >= dims[1] ||
            axis_int[2] < 0 || axis_int[2] >= dims[2])
          continue; /* still outside the grid */
        blk_id = in_voxel[(int64_t)axis_int[0] * strides[0] + (int64_t)axis_int[1] * strides[1] +
                          (int64_t)axis_int[2] * strides[2]];
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
      out_depth[pix * max_samples + plane] = t;
      out_depth[npix * max_samples + pix * max_samples + plane] = t2;
      out_voxel_id[pix * max_samples + plane] = blk_id;
    }
  }
}
