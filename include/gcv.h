/*
 * gcv.h -- C ABI of the MI355X-native point-generation / visibility path (libgcv_hip.so).
 *
 * the step immediately before the rasterizer in GaussianCity's
 * inference and dataset generation -- BEV maps -> extruded points -> voxel volume -> per-pixel
 * first-hit point id (scripts/dataset_generator.py:1251-1461).  Reference interfaces replaced:
 *
 *   gcv_extrude_count / gcv_extrude_emit   footprint_extruder.get_points_from_projection
 *                                          (extensions/footprint_extruder/footprint_extruder.cpp:143-213;
 *                                           a CPU loop upstream, "the end-to-end bottleneck", README.md:101)
 *   gcv_points_to_volume                   voxlib.points_to_volume
 *                                          (extensions/voxlib/points_to_volume.cu:21-81, bindings.cpp:36)
 *   gcv_maps_to_volume                     voxlib.maps_to_volume
 *                                          (extensions/voxlib/maps_to_volume.cu:21-142, bindings.cpp:38; no in-tree caller)
 *   gcv_ray_voxel_intersection             voxlib.ray_voxel_intersection_perspective
 *                                          (extensions/voxlib/ray_voxel_intersection.cu:54-332, bindings.cpp:33)
 *   gcv_build_occupancy                    (none upstream: 1 bit per 16x16x16 macro cell; lets the traversal
 *                                           jump across empty macro cells, reproducing the reference walk
 *                                           exactly -- results unchanged)
 *
 * Plain C: device pointers + sizes + a HIP stream, no torch types.  Every function returns 0 on
 * success or a negative gcv_status; gcv_last_error() gives the message (per host thread).
 * All pointers are DEVICE pointers unless the name ends in _host.
 */
#ifndef GCV_H
#define GCV_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCV_ABI_VERSION 4

enum gcv_status {
  GCV_OK = 0,
  GCV_ERR_INVALID_ARGUMENT = -1,
  GCV_ERR_HIP = -2,
  GCV_ERR_UNKNOWN_CLASS = -3, /* a pixel's semantic id has no positive scale (upstream never terminates) */
  GCV_ERR_BUFFER_TOO_SMALL = -4
};

int gcv_abi_version(void);
const char* gcv_last_error(void);

/* segInsMap of footprint_extruder.cpp:90-100,201 (scripts/dataset_generator.py:984-1004) */
typedef struct gcv_seg_ins {
  int16_t bldg_ins_min_id, car_ins_min_id, car_semantic_id, bldg_facade_semantic_id, roof_ins_offset;
} gcv_seg_ins;

/* ---- K15: footprint extruder --------------------------------------------------------------
 * Maps are row-major [height][width]: seg/td/bu int16, pts uint8 (numpy bool).
 * scale_of_semantic: int16[32768], scale_of_semantic[s] = scales[classes[s]] (<= 0: unknown).
 * scratch: gcv_extrude_scratch_bytes(height, width) bytes, carried from _count to _emit.
 * _count enqueues the counting pass, waits for it and stores the number of points (upstream's
 * points.size()) in *n_points_host; _emit writes them in upstream's order (row-major pixels,
 * z ascending) as [n][5] int16 = (x, y, z, scale, instanceID) -- the NPY_UINT16 rows of
 * getNumpyArrayFromVector (:66-84) bit for bit. */
size_t gcv_extrude_scratch_bytes(int32_t height, int32_t width);
int gcv_extrude_count(int32_t include_bottom_points, const int16_t* scale_of_semantic, const gcv_seg_ins* seg_ins_host,
                      int32_t height, int32_t width, const int16_t* seg_map, const int16_t* td_hf,
                      const int16_t* bu_hf, const uint8_t* pts_map, void* scratch, size_t scratch_bytes,
                      int64_t* n_points_host, void* hip_stream);
int gcv_extrude_emit(int32_t include_bottom_points, const int16_t* scale_of_semantic, const gcv_seg_ins* seg_ins_host,
                     int32_t height, int32_t width, const int16_t* seg_map, const int16_t* td_hf,
                     const int16_t* bu_hf, const uint8_t* pts_map, const void* scratch, size_t scratch_bytes,
                     int16_t* points_out, int64_t n_points, void* hip_stream);

/* ---- K13: BEV maps -> instance volume (voxlib.maps_to_volume, extensions/voxlib/maps_to_volume.cu:21-142) ----
 * inst_map/td_hf/bu_hf int16 [height][width], pts_map uint8, scales int8 [n_scales] indexed by semantic class
 * (instance < 10 ? instance : 2, as upstream :16-19,44); volume int16 [height][width][depth] (upstream: depth 504)
 * is zeroed and receives the instance id at every z of a border column (roof = instance + 1).  z outside
 * [0, depth) is skipped (upstream writes out of bounds).  scratch8: 8 device bytes.  Waits for completion. */
int gcv_maps_to_volume(const int16_t* inst_map, const int16_t* td_hf, const int16_t* bu_hf, const uint8_t* pts_map,
                       const int8_t* scales, int32_t n_scales, int32_t height, int32_t width, int32_t depth,
                       int16_t* volume, void* scratch8, void* hip_stream);

/* ---- K14: points -> dense volume ------------------------------------------------------------
 * points [n][3] int16 (x, y, z), pt_ids [n] int32, scales [n][3] int16; volume int32 [h][w][d]
 * (d fastest) is zeroed here (upstream torch::zeros) and every point writes its id into the cube
 * [x,x+sx) x [y,y+sy) x [z,z+sz) clipped to the volume.  Where cubes overlap the HIGHEST id wins
 * (upstream's plain stores race; this is the outcome of a sequential loop in point order).
 * occupancy (nullable): gcv_occupancy_bytes(h, w, d) bytes, filled as by gcv_build_occupancy. */
size_t gcv_occupancy_bytes(int32_t h, int32_t w, int32_t d);
int gcv_points_to_volume(int64_t n_points, const int16_t* points, const int32_t* pt_ids, const int16_t* scales,
                         int32_t h, int32_t w, int32_t d, int32_t* volume, uint32_t* occupancy, void* hip_stream);
int gcv_build_occupancy(const int32_t* volume, int32_t h, int32_t w, int32_t d, uint32_t* occupancy, void* hip_stream);

/* Fused form of scripts/dataset_generator.py:1366-1388 (_get_volume) for rows as the extruder writes them:
 * gcv_points_bounds: per-axis min / max of columns 0..2 of int16 rows with `row_stride` elements (3 or 5);
 *   waits for the result (upstream: six .item() calls).  scratch: gcv_bounds_scratch_bytes() device bytes.
 * gcv_rows_to_volume: rows [n][5] = (x, y, z, scale, instance); voxel id = row index + 1, position =
 *   (x, y, z) - offset in int16 arithmetic, cube of `scale` voxels per side -- what _get_volume + points_to_volume
 *   produce for scales = get_point_scales(rows[:, 3]) (utils/helpers.py:197-222, no special classes).
 *   volume_is_zero != 0: the caller guarantees the h*w*d ints are already 0 (a resident workspace restored by
 *   gcv_rows_erase_volume), so the clear -- the longest stage of a frame -- is skipped.
 * gcv_rows_erase_volume: writes 0 over exactly the cubes gcv_rows_to_volume(rows, offset, ...) wrote. */
size_t gcv_bounds_scratch_bytes(void);
int gcv_points_bounds(int64_t n_points, const int16_t* rows, int32_t row_stride, void* scratch, int32_t min_host[3],
                      int32_t max_host[3], void* hip_stream);
int gcv_rows_to_volume(int64_t n_points, const int16_t* rows, const int32_t offset[3], int32_t h, int32_t w, int32_t d,
                       int32_t* volume, uint32_t* occupancy, int32_t volume_is_zero, void* hip_stream);
int gcv_rows_erase_volume(int64_t n_points, const int16_t* rows, const int32_t offset[3], int32_t h, int32_t w,
                          int32_t d, int32_t* volume, void* hip_stream);

/* ---- K12: perspective ray / voxel traversal -----------------------------------------------------
 * volume int32 [dims0][dims1][dims2] with element strides (any layout torch can hand over);
 * cam_ori/cam_dir/cam_up are HOST float[3] (upstream copies them to the CPU, :256-266);
 * cam_c = (c_row, c_col), img_dims = (rows, cols).  Outputs as upstream (:36-43):
 *   out_voxel_id int32 [rows][cols][max_samples]      0 = no hit
 *   out_depth    float [2][rows][cols][max_samples]   entry t and exit t2; quiet NaN 0x7fc00000 = no hit
 *   out_raydirs  float [rows][cols][3]
 * occupancy (nullable) must describe `volume` with contiguous [h][w][d] strides; it removes steps and
 * memory reads, never changes an output. */
int gcv_ray_voxel_intersection(const int32_t* volume, const int32_t dims[3], const int64_t strides[3],
                               const uint32_t* occupancy, const float cam_ori_host[3], const float cam_dir_host[3],
                               const float cam_up_host[3], float cam_f, const float cam_c[2],
                               const int32_t img_dims[2], int32_t max_samples, int32_t* out_voxel_id,
                               float* out_depth, float* out_raydirs, void* hip_stream);

/* avg device ms per stage since the last call (option "timing" of gcv_set_option):
 * 0 extrude_count, 1 extrude_emit, 2 volume_clear, 3 volume_scatter, 4 occupancy, 5 traversal */
int gcv_set_option(const char* name, int value);
int gcv_get_stage_ms(float* out, int n);

#ifdef __cplusplus
}
#endif
#endif
