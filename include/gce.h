/*
 * gce.h -- C ABI of the MI355X-native multi-resolution hash-grid encoder (libgce_hip.so).
 *
 * SURVEY.md section 8 row f3.  Replaces the reference's native module `grid_encoder_ext`
 * (extensions/grid_encoder/bindings.cpp:19-40):
 *
 *   gce_forward    grid_encode_forward   (grid_encoder_ext.cu:459-494 -> kernel_grid, :97-256)
 *   gce_backward   grid_encode_backward  (grid_encoder_ext.cu:496-539 -> kernel_grid_backward :258-337,
 *                                         kernel_input_backward :339-366)
 *
 * Same argument meaning and order as upstream's functions, with raw DEVICE pointers for the tensors
 * and a HIP stream appended:
 *   inputs      float [B][D]   in [0,1]; a point with any coordinate outside encodes to 0 and has no gradient
 *   embeddings  float [offsets[L]][C]
 *   offsets     int32 [L+1]    first row of every level
 *   outputs     float [L][B][C]            (the Python side permutes to [B][L*C], __init__.py:75)
 *   dy_dx       float [B][L][D][C]         only touched when calc_grad_inputs
 *   grad        float [L][B][C];  grad_embeddings as embeddings (ACCUMULATED into: zero it first, as
 *               upstream's torch.zeros_like); grad_inputs float [B][D]
 *   B points, D in 2..5 input dims, C in {1,2,4,8} channels per level, L <= 32 levels,
 *   S = log2(per_level_scale), H = base resolution, gridtype 0 = hash / 1 = tiled.
 * gce_forward / gce_backward are the float32 entry points (GaussianCity's embeddings are float32).  gce_forward_t /
 * gce_backward_t (ABI v2) take the dtype upstream dispatches over (AT_DISPATCH_FLOATING_TYPES_AND_HALF, grid_encoder_ext.cu:
 * 555,597): `inputs` stay float; embeddings, outputs, dy_dx, grad, grad_embeddings and grad_inputs are GCE_F32 float,
 * GCE_F16 IEEE binary16 or GCE_F64 double, every accumulator in that type with upstream's roundings (c10::Half rounds
 * after every product and every sum).
 * The per-level scale exp2f(level*S)*H - 1 is evaluated on the host (gce_level_scales) -- see
 * oracle/gce_oracle.c "gce-fp32-v1".  Returns 0 or a negative gce_status; gce_last_error() has the text.
 */
#ifndef GCE_H
#define GCE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCE_ABI_VERSION 2
#define GCE_MAX_LEVELS 32

enum gce_status { GCE_OK = 0, GCE_ERR_INVALID_ARGUMENT = -1, GCE_ERR_HIP = -2, GCE_ERR_UNSUPPORTED = -3 };

int gce_abi_version(void);
const char* gce_last_error(void);

/* scales_host[l] = exp2f(l * S) * H - 1.0f  (grid_encoder_ext.cu:132,277), l < L <= GCE_MAX_LEVELS */
int gce_level_scales(uint32_t L, float S, uint32_t H, float* scales_host);

int gce_forward(const float* inputs, const float* embeddings, const int32_t* offsets, float* outputs, uint32_t B,
                uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, float* dy_dx,
                uint32_t gridtype, int align_corners, void* hip_stream);

int gce_backward(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets,
                 float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                 int calc_grad_inputs, const float* dy_dx, float* grad_inputs, uint32_t gridtype, int align_corners,
                 void* hip_stream);

enum gce_dtype { GCE_F32 = 0, GCE_F16 = 1, GCE_F64 = 2 };
int gce_forward_t(int dtype, const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs, uint32_t B,
                  uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, void* dy_dx,
                  uint32_t gridtype, int align_corners, void* hip_stream);
int gce_backward_t(int dtype, const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                   void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                   int calc_grad_inputs, const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners,
                   void* hip_stream);

/* avg device ms per stage since the last call (option "timing"): 0 forward, 1 backward_embeddings, 2 backward_inputs */
int gce_set_option(const char* name, int value);
int gce_get_stage_ms(float* out, int n);

#ifdef __cplusplus
}
#endif
#endif
