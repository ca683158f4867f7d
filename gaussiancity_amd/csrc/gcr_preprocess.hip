// gcr_preprocess.hip -- per-Gaussian streaming kernels for gfx950 (HBM-bound):
//   K0 mark_visible, K1 forward preprocess (+ per-block tile counts), K2 scan of block
//   counts, K8 fused backward preprocess (reference K8a computeCov2DCUDA + K8b preprocessCUDA).
// One thread per Gaussian, 256-thread blocks (4 wave64).  Camera matrices are wave-uniform
// and are fetched through the scalar cache (s_load), per-Gaussian attributes through
// per-lane vector loads.  Arithmetic follows gcr-fp32-v1 (gcr_device.h) in the operation
// order of cr/forward.cu / cr/backward.cu so that results are bit-identical to the oracle.
#include <cstdlib>

#include "gcr_device.h"
#include "gcr_internal.h"

namespace {

// cr/auxiliary.h:22-30
__device__ const float SH_C0 = 0.28209479177387814f;
__device__ const float SH_C1 = 0.4886025119029199f;
__device__ const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                   -1.0925484305920792f, 0.5462742152960396f};
__device__ const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                   0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                   -0.5900435899266435f};

struct V3 {
  float x, y, z;
};

// cr/auxiliary.h:48-56
GCR_DEV V3 transform_point_4x3(const V3 p, const float* __restrict__ m) {
  V3 o;
  o.x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
  o.y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
  o.z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
  return o;
}

// Non-zero part of T = W*J (glm T[c][r], c in {0,1}) plus the clamped camera-space mean;
// shared by the forward (cr/forward.cu:69-105) and backward (cr/backward.cu:160-187) paths.
struct Cov2DCtx {
  V3 t;
  float txtz, tytz, limx, limy;
  float T[2][3];
};

GCR_DEV void cov2d_setup(const V3 mean, float fx, float fy, float tan_fovx, float tan_fovy,
                         const float* __restrict__ vm, Cov2DCtx& c) {
  c.t = transform_point_4x3(mean, vm);
  c.limx = 1.3f * tan_fovx;
  c.limy = 1.3f * tan_fovy;
  c.txtz = c.t.x / c.t.z;
  c.tytz = c.t.y / c.t.z;
  c.t.x = gcr_min(c.limx, gcr_max(-c.limx, c.txtz)) * c.t.z;
  c.t.y = gcr_min(c.limy, gcr_max(-c.limy, c.tytz)) * c.t.z;
  const float tz = c.t.z;
  const float J00 = fx / tz, J02 = -(fx * c.t.x) / (tz * tz);
  const float J11 = fy / tz, J12 = -(fy * c.t.y) / (tz * tz);
  c.T[0][0] = vm[0] * J00 + vm[2] * J02;
  c.T[0][1] = vm[4] * J00 + vm[6] * J02;
  c.T[0][2] = vm[8] * J00 + vm[10] * J02;
  c.T[1][0] = vm[1] * J11 + vm[2] * J12;
  c.T[1][1] = vm[5] * J11 + vm[6] * J12;
  c.T[1][2] = vm[9] * J11 + vm[10] * J12;
}

// cov = transpose(T) * transpose(Vrk) * T ; returns (cov[0][0]+0.3, cov[0][1], cov[1][1]+0.3)
GCR_DEV void cov2d_eval(const Cov2DCtx& c, const float (&cv)[6], float (&out)[3]) {
  const float V[3][3] = {{cv[0], cv[1], cv[2]}, {cv[1], cv[3], cv[4]}, {cv[2], cv[4], cv[5]}};
  float A[3][2];
#pragma unroll
  for (int cc = 0; cc < 3; cc++)
#pragma unroll
    for (int r = 0; r < 2; r++)
      A[cc][r] = c.T[r][0] * V[0][cc] + c.T[r][1] * V[1][cc] + c.T[r][2] * V[2][cc];
  const float c00 = A[0][0] * c.T[0][0] + A[1][0] * c.T[0][1] + A[2][0] * c.T[0][2];
  const float c01 = A[0][1] * c.T[0][0] + A[1][1] * c.T[0][1] + A[2][1] * c.T[0][2];
  const float c11 = A[0][1] * c.T[1][0] + A[1][1] * c.T[1][1] + A[2][1] * c.T[1][2];
  out[0] = c00 + 0.3f;
  out[1] = c01;
  out[2] = c11 + 0.3f;
}

// cr/forward.cu:110-144 (quaternion (r,x,y,z) deliberately not normalised)
GCR_DEV void compute_cov3d(const V3 scale, float mod, const float4 rot, float (&cov3D)[6]) {
  const float s[3] = {mod * scale.x, mod * scale.y, mod * scale.z};
  const float r = rot.x, x = rot.y, y = rot.z, z = rot.w;
  const float R[3][3] = {
      {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
      {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
      {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
  float M[3][3];
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int k = 0; k < 3; k++) M[c][k] = s[k] * R[c][k];
#define GCR_SIG(c, r) (M[r][0] * M[c][0] + M[r][1] * M[c][1] + M[r][2] * M[c][2])
  cov3D[0] = GCR_SIG(0, 0);
  cov3D[1] = GCR_SIG(0, 1);
  cov3D[2] = GCR_SIG(0, 2);
  cov3D[3] = GCR_SIG(1, 1);
  cov3D[4] = GCR_SIG(1, 2);
  cov3D[5] = GCR_SIG(2, 2);
#undef GCR_SIG
}

// ------------------------------------------------------------------------------------- K0
// cr/rasterizer_impl.cu:52-62 with in_frustum of cr/auxiliary.h:135-156 (near plane only).
__global__ __launch_bounds__(256) void k_mark_visible(int P, const float* __restrict__ means3D,
                                                      const float* __restrict__ view,
                                                      uint8_t* __restrict__ present) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  const V3 p = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
  const V3 pv = transform_point_4x3(p, view);
  present[idx] = !(pv.z <= 0.2f);
}

// ------------------------------------------------------------------------------------- K1
// cr/forward.cu:147-233, split in two for wave64 lane efficiency.  In city-scale frames only a few
// percent of the Gaussians survive culling, and the exact per-Gaussian math (~460 VALU ops: IEEE
// divisions, sqrt, double-precision ndc2Pix) plus the SH gather are latency-heavy.  Measured on 5M
// Gaussians: inputs alone stream in 35 us (6.2 TB/s); doing the exact math in the streaming loop
// cost 92 us; queueing candidates in LDS and draining them inside the same kernel still cost
// 101 us because every wave fills its queue only at the end of its chunk, so all the gathers pile up
// where nothing is left to overlap them.  Hence:
//   K1a k_preprocess_cull (persistent grid, prefetch one iteration ahead): phase A0 -- the exact
//       near-plane test plus a CONSERVATIVE fast-math screen test -- writes radii = 0 for certain
//       culls and appends everything else to the block's candidate list.  39 us.
//   K1b k_preprocess_project (one candidate per thread, dense): phase A1 -- the reference's exact
//       arithmetic (gcr-fp32-v1) takes EVERY decision (det == 0, radius, tile rect, area == 0) --
//       then phase B: SH -> RGB, the 64-byte record (clamp mask in its fourth quad), and the compacted visible list
//       that the binning kernels and the backward preprocess iterate.
// A0 only ever skips Gaussians whose exact result is radius 0 / no tiles, so outputs are unchanged.
struct PhaseAIn {
  V3 p;
  float c0, c1, c2, c3, c4, c5, c6;  // scale.xyz + rot.rxyz, or cov3D[0..5] (scalars: stay in VGPRs)
};

// NT (round 6): the stream's loads carry the non-temporal policy -- the instruction stream is otherwise the same.
// A streaming cull that has the memory system to itself is latency-bound (C3: 330 MB in 75 us = 4.4 TB/s) and its
// lines are read once: with `nt` they land 9 % sooner (75 -> 68.6 us).  Three culls of three frames in flight read the
// SAME arrays a few microseconds apart and live on each other's lines in L2 / Infinity Cache: there `nt` costs 8 %
// (149 -> 161 us each, 5 740 -> 5 550 frames/s).  So the host picks: GcrPreprocessArgs.nt_stream, set by the entry
// points whose caller waits for num_rendered in every frame -- no second cull of that caller can be running
// (profiles/r06_cache_policy_ab.jsonl).
template <bool NT, typename T>
GCR_DEV T k1_stream_load(const T* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}
// The load is unconditional (callers clamp idx into range): a predicated prefetch makes the
// compiler's s_waitcnt placement lose count across the loop back-edge and wait for everything.
template <bool PRECOMP_COV, bool NT = false>
GCR_DEV void phase_a_load(const GcrPreprocessArgs& a, long long idx, PhaseAIn& in) {
  const float* __restrict__ mp = a.means3D + (size_t)idx * a.s_mean;  // row strides: 3 / 3 / 4 floats when dense,
  in.p = {k1_stream_load<NT>(mp), k1_stream_load<NT>(mp + 1), k1_stream_load<NT>(mp + 2)};  // 14 for column slices of a [N,14] tensor
  if (PRECOMP_COV) {
    const float* __restrict__ c = a.cov3D_precomp + 6 * (size_t)idx;
    in.c0 = k1_stream_load<NT>(c); in.c1 = k1_stream_load<NT>(c + 1); in.c2 = k1_stream_load<NT>(c + 2);
    in.c3 = k1_stream_load<NT>(c + 3); in.c4 = k1_stream_load<NT>(c + 4); in.c5 = k1_stream_load<NT>(c + 5);
    in.c6 = 0.0f;
  } else {
    const float* __restrict__ sp = a.scales + (size_t)idx * a.s_scale;
    in.c0 = k1_stream_load<NT>(sp);
    in.c1 = k1_stream_load<NT>(sp + 1);
    in.c2 = k1_stream_load<NT>(sp + 2);
    const float* __restrict__ rp = a.rotations + (size_t)idx * a.s_rot;
    if (a.s_rot == 4) {  // dense: one 16-byte load (the array is 16-byte aligned: torch allocations are)
      typedef float gcr_f4 __attribute__((ext_vector_type(4)));
      const gcr_f4 rot = k1_stream_load<NT>(reinterpret_cast<const gcr_f4*>(rp));
      in.c3 = rot.x; in.c4 = rot.y; in.c5 = rot.z; in.c6 = rot.w;
    } else {
      in.c3 = k1_stream_load<NT>(rp); in.c4 = k1_stream_load<NT>(rp + 1); in.c5 = k1_stream_load<NT>(rp + 2); in.c6 = k1_stream_load<NT>(rp + 3);
    }
  }
}

// Wave-uniform float forced into an SGPR.  The camera matrices are read once per kernel this way:
// left to itself the compiler re-fetched them with VECTOR loads in every loop iteration (it
// cannot prove them invariant next to the kernel's stores, so no s_load).
GCR_DEV float gcr_uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// Phase A0.  True = the exact path is CERTAIN to give radius 0 for this Gaussian.
//  - near plane: the reference's own test on the exactly computed view depth;
//  - screen: pixel centre known to ~1e-6 relative; radius_ref = ceil(3*sqrt(lambda_1)) with
//    |cov2D entries| <= E = |T|_F^2 * rho(Sigma),  |T|_F <= |W|_2 * |J|_F  (wf2 >= |W|_2^2 by
//    Gershgorin on W^T W, once per kernel; 1 for a rigid view matrix),
//    rho(Sigma) <= (mod*s_max)^2 * |R(q)|_F^2 <= 9 (mod*s_max)^2 rmax^2  with every entry of R(q)
//    as written at cr/forward.cu:126-130 bounded by rmax = max(1, 2|q|^2 - 1); for a
//    caller-supplied covariance (need not be PSD) rho <= Frobenius norm.  Then
//      Sigma PSD:      lambda_1 <= (a + c) + 0.317 <= E + 0.917
//      any symmetric:  mid <= E + 0.3, mid^2 - det <= 2 E^2  =>  lambda_1 <= 2.42 E + 0.62.
//    All terms are sums of squares (no cancellation), so 1-ulp rcp/sqrt and FMAs are covered many
//    times over by the 2 % + 2 px inflation.  NaN anywhere makes the comparisons false -> candidate.
//    The radius bound is about 2x the true radius for unit quaternions and a rigid camera, so the
//    extra candidates are the Gaussians within a couple of radii of the screen border.
// rho >= spectral radius of the Gaussian's world-space covariance, scale_modifier included: the one per-Gaussian
// quantity of the bound above that does not depend on the camera (what gcr_build_cull_cache stores beside the mean).
template <bool PRECOMP_COV>
GCR_DEV float cull_rho(float scale_modifier, const PhaseAIn& in) {
  if (PRECOMP_COV) {
    return __builtin_amdgcn_sqrtf(in.c0 * in.c0 + in.c3 * in.c3 + in.c5 * in.c5 +
                                  2.0f * (in.c1 * in.c1 + in.c2 * in.c2 + in.c4 * in.c4)) * 1.001f;
  } else {
    const float q2 = __builtin_fmaf(in.c3, in.c3, __builtin_fmaf(in.c4, in.c4, __builtin_fmaf(in.c5, in.c5, in.c6 * in.c6)));
    // diagonal 1 - 2(u^2+v^2) lies in [1 - 2|q|^2, 1]; off-diagonal 2(uv +- rw) <= |q|^2 (AM-GM)
    const float rmax = __builtin_fmaxf(1.0f, __builtin_fmaf(2.0f, q2, -1.0f)) * 1.0001f;  // >= every |R_ij|
    const float smax = scale_modifier * __builtin_fmaxf(__builtin_fabsf(in.c0),
                                                        __builtin_fmaxf(__builtin_fabsf(in.c1), __builtin_fabsf(in.c2)));
    return 9.0f * (smax * smax) * (rmax * rmax);
  }
}

template <bool PRECOMP_COV>
GCR_DEV bool phase_a0_certainly_culled(const GcrPreprocessArgs& a, const float (&vm)[16], const float (&pm)[16],
                                       float wf2, const PhaseAIn& in, float rho) {
  const float tz = vm[2] * in.p.x + vm[6] * in.p.y + vm[10] * in.p.z + vm[14];  // exact, as transformPoint4x3
  if (tz <= 0.2f) return true;  // in_frustum, cr/auxiliary.h:145
  const float tx = __builtin_fmaf(vm[0], in.p.x, __builtin_fmaf(vm[4], in.p.y, __builtin_fmaf(vm[8], in.p.z, vm[12])));
  const float ty = __builtin_fmaf(vm[1], in.p.x, __builtin_fmaf(vm[5], in.p.y, __builtin_fmaf(vm[9], in.p.z, vm[13])));
  const float hx = __builtin_fmaf(pm[0], in.p.x, __builtin_fmaf(pm[4], in.p.y, __builtin_fmaf(pm[8], in.p.z, pm[12])));
  const float hy = __builtin_fmaf(pm[1], in.p.x, __builtin_fmaf(pm[5], in.p.y, __builtin_fmaf(pm[9], in.p.z, pm[13])));
  const float hw = __builtin_fmaf(pm[3], in.p.x, __builtin_fmaf(pm[7], in.p.y, __builtin_fmaf(pm[11], in.p.z, pm[15])));
  const float pw = __builtin_amdgcn_rcpf(hw + 0.0000001f);
  const float half_w = 0.5f * (float)a.W, half_h = 0.5f * (float)a.H;
  const float px = __builtin_fmaf(hx * pw, half_w, half_w - 0.5f);  // ((ndc+1)*W-1)/2
  const float py = __builtin_fmaf(hy * pw, half_h, half_h - 0.5f);
  const float rtz = __builtin_amdgcn_rcpf(tz);
  const float limx = 1.3f * a.tanfovx, limy = 1.3f * a.tanfovy;
  const float cx = __builtin_fminf(limx, __builtin_fmaxf(-limx, tx * rtz));
  const float cy = __builtin_fminf(limy, __builtin_fmaxf(-limy, ty * rtz));
  const float j0 = a.focal_x * rtz, j1 = a.focal_y * rtz;  // J00, J11; J02 = -j0*cx, J12 = -j1*cy
  const float jf2 = __builtin_fmaf(j0 * j0, __builtin_fmaf(cx, cx, 1.0f), (j1 * j1) * __builtin_fmaf(cy, cy, 1.0f));
  const float E = wf2 * jf2 * rho;
  const float lam = __builtin_fmaf(PRECOMP_COV ? 2.42f : 1.0f, E, 1.0f);
  const float rb = __builtin_fmaf(3.06f, __builtin_amdgcn_sqrtf(lam), 2.0f);  // 3 * 1.02 * sqrt + 2 >= radius_ref
  const float slack = 2.0f;
  // area == 0 upstream  <=>  px + r < 1  or  px - r >= 16*grid_x  (same in y), cr/auxiliary.h:36-46
  return px + rb < -slack || px - rb > 16.0f * (float)a.gx + slack || py + rb < -slack ||
         py - rb > 16.0f * (float)a.gy + slack;
}

// gcr_camera.prefiltered: does this Gaussian fail the near-plane test (the only test of in_frustum, cr/auxiliary.h:145)?
GCR_DEV bool near_plane_violation(const float (&vm)[16], const PhaseAIn& in) {
  const float tz = vm[2] * in.p.x + vm[6] * in.p.y + vm[10] * in.p.z + vm[14];  // exact, as transformPoint4x3
  return tz <= 0.2f;
}

// Phase A1: the reference's exact per-Gaussian arithmetic.  Returns whether the Gaussian is
// rendered; fills the projected state and the integer radius (0 if not rendered).
struct Projected {
  float px, py, conx, cony, conz, depth;
  uint32_t rect_x, rect_y;
};

template <bool PRECOMP_COV>
GCR_DEV bool phase_a1_exact(const GcrPreprocessArgs& a, const float (&vm)[16], const float (&pm)[16], int idx,
                            const PhaseAIn& in, Projected& out, int& radius_out) {
  radius_out = 0;
  const V3 p_orig = in.p;
  const V3 p_view = transform_point_4x3(p_orig, vm);
  if (p_view.z <= 0.2f) return false;  // in_frustum
  const float hx = pm[0] * p_orig.x + pm[4] * p_orig.y + pm[8] * p_orig.z + pm[12];
  const float hy = pm[1] * p_orig.x + pm[5] * p_orig.y + pm[9] * p_orig.z + pm[13];
  const float hw = pm[3] * p_orig.x + pm[7] * p_orig.y + pm[11] * p_orig.z + pm[15];
  const float p_w = 1.0f / (hw + 0.0000001f);
  const float projx = hx * p_w, projy = hy * p_w;
  float cov3D[6];
  if (PRECOMP_COV) {
    cov3D[0] = in.c0; cov3D[1] = in.c1; cov3D[2] = in.c2;
    cov3D[3] = in.c3; cov3D[4] = in.c4; cov3D[5] = in.c5;
  } else {
    const V3 sc = {in.c0, in.c1, in.c2};
    compute_cov3d(sc, a.scale_modifier, make_float4(in.c3, in.c4, in.c5, in.c6), cov3D);
  }
  Cov2DCtx cc;
  float cov[3];
  cov2d_setup(p_orig, a.focal_x, a.focal_y, a.tanfovx, a.tanfovy, vm, cc);
  cov2d_eval(cc, cov3D, cov);
  const float det = (cov[0] * cov[2] - cov[1] * cov[1]);
  if (det == 0.0f) return false;
  const float det_inv = 1.f / det;
  const float conx = cov[2] * det_inv, cony = -cov[1] * det_inv, conz = cov[0] * det_inv;
  const float mid = 0.5f * (cov[0] + cov[2]);
  const float lambda1 = mid + __builtin_sqrtf(gcr_max(0.1f, mid * mid - det));
  const float lambda2 = mid - __builtin_sqrtf(gcr_max(0.1f, mid * mid - det));
  const float my_radius = __builtin_ceilf(3.f * __builtin_sqrtf(gcr_max(lambda1, lambda2)));
  const float px = gcr_ndc2pix(projx, a.W), py = gcr_ndc2pix(projy, a.H);
  // getRect, cr/auxiliary.h:36-46.  min(grid, max(0, (int)x)) == (int)clamp(x, 0, grid) for every x
  // incl. NaN/inf (the float clamp returns the non-NaN operand); 3 VALU ops instead of ~9 per bound.
  const int ri = gcr_f2i_sat(my_radius);
  const float rf = (float)ri;
  const float gxf = (float)a.gx, gyf = (float)a.gy;
  const int minx = (int)__builtin_fminf(__builtin_fmaxf((px - rf) / 16.0f, 0.0f), gxf);
  const int miny = (int)__builtin_fminf(__builtin_fmaxf((py - rf) / 16.0f, 0.0f), gyf);
  const int maxx = (int)__builtin_fminf(__builtin_fmaxf((px + rf + 16.0f - 1.0f) / 16.0f, 0.0f), gxf);
  const int maxy = (int)__builtin_fminf(__builtin_fmaxf((py + rf + 16.0f - 1.0f) / 16.0f, 0.0f), gyf);
  if ((uint32_t)(maxx - minx) * (uint32_t)(maxy - miny) == 0) return false;
  // Deliberate deviation shared with the oracle: a radius that converts to an int <= 0 (NaN scale /
  // covariance, or lambda <= 0 for an indefinite supplied covariance) is not rendered.  Upstream
  // counts such a Gaussian in num_rendered but never emits its keys (radii == 0), so its sort reads
  // uninitialised slots (cr/forward.cu:228-232 vs cr/rasterizer_impl.cu:78).
  if (ri <= 0) return false;
  radius_out = ri;
  out.px = px; out.py = py; out.conx = conx; out.cony = cony; out.conz = conz; out.depth = p_view.z;
  out.rect_x = (uint32_t)minx | ((uint32_t)maxx << 16);
  out.rect_y = (uint32_t)miny | ((uint32_t)maxy << 16);
  // (The covariance is NOT kept in the geometry buffer any more -- round 6, late.  Upstream stores it for its backward
  // (cr/forward.cu:185-188, cr/backward.cu:384-391); here the one kernel that needs it, K8, recomputes it from the same
  // scales and rotation with the same compute_cov3d -- the same bits -- instead of every forward paying a 32-byte store
  // into a fresh sector per survivor: at C3 the cull alone 76.8 -> 70 us, 153 -> 140 us each with three frames in flight,
  // 5 720 -> 5 860 frames/s (profiles/r06_k1_cov3d_store_ab.jsonl).  gcr_layout.geom_cov3D is still carved, and unused.)
  return true;
}

// Phase B: colour (SH evaluation or the caller's colour), record, visible-list entry (and tile counts in the
// global-cursor variant).
//
// computeColorFromSH, cr/forward.cu:20-66.  With M == 16 (the degree-3 layout) a Gaussian's 192 contiguous bytes are
// consumed in FOUR groups of three dwordx4 loads = four coefficients x three channels each, in coefficient order:
// every channel's sum is built in exactly the reference's order (the reference adds the terms of a degree left to
// right, degree after degree), but only 12 coefficient registers are live at a time instead of 48 -- the exact pass
// shares its registers with the streaming loop of the fused kernel, and fewer registers mean more of its workgroups
// fit beside another frame's blend.  The memory clobbers between the groups keep the compiler from hoisting all
// twelve loads to the front again.  (48 scalar loads with a 192-byte lane stride cost ~11 M cache-line requests per
// frame; that is why the loads are 16-byte ones.)
struct ShDir {
  float x, y, z, xx, yy, zz, xy, yz, xz;
};

// term(i) for one channel; the parenthesisation is the reference's left-to-right product
GCR_DEV float sh_term(int i, const ShDir& d, float s) {
  switch (i) {
    case 4: return SH_C2[0] * d.xy * s;
    case 5: return SH_C2[1] * d.yz * s;
    case 6: return SH_C2[2] * (2.0f * d.zz - d.xx - d.yy) * s;
    case 7: return SH_C2[3] * d.xz * s;
    case 8: return SH_C2[4] * (d.xx - d.yy) * s;
    case 9: return SH_C3[0] * d.y * (3.0f * d.xx - d.yy) * s;
    case 10: return SH_C3[1] * d.xy * d.z * s;
    case 11: return SH_C3[2] * d.y * (4.0f * d.zz - d.xx - d.yy) * s;
    case 12: return SH_C3[3] * d.z * (2.0f * d.zz - 3.0f * d.xx - 3.0f * d.yy) * s;
    case 13: return SH_C3[4] * d.x * (4.0f * d.zz - d.xx - d.yy) * s;
    case 14: return SH_C3[5] * d.z * (d.xx - d.yy) * s;
    default: return SH_C3[6] * d.x * (d.xx - 3.0f * d.yy) * s;
  }
}

// Adds coefficient i (value s) to a channel's running sum exactly as the reference's expression does.
GCR_DEV float sh_accumulate(int i, int deg, const ShDir& d, float result, float s) {
  if (i == 0) return SH_C0 * s;
  if (i <= 3) {
    if (deg < 1) return result;
    if (i == 1) return result - SH_C1 * d.y * s;
    if (i == 2) return result + SH_C1 * d.z * s;
    return result - SH_C1 * d.x * s;
  }
  if (i <= 8) return deg > 1 ? result + sh_term(i, d, s) : result;
  return deg > 2 ? result + sh_term(i, d, s) : result;
}

template <bool HAVE_OPACITY = false>
GCR_DEV void preprocess_phase_b(const GcrPreprocessArgs& a, int idx, const V3 mean, const Projected& pr,
                                uint32_t list_pos, uint32_t* __restrict__ vis_list, float cached_opacity = 0.0f) {
  const float opacity = HAVE_OPACITY ? cached_opacity : a.opacities[(size_t)idx * a.s_opac];
  float cr, cg, cb;
  uint32_t clamp_mask = 0u;
  if (a.colors_precomp == nullptr) {
    const float ox = mean.x - GCR_CAM(a, campos, a.campos, 0), oy = mean.y - GCR_CAM(a, campos, a.campos, 1),
                oz = mean.z - GCR_CAM(a, campos, a.campos, 2);
    const float len = __builtin_sqrtf(ox * ox + oy * oy + oz * oz);
    ShDir d;
    d.x = ox / len; d.y = oy / len; d.z = oz / len;
    d.xx = d.x * d.x; d.yy = d.y * d.y; d.zz = d.z * d.z;
    d.xy = d.x * d.y; d.yz = d.y * d.z; d.xz = d.x * d.z;
    const int deg = a.D;
    const float* __restrict__ shp = a.shs + (size_t)idx * a.M * 3;
    float res[3] = {0.0f, 0.0f, 0.0f};
    if (a.M == 16) {
      const float4* __restrict__ sh4 = reinterpret_cast<const float4*>(shp);
      // The 192-byte row lies in two 128-byte lines and the four groups below are four round trips in a row (each
      // group's registers are reused by the next): the row's LAST float is asked for first, so the second line's miss
      // runs beside the first one's instead of starting two round trips later.
      // (round 5; K1 -0.7 us stateless / -2.3 us cached at C3, -4 us cached at C5: profiles/r05_k1_ab_*.jsonl)
      const float sh_last = shp[47];
      asm volatile("" ::: "memory");
#pragma unroll
      for (int grp = 0; grp < 4; grp++) {  // coefficients 4*grp .. 4*grp+3
        const float4 v0 = sh4[3 * grp], v1 = sh4[3 * grp + 1];
        float4 v2 = sh4[3 * grp + 2];
        if (grp == 3) v2.w = sh_last;  // (the same bits)
        const float f[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
          for (int ch = 0; ch < 3; ch++) res[ch] = sh_accumulate(4 * grp + k, deg, d, res[ch], f[3 * k + ch]);
        asm volatile("" ::: "memory");
      }
    } else {
      const int ncoef = (deg + 1) * (deg + 1);
#pragma unroll
      for (int i = 0; i < 16; i++)
        if (i < ncoef)
#pragma unroll
          for (int ch = 0; ch < 3; ch++) res[ch] = sh_accumulate(i, deg, d, res[ch], shp[3 * i + ch]);
    }
#pragma unroll
    for (int ch = 0; ch < 3; ch++) res[ch] = res[ch] + 0.5f;
    clamp_mask = (res[0] < 0 ? 1u : 0u) | (res[1] < 0 ? 2u : 0u) | (res[2] < 0 ? 4u : 0u);
    cr = gcr_max(res[0], 0.0f);
    cg = gcr_max(res[1], 0.0f);
    cb = gcr_max(res[2], 0.0f);
  } else {
    const float* __restrict__ cpp = a.colors_precomp + (size_t)idx * a.s_col;
    cr = cpp[0];
    cg = cpp[1];
    cb = cpp[2];
  }
  float4* __restrict__ rec = a.rec + (size_t)idx * GCR_REC_QUADS;
  rec[0] = make_float4(pr.px, pr.py, pr.conx, pr.cony);
  rec[1] = make_float4(pr.conz, opacity, cr, cg);
  rec[2] = make_float4(cb, pr.depth, __uint_as_float(pr.rect_x), __uint_as_float(pr.rect_y));
  rec[3] = make_float4(__uint_as_float(clamp_mask), 0.0f, 0.0f, 0.0f);
  vis_list[list_pos] = (uint32_t)idx;
  // what the binning kernels need of a survivor, at the same position of the block's list (ABI v9): they read the lists
  // front to back instead of gathering rec[2] of every survivor out of P x 64 bytes
  a.vis_rec[(vis_list - a.vis_list) + list_pos] =
      make_uint4((uint32_t)idx, __float_as_uint(pr.depth), pr.rect_x, pr.rect_y);
  // per-tile instance counts, global-cursor variant only (gcr_binning.hip explains why the
  // default path counts in LDS instead)
  if (a.tile_count != nullptr) {
    const int minx = (int)(pr.rect_x & 0xffffu), maxx = (int)(pr.rect_x >> 16);
    const int miny = (int)(pr.rect_y & 0xffffu), maxy = (int)(pr.rect_y >> 16);
    for (int ty = miny; ty < maxy; ty++)
      for (int tx = minx; tx < maxx; tx++)
        atomicAdd(&a.tile_count[(size_t)(ty * a.gx + tx) * GCR_CURSOR_STRIDE], 1u);
  }
}

// K1a: streaming cull.  Persistent grid, inputs prefetched one iteration ahead; writes radii = 0
// for everything phase A0 rules out and appends the rest to the block's candidate list
// (wave-level ballot compaction, one LDS atomic per wave and iteration).
template <bool PRECOMP_COV>
__global__ __launch_bounds__(256) void k_preprocess_cull(const GcrPreprocessArgs a) {
  __shared__ uint32_t list_tail;  // length of this block's candidate list
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) list_tail = 0;
  __syncthreads();
  float vm[16], pm[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    vm[i] = gcr_uniform(GCR_CAM(a, view, a.view, i));
    pm[i] = gcr_uniform(GCR_CAM(a, proj, a.proj, i));
  }
  // wf2 >= |W|_2^2 for the 3x3 block W the covariance projection uses: Gershgorin row sums of W^T W
  float wf2 = 0.0f;
  {
    const float wc[3][3] = {{vm[0], vm[1], vm[2]}, {vm[4], vm[5], vm[6]}, {vm[8], vm[9], vm[10]}};
#pragma unroll
    for (int i = 0; i < 3; i++) {
      float row = 0.0f;
#pragma unroll
      for (int j = 0; j < 3; j++)
        row += __builtin_fabsf(wc[0][i] * wc[0][j] + wc[1][i] * wc[1][j] + wc[2][i] * wc[2][j]);
      wf2 = __builtin_fmaxf(wf2, row);
    }
    wf2 *= 1.001f;
  }
  const long long chunk_begin = (long long)blockIdx.x * a.chunk;
  const long long chunk_end = chunk_begin + a.chunk < a.P ? chunk_begin + a.chunk : a.P;
  uint32_t* __restrict__ my_cand = a.cand_list + chunk_begin;
  const uint64_t lt_mask = (1ull << lane) - 1ull;

  // inputs are requested two iterations ahead (when the kernel shares the GPU with another frame's blend it gets a
  // fraction of the wave slots, and bytes in flight per wave are what keeps the stream at HBM speed).  What the hardware
  // sees is less: the rotation cur <- nxt <- nx2 at the end of an iteration copies the loads' destination registers, so
  // the loop waits for all but the newest loads -- see k_preprocess_fused_cached for the copy-free form and why this
  // 40-byte stream does not gain from it
  PhaseAIn cur, nxt, nx2;
  bool viol = false;  // gcr_camera.prefiltered and a Gaussian behind the near plane
  long long idx64 = chunk_begin + tid;
  const long long last = chunk_end - 1;  // the prefetch stays inside the block's own chunk (see k_preprocess_fused)
  phase_a_load<PRECOMP_COV>(a, idx64 < last ? idx64 : last, cur);
  phase_a_load<PRECOMP_COV>(a, idx64 + 256 < last ? idx64 + 256 : last, nxt);
  for (long long base = chunk_begin; base < chunk_end; base += 256) {
    idx64 = base + tid;
    phase_a_load<PRECOMP_COV>(a, idx64 + 512 < last ? idx64 + 512 : last, nx2);  // prefetch, two iterations ahead
    bool candidate = false;
    if (idx64 < chunk_end) {
      candidate = !phase_a0_certainly_culled<PRECOMP_COV>(a, vm, pm, wf2, cur, cull_rho<PRECOMP_COV>(a.scale_modifier, cur));
      if (!candidate) a.radii[idx64] = 0;
      if (a.prefiltered) viol |= near_plane_violation(vm, cur);
    }
    const uint64_t m = __ballot(candidate);
    if (m != 0ull) {
      uint32_t wbase = 0;
      if (lane == 0) wbase = atomicAdd(&list_tail, (uint32_t)__popcll(m));
      wbase = __shfl(wbase, 0, 64);
      if (candidate) my_cand[wbase + (uint32_t)__popcll(m & lt_mask)] = (uint32_t)idx64;
    }
    cur = nxt;
    nxt = nx2;
  }
  const int any_viol = __syncthreads_or(viol ? 1 : 0);
  if (tid == 0) a.cand_count[blockIdx.x] = list_tail | (any_viol ? 0x80000000u : 0u);  // (a block holds < 2^31 candidates)
}

// K1b: dense pass over the candidates of K1a block `blockIdx.x`, one candidate per thread: the
// reference's exact arithmetic takes every decision, survivors get colour + record and are
// compacted into the block's visible list.  Gather latency is hidden by occupancy here instead of
// stalling the streaming kernel.  (Tried and measured slower: one-wave workgroups with a global
// atomic per wave, 115 us; prefetching the SH coefficients before the exact math, no gain.)
template <bool PRECOMP_COV>
__global__ __launch_bounds__(256) void k_preprocess_project(const GcrPreprocessArgs a) {
  __shared__ uint32_t list_tail;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) list_tail = 0;
  __syncthreads();
  float vm[16], pm[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    vm[i] = gcr_uniform(GCR_CAM(a, view, a.view, i));
    pm[i] = gcr_uniform(GCR_CAM(a, proj, a.proj, i));
  }
  const size_t chunk_begin = (size_t)blockIdx.x * a.chunk;
  const uint32_t* __restrict__ my_cand = a.cand_list + chunk_begin;
  uint32_t* __restrict__ my_list = a.vis_list + chunk_begin;
  const uint32_t ncand_word = a.cand_count[blockIdx.x];
  const uint32_t ncand = ncand_word & 0x7fffffffu;  // (bit 31: the cull kernel met a prefilter violation)
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  uint32_t my_tiles = 0;  // (Gaussian, tile) instances of this thread's survivors
  for (uint32_t it0 = 0; it0 < ncand; it0 += 256) {  // block-uniform trip count
    const uint32_t it = it0 + tid;
    bool keep = false;
    int idx = 0, radius = 0;
    PhaseAIn in;
    Projected pr;
    if (it < ncand) {
      idx = (int)my_cand[it];
      phase_a_load<PRECOMP_COV>(a, idx, in);
      keep = phase_a1_exact<PRECOMP_COV>(a, vm, pm, idx, in, pr, radius);
      a.radii[idx] = radius;
    }
    const uint64_t m = __ballot(keep);
    if (m != 0ull) {
      uint32_t lbase = 0;
      if (lane == 0) lbase = atomicAdd(&list_tail, (uint32_t)__popcll(m));
      lbase = __shfl(lbase, 0, 64);
      if (keep) {
        my_tiles += ((pr.rect_x >> 16) - (pr.rect_x & 0xffffu)) * ((pr.rect_y >> 16) - (pr.rect_y & 0xffffu));
        preprocess_phase_b(a, idx, in.p, pr, lbase + (uint32_t)__popcll(m & lt_mask), my_list);
      }
    }
  }
  // num_rendered = sum of the survivors' tile counts (what the reference obtains from its inclusive scan,
  // cr/rasterizer_impl.cu:228-238), accumulated per block here so that the host can have R right after K1
  // instead of after the tile-count kernels.
  __shared__ unsigned long long blk_tiles;
  if (tid == 0) blk_tiles = 0ull;
  __syncthreads();
  const uint32_t wsum = gcr_wave_sum_u32(my_tiles);
  if (lane == 0 && wsum) atomicAdd(&blk_tiles, (unsigned long long)wsum);
  __syncthreads();
  if (tid == 0) {
    a.vis_count[blockIdx.x] = list_tail;
    // this block's share of num_rendered; the first workgroup of the kernel that follows sums the blocks' shares
    // and publishes the total to the host (no atomics, nothing to zero beforehand)
    a.block_tiles[blockIdx.x] = blk_tiles | ((ncand_word & 0x80000000u) ? GCR_PREFILTER_FLAG : 0ull);
  }
}

// K1 fused (default): ONE kernel whose workgroups alternate between STREAMING their chunk through the cull
// (K1a's loop) and PROCESSING the candidates they have collected -- exact projection, colour, record (K1b's body).
// The candidates' inputs (index + 10 floats) wait in LDS, so they never go back to HBM and K1b's re-gather of three
// 128-byte lines per candidate disappears; and while one workgroup is in its latency-bound processing pass the
// other workgroups of the CU keep the HBM stream going, so the pass is hidden instead of being a second kernel
// that starts only when the slowest streaming block has finished.  A processing pass runs whenever 256 candidates
// are waiting (all lanes busy) and once more at the end of the chunk.
constexpr int FUSED_CAP = 512;  // <= 255 left over + <= 256 new candidates per iteration

template <bool PRECOMP_COV, bool NT>
__global__ __launch_bounds__(256) void k_preprocess_fused(const GcrPreprocessArgs a) {
  __shared__ uint32_t sIdx[FUSED_CAP];
  __shared__ float sIn[10][FUSED_CAP];
  __shared__ uint32_t cand_tail, vis_tail;
  __shared__ uint32_t wave_new[2][4];  // candidates each wave found in this iteration, double-buffered by parity
  __shared__ unsigned long long blk_tiles;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) {
    cand_tail = 0;
    vis_tail = 0;
    blk_tiles = 0ull;
  }
  __syncthreads();
  float vm[16], pm[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    vm[i] = gcr_uniform(GCR_CAM(a, view, a.view, i));
    pm[i] = gcr_uniform(GCR_CAM(a, proj, a.proj, i));
  }
  float wf2 = 0.0f;
  {
    const float wc[3][3] = {{vm[0], vm[1], vm[2]}, {vm[4], vm[5], vm[6]}, {vm[8], vm[9], vm[10]}};
#pragma unroll
    for (int i = 0; i < 3; i++) {
      float row = 0.0f;
#pragma unroll
      for (int j = 0; j < 3; j++)
        row += __builtin_fabsf(wc[0][i] * wc[0][j] + wc[1][i] * wc[1][j] + wc[2][i] * wc[2][j]);
      wf2 = __builtin_fmaxf(wf2, row);
    }
    wf2 *= 1.001f;
  }
  const long long chunk_begin = (long long)blockIdx.x * a.chunk;
  const long long chunk_end = chunk_begin + a.chunk < a.P ? chunk_begin + a.chunk : a.P;
  uint32_t* __restrict__ my_list = a.vis_list + chunk_begin;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  uint32_t my_tiles = 0;

  // `waiting` = candidates in the LDS queue.  It must be block-uniform (it decides whether the workgroup enters a
  // processing pass with its barriers), so it is NOT read back from cand_tail -- on an iteration without a pass no
  // barrier separates that read from the next iteration's atomicAdd of a faster wave -- but kept in a
  // register: every wave publishes how many candidates it appended (slot parity = iteration parity, so the next
  // iteration's writes cannot overtake a slow reader; the one after that is behind the next barrier) and every
  // thread adds up the four counts after the barrier.
  uint32_t waiting = 0;
  uint32_t parity = 0;

  PhaseAIn cur, nxt, nx2;  // (requested two iterations ahead; what the rotation below leaves of that: see k_preprocess_cull)
  bool viol = false;  // gcr_camera.prefiltered and a Gaussian behind the near plane
  long long idx64 = chunk_begin + tid;
  // The prefetch never leaves the block's own chunk: lanes that would read past it re-read the chunk's last Gaussian (an
  // L2-hot line).  Until round 5 they read on into the next block's chunk -- 20 KB x 2 048 blocks = 42 MB of the 313 MB the
  // kernel fetched at C3.  K1 alone: C3 82.3 -> 83.2 us, C5 355 -> 347 us, C2 18.3 -> 17.1 us
  // (profiles/r05_k1_ab_stateless_clamp_touch.jsonl, one process, alternating).
  const long long last = chunk_end - 1;
  phase_a_load<PRECOMP_COV, NT>(a, idx64 < last ? idx64 : last, cur);
  phase_a_load<PRECOMP_COV, NT>(a, idx64 + 256 < last ? idx64 + 256 : last, nxt);
  for (long long base = chunk_begin; base < chunk_end; base += 256, parity ^= 1u) {
    idx64 = base + tid;
    phase_a_load<PRECOMP_COV, NT>(a, idx64 + 512 < last ? idx64 + 512 : last, nx2);  // prefetch, two iterations ahead
    bool candidate = false;
    if (idx64 < chunk_end) {
      candidate = !phase_a0_certainly_culled<PRECOMP_COV>(a, vm, pm, wf2, cur, cull_rho<PRECOMP_COV>(a.scale_modifier, cur));
      if (!candidate) {  // (nobody reads a culled Gaussian's zero again in this frame)
        if (NT) __builtin_nontemporal_store(0, a.radii + idx64);
        else a.radii[idx64] = 0;
      }
      if (a.prefiltered) viol |= near_plane_violation(vm, cur);
    }
    const uint64_t m = __ballot(candidate);
    if (lane == 0) wave_new[parity][tid >> 6] = (uint32_t)__popcll(m);
    if (m != 0ull) {
      uint32_t wbase = 0;
      if (lane == 0) wbase = atomicAdd(&cand_tail, (uint32_t)__popcll(m));
      wbase = __shfl(wbase, 0, 64);
      if (candidate) {
        const uint32_t slot = wbase + (uint32_t)__popcll(m & lt_mask);
        sIdx[slot] = (uint32_t)idx64;
        sIn[0][slot] = cur.p.x; sIn[1][slot] = cur.p.y; sIn[2][slot] = cur.p.z;
        sIn[3][slot] = cur.c0; sIn[4][slot] = cur.c1; sIn[5][slot] = cur.c2;
        sIn[6][slot] = cur.c3; sIn[7][slot] = cur.c4; sIn[8][slot] = cur.c5;
        sIn[9][slot] = cur.c6;
      }
    }
    cur = nxt;
    nxt = nx2;
    __syncthreads();
    waiting += wave_new[parity][0] + wave_new[parity][1] + wave_new[parity][2] + wave_new[parity][3];
    const bool last_iter = base + 256 >= chunk_end;
    if (waiting >= 256u || (last_iter && waiting > 0u)) {
      // ---- processing passes over the waiting candidates: full 256-lane passes, plus the remainder at the end
      uint32_t done = 0;
      while (done + 256u <= waiting || (last_iter && done < waiting)) {
        const uint32_t it = done + (uint32_t)tid;
        bool keep = false;
        int idx = 0, radius = 0;
        PhaseAIn in;
        Projected pr;
        if (it < waiting) {
          idx = (int)sIdx[it];
          in.p = {sIn[0][it], sIn[1][it], sIn[2][it]};
          in.c0 = sIn[3][it]; in.c1 = sIn[4][it]; in.c2 = sIn[5][it];
          in.c3 = sIn[6][it]; in.c4 = sIn[7][it]; in.c5 = sIn[8][it];
          in.c6 = sIn[9][it];
          keep = phase_a1_exact<PRECOMP_COV>(a, vm, pm, idx, in, pr, radius);
          a.radii[idx] = radius;
        }
        const uint64_t mk = __ballot(keep);
        if (mk != 0ull) {
          uint32_t lbase = 0;
          if (lane == 0) lbase = atomicAdd(&vis_tail, (uint32_t)__popcll(mk));
          lbase = __shfl(lbase, 0, 64);
          if (keep) {
            my_tiles += ((pr.rect_x >> 16) - (pr.rect_x & 0xffffu)) * ((pr.rect_y >> 16) - (pr.rect_y & 0xffffu));
            preprocess_phase_b(a, idx, in.p, pr, lbase + (uint32_t)__popcll(mk & lt_mask), my_list);
          }
        }
        done += 256u;
      }
      if (done > waiting) done = waiting;
      // the (< 256) candidates that did not fill a pass move to the front and wait for the next one
      const uint32_t left = waiting - done;
      __syncthreads();  // every lane has read its candidate
      uint32_t mv_idx = 0;
      float mv[10];
      if ((uint32_t)tid < left) {
        mv_idx = sIdx[done + tid];
#pragma unroll
        for (int k = 0; k < 10; k++) mv[k] = sIn[k][done + tid];
      }
      __syncthreads();
      if ((uint32_t)tid < left) {
        sIdx[tid] = mv_idx;
#pragma unroll
        for (int k = 0; k < 10; k++) sIn[k][tid] = mv[k];
      }
      if (tid == 0) cand_tail = left;
      waiting = left;
      __syncthreads();
    }
  }
  const uint32_t wsum = gcr_wave_sum_u32(my_tiles);
  if (lane == 0 && wsum) atomicAdd(&blk_tiles, (unsigned long long)wsum);
  const int any_viol = __syncthreads_or(viol ? 1 : 0);
  if (tid == 0) {
    a.vis_count[blockIdx.x] = vis_tail;
    // summed (= num_rendered) by the first workgroup of the next kernel; bit 63: a prefilter violation
    a.block_tiles[blockIdx.x] = blk_tiles | (any_viol ? GCR_PREFILTER_FLAG : 0ull);
  }
}

// K1 fused, a static scene's variant (gcr_gaussians.cull_cache -- a second line, never the headline).  The stream is part A
// of the cache, ONE 16-byte (mean, rho) record per Gaussian instead of 40 B out of three arrays (56 B when the rows are
// [N,14]); a candidate waits in LDS with its index and mean and reads part B -- scales, opacity, rotation as ONE 32-byte
// record -- at the start of its processing pass, instead of a 128-byte line of each of three arrays.  The cull's decisions
// are those of the stateless kernel (the same mean, the same rho, the same test), A0 only ever skips Gaussians whose
// exact result is radius 0, and the records are copies: radii / lists / image are the same bits either way
// (tests/test_gpu_cull_cache.py).  K1 alone at C3: 86.8 -> 55.9 us, at C5: 365 -> 266 us
// (profiles/r05_k1_ab_same_process.jsonl).
//
// Its streaming loop runs in GROUPS of NS iterations with one register set per iteration of the group:
//   * stage k's registers are loaded for the NEXT group right after their last use in this one, so NS - 1 iterations of
//     records are in flight per wave and no register is ever copied (the stateless loop above rotates cur <- nxt <- nx2:
//     the copies read the loads' destination registers, so it waits for all but the newest loads -- measured as good
//     for a 40-byte stream, 86.8 vs 86.1 us with this structure, and better at C5, 365 vs 377 us, where this one's
//     candidates would have to re-read their scales / rotations);
//   * one barrier pair per group instead of one barrier per iteration; a processing pass runs at the end of a group when
//     256 candidates are waiting (all lanes busy) and at the end of the chunk; the queue holds a whole group's worth.
#ifndef GCR_K1_STAGES_CACHED  /* A/B builds: iterations per group of the cached stream (16 B per lane and iteration) */
#define GCR_K1_STAGES_CACHED 5
#endif

template <bool PRECOMP_COV>
__global__ __launch_bounds__(256) void k_preprocess_fused_cached(const GcrPreprocessArgs a) {
  constexpr int NS = GCR_K1_STAGES_CACHED;
  constexpr int CAP = 256 * (NS + 1);  // <= 255 left over + <= 256 NS new candidates per group
  __shared__ uint32_t sIdx[CAP];
  __shared__ float sP[3][CAP];
  __shared__ uint32_t cand_tail, vis_tail;
  __shared__ unsigned long long blk_tiles;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) {
    cand_tail = 0;
    vis_tail = 0;
    blk_tiles = 0ull;
  }
  __syncthreads();
  float vm[16], pm[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    vm[i] = gcr_uniform(GCR_CAM(a, view, a.view, i));
    pm[i] = gcr_uniform(GCR_CAM(a, proj, a.proj, i));
  }
  float wf2 = 0.0f;
  {
    const float wc[3][3] = {{vm[0], vm[1], vm[2]}, {vm[4], vm[5], vm[6]}, {vm[8], vm[9], vm[10]}};
#pragma unroll
    for (int i = 0; i < 3; i++) {
      float row = 0.0f;
#pragma unroll
      for (int j = 0; j < 3; j++)
        row += __builtin_fabsf(wc[0][i] * wc[0][j] + wc[1][i] * wc[1][j] + wc[2][i] * wc[2][j]);
      wf2 = __builtin_fmaxf(wf2, row);
    }
    wf2 *= 1.001f;
  }
  const long long chunk_begin = (long long)blockIdx.x * a.chunk;
  const long long chunk_end = chunk_begin + a.chunk < a.P ? chunk_begin + a.chunk : a.P;
  const long long clast = chunk_end - 1;  // (every block owns at least one Gaussian: gcr_preprocess_grid)
  uint32_t* __restrict__ my_list = a.vis_list + chunk_begin;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  uint32_t my_tiles = 0;
  bool viol = false;  // gcr_camera.prefiltered and a Gaussian behind the near plane

  float4 q[NS];  // (mean, rho) of this thread's Gaussian in iteration k of the group
#pragma unroll
  for (int k = 0; k < NS; k++) {
    const long long i = chunk_begin + 256 * k + tid;
    q[k] = a.cull_cache[i < clast ? i : clast];
  }
  for (long long gbase = chunk_begin; gbase < chunk_end; gbase += 256 * NS) {
#pragma unroll
    for (int k = 0; k < NS; k++) {
      const long long idx64 = gbase + 256 * k + tid;
      bool candidate = false;
      if (idx64 < chunk_end) {
        PhaseAIn in;
        in.p = {q[k].x, q[k].y, q[k].z};
        in.c0 = in.c1 = in.c2 = in.c3 = in.c4 = in.c5 = in.c6 = 0.0f;  // (the cull reads the mean and rho only)
        candidate = !phase_a0_certainly_culled<PRECOMP_COV>(a, vm, pm, wf2, in, q[k].w);
        if (!candidate) a.radii[idx64] = 0;
        if (a.prefiltered) viol |= near_plane_violation(vm, in);
      }
      const uint64_t m = __ballot(candidate);
      if (m != 0ull) {
        uint32_t wbase = 0;
        if (lane == 0) wbase = atomicAdd(&cand_tail, (uint32_t)__popcll(m));
        wbase = __shfl(wbase, 0, 64);
        if (candidate) {
          const uint32_t slot = wbase + (uint32_t)__popcll(m & lt_mask);
          sIdx[slot] = (uint32_t)idx64;
          sP[0][slot] = q[k].x; sP[1][slot] = q[k].y; sP[2][slot] = q[k].z;
        }
      }
      // stage k's registers are free now: the same iteration of the NEXT group goes into them (the compiler must not
      // hoist the load above the uses -- it would need a second register set and copy it at the back-edge)
      asm volatile("" ::: "memory");
      const long long inext = idx64 + 256 * NS;
      q[k] = a.cull_cache[inext < clast ? inext : clast];
    }
    __syncthreads();
    const uint32_t waiting = cand_tail;  // block-uniform: nobody appends again before the barrier(s) below
    const bool last_group = gbase + 256 * NS >= chunk_end;
    if (waiting >= 256u || (last_group && waiting > 0u)) {
      // ---- processing passes over the waiting candidates: full 256-lane passes, plus the remainder at the end
      uint32_t done = 0;
      while (done + 256u <= waiting || (last_group && done < waiting)) {
        const uint32_t it = done + (uint32_t)tid;
        bool keep = false;
        int idx = 0, radius = 0;
        float opac = 0.0f;
        PhaseAIn in;
        Projected pr;
        if (it < waiting) {
          idx = (int)sIdx[it];
          in.p = {sP[0][it], sP[1][it], sP[2][it]};
          {  // ONE 32-byte record: scales (or covariance), opacity, rotation
            const float4 b0 = a.cull_shape[2 * (size_t)idx], b1 = a.cull_shape[2 * (size_t)idx + 1];
            if (PRECOMP_COV) {
              in.c0 = b0.x; in.c1 = b0.y; in.c2 = b0.z; in.c3 = b0.w; in.c4 = b1.x; in.c5 = b1.y; in.c6 = 0.0f;
              opac = b1.z;
            } else {
              in.c0 = b0.x; in.c1 = b0.y; in.c2 = b0.z; opac = b0.w;
              in.c3 = b1.x; in.c4 = b1.y; in.c5 = b1.z; in.c6 = b1.w;
            }
          }
          keep = phase_a1_exact<PRECOMP_COV>(a, vm, pm, idx, in, pr, radius);
          a.radii[idx] = radius;
        }
        const uint64_t mk = __ballot(keep);
        if (mk != 0ull) {
          uint32_t lbase = 0;
          if (lane == 0) lbase = atomicAdd(&vis_tail, (uint32_t)__popcll(mk));
          lbase = __shfl(lbase, 0, 64);
          if (keep) {
            my_tiles += ((pr.rect_x >> 16) - (pr.rect_x & 0xffffu)) * ((pr.rect_y >> 16) - (pr.rect_y & 0xffffu));
            preprocess_phase_b<true>(a, idx, in.p, pr, lbase + (uint32_t)__popcll(mk & lt_mask), my_list, opac);
          }
        }
        done += 256u;
      }
      if (done > waiting) done = waiting;
      // the (< 256) candidates that did not fill a pass move to the front and wait for the next one
      const uint32_t left = waiting - done;
      __syncthreads();  // every lane has read its candidate
      uint32_t mv_idx = 0;
      float mv[3];
      if ((uint32_t)tid < left) {
        mv_idx = sIdx[done + tid];
#pragma unroll
        for (int k = 0; k < 3; k++) mv[k] = sP[k][done + tid];
      }
      __syncthreads();
      if ((uint32_t)tid < left) {
        sIdx[tid] = mv_idx;
#pragma unroll
        for (int k = 0; k < 3; k++) sP[k][tid] = mv[k];
      }
      if (tid == 0) cand_tail = left;
    }
    __syncthreads();  // everybody has read cand_tail (and sees the compacted queue) before the next group appends
  }
  const uint32_t wsum = gcr_wave_sum_u32(my_tiles);
  if (lane == 0 && wsum) atomicAdd(&blk_tiles, (unsigned long long)wsum);
  const int any_viol = __syncthreads_or(viol ? 1 : 0);
  if (tid == 0) {
    a.vis_count[blockIdx.x] = vis_tail;
    // summed (= num_rendered) by the first workgroup of the next kernel; bit 63: a prefilter violation
    a.block_tiles[blockIdx.x] = blk_tiles | (any_viol ? GCR_PREFILTER_FLAG : 0ull);
  }
}

// ------------------------------------------------------------------------------------- K2
// Exclusive scan (in place) of the n per-block tile counts; *total = num_rendered
// (cr/rasterizer_impl.cu:228-238 scans all P counts; the per-Gaussian offsets are rebuilt
// inside K3 from a block-local scan).  Single 1024-thread block: n = P/256 is small.
__global__ __launch_bounds__(1024) void k_scan_block_sums(uint32_t* __restrict__ sums, int n,
                                                          unsigned long long* __restrict__ total) {
  __shared__ unsigned long long part[1024];
  const int tid = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int beg = min(n, tid * per), end = min(n, beg + per);
  unsigned long long s = 0;
  for (int i = beg; i < end; i++) s += sums[i];
  part[tid] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    unsigned long long t = (tid >= o) ? part[tid - o] : 0ull;
    __syncthreads();
    part[tid] += t;
    __syncthreads();
  }
  unsigned long long run = part[tid] - s;
  for (int i = beg; i < end; i++) {
    const uint32_t t = sums[i];
    sums[i] = (uint32_t)run;
    run += t;
  }
  if (tid == 1023) *total = part[1023];
}

// Tile-count scan for the counting-sort binning: in: cursor[t*stride] = #instances of tile t;
// out: ranges[t] = [start,end) (what identifyTileRanges produces upstream,
// cr/rasterizer_impl.cu:104-124), cursor[t*stride] = start, frame[0] = num_rendered,
// frame[1] = longest tile list, frame[2] = 1 if the speculatively enqueued rest of the frame may
// run (num_rendered <= cap_instances and longest list <= cap_list), else 0.
// One 1024-thread block; thread i owns the contiguous slice [i*per, (i+1)*per) so that all its
// loads are in flight at once; wave-shuffle scan of the 1024 slice sums -> two barriers in total.
__global__ __launch_bounds__(1024) void k_scan_tiles(uint32_t* __restrict__ cursor, int stride,
                                                     uint32_t* __restrict__ ranges, int T,
                                                     unsigned long long* __restrict__ frame,
                                                     unsigned long long cap_instances,
                                                     unsigned long long cap_list,
                                                     unsigned long long* __restrict__ host_R, unsigned int seq,
                                                     const unsigned long long* __restrict__ block_tiles, int nblocks_k1) {
  __shared__ unsigned long long wsum[16];
  __shared__ uint32_t wmax[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  unsigned long long pf = 0ull;  // a K1 block met a prefilter violation (GCR_PREFILTER_FLAG in its share)
  for (int b = tid; b < nblocks_k1; b += 1024) pf |= block_tiles[b] & GCR_PREFILTER_FLAG;
  const int prefilter_violation = __syncthreads_or(pf != 0ull ? 1 : 0);
  const int per = (T + 1023) / 1024;
  const int beg = min(T, tid * per), end = min(T, beg + per);
  unsigned long long s = 0;
  uint32_t mx = 0;
  for (int i = beg; i < end; i++) {
    const uint32_t c = cursor[(size_t)i * stride];
    s += c;
    mx = c > mx ? c : mx;
  }
  // inclusive scan of the slice sums across the wave (64-bit), then across the 16 waves
  unsigned long long incl = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  const uint32_t m = gcr_wave_max_u32(mx);
  if (lane == 63) wsum[w] = incl;
  if (lane == 0) wmax[w] = m;
  __syncthreads();
  unsigned long long before = 0, total = 0;
  uint32_t mm = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const unsigned long long v = wsum[k];
    if (k < w) before += v;
    total += v;
    mm = wmax[k] > mm ? wmax[k] : mm;
  }
  unsigned long long run = before + incl - s;
  for (int i = beg; i < end; i++) {
    const uint32_t c = cursor[(size_t)i * stride];
    // saturate: an overflowing frame is rejected on the host (num_rendered > 2^31-1)
    const unsigned long long e64 = run + c;
    const uint32_t st = run > 0xffffffffull ? 0xffffffffu : (uint32_t)run;
    ranges[2 * i + 0] = st;
    ranges[2 * i + 1] = e64 > 0xffffffffull ? 0xffffffffu : (uint32_t)e64;
    cursor[(size_t)i * stride] = st;
    run = e64;
  }
  if (tid == 0) {
    if (prefilter_violation) total = GCR_PREFILTER_MARK;  // (beyond every capacity: the rest of the frame is vetoed)
    if (host_R != nullptr)  // (frame tag << 32 | R) for the polling host thread, see k_tile_table<false>
      gcr_store_to_host(host_R, ((unsigned long long)seq << 32) | (total > 0xffffffffull ? 0xffffffffull : total));
    frame[0] = total;
    frame[1] = mm;
    frame[2] = (total <= cap_instances && (unsigned long long)mm <= cap_list) ? 1ull : 0ull;
    frame[GCR_FRAME_PIECE] = 0ull;  // no backward state yet (set by a forward blend that writes it)
  }
}

// ------------------------------------------------------------------------------------- K8
// Fused cr/backward.cu:143-293 (computeCov2DCUDA, "K8a") and :378-425 (preprocessCUDA, "K8b").
// K8a assigns dL_dmean3D, K8b adds the projection and SH terms -- done here in registers in
// the same order: cov2D part, + projection part, + SH part.
__global__ __launch_bounds__(256) void k_preprocess_bwd(const GcrPreprocessBwdArgs a) {
  // dense walk over K1's survivors (== the Gaussians with radii > 0 upstream, cr/backward.cu:151,387)
  // There are only P_v threads of work here (C2: 84 k = 1.3 waves per SIMD), so the kernel lasts as long as one
  // thread's chain of memory round trips.  Left to the compiler that chain was nine deep (index; mean/cov/record;
  // the camera matrices re-fetched with vector loads every iteration because the stores could alias them; SH
  // coefficients degree by degree behind each `deg >` branch; scale and rotation at the very end).  Here: matrices
  // once per kernel into SGPRs, and every per-Gaussian input requested right after the index -- two round trips.
  float vm[16], proj[16], cp[3];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    vm[i] = gcr_uniform(GCR_CAM(a, view, a.view, i));
    proj[i] = gcr_uniform(GCR_CAM(a, proj, a.proj, i));
  }
#pragma unroll
  for (int i = 0; i < 3; i++) cp[i] = gcr_uniform(GCR_CAM(a, campos, a.campos, i));
  // gcr_backward on a frame whose forward left no backward state (gcr_camera.backward == 0), or whose state does not
  // fit the buffer handed in: the blend gradient kernel did nothing -- say so with NaN, never with plausible zeros
  const bool poison = a.frame != nullptr && !gcr_frame_has_state(a.frame, a.binning_bytes);
  const uint32_t nvis = a.vis_count[blockIdx.x];
  const uint32_t* __restrict__ my_list = a.vis_list + (size_t)blockIdx.x * a.chunk;
  for (uint32_t it = threadIdx.x; it < nvis; it += 256) {
  const int idx = (int)my_list[it];
  const float* __restrict__ mp = a.means3D + (size_t)idx * a.s_mean;
  const V3 mean = {mp[0], mp[1], mp[2]};
  float cv[6];  // the caller's precomputed covariance; else derived below, once scales and rotation are here
  if (a.scales == nullptr) {
#pragma unroll
    for (int i = 0; i < 6; i++) cv[i] = a.cov3D[(size_t)a.s_cov3d * idx + i];
  }
  float shv[48];  // SH coefficients [i][channel] (zeros beyond M)
  uint8_t cl = 0;
  if (a.shs != nullptr) {
    const float* __restrict__ shp = a.shs + (size_t)idx * a.M * 3;
    if (a.M == 16) {
      const float4* __restrict__ sh4 = reinterpret_cast<const float4*>(shp);
#pragma unroll
      for (int q = 0; q < 12; q++) {
        const float4 v = sh4[q];
        shv[4 * q] = v.x; shv[4 * q + 1] = v.y; shv[4 * q + 2] = v.z; shv[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 48; q++) shv[q] = q < 3 * a.M ? shp[q] : 0.0f;
    }
    cl = (uint8_t)__float_as_uint(a.rec[(size_t)idx * GCR_REC_QUADS + 3].x);  // the clamp mask K1 left in the record
  }
  float4 rot = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  float scl[3] = {0.0f, 0.0f, 0.0f};
  if (a.scales != nullptr) {
    const float* __restrict__ rp = a.rotations + (size_t)idx * a.s_rot;
    rot = a.s_rot == 4 ? *reinterpret_cast<const float4*>(rp) : make_float4(rp[0], rp[1], rp[2], rp[3]);
#pragma unroll
    for (int i = 0; i < 3; i++) scl[i] = a.scales[(size_t)idx * a.s_scale + i];
    // what K1 computed for this Gaussian (cr/forward.cu:110-143), from the same inputs with the same function: the same bits
    const V3 sc3 = {scl[0], scl[1], scl[2]};
    compute_cov3d(sc3, a.scale_modifier, rot, cv);
  }
  // K7's accumulation record of this Gaussian (gcr_internal.h); the API's per-Gaussian outputs of the
  // blend gradient are written from it here
  float4 g0, g1, g2;  // dcolor.rgb, dopacity | dmean2D.xy, dconic.x, dconic.y | dconic.w
  if (!a.deterministic) {
    g0 = a.grad_rec[(size_t)idx * (GCR_GRAD_REC_FLOATS / 4) + 0];
    g1 = a.grad_rec[(size_t)idx * (GCR_GRAD_REC_FLOATS / 4) + 1];
    g2 = a.grad_rec[(size_t)idx * (GCR_GRAD_REC_FLOATS / 4) + 2];
  } else {  // nine fixed-point sums with the Gaussian's own binary point (option "deterministic_backward")
    const long long* __restrict__ r64 =
        reinterpret_cast<const long long*>(a.grad_rec) + (size_t)idx * (GCR_GRAD_REC_FLOATS_DET / 2);
    const long long kslot = r64[GCR_DET_K_SLOT];
    const int kc = kslot != 0 ? (int)(kslot & 0xff) - 64 : 32;  // (never flushed: every sum is zero)
    const int ko = kslot != 0 ? (int)((kslot >> 8) & 0xff) - 64 : 32;
    float f[9];
#pragma unroll
    for (int k = 0; k < 9; k++) f[k] = (float)__builtin_ldexp((double)r64[k], k >= 4 ? -kc : -ko);
    g0 = make_float4(f[0], f[1], f[2], f[3]);
    g1 = make_float4(f[4], f[5], f[6], f[7]);
    g2 = make_float4(f[8], 0.0f, 0.0f, 0.0f);
  }
  if (poison) {
    const float nan = __builtin_nanf("");
    g0 = make_float4(nan, nan, nan, nan);
    g1 = g0;
    g2 = g0;
  }
  // The record holds MOMENTS of u = G * dL/dalpha over the Gaussian's pixels (gcr_blend.hip "MOMENTS", gcr_internal.h):
  // g0.w = S, g1 = (Sx, Sy, Sxx, Sxy), g2.x = Syy.  The per-Gaussian factors of cr/backward.cu:540-575 -- opacity (dL_dG =
  // o dL/dalpha), the conic in dG/d(delta), -0.5, the pixel scale W/2, H/2 -- are applied here, once per Gaussian.
  const float4 pr0 = a.rec[(size_t)idx * GCR_REC_QUADS], pr1 = a.rec[(size_t)idx * GCR_REC_QUADS + 1];
  const float con_x = pr0.z, con_y = pr0.w, con_z = pr1.x, opac = pr1.y;
  const float ddelx_dx = (float)(0.5 * a.W), ddely_dy = (float)(0.5 * a.H);
  const float hop = -0.5f * opac;
  // A Gaussian no pixel consumed has five exact zeros here, and upstream -- which adds per-pixel terms and has none to add --
  // leaves its gradients 0 whatever its record holds: the factors are skipped, so that a non-finite opacity or conic of
  // such a Gaussian does not turn 0 into NaN.  (A NaN moment compares unequal to 0: it goes through.)
  const bool no_moment = g1.x == 0.0f && g1.y == 0.0f && g1.z == 0.0f && g1.w == 0.0f && g2.x == 0.0f;
  const float dcx = no_moment ? 0.0f : hop * g1.z, dcy = no_moment ? 0.0f : hop * g1.w, dcz = no_moment ? 0.0f : hop * g2.x;
  const float dm2x = no_moment ? 0.0f : -(opac * ddelx_dx) * (con_x * g1.x + con_y * g1.y);
  const float dm2y = no_moment ? 0.0f : -(opac * ddely_dy) * (con_z * g1.y + con_y * g1.x);
  a.dL_dmean2D[3 * (size_t)idx] = dm2x;
  a.dL_dmean2D[3 * (size_t)idx + 1] = dm2y;
  a.dL_dcolor[(size_t)idx * a.g_col] = g0.x;
  a.dL_dcolor[(size_t)idx * a.g_col + 1] = g0.y;
  a.dL_dcolor[(size_t)idx * a.g_col + 2] = g0.z;
  a.dL_dopacity[(size_t)idx * a.g_opac] = g0.w;

  // ---- K8a
  Cov2DCtx c;
  cov2d_setup(mean, a.focal_x, a.focal_y, a.tanfovx, a.tanfovy, vm, c);
  const float x_grad_mul = (c.txtz < -c.limx || c.txtz > c.limx) ? 0.f : 1.f;
  const float y_grad_mul = (c.tytz < -c.limy || c.tytz > c.limy) ? 0.f : 1.f;
  float cov[3];
  cov2d_eval(c, cv, cov);
  const float ca = cov[0], cb = cov[1], cc = cov[2];
  const float denom = ca * cc - cb * cb;
  float dL_da = 0, dL_db = 0, dL_dc = 0;
  const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
  float dcov[6];
#define T(i, j) c.T[i][j]
  if (denom2inv != 0) {
    dL_da = denom2inv * (-cc * cc * dcx + 2 * cb * cc * dcy + (denom - ca * cc) * dcz);
    dL_dc = denom2inv * (-ca * ca * dcz + 2 * ca * cb * dcy + (denom - ca * cc) * dcx);
    dL_db = denom2inv * 2 * (cb * cc * dcx - (denom + 2 * cb * cb) * dcy + ca * cb * dcz);
    dcov[0] = (T(0, 0) * T(0, 0) * dL_da + T(0, 0) * T(1, 0) * dL_db + T(1, 0) * T(1, 0) * dL_dc);
    dcov[3] = (T(0, 1) * T(0, 1) * dL_da + T(0, 1) * T(1, 1) * dL_db + T(1, 1) * T(1, 1) * dL_dc);
    dcov[5] = (T(0, 2) * T(0, 2) * dL_da + T(0, 2) * T(1, 2) * dL_db + T(1, 2) * T(1, 2) * dL_dc);
    dcov[1] = 2 * T(0, 0) * T(0, 1) * dL_da + (T(0, 0) * T(1, 1) + T(0, 1) * T(1, 0)) * dL_db +
              2 * T(1, 0) * T(1, 1) * dL_dc;
    dcov[2] = 2 * T(0, 0) * T(0, 2) * dL_da + (T(0, 0) * T(1, 2) + T(0, 2) * T(1, 0)) * dL_db +
              2 * T(1, 0) * T(1, 2) * dL_dc;
    dcov[4] = 2 * T(0, 2) * T(0, 1) * dL_da + (T(0, 1) * T(1, 2) + T(0, 2) * T(1, 1)) * dL_db +
              2 * T(1, 1) * T(1, 2) * dL_dc;
  } else {
#pragma unroll
    for (int i = 0; i < 6; i++) dcov[i] = 0;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) a.dL_dcov3D[6 * (size_t)idx + i] = dcov[i];

  const float V[3][3] = {{cv[0], cv[1], cv[2]}, {cv[1], cv[3], cv[4]}, {cv[2], cv[4], cv[5]}};
#define ROWDOT(i, k) (T(i, 0) * V[k][0] + T(i, 1) * V[k][1] + T(i, 2) * V[k][2])
  const float dL_dT00 = 2 * ROWDOT(0, 0) * dL_da + ROWDOT(1, 0) * dL_db;
  const float dL_dT01 = 2 * ROWDOT(0, 1) * dL_da + ROWDOT(1, 1) * dL_db;
  const float dL_dT02 = 2 * ROWDOT(0, 2) * dL_da + ROWDOT(1, 2) * dL_db;
  const float dL_dT10 = 2 * ROWDOT(1, 0) * dL_dc + ROWDOT(0, 0) * dL_db;
  const float dL_dT11 = 2 * ROWDOT(1, 1) * dL_dc + ROWDOT(0, 1) * dL_db;
  const float dL_dT12 = 2 * ROWDOT(1, 2) * dL_dc + ROWDOT(0, 2) * dL_db;
#undef ROWDOT
#undef T
  const float dL_dJ00 = vm[0] * dL_dT00 + vm[4] * dL_dT01 + vm[8] * dL_dT02;
  const float dL_dJ02 = vm[2] * dL_dT00 + vm[6] * dL_dT01 + vm[10] * dL_dT02;
  const float dL_dJ11 = vm[1] * dL_dT10 + vm[5] * dL_dT11 + vm[9] * dL_dT12;
  const float dL_dJ12 = vm[2] * dL_dT10 + vm[6] * dL_dT11 + vm[10] * dL_dT12;
  const float h_x = a.focal_x, h_y = a.focal_y;
  const float tz = 1.f / c.t.z;
  const float tz2 = tz * tz;
  const float tz3 = tz2 * tz;
  const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
  const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
  const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * c.t.x) * tz3 * dL_dJ02 +
                       (2 * h_y * c.t.y) * tz3 * dL_dJ12;
  float dmx = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;
  float dmy = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
  float dmz = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;

  // ---- K8b: 2D mean -> 3D mean (cr/backward.cu:392-413)
  {
    const float mhw = proj[3] * mean.x + proj[7] * mean.y + proj[11] * mean.z + proj[15];
    const float m_w = 1.0f / (mhw + 0.0000001f);
    const float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
    const float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
    const float d2x = dm2x, d2y = dm2y;
    const float ax = (proj[0] * m_w - proj[3] * mul1) * d2x + (proj[1] * m_w - proj[3] * mul2) * d2y;
    const float ay = (proj[4] * m_w - proj[7] * mul1) * d2x + (proj[5] * m_w - proj[7] * mul2) * d2y;
    const float az = (proj[8] * m_w - proj[11] * mul1) * d2x + (proj[9] * m_w - proj[11] * mul2) * d2y;
    dmx += ax;
    dmy += ay;
    dmz += az;
  }

  // ---- K8b: SH backward (cr/backward.cu:20-138)
  if (a.shs != nullptr) {
    const float ox = mean.x - cp[0], oy = mean.y - cp[1], oz = mean.z - cp[2];
    const float len = __builtin_sqrtf(ox * ox + oy * oy + oz * oz);
    const float x = ox / len, y = oy / len, z = oz / len;
    float* __restrict__ dsh = a.dL_dsh + (size_t)idx * a.M * 3;
    float dRGB[3];
    const float dcol[3] = {g0.x, g0.y, g0.z};
#pragma unroll
    for (int ch = 0; ch < 3; ch++) dRGB[ch] = dcol[ch] * (((cl >> ch) & 1) ? 0.f : 1.f);
    float gdx[3] = {0, 0, 0}, gdy[3] = {0, 0, 0}, gdz[3] = {0, 0, 0};
    const int deg = a.D;
#define SHV(i) shv[3 * (i) + ch]
#define DSH(i, w)                  \
  _Pragma("unroll") for (int ch = 0; ch < 3; ch++) dsh[3 * (i) + ch] = (w) * dRGB[ch]
    DSH(0, SH_C0);
    if (deg > 0) {
      const float d1 = -SH_C1 * y, d2 = SH_C1 * z, d3 = -SH_C1 * x;
      DSH(1, d1);
      DSH(2, d2);
      DSH(3, d3);
#pragma unroll
      for (int ch = 0; ch < 3; ch++) {
        gdx[ch] = -SH_C1 * SHV(3);
        gdy[ch] = -SH_C1 * SHV(1);
        gdz[ch] = SH_C1 * SHV(2);
      }
      if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z;
        const float xy = x * y, yz = y * z, xz = x * z;
        const float d4 = SH_C2[0] * xy, d5 = SH_C2[1] * yz, d6 = SH_C2[2] * (2.f * zz - xx - yy),
                    d7 = SH_C2[3] * xz, d8 = SH_C2[4] * (xx - yy);
        DSH(4, d4);
        DSH(5, d5);
        DSH(6, d6);
        DSH(7, d7);
        DSH(8, d8);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
          gdx[ch] += SH_C2[0] * y * SHV(4) + SH_C2[2] * 2.f * -x * SHV(6) + SH_C2[3] * z * SHV(7) +
                     SH_C2[4] * 2.f * x * SHV(8);
          gdy[ch] += SH_C2[0] * x * SHV(4) + SH_C2[1] * z * SHV(5) + SH_C2[2] * 2.f * -y * SHV(6) +
                     SH_C2[4] * 2.f * -y * SHV(8);
          gdz[ch] += SH_C2[1] * y * SHV(5) + SH_C2[2] * 2.f * 2.f * z * SHV(6) + SH_C2[3] * x * SHV(7);
        }
        if (deg > 2) {
          const float d9 = SH_C3[0] * y * (3.f * xx - yy), d10 = SH_C3[1] * xy * z,
                      d11 = SH_C3[2] * y * (4.f * zz - xx - yy),
                      d12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy),
                      d13 = SH_C3[4] * x * (4.f * zz - xx - yy), d14 = SH_C3[5] * z * (xx - yy),
                      d15 = SH_C3[6] * x * (xx - 3.f * yy);
          DSH(9, d9);
          DSH(10, d10);
          DSH(11, d11);
          DSH(12, d12);
          DSH(13, d13);
          DSH(14, d14);
          DSH(15, d15);
#pragma unroll
          for (int ch = 0; ch < 3; ch++) {
            gdx[ch] += (SH_C3[0] * SHV(9) * 3.f * 2.f * xy + SH_C3[1] * SHV(10) * yz +
                        SH_C3[2] * SHV(11) * -2.f * xy + SH_C3[3] * SHV(12) * -3.f * 2.f * xz +
                        SH_C3[4] * SHV(13) * (-3.f * xx + 4.f * zz - yy) +
                        SH_C3[5] * SHV(14) * 2.f * xz + SH_C3[6] * SHV(15) * 3.f * (xx - yy));
            gdy[ch] += (SH_C3[0] * SHV(9) * 3.f * (xx - yy) + SH_C3[1] * SHV(10) * xz +
                        SH_C3[2] * SHV(11) * (-3.f * yy + 4.f * zz - xx) +
                        SH_C3[3] * SHV(12) * -3.f * 2.f * yz + SH_C3[4] * SHV(13) * -2.f * xy +
                        SH_C3[5] * SHV(14) * -2.f * yz + SH_C3[6] * SHV(15) * -3.f * 2.f * xy);
            gdz[ch] += (SH_C3[1] * SHV(10) * xy + SH_C3[2] * SHV(11) * 4.f * 2.f * yz +
                        SH_C3[3] * SHV(12) * 3.f * (2.f * zz - xx - yy) +
                        SH_C3[4] * SHV(13) * 4.f * 2.f * xz + SH_C3[5] * SHV(14) * (xx - yy));
          }
        }
      }
    }
#undef SHV
#undef DSH
    const float ddx = gdx[0] * dRGB[0] + gdx[1] * dRGB[1] + gdx[2] * dRGB[2];
    const float ddy = gdy[0] * dRGB[0] + gdy[1] * dRGB[1] + gdy[2] * dRGB[2];
    const float ddz = gdz[0] * dRGB[0] + gdz[1] * dRGB[1] + gdz[2] * dRGB[2];
    // dnormvdv(float3), cr/auxiliary.h:97-113
    const float sum2 = ox * ox + oy * oy + oz * oz;
    const float invsum32 = 1.0f / __builtin_sqrtf(sum2 * sum2 * sum2);
    dmx += ((+sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * invsum32;
    dmy += (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * invsum32;
    dmz += (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * invsum32;
  }
  a.dL_dmean3D[(size_t)idx * a.g_mean] = dmx;
  a.dL_dmean3D[(size_t)idx * a.g_mean + 1] = dmy;
  a.dL_dmean3D[(size_t)idx * a.g_mean + 2] = dmz;

  // ---- K8b: cov3D -> scale / rotation (cr/backward.cu:297-373)
  if (a.scales != nullptr) {
    const float r = rot.x, x = rot.y, y = rot.z, z = rot.w;
    const float Rm[3][3] = {
        {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
        {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
        {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
    const float s[3] = {a.scale_modifier * scl[0], a.scale_modifier * scl[1], a.scale_modifier * scl[2]};
    float M2[3][3];
#pragma unroll
    for (int cidx = 0; cidx < 3; cidx++)
#pragma unroll
      for (int k = 0; k < 3; k++) M2[cidx][k] = 2.0f * (s[k] * Rm[cidx][k]);
    const float dS[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                            {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                            {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
    float Mt[3][3];  // dL_dMt[c][r] = dL_dM[r][c], dL_dM[c][r] = sum_k M2[k][r]*dS[c][k]
#pragma unroll
    for (int cidx = 0; cidx < 3; cidx++)
#pragma unroll
      for (int rr = 0; rr < 3; rr++)
        Mt[rr][cidx] = M2[0][rr] * dS[cidx][0] + M2[1][rr] * dS[cidx][1] + M2[2][rr] * dS[cidx][2];
#pragma unroll
    for (int cidx = 0; cidx < 3; cidx++)
      a.dL_dscale[(size_t)idx * a.g_scale + cidx] =
          Rm[0][cidx] * Mt[cidx][0] + Rm[1][cidx] * Mt[cidx][1] + Rm[2][cidx] * Mt[cidx][2];
#pragma unroll
    for (int cidx = 0; cidx < 3; cidx++)
#pragma unroll
      for (int rr = 0; rr < 3; rr++) Mt[cidx][rr] *= s[cidx];
    float4 dq;
    dq.x = 2 * z * (Mt[0][1] - Mt[1][0]) + 2 * y * (Mt[2][0] - Mt[0][2]) + 2 * x * (Mt[1][2] - Mt[2][1]);
    dq.y = 2 * y * (Mt[1][0] + Mt[0][1]) + 2 * z * (Mt[2][0] + Mt[0][2]) + 2 * r * (Mt[1][2] - Mt[2][1]) -
           4 * x * (Mt[2][2] + Mt[1][1]);
    dq.z = 2 * x * (Mt[1][0] + Mt[0][1]) + 2 * r * (Mt[2][0] - Mt[0][2]) + 2 * z * (Mt[1][2] + Mt[2][1]) -
           4 * y * (Mt[2][2] + Mt[0][0]);
    dq.w = 2 * r * (Mt[0][1] - Mt[1][0]) + 2 * x * (Mt[2][0] + Mt[0][2]) + 2 * y * (Mt[1][2] + Mt[2][1]) -
           4 * z * (Mt[1][1] + Mt[0][0]);
    float* __restrict__ dqp = a.dL_drot + (size_t)idx * a.g_rot;
    if (a.g_rot == 4) {
      *reinterpret_cast<float4*>(dqp) = dq;
    } else {
      dqp[0] = dq.x; dqp[1] = dq.y; dqp[2] = dq.z; dqp[3] = dq.w;
    }
  }
}
}

}  // namespace

hipError_t gcr_launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present,
                                   hipStream_t s) {
  if (P <= 0) return hipSuccess;
  k_mark_visible<<<(P + 255) / 256, 256, 0, s>>>(P, means3D, view, present);
  return hipGetLastError();
}

// Grid of the persistent K1 kernels (and of every kernel that walks K1's per-block survivor lists).
// A MEASURED constant, not an occupancy premise: 8 workgroups per CU, at most GCR_K1_MAX_BLOCKS.  That is one
// resident round of the register-light streaming cull (split mode) and TWO rounds of the fused kernel (119 VGPRs,
// 22.5 KB LDS: four workgroups per CU) -- for the fused kernel many short blocks measured better than one resident
// round (4 660 vs 4 470 frames/s at C3: short blocks co-schedule with the other frames' blend workgroups, and a
// block's latency-bound processing pass is covered by its neighbours' streaming).  Both modes use the same grid so
// that the survivor-list layout in the geometry buffer does not depend on the mode (the backward re-derives it).
int gcr_preprocess_resident_blocks(bool split) {
  (void)split;
  static int cached = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    }
    int n = 8 * cus;
    if (const char* e = getenv("GCR_K1_BLOCKS")) n = atoi(e);  // experiments only
    return n > GCR_K1_MAX_BLOCKS ? GCR_K1_MAX_BLOCKS : (n < 1 ? 1 : n);
  }();
  return cached;
}

// gcr_build_cull_cache.  Two arrays: A[i] = (mean, rho) -- everything the streaming cull needs of Gaussian i, 16 bytes; B[i] =
// (scales, opacity, rotation) or (covariance, opacity, 0) -- everything a candidate needs besides, ONE 32-byte record instead
// of a 128-byte line each of scales, rotations and opacities.  Only the inputs, their strides, P and scale_modifier of `a`
// are read.
template <bool PRECOMP_COV>
__global__ __launch_bounds__(256) void k_build_cull_cache(const GcrPreprocessArgs a, float4* __restrict__ outA,
                                                          float4* __restrict__ outB) {
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < (long long)a.P; idx += (long long)gridDim.x * 256) {
    PhaseAIn in;
    phase_a_load<PRECOMP_COV>(a, idx, in);
    const float opacity = a.opacities[(size_t)idx * a.s_opac];
    outA[idx] = make_float4(in.p.x, in.p.y, in.p.z, cull_rho<PRECOMP_COV>(a.scale_modifier, in));
    if (PRECOMP_COV) {
      outB[2 * idx] = make_float4(in.c0, in.c1, in.c2, in.c3);
      outB[2 * idx + 1] = make_float4(in.c4, in.c5, opacity, 0.0f);
    } else {
      outB[2 * idx] = make_float4(in.c0, in.c1, in.c2, opacity);
      outB[2 * idx + 1] = make_float4(in.c3, in.c4, in.c5, in.c6);
    }
  }
}

hipError_t gcr_launch_build_cull_cache(const GcrPreprocessArgs& a, float4* outA, float4* outB, hipStream_t s) {
  if (a.P <= 0) return hipSuccess;
  long long nb = ((long long)a.P + 255) / 256;
  if (nb > 8192) nb = 8192;
  if (a.cov3D_precomp != nullptr)
    k_build_cull_cache<true><<<(unsigned int)nb, 256, 0, s>>>(a, outA, outB);
  else
    k_build_cull_cache<false><<<(unsigned int)nb, 256, 0, s>>>(a, outA, outB);
  return hipGetLastError();
}

hipError_t gcr_launch_preprocess(const GcrPreprocessArgs& a, bool split, hipStream_t s) {
  if (a.P <= 0) return hipSuccess;
  if (!split) {  // default: streaming cull and exact pass in one kernel
    if (a.cull_cache != nullptr) {  // a static scene's cull cache (gcr_gaussians.cull_cache): same results, fewer bytes
      if (a.cov3D_precomp != nullptr)
        k_preprocess_fused_cached<true><<<a.nblocks, 256, 0, s>>>(a);
      else
        k_preprocess_fused_cached<false><<<a.nblocks, 256, 0, s>>>(a);
    } else if (a.cov3D_precomp != nullptr) {
      if (a.nt_stream) k_preprocess_fused<true, true><<<a.nblocks, 256, 0, s>>>(a);
      else k_preprocess_fused<true, false><<<a.nblocks, 256, 0, s>>>(a);
    } else {
      if (a.nt_stream) k_preprocess_fused<false, true><<<a.nblocks, 256, 0, s>>>(a);
      else k_preprocess_fused<false, false><<<a.nblocks, 256, 0, s>>>(a);
    }
    return hipGetLastError();
  }
  if (a.cov3D_precomp != nullptr) {
    k_preprocess_cull<true><<<a.nblocks, 256, 0, s>>>(a);
    k_preprocess_project<true><<<a.nblocks, 256, 0, s>>>(a);
  } else {
    k_preprocess_cull<false><<<a.nblocks, 256, 0, s>>>(a);
    k_preprocess_project<false><<<a.nblocks, 256, 0, s>>>(a);
  }
  return hipGetLastError();
}

hipError_t gcr_launch_scan_block_sums(uint32_t* block_sums, int n, unsigned long long* total,
                                      hipStream_t s) {
  k_scan_block_sums<<<1, 1024, 0, s>>>(block_sums, n, total);
  return hipGetLastError();
}

hipError_t gcr_launch_scan_tiles(uint32_t* tile_cursor, int stride, uint32_t* ranges, int T,
                                 unsigned long long* frame, unsigned long long cap_instances,
                                 unsigned long long cap_list, unsigned long long* host_R, unsigned int seq,
                                 const unsigned long long* block_tiles, int nblocks_k1, hipStream_t s) {
  k_scan_tiles<<<1, 1024, 0, s>>>(tile_cursor, stride, ranges, T, frame, cap_instances, cap_list, host_R, seq, block_tiles, nblocks_k1);
  return hipGetLastError();
}

namespace {
// K7 adds into the records of the Gaussians in its tile lists, all of them survivors of K1: those are the only
// records that have to start at zero (and the only ones K8 reads).
__global__ __launch_bounds__(256) void k_zero_grad_records(int chunk, const uint32_t* __restrict__ vis_list,
                                                           const uint32_t* __restrict__ vis_count,
                                                           float4* __restrict__ grad_rec, int rec_quads) {
  const uint32_t nvis = vis_count[blockIdx.x];
  const uint32_t* __restrict__ my_list = vis_list + (size_t)blockIdx.x * chunk;
  const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  // rec_quads (4, or 8 for the fixed-point records) lanes per record: one 64-byte line per quad of lanes
  for (uint32_t it = threadIdx.x; it < nvis * (uint32_t)rec_quads; it += 256)
    grad_rec[(size_t)my_list[it / (uint32_t)rec_quads] * (uint32_t)rec_quads + (it % (uint32_t)rec_quads)] = z;
}

__global__ __launch_bounds__(256) void k_fill_zero(const GcrFillArgs f) {
  for (int sgi = 0; sgi < f.nseg; sgi++)
    gcr_fill_zero_segment(f.ptr[sgi], f.n[sgi], (int)blockIdx.x, f.blocks, (int)threadIdx.x);
}
}  // namespace

hipError_t gcr_launch_zero_grad_records(int nblocks, int chunk, const uint32_t* vis_list, const uint32_t* vis_count,
                                        float4* grad_rec, int rec_quads, hipStream_t s) {
  if (nblocks <= 0) return hipSuccess;
  k_zero_grad_records<<<nblocks, 256, 0, s>>>(chunk, vis_list, vis_count, grad_rec, rec_quads);
  return hipGetLastError();
}

hipError_t gcr_launch_fill(const GcrFillArgs& f, hipStream_t s) {
  if (f.blocks <= 0 || f.nseg <= 0) return hipSuccess;
  k_fill_zero<<<f.blocks, 256, 0, s>>>(f);
  return hipGetLastError();
}

hipError_t gcr_launch_preprocess_bwd(const GcrPreprocessBwdArgs& a, hipStream_t s) {
  if (a.P <= 0) return hipSuccess;
  k_preprocess_bwd<<<a.nblocks, 256, 0, s>>>(a);
  return hipGetLastError();
}
