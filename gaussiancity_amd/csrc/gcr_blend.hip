// gcr_blend.hip -- K6 forward alpha compositing and K7 reverse-walk gradient for gfx950.
//
// One 256-thread workgroup (4 x wave64) per 16x16 tile; wave w owns the 8x8-pixel QUADRANT
// (w&1, w>>1) of the tile.  A tile's depth-sorted list is consumed in chunks of 256 entries: the
// 256 threads gather one 48-byte Gaussian record each (3 x dwordx4 from <= 2 cache lines) and
// stage it in LDS as SoA quads, so the blend loop reads wave-uniform (broadcast) LDS and never
// touches global memory.  Differences from cr/forward.cu:238-346 that do not change results:
//   * colour is staged in LDS too (the reference gathers it per contributing pixel, :328);
//   * a per-Gaussian conservative bound pmin = -ln(255*opacity) - 1e-3 is staged; pixels with
//     power < pmin are skipped before exp() -- exactly pixels the alpha < 1/255 test (:318)
//     would skip anyway;
//   * QUADRANT CULLING: the staging thread also computes the axis-aligned box of the ellipse
//     {power >= pmin} (inflated by 0.1 % + 0.01 px) and a 4-bit mask of the quadrants it
//     overlaps.  Each wave ballot-compacts the chunk into a 256-bit set and walks only its own
//     entries with a scalar bit scan.  An entry outside a quadrant's box has power < pmin for
//     all 64 pixels, i.e. it is skipped by every pixel upstream as well, and skipped entries
//     never change a pixel's state -- so n_contrib/final_T/colour are unchanged;
//   * `contributor` is the (wave-uniform) list position instead of a per-lane counter;
//   * a wave stops as soon as its own 64 pixels are done, on top of the block vote (:284-286).
// Arithmetic: gcr-fp32-v1 (gcr_device.h) -> out_color / final_T / n_contrib are bit-identical
// to the oracle.
#include "gcr_device.h"
#include "gcr_internal.h"

namespace {

constexpr int CHUNK = 256;

// One staged list entry: 48 bytes so that a single address (j * 48) + immediate offsets serves
// the three wave-uniform (broadcast) reads of the blend loop.
struct __attribute__((aligned(16))) StagedEntry {
  float4 a;     // x, y, conic.x, conic.y
  float4 b;     // conic.z, opacity, r, g
  float2 c;     // b, pmin
  uint32_t id;  // Gaussian index (backward only)
  uint32_t mask;  // quadrant mask
};

// pmin such that power < pmin  =>  opacity*exp(power) < 1/255 with a 1e-3 safety margin.
// Clamped to >= -87 so the blend loops may use the guard-free exponential (see gcr_device.h).
GCR_DEV float gcr_alpha_skip_bound(float opacity) {
  if (!(opacity > 0.0f)) return __builtin_inff();  // alpha <= 0 < 1/255: always skipped
  return gcr_max(-87.0f, -__builtin_logf(255.0f * opacity) - 1.0e-3f);
}

// 4-bit mask of the 8x8 quadrants of tile (tile_x0, tile_y0) that the region {power >= pmin} of
// a Gaussian can reach.  Conservative: anything it cannot bound returns 0xF.
GCR_DEV uint32_t gcr_quadrant_mask(float gx, float gy, float cx, float cy, float cz, float pmin,
                                   float tile_x0, float tile_y0) {
  if (!(pmin < 0.0f)) return 0u;  // alpha < 1/255 everywhere (power <= 0 always)
  const float det = cx * cz - cy * cy;
  if (!(det > 0.0f)) return 0xFu;
  const float tau = -2.0f * pmin;
  float ex = __builtin_sqrtf(tau * cz / det), ey = __builtin_sqrtf(tau * cx / det);
  if (!(ex == ex) || !(ey == ey)) return 0xFu;
  ex = ex * 1.001f + 0.01f;
  ey = ey * 1.001f + 0.01f;
  const float lox = gx - ex, hix = gx + ex, loy = gy - ey, hiy = gy + ey;
  const bool x0 = hix >= tile_x0 && lox <= tile_x0 + 7.0f;
  const bool x1 = hix >= tile_x0 + 8.0f && lox <= tile_x0 + 15.0f;
  const bool y0 = hiy >= tile_y0 && loy <= tile_y0 + 7.0f;
  const bool y1 = hiy >= tile_y0 + 8.0f && loy <= tile_y0 + 15.0f;
  return (x0 && y0 ? 1u : 0u) | (x1 && y0 ? 2u : 0u) | (x0 && y1 ? 4u : 0u) | (x1 && y1 ? 8u : 0u);
}

template <bool FAST_EXP>
GCR_DEV float blend_exp(float x) {
  return FAST_EXP ? gcr_expf_fast(x) : gcr_expf_noguard(x);
}

// ------------------------------------------------------------------------------------- K6
// Scalar-unit budget: a CU has ONE scalar ALU for its four SIMDs, and the first versions of this
// loop were bound by it (exec-mask bookkeeping of nested branches, 64-bit bit scans, lane masks
// carried in SGPRs).  Hence
//   * each wave first compacts the indices of the chunk entries that can touch its quadrant into
//     a private LDS list, so the hot loop is a plain counted loop;
//   * the body is branch-free (selects), with one wave-uniform skip;
//   * "done" is folded into a working transmittance Tw that drops to 0 when the pixel is finished
//     (T*(1-a) < 1e-4 upstream): test_T = Tw*(1-a) is then 0 and the entry can never be `use`d;
//     Tout keeps the value upstream leaves in T.
template <bool FAST_EXP>
__global__ __launch_bounds__(256) void k_blend_fwd(const GcrBlendArgs a) {
  __shared__ StagedEntry sE[CHUNK];
  __shared__ uint2 sList[4][CHUNK + 2];  // per-wave compacted entries: {byte offset into sE,
                                         // contributor number = list position + 1}

  if (a.frame != nullptr && a.frame[2] == 0ull) return;  // speculative launch vetoed
  const int tile = blockIdx.x;
  const int tx = tile % a.gx, ty = tile / a.gx;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int pxi = tx * GCR_TILE_X + (w & 1) * 8 + (lane & 7);
  const int pyi = ty * GCR_TILE_Y + (w >> 1) * 8 + (lane >> 3);
  const bool inside = pxi < a.W && pyi < a.H;
  const float pixx = (float)pxi, pixy = (float)pyi;
  const float tile_x0 = (float)(tx * GCR_TILE_X), tile_y0 = (float)(ty * GCR_TILE_Y);
  const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
  const int total = (int)(r1 - r0);
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const char* const sEb = reinterpret_cast<const char*>(sE);

  // Working transmittance: Tw > 0 while the pixel is live and holds upstream's T; when the pixel
  // finishes (T*(1-a) < 1e-4 upstream) the sign is flipped, which keeps |Tw| = the T upstream
  // leaves behind while making test_T = Tw*(1-a) <= 0, so no later entry can be `use`d.
  float Tw = inside ? 1.0f : -1.0f;
  float C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
  uint32_t last_contributor = 0;

// One list entry (record QA/QB/QC, contributor CON) against this lane's pixel; sets `live` to
// whether any lane of the wave still has a live pixel.
#define GCR_BLEND_STEP(QA, QB, QC, CON)                                                      \
  {                                                                                          \
    const float dx = QA.x - pixx, dy = QA.y - pixy;                                          \
    const float power = gcr_power(QA.z, QA.w, QB.x, dx, dy);                                 \
    const bool in_range = !(power > 0.0f) && !(power < QC.y);                                \
    if (__ballot(in_range) != 0ull) { /* wave-uniform */                                     \
      /* lanes outside [pmin, 0] may produce garbage; every use below is behind a select */  \
      const float araw = __builtin_fminf(0.99f, QB.y * blend_exp<FAST_EXP>(power));          \
      const bool valid = in_range && !(araw < 1.0f / 255.0f);                                \
      const float test_T = Tw * (1 - araw);                                                  \
      const bool use = valid && !(test_T < 0.0001f);                                         \
      const float n0 = __builtin_fmaf(QB.z * araw, Tw, C0);                                  \
      const float n1 = __builtin_fmaf(QB.w * araw, Tw, C1);                                  \
      const float n2 = __builtin_fmaf(QC.x * araw, Tw, C2);                                  \
      C0 = use ? n0 : C0;                                                                    \
      C1 = use ? n1 : C1;                                                                    \
      C2 = use ? n2 : C2;                                                                    \
      last_contributor = use ? (CON) : last_contributor;                                     \
      Tw = use ? test_T : (valid ? -__builtin_fabsf(Tw) : Tw);                               \
      live = __ballot(Tw > 0.0f) != 0ull;                                                    \
    }                                                                                        \
  }
#define GCR_BLEND_LOAD(QA, QB, QC, OFF)                            \
  QA = *reinterpret_cast<const float4*>(sEb + (OFF));             \
  QB = *reinterpret_cast<const float4*>(sEb + (OFF) + 16);         \
  QC = *reinterpret_cast<const float2*>(sEb + (OFF) + 32);

  for (int base = 0; base < total; base += CHUNK) {
    // block-wide vote (cr/forward.cu:284-286); also fences the previous chunk's LDS reads
    if (__syncthreads_count(!(Tw > 0.0f)) == 256) break;
    const int n = min(CHUNK, total - base);
    uint32_t my_mask = 0;
    if (tid < n) {
      const uint32_t id = a.list[r0 + base + tid];
      const float4* __restrict__ rec = a.rec + (size_t)id * GCR_REC_QUADS;
      const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
      const float pmin = gcr_alpha_skip_bound(q1.y);
      sE[tid].a = q0;
      sE[tid].b = q1;
      sE[tid].c = make_float2(q2.x, pmin);
      my_mask = gcr_quadrant_mask(q0.x, q0.y, q0.z, q0.w, q1.x, pmin, tile_x0, tile_y0);
    }
    sE[tid].mask = my_mask;
    __syncthreads();
    if (__ballot(Tw > 0.0f) == 0ull) continue;  // this wave's quadrant is finished; keep voting
    // compact this wave's entries (ascending list order is preserved)
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int jj = k * 64 + lane;
      const bool rel = (sE[jj].mask >> w) & 1u;
      const uint64_t m = __ballot(rel);
      if (rel)
        sList[w][cnt + __popcll(m & lt_mask)] =
            make_uint2((uint32_t)(jj * (int)sizeof(StagedEntry)), (uint32_t)(base + jj + 1));
      cnt += __popcll(m);
    }
    // two pad slots so the pipeline below may read past the end (the values are never consumed)
    if (lane < 2) sList[w][cnt + lane] = make_uint2(0u, 0u);
    __builtin_amdgcn_wave_barrier();
    // Software pipeline, two entries per trip with ping-pong registers: while entry i blends,
    // the record of entry i+1 and the list slot of entry i+2 are already in flight.
    const uint2* lp = &sList[w][0];
    float4 qa0, qb0, qa1, qb1;
    float2 qc0, qc1;
    uint2 e0 = lp[0], e1 = lp[1];
    GCR_BLEND_LOAD(qa0, qb0, qc0, e0.x)
    bool live = true;
    for (int i = 0; i < cnt; i += 2, lp += 2) {
      GCR_BLEND_LOAD(qa1, qb1, qc1, e1.x)
      const uint32_t con0 = e0.y;
      e0 = lp[2];
      GCR_BLEND_STEP(qa0, qb0, qc0, con0)
      if (!live || i + 1 >= cnt) break;
      GCR_BLEND_LOAD(qa0, qb0, qc0, e0.x)
      const uint32_t con1 = e1.y;
      e1 = lp[3];
      GCR_BLEND_STEP(qa1, qb1, qc1, con1)
      if (!live) break;
    }
  }
#undef GCR_BLEND_STEP
#undef GCR_BLEND_LOAD
  if (inside) {
    const float Tout = __builtin_fabsf(Tw);
    const size_t pix_id = (size_t)a.W * pyi + pxi;
    const size_t plane = (size_t)a.H * a.W;
    a.final_T[pix_id] = Tout;
    a.n_contrib[pix_id] = last_contributor;
    a.out_color[pix_id] = C0 + Tout * a.bg[0];
    a.out_color[plane + pix_id] = C1 + Tout * a.bg[1];
    a.out_color[2 * plane + pix_id] = C2 + Tout * a.bg[2];
  }
}

// ------------------------------------------------------------------------------------- K7
// cr/backward.cu:428-581.  Per pixel the reverse walk is the reference's; what changes is how
// the nine per-(pixel,Gaussian) gradient terms reach memory.  The reference issues nine global
// float atomics per pixel per Gaussian; here
//   1. each wave reduce-scatters the nine terms over its four 16-lane DPP rows (31 VALU ops,
//      gcr_row_reduce_scatter9; no LDS traffic),
//   2. lanes 0..8 of each row add the row sums into a per-chunk LDS accumulator (one ds_add_f32),
//   3. after the chunk, thread t flushes entry t with nine global_atomic_add_f32
// so global atomics drop from 9 per (pixel,Gaussian) to 9 per (tile,Gaussian).
// Only entries [0, max n_contrib of the tile) are visited: later entries are skipped by every
// pixel in the reference too (contributor >= last_contributor, :511-512); and each wave visits
// only the entries whose quadrant mask includes its 8x8 quadrant (see K6).
template <bool FAST_EXP>
__global__ __launch_bounds__(256) void k_blend_bwd(const GcrBlendArgs a) {
  __shared__ StagedEntry sE[CHUNK];
  __shared__ uint2 sList[4][CHUNK + 2];   // per-wave compacted slots: {byte offset into sE, list entry}
  __shared__ uint16_t sAcc4[4][CHUNK + 2];  // ... and byte offset of the slot's column in sAcc
  __shared__ float sAcc[9][CHUNK];
  __shared__ uint32_t sMax[4];

  const int tile = blockIdx.x;
  const int tx = tile % a.gx, ty = tile / a.gx;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int pxi = tx * GCR_TILE_X + (w & 1) * 8 + (lane & 7);
  const int pyi = ty * GCR_TILE_Y + (w >> 1) * 8 + (lane >> 3);
  const bool inside = pxi < a.W && pyi < a.H;
  const float pixx = (float)pxi, pixy = (float)pyi;
  const float tile_x0 = (float)(tx * GCR_TILE_X), tile_y0 = (float)(ty * GCR_TILE_Y);
  const uint32_t r0 = a.ranges[2 * tile];
  const size_t pix_id = (size_t)a.W * pyi + pxi;
  const size_t plane = (size_t)a.H * a.W;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const int acc_slot = (lane & 15) <= 8 ? (lane & 15) : -1;  // which of the 9 terms this lane flushes

  const float T_final = inside ? a.final_T[pix_id] : 0.0f;
  const uint32_t last_contributor = inside ? a.n_contrib[pix_id] : 0u;
  float dLp0 = 0.0f, dLp1 = 0.0f, dLp2 = 0.0f;
  if (inside) {
    dLp0 = a.dL_dpix[pix_id];
    dLp1 = a.dL_dpix[plane + pix_id];
    dLp2 = a.dL_dpix[2 * plane + pix_id];
  }
  const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
  float bg_dot_dpixel = 0;
  bg_dot_dpixel += bg0 * dLp0;
  bg_dot_dpixel += bg1 * dLp1;
  bg_dot_dpixel += bg2 * dLp2;
  const float ddelx_dx = (float)(0.5 * a.W), ddely_dy = (float)(0.5 * a.H);

  // entries any pixel of this tile consumed
  const uint32_t wave_max = gcr_wave_max_u32(last_contributor);
  if (lane == 0) sMax[w] = wave_max;
  __syncthreads();
  const int total = (int)max(max(sMax[0], sMax[1]), max(sMax[2], sMax[3]));
  if (total == 0) return;

  float T = T_final;
  const float neg_T_final = -T_final;
  const char* const sEb = reinterpret_cast<const char*>(sE);
  char* const acc_base = reinterpret_cast<char*>(&sAcc[acc_slot >= 0 ? acc_slot : 0][0]);
  float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;  // accum_rec
  float last_alpha = 0.0f, lc0 = 0.0f, lc1 = 0.0f, lc2 = 0.0f;

  for (int base = 0; base < total; base += CHUNK) {
    const int n = min(CHUNK, total - base);
    __syncthreads();  // previous chunk fully flushed before its LDS is reused
    uint32_t my_mask = 0;
    if (tid < n) {
      // back to front: chunk slot `tid` holds list entry e = total-1-(base+tid)
      const uint32_t id = a.list[r0 + (uint32_t)(total - 1 - (base + tid))];
      const float4* __restrict__ rec = a.rec + (size_t)id * GCR_REC_QUADS;
      const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
      const float pmin = gcr_alpha_skip_bound(q1.y);
      sE[tid].a = q0;
      sE[tid].b = q1;
      sE[tid].c = make_float2(q2.x, pmin);
      sE[tid].id = id;
      my_mask = gcr_quadrant_mask(q0.x, q0.y, q0.z, q0.w, q1.x, pmin, tile_x0, tile_y0);
    }
    sE[tid].mask = my_mask;
#pragma unroll
    for (int k = 0; k < 9; k++) sAcc[k][tid] = 0.0f;
    __syncthreads();

    // compact the slots this wave still consumes: quadrant bit set and list entry < wave_max
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int jj = k * 64 + lane;
      const uint32_t entry_l = (uint32_t)(total - 1 - (base + jj));  // == `contributor` upstream
      const bool rel = jj < n && ((sE[jj].mask >> w) & 1u) && entry_l < wave_max;
      const uint64_t m = __ballot(rel);
      if (rel) {
        const int slot = cnt + __popcll(m & lt_mask);
        sList[w][slot] = make_uint2((uint32_t)(jj * (int)sizeof(StagedEntry)), entry_l);
        sAcc4[w][slot] = (uint16_t)(jj * 4);
      }
      cnt += __popcll(m);
    }
    // two pad slots so the pipeline below may read past the end (the values are never consumed)
    if (lane < 2) {
      sList[w][cnt + lane] = make_uint2(0u, 0u);
      sAcc4[w][cnt + lane] = 0;
    }
    __builtin_amdgcn_wave_barrier();

// One list entry against this lane's pixel (cr/backward.cu:505-580).  Branch-free body: lanes
// that upstream would `continue` keep their state via selects and contribute exact zeros to the
// wave reduction (see K6: scalar-unit pressure).  `power` is forced to 0 on those lanes so that
// every intermediate stays finite and the three masked factors (dchannel_dcolor, dL_dalpha) zero
// all nine terms.  The two quotients share the divisor 1-alpha: one v_rcp_f32 + one Newton step
// (<= 1 ulp) instead of two IEEE division expansions -- the only place the HIP path leaves
// gcr-fp32-v1; K7's sums are order-dependent (atomics) and tolerance-checked anyway.
#define GCR_BWD_STEP(QA, QB, QC, ENTRY, ACC4)                                                  \
  {                                                                                            \
    const float dx = QA.x - pixx, dy = QA.y - pixy;                                            \
    const float power_raw = gcr_power(QA.z, QA.w, QB.x, dx, dy);                               \
    const bool in_range = (ENTRY) < last_contributor && !(power_raw > 0.0f) && !(power_raw < QC.y); \
    if (__ballot(in_range) != 0ull) { /* else the whole wave skips this Gaussian */            \
      const float power = in_range ? power_raw : 0.0f;                                         \
      const float G = blend_exp<FAST_EXP>(power);                                              \
      const float alpha = __builtin_fminf(0.99f, QB.y * G);                                    \
      const bool use = in_range && !(alpha < 1.0f / 255.0f);                                   \
      const float om = 1.f - alpha;                                                            \
      const float r0 = __builtin_amdgcn_rcpf(om);                                              \
      const float rcp = __builtin_fmaf(r0, __builtin_fmaf(-om, r0, 1.0f), r0);                 \
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
      /* reduce-scatter over each 16-lane row, then ONE ds_add_f32: lanes 0..8 of the four */  \
      /* rows add their row's sum of term (lane & 15) into the chunk accumulator */            \
      const float rsum = gcr_row_reduce_scatter9(v, lane);                                     \
      if (acc_slot >= 0) atomicAdd(reinterpret_cast<float*>(acc_base + (ACC4)), rsum);         \
    }                                                                                          \
  }
#define GCR_BWD_LOAD(QA, QB, QC, OFF)                              \
  QA = *reinterpret_cast<const float4*>(sEb + (OFF));             \
  QB = *reinterpret_cast<const float4*>(sEb + (OFF) + 16);         \
  QC = *reinterpret_cast<const float2*>(sEb + (OFF) + 32);

    // software pipeline, two entries per trip (see K6)
    const uint2* lp = &sList[w][0];
    const uint16_t* ap = &sAcc4[w][0];
    float4 qa0, qb0, qa1, qb1;
    float2 qc0, qc1;
    uint2 e0 = lp[0], e1 = lp[1];
    uint32_t c0 = ap[0], c1 = ap[1];
    GCR_BWD_LOAD(qa0, qb0, qc0, e0.x)
    for (int i = 0; i < cnt; i += 2, lp += 2, ap += 2) {
      GCR_BWD_LOAD(qa1, qb1, qc1, e1.x)
      const uint32_t en0 = e0.y, ac0 = c0;
      e0 = lp[2];
      c0 = ap[2];
      GCR_BWD_STEP(qa0, qb0, qc0, en0, ac0)
      if (i + 1 >= cnt) break;
      GCR_BWD_LOAD(qa0, qb0, qc0, e0.x)
      const uint32_t en1 = e1.y, ac1 = c1;
      e1 = lp[3];
      c1 = ap[3];
      GCR_BWD_STEP(qa1, qb1, qc1, en1, ac1)
    }
#undef GCR_BWD_STEP
#undef GCR_BWD_LOAD
    __syncthreads();
    if (tid < n) {
      const uint32_t id = sE[tid].id;
      float g[9];
#pragma unroll
      for (int k = 0; k < 9; k++) g[k] = sAcc[k][tid];
      if (g[0] != 0.0f) atomicAdd(&a.dL_dcolor[3 * (size_t)id + 0], g[0]);
      if (g[1] != 0.0f) atomicAdd(&a.dL_dcolor[3 * (size_t)id + 1], g[1]);
      if (g[2] != 0.0f) atomicAdd(&a.dL_dcolor[3 * (size_t)id + 2], g[2]);
      if (g[3] != 0.0f) atomicAdd(&a.dL_dmean2D[3 * (size_t)id + 0], g[3]);
      if (g[4] != 0.0f) atomicAdd(&a.dL_dmean2D[3 * (size_t)id + 1], g[4]);
      if (g[5] != 0.0f) atomicAdd(&a.dL_dconic[4 * (size_t)id + 0], g[5]);
      if (g[6] != 0.0f) atomicAdd(&a.dL_dconic[4 * (size_t)id + 1], g[6]);
      if (g[7] != 0.0f) atomicAdd(&a.dL_dconic[4 * (size_t)id + 3], g[7]);
      if (g[8] != 0.0f) atomicAdd(&a.dL_dopacity[id], g[8]);
    }
  }
}

}  // namespace

hipError_t gcr_launch_blend_fwd(const GcrBlendArgs& a, bool fast_exp, hipStream_t s) {
  const int T = a.gx * a.gy;
  if (T <= 0) return hipSuccess;
  if (fast_exp)
    k_blend_fwd<true><<<T, 256, 0, s>>>(a);
  else
    k_blend_fwd<false><<<T, 256, 0, s>>>(a);
  return hipGetLastError();
}

hipError_t gcr_launch_blend_bwd(const GcrBlendArgs& a, bool fast_exp, hipStream_t s) {
  const int T = a.gx * a.gy;
  if (T <= 0) return hipSuccess;
  if (fast_exp)
    k_blend_bwd<true><<<T, 256, 0, s>>>(a);
  else
    k_blend_bwd<false><<<T, 256, 0, s>>>(a);
  return hipGetLastError();
}
