// gcr_sort.h -- workgroup-level sorting of one tile's (depth << 32 | Gaussian index) keys, shared by the tile sort
// (gcr_binning.hip, K4) and the forward blend (gcr_blend.hip, K6), which extends a lazily sorted list on demand.
// All functions are called by a whole 256-thread workgroup with block-uniform arguments.
#pragma once
#include "gcr_device.h"

// In-LDS bitonic network over s[0, N2) (N2 a power of two >= 2, keys padded with ~0), 256 threads.
// Wave w owns the contiguous span [w*N2/4, (w+1)*N2/4): a stage whose pairs (a, a|j) stay inside a span
// (2j <= span) only needs the wave's own LDS ordering, so block barriers are paid only for the few
// long-stride stages (3 of 55 for N2 = 1024).  Ends with a block barrier.
GCR_DEV void gcr_bitonic_sort_lds(uint64_t* s, int N2, int tid) {
  const int half = N2 >> 1, wave_pairs = half >> 2;  // pairs per wave and stage
  const int lane = tid & 63, w = tid >> 6;
  const int span = N2 >> 2;
  bool prev_local = false;
  for (int k = 2; k <= N2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      const bool local = 2 * j <= span;
      if (!local && prev_local) __syncthreads();  // other waves' spans are about to be read
      for (int t = lane; t < wave_pairs; t += 64) {
        const int i = w * wave_pairs + t;
        const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        const int b = a | j;
        const uint64_t x = s[a], y = s[b];
        const bool ascending = (a & k) == 0;
        if ((x > y) == ascending) {
          s[a] = y;
          s[b] = x;
        }
      }
      if (local)
        __builtin_amdgcn_wave_barrier();
      else
        __syncthreads();
      prev_local = local;
    }
  }
  __syncthreads();
}

// A tile list LONGER than the LDS this launch was given (dense general 3DGS scenes beyond the 4096-key maximum;
// or simply a list that outgrew the caller's length hint -- GaussianCity's own scenes have ~150 entries per
// tile): the same workgroup sorts it in three steps, everything else of the frame is untouched.
//   1. runs of `run` keys (= the LDS capacity of the launch, a power of two) are sorted in LDS with the bitonic
//      network and written back in place;
//   2. log2(#runs) merge passes ping-pong between the key buffer and its spare half: every thread produces
//      a contiguous slice of each merged pair, located with a merge-path binary search (keys are unique);
//   3. the Gaussian indices (low 32 bits) go to the sorted list.
// Round 1 sent the WHOLE frame to the global radix sort when one list was long (cr/rasterizer_impl.cu:252-260 sorts
// everything globally; the order produced is the same).
GCR_DEV void gcr_tile_sort_long(uint64_t* s, uint32_t run0, uint64_t* __restrict__ A, uint64_t* __restrict__ B,
                                uint32_t* __restrict__ out, uint32_t n, int tid) {
  // 1. sorted runs
  for (uint32_t c0 = 0; c0 < n; c0 += run0) {
    const uint32_t len = min(run0, n - c0);
    for (uint32_t i = tid; i < run0; i += 256) s[i] = i < len ? A[c0 + i] : ~0ull;
    __syncthreads();
    gcr_bitonic_sort_lds(s, (int)run0, tid);
    for (uint32_t i = tid; i < len; i += 256) A[c0 + i] = s[i];
    __syncthreads();
  }
  // 2. merge passes.  Every pair of runs (a, b) is merged window by window: a window = `run0` consecutive OUTPUT
  //    keys; merge-path binary searches (one thread per window boundary, all boundaries of the pair at once) say
  //    which slices of a and b produce it; the two slices (run0 keys together) are loaded into LDS with coalesced
  //    reads, every key finds its output position as (own index + lower bound in the other slice) by a binary
  //    search in LDS, and is stored into the window.  Only the boundary searches chase pointers through global
  //    memory (the first version merged element by element from global loads: dense scene D1 1.39 -> 1.27 ms; what
  //    remains is the 78-stage bitonic network of the 4096-key runs, LDS-bandwidth-bound with five tiles per CU --
  //    an LDS radix sort of the runs was measured too: no faster (8 ballots + selects per key and pass), dropped).
  uint64_t* src = A;
  uint64_t* dst = B;
  __shared__ uint32_t win_ia[258];  // a-index of every window boundary of the current pair, 256 windows at a time
  for (uint32_t run = run0; run < n; run <<= 1) {
    for (uint32_t p0 = 0; p0 < n; p0 += 2 * run) {
      const uint32_t la = min(run, n - p0);
      const uint32_t lb = p0 + run < n ? min(run, n - p0 - run) : 0u;
      const uint64_t* __restrict__ a = src + p0;
      const uint64_t* __restrict__ b = src + p0 + run;
      uint64_t* __restrict__ o = dst + p0;
      const uint32_t tot = la + lb;
      if (lb == 0u) {  // an unpaired run at the end: copy
        for (uint32_t i = tid; i < la; i += 256) o[i] = a[i];
        continue;
      }
      const uint32_t nwin = (tot + run0 - 1) / run0;
      for (uint32_t w0 = 0; w0 < nwin; w0 += 256) {
        const uint32_t wn = min(256u, nwin - w0);
        __syncthreads();  // win_ia / s free again
        for (uint32_t q = tid; q <= wn; q += 256) {
          // boundary d = first output index of window w0+q (or `tot`): i keys of a and d-i of b precede it, with
          // i the smallest index such that a[i] > b[d-1-i] (unique keys: no ties)
          const uint32_t d = min(tot, (w0 + q) * run0);
          uint32_t lo = d > lb ? d - lb : 0u, hi = min(d, la);
          while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (a[mid] < b[d - 1 - mid])
              lo = mid + 1;
            else
              hi = mid;
          }
          win_ia[q] = lo;
        }
        __syncthreads();
        for (uint32_t w = 0; w < wn; w++) {
          const uint32_t d0 = (w0 + w) * run0, d1 = min(tot, d0 + run0);
          const uint32_t ia0 = win_ia[w], ia1 = win_ia[w + 1];
          const uint32_t ib0 = d0 - ia0, ib1 = d1 - ia1;
          const uint32_t na = ia1 - ia0, nb = ib1 - ib0;  // na + nb = d1 - d0 <= run0
          for (uint32_t i = tid; i < na; i += 256) s[i] = a[ia0 + i];
          for (uint32_t i = tid; i < nb; i += 256) s[na + i] = b[ib0 + i];
          __syncthreads();
          for (uint32_t i = tid; i < na + nb; i += 256) {
            const uint64_t key = s[i];
            const bool from_a = i < na;
            const uint64_t* other = from_a ? s + na : s;
            uint32_t lo = 0, hi = from_a ? nb : na;  // lower bound of key in the other slice
            while (lo < hi) {
              const uint32_t mid = (lo + hi) >> 1;
              if (other[mid] < key)
                lo = mid + 1;
              else
                hi = mid;
            }
            o[d0 + (from_a ? i : i - na) + lo] = key;
          }
          __syncthreads();
        }
      }
    }
    __syncthreads();  // the pass is complete (workgroup-scope visibility of the global stores) before it is read
    uint64_t* t = src;
    src = dst;
    dst = t;
  }
  // 3. Gaussian indices
  for (uint32_t i = tid; i < n; i += 256) out[i] = (uint32_t)src[i];
}

// ---------------------------------------------------------------------------------------------- lazy tile sort
// cr/rasterizer_impl.cu:252-260 sorts EVERY instance; the blend (cr/forward.cu:284-286) stops reading a tile's list
// when its 256 pixels are saturated -- in a saturating scene most of what was sorted is never read (dense stress
// scene D1: 85 % of 33.7 M instances; C5: 64 %).  The binning state is opaque to the caller, so the order of entries
// nobody reads is not observable: a list longer than GCR_LAZY_MIN is sorted SEGMENT BY SEGMENT, front to back, and only
// as far as the forward blend actually walks.
//   state of a tile (uint4 in the image buffer): x = n_sorted -- list[0, n_sorted) is the final order;
//                                                (z, w) = L -- every key >= L is not in that prefix yet.
//   one extension: choose an upper bound U from 256 strided samples of the keys (the candidate samples >= L are
//   rank-sorted in LDS; sample j leaves ~rem * j / m keys below it), compact the keys in [L, U) into LDS with ballots
//   and one LDS atomic per wave (any order: they are sorted next), bitonic sort, append the Gaussian indices.  A
//   segment that does not fit GCR_LAZY_CAP keys is retried with a lower sample; when the samples cannot separate the
//   keys any more (hundreds of thousands of entries in one tile, or equal depths) the interval is bisected between the
//   smallest and the largest key seen -- keys are unique, so every bisection keeps at least one key and loses at least
//   one: it terminates, any list is sorted correctly.
// The key buffer is only read; the tile sort kernel extracts the first segment, K6 the following ones (its workgroup is
// about to gather those entries anyway).  Entries behind the last one consumed stay unsorted.
constexpr int GCR_LAZY_MIN = 1024;      // lists up to this length are sorted whole by the tile sort kernel
constexpr int GCR_LAZY_SAMPLES = 256;
// keys per segment (LDS capacity; a sampled segment aims at 3/4 of it): what K6 can hold in its record buffer, and the
// first segment the tile sort kernel extracts
constexpr int GCR_LAZY_CAP_K6 = 1024;
#ifndef GCR_LAZY_CAP_K4
#define GCR_LAZY_CAP_K4 1024
#endif
constexpr int gcr_lazy_lds_keys(int cap) { return cap + GCR_LAZY_SAMPLES + 3; }  // 64-bit LDS words gcr_lazy_extend() needs

// Appends the next segment of the tile's order to out[n_sorted ...].  `s`: gcr_lazy_lds_keys(CAP) 64-bit LDS words nobody
// else is using (the function starts and ends with a workgroup barrier); keys[0, n): the tile's unsorted keys;
// n_sorted < n and L as above, both updated.
template <int GCR_LAZY_CAP>
GCR_DEV void gcr_lazy_extend(uint64_t* s, const uint64_t* keys, uint32_t n, uint32_t& n_sorted, uint64_t& L,
                             uint32_t* out, int tid) {
  constexpr int GCR_LAZY_TARGET = GCR_LAZY_CAP * 3 / 4;
  uint64_t* samp = s + GCR_LAZY_CAP;                                                        // c[0, m): candidates
  uint32_t* cnt_w = reinterpret_cast<uint32_t*>(s + GCR_LAZY_CAP + GCR_LAZY_SAMPLES);       // compaction counter
  unsigned long long* lo_w = reinterpret_cast<unsigned long long*>(s + GCR_LAZY_CAP + GCR_LAZY_SAMPLES + 1);
  unsigned long long* hi_w = lo_w + 1;
  const uint32_t rem = n - n_sorted;
  const int lane = tid & 63;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  uint32_t m = 0, j = 0;
  uint64_t U = ~0ull;  // no key is ~0: depth bits of a positive float in the high half
  __syncthreads();
  if (rem > (uint32_t)GCR_LAZY_CAP) {  // (then n > 256: the 256 sample positions are distinct)
    const uint64_t k = keys[((uint64_t)tid * n) >> 8];
    const bool cand = k >= L;
    const uint64_t mine = cand ? k : ~0ull;
    s[tid] = mine;
    m = (uint32_t)__syncthreads_count(cand);
    const ulonglong2* s2 = reinterpret_cast<const ulonglong2*>(s);
    uint32_t rank = 0;
#pragma unroll 2
    for (int q = 0; q < GCR_LAZY_SAMPLES / 2; q++) {  // wave-uniform (broadcast) reads
      const ulonglong2 kk = s2[q];
      rank += (kk.x < mine ? 1u : 0u) + (kk.y < mine ? 1u : 0u);
    }
    if (cand) samp[rank] = mine;
    j = (uint32_t)(((uint64_t)GCR_LAZY_TARGET * m) / rem);
    j = max(1u, min(j, m));
    __syncthreads();  // samples ranked; s[0, 256) is free for the compaction
    if (m > 0u && j < m) U = samp[j];
  }
  uint32_t cnt = 0;
  for (int guard = 0; guard < 160; guard++) {
    if (tid == 0) {
      *cnt_w = 0u;
      *lo_w = ~0ull;
      *hi_w = 0ull;
    }
    __syncthreads();
    uint64_t lo_k = ~0ull, hi_k = 0ull;
#pragma unroll 1
    for (uint32_t i0 = 0; i0 < n; i0 += 256u) {
      const uint32_t i = i0 + (uint32_t)tid;
      const uint64_t k = i < n ? keys[i] : ~0ull;
      const bool in = k >= L && k < U;  // (~0 is never in range)
      const uint64_t bal = __ballot(in);
      if (bal != 0ull) {  // wave-uniform
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(cnt_w, (uint32_t)__popcll(bal));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (in) {
          const uint32_t pos = base + (uint32_t)__popcll(bal & lt_mask);
          if (pos < (uint32_t)GCR_LAZY_CAP) s[pos] = k;
          lo_k = k < lo_k ? k : lo_k;
          hi_k = k > hi_k ? k : hi_k;
        }
      }
    }
    __syncthreads();
    cnt = *cnt_w;
    if (cnt <= (uint32_t)GCR_LAZY_CAP) break;
    // too many keys below U: a lower sample while the samples can still tell them apart, else bisect the key interval
    if (j > 1u) {
      j = max(1u, min(j - 1u, (uint32_t)(((uint64_t)j * GCR_LAZY_TARGET) / cnt)));
      U = samp[j];
    } else {
      if (lo_k <= hi_k) {
        atomicMin(lo_w, (unsigned long long)lo_k);
        atomicMax(hi_w, (unsigned long long)hi_k);
      }
      __syncthreads();
      const uint64_t lo = *lo_w, hi = *hi_w;
      U = lo + ((hi - lo) >> 1) + 1ull;  // keeps lo, drops hi (lo < hi: more than one unique key is in range)
      j = 0u;
    }
    __syncthreads();  // everybody has read the counters before they are reset
  }
  if (cnt == 0u || cnt > (uint32_t)GCR_LAZY_CAP) {  // cannot happen (see above); never loop forever on a broken invariant
    n_sorted = n;
    L = ~0ull;
    __syncthreads();
    return;
  }
  int N2 = 256;
  while (N2 < (int)cnt) N2 <<= 1;
  for (int i = (int)cnt + tid; i < N2; i += 256) s[i] = ~0ull;
  __syncthreads();
  gcr_bitonic_sort_lds(s, N2, tid);
  for (uint32_t i = tid; i < cnt; i += 256u) out[n_sorted + i] = (uint32_t)s[i];
  n_sorted += cnt;
  L = U;
  __syncthreads();  // LDS free again; the list entries are visible to the workgroup
}
