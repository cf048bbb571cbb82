// A4: adaptive selection + deterministic compaction (select_kernel / scan_blocks_kernel / expand_kernel), the dense-mode
// expansion and the sampling-network debug view.
// Device code only (gfx950, wave64); part of kernels.hip.hpp.
#pragma once
#include "k_common.hip.hpp"
#include "tuning.hpp"
#include "k_select_pair.hip.hpp"

namespace adanerf {

// ------------------------------------------------------------------------------------------
// A4: adaptive selection + deterministic compaction
// ------------------------------------------------------------------------------------------

// Rays per workgroup of select_kernel (4 waves x kSelRaysPerBlock/4 rays, one ray at a time per wave) = rays per
// entry of the block-total scan.  Small on purpose: a wave's serial loop over its rays is the critical path of a
// small batch (an 83 200-ray shard of an 8-GPU frame), and more, shorter waves also schedule better on a whole
// frame (measured 0.207 ms at 256 rays, 0.167 ms at 64 for 640 000 rays).
using tune::kSelRaysPerBlock;
static_assert(kSelRaysPerBlock == 32 || kSelRaysPerBlock == 64, "segment must fit one wave");
constexpr int kSelSegShift = kSelRaysPerBlock == 64 ? 6 : 5;

// Selection rule (src/nerf_raymarch_common.py:699-757 as a set rule):
// keep the n_max largest values (ties: lower bin first) that are >= thr; if none is >= thr keep the
// arg-max alone.  One wave per ray: lane holds bins (lane, lane + 64); the kept set lives in two
// 64-bit ballot masks, so ascending-bin output order is a popcount.
__device__ __forceinline__ void select_ray(float v0, float v1, int lane, int n_max, float thr, uint64_t* s0, uint64_t* s1) {
  const uint64_t b0 = __ballot(v0 >= thr), b1 = __ballot(v1 >= thr);
  const int c = __popcll(b0) + __popcll(b1);
  uint64_t sel0, sel1;
  if (c <= n_max && c > 0) {
    sel0 = b0;
    sel1 = b1;
  } else {
    const float m = wave_max_f32(fmaxf(v0, v1));
    const uint64_t e0 = __ballot(v0 == m), e1 = __ballot(v1 == m);
    if (c == 0) {
      // nothing clears the threshold: keep the arg-max alone (lowest bin among equal maxima)
      sel0 = e0 & (~e0 + 1);
      sel1 = e0 ? 0 : (e1 & (~e1 + 1));
      if ((sel0 | sel1) == 0) sel0 = 1;   // all-NaN row: keep bin 0 (undefined in the reference)
    } else {
      // More than n_max candidates: bisect a value threshold t in [thr, max] until exactly n_max values
      // are >= t (v_cmp yields the lane mask directly, ~10 instructions per step, ~log2(range / gap)
      // steps).  If the interval closes on a tie that straddles the cut-off, keep everything above the
      // tie value plus the lowest-index members of the tie (the set rule's "lower bin first").
      float lo = thr, hi = m;                    // count(v >= lo) = c > n_max
      uint64_t g0 = e0, g1 = e1;                 // {v >= hi}
      int ch = __popcll(e0) + __popcll(e1);
      uint64_t t0 = b0, t1 = b1;                 // {v >= lo}
      while (ch < n_max) {
        const float mid = lo + (hi - lo) * 0.5f;
        if (!(mid > lo) || !(mid < hi)) break;   // lo and hi are adjacent floats
        const uint64_t m0 = __ballot(v0 >= mid), m1 = __ballot(v1 >= mid);
        const int cm = __popcll(m0) + __popcll(m1);
        if (cm > n_max) {
          lo = mid;
          t0 = m0;
          t1 = m1;
        } else {
          hi = mid;
          g0 = m0;
          g1 = m1;
          ch = cm;
        }
      }
      if (ch >= n_max) {
        // ch == n_max: {v >= hi} is the answer; ch > n_max only when more than n_max values equal the
        // maximum (then lo..hi never moved): fall through to the tie rule with an empty "above" set
        if (ch == n_max) {
          sel0 = g0;
          sel1 = g1;
        } else {
          g0 = 0;
          g1 = 0;
          ch = 0;
          t0 = e0;
          t1 = e1;
          goto tie;
        }
      } else {
      tie:
        // every value in {v >= lo} \ {v >= hi} equals lo: take the first (n_max - ch) of them by bin index
        const uint64_t q0 = t0 & ~g0, q1 = t1 & ~g1;
        const int need = n_max - ch;
        const int r0 = mbcnt64(q0), r1 = __popcll(q0) + mbcnt64(q1);
        const uint64_t k0 = __ballot(((q0 >> lane) & 1) && r0 < need), k1 = __ballot(((q1 >> lane) & 1) && r1 < need);
        sel0 = g0 | k0;
        sel1 = g1 | k1;
      }
    }
  }
  *s0 = sel0;
  *s1 = sel1;
}

// wave-wide sum (every lane gets it)
__device__ __forceinline__ float wave_sum_all_f32(float v) { return wave_sum_dpp_f32(v); }

// the sampler's transform of a ray's 128 raw outputs held two per lane (kOracle*, k_select_pair.hip.hpp)
__device__ __forceinline__ void oracle_transform_wave(int transform, float* v0, float* v1) {
  if (transform == kOracleSigmoid) {
    *v0 = 1.0f / (1.0f + expf(-*v0));
    *v1 = 1.0f / (1.0f + expf(-*v1));
  } else if (transform == kOracleSoftmax) {
    const float m = wave_max_f32(fmaxf(*v0, *v1));
    const float e0 = expf(*v0 - m), e1 = expf(*v1 - m);
    const float s = wave_sum_all_f32(e0 + e1);
    *v0 = e0 / s;
    *v1 = e1 / s;
  }
}

__global__ __launch_bounds__(256) void select_kernel(const float* __restrict__ oracle, int n_rays, int n_max, float thr, int transform,
                                                     int32_t* __restrict__ counts, uint8_t* __restrict__ selbin,
                                                     float* __restrict__ selw, int32_t* __restrict__ block_total) {
  __shared__ int wave_tot[4];
  constexpr int RPW = kSelRaysPerBlock / 4;   // rays per wave
  const int lane = lane_id();
  const int wave = static_cast<int>(threadIdx.x) >> 6;
  const int base = blockIdx.x * kSelRaysPerBlock + wave * RPW;
  int total = 0;
  // rows of the next group of 4 rays are requested before the current group is processed, so each wave keeps
  // 8 row loads in flight while it computes (the kernel is bound by latency x bytes in flight, not by issue)
  float n0[4], n1[4];
  auto fetch = [&](int i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = base + i + u;
      const float* row = oracle + static_cast<size_t>(r < n_rays ? r : 0) * kBins;
      n0[u] = row[lane];
      n1[u] = row[64 + lane];
    }
  };
  fetch(0);
#pragma unroll
  for (int i = 0; i < RPW; i += 4) {
    float v0[4], v1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v0[u] = n0[u];
      v1[u] = n1[u];
    }
    if (i + 4 < RPW) fetch(i + 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = base + i + u;
      if (r >= n_rays) break;     // wave-uniform
      uint64_t s0, s1;
      oracle_transform_wave(transform, &v0[u], &v1[u]);
      select_ray(v0[u], v1[u], lane, n_max, thr, &s0, &s1);
      const int c0 = __popcll(s0);
      const int cnt = c0 + __popcll(s1);
      const size_t o = static_cast<size_t>(r) * n_max;
      if ((s0 >> lane) & 1) {
        const int rank = mbcnt64(s0);
        selbin[o + rank] = static_cast<uint8_t>(lane);
        selw[o + rank] = v0[u];
      }
      if ((s1 >> lane) & 1) {
        const int rank = c0 + mbcnt64(s1);
        selbin[o + rank] = static_cast<uint8_t>(64 + lane);
        selw[o + rank] = v1[u];
      }
      if (lane == 0) counts[r] = cnt;
      total += cnt;
    }
  }
  if (lane == 0) wave_tot[wave] = total;
  __syncthreads();
  if (threadIdx.x == 0) block_total[blockIdx.x] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
}

// Lane-pair selection (k_select_pair.hip.hpp), standalone form: rows of [R,128] fp32 in global memory (stage API / parity tests).  4 waves x 32 rays per workgroup.
// n_list != null: the refinement pass of the guarded selection -- rows so.refine_list[0 .. *n_list) only (n_rays bounds the launch).
__global__ __launch_bounds__(256) void select_rows_kernel(const float* __restrict__ oracle, int n_rays, SelectOut so,
                                                          const int32_t* __restrict__ n_list = nullptr) {
  __shared__ __attribute__((aligned(16))) char lds[4 * kPairLdsBytesPerWave];
  const int lane = lane_id();
  const int wave = static_cast<int>(threadIdx.x) >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int first = (blockIdx.x * 4 + wave) * 32;
  if (n_list) n_rays = min(n_rays, *n_list);
  if (first >= n_rays) return;                       // wave-uniform
  const int local = first + j;
  const bool valid = local < n_rays;
  const int lidx = valid ? local : n_rays - 1;
  const int entry = n_list ? so.refine_list[lidx] : 0;
  const float* row = oracle + static_cast<size_t>(n_list ? (entry & kRefineRayMask) : lidx) * kBins;
  float x[64];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v = *reinterpret_cast<const float4*>(row + 32 * m + 8 * g + 4 * h);
      x[16 * m + 4 * g + 0] = v.x;
      x[16 * m + 4 * g + 1] = v.y;
      x[16 * m + 4 * g + 2] = v.z;
      x[16 * m + 4 * g + 3] = v.w;
    }
  const uint32_t stage = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds)) + wave * kPairLdsBytesPerWave + lane * 16;
  bool bad = false;
  if (so.guard_mask) {      // guard mode: a non-finite row is undecided by definition (the sampling kernels do the same)
    float z = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) z = __builtin_fmaf(x[i], 0.f, z);
    bad = (z != z) | (pair_xchg(static_cast<uint32_t>(z != z)) != 0u);
  }
  GuardAcc gacc;
  pair_epilogue(x, lane, local, valid, stage, so, bad, entry, &gacc);
  if (n_list) guard_flush(so, gacc, lane);
}

// max |a - b| over n floats -> *out (float bits of a non-negative value, atomicMax); a non-finite difference sets out[1]
__global__ __launch_bounds__(256) void max_abs_diff_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, uint32_t* __restrict__ out) {
  float m = 0.f;
  bool bad = false;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const float d = fabsf(a[i] - b[i]);
    if (d == d && d < INFINITY) m = fmaxf(m, d);
    else bad = true;
  }
  m = wave_max_nonneg(m);
  const uint64_t anybad = __ballot(bad);
  if ((threadIdx.x & 63) == 0) {
    if (m > 0.f) atomicMax(&out[0], __builtin_bit_cast(uint32_t, m));
    if (anybad) atomicAdd(&out[1], 1u);
  }
}

// Calibration of the guarded selection's two bounds on [n_rays,128] raw outputs of the plain-fp16 engine (y) and the split engine (x);
// the statistics pair_monitor measures on every frame, over whole rows.  One wave per ray (lane: bins lane and lane + 64).
//   out[0]  max |y - x|                                     (float bits, atomicMax)
//   out[1]  rays with a non-finite difference               (atomicAdd)
//   out[2]  two_eps > 0 only: max over rays of (y_i - x_i) - (y_j - x_j), i kept by the selection from y (n_max largest that reach thr,
//           or the arg-max), j a candidate that is not: max(v_n - two_eps, thr - two_eps / 2) <= y_j < cut value (arg-max fallback:
//           v_1 - two_eps).  The sign that could make j overtake i; see pair_select.
__global__ __launch_bounds__(256) void guard_stats_kernel(const float* __restrict__ y, const float* __restrict__ x, int n_rays, int n_max, float thr,
                                                          float two_eps, uint32_t* __restrict__ out) {
  const int lane = lane_id();
  const int r = blockIdx.x * 4 + (static_cast<int>(threadIdx.x) >> 6);
  if (r >= n_rays) return;   // wave-uniform
  const float* yr = y + static_cast<size_t>(r) * kBins;
  const float* xr = x + static_cast<size_t>(r) * kBins;
  const float y0 = yr[lane], y1 = yr[64 + lane];
  const float d0 = y0 - xr[lane], d1 = y1 - xr[64 + lane];
  const bool f0 = fabsf(d0) < INFINITY, f1 = fabsf(d1) < INFINITY;
  const float ma = wave_max_f32(fmaxf(f0 ? fabsf(d0) : 0.f, f1 ? fabsf(d1) : 0.f));
  const uint64_t bad = __ballot(!f0 || !f1);
  float mp = 0.f;
  if (two_eps > 0.f && !bad) {
    float a0 = y0, a1 = y1, c0 = 0.f, tn = 0.f;
    for (int k = 0; k < n_max; ++k) {      // the n_max largest values of the row, one per round
      const float m = wave_max_f32(fmaxf(a0, a1));
      if (k == 0) c0 = m;
      tn = m;
      const uint64_t e0 = __ballot(a0 == m), e1 = __ballot(a1 == m);
      if (e0) {
        if (lane == __builtin_ctzll(e0)) a0 = -INFINITY;
      } else if (lane == __builtin_ctzll(e1)) a1 = -INFINITY;
    }
    const bool none = c0 < thr;
    const float t = none ? c0 : fmaxf(tn, thr);
    const float cut = none ? c0 - two_eps : fmaxf(tn - two_eps, thr - 0.5f * two_eps);
    const float dk = wave_max_f32(fmaxf(y0 >= t ? d0 : -INFINITY, y1 >= t ? d1 : -INFINITY));
    const float dn = -wave_max_f32(fmaxf((y0 >= cut && y0 < t) ? -d0 : -INFINITY, (y1 >= cut && y1 < t) ? -d1 : -INFINITY));
    const float pr = dk - dn;
    mp = (pr > 0.f && pr < INFINITY) ? pr : 0.f;
  }
  if (lane == 0) {
    if (ma > 0.f) atomicMax(&out[0], __builtin_bit_cast(uint32_t, ma));
    if (bad) atomicAdd(&out[1], 1u);
    if (mp > 0.f) atomicMax(&out[2], __builtin_bit_cast(uint32_t, mp));
  }
}

// exclusive scan of the per-block totals by one workgroup; writes S to *total
__global__ __launch_bounds__(1024) void scan_blocks_kernel(const int32_t* __restrict__ block_total, int n_blocks,
                                                           int32_t* __restrict__ block_offset, int32_t* __restrict__ total) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const int per = (n_blocks + 1023) / 1024;
  const int lo = t * per;
  int s = 0;
  for (int i = 0; i < per; ++i) {
    const int k = lo + i;
    if (k < n_blocks) s += block_total[k];
  }
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;   // exclusive prefix of this thread's chunk
  for (int i = 0; i < per; ++i) {
    const int k = lo + i;
    if (k < n_blocks) {
      block_offset[k] = run;
      run += block_total[k];
    }
  }
  if (t == 1023) *total = part[1023];
}

// ray offsets + compacted (key, weight) arrays, ray-major / bins ascending
// One thread per ray; the rays behind one entry of the segment totals (2^seg_shift rays: 64 for select_kernel's workgroups,
// 32 for the wave-sized segments of the fused / pair selection, k_select_pair.hip.hpp) are part of one wave, so the in-segment
// prefix is a width-limited shuffle scan (no LDS, no barrier).  INLINE_SCAN: the workgroup also derives its own base offset by
// summing the segment totals in front of it (a cooperative reduction over <= kInlineScanMaxBlocks ints from L2) instead of
// reading the output of scan_blocks_kernel -- one launch and its ~10 us of single-workgroup latency less per batch; the host
// keeps the separate scan for larger batches, where every workgroup re-reading the totals would be quadratic.
template <bool INLINE_SCAN>
__global__ __launch_bounds__(256) void expand_kernel(const int32_t* __restrict__ counts, const uint8_t* __restrict__ selbin,
                                                     const float* __restrict__ selw, const int32_t* __restrict__ block_offset,
                                                     const int32_t* __restrict__ block_total, int n_blocks, int n_rays, int n_max,
                                                     int seg_shift, int32_t* __restrict__ ray_offsets, uint32_t* __restrict__ sample_key,
                                                     float* __restrict__ sample_w, int32_t* __restrict__ total) {
  const int segs = 256 >> seg_shift;                           // segments covered by this workgroup
  const int t = static_cast<int>(threadIdx.x);
  const int r = blockIdx.x * 256 + t;
  const int c = (r < n_rays) ? counts[r] : 0;
  const int x = seg_shift == 5 ? wave_incl_sum_dpp_i32<32>(c) : wave_incl_sum_dpp_i32<64>(c);      // inclusive prefix inside the segment (DPP, no LDS crossbar)
  const int seg = r >> seg_shift;                              // segment of this ray
  int seg_base;
  if (INLINE_SCAN) {
    __shared__ int part[4];
    const int b0 = blockIdx.x * segs;                          // first segment of this workgroup
    int s = 0;
    for (int i = t; i < b0; i += 256) s += block_total[i];
    s = wave_sum_dpp_i32(s);
    if ((t & 63) == 0) part[t >> 6] = s;
    __syncthreads();
    seg_base = part[0] + part[1] + part[2] + part[3];
    for (int i = b0; i < seg && i < n_blocks; ++i) seg_base += block_total[i];
    if (blockIdx.x == gridDim.x - 1 && t == 0) {               // the last workgroup knows the grand total
      int tot = part[0] + part[1] + part[2] + part[3];
      for (int i = b0; i < n_blocks; ++i) tot += block_total[i];
      *total = tot;
    }
  } else {
    seg_base = (seg < n_blocks) ? block_offset[seg] : 0;
  }
  if (r >= n_rays) return;
  const int o = seg_base + x - c;
  ray_offsets[r] = o;
  const size_t src = static_cast<size_t>(r) * n_max;
  for (int k = 0; k < c; ++k) {
    sample_key[o + k] = (static_cast<uint32_t>(r) << 7) | selbin[src + k];
    sample_w[o + k] = selw[src + k];
  }
}

// Guarded selection, between its two passes: the ascending list of the rays to re-evaluate and their number -- the rays whose guard
// bit is set (one word per 32 rays, written by pair_epilogue) and the rays audited in this frame (audit_bits: a rotating 1 / period of all
// rays; an audited ray that pass 1 had decided carries kRefineAuditBit in its entry).  256 words (8 192 rays) per workgroup.  Every
// workgroup reads the whole mask (<= 80 KB from L2 at 800 x 800) for the totals and its own prefix instead of waiting for a scan.
// Deterministic order.
//   cap_round > 0 ("fill" mode, ADANERF_FLAG_GUARD_AUDIT_FILL): the refinement pass runs in rounds of cap_round rays (its grid x 128); the
//   undecided rays fix the number of rounds, and only as many audited-only rays are listed as the last round has room for -- the audit
//   then never costs a round of its own (an 80 000-ray share of an 8-GPU frame: 31 000 undecided rays are one round, 34 000 with the
//   audit were two).  Which ones: a window of the frame's audit candidates, in ray order, that moves on by its own length every time the
//   same audit phase comes round again (`cycle`), so no ray is left out for good.
struct RefinePlan {
  int tot_und, tot_aud, budget, offset;
};
__device__ __forceinline__ int refine_accepted_before(const RefinePlan& p, int pa) {      // accepted audit candidates with rank < pa
  if (p.budget >= p.tot_aud) return pa;
  const int wrap = p.offset + p.budget - p.tot_aud;      // > 0: the window wraps round to the first `wrap` candidates
  if (wrap <= 0) return min(max(pa - p.offset, 0), p.budget);
  return min(pa, wrap) + max(pa - p.offset, 0);
}
__device__ __forceinline__ bool refine_accepts(const RefinePlan& p, int pa) {
  if (p.budget >= p.tot_aud) return true;
  int d = pa - p.offset;
  if (d < 0) d += p.tot_aud;
  return d < p.budget;
}
__global__ __launch_bounds__(256) void refine_list_kernel(const uint32_t* __restrict__ mask, int n_words, int n_rays, int period, int phase,
                                                          int cap_round, int cycle, int32_t* __restrict__ list, int32_t* __restrict__ count) {
  __shared__ int red[4][4], wtot[2][4];
  const int t = static_cast<int>(threadIdx.x);
  const int b0 = blockIdx.x * 256;
  auto word = [&](int i, uint32_t m, uint32_t* und, uint32_t* aud) {      // bits of word i (value m): undecided rays, audited-only rays
    const int left = n_rays - i * 32;                                       // rays this word covers (the last word may be partial)
    const uint32_t in_range = left >= 32 ? 0xFFFFFFFFu : (left > 0 ? low_bits32(left) : 0u);
    const uint32_t u = m & in_range;
    *und = u;
    *aud = audit_bits(period, phase, i) & ~u & in_range;
  };
  int s[4] = {0, 0, 0, 0};      // undecided / audited-only in front of this workgroup, and in the whole batch
  auto tally = [&](int i, uint32_t m) {
    uint32_t u, a;
    word(i, m, &u, &a);
    const int cu = __popc(u), ca = __popc(a);
    if (i < b0) {
      s[0] += cu;
      s[1] += ca;
    }
    s[2] += cu;
    s[3] += ca;
  };
  // every workgroup walks the whole mask (80 KB at 800 x 800, from L2): 16-byte loads, or the walk is 79 dependent-looking
  // 4-byte round trips per thread (27 us instead of 9 for the 800 x 800 frame)
  const int n_quads = n_words >> 2;
  const uint4* __restrict__ mask4 = reinterpret_cast<const uint4*>(mask);
#pragma unroll 2
  for (int q = t; q < n_quads; q += 256) {
    const uint4 v = mask4[q];
    tally(4 * q, v.x);
    tally(4 * q + 1, v.y);
    tally(4 * q + 2, v.z);
    tally(4 * q + 3, v.w);
  }
  for (int i = 4 * n_quads + t; i < n_words; i += 256) tally(i, mask[i]);
#pragma unroll
  for (int k = 0; k < 4; ++k) s[k] = wave_sum_dpp_i32(s[k]);
  const int wi = b0 + t;
  uint32_t und = 0u, aud = 0u;
  if (wi < n_words) word(wi, mask[wi], &und, &aud);
  const int cu = __popc(und), ca = __popc(aud);
  const int xu = wave_incl_sum_dpp_i32<64>(cu), xa = wave_incl_sum_dpp_i32<64>(ca);       // inclusive scans inside the wave
  if ((t & 63) == 63) {
    wtot[0][t >> 6] = xu;
    wtot[1][t >> 6] = xa;
  }
  if ((t & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) red[k][t >> 6] = s[k];
  }
  __syncthreads();
  int tot[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) tot[k] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
  RefinePlan p;
  p.tot_und = tot[2];
  p.tot_aud = tot[3];
  p.budget = p.tot_aud;
  p.offset = 0;
  if (cap_round > 0) {
    const int rounds = max(1, (p.tot_und + cap_round - 1) / cap_round);
    int room = max(0, rounds * cap_round - p.tot_und);
    if (room < (p.tot_aud + 3) / 4) room += cap_round;      // never less than a quarter of the frame's quota: then the audit does take a round
    p.budget = min(p.tot_aud, room);
    if (p.budget > 0 && p.budget < p.tot_aud) p.offset = static_cast<int>((static_cast<long long>(cycle) * p.budget) % p.tot_aud);
  }
  int pu = tot[0] + xu - cu, pa = tot[1] + xa - ca;      // ranks of this word's first undecided / audited-only ray
  for (int w = 0; w < (t >> 6); ++w) {
    pu += wtot[0][w];
    pa += wtot[1][w];
  }
  for (uint32_t r = und | aud; r; r &= r - 1u) {
    const int bit = __builtin_ctz(r);
    const int ray = wi * 32 + bit;
    if ((und >> bit) & 1u) {
      list[pu + refine_accepted_before(p, pa)] = ray;
      ++pu;
    } else {
      if (refine_accepts(p, pa)) list[pu + refine_accepted_before(p, pa)] = ray | kRefineAuditBit;
      ++pa;
    }
  }
  if (blockIdx.x == 0 && t == 0) *count = p.tot_und + p.budget;
}

// Debug view of the sampling network (viewer 'O' key: copyResultSamplingNetwork -> samplesToImage,
// adanerf_real_time_viewer/src/cuda/base_cuda_kernels.cu:487-528): pixel = ((0.5 + bin) / 128) of the three largest
// outputs of the ray, largest first, in R, G, B.  The viewer sorts with a stable block radix sort, so equal values
// rank lower bin first.  One wave per ray, three arg-max rounds.
__global__ __launch_bounds__(256) void oracle_view_kernel(const float* __restrict__ oracle, int n_rays, uchar4* __restrict__ rgba8) {
  const int lane = lane_id();
  const int r = blockIdx.x * 4 + (static_cast<int>(threadIdx.x) >> 6);
  if (r >= n_rays) return;   // wave-uniform
  const float* row = oracle + static_cast<size_t>(r) * kBins;
  float v0 = row[lane], v1 = row[64 + lane];
  int bin[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float m = wave_max_f32(fmaxf(v0, v1));
    const uint64_t e0 = __ballot(v0 == m), e1 = __ballot(v1 == m);
    int b = k;   // all-NaN row: undefined in the reference
    if (e0) b = __builtin_ctzll(e0);
    else if (e1) b = 64 + __builtin_ctzll(e1);
    bin[k] = b;
    if (b == lane) v0 = -INFINITY;
    if (b == 64 + lane) v1 = -INFINITY;
  }
  if (lane == 0) {
    uchar4 px;
    px.x = static_cast<unsigned char>((0.5f + static_cast<float>(bin[0])) / 128.0f * 255.0f);
    px.y = static_cast<unsigned char>((0.5f + static_cast<float>(bin[1])) / 128.0f * 255.0f);
    px.z = static_cast<unsigned char>((0.5f + static_cast<float>(bin[2])) / 128.0f * 255.0f);
    px.w = 255;
    rgba8[r] = px;
  }
}

// thr == 0 inside adanerf_render: only what the compositing kernel needs per ray -- the samples' keys are their indices and their
// kept values are the oracle buffer itself, so neither array is written (the shading / compositing kernels take a null key
// array and the oracle buffer as the weights)
__global__ __launch_bounds__(256) void dense_offsets_kernel(int n_rays, int32_t* __restrict__ ray_offsets, int32_t* __restrict__ counts,
                                                            int32_t* __restrict__ total) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r == 0) *total = n_rays * kBins;
  if (r >= n_rays) return;
  ray_offsets[r] = r * kBins;
  counts[r] = kBins;
}

// thr == 0, stage API (explicit arrays): every bin of every ray (src/nerf_raymarch_common.py:708-720)
__global__ __launch_bounds__(256) void dense_expand_kernel(const float* __restrict__ oracle, int n_rays, int32_t* __restrict__ ray_offsets,
                                                           int32_t* __restrict__ counts, uint32_t* __restrict__ sample_key,
                                                           float* __restrict__ sample_w, int32_t* __restrict__ total) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t n = static_cast<size_t>(n_rays) * kBins;
  if (i == 0) *total = static_cast<int32_t>(n);
  if (i >= n) return;
  sample_key[i] = static_cast<uint32_t>(i);
  sample_w[i] = oracle[i];
  if ((i & (kBins - 1)) == 0) {
    const int r = static_cast<int>(i >> 7);
    ray_offsets[r] = static_cast<int32_t>(i);
    counts[r] = kBins;
  }
}

}  // namespace adanerf
