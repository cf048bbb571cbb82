// A4 (selection half) on the register layout the sampling MLP leaves its outputs in, so that the adaptive selection runs
// in the epilogue of the kernel that produces the oracle values and the [R,128] fp32 round trip through HBM disappears
// (SURVEY 8d prices the compaction read at "0 if fused after A3").  select_rows_kernel (k_compact.hip.hpp) runs the very same
// device code over [R,128] rows in global memory (stage API adanerf_compact, parity tests).
// Device code only (gfx950, wave64); part of kernels.hip.hpp.
#pragma once
#include "k_common.hip.hpp"
#include "tuning.hpp"

namespace adanerf {

// Where the selection stage leaves its results (the inputs of expand_kernel).
struct SelectOut {
  int32_t* counts;      // [R]            kept samples per ray, 1..n_max
  uint8_t* selbin;      // [R, n_max]     kept bins, ascending
  float* selw;          // [R, n_max]     oracle value of each kept bin
  int32_t* seg_total;   // [ceil(R / 32)] samples kept by each 32-ray segment (one wave's rays)
  int32_t n_max;
  float thr;
  int32_t transform;    // kOracle*: what the sampler applies to the raw network outputs first (losses[0])
  // Guard-banded two-precision selection (ADANERF_SAMPLING_GUARDED, see pair_select):
  uint32_t* guard_mask;        // pass 1 (plain-fp16 engine): [ceil(R / 32)] bit j of word s set <=> ray 32 s + j is undecided
  float guard_eps;             //   guard_band_of(transform, eps): bound on |fp16-engine output - split-engine output| per raw value
  float guard_pair;            //   guard_pair_of(transform, eps, eps_pair): bound on the error of a DIFFERENCE of two candidate values
  int32_t audit_period;        //   > 0 (a power of two <= 32): besides the undecided rays, ray j of 32-ray segment s is re-evaluated when
  int32_t audit_phase;         //   ((j - phase - s) & (period - 1)) == 0 -- a rotating 1 / period of ALL rays, the decided ones included
  const int32_t* refine_list;  // pass 2 (split engine on the listed rays only): `local` indexes this list of ray ids (bit 30 set: the
                               //   ray was DECIDED by pass 1 and is audited: its row is compared, not overwritten); otherwise the ray's
                               //   counts / selbin / selw are overwritten and seg_total corrected by the count difference
  // Monitor of the assumptions behind the band, on every re-evaluated ray: pass 1 leaves the ray's RAW first-pass outputs in
  // guard_rows[ray] and the value from which a bin is a candidate for the kept set in guard_probe[ray]; pass 2 compares the whole row with its own
  // raw outputs and keeps (guard_seen, cumulative): [0] the largest |difference| (float bits), [1] rays where a bound was exceeded,
  // [2] the largest error of a (kept, candidate-not-kept) difference (float bits), [3] audited rays whose pass-1 selection differs from the
  // exact one, [4] audited rays.
  float* guard_probe;          // [R] or null
  float* guard_rows;           // [R,128] or null (pass 1 writes, pass 2 reads)
  uint32_t* guard_seen;        // [5] or null
  float guard_band;            // pass 2: the RAW-unit bound on a value's error pass 1 ran with (guard_eps is 0 there)
  float guard_band_pair;       // pass 2: the RAW-unit bound on a difference's error (0: the sampler transforms its values, no such bound in use)
};

constexpr int32_t kRefineAuditBit = 1 << 30;      // refine_list entry: the ray was decided by pass 1 (audit only)
constexpr int32_t kRefineRayMask = kRefineAuditBit - 1;

// the rays of segment `seg` (32 consecutive rays) that are audited at `phase`: bits j with ((j - phase - seg) & (period - 1)) == 0
__host__ __device__ inline uint32_t audit_bits(int period, int phase, int seg) {
  if (period <= 0) return 0u;
  const uint32_t pattern = period >= 32 ? 1u : 0xFFFFFFFFu / ((1u << period) - 1u);      // bits 0, period, 2 period, ...
  return pattern << ((phase + seg) & (period - 1));
}

// src/nerf_raymarch_common.py:624-630 / 686-690: BCEWithLogitsLoss -> sigmoid, CrossEntropyLoss[Weighted] -> softmax over the bins
constexpr int kOracleRaw = 0, kOracleSigmoid = 1, kOracleSoftmax = 2;

// SelectOut::guard_eps for a bound eps on the raw outputs (host side; see pair_select)
inline float guard_band_of(int transform, float eps) {
  return transform == kOracleSigmoid ? 0.25f * eps : (transform == kOracleSoftmax ? 1.01f * (__builtin_expf(2.0f * eps) - 1.0f) : eps);
}
// SelectOut::guard_pair: the bound on the error of a difference of two values.  Raw outputs: eps_pair as measured (never more than
// 2 eps, which the bound on single values already implies).  A transformed sampler (sigmoid / softmax) scales the two errors of a pair
// by different slopes, so only 2 x the transformed single-value bound holds there.
inline float guard_pair_of(int transform, float eps, float eps_pair) {
  if (transform != kOracleRaw || !(eps_pair > 0.f) || eps_pair > 2.0f * eps) return 2.0f * guard_band_of(transform, eps);
  return eps_pair;
}

constexpr int kPairSegShift = 5;                 // a wave selects for 32 rays = one entry of seg_total
constexpr int kPairMaxN = 16;                    // largest n_max this path handles (sorted lists live in registers)
constexpr int kPairLdsBytesPerWave = 32 * 64 * 4;   // staging of 32 values per lane (two passes)

// The MFMA layout (layout.hpp): lane l = (j = l & 31, h = l >> 5) of a wave holds, for ray j, the 64 values
// x[i], i = 16 m + r, of bins act_feature(i, h) = 32 m + 8 (r >> 2) + 4 h + (r & 3): within every group of 8 consecutive
// bins the first four sit in lane j, the next four in lane j + 32.  All set arithmetic below is per lane on 64-bit masks over i
// plus one exchange with the partner lane.

// the partner lane's value (lane ^ 32): v_permlane32_swap_b32 (gfx950) swaps lanes 32..63 of one register with lanes 0..31 of another --
// applied to two copies of v it leaves v[0..31] | v[0..31] in the first and v[32..63] | v[32..63] in the second; each half picks the
// one that holds its partner.  Stays in the VALU (the __shfl_xor form was a ds_bpermute round trip through the LDS crossbar).
__device__ __forceinline__ uint32_t pair_xchg(uint32_t v) {
  typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
  const u32x2_t r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return (lane_id() < 32) ? r[1] : r[0];
}
__device__ __forceinline__ float pair_xchg(float v) { return __builtin_bit_cast(float, pair_xchg(__builtin_bit_cast(uint32_t, v))); }

// bits [0, n) of a 32-bit word, 0 <= n <= 32
__device__ __forceinline__ uint32_t low_bits32(int n) { return static_cast<uint32_t>((1ull << n) - 1ull); }

// 64-bit mask {x[i] >= t} / {x[i] > t} over the lane's values (NaN compares false, as in the reference's `samples >= threshold`)
template <bool STRICT>
__device__ __forceinline__ void pair_mask(const float* x, float t, uint32_t* lo, uint32_t* hi) {
  uint32_t a = 0, b = 0;
#pragma unroll
  for (int i = 31; i >= 0; --i) {
    a = a + a + ((STRICT ? x[i] > t : x[i] >= t) ? 1u : 0u);
    b = b + b + ((STRICT ? x[32 + i] > t : x[32 + i] >= t) ? 1u : 0u);
  }
  *lo = a;
  *hi = b;
}

// number of PARTNER elements whose bin is below the bins of this lane's group g = i >> 2 (0..15): the partner's groups
// 0..g-1, plus its group g when this lane is the upper half.  q = partner mask pre-shifted by pair_align().
__device__ __forceinline__ void pair_align(uint32_t plo, uint32_t phi, int h, uint32_t* qlo, uint32_t* qhi) {
  // lower half: groups < g <=> partner indices < 4 g <=> (mask << 4) indices < 4 g + 4; upper half: indices < 4 g + 4
  *qlo = h ? plo : (plo << 4);
  *qhi = h ? phi : ((phi << 4) | (plo >> 28));
}
__device__ __forceinline__ int pair_below(uint32_t qlo, uint32_t qhi, int g) {   // g may be a run-time value
  const int n = 4 * g + 4;
  return n <= 32 ? __popc(qlo & low_bits32(n)) : __popc(qlo) + __popc(qhi & low_bits32(n - 32));
}

// Selection rule of src/nerf_raymarch_common.py:699-757 as a set rule (the same rule select_ray in
// k_compact.hip.hpp implements with wave ballots): keep the n_max largest values (ties: lower bin first) that are >= thr; if
// none is, keep the arg-max alone.  Returns this lane's kept set as a mask over i and the ray's kept count.
//   1. per lane, the NB largest of its 64 values by branch-free insertion into a sorted register list (v_med3_f32 per slot),
//   2. partner's list by one cross-half exchange, bitonic merge -> the ray's n_max-th largest value,
//   3. cut value t = max(that, thr) (or the maximum when nothing reaches thr); kept = {x >= t}; only if more than n_max
//      values pass -- a tie at the cut-off -- the slower exact tie rule runs (wave-uniform branch).
//
// Guard band (eps_raw > 0; the caller runs a cheaper, less accurate network whose raw outputs y are within eps of the exact
// engine's x): *undecided is set unless the selection from y provably equals the selection from ANY x with |x - y| <= eps.
// With e = the bound carried through the sampler's transform (kOracleRaw: eps; sigmoid: eps / 4 since sigma' <= 1/4; softmax:
// p~ / p lies in [exp(-2 eps), exp(2 eps)], so |p~ - p| <= max(p~) (exp(2 eps) - 1); the host passes the transform's factor
// as eps_raw, guard_band_of(), so that no loop-invariant VGPR is held across the MLP for it), a ray is decided iff
//   (a) none of its n_max largest values lies within e of thr (membership of {x >= thr} is then the same for x and y; smaller
//       values are either below thr - e or cut off by (b)),
//   (b) exactly as many values reach u = max(v_n - ep, thr - e) as reach the cut value t (no value that could overtake the
//       n_max-th one; for the arg-max fallback u = v_1 - ep: the runner-up is more than ep behind).  ep bounds the error of a
//       DIFFERENCE between a kept value and a not-kept candidate (a value from max(v_n - 2 e, thr - e) upwards; anything below that is
//       out of reach by the bound on single values): x_j >= x_i needs (y_i - x_i) - (y_j - x_j) >= y_i - y_j > ep.  ep = 2 e
//       always holds; for untransformed outputs the host may pass a smaller measured bound (guard_pair_of: errors of the outputs of one
//       ray are correlated, the measured one-sided pair error is ~0.6 of 2 max|error|),
//   (c) no tie at the cut and no non-finite value.
// *cand_cut (when asked for) = max(v_n - 2 e, thr - e) (arg-max fallback: v_1 - 2 e), *kept_cut = t: what the monitor of pass 2 needs to
// find the kept set and the candidates of this ray again.
// Everything is derived from the merged sorted list c[] plus ONE counting pass over the lane's 64 values.
template <int NB>
__device__ __forceinline__ int pair_select(const float* x, int h, int n_max, float thr, uint32_t* sel_lo, uint32_t* sel_hi,
                                           float eps_raw = 0.f, int transform = 0, bool* undecided = nullptr, float pair_raw = 0.f,
                                           float* cand_cut = nullptr, float* kept_cut = nullptr) {
  static_assert(NB == 4 || NB == 8 || NB == 16, "bitonic merge");
  float s[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) s[k] = -INFINITY;
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    // NaN never ranks (fmaxf in select_ray ignores it, too) and needs no per-value replacement by -inf: v_max_f32 returns its non-NaN operand
    // and v_med3_f32 with a NaN operand returns the minimum of the other two, which in a descending list is s[k] itself (edge-case fixture:
    // NaN / inf rows of test_fused_selection_* and the fuzz cases; round 6: 2 VALU per value less, profiles/r06_variants_pe_scrub.log).
    const float v = x[i];
#pragma unroll
    for (int k = NB - 1; k >= 1; --k) s[k] = __builtin_amdgcn_fmed3f(s[k - 1], s[k], v);
    s[0] = fmaxf(s[0], v);
  }
  float c[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) c[k] = pair_xchg(s[k]);
  {
    float p[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) p[k] = fmaxf(s[k], c[NB - 1 - k]);    // the NB largest of the union, bitonic order
#pragma unroll
    for (int k = 0; k < NB; ++k) c[k] = p[k];
  }
#pragma unroll
  for (int d = NB / 2; d >= 1; d >>= 1)
#pragma unroll
    for (int k = 0; k < NB; ++k)
      if ((k & d) == 0) {
        const float a = c[k], b = c[k + d];
        c[k] = fmaxf(a, b);
        c[k + d] = fminf(a, b);
      }
  float tn = c[0];
#pragma unroll
  for (int k = 1; k < NB; ++k) tn = (k == n_max - 1) ? c[k] : tn;
  const bool none = c[0] < thr;                       // nothing reaches the threshold: arg-max alone
  const float t = none ? c[0] : fmaxf(tn, thr);
  const int n_eff = none ? 1 : n_max;

  uint32_t lo, hi;
  pair_mask<false>(x, t, &lo, &hi);
  int own = __popc(lo) + __popc(hi);
  int total = own + static_cast<int>(pair_xchg(static_cast<uint32_t>(own)));
  const bool tie = total > n_eff;
  if (eps_raw > 0.f) {      // wave-uniform
    const float e = transform == kOracleSoftmax ? eps_raw * c[0] : eps_raw;      // eps_raw: see guard_band_of()
    bool und = tie;
#pragma unroll
    for (int k = 0; k < NB; ++k) und |= (k < n_max) && (fabsf(c[k] - thr) <= e);
    const float ep = transform == kOracleSoftmax ? pair_raw * c[0] : pair_raw;      // guard_pair_of(): 0 < ep <= 2 e (may be below e)
    const float u = none ? c[0] - ep : fmaxf(tn - ep, thr - e);
    if (cand_cut) *cand_cut = none ? c[0] - 2.0f * e : fmaxf(tn - 2.0f * e, thr - e);
    if (kept_cut) *kept_cut = t;
    int cu = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) cu += (x[i] >= u) ? 1 : 0;
    cu += static_cast<int>(pair_xchg(static_cast<uint32_t>(cu)));
    und |= cu != total;
    und |= !(c[0] - c[0] == 0.f) | !(e - e == 0.f);      // inf / NaN among the candidates or in the bound
    *undecided = und | (pair_xchg(static_cast<uint32_t>(und)) != 0u);
  }
  if (__ballot(tie) != 0ull) {
    // exact tie rule: everything above t, then the lowest-bin members of {x == t} until n_eff are kept
    uint32_t glo, ghi;
    pair_mask<true>(x, t, &glo, &ghi);
    const uint32_t elo = lo & ~glo, ehi = hi & ~ghi;
    const int gown = __popc(glo) + __popc(ghi);
    const int need = n_eff - (gown + static_cast<int>(pair_xchg(static_cast<uint32_t>(gown))));
    uint32_t qlo, qhi;
    pair_align(pair_xchg(elo), pair_xchg(ehi), h, &qlo, &qhi);
    // rank of every tie member among the tie members of the ray, in bin order; walked bit by bit (ties are few, and
    // this path is rare: no unrolled per-index constants)
    uint32_t klo = 0, khi = 0;
    for (uint32_t m = tie ? elo : 0u; m; m &= m - 1u) {
      const int i = __builtin_ctz(m);
      const int rank = __popc(elo & low_bits32(i)) + pair_below(qlo, qhi, i >> 2);
      klo |= (rank < need ? 1u : 0u) << i;
    }
    for (uint32_t m = tie ? ehi : 0u; m; m &= m - 1u) {
      const int i = __builtin_ctz(m);
      const int rank = __popc(elo) + __popc(ehi & low_bits32(i)) + pair_below(qlo, qhi, 8 + (i >> 2));
      khi |= (rank < need ? 1u : 0u) << i;
    }
    if (tie) {
      lo = glo | klo;
      hi = ghi | khi;
    }
    own = __popc(lo) + __popc(hi);
    total = own + static_cast<int>(pair_xchg(static_cast<uint32_t>(own)));
  }
  if (total == 0) {          // all-NaN row: undefined in the reference; bin 0, like select_ray
    if (h == 0) lo = 1u;
    total = 1;
  }
  *sel_lo = lo;
  *sel_hi = hi;
  return total;
}

// wave64 max of non-negative floats, every lane gets the result (in-row DPP steps + readlanes: stays in the VALU, k_common.hip.hpp)
__device__ __forceinline__ float wave_max_nonneg(float v) { return wave_max_f32(v); }

// hand-issued LDS accesses (hipcc would otherwise order them against the LDS-DMA weight ring with vmcnt(0): see k_mlp16.hip.hpp)
__device__ __forceinline__ void pair_lds_write4(uint32_t byte_addr, float a, float b, float c, float d) {
  const f32x4 v = {a, b, c, d};
  asm volatile("ds_write_b128 %0, %1" ::"v"(byte_addr), "v"(v) : "memory");
}
__device__ __forceinline__ float pair_lds_read(uint32_t byte_addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(byte_addr) : "memory");
  return v;
}

// Writes the kept (bin, value) pairs of this lane in ascending-bin order of the RAY: slot = rank among the ray's kept bins.
// The values are staged in a wave-private LDS block so that a lane can fetch x[i] for a run-time i (a register array cannot be
// indexed dynamically); `stage` = LDS byte address of the wave's block + lane * 16.
// write == false (per lane; the audit of a decided ray in pass 2 of the guarded selection): nothing is stored -- the bins in place are
// compared with this lane's instead; returns whether any differs.
__device__ __forceinline__ bool pair_emit(const float* x, uint32_t lo, uint32_t hi, int h, uint32_t stage, bool valid, size_t slot0,
                                          uint8_t* __restrict__ selbin, float* __restrict__ selw, bool write = true) {
  uint32_t qlo, qhi;
  pair_align(pair_xchg(lo), pair_xchg(hi), h, &qlo, &qhi);
  const int nlo = __popc(lo);
  bool differs = false;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous pass's reads are done before the block is overwritten
#pragma unroll
    for (int q = 0; q < 8; ++q)
      pair_lds_write4(stage + q * 1024, x[32 * pass + 4 * q], x[32 * pass + 4 * q + 1], x[32 * pass + 4 * q + 2], x[32 * pass + 4 * q + 3]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint32_t own = pass ? hi : lo;
    uint32_t m = valid ? own : 0u;
    while (m) {
      const int i = __builtin_ctz(m);
      m &= m - 1u;
      const float v = pair_lds_read(stage + (i >> 2) * 1024 + (i & 3) * 4);
      const int rank = (pass ? nlo : 0) + __popc(own & low_bits32(i)) + pair_below(qlo, qhi, (32 * pass + i) >> 2);
      const uint8_t bin = static_cast<uint8_t>(act_feature(32 * pass + i, h));
      if (write) {
        selbin[slot0 + rank] = bin;
        selw[slot0 + rank] = v;
      } else {
        differs |= selbin[slot0 + rank] != bin;
      }
    }
  }
  return differs;
}

// Monitor of the guarded selection (pass 2): this lane's 64 exact raw outputs x against the ray's first-pass raw outputs -- that ray's
// [128] floats in HBM, the layout pass 1 / the oracle buffer use: lane (j, h) owns the 16 float4 at floats 32 m + 8 g + 4 h.
// The 16 loads and their wait are ONE asm statement: hipcc, left to itself, re-used one destination quad and exposed 16 memory
// latencies per 32-ray tile of a one-wave-per-SIMD kernel (+18 % on the whole refinement pass, profiles/r04_guard_monitor_cost.md);
// this exposes one.  (Issuing them before the selection and waiting after it -- two asm statements -- is not safe: between the two
// the compiler believes the destinations already hold their values and, under this kernel's register pressure, moved some of them to
// other registers before the data had landed: the monitor then compared garbage.)
struct RowRegs {
  f32x4 v[16];
};
__device__ __forceinline__ void row_load(const float* __restrict__ row, int h, RowRegs& r) {
  const float* p = row + 4 * h;
  asm volatile(
      "global_load_dwordx4 %0, %16, off\n\tglobal_load_dwordx4 %1, %16, off offset:32\n\tglobal_load_dwordx4 %2, %16, off offset:64\n\t"
      "global_load_dwordx4 %3, %16, off offset:96\n\tglobal_load_dwordx4 %4, %16, off offset:128\n\tglobal_load_dwordx4 %5, %16, off offset:160\n\t"
      "global_load_dwordx4 %6, %16, off offset:192\n\tglobal_load_dwordx4 %7, %16, off offset:224\n\t"
      "global_load_dwordx4 %8, %16, off offset:256\n\tglobal_load_dwordx4 %9, %16, off offset:288\n\tglobal_load_dwordx4 %10, %16, off offset:320\n\t"
      "global_load_dwordx4 %11, %16, off offset:352\n\tglobal_load_dwordx4 %12, %16, off offset:384\n\tglobal_load_dwordx4 %13, %16, off offset:416\n\t"
      "global_load_dwordx4 %14, %16, off offset:448\n\tglobal_load_dwordx4 %15, %16, off offset:480\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7]), "=&v"(r.v[8]),
        "=&v"(r.v[9]), "=&v"(r.v[10]), "=&v"(r.v[11]), "=&v"(r.v[12]), "=&v"(r.v[13]), "=&v"(r.v[14]), "=&v"(r.v[15])
      : "v"(p)
      : "memory");
}
// *m_abs = largest |y - x| of the ray, *m_pair = largest (y_i - x_i) - (y_j - x_j) over kept bins i (y_i >= kept_cut) and candidates j
// (cand_cut <= y_j < kept_cut); both for the whole ray (the two lanes that share it agree).  A non-finite difference counts as 0:
// such a ray is re-evaluated because of it.
__device__ __forceinline__ void pair_monitor(const float* x, const RowRegs& yr, float cand_cut, float kept_cut, bool pairs, float* m_abs, float* m_pair) {
  float ma = 0.f, dk = -INFINITY, dn = INFINITY;
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float y = yr.v[q][e];
      const float d = y - x[4 * q + e];          // q = 4 m + g: x index 16 m + 4 g + e
      const bool fin = fabsf(d) < INFINITY;      // false for NaN, too
      ma = fin ? fmaxf(ma, fabsf(d)) : ma;
      if (pairs) {
        dk = (fin && y >= kept_cut) ? fmaxf(dk, d) : dk;
        dn = (fin && y >= cand_cut && y < kept_cut) ? fminf(dn, d) : dn;
      }
    }
  ma = fmaxf(ma, pair_xchg(ma));
  dk = fmaxf(dk, pair_xchg(dk));
  dn = fminf(dn, pair_xchg(dn));
  *m_abs = ma;
  const float pr = dk - dn;      // -inf when the ray has no kept value or no candidate below the cut
  *m_pair = (pairs && pr > 0.f && pr < INFINITY) ? pr : 0.f;
}

// What pass 2 of the guarded selection learns about its rays, accumulated per WAVE over all the tiles a persistent kernel walks and
// written with five atomics at the end of the kernel (guard_flush).  Per tile they were ~4 same-address atomics per wave, counted
// by vmcnt like every other memory operation -- the weight ring's next counted wait then sat behind them: +17 % on the whole
// refinement pass (profiles/r04_guard_monitor_cost.md).  All fields are wave-uniform (they live in SGPRs).
struct GuardAcc {
  uint32_t max_bits = 0, pair_bits = 0;      // float bits of non-negative values: order like unsigned integers
  uint32_t over = 0, audited = 0, mismatch = 0;
};
__device__ __forceinline__ void guard_flush(const SelectOut& so, const GuardAcc& g, int lane) {
  if (!so.guard_seen || lane != 0) return;
  if (g.max_bits) atomicMax(&so.guard_seen[0], g.max_bits);
  if (g.over) atomicAdd(&so.guard_seen[1], g.over);
  if (g.pair_bits) atomicMax(&so.guard_seen[2], g.pair_bits);
  if (g.mismatch) atomicAdd(&so.guard_seen[3], g.mismatch);
  if (g.audited) atomicAdd(&so.guard_seen[4], g.audited);
}

// The whole epilogue for one wave's 32 rays: select, emit, per-ray counts, segment total.
//   x        64 values of ray j in lane (j, h) (layout above)
//   local    ray index of lane j inside the batch, valid = local < n_rays (invalid lanes hold a duplicate ray)
//   stage    see pair_emit
//   force_undecided  guard mode: the caller already knows the ray needs the exact engine (non-finite raw outputs)
//   entry    pass 2 of the guarded selection: the caller's copy of so.refine_list[local] (it needed the ray id anyway), else unused
//   gacc     pass 2: the wave's running monitor / audit record, flushed by the caller (guard_flush) when it has no more tiles
//   REFINE   the kernel can serve as pass 2 of the guarded selection (so.refine_list decides at run time); false in the plain-fp16
//            first-pass kernel, whose 256-register budget has no room for the monitor's row (it spilled 185 registers with it)
template <bool REFINE = true>
__device__ __forceinline__ void pair_epilogue(const float* x_raw, int lane, int local, bool valid, uint32_t stage, const SelectOut& so,
                                              bool force_undecided = false, int entry = 0, GuardAcc* gacc = nullptr) {
  const int h = lane >> 5;
  // pass 2 of the guarded selection: what the monitor and the audit read from HBM is requested first and used after the selection
  const bool monitored = REFINE && so.refine_list && so.guard_rows && so.guard_seen;      // wave-uniform
  const int target = entry & kRefineRayMask;
  float2 cuts = make_float2(0.f, 0.f);
  int old_count = 0;
  if (REFINE && so.refine_list) {
    if (monitored && so.guard_band_pair > 0.f && so.guard_probe) cuts = *reinterpret_cast<const float2*>(so.guard_probe + 2 * static_cast<size_t>(target));
    old_count = so.counts[target];
  }
  float x[64];
  if (so.transform == kOracleSigmoid) {
#pragma unroll
    for (int i = 0; i < 64; ++i) x[i] = 1.0f / (1.0f + expf(-x_raw[i]));
  } else if (so.transform == kOracleSoftmax) {
    float m = x_raw[0];
#pragma unroll
    for (int i = 1; i < 64; ++i) m = fmaxf(m, x_raw[i]);
    m = fmaxf(m, pair_xchg(m));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      x[i] = expf(x_raw[i] - m);
      sum += x[i];
    }
    sum += pair_xchg(sum);
#pragma unroll
    for (int i = 0; i < 64; ++i) x[i] = x[i] / sum;
  } else {
#pragma unroll
    for (int i = 0; i < 64; ++i) x[i] = x_raw[i];
  }
  uint32_t lo, hi;
  int total;
  bool und = false;
  float cand_cut = 0.f, kept_cut = 0.f;
  const float eps = so.guard_eps;      // pass 1: the band; 0 otherwise (host) -- a plain kernel argument, no per-lane select
  if (so.n_max <= 4) total = pair_select<4>(x, h, so.n_max, so.thr, &lo, &hi, eps, so.transform, &und, so.guard_pair, &cand_cut, &kept_cut);
  else if (so.n_max <= 8) total = pair_select<8>(x, h, so.n_max, so.thr, &lo, &hi, eps, so.transform, &und, so.guard_pair, &cand_cut, &kept_cut);
  else total = pair_select<16>(x, h, so.n_max, so.thr, &lo, &hi, eps, so.transform, &und, so.guard_pair, &cand_cut, &kept_cut);
  if (REFINE && so.refine_list) {
    // pass 2 of the guarded selection: this wave's rays are scattered over the batch.  An undecided ray: replace its row and correct
    // the total of its 32-ray segment by the difference (integer atomics: the result does not depend on their order).  A ray pass 1
    // had decided (audit): compare the row in place with the exact selection, store nothing -- the frame does not depend on which
    // rays were audited.
    const bool audit = (entry & kRefineAuditBit) != 0;
    if (monitored) {
      // the assumptions behind the band, measured on the whole row in RAW units (x_raw: before the sampler's transform)
      const bool pairs = so.guard_band_pair > 0.f && so.guard_probe != nullptr;
      float ma, mp;
      RowRegs yrow;
      row_load(so.guard_rows + static_cast<size_t>(target) * kBins, h, yrow);
      pair_monitor(x_raw, yrow, cuts.x, cuts.y, pairs, &ma, &mp);
      if (!(valid && h == 0)) ma = mp = 0.f;
      const uint64_t over = __ballot(ma > so.guard_band || (pairs && mp > so.guard_band_pair));
      ma = wave_max_nonneg(ma);
      mp = wave_max_nonneg(mp);
      const uint32_t mab = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(__builtin_bit_cast(uint32_t, ma))));
      const uint32_t mpb = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(__builtin_bit_cast(uint32_t, mp))));
      gacc->max_bits = gacc->max_bits > mab ? gacc->max_bits : mab;
      gacc->pair_bits = gacc->pair_bits > mpb ? gacc->pair_bits : mpb;
      gacc->over += static_cast<uint32_t>(__popcll(over));
    }
    const int old = valid ? old_count : 0;
    bool differs = pair_emit(x, lo, hi, h, stage, valid, static_cast<size_t>(target) * so.n_max, so.selbin, so.selw, !audit);
    differs = (differs | (pair_xchg(static_cast<uint32_t>(differs)) != 0u) | (total != old)) && audit && valid;
    if (so.guard_seen) {
      const uint64_t aud = __ballot(audit && valid && h == 0), bad = __ballot(differs && h == 0);
      gacc->audited += static_cast<uint32_t>(__popcll(aud));
      gacc->mismatch += static_cast<uint32_t>(__popcll(bad));
    }
    if (valid && h == 0 && !audit) {
      so.counts[target] = total;
      if (total != old) atomicAdd(&so.seg_total[target >> kPairSegShift], total - old);
    }
    return;
  }
  // (tune::kAblateSample & 2048: timing ablation without the rows.  Round 6 priced pair_emit at 0.024 of the kernel's 1.21 ms this way, but neither
  // fetching the staged values eight at a time (one LDS wait per pass) nor writing the rows through an LDS tile with two full-wave stores instead of up
  // to 32 partial ones moved the kernel: profiles/r06_variants_emit_ablation.log, _emit2.log, _emit_tile.log.  The simple form stays.)
  if (!(tune::kAblateSample & 2048)) pair_emit(x, lo, hi, h, stage, valid, static_cast<size_t>(local) * so.n_max, so.selbin, so.selw);
  int t = (valid && h == 0) ? total : 0;
  if (valid && h == 0) so.counts[local] = total;
  t = wave_sum_dpp_i32(t);      // lanes 0..31 hold the rays, lanes 32..63 contribute 0
  if (so.guard_mask) {
    const bool u = (und | force_undecided) && valid;
    const uint64_t b = __ballot(u);      // lanes j and j + 32 agree; bits 0..31 = the rays
    if (lane == 0 && valid) so.guard_mask[local >> kPairSegShift] = static_cast<uint32_t>(b);
    // what pass 2 monitors: the raw first-pass row and the two cut values of every ray it will look at (undecided or audited)
    const bool aud = so.audit_period > 0 && (((lane & 31) - so.audit_phase - (local >> kPairSegShift)) & (so.audit_period - 1)) == 0;
    const bool keep = (u | aud) && valid;
    if (so.guard_probe && keep && h == 0) *reinterpret_cast<float2*>(so.guard_probe + 2 * static_cast<size_t>(local)) = make_float2(cand_cut, kept_cut);
    if (so.guard_rows && keep) {
      float* row = so.guard_rows + static_cast<size_t>(local) * kBins;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(row + 32 * m + 8 * g + 4 * h) =
              make_float4(x_raw[16 * m + 4 * g], x_raw[16 * m + 4 * g + 1], x_raw[16 * m + 4 * g + 2], x_raw[16 * m + 4 * g + 3]);
    }
  }
  if (lane == 0 && valid) so.seg_total[local >> kPairSegShift] = t;   // lane 0 invalid: the whole wave is beyond n_rays
}

}  // namespace adanerf
