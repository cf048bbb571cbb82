// A1+A2+A3 on the 16-bit MFMA pipe: the split-precision sampling kernel sample_mlp16x3_kernel (default) and the
// opt-in plain-fp16 kernel sample_mlp16_kernel.
// Device code only (gfx950, wave64); part of kernels.hip.hpp.
#pragma once
#include "k_mlp16.hip.hpp"

namespace adanerf {

// ---- split-precision sampling MLP: fp16 hi + 2^-11 * fp16 lo', three MFMAs per term --------------
// x ~= hi + lo' / 2048 with hi = fp16(x), lo' = fp16((x - hi) * 2048): 22 significant bits and no
// dependence on fp16 subnormals.  W.x = Whi.xhi + (Whi.xlo' + Wlo'.xhi) / 2048 (the lo'.lo' term is
// 2^-22 relative and dropped).  Main and cross products accumulate in separate fp32 accumulators.
// Measured against an fp64 reference on the shipped weights: max error 1.9e-6 (numpy sgemm: 2.4e-6),
// identical selections on 100 % of rays -- at 3/16 of the fp32-MFMA cycle count.
// kSplitScale (2^11) is defined in pack.hpp


// hi = fp16(v), lo' = fp16((v - hi) * 2^11): v - hi is exact in fp32, and so is its product with 2^11.  Scalar arithmetic with compiler-visible
// conversions (8 instructions per pair; the library is built with -fno-slp-vectorize, so nothing re-packs the two chains into v_pk_*_f32, and there is
// no inline asm between MFMAs and their consumers: the hazard recogniser does not look inside asm -- k_mlp16.hip.hpp mfma_guard).  Forms measured and
// dropped: packed fp32 (round 5: same speed, and the library holds no packed-fp32 VALU since), v_fma_mix / v_fma_mixlo (rounds 5 / 6: 5-7 instructions,
// not faster, and they round a SUBNORMAL lo' differently from v_cvt_pk_f16_f32: the NDC fixture fails -- profiles/r06_variants_split_pack3.log).
__device__ __forceinline__ void split_pack(float v0, float v1, uint32_t* hi, uint32_t* lo) {
  const f32x2 v = {v0, v1};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  *hi = __builtin_bit_cast(uint32_t, h);
  const f32x2 r = {(v0 - static_cast<float>(h[0])) * kSplitScale, (v1 - static_cast<float>(h[1])) * kSplitScale};
  *lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
}

// One layer, fragments arrive as (hi, lo') pairs per k-step.  FPOS: first fragment position mod the chunk size.
// epilogue of one accumulator pair (values 2*pi, 2*pi+1 of tile m): v = acc + cross / 2048, then either
// the fp32 output (last layer) or ReLU + hi/lo' split for the next layer
template <bool LAST>
__device__ __forceinline__ void epilogue_pair_16x3(const f32x16& acc, const f32x16& cross, int m, int pi, uint32_t* out_hi,
                                                   uint32_t* out_lo, float* out_f32) {
  float v0 = __builtin_fmaf(cross[2 * pi], 1.0f / kSplitScale, acc[2 * pi]);
  float v1 = __builtin_fmaf(cross[2 * pi + 1], 1.0f / kSplitScale, acc[2 * pi + 1]);
  if (LAST) {
    out_f32[16 * m + 2 * pi] = v0;
    out_f32[16 * m + 2 * pi + 1] = v1;
  } else {
    split_pack(relu_bits(v0), relu_bits(v1), &out_hi[8 * m + pi], &out_lo[8 * m + pi]);
  }
}


// A finished output tile whose epilogue (combine, ReLU, hi / lo' split) has not run yet.  tune::kSplitCarry: the LAST tile of a hidden layer
// is handed to the next layer this way and converted under the MFMAs of that layer's first tile (its outputs feed k-steps 14 and 15 of the
// next layer only), instead of 96 VALU instructions + the MFMA -> VALU wait states with the matrix pipe idle at every layer boundary.
struct PendingTile3 {
  f32x16 acc, cross;
};

// HAS_PEND: `pend` holds tile 7 of the previous (hidden, 8-tile) layer; its outputs go to in_hi / in_lo[56 .. 63].
template <class WS, int KS, int MT, bool LAST, int FPOS, bool HAS_PEND>
__device__ __forceinline__ void layer_16x3(WS& st, uint32_t bias_addr, int lane, uint32_t* in_hi, uint32_t* in_lo, uint32_t* out_hi,
                                           uint32_t* out_lo, float* out_f32, PendingTile3& pend) {
  // Software pipeline across output tiles (one wave per SIMD: nothing else hides these latencies):
  //  - the bias block of tile m+1 is requested right after tile m's accumulators are initialised,
  //  - the epilogue of tile m-1 is spread, one accumulator pair per k-step, over tile m's MFMAs.
  constexpr bool PIPE = !(tune::kAblateSample & 64);
  constexpr bool CARRY = tune::kSplitCarry && PIPE && !(tune::kAblateSample & 8);
  static_assert(!HAS_PEND || KS == 16, "a pending tile writes inputs 56 .. 63: the k-steps 14 and 15 of a 256-wide layer");
  constexpr bool kCounted = tune::kSplitBiasCounted && tune::kSchedGroupsSampling && !(tune::kAblateSample & 2);
  BiasRegs br;
  f32x16 pacc, pcross;   // previous tile's accumulators (PIPE)
  if (HAS_PEND && CARRY) {
    pacc = pend.acc;
    pcross = pend.cross;
  }
  if (!(tune::kAblateSample & 4)) lds_bias_issue(bias_addr, br, st.rd_cur, st.rd_next);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    f32x16 acc, cross;
    if (tune::kSchedGroupsSampling) __builtin_amdgcn_sched_barrier(0);      // one scheduling region per tile
    if (tune::kAblateSample & 4) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    } else {
      // m > 0: the block was requested at the head of tile m - 1, in front of that tile's 2 KS fragment re-fills
      constexpr int kYounger = kCounted ? (2 * KS < 15 ? 2 * KS : 15) : 0;
      if (m == 0) lds_bias_take<0>(br, &acc);
      else lds_bias_take<kYounger>(br, &acc);
      if (m + 1 < MT) lds_bias_issue(bias_addr + (m + 1) * 128, br, st.rd_cur, st.rd_next);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) cross[r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int f = (FPOS + 2 * (m * KS + s)) % tune::kChunkFragsSampling;   // compile-time after unrolling; always even
      ws_position<tune::kAblateSample>(st, f, s == 0);
      const u32x4 bh = {in_hi[4 * s], in_hi[4 * s + 1], in_hi[4 * s + 2], in_hi[4 * s + 3]};
      const u32x4 bl = {in_lo[4 * s], in_lo[4 * s + 1], in_lo[4 * s + 2], in_lo[4 * s + 3]};
      acc = Fp16::mfma(st.R[f % WS::kRegs], bh, acc);
      cross = Fp16::mfma(st.R[f % WS::kRegs], bl, cross);
      cross = Fp16::mfma(st.R[(f + 1) % WS::kRegs], bh, cross);
      ws_refill<tune::kAblateSample>(st, f);
      ws_refill<tune::kAblateSample>(st, f + 1);
      if (PIPE && !(tune::kAblateSample & 8) && (m > 0 || (HAS_PEND && CARRY))) {
        // the 8 pairs of the previous tile, spread over the k-steps from E0 = 1 on (PER per k-step): the first k-step keeps the MFMA -> VALU wait
        // states of the previous tile's last MFMAs out of the stream
        constexpr int E0 = KS > 2 ? 1 : 0;
        constexpr int PER = (KS - E0 >= 8) ? 1 : (8 + (KS - E0) - 1) / (KS - E0);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
          const int pi = (s - E0) * PER + k;
          if (s >= E0 && pi < 8) {
            if (m > 0) epilogue_pair_16x3<LAST>(pacc, pcross, m - 1, pi, out_hi, out_lo, out_f32);
            else epilogue_pair_16x3<false>(pacc, pcross, 7, pi, in_hi, in_lo, nullptr);      // the previous layer's last tile: read by k-steps 14, 15
          }
        }
      }
      if (tune::kSchedGroupsSampling) {
        // pin the k-step's interleave: three MFMAs, two fragment re-fills, the VALU of one epilogue pair spread between them
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, tune::kSgbValuSampling, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, tune::kSgbValuSampling, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, tune::kSgbValuSampling, 0);
        if (tune::kSplitKstepFence) __builtin_amdgcn_sched_barrier(0);
      }
    }
    if ((tune::kAblateSample & 8) && !LAST) {
      asm volatile("" ::"v"(acc), "v"(cross));
#pragma unroll
      for (int g = 0; g < 8; ++g) asm volatile("" : "=v"(out_hi[8 * m + g]), "=v"(out_lo[8 * m + g]));
      continue;
    }
    if (PIPE && m + 1 < MT) {
      pacc = acc;
      pcross = cross;
    } else if (CARRY && !LAST) {
      static_assert(LAST || !tune::kSplitCarry || MT == 8, "the carried tile is tile 7");
      pend.acc = acc;
      pend.cross = cross;
    } else {
#pragma unroll
      for (int pi = 0; pi < 8; ++pi) epilogue_pair_16x3<LAST>(acc, cross, m, pi, out_hi, out_lo, out_f32);
    }
  }
}

// A1+A2+A3 on the split-precision engine.  Workgroup = 4 waves (one per SIMD, <= 512 registers:
// 2 x (hi, lo') activation sets of 64 VGPRs + 2 accumulators) x 32 rays = 128-ray tile; persistent
// over tiles; weights streamed once per workgroup through the LDS ring like the shading kernel.
template <int FP, int FD>
__global__ __launch_bounds__(256) void sample_mlp16x3_kernel(SampleArgs a) {
  constexpr int QD = pe_slots(FD), QP = pe_slots(FP), Q0 = QD + QP;
  constexpr int WAVES = 4, CF = tune::kChunkFragsSampling, RS = tune::kRingSlotsSampling, LPW = CF / WAVES, TILE = WAVES * 32;
  constexpr int F0 = 2 * (Q0 / 8) * 8;                  // layer-0 fragments (hi + lo')
  constexpr int FRAGS = F0 + 6 * 256 + 128;
  static_assert(F0 % CF == 0 && FRAGS % CF == 0 && CF % WAVES == 0 && CF % tune::kRegFragsSampling == 0 && CF <= 32, "chunk geometry");
  typedef WStream<CF, RS, LPW, tune::kRegFragsSampling> WS;
  constexpr int kRingBytes = CF * RS * 1024, kBiasFloats = 7 * 256 + 128;
  __shared__ __attribute__((aligned(16))) char lds[kRingBytes + kBiasFloats * 4 + WAVES * kPairLdsBytesPerWave];
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const int j = lane & 31, h = lane >> 5;
  // refinement pass of the guarded selection: the rays are listed in a.ray_list, their number is known on the device only
  int n_rays = a.n_rays;
  if (a.ray_list) n_rays = min(n_rays, __builtin_amdgcn_readfirstlane(*a.n_list));
  const int ntiles = (n_rays + TILE - 1) / TILE;
  if (static_cast<int>(blockIdx.x) >= ntiles) return;
  // staging block of the fused selection (pair_emit), wave-private
  const uint32_t sel_stage = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds)) + kRingBytes + kBiasFloats * 4 +
                             wave * kPairLdsBytesPerWave + lane * 16;

  {
    float* lds_bias = reinterpret_cast<float*>(lds + kRingBytes);
    for (int i = threadIdx.x; i < kBiasFloats; i += blockDim.x) lds_bias[i] = a.net16.bias[i];
  }
  __syncthreads();
  WS st;
  ws_start(st, a.net16.w, FRAGS * 1024, lds, wave, lane);
  const uint32_t bias0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds)) + kRingBytes + h * 64;
  const uint32_t* bo = a.net16.b_off;

  GuardAcc gacc;      // refinement pass: the wave's monitor / audit record over all its tiles
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int local = tile * TILE + wave * 32 + j;
    const bool valid = local < n_rays;
    const int lidx = valid ? local : n_rays - 1;
    const int entry = a.ray_list ? a.ray_list[lidx] : 0;      // refinement pass: ray id (+ kRefineAuditBit), kept for the epilogue
    const int ray = a.first_ray + (a.ray_list ? (entry & kRefineRayMask) : lidx);
    int col, row;
    ray_pixel(a.g, ray, &col, &row);
    float nds[3], p[3], u[3];
    gen_ray(a.g, col, row, nds, p);
    unit3(nds, u);

    uint32_t aH[64], aL[64], bH[64], bL[64];
    if (tune::kAblateSample & 256) {      // timing ablation (wrong results): no encoding, inputs = cheap functions of the ray
#pragma unroll
      for (int q = 0; q < Q0 / 2; ++q) {
        aH[q] = __builtin_bit_cast(uint32_t, u[q % 3]) >> 3;
        aL[q] = __builtin_bit_cast(uint32_t, p[q % 3]) >> 5;
      }
    } else {
      float t[Q0];
      pe_eval<FD, !(tune::kAblateSample & 128)>(u, h, t);            // [dir PE | pos PE]  (src/features.py:868-874)
      pe_eval<FP, !(tune::kAblateSample & 128)>(p, h, t + QD);
#pragma unroll
      for (int q = 0; q < Q0 / 2; ++q) split_pack(t[2 * q], t[2 * q + 1], &aH[q], &aL[q]);
    }
    PendingTile3 pend;      // a hidden layer's last tile, converted under the next layer's first MFMAs (tune::kSplitCarry)
    layer_16x3<WS, Q0 / 8, 8, false, 0, false>(st, bias0 + bo[0] * 4, lane, aH, aL, bH, bL, nullptr, pend);
#pragma unroll 1
    for (int l = 1; l <= 5; l += 2) {
      layer_16x3<WS, 16, 8, false, 0, true>(st, bias0 + bo[l] * 4, lane, bH, bL, aH, aL, nullptr, pend);
      layer_16x3<WS, 16, 8, false, 0, true>(st, bias0 + bo[l + 1] * 4, lane, aH, aL, bH, bL, nullptr, pend);
    }
    float out[64];
    layer_16x3<WS, 16, 4, true, 0, true>(st, bias0 + bo[7] * 4, lane, bH, bL, nullptr, nullptr, out, pend);

    if (a.fused_select) {
      // A4 in the epilogue: the 128 raw outputs of ray j sit in lanes j and j + 32 (k_select_pair.hip.hpp)
      float z = 0.f;
#pragma unroll
      for (int i = 0; i < 64; ++i) z = __builtin_fmaf(out[i], 0.f, z);    // NaN iff some output is inf / NaN (fp16 range left)
      const bool bad_ray = (z != z) | (pair_xchg(static_cast<uint32_t>(z != z)) != 0u);
      if (bad_ray && valid && h == 0 && a.overflow_flag) atomicAdd(a.overflow_flag, 1);
      if (tune::kAblateSample & 512) {      // timing ablation (wrong results): no selection, one count per ray from a cheap function of the outputs
        if (valid && h == 0) a.sel.counts[local] = 1 + (__builtin_bit_cast(int, out[0] + out[17] + out[35] + out[63]) & 3);
        if (lane == 0 && valid) a.sel.seg_total[local >> kPairSegShift] = 64;
      } else {
        pair_epilogue(out, lane, local, valid, sel_stage, a.sel, false, entry, &gacc);
      }
    }
    if (valid) {
      if (a.oracle_out) {
        float* o = a.oracle_out + static_cast<size_t>(local) * kBins;
        bool bad = false;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float4 v = make_float4(out[16 * m + 4 * g], out[16 * m + 4 * g + 1], out[16 * m + 4 * g + 2], out[16 * m + 4 * g + 3]);
            bad |= !(fabsf(v.x) < 3.0e38f) | !(fabsf(v.y) < 3.0e38f) | !(fabsf(v.z) < 3.0e38f) | !(fabsf(v.w) < 3.0e38f);
            *reinterpret_cast<float4*>(o + 32 * m + 8 * g + 4 * h) = v;
          }
        // an activation beyond the fp16 range (65504) shows up as inf/NaN here; a ray is owned by lanes j and j + 32
        // (64 outputs each), counted once
        const bool bad_ray = bad | (pair_xchg(static_cast<uint32_t>(bad)) != 0u);
        if (bad_ray && h == 0 && a.overflow_flag) atomicAdd(a.overflow_flag, 1);
      }
      if (a.rays_out) {
        float ro[3] = {p[0], p[1], p[2]}, rd[3] = {nds[0], nds[1], nds[2]};
        if (a.g.use_ndc) ndc_ray(a.g, p, nds, ro, rd);
        float4* r = reinterpret_cast<float4*>(a.rays_out + static_cast<size_t>(local) * 8);
        if (h == 0) r[0] = make_float4(ro[0], ro[1], ro[2], 0.f);
        else r[1] = make_float4(rd[0], rd[1], rd[2], 0.f);
      }
    }
  }
  if (a.fused_select && a.ray_list) guard_flush(a.sel, gacc, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// A1+A2+A3 in plain fp16 (ADANERF_SAMPLING_FP16): one MFMA per term, fp32 accumulate -- the arithmetic the
// reference VIEWER runs its sampling network in (TensorRT kFP16, adanerf_real_time_viewer/src/imagegenerator.cpp:155-156).
// 11-bit operands move a few outputs across the threshold / the N-th rank (raw error <= 3e-3), so the selected bins
// differ from the fp32 PyTorch path on 0.3-1.5 % of rays: an opt-in speed mode, never the default.  Same engine as the shading
// kernel (8 waves x 32 rays per workgroup, activations in registers, weights through the LDS ring).
// Revision of the arithmetic of the two engines the guarded selection compares (this kernel: encoding, operand rounding, summation
// order; sample_mlp16x3_kernel likewise): part of the key of the calibration record (adanerf_guard_calibration_file) -- bump it
// with any change that moves the raw outputs, so that recorded error bounds of another engine are measured again.
constexpr int kGuardEngineRev = 4;

template <int FP, int FD>
constexpr int sample16_frags() { return ((pe_slots(FD) + pe_slots(FP)) / 8) * 8 + 6 * 128 + 64; }

// Occupancy: one 8-wave workgroup per CU by design (two waves per SIMD from the SAME workgroup, staggered half a chunk apart):
// __launch_bounds__(512, 2) = two waves per SIMD = 256 registers; the LDS footprint (64 KB ring + 7.5 KB biases + 8 x 8 KB
// selection staging = 135.5 KB) admits no second workgroup either, with or without the fused selection.
template <int FP, int FD>
__global__ __launch_bounds__(512, 2) void sample_mlp16_kernel(SampleArgs a) {
  constexpr int QD = pe_slots(FD), QP = pe_slots(FP), Q0 = QD + QP;
  constexpr bool kOneGroupDma = tune::kDmaGroup >= 0;
  constexpr int WAVES = 8, CF = 16, RS = 4, LPW = kOneGroupDma ? CF / 4 : CF / WAVES, TILE = WAVES * 32;   // 880 fragments = 55 chunks of 16
  constexpr int F0 = (Q0 / 8) * 8, FRAGS = sample16_frags<FP, FD>();
  static_assert(F0 % CF == 0 && FRAGS % CF == 0 && CF % WAVES == 0 && CF % kRegFrags == 0 && CF <= 32, "chunk geometry");
  typedef WStream<CF, RS, LPW> WS;
  constexpr int kRingBytes = CF * RS * 1024, kBiasFloats = 7 * 256 + 128;
  __shared__ __attribute__((aligned(16))) char lds[kRingBytes + kBiasFloats * 4 + WAVES * kPairLdsBytesPerWave];
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const int j = lane & 31, h = lane >> 5;
  // Work is dealt out in wave tiles (32 rays): q full rounds of 8 wave tiles per workgroup, then the remaining rem < 8 G wave tiles
  // in ONE partial round of k = ceil(rem / G) waves per workgroup instead of a full round on some workgroups and none on the
  // others.  A round with <= 4 active waves has every SIMD to itself and takes a little over half a full round, which is what a
  // small batch (an 80 000-ray shard of an 8-GPU frame: 1.22 rounds) gains.  The idle waves of that round only keep the weight
  // ring's barriers and DMA going (ring_walk).
  const int WT = (a.n_rays + 31) >> 5, G = static_cast<int>(gridDim.x);
  const int q = WT / (WAVES * G), rem = WT - WAVES * G * q, k = (rem + G - 1) / G;
  const bool has_partial = static_cast<int>(blockIdx.x) * k < rem;          // workgroup-uniform
  const int rounds = q + (has_partial ? 1 : 0);
  if (rounds == 0) return;
  const uint32_t sel_stage = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds)) + kRingBytes + kBiasFloats * 4 +
                             wave * kPairLdsBytesPerWave + lane * 16;
  {
    float* lds_bias = reinterpret_cast<float*>(lds + kRingBytes);
    for (int i = threadIdx.x; i < kBiasFloats; i += blockDim.x) lds_bias[i] = a.net16.bias[i];
  }
  __syncthreads();
  WS st;
  ws_start(st, a.net16.w, FRAGS * 1024, lds, kOneGroupDma ? (wave & 3) : wave, lane, tune::kStagger ? (wave >> 2) : -1,
           !kOneGroupDma || (wave >> 2) == tune::kDmaGroup);
  const uint32_t bias0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds)) + kRingBytes + h * 64;
  const uint32_t* bo = a.net16.b_off;

  for (int rd = 0; rd < rounds; ++rd) {
    const int wt = rd < q ? (rd * G + static_cast<int>(blockIdx.x)) * WAVES + wave : WAVES * G * q + static_cast<int>(blockIdx.x) * k + wave;
    if (rd >= q && (wave >= k || wt >= WT)) {      // wave-uniform: an idle wave of the partial round
#pragma unroll 1
      for (int c = 0; c < FRAGS / CF; ++c) {
#pragma unroll
        for (int pp = 0; pp < CF; pp += CF / LPW) ws_position<0>(st, pp, false);      // every synchronisation point and piece position of a chunk
      }
      continue;
    }
    const int local = wt * 32 + j;
    const bool valid = local < a.n_rays;
    const int ray = a.first_ray + (valid ? local : a.n_rays - 1);
    int col, row;
    ray_pixel(a.g, ray, &col, &row);
    float nds[3], p[3], u[3];
    gen_ray(a.g, col, row, nds, p);
    unit3(nds, u);
    if (valid && a.rays_out) {
      float ro[3] = {p[0], p[1], p[2]}, rd[3] = {nds[0], nds[1], nds[2]};
      if (a.g.use_ndc) ndc_ray(a.g, p, nds, ro, rd);
      float4* r = reinterpret_cast<float4*>(a.rays_out + static_cast<size_t>(local) * 8);
      if (h == 0) r[0] = make_float4(ro[0], ro[1], ro[2], 0.f);
      else r[1] = make_float4(rd[0], rd[1], rd[2], 0.f);
    }
    uint32_t hA[64], hB[64];
    {
      float t[Q0];
      if (tune::kAblateSample & 256) {      // timing ablation (wrong results): no encoding
#pragma unroll
        for (int q = 0; q < Q0; ++q) t[q] = (q & 1) ? u[q % 3] : p[q % 3];
      } else {
        pe_eval<FD, !(tune::kAblateSample & 128) && !tune::kFastPeFp16Pass>(u, h, t);            // [dir PE | pos PE]  (src/features.py:868-874)
        pe_eval<FP, !(tune::kAblateSample & 128) && !tune::kFastPeFp16Pass>(p, h, t + QD);
      }
      uint32_t in0[Q0 / 2];
#pragma unroll
      for (int q = 0; q < Q0 / 2; ++q) in0[q] = Fp16::pack(t[2 * q], t[2 * q + 1]);
      layer_16<Fp16, WS, Q0 / 8, 0, 8, true, 0>(st, bias0 + bo[0] * 4, lane, in0, in0, hA);
    }
#pragma unroll 1
    for (int l = 1; l <= 5; l += 2) {
      layer_16<Fp16, WS, 16, 0, 8, true, F0 % CF>(st, bias0 + bo[l] * 4, lane, hA, hA, hB);
      layer_16<Fp16, WS, 16, 0, 8, true, F0 % CF>(st, bias0 + bo[l + 1] * 4, lane, hB, hB, hA);
    }
    f32x16 out[4];
    layer_16<Fp16, WS, 16, 0, 4, false, F0 % CF, kKeepAllF32>(st, bias0 + bo[7] * 4, lane, hA, hA, hB, out);
    if (a.fused_select) {
      float x[64];
      float z = 0.f;
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        x[i] = out[i >> 4][i & 15];
        z = __builtin_fmaf(x[i], 0.f, z);
      }
      const bool bad_ray = (z != z) | (pair_xchg(static_cast<uint32_t>(z != z)) != 0u);
      // guard mode: a non-finite ray goes to the refinement pass, which does the counting
      if (bad_ray && valid && h == 0 && a.overflow_flag && !a.sel.guard_mask) atomicAdd(a.overflow_flag, 1);
      pair_epilogue<false>(x, lane, local, valid, sel_stage, a.sel, bad_ray);
    }
    if (valid && a.oracle_out) {
      float* o = a.oracle_out + static_cast<size_t>(local) * kBins;
      bool bad = false;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = make_float4(out[m][4 * g], out[m][4 * g + 1], out[m][4 * g + 2], out[m][4 * g + 3]);
          bad |= !(fabsf(v.x) < 3.0e38f) | !(fabsf(v.y) < 3.0e38f) | !(fabsf(v.z) < 3.0e38f) | !(fabsf(v.w) < 3.0e38f);
          *reinterpret_cast<float4*>(o + 32 * m + 8 * g + 4 * h) = v;
        }
      const bool bad_ray = bad | (pair_xchg(static_cast<uint32_t>(bad)) != 0u);   // lanes j and j + 32 own one ray
      if (bad_ray && h == 0 && a.overflow_flag) atomicAdd(a.overflow_flag, 1);    // an activation left the fp16 range
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// The same network in plain fp16 with TWO 32-ray blocks per wave (layer_16x2, the shading kernel's engine): 4 waves x 64 rays = 256 rays per weight pass,
// every fragment read from LDS feeds two MFMAs, half the LDS-DMA bytes / barriers / bias reads per ray of sample_mlp16_kernel.  Same packed weights, same
// k-step order, same epilogue arithmetic: the raw outputs are sample_mlp16_kernel's bit for bit (so the guard band's calibration record holds for both,
// kGuardEngineRev unchanged; tests/test_gpu_parity.py compares the two).  Used for batches of at least kSample16x2MinRays rays with the fused selection (the
// first pass of the guarded mode, the plain-fp16 speed mode); smaller batches -- the strip shares of an N-GPU frame -- keep sample_mlp16_kernel, whose last
// round is dealt out wave by wave.
constexpr int kSample16x2MinRays = 4 * 256 * 256;      // four full rounds of a 256-CU grid

template <int FP, int FD>
__global__ __launch_bounds__(256) void sample_mlp16x2_kernel(SampleArgs a) {
  constexpr int QD = pe_slots(FD), QP = pe_slots(FP), Q0 = QD + QP;
  constexpr int WAVES = 4, CF = 16, RS = 6, LPW = CF / WAVES, TILE = WAVES * 64;
  constexpr int F0 = (Q0 / 8) * 8, FRAGS = sample16_frags<FP, FD>();
  static_assert(F0 % CF == 0 && FRAGS % CF == 0 && CF % tune::kRegFrags2 == 0, "chunk geometry");
  typedef WStream<CF, RS, LPW, tune::kRegFrags2> WS;
  constexpr int kRingBytes = CF * RS * 1024, kBiasFloats = 7 * 256 + 128;
  __shared__ __attribute__((aligned(16))) char lds[kRingBytes + kBiasFloats * 4 + WAVES * kPairLdsBytesPerWave];
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int ntiles = (a.n_rays + TILE - 1) / TILE;
  if (static_cast<int>(blockIdx.x) >= ntiles) return;
  const uint32_t sel_stage = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds)) + kRingBytes + kBiasFloats * 4 +
                             wave * kPairLdsBytesPerWave + lane * 16;
  {
    float* lds_bias = reinterpret_cast<float*>(lds + kRingBytes);
    for (int i = threadIdx.x; i < kBiasFloats; i += blockDim.x) lds_bias[i] = a.net16.bias[i];
  }
  __syncthreads();
  WS st;
  ws_start(st, a.net16.w, FRAGS * 1024, lds, wave, lane);
  const uint32_t bias0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds)) + kRingBytes + h * 64;
  const uint32_t* bo = a.net16.b_off;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    uint32_t hA0[64], hB0[64], hA1[64], hB1[64];
    PendingTile2 pend;
    int local[2];
    bool valid[2];
    {
      uint32_t in0[2][Q0 / 2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        local[b] = tile * TILE + wave * 64 + 32 * b + j;
        valid[b] = local[b] < a.n_rays;
        const int ray = a.first_ray + (valid[b] ? local[b] : a.n_rays - 1);
        int col, row;
        ray_pixel(a.g, ray, &col, &row);
        float nds[3], p[3], u[3];
        gen_ray(a.g, col, row, nds, p);
        unit3(nds, u);
        if (valid[b] && a.rays_out) {
          float ro[3] = {p[0], p[1], p[2]}, rd[3] = {nds[0], nds[1], nds[2]};
          if (a.g.use_ndc) ndc_ray(a.g, p, nds, ro, rd);
          float4* r = reinterpret_cast<float4*>(a.rays_out + static_cast<size_t>(local[b]) * 8);
          if (h == 0) r[0] = make_float4(ro[0], ro[1], ro[2], 0.f);
          else r[1] = make_float4(rd[0], rd[1], rd[2], 0.f);
        }
        float t[Q0];
        pe_eval<FD, !tune::kFastPeFp16Pass>(u, h, t);            // [dir PE | pos PE]  (src/features.py:868-874), as sample_mlp16_kernel
        pe_eval<FP, !tune::kFastPeFp16Pass>(p, h, t + QD);
#pragma unroll
        for (int q = 0; q < Q0 / 2; ++q) in0[b][q] = Fp16::pack(t[2 * q], t[2 * q + 1]);
      }
      layer_16x2<Fp16, WS, Q0 / 8, 0, 8, true, 0, -1, -1, true, true>(st, bias0 + bo[0] * 4, in0[0], in0[0], in0[1], in0[1], hA0, hA1, pend);
    }
#pragma unroll 1
    for (int l = 1; l <= 5; l += 2) {
      layer_16x2<Fp16, WS, 16, 0, 8, true, 0, -1, 7, true, true>(st, bias0 + bo[l] * 4, hA0, hA0, hA1, hA1, hB0, hB1, pend, hA0, hA1);
      layer_16x2<Fp16, WS, 16, 0, 8, true, 0, -1, 7, true, true>(st, bias0 + bo[l + 1] * 4, hB0, hB0, hB1, hB1, hA0, hA1, pend, hB0, hB1);
    }
    f32x16 outA[4], outB[4];
    layer_16x2<Fp16, WS, 16, 0, 4, false, 0, kKeepAllF32, 7, true, false>(st, bias0 + bo[7] * 4, hA0, hA0, hA1, hA1, hB0, hB1, pend, hA0, hA1, outA, outB);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float x[64];
      float z = 0.f;
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        x[i] = b ? outB[i >> 4][i & 15] : outA[i >> 4][i & 15];
        z = __builtin_fmaf(x[i], 0.f, z);
      }
      const bool bad_ray = (z != z) | (pair_xchg(static_cast<uint32_t>(z != z)) != 0u);
      // guard mode: a non-finite ray goes to the refinement pass, which does the counting
      if (bad_ray && valid[b] && h == 0 && a.overflow_flag && !a.sel.guard_mask) atomicAdd(a.overflow_flag, 1);
      pair_epilogue<false>(x, lane, local[b], valid[b], sel_stage, a.sel, bad_ray);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace adanerf
