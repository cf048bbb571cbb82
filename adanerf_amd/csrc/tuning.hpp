// Build-time tuning constants of the MLP engines, in one place.  The shipped library is built with the defaults
// below.  tools/ablate.sh builds experiment variants with -DADN_EXPERIMENT -DADN_<NAME>=<value> (ring geometry sweeps,
// timing ablations for the tables in profiles/); without ADN_EXPERIMENT the -D overrides are ignored.
#pragma once

namespace adanerf {
namespace tune {


#if defined(ADN_EXPERIMENT)
#define ADN_OVERRIDABLE 1
#else
#define ADN_OVERRIDABLE 0
#endif

// ---- LDS weight ring of the 16-bit engines (k_mlp16.hip.hpp) ---------------------------------------------------------
// CF = fragments (KiB) per chunk = MFMAs per wave between two synchronisation points; RS = ring slots.  At
// synchronisation point k a wave waits for its own pieces of chunk k+1, so RS-3 further chunks stay in flight.
#if ADN_OVERRIDABLE && defined(ADN_CF)
constexpr int kChunkFrags = ADN_CF;
#else
constexpr int kChunkFrags = 16;
#endif
#if ADN_OVERRIDABLE && defined(ADN_RS)
constexpr int kRingSlots = ADN_RS;
#else
constexpr int kRingSlots = 4;
#endif
// the split-precision sampling kernel (one wave per SIMD) has its own geometry
#if ADN_OVERRIDABLE && defined(ADN_CF_S)
constexpr int kChunkFragsSampling = ADN_CF_S;
#else
constexpr int kChunkFragsSampling = 32;      // round 6: 32 fragments x 3 slots (a barrier every 48 MFMAs, pieces issued in the first half of a chunk): 1.228 -> 1.218 ms
#endif
#if ADN_OVERRIDABLE && defined(ADN_RS_S)
constexpr int kRingSlotsSampling = ADN_RS_S;
#else
constexpr int kRingSlotsSampling = 3;
#endif
// plain-fp16 sampling kernel (speed mode, first pass of the guarded mode): encoding through one v_sin_f32 per slot instead of the
// fp32-parity sin_or_cos -- its values are rounded to fp16 right after (2.4e-4), the guard band is calibrated with whatever this
// kernel computes.  Measured (profiles/r03_variants_fast_pe.log): sampling stage 1.014 -> 0.980 ms, calibrated band and refined
// rays unchanged (6.5525e-3 / 249 183 -> 6.5535e-3 / 249 226)
#if ADN_OVERRIDABLE && defined(ADN_FAST_PE16)
constexpr bool kFastPeFp16Pass = ADN_FAST_PE16 != 0;
#else
constexpr bool kFastPeFp16Pass = true;
#endif
// ---- run-time-shaped 16-bit kernels (k_generic16.hip.hpp) -------------------------------------------------------------
// kGenericStaged: weights of one output tile staged through LDS for the whole workgroup (false: every wave fetches its own
// fragments from L2 -- the experiment baseline of profiles/r03_generic_staged.md); kGenericBlocks128 / 256: 32-sample blocks per
// wave of the staged shading kernel at that width (10-4 layout; k_generic16.hip.hpp gen_blocks / gen_occupancy).  Measured: 2 at
// width 256 (2.82 vs 3.42 ms, round 3); at width 128 round 3 shipped 1 block at 3 workgroups per CU (0.889 ms; two blocks spilled),
// round 4 -- bias table in LDS, no bias registers -- 2 blocks at 2 per CU (0.709 vs 0.758 ms on one box, profiles/r04_lab_log.md).
#if ADN_OVERRIDABLE && defined(ADN_GEN_STAGED)
constexpr bool kGenericStaged = ADN_GEN_STAGED != 0;
#else
constexpr bool kGenericStaged = true;
#endif
#if ADN_OVERRIDABLE && defined(ADN_GEN_NB128)
constexpr int kGenericBlocks128 = ADN_GEN_NB128;
#else
constexpr int kGenericBlocks128 = 2;
#endif
// workgroups per CU asked of the compiler at width 128 (10-4 layout; the 16-band layout runs one block at 2 per CU)
#if ADN_OVERRIDABLE && defined(ADN_GEN_OCC128)
constexpr int kGenericOcc128 = ADN_GEN_OCC128;
#else
constexpr int kGenericOcc128 = 2;
#endif
// workgroups per CU at width 64 (two blocks per wave; the 16-band layout's LDS footprint admits two)
#if ADN_OVERRIDABLE && defined(ADN_GEN_OCC64)
constexpr int kGenericOcc64 = ADN_GEN_OCC64;
#else
constexpr int kGenericOcc64 = 2;
#endif
// workgroups per CU at width 256 (10-4 layout): 1 with two blocks per wave (512 registers); 2 is only meaningful with one block
#if ADN_OVERRIDABLE && defined(ADN_GEN_OCC256)
constexpr int kGenericOcc256 = ADN_GEN_OCC256;
#else
constexpr int kGenericOcc256 = 1;
#endif
// kGenericSpread: a tile's conversions are spread over the k-steps of the NEXT tile, between its MFMAs, out of a second accumulator set
// (k_generic16.hip.hpp layer_16_staged).  1: at width 256 only (one wave per SIMD, registers to spare); 2: every width; 0: off.
// Measured (profiles/r04_variants_generic_spread.log): 5 x 256 2.496 -> 2.444 ms with 1; nothing at width 128, slower at width 64 with 2.
#if ADN_OVERRIDABLE && defined(ADN_GEN_SPREAD)
constexpr int kGenericSpread = ADN_GEN_SPREAD;
#else
constexpr int kGenericSpread = 1;
#endif
// kGenericDefer: a tile's conversions run behind the NEXT tile's barrier and first fragment requests (k_generic16.hip.hpp, layer_16_staged)
#if ADN_OVERRIDABLE && defined(ADN_GEN_DEFER)
constexpr bool kGenericDefer = ADN_GEN_DEFER != 0;
#else
constexpr bool kGenericDefer = false;      // measured: no effect with three waves per SIMD to cover for each other (r04_lab_log.md)
#endif
// kGenericBiasDirect: a tile's bias block is read from the LDS table straight into its accumulators at the head of the tile
// (false: requested a tile ahead into 16 registers of their own and copied)
#if ADN_OVERRIDABLE && defined(ADN_GEN_BIAS_DIRECT)
constexpr bool kGenericBiasDirect = ADN_GEN_BIAS_DIRECT != 0;
#else
constexpr bool kGenericBiasDirect = true;
#endif
// kGenericAhead: fragments a wave requests from LDS ahead of the k-step that consumes them
#if ADN_OVERRIDABLE && defined(ADN_GEN_AHEAD)
constexpr int kGenericAhead = ADN_GEN_AHEAD;
#else
constexpr int kGenericAhead = 4;
#endif
#if ADN_OVERRIDABLE && defined(ADN_GEN_NB256)
constexpr int kGenericBlocks256 = ADN_GEN_NB256;
#else
constexpr int kGenericBlocks256 = 2;
#endif
// Fragments held in registers per wave (= LDS prefetch distance in MFMAs).  Two waves per SIMD (256-register cap): 4
// (2: 3.71-3.82 ms, 8: 3.66-3.76 ms with 4 spilled registers, against 3.58-3.62 on the same box).  The one-wave-per-SIMD
// split sampling kernel keeps a whole chunk (4: 1.42, 8: 1.37, 16: 1.32 ms).
#if ADN_OVERRIDABLE && defined(ADN_NR)
constexpr int kRegFrags = ADN_NR;
#else
constexpr int kRegFrags = 4;
#endif
#if ADN_OVERRIDABLE && defined(ADN_NR_S)
constexpr int kRegFragsSampling = ADN_NR_S;
#else
constexpr int kRegFragsSampling = 16;
#endif
// 8-wave workgroups: waves 4-7 synchronise half a chunk after waves 0-3, so the two waves of a SIMD run half an output
// tile apart (ws_sync); -1 / 0 / 1: every wave / only waves 0-3 / only waves 4-7 DMA-copy the weight pieces.
#if ADN_OVERRIDABLE && defined(ADN_STAGGER)
constexpr bool kStagger = ADN_STAGGER != 0;
#else
constexpr bool kStagger = true;
#endif
#if ADN_OVERRIDABLE && defined(ADN_DMA_GRP)
constexpr int kDmaGroup = ADN_DMA_GRP;
#else
constexpr int kDmaGroup = 0;
#endif
// s_nop argument (wait states - 1) on the short side of a branch that follows a tile's last MFMA (ws_position);
// -1: no padding (profiles/r02_stagger_hazard.md: wrong results in some variants)
#if ADN_OVERRIDABLE && defined(ADN_PAD)
constexpr int kSkipPad = ADN_PAD;
#else
constexpr int kSkipPad = 11;
#endif

// ---- shading kernel with two sample blocks per wave (shade_mlp16x2_kernel, one wave per SIMD) ----------------------------
// kShadeBlocks: 1 = shade_mlp16_kernel (8 waves x 32 samples), 2 = shade_mlp16x2_kernel (4 waves x 64 samples; shipped since
// round 3: profiles/r03_shade_two_blocks.md)
#if ADN_OVERRIDABLE && defined(ADN_SHADE_BLOCKS)
constexpr int kShadeBlocks = ADN_SHADE_BLOCKS;
#else
constexpr int kShadeBlocks = 2;
#endif
#if ADN_OVERRIDABLE && defined(ADN_CF2)
constexpr int kChunkFrags2 = ADN_CF2;
#else
constexpr int kChunkFrags2 = 32;      // round 6: 32 fragments x 3 slots (a barrier every 64 MFMAs, the 8 pieces in the first half of a chunk): 3.52 -> 3.48 ms
#endif
#if ADN_OVERRIDABLE && defined(ADN_RS2)
constexpr int kRingSlots2 = ADN_RS2;
#else
constexpr int kRingSlots2 = 3;
#endif
// counted lgkmcnt wait in front of a tile's bias block instead of a full drain (layer_16x2)
#if ADN_OVERRIDABLE && defined(ADN_BIASWAIT)
constexpr bool kBiasWaitCounted = ADN_BIASWAIT != 0;
#else
constexpr bool kBiasWaitCounted = true;      // round 6: the request carries the re-fill addresses as operands (lds_bias_issue), one scheduling region per tile
#endif
// kShadeKstepFence: a scheduling barrier behind every k-step of layer_16x2 (the inline-asm conversions stay in their k-step)
#if ADN_OVERRIDABLE && defined(ADN_KSTEP_FENCE)
constexpr bool kShadeKstepFence = ADN_KSTEP_FENCE != 0;
#else
constexpr bool kShadeKstepFence = true;
#endif
// kShadeGuardStep: k-step of a tile in which the previous tile's mfma_guards are taken and its first conversions run (layer_16x2)
#if ADN_OVERRIDABLE && defined(ADN_GUARD_STEP)
constexpr int kShadeGuardStep = ADN_GUARD_STEP;
#else
constexpr int kShadeGuardStep = 2;      // round 6: 0 -> 1 -> 2: 3.357 -> 3.327 -> 3.285 ms (3: no further gain)
#endif
// kShadeCarry: the last tile of a shading layer is converted under the first tile of the next layer (layer_16x2, PendingTile2).
// (Requesting the next layer's first bias block a layer ahead, so that the wait at a layer boundary is counted too, was measured and dropped:
// the 16 bias registers live across the boundary next to the carried tile and both kernels spill -- 512 registers + scratch, sampling 1.23 -> 2.29 ms.)
#if ADN_OVERRIDABLE && defined(ADN_SHADE_CARRY)
constexpr bool kShadeCarry = ADN_SHADE_CARRY != 0;
#else
constexpr bool kShadeCarry = true;
#endif
// the k-step interleave pinned with sched_group_barrier (layer_16x2; bias blocks through compiler-visible LDS loads were measured slower in round 3 and are gone)
#if ADN_OVERRIDABLE && defined(ADN_SGB)
constexpr bool kSchedGroups = ADN_SGB != 0;
#else
constexpr bool kSchedGroups = true;
#endif
#if ADN_OVERRIDABLE && defined(ADN_SGB_S)
constexpr bool kSchedGroupsSampling = ADN_SGB_S != 0;
#else
constexpr bool kSchedGroupsSampling = true;      // round 6: 1.261 -> 1.219 ms on top of the re-worked ring (profiles/r06_lab_log.md); round 3 saw no effect
#endif
#if ADN_OVERRIDABLE && defined(ADN_SGB_S_VALU)
constexpr int kSgbValuSampling = ADN_SGB_S_VALU;
#else
constexpr int kSgbValuSampling = 4;
#endif
#if ADN_OVERRIDABLE && defined(ADN_NR2)
constexpr int kRegFrags2 = ADN_NR2;
#else
constexpr int kRegFrags2 = 16;
#endif

// Split engine, k-step details (k_sampling16.hip.hpp layer_16x3):
//  kSplitBiasCounted the wait in front of a tile's bias block (requested a tile earlier) is lgkmcnt(min(15, 2 KS)) instead of a full drain: the tile's
//                    own 2 KS fragment re-fills were issued behind the request (one scheduling region per tile: needs kSchedGroupsSampling; the request
//                    carries the re-fill addresses as operands) and LDS returns in order -- tests/test_host_cpu.py counts them on the assembly
//  kSplitCarry       the last output tile of a hidden layer is converted under the MFMAs of the NEXT layer's first tile (PendingTile3)
// (measured in round 6 and not kept as knobs: re-filling the lo' fragment register first so that one wait covers both -- no effect; the first k-step of a
// tile that carries a pair of the previous tile's epilogue -- 1, fixed; profiles/r06_variants_ring_2.log)
#if ADN_OVERRIDABLE && defined(ADN_KSTEP_FENCE_S)
constexpr bool kSplitKstepFence = ADN_KSTEP_FENCE_S != 0;      // a scheduling fence per k-step of layer_16x3 as in layer_16x2
#else
constexpr bool kSplitKstepFence = true;       // round 6: 1.247 -> 1.231 ms
#endif
#if ADN_OVERRIDABLE && defined(ADN_SPLIT_CARRY)
constexpr bool kSplitCarry = ADN_SPLIT_CARRY != 0;
#else
constexpr bool kSplitCarry = true;
#endif
#if ADN_OVERRIDABLE && defined(ADN_BIASWAIT_S)
constexpr bool kSplitBiasCounted = ADN_BIASWAIT_S != 0;
#else
constexpr bool kSplitBiasCounted = true;
#endif

// ---- selection (k_select_pair.hip.hpp, k_compact.hip.hpp)
// kSelScrubNaN: the sorted-list insertion of pair_select replaces NaN by -inf first (2 VALU per value); false: relies on v_max / v_med3 ignoring NaN
#if ADN_OVERRIDABLE && defined(ADN_SEL_SCRUB)
constexpr bool kSelScrubNaN = ADN_SEL_SCRUB != 0;
#else
constexpr bool kSelScrubNaN = false;
#endif
// Rays per workgroup of the wave-per-ray select_kernel (4 waves x kSelRaysPerBlock / 4 rays) = rays per segment total
#if ADN_OVERRIDABLE && defined(ADN_SEL_RPB)
constexpr int kSelRaysPerBlock = ADN_SEL_RPB;
#else
constexpr int kSelRaysPerBlock = 64;
#endif

// ---- timing ablations (results become WRONG; tools/ablate.sh, tables in profiles/*ablation*.md) ------------------------
//   1: no chunk boundary (no wait, no barrier, no DMA)   2: no LDS re-fill of the fragment registers
//   4: no bias read (acc starts at 0)                    8: no ReLU/convert epilogue
//  16: boundary without the DMA issue                   32: boundary without wait + barrier
//  64: (sampling kernel) no cross-tile software pipeline of bias reads / epilogue
// 128: (sampling kernels) v_sin_f32 instead of the libm-grade sincosf in the oracle-feature encoding   256: (sampling kernels) no encoding at all
// 512: (split sampling kernel) no selection epilogue   2048: (fused selection) the kept (bin, value) rows are not written (pair_emit skipped)
// kAblateShade applies to shade_mlp16_kernel, kAblateSample to sample_mlp16x3_kernel.
#if ADN_OVERRIDABLE && defined(ADN_ABLATE)
constexpr int kAblateShade = ADN_ABLATE;
#else
constexpr int kAblateShade = 0;
#endif
#if ADN_OVERRIDABLE && defined(ADN_ABLATE_S)
constexpr int kAblateSample = ADN_ABLATE_S;
#else
constexpr int kAblateSample = 0;
#endif


// kAblateDmaBytes: every LDS-DMA piece of the weight rings moves 4 instead of 16 bytes per lane (same instruction count, a quarter of the traffic)
#if ADN_OVERRIDABLE && defined(ADN_ABLATE_DMA_BYTES)
constexpr bool kAblateDmaBytes = ADN_ABLATE_DMA_BYTES != 0;
#else
constexpr bool kAblateDmaBytes = false;
#endif

// kAblateGeneric (run-time-shaped 16-bit kernels, k_generic16.hip.hpp; wrong results): 1 = the per-tile wait for the staged weights is skipped
// (barrier only) -- the time the kernel would take if the copies always arrived in time, i.e. the most a deeper prefetch can buy;
// 2 = no bias loads (accumulators start at 0); 3 = both
#if ADN_OVERRIDABLE && defined(ADN_ABLATE_G)
constexpr int kAblateGeneric = ADN_ABLATE_G;
#else
constexpr int kAblateGeneric = 0;
#endif

#undef ADN_OVERRIDABLE

}  // namespace tune
}  // namespace adanerf
