// Build-time constants of the MLP engines, in one place.  Part 1: what the shipped kernels are built with -- every line is a closed
// experiment (the log that decided it is named; profiles/rNN_lab_log.md).  Part 2: the few constants tools/ablate.sh may still override
// (-DADN_EXPERIMENT -DADN_<NAME>=<value>): ring geometry of the two one-wave-per-SIMD kernels and the timing ablations behind the
// per-lever tables.  An ADN_<NAME> override without ADN_EXPERIMENT is an error, not ignored.
#pragma once

namespace adanerf {
namespace tune {

// ---- Part 1: fixed ----------------------------------------------------------------------------------------------------------
// LDS weight ring of the 8-wave kernels (shade_mlp16_kernel, sample_mlp16_kernel; k_mlp16.hip.hpp).  CF = fragments (KiB) per chunk =
// MFMAs per wave between two synchronisation points, RS = ring slots, NR = fragments a wave holds in registers (LDS prefetch distance).
constexpr int kChunkFrags = 16, kRingSlots = 4;      // r02 ring sweeps
constexpr int kRegFrags = 4;                         // 2: 3.71-3.82 ms, 8: 3.66-3.76 with 4 spilled registers, 4: 3.58-3.62 (256-register cap)
constexpr bool kStagger = true;                      // waves 4-7 synchronise half a chunk after waves 0-3 (ws_sync)
constexpr int kDmaGroup = 0;                         // 8-wave kernels: -1 / 0 / 1 = every wave / only waves 0-3 / only waves 4-7 copy the weight pieces
constexpr int kSkipPad = 11;                         // s_nop argument on the short side of a branch behind a tile's last MFMA (profiles/r02_stagger_hazard.md)
constexpr int kShadeBlocks = 2;                      // 2 = shade_mlp16x2_kernel (4 waves x 64 samples; profiles/r03_shade_two_blocks.md), 1 = shade_mlp16_kernel
// plain-fp16 sampling pass (speed mode, first pass of the guarded mode): encoding through one v_sin_f32 per slot -- its values are rounded to
// fp16 right after, the guard band is calibrated with whatever this kernel computes (profiles/r03_variants_fast_pe.log: 1.014 -> 0.980 ms)
constexpr bool kFastPeFp16Pass = true;
// layer_16x2 / layer_16x3 (round 6, profiles/r06_lab_log.md section 1.2; each was one A/B in one session):
constexpr bool kSchedGroups = true, kSchedGroupsSampling = true;      // k-step interleave pinned with sched_group_barrier (sampling: 1.261 -> 1.219 ms)
constexpr int kSgbValuSampling = 4;                                   // VALU per group (3 / 5: within 0.3 %, r06_variants_ring_2.log)
constexpr bool kBiasWaitCounted = true, kSplitBiasCounted = true;     // counted lgkmcnt in front of a tile's bias block: the request carries the re-fill
                                                                      // addresses as operands (lds_bias_issue); tests/test_host_cpu.py counts the younger reads
constexpr bool kShadeCarry = true, kSplitCarry = true;                // a layer's last tile is converted under the next layer's first (PendingTile2 / 3)
constexpr bool kShadeKstepFence = true, kSplitKstepFence = true;      // sched_barrier(0) behind every k-step: pins the inline-asm conversions
constexpr int kShadeGuardStep = 2;                                    // k-step that takes the previous tile's mfma_guards: 0 -> 1 -> 2: 3.357 -> 3.327 -> 3.285 ms
// run-time-shaped 16-bit kernels (k_generic16.hip.hpp; profiles/r03_generic_staged.md, r04_lab_log.md)
constexpr bool kGenericStaged = true;                // weights of an output tile staged through LDS once per workgroup (false: layer_16x3_direct, the baseline)
constexpr int kGenericBlocks128 = 2, kGenericBlocks256 = 2;           // 32-sample blocks per wave of the staged shading kernel at that width
constexpr int kGenericOcc64 = 2, kGenericOcc128 = 2, kGenericOcc256 = 1;      // workgroups per CU asked of the compiler
constexpr int kGenericSpread = 1;                    // a tile's conversions spread over the next tile's k-steps: 1 = width 256 only (r04_variants_generic_spread.log)
constexpr bool kGenericBiasDirect = true;            // a tile's bias block read from the LDS table straight into its accumulators
constexpr int kGenericAhead = 4;                     // fragments a wave requests from LDS ahead of the k-step that consumes them
// selection: rays per workgroup of the wave-per-ray select_kernel (4 waves x kSelRaysPerBlock / 4 rays) = rays per segment total
constexpr int kSelRaysPerBlock = 64;

// ---- Part 2: overridable in experiment builds -------------------------------------------------------------------------------
#if !defined(ADN_EXPERIMENT) && (defined(ADN_CF_S) || defined(ADN_RS_S) || defined(ADN_NR_S) || defined(ADN_CF2) || defined(ADN_RS2) || defined(ADN_NR2) || \
                                 defined(ADN_ABLATE) || defined(ADN_ABLATE_S) || defined(ADN_ABLATE_G) || defined(ADN_ABLATE_DMA_BYTES))
#error "ADN_<NAME> overrides are for experiment builds: add -DADN_EXPERIMENT (tools/ablate.sh does)"
#endif
// split-precision sampling kernel: 32 fragments x 3 slots (a barrier every 48 MFMAs, pieces issued in the first half of a chunk: 1.228 -> 1.218 ms),
// a whole half chunk of fragments in registers (4: 1.42, 8: 1.37, 16: 1.32 ms)
#ifndef ADN_CF_S
#define ADN_CF_S 32
#endif
#ifndef ADN_RS_S
#define ADN_RS_S 3
#endif
#ifndef ADN_NR_S
#define ADN_NR_S 16
#endif
// shade_mlp16x2_kernel: 32 fragments x 3 slots (a barrier every 64 MFMAs, the 8 pieces in the first half of a chunk: 3.52 -> 3.48 ms)
#ifndef ADN_CF2
#define ADN_CF2 32
#endif
#ifndef ADN_RS2
#define ADN_RS2 3
#endif
#ifndef ADN_NR2
#define ADN_NR2 16
#endif
constexpr int kChunkFragsSampling = ADN_CF_S, kRingSlotsSampling = ADN_RS_S, kRegFragsSampling = ADN_NR_S;
constexpr int kChunkFrags2 = ADN_CF2, kRingSlots2 = ADN_RS2, kRegFrags2 = ADN_NR2;

// Timing ablations (results become WRONG; tables: profiles/r06_lab_log.md sections 1.3 / 1.4).  kAblateShade: the shading kernels, kAblateSample: the sampling kernels.
//   1: no chunk boundary (no wait, no barrier, no DMA)   2: no LDS re-fill of the fragment registers
//   4: no bias read (acc starts at 0)                    8: no ReLU/convert epilogue
//  16: boundary without the DMA issue                   32: boundary without wait + barrier
//  64: (sampling) no cross-tile software pipeline of bias reads / epilogue
// 128: (sampling) v_sin_f32 instead of the libm-grade sincosf in the oracle-feature encoding   256: (sampling) no encoding at all
// 512: (split sampling kernel) no selection epilogue   2048: (fused selection) the kept (bin, value) rows are not written (pair_emit skipped)
#ifndef ADN_ABLATE
#define ADN_ABLATE 0
#endif
#ifndef ADN_ABLATE_S
#define ADN_ABLATE_S 0
#endif
// run-time-shaped 16-bit kernels: 1 = the per-tile wait for the staged weights is skipped (barrier only: the most a deeper prefetch can buy); 2 = no bias loads
#ifndef ADN_ABLATE_G
#define ADN_ABLATE_G 0
#endif
// every LDS-DMA piece of the weight rings moves 4 instead of 16 bytes per lane (same instruction count, a quarter of the traffic)
#ifndef ADN_ABLATE_DMA_BYTES
#define ADN_ABLATE_DMA_BYTES 0
#endif
constexpr int kAblateShade = ADN_ABLATE, kAblateSample = ADN_ABLATE_S, kAblateGeneric = ADN_ABLATE_G;
constexpr bool kAblateDmaBytes = ADN_ABLATE_DMA_BYTES != 0;

}  // namespace tune
}  // namespace adanerf
