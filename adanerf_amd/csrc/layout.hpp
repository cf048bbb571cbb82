// Register-resident activation layout shared by the weight packer (host) and the MLP kernels
// (device).
//
// Both MLP engines compute the TRANSPOSED product  H'[feature][sample] = W[feature][k] * H[k][sample]
// with v_mfma_f32_32x32x{2_f32,16_bf16,16_f16}: A = weights (streamed, pre-packed per fragment),
// B = activations (registers), D = next activations (registers).  For every 32x32 D tile the MFMA
// leaves lane l = (j = l & 31 sample column, h = l >> 5 half) holding rows
//     (r & 3) + 8 * (r >> 2) + 4 * h,   r = 0..15
// so lane-half h owns, for output tile m, the 16 "slots" q = 16 m + r that stand for features
//     act_feature(q, h) = 32 (q >> 4) + 8 ((q & 15) >> 2) + 4 h + (q & 3).
// The next layer consumes slots in q order as its k index (K-permutation is free as long as the
// A operand is packed with the same permutation), so activations never leave registers and never
// need a cross-lane shuffle:
//   fp32 engine : k-step s uses slot s           (B = 1 VGPR,  A = W[row][col(s, h)])
//   16-bit eng. : k-step s uses slots 8s..8s+7   (B = 4 VGPRs, A = 8 packed values)
// Positional-encoding inputs use the same idea: lane-half 0 evaluates sin, lane-half 1 cos of the
// same (band, component), so each PE value is computed exactly once per sample.
#pragma once

#if defined(__HIPCC__)
#define ADN_HD __host__ __device__
#else
#define ADN_HD
#endif

namespace adanerf {

ADN_HD inline int act_feature(int q, int h) { return 32 * (q >> 4) + 8 * ((q & 15) >> 2) + 4 * h + (q & 3); }

// number of slots (per lane-half) a PE with F frequency bands occupies: 3F (sin|cos pairs) + 2
// identity slots, rounded up to a multiple of 8 (one 16-bit k-step)
ADN_HD constexpr int pe_slots(int F) { return ((3 * F + 2) + 7) & ~7; }

// source column (within the PE block [x, sin(2^0 x), cos(2^0 x), ...], 3-vectors interleaved as
// src/util/feature_encoding.py:54-73) of slot q for lane-half h; -1 = zero padding.
// FL >= F ("layout bands"): the slot layout of an FL-band encoding carrying only F bands -- bands F..FL-1 get no source column
// (zero weights), the identity slots sit at 3 FL.  The kernels are instantiated for a few layouts ((10,4), (2,2) and the
// catch-all kMaxBands); any other posEncArgs F <= kMaxBands is packed into the catch-all layout.
constexpr int kMaxBands = 16;
ADN_HD inline int pe_col(int F, int q, int h, int FL = 0) {
  if (FL <= 0) FL = F;
  if (q < 3 * FL) {
    int b = q / 3, c = q - 3 * b;
    return b < F ? 3 + 6 * b + 3 * h + c : -1;
  }
  if (q == 3 * FL) return h ? 2 : 0;      // x | z
  if (q == 3 * FL + 1) return h ? -1 : 1; // y | pad
  return -1;
}

constexpr int kBins = 128;   // multiDepthFeatures

}  // namespace adanerf
