// SURVEY 8f N4, second half: the shading network in any exportable topology (NeRF trunk of 1..8 layers, width 64 / 128 / 256,
// one skip or none, src/models.py:199-277) and any encoding layout on the 16-bit MFMA pipe.
// The specialised 8 x 256 kernels stage one fragment stream through an LDS ring whose chunk positions are compile-time
// constants of that topology.  Here the layer table is a run-time argument, so every wave fetches its A fragments straight from
// global memory (all waves of the chip read the same <= 1.2 MB: L2 / L1 hits) in a loop over the hidden layers, with the layer's
// k-steps and tiles unrolled.  The fragment loads are ordinary compiler-visible loads, issued a tile ahead by the compiler.
// Device code only (gfx950, wave64); part of kernels.hip.hpp (main translation unit: accumulators in architectural VGPRs).
#pragma once
#include "k_generic_f32.hip.hpp"      // GenericTopo
#include "k_mlp16.hip.hpp"

namespace adanerf {

// One 16-bit layer, fragments [m][s][lane][8 x 16 bit] from global memory, bias blocks [m][h][16] fp32 from global memory.
// Input = two register segments of S1 / S2 k-steps (4 packed dwords each); KEEP_F32_TILE as layer_16.
template <class ET, int S1, int S2, int MT, bool RELU, int KEEP_F32_TILE = -1>
__device__ __forceinline__ void layer_16_direct(const u32x4* __restrict__ w, const float* __restrict__ bias, int lane, const uint32_t* in1,
                                                const uint32_t* in2, uint32_t* out, f32x16* keep = nullptr) {
  constexpr int KS = S1 + S2;
  const int h = lane >> 5;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    f32x16 acc;
    const float4* bp = reinterpret_cast<const float4*>(bias + (m * 2 + h) * 16);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b = bp[g];
      acc[4 * g + 0] = b.x;
      acc[4 * g + 1] = b.y;
      acc[4 * g + 2] = b.z;
      acc[4 * g + 3] = b.w;
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4 a = w[(m * KS + s) * 64 + lane];
      const uint32_t* src = (s < S1) ? (in1 + 4 * s) : (in2 + 4 * (s - S1));
      const u32x4 b = {src[0], src[1], src[2], src[3]};
      acc = ET::mfma(a, b, acc);
    }
    if (KEEP_F32_TILE == m) {
      *keep = acc;
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) epilogue_quad_16<ET, RELU>(acc, m, g, out);
    }
  }
}

// A5 + A6 for any shading-net topology on the 16-bit engine.  Workgroup = 4 waves x 32 samples (two workgroups per CU).
template <class ET, int FP, int FD, int W>
__global__ __launch_bounds__(256, 2) void shade_mlp16_gen_kernel(ShadeArgs a, GenericTopo t) {
  constexpr int QP = pe_slots(FP), QD = pe_slots(FD), MT = W / 32, KW = W / 16;      // KW: k-steps of a W-wide input
  constexpr int TILE = 4 * 32;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const int j = lane & 31, h = lane >> 5;
  int total = a.total ? *a.total : a.max_samples;
  if (total > a.max_samples) total = a.max_samples;
  const u32x4* w = a.net.w;
  const float* b = a.net.bias;
  for (int tile = blockIdx.x; tile * TILE < total; tile += gridDim.x) {
    const int s = tile * TILE + wave * 32 + j;
    if (tile * TILE + wave * 32 >= total) continue;      // wave-uniform
    asm volatile("" : "+v"(w), "+v"(b));                  // keep the fragment loads inside the loops (see shade_mlp32_kernel)
    float x[3], dpe[3];
    load_sample(a, s, total, x, dpe);
    uint32_t pts[QP / 2], dirs[QD / 2], hA[W / 4], hB[W / 4 + 8];      // hB also receives the (MT + 1)-tile feature (+ alpha) layer's packed part
    pe_pack<ET, FP>(x, h, pts);
    pe_pack<ET, FD>(dpe, h, dirs);
    layer_16_direct<ET, QP / 8, 0, MT, true>(w + a.net.w_off[0], b + a.net.b_off[0], lane, pts, pts, hA);
#pragma unroll 1
    for (int l = 1; l < t.depth; ++l) {
      asm volatile("" : "+v"(w), "+v"(b));
      if (l == t.skip + 1) layer_16_direct<ET, QP / 8, KW, MT, true>(w + a.net.w_off[l], b + a.net.b_off[l], lane, pts, hA, hB);      // cat([pts, h])
      else layer_16_direct<ET, KW, 0, MT, true>(w + a.net.w_off[l], b + a.net.b_off[l], lane, hA, hA, hB);
#pragma unroll
      for (int i = 0; i < W / 4; ++i) hA[i] = hB[i];
    }
    const int lf = t.depth;
    f32x16 alpha_tile;
    layer_16_direct<ET, KW, 0, MT + 1, false, MT>(w + a.net.w_off[lf], b + a.net.b_off[lf], lane, hA, hA, hB, &alpha_tile);        // feature (+ alpha row)
    layer_16_direct<ET, KW, QD / 8, MT / 2, true>(w + a.net.w_off[lf + 1], b + a.net.b_off[lf + 1], lane, hB, dirs, hA);          // cat([feature, dir])
    f32x16 rgb_tile;
    layer_16_direct<ET, KW / 2, 0, 1, false, 0>(w + a.net.w_off[lf + 2], b + a.net.b_off[lf + 2], lane, hA, hA, hB, &rgb_tile);
    if (h == 0 && s < total)
      *reinterpret_cast<float4*>(a.raw_out + static_cast<size_t>(s) * 4) = make_float4(rgb_tile[0], rgb_tile[1], rgb_tile[2], alpha_tile[0]);
  }
}

}  // namespace adanerf
