// the two networks in any topology the reference's model classes can export (BaseNet / NeRF,
// src/models.py:18-82, 199-277: depth 2..8, width 64 / 128 / 256, trunk skips at any layers or none) and the
// raySampleInput oracle input (src/features.py:876-888; viewer: updateSpherePosDirBatchedUnrolledEnc with additional
// samples, adanerf_real_time_viewer/src/cuda/base_cuda_kernels.cu:99-145) -- on the exact-fp32 MFMA engine
// (v_mfma_f32_32x32x2_f32, k_mlp_f32.hip.hpp) with a run-time loop over the hidden layers.  Every shipped config is
// 8 x 256 / skip 4 and runs on the specialised kernels instead; these exist so that other exports load and render
// (at the fp32-MFMA rate, 1/16 of the 16-bit engines' peak, on networks that are several times smaller).
// Device code only (gfx950, wave64); compiled in launch_f32.hip (without -amdgpu-mfma-vgpr-form).
#pragma once
#include "k_mlp16.hip.hpp"
#include "k_mlp_f32.hip.hpp"

namespace adanerf {

struct GenericTopo {
  int32_t depth;        // Linear layers of the trunk (sampling net: all layers)
  int32_t cat_mask;     // shading trunk: bit l set <=> layer l takes cat([pts, h]) (l - 1 in the reference's `skips`, src/models.py:226-228, 260-261); 0 none
  int32_t ray_samples;  // sampling net: raySampleInput
  uint32_t rsi_w_off;   // 16-byte offset of the raySampleInput fragments of layer 0, [a][s4][m][lane]
  const float* rsi_z;   // [ray_samples] world depth of every extra point
  float rsi_d1;         // upper end of the warped depth range
};

// Layer 0 of a sampling net with raySampleInput: K = [dir PE | pos PE | A x PE(point a)].  The A points are walked in a
// run-time loop with all MT accumulators live (K-major fragments, pack.cpp emit_ray_samples); the first part is the usual
// tile-major block.  PE(point) = encode((p + d z_a) / d1) with the identity slots scaled back by d1, as the reference does.
template <int FP, int Q0, int MT>
__device__ __forceinline__ void layer0_ray_samples(const u32x4* __restrict__ w0, const u32x4* __restrict__ wr, const float* __restrict__ bias,
                                                   int lane, const float* in0, const float p[3], const float d[3], const GenericTopo& t,
                                                   float* out) {
  constexpr int QP = pe_slots(FP);
  const int h = lane >> 5;
  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const float4* bp = reinterpret_cast<const float4*>(bias + (m * 2 + h) * 16);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b = bp[g];
      acc[m][4 * g + 0] = b.x;
      acc[m][4 * g + 1] = b.y;
      acc[m][4 * g + 2] = b.z;
      acc[m][4 * g + 3] = b.w;
    }
#pragma unroll
    for (int s4 = 0; s4 < Q0 / 4; ++s4) {
      const f32x4 a = reinterpret_cast<const f32x4*>(w0)[(m * (Q0 / 4) + s4) * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], in0[4 * s4 + e], acc[m], 0, 0, 0);
    }
  }
#pragma unroll 1
  for (int a = 0; a < t.ray_samples; ++a) {
    const float z = t.rsi_z[a];
    float x[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) x[i] = __fadd_rn(p[i], __fmul_rn(d[i], z)) / t.rsi_d1;
    float tt[QP];
    pe_eval<FP, true>(x, h, tt);
    tt[3 * FP] = __fmul_rn(tt[3 * FP], t.rsi_d1);             // enc[..., :3] *= d1  (src/features.py:886)
    tt[3 * FP + 1] = __fmul_rn(tt[3 * FP + 1], t.rsi_d1);
    const f32x4* wa = reinterpret_cast<const f32x4*>(wr) + static_cast<size_t>(a) * (QP / 4) * MT * 64 + lane;
#pragma unroll
    for (int s4 = 0; s4 < QP / 4; ++s4)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const f32x4 f = wa[(s4 * MT + m) * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[e], tt[4 * s4 + e], acc[m], 0, 0, 0);
      }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[16 * m + r] = fmaxf(acc[m][r], 0.f);
}

// A1+A2+A3 for any sampling-net topology.  One wave = 32 rays, 4 waves per workgroup, like sample_mlp_kernel.
template <int FP, int FD, int W>
__global__ __launch_bounds__(256) void sample_mlp_gen_kernel(SampleArgs a, GenericTopo t) {
  constexpr int QD = pe_slots(FD), QP = pe_slots(FP), Q0 = QD + QP, QW = W / 2, MT = W / 32;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int blk = blockIdx.x * 4 + wave;
  if (blk * 32 >= a.n_rays) return;
  const int local = blk * 32 + j;
  const bool valid = local < a.n_rays;
  const int ray = a.first_ray + (valid ? local : a.n_rays - 1);
  int col, row;
  ray_pixel(a.g, ray, &col, &row);
  float nds[3], p[3], u[3];
  gen_ray(a.g, col, row, nds, p);
  unit3(nds, u);

  float in0[Q0], bufA[QW], bufB[QW];
  pe_eval<FD, true>(u, h, in0);          // [dir PE | pos PE]  (src/features.py:868-874)
  pe_eval<FP, true>(p, h, in0 + QD);
  const u32x4* w = a.net.w;
  const float* b = a.net.bias;
  if (t.ray_samples > 0)
    layer0_ray_samples<FP, Q0, MT>(w + a.net.w_off[0], w + t.rsi_w_off, b + a.net.b_off[0], lane, in0, p, nds, t, bufB);
  else
    layer_f32<Q0, 0, MT, true>(w + a.net.w_off[0], b + a.net.b_off[0], lane, in0, in0, bufB);
#pragma unroll 1
  for (int l = 1; l + 1 < t.depth; ++l) {
    asm volatile("" : "+v"(w), "+v"(b));   // keep the fragment loads inside the loop (see shade_mlp32_kernel)
    layer_f32<QW, 0, MT, true>(w + a.net.w_off[l], b + a.net.b_off[l], lane, bufB, bufB, bufA);
#pragma unroll
    for (int i = 0; i < QW; ++i) bufB[i] = bufA[i];
  }
  float out[64];
  layer_f32<QW, 0, 4, false>(w + a.net.w_off[t.depth - 1], b + a.net.b_off[t.depth - 1], lane, bufB, bufB, out);

  if (valid) {
    if (a.oracle_out) {
      float* o = a.oracle_out + static_cast<size_t>(local) * kBins;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(o + 32 * m + 8 * g + 4 * h) =
              make_float4(out[16 * m + 4 * g], out[16 * m + 4 * g + 1], out[16 * m + 4 * g + 2], out[16 * m + 4 * g + 3]);
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

// A5+A6 for any shading-net topology (NeRF trunk of t.depth layers of width W, skip after layer t.skip, then
// feature (+ alpha row), views, rgb).
template <int FP, int FD, int W>
__global__ __launch_bounds__(256) void shade_mlp32_gen_kernel(ShadeArgs a, GenericTopo t) {
  constexpr int QP = pe_slots(FP), QD = pe_slots(FD), QW = W / 2, MT = W / 32;
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
    if (tile * TILE + wave * 32 >= total) continue;
    asm volatile("" : "+v"(w), "+v"(b));
    float x[3], dpe[3];
    load_sample(a, s, total, x, dpe);
    float pts[QP], dirs[QD], hA[QW], hB[QW + 16];      // hB also receives the (MT + 1)-tile feature (+ alpha) layer
    pe_eval<FP, true>(x, h, pts);
    pe_eval<FD, true>(dpe, h, dirs);
    layer_f32<QP, 0, MT, true>(w + a.net.w_off[0], b + a.net.b_off[0], lane, pts, pts, hA);
#pragma unroll 1
    for (int l = 1; l < t.depth; ++l) {
      asm volatile("" : "+v"(w), "+v"(b));
      if ((t.cat_mask >> l) & 1) layer_f32<QP, QW, MT, true>(w + a.net.w_off[l], b + a.net.b_off[l], lane, pts, hA, hB);      // cat([pts, h])
      else layer_f32<QW, 0, MT, true>(w + a.net.w_off[l], b + a.net.b_off[l], lane, hA, hA, hB);
#pragma unroll
      for (int i = 0; i < QW; ++i) hA[i] = hB[i];
    }
    const int lf = t.depth;
    layer_f32<QW, 0, MT + 1, false>(w + a.net.w_off[lf], b + a.net.b_off[lf], lane, hA, hA, hB);        // feature (+ alpha row)
    const float alpha = hB[QW];
    layer_f32<QW, QD, MT / 2, true>(w + a.net.w_off[lf + 1], b + a.net.b_off[lf + 1], lane, hB, dirs, hA);   // cat([feature, dir])
    float rgb[16];
    layer_f32<QW / 2, 0, 1, false>(w + a.net.w_off[lf + 2], b + a.net.b_off[lf + 2], lane, hA, hA, rgb);
    if (h == 0 && s < total)
      *reinterpret_cast<float4*>(a.raw_out + static_cast<size_t>(s) * 4) = make_float4(rgb[0], rgb[1], rgb[2], alpha);
  }
}

}  // namespace adanerf
