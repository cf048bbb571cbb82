// fp32 MFMA MLP engine (v_mfma_f32_32x32x2_f32, activations fp32 in registers), the exact-fp32 sampling kernel
// sample_mlp_kernel (A1+A2+A3) and the debug kernel ray_features_kernel.
// Device code only (gfx950, wave64); part of kernels.hip.hpp.
#pragma once
#include "k_common.hip.hpp"
#include "k_select_pair.hip.hpp"

namespace adanerf {

// ------------------------------------------------------------------------------------------
// fp32 MFMA MLP engine: v_mfma_f32_32x32x2_f32, activations fp32 in registers
// ------------------------------------------------------------------------------------------

// One layer for one 32-sample column block.  QS input slots (per lane-half), MT output tiles.
// Input = two register segments (Q1 then Q2 slots; a concatenation costs nothing).
template <int Q1, int Q2, int MT, bool RELU>
__device__ __forceinline__ void layer_f32(const u32x4* __restrict__ w, const float* __restrict__ bias, int lane,
                                          const float* in1, const float* in2, float* out) {
  constexpr int QS = Q1 + Q2;
  static_assert(Q1 % 4 == 0 && Q2 % 4 == 0, "fp32 engine groups 4 k-steps per 16-byte fragment");
  const int h = lane >> 5;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    f32x16 acc;
    const float4* bp = reinterpret_cast<const float4*>(bias + (m * 2 + h) * 16);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 b = bp[g];
      acc[4 * g + 0] = b.x;
      acc[4 * g + 1] = b.y;
      acc[4 * g + 2] = b.z;
      acc[4 * g + 3] = b.w;
    }
#pragma unroll
    for (int s4 = 0; s4 < QS / 4; ++s4) {
      // NB: load as a float vector.  __builtin_bit_cast(float, u32x4_value[i]) miscompiles on
      // ROCm 7.2 hipcc (every element reads lane register 0).
      const f32x4 a = reinterpret_cast<const f32x4*>(w)[(m * (QS / 4) + s4) * 64 + lane];
      const float* in = (4 * s4 < Q1) ? (in1 + 4 * s4) : (in2 + (4 * s4 - Q1));
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], in[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], in[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], in[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], in[3], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) out[16 * m + r] = RELU ? fmaxf(acc[r], 0.f) : acc[r];
  }
}

struct SampleArgs {
  RayGenParams g;
  NetParams net;         // fp32 fragments (exact engine)
  NetParams net16;       // fp16 hi/lo' fragment pairs (split-precision engine)
  int32_t* overflow_flag;
  int32_t first_ray, n_rays;
  float* oracle_out;     // [n_rays,128] or null
  float* rays_out;       // [n_rays,8] or null
  SelectOut sel;         // fused_select: adaptive selection in the kernel's epilogue (16-bit engines)
  int32_t fused_select;
  // pass 2 of the guarded selection (split engine): rays = ray_list[0 .. *n_list) instead of 0 .. n_rays (n_rays bounds the launch)
  const int32_t* ray_list;
  const int32_t* n_list;
};

// A1+A2+A3.  One wave = one block of 32 rays; 4 waves per workgroup (one per SIMD, up to 512 VGPRs).
template <int FP, int FD>
__global__ __launch_bounds__(256) void sample_mlp_kernel(SampleArgs a) {
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

  constexpr int QD = pe_slots(FD), QP = pe_slots(FP), Q0 = QD + QP;
  float bufA[128], bufB[128];
  pe_eval<FD, true>(u, h, bufA);          // [dir PE | pos PE]  (src/features.py:868-874)
  pe_eval<FP, true>(p, h, bufA + QD);

  const u32x4* w = a.net.w;
  const float* b = a.net.bias;
  layer_f32<Q0, 0, 8, true>(w + a.net.w_off[0], b + a.net.b_off[0], lane, bufA, bufA, bufB);
#pragma unroll 1
  for (int l = 1; l <= 5; l += 2) {
    layer_f32<128, 0, 8, true>(w + a.net.w_off[l], b + a.net.b_off[l], lane, bufB, bufB, bufA);
    layer_f32<128, 0, 8, true>(w + a.net.w_off[l + 1], b + a.net.b_off[l + 1], lane, bufA, bufA, bufB);
  }
  layer_f32<128, 0, 4, false>(w + a.net.w_off[7], b + a.net.b_off[7], lane, bufB, bufB, bufA);

  if (valid) {
    if (a.oracle_out) {
      float* o = a.oracle_out + static_cast<size_t>(local) * kBins;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v = make_float4(bufA[16 * m + 4 * g], bufA[16 * m + 4 * g + 1], bufA[16 * m + 4 * g + 2], bufA[16 * m + 4 * g + 3]);
          *reinterpret_cast<float4*>(o + 32 * m + 8 * g + 4 * h) = v;
        }
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

// Debug/parity: explicit oracle-net input features in the reference's column order.
// ray_samples > 0 (raySampleInput): A more blocks of 3 + 6 FP columns, the points p + d z_a encoded as the reference does
// (src/features.py:876-888: encode(x / d1), identity part scaled back by d1).
// FP / FD: bands of the position / direction encoding (run-time: any posEncArgs[0]).
static __global__ __launch_bounds__(256) void ray_features_kernel(RayGenParams g, int first_ray, int n_rays, float* feat, float* rays_out,
                                                           int ray_samples, const float* rsi_z, float rsi_d1, int FP, int FD) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rays) return;
  int col, row;
  ray_pixel(g, first_ray + i, &col, &row);
  float nds[3], p[3], u[3];
  gen_ray(g, col, row, nds, p);
  unit3(nds, u);
  if (feat) {
    const int ND = 3 + 6 * FD, NP = 3 + 6 * FP;
    float* f = feat + static_cast<size_t>(i) * (ND + NP + ray_samples * NP);
    for (int a = 0; a < ray_samples; ++a) {
      float* fa = f + ND + NP + a * NP;
      for (int c = 0; c < 3; ++c) {
        const float x = __fadd_rn(p[c], __fmul_rn(nds[c], rsi_z[a])) / rsi_d1;
        fa[c] = __fmul_rn(x, rsi_d1);
        for (int bnd = 0; bnd < FP; ++bnd) {
          float sn, co;
          sincosf(x * static_cast<float>(1 << bnd), &sn, &co);
          fa[3 + 6 * bnd + c] = sn;
          fa[3 + 6 * bnd + 3 + c] = co;
        }
      }
    }
    for (int c = 0; c < 3; ++c) {
      f[c] = u[c];
      f[ND + c] = p[c];
    }
    for (int b = 0; b < FD; ++b)
      for (int c = 0; c < 3; ++c) {
        float s, co;
        sincosf(u[c] * static_cast<float>(1 << b), &s, &co);
        f[3 + 6 * b + c] = s;
        f[3 + 6 * b + 3 + c] = co;
      }
    for (int b = 0; b < FP; ++b)
      for (int c = 0; c < 3; ++c) {
        float s, co;
        sincosf(p[c] * static_cast<float>(1 << b), &s, &co);
        f[ND + 3 + 6 * b + c] = s;
        f[ND + 3 + 6 * b + 3 + c] = co;
      }
  }
  if (rays_out) {
    float ro[3] = {p[0], p[1], p[2]}, rd[3] = {nds[0], nds[1], nds[2]};
    if (g.use_ndc) ndc_ray(g, p, nds, ro, rd);
    float4* r = reinterpret_cast<float4*>(rays_out + static_cast<size_t>(i) * 8);
    r[0] = make_float4(ro[0], ro[1], ro[2], 0.f);
    r[1] = make_float4(rd[0], rd[1], rd[2], 0.f);
  }
}

}  // namespace adanerf
