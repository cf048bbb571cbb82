// gfx950 (CDNA4, wave64) kernels of the AdaNeRF per-frame hot path.  Device code only; the C ABI
// in adanerf_hip.hip launches these.  Stage map:
//   A1+A2+A3  sample_mlp_kernel      ray gen -> sphere exit -> oracle PE -> 8-layer sampling MLP (fp32 MFMA)
//   A4        select_kernel / scan_blocks_kernel / expand_kernel   top-N + threshold, deterministic compaction
//   A5+A6     shade_mlp16_kernel / shade_mlp32_kernel   fused PE + 8x256 shading MLP (bf16/f16/f32 MFMA)
//   A7        composite_kernel       sigmoid + alpha * oracle weight, front-to-back
// plus explicit-feature debug kernels (ray_features_kernel, shade_features_kernel) that materialise
// what the reference launchers wrote to memory, for parity tests only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.hpp"
#include "pack.hpp"

namespace adanerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

constexpr int kMaxLayers = 12;

struct NetParams {
  const u32x4* w;            // packed A fragments (16 B each)
  const float* bias;         // packed bias blocks
  uint32_t w_off[kMaxLayers];
  uint32_t b_off[kMaxLayers];
};

// Everything ray generation needs (A1 + A2).  Doubles mirror the float64 numpy ray table of
// src/util/raygeneration.py:10-26.
struct RayGenParams {
  double start_x, x_pp, start_y, y_pp, focal;
  int32_t w, h;
  int32_t strip_rows, world, rank;     // round-robin strip sharding of image rows
  int32_t use_ndc;
  float rot[9];                        // row-major c2w
  float pos[3];
  float center[3];
  float rad2;                          // ||view_cell_size/2||^2
  float ndc_sw, ndc_sh;                // -1/(W/(2 focal)), -1/(H/(2 focal))
};

struct ShadeParams {
  float center[3];
  float inv_sqrt_max_depth_unused;
  float sqrt_max_depth;
  int32_t normalize;                   // 1: InverseSqrtDistCentered, 0: None
  int32_t unit_dir;                    // 1: PE(dir/|dir|) (NDC), 0: PE(dir) as received
  const float* ztab;                   // [128] world depth per bin
};

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & 63; }

// local ray index -> (col, row) under round-robin row-strip sharding
__device__ __forceinline__ void ray_pixel(const RayGenParams& g, int i, int* col, int* row) {
  const int per_strip = g.w * g.strip_rows;
  const int sl = i / per_strip;
  const int within = i - sl * per_strip;
  const int r = within / g.w;
  *col = within - r * g.w;
  *row = (sl * g.world + g.rank) * g.strip_rows + r;
}

// A1: camera-space unit direction (float64 math, cast to float32), then A2: world dir + sphere exit.
// Follows src/util/raygeneration.py:10-26 and src/features.py:769-791, 845-866.
__device__ __forceinline__ void gen_ray(const RayGenParams& g, int col, int row, float nds[3], float p[3]) {
  double vx = __dadd_rn(g.start_x, __dmul_rn(g.x_pp, static_cast<double>(col)));
  double vy = __dadd_rn(g.start_y, __dmul_rn(g.y_pp, static_cast<double>(row)));
  double vz = g.focal;
  double n = sqrt(__dadd_rn(__dadd_rn(__dmul_rn(vx, vx), __dmul_rn(vy, vy)), __dmul_rn(vz, vz)));
  float dx = static_cast<float>(vx / n);
  float dy = static_cast<float>(-(vy / n));
  float dz = static_cast<float>(-(vz / n));
#pragma unroll
  for (int i = 0; i < 3; ++i)
    nds[i] = __fadd_rn(__fadd_rn(__fmul_rn(g.rot[3 * i], dx), __fmul_rn(g.rot[3 * i + 1], dy)), __fmul_rn(g.rot[3 * i + 2], dz));
  float q[3] = {g.pos[0] - g.center[0], g.pos[1] - g.center[1], g.pos[2] - g.center[2]};
  float udot = __fadd_rn(__fadd_rn(__fmul_rn(q[0], nds[0]), __fmul_rn(q[1], nds[1])), __fmul_rn(q[2], nds[2]));
  float qq = __fadd_rn(__fadd_rn(__fmul_rn(q[0], q[0]), __fmul_rn(q[1], q[1])), __fmul_rn(q[2], q[2]));
  float delta = __fsub_rn(__fmul_rn(udot, udot), __fsub_rn(qq, g.rad2));
  float dist = __fadd_rn(-udot, sqrtf(fmaxf(delta, 0.f)));
#pragma unroll
  for (int i = 0; i < 3; ++i) p[i] = __fadd_rn(g.pos[i], __fmul_rn(nds[i], dist));
}

__device__ __forceinline__ void unit3(const float v[3], float out[3]) {
  float n = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(v[0], v[0]), __fmul_rn(v[1], v[1])), __fmul_rn(v[2], v[2])));
  out[0] = v[0] / n;
  out[1] = v[1] / n;
  out[2] = v[2] / n;
}

// src/nerf_raymarch_common.py:71-88 (near = 1)
__device__ __forceinline__ void ndc_ray(const RayGenParams& g, const float o[3], const float d[3], float on[3], float dn[3]) {
  const float near = 1.0f;
  float t = -(near + o[2]) / d[2];
  float ox = __fadd_rn(o[0], __fmul_rn(t, d[0])), oy = __fadd_rn(o[1], __fmul_rn(t, d[1])), oz = __fadd_rn(o[2], __fmul_rn(t, d[2]));
  on[0] = g.ndc_sw * ox / oz;
  on[1] = g.ndc_sh * oy / oz;
  on[2] = 1.0f + 2.0f * near / oz;
  dn[0] = g.ndc_sw * (d[0] / d[2] - ox / oz);
  dn[1] = g.ndc_sh * (d[1] / d[2] - oy / oz);
  dn[2] = -2.0f * near / oz;
}

// sin(a) (h = 0) or cos(a) (h = 1) at libm accuracy (<= 1.6 ulp, max abs error 9.2e-8 for |a| < 1e5, checked against
// fp64 on 2e7 arguments): three-term FMA Cody-Waite reduction by pi/2, degree-7 / degree-8 minimax polynomials on
// [-pi/4, pi/4] (Cephes sinf/cosf coefficients); cos(a) = sin(a + pi/2) is applied to the integer quadrant, so it is
// exact.  ~25 VALU instructions; the device libm's sincosf (Payne-Hanek capable, both results) costs ~5x that, which
// was 0.14 ms per frame in the sampling kernel.
__device__ __forceinline__ float sin_or_cos(float a, int h) {
  float r;
  int n;
  if (__builtin_expect(fabsf(a) < 1.0e5f, 1)) {
    const float j = __builtin_rintf(a * 0.636619747f);             // a * 2/pi
    r = __builtin_fmaf(j, -1.57079601e+00f, a);                    // pi/2 = 1.57079601 + 3.13916473e-7 + 5.39030253e-15
    r = __builtin_fmaf(j, -3.13916473e-07f, r);
    r = __builtin_fmaf(j, -5.39030253e-15f, r);
    n = static_cast<int>(j) + h;
  } else {
    // rare: the same reduction in fp64 (two-term pi/2), exact to ~1e-16 while the quotient fits a double's integers
    // (|a| < ~1e15).  Beyond that the argument's own fp32 spacing spans > 1e7 periods and the value carries no
    // information: the reduced argument is clamped so the result stays in [-1, 1], but it is not libm's value.
    // inf/NaN -> NaN like libm.
    const double ad = static_cast<double>(a);
    const double k = __builtin_rint(ad * 0.6366197723675814);
    double rd = __builtin_fma(k, -1.5707963267948966, ad);
    rd = __builtin_fma(k, -6.123233995736766e-17, rd);
    rd = __builtin_fmin(__builtin_fmax(rd, -0.7853981633974483), 0.7853981633974483);   // NaN stays NaN: see below
    r = (a != a || fabsf(a) == INFINITY) ? __builtin_nanf("") : static_cast<float>(rd);
    n = static_cast<int>(k - 4.0 * __builtin_floor(k * 0.25)) + h;
  }
  const float s = r * r;
  float t = __builtin_fmaf(s, -1.9515295891e-4f, 8.3321608736e-3f);
  t = __builtin_fmaf(t, s, -1.6666654611e-1f);
  const float ps = __builtin_fmaf(t * s, r, r);
  float u = __builtin_fmaf(s, 2.443315711809948e-5f, -1.388731625493765e-3f);
  u = __builtin_fmaf(u, s, 4.166664568298827e-2f);
  const float pc = __builtin_fmaf(u, s * s, __builtin_fmaf(s, -0.5f, 1.0f));
  const float v = (n & 1) ? pc : ps;
  return (n & 2) ? -v : v;
}

// PE slots of lane-half h (layout.hpp): slot q < 3F -> h ? cos : sin of 2^(q/3) * x[q%3];
// then two identity slots.  ACCURATE: libm-grade sin_or_cos (fp32 parity path);
// !ACCURATE: one v_sin_f32 per slot (cos = sin shifted by a quarter revolution).
template <int F, bool ACCURATE>
__device__ __forceinline__ void pe_eval(const float x[3], int h, float* out) {
#pragma unroll
  for (int q = 0; q < 3 * F; ++q) {
    const int b = q / 3, c = q - 3 * b;
    const float a = x[c] * static_cast<float>(1 << b);
    if (ACCURATE) {
      out[q] = sin_or_cos(a, h);
    } else {
      out[q] = __builtin_amdgcn_sinf(__builtin_fmaf(a, 0.15915494309189535f, h ? 0.25f : 0.0f));
    }
  }
  out[3 * F] = h ? x[2] : x[0];
  out[3 * F + 1] = h ? 0.f : x[1];
#pragma unroll
  for (int q = 3 * F + 2; q < pe_slots(F); ++q) out[q] = 0.f;
}

// A5: sample position + normalisation (src/features.py:458-467, src/nerf_raymarch_common.py:226-230)
__device__ __forceinline__ void sample_position(const ShadeParams& sp, const float o[3], const float d[3], float z, float x[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) x[i] = __fadd_rn(o[i], __fmul_rn(d[i], z));
  if (sp.normalize) {
    float l[3] = {x[0] - sp.center[0], x[1] - sp.center[1], x[2] - sp.center[2]};
    float n2 = __fadd_rn(__fadd_rn(__fmul_rn(l[0], l[0]), __fmul_rn(l[1], l[1])), __fmul_rn(l[2], l[2]));
    float local = sqrtf(sqrtf(n2));
    float den = __fmul_rn(sp.sqrt_max_depth, local);
    x[0] = l[0] / den;
    x[1] = l[1] / den;
    x[2] = l[2] / den;
  }
}

// wave64 max of a float (every lane gets the result): 4 in-row DPP butterflies, then 4 readlanes
__device__ __forceinline__ float wave_max_f32(float v) {
  int x = __builtin_bit_cast(int, v);
#define ADN_DPP_MAX(ctrl)                                                                         \
  {                                                                                                \
    int y = __builtin_amdgcn_update_dpp(x, x, ctrl, 0xF, 0xF, false);                               \
    x = __builtin_bit_cast(int, fmaxf(__builtin_bit_cast(float, x), __builtin_bit_cast(float, y))); \
  }
  ADN_DPP_MAX(0xB1)    // quad_perm [1,0,3,2]
  ADN_DPP_MAX(0x4E)    // quad_perm [2,3,0,1]
  ADN_DPP_MAX(0x141)   // row_half_mirror
  ADN_DPP_MAX(0x140)   // row_mirror
#undef ADN_DPP_MAX
  float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 0));
  float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 16));
  float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 32));
  float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

__device__ __forceinline__ float sigmoidf_dev(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ int mbcnt64(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0));
}

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
template <int FP, int FD>
__global__ __launch_bounds__(256) void ray_features_kernel(RayGenParams g, int first_ray, int n_rays, float* feat, float* rays_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rays) return;
  int col, row;
  ray_pixel(g, first_ray + i, &col, &row);
  float nds[3], p[3], u[3];
  gen_ray(g, col, row, nds, p);
  unit3(nds, u);
  if (feat) {
    constexpr int ND = 3 + 6 * FD, NP = 3 + 6 * FP;
    float* f = feat + static_cast<size_t>(i) * (ND + NP);
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

// ------------------------------------------------------------------------------------------
// A4: adaptive selection + deterministic compaction
// ------------------------------------------------------------------------------------------

// Rays per workgroup of select_kernel (4 waves x kSelRaysPerBlock/4 rays, one ray at a time per wave) = rays per
// entry of the block-total scan.  Small on purpose: a wave's serial loop over its rays is the critical path of a
// small batch (an 83 200-ray shard of an 8-GPU frame), and more, shorter waves also schedule better on a whole
// frame (measured 0.207 ms at 256 rays, 0.167 ms at 64 for 640 000 rays).
#ifndef ADN_SEL_RPB
#define ADN_SEL_RPB 64
#endif
constexpr int kSelRaysPerBlock = ADN_SEL_RPB;
static_assert(kSelRaysPerBlock == 16 || kSelRaysPerBlock == 32 || kSelRaysPerBlock == 64, "segment must fit one wave");

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

__global__ __launch_bounds__(256) void select_kernel(const float* __restrict__ oracle, int n_rays, int n_max, float thr,
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
// One thread per ray; the rays of one select_kernel workgroup are one wave segment, so the in-segment prefix is a
// width-limited shuffle scan (no LDS, no barrier).
__global__ __launch_bounds__(256) void expand_kernel(const int32_t* __restrict__ counts, const uint8_t* __restrict__ selbin,
                                                     const float* __restrict__ selw, const int32_t* __restrict__ block_offset,
                                                     int n_rays, int n_max, int32_t* __restrict__ ray_offsets,
                                                     uint32_t* __restrict__ sample_key, float* __restrict__ sample_w) {
  const int r = blockIdx.x * 256 + static_cast<int>(threadIdx.x);
  const int c = (r < n_rays) ? counts[r] : 0;
  const int seg_lane = r & (kSelRaysPerBlock - 1);
  int x = c;
#pragma unroll
  for (int off = 1; off < kSelRaysPerBlock; off <<= 1) {
    const int y = __shfl_up(x, off, kSelRaysPerBlock);
    if (seg_lane >= off) x += y;
  }
  if (r >= n_rays) return;
  const int o = block_offset[r / kSelRaysPerBlock] + x - c;
  ray_offsets[r] = o;
  const size_t src = static_cast<size_t>(r) * n_max;
  for (int k = 0; k < c; ++k) {
    sample_key[o + k] = (static_cast<uint32_t>(r) << 7) | selbin[src + k];
    sample_w[o + k] = selw[src + k];
  }
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

// thr == 0: every bin of every ray (src/nerf_raymarch_common.py:708-720); keys are implicit
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

// ------------------------------------------------------------------------------------------
// A5 + A6: fused PE + shading MLP
// ------------------------------------------------------------------------------------------

struct ShadeArgs {
  ShadeParams sp;
  NetParams net;
  const float* rays;          // [*,8]
  const uint32_t* sample_key; // [S]
  const float* sample_z;      // [S] world depth per sample (inverse-CDF sampler); null -> ztab[bin]
  const int32_t* total;       // device S (may be null -> max_samples)
  int32_t max_samples;
  float* raw_out;             // [S,4]
};

struct Bf16 {
  typedef bf16x8 vec8;
  static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  }
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
struct Fp16 {
  typedef f16x8 vec8;
  static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
  }
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

// ---- weight streaming through LDS -------------------------------------------------------------
// The packed 16-bit shading net is one linear stream of 1 KiB A fragments in consumption order
// (layer, tile m, k-step s).  All waves of a workgroup consume it in lockstep, so it is staged ONCE
// per workgroup: 8 KiB chunks (8 fragments; each of the 8 waves DMA-copies one fragment with
// global_load_lds_dwordx4, LDS image lane-linear = fragment order, so ds_read_b128 is conflict
// free) into a 4-slot ring.  One s_barrier per chunk; counted vmcnt keeps 2-3 chunks in flight
// across the barrier (never vmcnt(0) in the loop).  Each wave keeps the current chunk's 8
// fragments in registers and re-fills fragment i from the NEXT chunk right after the MFMA that
// consumed it, so LDS latency hides behind the other 7 MFMAs.
// Timing-ablation switches for tools/ablate.sh (results become WRONG; never defined in the shipped build):
//   1: no chunk boundary (no wait, no barrier, no DMA)   2: no LDS re-fill of the fragment registers
//   4: no bias read (acc starts at 0)                    8: no ReLU/convert epilogue
//  16: boundary without the DMA issue                   32: boundary without wait + barrier
//  64: (sampling kernel) no cross-tile software pipeline of bias reads / epilogue
// 128: (sampling kernels) v_sin_f32 instead of the libm-grade sincosf in the oracle-feature encoding
// ADN_ABLATE applies to shade_mlp16_kernel, ADN_ABLATE_S to sample_mlp16x3_kernel.
#ifndef ADN_ABLATE
#define ADN_ABLATE 0
#endif
#ifndef ADN_ABLATE_S
#define ADN_ABLATE_S 0
#endif
// Ring geometry.  CF = fragments (KiB) per chunk = MFMAs per wave between barriers; RS = ring slots.
// At boundary k a wave waits for its own pieces of chunk k+1, so RS-3 further chunks stay in flight.
#ifndef ADN_CF
#define ADN_CF 16
#endif
#ifndef ADN_RS
#define ADN_RS 4
#endif
#ifndef ADN_CF_S
#define ADN_CF_S 16
#endif
#ifndef ADN_RS_S
#define ADN_RS_S 6
#endif
constexpr int kRegFrags = 4;     // fragments held in registers per wave (re-fill distance in MFMAs)
constexpr int kShadeFrags16 = 32 + 4 * 128 + 160 + 2 * 128 + 144 + 72 + 8;   // 1184 per pass (FP=10, FD=4)
constexpr int kShadeBiasFloats = 8 * 256 + 288 + 128 + 32;                     // 2496

// CF / RS / LPW (fragments each wave DMA-copies per chunk = CF / waves) are compile-time; the slot a
// chunk lives in is a run-time counter, so any tile length that is a multiple of CF works.
template <int CF, int RS, int LPW>
struct WStream {
  static constexpr int kChunkBytes = CF * 1024;
  const char* gbase;     // stream start (global)
  uint32_t gbytes;       // stream length in bytes (multiple of kChunkBytes)
  uint32_t goff;         // byte offset of the next chunk to issue
  uint32_t lane_off;     // lane * 16
  uint32_t wave_off;     // byte offset of this wave's first fragment inside a chunk
  uint32_t slot_cur;     // ring slot of the chunk being consumed (wave-uniform)
  uint32_t rd_cur;       // LDS byte address of (current chunk, this lane)
  uint32_t rd_next;      // LDS byte address of (next chunk, this lane)
  uint32_t lds_base;     // LDS byte address of the ring
  u32x4 R[kRegFrags];    // register ring: fragment p (position inside the chunk) lives in R[p % kRegFrags]
};

__device__ __forceinline__ u32x4 lds_read128(uint32_t byte_addr) {
  typedef const __attribute__((address_space(3))) u32x4* lds_u32x4_ptr;
  return *((lds_u32x4_ptr)(uintptr_t)byte_addr);
}

template <int CF, int RS, int LPW>
__device__ __forceinline__ void ws_issue(WStream<CF, RS, LPW>& st, uint32_t slot) {
#pragma unroll
  for (int i = 0; i < LPW; ++i) {
    const char* src = st.gbase + st.goff + st.wave_off + i * 1024 + st.lane_off;
    const uint32_t dst = st.lds_base + slot * (CF * 1024) + st.wave_off + i * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)static_cast<uintptr_t>(dst), 16, 0, 0);
  }
  st.goff += CF * 1024;
  if (st.goff >= st.gbytes) st.goff = 0;
}

// chunk boundary k: own pieces of chunk k+1 have landed (<= (RS-3) LPW younger DMAs outstanding); barrier =>
// chunk k+1 complete in LDS for every wave and every wave has consumed chunk k-1 (its MFMAs were
// issued before the barrier, so its ds_reads returned) => refill the slot of chunk k-1 with chunk k+RS-1.
template <int ABL, int CF, int RS, int LPW>
__device__ __forceinline__ void ws_boundary(WStream<CF, RS, LPW>& st) {
  static_assert(RS >= 3 && (RS - 2) * LPW < 64, "vmcnt is a 6-bit counter");
  if (ABL & 1) return;
  if (!(ABL & 32)) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((RS - 3) * LPW) : "memory");   // 32: no wait/barrier
  const uint32_t old = st.slot_cur;
  st.slot_cur = (old + 1 == RS) ? 0 : old + 1;
  const uint32_t nxt = (st.slot_cur + 1 == RS) ? 0 : st.slot_cur + 1;
  if (!(ABL & 16)) ws_issue(st, old);                                                                       // 16: no DMA
  st.rd_cur = st.rd_next;
  st.rd_next = st.lds_base + nxt * (CF * 1024) + st.lane_off;
}

// fragment position p inside the current chunk has just been consumed: re-fill its register with fragment
// p + kRegFrags (same chunk, or the next chunk -- already landed: see ws_boundary)
template <int ABL, int CF, int RS, int LPW>
__device__ __forceinline__ void ws_refill(WStream<CF, RS, LPW>& st, int p) {
  if (ABL & 2) {
    asm volatile("" : "+v"(st.R[p % kRegFrags]));
    return;
  }
  const int q = p + kRegFrags;
  st.R[p % kRegFrags] = (q < CF) ? lds_read128(st.rd_cur + q * 1024) : lds_read128(st.rd_next + (q - CF) * 1024);
}

template <int CF, int RS, int LPW>
__device__ __forceinline__ void ws_start(WStream<CF, RS, LPW>& st, const void* gbase, uint32_t gbytes, char* lds, int wave, int lane) {
  st.gbase = reinterpret_cast<const char*>(gbase);
  st.gbytes = gbytes;
  st.goff = 0;
  st.lane_off = lane * 16;
  st.wave_off = wave * LPW * 1024;
  st.lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds));
#pragma unroll
  for (int k = 0; k < RS - 1; ++k) ws_issue(st, k);
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((RS - 2) * LPW) : "memory");
  st.slot_cur = RS - 1;                       // the first boundary moves to slot 0 = chunk 0
  st.rd_cur = st.lds_base + st.lane_off;      // unused until then
  st.rd_next = st.lds_base + st.lane_off;     // chunk 0
#pragma unroll
  for (int i = 0; i < kRegFrags; ++i) st.R[i] = lds_read128(st.rd_next + i * 1024);
}

// ReLU on the raw bits: max(int(x), 0) is +0.0 for every negative float and the identity for positive
// ones -- one v_max_i32, no canonicalising v_max_f32 pair.
__device__ __forceinline__ float relu_bits(float x) {
  int i = __builtin_bit_cast(int, x);
  i = i > 0 ? i : 0;
  return __builtin_bit_cast(float, i);
}

template <class ET, int F>
__device__ __forceinline__ void pe_pack(const float x[3], int h, uint32_t* out) {
  float t[pe_slots(F)];
  pe_eval<F, false>(x, h, t);
#pragma unroll
  for (int q = 0; q < pe_slots(F) / 2; ++q) out[q] = ET::pack(t[2 * q], t[2 * q + 1]);
}

// Bias block [m][h][16] for this lane-half from LDS with hand-issued reads: hipcc cannot see an asm
// ds_read, so it neither assumes aliasing with the LDS-DMA ring (which costs an s_waitcnt vmcnt(0) drain
// per tile) nor needs the 3-VALU-per-value SGPR select that scalar loads cost.  The wait statement names
// every destination "+v" so no consumer is scheduled above it (cdna_hip_programming.md 5.7 form ii).
struct BiasRegs {
  f32x4 b0, b1, b2, b3;
};
// issue now, consume later: the reads stay in flight behind the MFMAs of the current tile
__device__ __forceinline__ void lds_bias_issue(uint32_t byte_addr, BiasRegs& r) {
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48"
               : "=&v"(r.b0), "=&v"(r.b1), "=&v"(r.b2), "=&v"(r.b3)
               : "v"(byte_addr));
}
__device__ __forceinline__ void lds_bias_take(BiasRegs& r, f32x16* acc) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.b0), "+v"(r.b1), "+v"(r.b2), "+v"(r.b3));
  f32x16 a;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    a[e] = r.b0[e];
    a[4 + e] = r.b1[e];
    a[8 + e] = r.b2[e];
    a[12 + e] = r.b3[e];
  }
  *acc = a;
}

__device__ __forceinline__ void lds_bias16(uint32_t byte_addr, f32x16* acc) {
  f32x4 b0, b1, b2, b3;
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48"
               : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
               : "v"(byte_addr));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
  f32x16 a;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    a[e] = b0[e];
    a[4 + e] = b1[e];
    a[8 + e] = b2[e];
    a[12 + e] = b3[e];
  }
  *acc = a;
}

// One 16-bit layer for one 32-sample column block.  Input = two register segments (S1 then S2
// k-steps of 8 slots = 4 packed dwords each); output tile m lands in out[8m .. 8m+7] (packed pairs).
// FPOS = position of the layer's first fragment in the stream modulo the chunk size.
// KEEP_F32_TILE >= 0: that tile's raw accumulator is returned in *keep instead (alpha / rgb rows);
// kKeepAllF32: all of them, in keep[0 .. MT-1] (the sampling net's 128 raw outputs).
// epilogue of one accumulator quad g (values 4g..4g+3 of tile m): convert, ReLU on the packed pairs
template <class ET, bool RELU>
__device__ __forceinline__ void epilogue_quad_16(const f32x16& acc, int m, int g, uint32_t* out) {
  // convert first, then ReLU on the packed pair: max(int16, 0) clears every negative bf16/f16
  // (one v_cvt_pk + one v_pk_max_i16 per two values)
  uint32_t p0 = ET::pack(acc[4 * g + 0], acc[4 * g + 1]), p1 = ET::pack(acc[4 * g + 2], acc[4 * g + 3]);
  if (RELU) {
    const s16x2 z = {0, 0};
    p0 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p0), z));
    p1 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p1), z));
  }
  out[8 * m + 2 * g + 0] = p0;
  out[8 * m + 2 * g + 1] = p1;
}

constexpr int kKeepAllF32 = -2;   // layer_16 KEEP_F32_TILE: every tile's raw accumulator goes to keep[m]
template <class ET, class WS, int S1, int S2, int MT, bool RELU, int FPOS, int KEEP_F32_TILE = -1>
__device__ __forceinline__ void layer_16(WS& st, uint32_t bias_addr, int lane, const uint32_t* in1, const uint32_t* in2,
                                         uint32_t* out, f32x16* keep = nullptr) {
  constexpr int CF = ADN_CF;
  constexpr int KS = S1 + S2;
  // bias_addr: LDS byte address of this layer's bias block for THIS lane-half ([m][h][16] floats)
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    f32x16 acc;
    if (ADN_ABLATE & 4) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    } else {
      lds_bias16(bias_addr + m * 128, &acc);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int f = (FPOS + m * KS + s) % CF;     // position inside the chunk; compile-time after unrolling
      if (f == 0) ws_boundary<ADN_ABLATE>(st);
      const uint32_t* src = (s < S1) ? (in1 + 4 * s) : (in2 + 4 * (s - S1));
      u32x4 b = {src[0], src[1], src[2], src[3]};
      acc = ET::mfma(st.R[f % kRegFrags], b, acc);
      ws_refill<ADN_ABLATE>(st, f);
    }
    if (KEEP_F32_TILE == kKeepAllF32) {
      keep[m] = acc;
    } else if (KEEP_F32_TILE == m) {
      *keep = acc;
    } else if (ADN_ABLATE & 8) {
      asm volatile("" ::"v"(acc));
#pragma unroll
      for (int g = 0; g < 8; ++g) asm volatile("" : "=v"(out[8 * m + g]));
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) epilogue_quad_16<ET, RELU>(acc, m, g, out);
    }
  }
}

// Loads the sample's ray record and evaluates position (+ optional unit direction).
__device__ __forceinline__ void load_sample(const ShadeArgs& a, int s, int total, float x[3], float dpe[3]) {
  const int si = (s < total) ? s : (total > 0 ? total - 1 : 0);
  const uint32_t key = a.sample_key[si];
  const uint32_t ray = key >> 7;
  const int bin = static_cast<int>(key & 127u);
  const float4* rr = reinterpret_cast<const float4*>(a.rays + static_cast<size_t>(ray) * 8);
  const float4 o4 = rr[0], d4 = rr[1];
  const float o[3] = {o4.x, o4.y, o4.z}, d[3] = {d4.x, d4.y, d4.z};
  sample_position(a.sp, o, d, a.sample_z ? a.sample_z[si] : a.sp.ztab[bin], x);
  if (a.sp.unit_dir) unit3(d, dpe);
  else {
    dpe[0] = d[0];
    dpe[1] = d[1];
    dpe[2] = d[2];
  }
}

// Wave-private LDS stash for packed PE slots (hand-issued so hipcc neither orders them against the LDS-DMA
// ring with vmcnt(0) nor keeps 24 VGPRs alive across the layer stack).  Layout [dword group of 4][lane]:
// b128 accesses are lane-linear, hence conflict-free.
template <int NQ>   // NQ = number of b128 groups
__device__ __forceinline__ void lds_stash_write(uint32_t byte_addr, const uint32_t* v) {
#pragma unroll
  for (int g = 0; g < NQ; ++g) {
    const u32x4 t = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
    asm volatile("ds_write_b128 %0, %1" ::"v"(byte_addr + g * 1024), "v"(t) : "memory");
  }
}
template <int NQ>
__device__ __forceinline__ void lds_stash_read(uint32_t byte_addr, uint32_t* v) {
  u32x4 t[NQ];
#pragma unroll
  for (int g = 0; g < NQ; ++g) asm volatile("ds_read_b128 %0, %1" : "=&v"(t[g]) : "v"(byte_addr + g * 1024));
  if (NQ == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[NQ > 3 ? 3 : 0]));
  else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[NQ > 1 ? 1 : 0]));
#pragma unroll
  for (int g = 0; g < NQ; ++g) {
    v[4 * g] = t[g][0];
    v[4 * g + 1] = t[g][1];
    v[4 * g + 2] = t[g][2];
    v[4 * g + 3] = t[g][3];
  }
}

// A5+A6, 16-bit MFMA path.  Workgroup = WAVES waves x 32 samples; persistent over tiles; the weight
// stream is cyclic so DMA prefetch runs across tile boundaries.  WAVES = 4 with two workgroups per CU
// (two waves per SIMD from DIFFERENT workgroups): each workgroup has its own ring and barriers, so the
// two waves sharing a SIMD are not in lockstep and one computes while the other waits at its barrier.
template <class ET, int FP, int FD, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : 2) void shade_mlp16_kernel(ShadeArgs a) {
  static_assert(FP == 10 && FD == 4, "fragment positions below assume the 10-4 shading encoding");
  static_assert(WAVES == 4 || WAVES == 8, "chunk = 8 fragments");
  constexpr int QP = pe_slots(FP), QD = pe_slots(FD);
  constexpr int TILE = WAVES * 32, CF = ADN_CF, RS = ADN_RS, LPW = CF / WAVES;
  constexpr int kRingBytes = CF * RS * 1024;
  static_assert(CF % WAVES == 0 && CF % kRegFrags == 0 && kShadeFrags16 % CF == 0 && CF % 8 == 0 && CF <= 32, "chunk geometry");
  typedef WStream<CF, RS, LPW> WS;
#ifndef ADN_STASH
#define ADN_STASH 1   // 1: PE slots computed once per tile and parked in LDS; 0: sample re-loaded at layers 5 / view
#endif
  constexpr int kStashBytes = ADN_STASH ? WAVES * (QP / 8 + QD / 8) * 1024 : 0;
  __shared__ __attribute__((aligned(16))) char lds[kRingBytes + kShadeBiasFloats * 4 + kStashBytes];
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const int j = lane & 31, h = lane >> 5;
  const uint32_t stash = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds)) + kRingBytes + kShadeBiasFloats * 4 +
                         wave * (QP / 8 + QD / 8) * 1024 + lane * 16;
  int total = a.total ? *a.total : a.max_samples;
  if (total > a.max_samples) total = a.max_samples;
  const int ntiles = (total + TILE - 1) / TILE;
  if (static_cast<int>(blockIdx.x) >= ntiles) return;    // workgroup-uniform

  {
    float* lds_bias = reinterpret_cast<float*>(lds + kRingBytes);
    for (int i = threadIdx.x; i < kShadeBiasFloats; i += blockDim.x) lds_bias[i] = a.net.bias[i];
  }
  __syncthreads();
  WS st;
  ws_start(st, a.net.w, kShadeFrags16 * 1024, lds, wave, lane);

  // LDS byte address of the bias blocks of this lane-half
  const uint32_t bias0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds)) + kRingBytes + h * 64;
  const uint32_t* bo = a.net.b_off;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int s = tile * TILE + wave * 32 + j;
    uint32_t hA[64], hB[64];
    {
      float x[3], dpe[3];
      load_sample(a, s, total, x, dpe);
      uint32_t pts[QP / 2];
      pe_pack<ET, FP>(x, h, pts);
      if (ADN_STASH) {
        uint32_t dirs[QD / 2];
        pe_pack<ET, FD>(dpe, h, dirs);
        lds_stash_write<QP / 8>(stash, pts);
        lds_stash_write<QD / 8>(stash + (QP / 8) * 1024, dirs);
      }
      layer_16<ET, WS, QP / 8, 0, 8, true, 0>(st, bias0 + bo[0] * 4, lane, pts, pts, hA);
    }
#pragma unroll 1
    for (int l = 1; l <= 3; l += 2) {
      layer_16<ET, WS, 16, 0, 8, true, 0>(st, bias0 + bo[l] * 4, lane, hA, hA, hB);
      layer_16<ET, WS, 16, 0, 8, true, 0>(st, bias0 + bo[l + 1] * 4, lane, hB, hB, hA);
    }
    {
      // the skip connection re-loads the sample and re-evaluates the 32 position slots instead of
      // holding 16 (+6) VGPRs across layers 1-4 (30 v_sin per lane vs ~600 MFMA issue slots)
      uint32_t pts[QP / 2];
      if (ADN_STASH) {
        lds_stash_read<QP / 8>(stash, pts);
      } else {
        float x[3], dpe[3];
        load_sample(a, s, total, x, dpe);
        pe_pack<ET, FP>(x, h, pts);
      }
      layer_16<ET, WS, QP / 8, 16, 8, true, 0>(st, bias0 + bo[5] * 4, lane, pts, hA, hB);   // cat([pts, h])
    }
    layer_16<ET, WS, 16, 0, 8, true, 0>(st, bias0 + bo[6] * 4, lane, hB, hB, hA);
    layer_16<ET, WS, 16, 0, 8, true, 0>(st, bias0 + bo[7] * 4, lane, hA, hA, hB);
    f32x16 alpha_tile;
    layer_16<ET, WS, 16, 0, 9, false, 0, 8>(st, bias0 + bo[8] * 4, lane, hB, hB, hA, &alpha_tile);      // feature (+alpha row)
    const float alpha = alpha_tile[0];
    {
      uint32_t dirs[QD / 2];
      if (ADN_STASH) {
        lds_stash_read<QD / 8>(stash + (QP / 8) * 1024, dirs);
      } else {
        float x[3], dpe[3];
        load_sample(a, s, total, x, dpe);
        pe_pack<ET, FD>(dpe, h, dirs);
      }
      layer_16<ET, WS, 16, QD / 8, 4, true, (32 + 4 * 128 + 160 + 2 * 128 + 144) % CF>(st, bias0 + bo[9] * 4, lane, hA, dirs, hB);             // cat([feature, dir])
    }
    f32x16 rgb_tile;
    layer_16<ET, WS, 8, 0, 1, false, (32 + 4 * 128 + 160 + 2 * 128 + 144 + 72) % CF, 0>(st, bias0 + bo[10] * 4, lane, hB, hB, hA, &rgb_tile);
    if (h == 0 && s < total)
      *reinterpret_cast<float4*>(a.raw_out + static_cast<size_t>(s) * 4) = make_float4(rgb_tile[0], rgb_tile[1], rgb_tile[2], alpha);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup's LDS allocation
}

// fp32 parity mode of the shading net: same structure on the fp32 MFMA engine, accurate sincos.
template <int FP, int FD>
__global__ __launch_bounds__(256) void shade_mlp32_kernel(ShadeArgs a) {
  constexpr int QP = pe_slots(FP), QD = pe_slots(FD);
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
    // the weight addresses do not depend on the tile: without this the compiler hoists every A-fragment
    // load out of the tile loop (loop-invariant code motion) and spills thousands of registers
    asm volatile("" : "+v"(w), "+v"(b));
    float x[3], dpe[3];
    load_sample(a, s, total, x, dpe);
    float pts[QP], dirs[QD], hA[144], hB[128];      // hA also receives the 9-tile feature(+alpha) layer
    pe_eval<FP, true>(x, h, pts);
    pe_eval<FD, true>(dpe, h, dirs);
    layer_f32<QP, 0, 8, true>(w + a.net.w_off[0], b + a.net.b_off[0], lane, pts, pts, hA);
#pragma unroll 1
    for (int l = 1; l <= 3; l += 2) {
      layer_f32<128, 0, 8, true>(w + a.net.w_off[l], b + a.net.b_off[l], lane, hA, hA, hB);
      layer_f32<128, 0, 8, true>(w + a.net.w_off[l + 1], b + a.net.b_off[l + 1], lane, hB, hB, hA);
    }
    layer_f32<QP, 128, 8, true>(w + a.net.w_off[5], b + a.net.b_off[5], lane, pts, hA, hB);       // cat([pts, h])
    layer_f32<128, 0, 8, true>(w + a.net.w_off[6], b + a.net.b_off[6], lane, hB, hB, hA);
    layer_f32<128, 0, 8, true>(w + a.net.w_off[7], b + a.net.b_off[7], lane, hA, hA, hB);
    layer_f32<128, 0, 9, false>(w + a.net.w_off[8], b + a.net.b_off[8], lane, hB, hB, hA);         // feature (+alpha row)
    const float alpha = hA[128];
    layer_f32<128, QD, 4, true>(w + a.net.w_off[9], b + a.net.b_off[9], lane, hA, dirs, hB);       // cat([feature, dir])
    float rgb[16];
    layer_f32<64, 0, 1, false>(w + a.net.w_off[10], b + a.net.b_off[10], lane, hB, hB, rgb);
    if (h == 0 && s < total)
      *reinterpret_cast<float4*>(a.raw_out + static_cast<size_t>(s) * 4) = make_float4(rgb[0], rgb[1], rgb[2], alpha);
  }
}

// Debug/parity: explicit shading-net input features in the reference's column order.
template <int FP, int FD>
__global__ __launch_bounds__(256) void shade_features_kernel(ShadeArgs a, float* feat) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.max_samples) return;
  float x[3], dpe[3];
  load_sample(a, s, a.max_samples, x, dpe);
  constexpr int NP = 3 + 6 * FP, ND = 3 + 6 * FD;
  float* f = feat + static_cast<size_t>(s) * (NP + ND);
  for (int c = 0; c < 3; ++c) {
    f[c] = x[c];
    f[NP + c] = dpe[c];
  }
  for (int b = 0; b < FP; ++b)
    for (int c = 0; c < 3; ++c) {
      float sn, co;
      sincosf(x[c] * static_cast<float>(1 << b), &sn, &co);
      f[3 + 6 * b + c] = sn;
      f[3 + 6 * b + 3 + c] = co;
    }
  for (int b = 0; b < FD; ++b)
    for (int c = 0; c < 3; ++c) {
      float sn, co;
      sincosf(dpe[c] * static_cast<float>(1 << b), &sn, &co);
      f[NP + 3 + 6 * b + c] = sn;
      f[NP + 3 + 6 * b + 3 + c] = co;
    }
}

// ---- split-precision sampling MLP: fp16 hi + 2^-11 * fp16 lo', three MFMAs per term --------------
// x ~= hi + lo' / 2048 with hi = fp16(x), lo' = fp16((x - hi) * 2048): 22 significant bits and no
// dependence on fp16 subnormals.  W.x = Whi.xhi + (Whi.xlo' + Wlo'.xhi) / 2048 (the lo'.lo' term is
// 2^-22 relative and dropped).  Main and cross products accumulate in separate fp32 accumulators.
// Measured against an fp64 reference on the shipped weights: max error 1.9e-6 (numpy sgemm: 2.4e-6),
// identical selections on 100 % of rays -- at 3/16 of the fp32-MFMA cycle count.
// kSplitScale (2^11) is defined in pack.hpp

__device__ __forceinline__ void split_pack(float v0, float v1, uint32_t* hi, uint32_t* lo) {
  f32x2 v = {v0, v1};
  f16x2 h = __builtin_convertvector(v, f16x2);
  f32x2 hf = __builtin_convertvector(h, f32x2);
  f32x2 r = (v - hf) * kSplitScale;
  *hi = __builtin_bit_cast(uint32_t, h);
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

template <class WS, int KS, int MT, bool LAST, int FPOS>
__device__ __forceinline__ void layer_16x3(WS& st, uint32_t bias_addr, int lane, const uint32_t* in_hi,
                                           const uint32_t* in_lo, uint32_t* out_hi, uint32_t* out_lo, float* out_f32) {
  // Software pipeline across output tiles (one wave per SIMD: nothing else hides these latencies):
  //  - the bias block of tile m+1 is requested right after tile m's accumulators are initialised,
  //  - the epilogue of tile m-1 is spread, one accumulator pair per k-step, over tile m's MFMAs.
  constexpr bool PIPE = !(ADN_ABLATE_S & 64);
  BiasRegs br;
  f32x16 pacc, pcross;   // previous tile's accumulators (PIPE)
  if (!(ADN_ABLATE_S & 4)) lds_bias_issue(bias_addr, br);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    f32x16 acc, cross;
    if (ADN_ABLATE_S & 4) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    } else {
      lds_bias_take(br, &acc);
      if (m + 1 < MT) lds_bias_issue(bias_addr + (m + 1) * 128, br);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) cross[r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int f = (FPOS + 2 * (m * KS + s)) % ADN_CF_S;   // compile-time after unrolling; always even
      if (f == 0) ws_boundary<ADN_ABLATE_S>(st);
      const u32x4 bh = {in_hi[4 * s], in_hi[4 * s + 1], in_hi[4 * s + 2], in_hi[4 * s + 3]};
      const u32x4 bl = {in_lo[4 * s], in_lo[4 * s + 1], in_lo[4 * s + 2], in_lo[4 * s + 3]};
      acc = Fp16::mfma(st.R[f % kRegFrags], bh, acc);
      cross = Fp16::mfma(st.R[f % kRegFrags], bl, cross);
      cross = Fp16::mfma(st.R[(f + 1) % kRegFrags], bh, cross);
      ws_refill<ADN_ABLATE_S>(st, f);
      ws_refill<ADN_ABLATE_S>(st, f + 1);
      if (PIPE && m > 0 && !(ADN_ABLATE_S & 8)) {
        // KS >= 2: spread the 8 pairs over the first k-steps (all 8 in step 0/1 when KS < 8)
        constexpr int PER = (KS >= 8) ? 1 : (8 + KS - 1) / KS;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
          const int pi = s * PER + k;
          if (pi < 8) epilogue_pair_16x3<LAST>(pacc, pcross, m - 1, pi, out_hi, out_lo, out_f32);
        }
      }
    }
    if ((ADN_ABLATE_S & 8) && !LAST) {
      asm volatile("" ::"v"(acc), "v"(cross));
#pragma unroll
      for (int g = 0; g < 8; ++g) asm volatile("" : "=v"(out_hi[8 * m + g]), "=v"(out_lo[8 * m + g]));
      continue;
    }
    if (PIPE && m + 1 < MT) {
      pacc = acc;
      pcross = cross;
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
  constexpr int WAVES = 4, CF = ADN_CF_S, RS = ADN_RS_S, LPW = CF / WAVES, TILE = WAVES * 32;
  constexpr int F0 = 2 * (Q0 / 8) * 8;                  // layer-0 fragments (hi + lo')
  constexpr int FRAGS = F0 + 6 * 256 + 128;
  static_assert(F0 % CF == 0 && FRAGS % CF == 0 && CF % WAVES == 0 && CF % kRegFrags == 0 && CF <= 32, "chunk geometry");
  typedef WStream<CF, RS, LPW> WS;
  constexpr int kRingBytes = CF * RS * 1024, kBiasFloats = 7 * 256 + 128;
  __shared__ __attribute__((aligned(16))) char lds[kRingBytes + kBiasFloats * 4];
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int ntiles = (a.n_rays + TILE - 1) / TILE;
  if (static_cast<int>(blockIdx.x) >= ntiles) return;

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
    const int local = tile * TILE + wave * 32 + j;
    const bool valid = local < a.n_rays;
    const int ray = a.first_ray + (valid ? local : a.n_rays - 1);
    int col, row;
    ray_pixel(a.g, ray, &col, &row);
    float nds[3], p[3], u[3];
    gen_ray(a.g, col, row, nds, p);
    unit3(nds, u);

    uint32_t aH[64], aL[64], bH[64], bL[64];
    {
      float t[Q0];
      pe_eval<FD, !(ADN_ABLATE_S & 128)>(u, h, t);            // [dir PE | pos PE]  (src/features.py:868-874)
      pe_eval<FP, !(ADN_ABLATE_S & 128)>(p, h, t + QD);
#pragma unroll
      for (int q = 0; q < Q0 / 2; ++q) split_pack(t[2 * q], t[2 * q + 1], &aH[q], &aL[q]);
    }
    layer_16x3<WS, Q0 / 8, 8, false, 0>(st, bias0 + bo[0] * 4, lane, aH, aL, bH, bL, nullptr);
#pragma unroll 1
    for (int l = 1; l <= 5; l += 2) {
      layer_16x3<WS, 16, 8, false, 0>(st, bias0 + bo[l] * 4, lane, bH, bL, aH, aL, nullptr);
      layer_16x3<WS, 16, 8, false, 0>(st, bias0 + bo[l + 1] * 4, lane, aH, aL, bH, bL, nullptr);
    }
    float out[64];
    layer_16x3<WS, 16, 4, true, 0>(st, bias0 + bo[7] * 4, lane, bH, bL, nullptr, nullptr, out);

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
        // an activation beyond the fp16 range (65504) shows up as inf/NaN here
        if (bad && a.overflow_flag) atomicAdd(a.overflow_flag, 1);
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
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// A1+A2+A3 in plain fp16 (ADANERF_SAMPLING_FP16): one MFMA per term, fp32 accumulate -- the arithmetic the
// reference VIEWER runs its sampling network in (TensorRT kFP16, adanerf_real_time_viewer/src/imagegenerator.cpp:155-156).
// 11-bit operands move a few outputs across the threshold / the N-th rank (raw error <= 3e-3), so the selected bins
// differ from the fp32 PyTorch path on 0.3-1.5 % of rays: an opt-in speed mode, never the default.  Same engine as the shading
// kernel (8 waves x 32 rays per workgroup, activations in registers, weights through the LDS ring).
template <int FP, int FD>
constexpr int sample16_frags() { return ((pe_slots(FD) + pe_slots(FP)) / 8) * 8 + 6 * 128 + 64; }

template <int FP, int FD>
__global__ __launch_bounds__(512, 2) void sample_mlp16_kernel(SampleArgs a) {
  constexpr int QD = pe_slots(FD), QP = pe_slots(FP), Q0 = QD + QP;
  constexpr int WAVES = 8, CF = ADN_CF, RS = ADN_RS, LPW = CF / WAVES, TILE = WAVES * 32;
  constexpr int F0 = (Q0 / 8) * 8, FRAGS = sample16_frags<FP, FD>();
  static_assert(F0 % CF == 0 && FRAGS % CF == 0 && CF % WAVES == 0 && CF % kRegFrags == 0 && CF <= 32, "chunk geometry");
  typedef WStream<CF, RS, LPW> WS;
  constexpr int kRingBytes = CF * RS * 1024, kBiasFloats = 7 * 256 + 128;
  __shared__ __attribute__((aligned(16))) char lds[kRingBytes + kBiasFloats * 4];
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int ntiles = (a.n_rays + TILE - 1) / TILE;
  if (static_cast<int>(blockIdx.x) >= ntiles) return;
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
    const int local = tile * TILE + wave * 32 + j;
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
      pe_eval<FD, !(ADN_ABLATE_S & 128)>(u, h, t);            // [dir PE | pos PE]  (src/features.py:868-874)
      pe_eval<FP, !(ADN_ABLATE_S & 128)>(p, h, t + QD);
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
      if (bad && a.overflow_flag) atomicAdd(a.overflow_flag, 1);   // an activation left the fp16 range
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// DONeRF inverse-CDF sampler (FromClassifiedDepth) + classic sigma/delta compositing
// ------------------------------------------------------------------------------------------

struct DepthMap {          // warped depth t in [0,1] -> world depth (src/util/depth_transformations.py:37-58)
  float d0, d1;
  int32_t log_transform;   // 1: (d1-d0+1)^t - 1 + d0, 0: t (d1-d0) + d0
};

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ float wave_incl_scan_f32(float v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float t = __shfl_up(v, off);
    if (lane >= off) v += t;
  }
  return v;
}

// FromClassifiedDepth.generate + nerf_sample_pdf(det=True) (src/nerf_raymarch_common.py:606-660, 160-192):
// sigmoid(oracle) + 1e-5 -> pdf -> cdf over the 129 bin edges -> invert at u = k/(n+1), k = 1..n.
// One wave per ray; the cdf goes through a wave-private LDS row and every lane inverts its own u.
__global__ __launch_bounds__(256) void pdf_sample_kernel(const float* __restrict__ oracle, int n_rays, int n, DepthMap dm,
                                                         int32_t* __restrict__ ray_offsets, int32_t* __restrict__ counts,
                                                         uint32_t* __restrict__ sample_key, float* __restrict__ sample_w,
                                                         float* __restrict__ sample_z, int32_t* __restrict__ total) {
  __shared__ float cdf_s[4][kBins + 1 + 3];
  const int lane = lane_id();
  const int wave = static_cast<int>(threadIdx.x) >> 6;
  float* cdf = cdf_s[wave];
  if (blockIdx.x == 0 && threadIdx.x == 0) *total = n_rays * n;
  for (int r = (blockIdx.x * 4 + wave); r < n_rays; r += gridDim.x * 4) {
    const float* row = oracle + static_cast<size_t>(r) * kBins;
    const float w0 = sigmoidf_dev(row[lane]) + 1e-5f, w1 = sigmoidf_dev(row[64 + lane]) + 1e-5f;
    const float tot = wave_sum_f32(w0 + w1);
    const float p0 = w0 / tot, p1 = w1 / tot;
    const float cA = wave_incl_scan_f32(p0, lane);
    const float totA = __shfl(cA, 63);
    const float cB = totA + wave_incl_scan_f32(p1, lane);
    if (lane == 0) cdf[0] = 0.f;
    cdf[1 + lane] = cA;
    cdf[65 + lane] = cB;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      ray_offsets[r] = r * n;
      counts[r] = n;
    }
    for (int k = lane; k < n; k += 64) {
      const float u = static_cast<float>(k + 1) / static_cast<float>(n + 1);     // linspace(0,1,n+2)[k+1]
      int lo = 0, hi = kBins + 1;                                                 // searchsorted(cdf, u, right=True)
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] <= u) lo = mid + 1;
        else hi = mid;
      }
      const int below = max(lo - 1, 0), above = min(lo, kBins);
      const float c0 = cdf[below], c1 = cdf[above];
      float denom = c1 - c0;
      denom = denom < 1e-5f ? 1.0f : denom;
      const float t = (u - c0) / denom;
      const float b0 = static_cast<float>(below) * (1.0f / kBins), b1 = static_cast<float>(above) * (1.0f / kBins);
      const float zw = __fadd_rn(b0, __fmul_rn(t, b1 - b0));
      float z;
      if (dm.log_transform) z = powf(static_cast<float>(static_cast<double>(dm.d1) - dm.d0 + 1.0), zw) - 1.0f + dm.d0;
      else z = zw * (dm.d1 - dm.d0) + dm.d0;
      const size_t o = static_cast<size_t>(r) * n + k;
      sample_z[o] = z;
      sample_key[o] = (static_cast<uint32_t>(r) << 7) | static_cast<uint32_t>(min(below, kBins - 1));
      sample_w[o] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// nerf_raw2outputs (src/nerf_raymarch_common.py:19-68): alpha = 1 - exp(-relu(raw_a) * (z[k+1]-z[k]) * |d|),
// last interval 1e10; rgb = sigmoid(raw); front-to-back with the 1e-10 transmittance floor.
__global__ __launch_bounds__(256) void composite_classic_kernel(const float4* __restrict__ raw, const float* __restrict__ sample_z,
                                                                const float* __restrict__ rays, int n_rays, int n,
                                                                float* __restrict__ rgb_out, uchar4* __restrict__ rgba8_out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const float4 d4 = reinterpret_cast<const float4*>(rays + static_cast<size_t>(r) * 8)[1];
  const float dn = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d4.x, d4.x), __fmul_rn(d4.y, d4.y)), __fmul_rn(d4.z, d4.z)));
  const size_t o = static_cast<size_t>(r) * n;
  float cr = 0.f, cg = 0.f, cb = 0.f, T = 1.f;
  float zk = sample_z[o];
  for (int k = 0; k < n; ++k) {
    const float4 v = raw[o + k];
    const float zn = (k + 1 < n) ? sample_z[o + k + 1] : 0.f;
    const float dist = __fmul_rn((k + 1 < n) ? __fsub_rn(zn, zk) : 1e10f, dn);
    const float al = __fsub_rn(1.0f, expf(-__fmul_rn(fmaxf(v.w, 0.f), dist)));
    const float wt = __fmul_rn(al, T);
    cr = __fadd_rn(cr, __fmul_rn(wt, sigmoidf_dev(v.x)));
    cg = __fadd_rn(cg, __fmul_rn(wt, sigmoidf_dev(v.y)));
    cb = __fadd_rn(cb, __fmul_rn(wt, sigmoidf_dev(v.z)));
    T = __fmul_rn(T, __fadd_rn(__fsub_rn(1.0f, al), 1e-10f));
    zk = zn;
  }
  if (rgb_out) {
    rgb_out[3 * static_cast<size_t>(r) + 0] = cr;
    rgb_out[3 * static_cast<size_t>(r) + 1] = cg;
    rgb_out[3 * static_cast<size_t>(r) + 2] = cb;
  }
  if (rgba8_out) {
    uchar4 px;
    px.x = static_cast<unsigned char>(fminf(fmaxf(cr, 0.f), 1.f) * 255.0f);
    px.y = static_cast<unsigned char>(fminf(fmaxf(cg, 0.f), 1.f) * 255.0f);
    px.z = static_cast<unsigned char>(fminf(fmaxf(cb, 0.f), 1.f) * 255.0f);
    px.w = 255;
    rgba8_out[r] = px;
  }
}

// ------------------------------------------------------------------------------------------
// A7: compositing (src/nerf_raymarch_common.py:91-144)
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// one sample of the front-to-back recurrence (src/nerf_raymarch_common.py:91-144): every product and sum rounds
// to fp32 where the reference's does
__device__ __forceinline__ void composite_step(const float4 v, float wv, int mult_mode, float& cr, float& cg, float& cb, float& T) {
  float al = sigmoidf(v.w);
  if (mult_mode == 1) al = __fmul_rn(al, wv);
  float wt = __fmul_rn(al, T);
  if (mult_mode == 2) wt = __fmul_rn(wt, wv);
  cr = __fadd_rn(cr, __fmul_rn(wt, sigmoidf(v.x)));
  cg = __fadd_rn(cg, __fmul_rn(wt, sigmoidf(v.y)));
  cb = __fadd_rn(cb, __fmul_rn(wt, sigmoidf(v.z)));
  T = __fmul_rn(T, __fadd_rn(__fsub_rn(1.0f, al), 1e-10f));
}

// Thread per ray, sequential over its samples (the reference's cumprod order, bit for bit).  The samples of the
// workgroup's 256 consecutive rays are one contiguous range of the compacted arrays, so they are first copied to LDS
// with coalesced 16-byte loads (`cap` samples of dynamic LDS, 20 B each; RB = rays per workgroup is chosen so that
// RB * N samples stay under 48 KB); a thread striding through global memory
// instead touches a different 128-byte line per lane and per step (measured 3.6x the algorithmic HBM bytes).
// Offsets that are not the compactor's (stage API called with a hand-made layout) fall back to direct loads.
template <int RB>
__global__ __launch_bounds__(RB) void composite_kernel(const float4* __restrict__ raw, const float* __restrict__ sample_w,
                                                       const int32_t* __restrict__ ray_offsets, const int32_t* __restrict__ counts,
                                                       int n_rays, int mult_mode, int cap, float* __restrict__ rgb_out,
                                                       uchar4* __restrict__ rgba8_out) {
  extern __shared__ __attribute__((aligned(16))) char comp_lds[];
  float4* s_raw = reinterpret_cast<float4*>(comp_lds);
  float* s_w = reinterpret_cast<float*>(comp_lds + static_cast<size_t>(cap) * sizeof(float4));
  const int t = threadIdx.x;
  const int r0 = blockIdx.x * RB;
  const int r1 = min(r0 + RB, n_rays) - 1;                      // last ray of the workgroup (uniform)
  const int base = ray_offsets[r0];
  const int n = ray_offsets[r1] + counts[r1] - base;            // samples of the workgroup if the layout is the compactor's
  const bool staged = cap > 0 && n >= 0 && n <= cap;            // uniform
  if (staged) {
    for (int i = t; i < n; i += RB) {
      s_raw[i] = raw[base + i];
      s_w[i] = sample_w[base + i];
    }
    __syncthreads();
  }
  const int r = r0 + t;
  if (r >= n_rays) return;
  const int o = ray_offsets[r], c = counts[r];
  const int ol = o - base;
  float cr = 0.f, cg = 0.f, cb = 0.f, T = 1.f;
  if (staged && ol >= 0 && ol + c <= n) {
    for (int k = 0; k < c; ++k) composite_step(s_raw[ol + k], s_w[ol + k], mult_mode, cr, cg, cb, T);
  } else {
    for (int k = 0; k < c; ++k) composite_step(raw[o + k], sample_w[o + k], mult_mode, cr, cg, cb, T);
  }
  if (rgb_out) {
    rgb_out[3 * static_cast<size_t>(r) + 0] = cr;
    rgb_out[3 * static_cast<size_t>(r) + 1] = cg;
    rgb_out[3 * static_cast<size_t>(r) + 2] = cb;
  }
  if (rgba8_out) {
    // viewer output contract: (uchar)(clamp(v,0,1)*255), A = 255 (adaptive_cuda_kernels.cu:846-851)
    uchar4 px;
    px.x = static_cast<unsigned char>(fminf(fmaxf(cr, 0.f), 1.f) * 255.0f);
    px.y = static_cast<unsigned char>(fminf(fmaxf(cg, 0.f), 1.f) * 255.0f);
    px.z = static_cast<unsigned char>(fminf(fmaxf(cb, 0.f), 1.f) * 255.0f);
    px.w = 255;
    rgba8_out[r] = px;
  }
}

// Long rays (dense mode: 128 samples each): one wave per ray, lane holds samples (lane, lane + 64);
// transmittance = exclusive product scan of (1 - alpha + 1e-10) across the wave, colour = wave sum.
// Coalesced 16-byte loads instead of one thread striding through 2 KiB per ray.
__device__ __forceinline__ float wave_incl_prod_f32(float v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float t = __shfl_up(v, off);
    if (lane >= off) v *= t;
  }
  return v;
}

__global__ __launch_bounds__(256) void composite_wave_kernel(const float4* __restrict__ raw, const float* __restrict__ sample_w,
                                                             const int32_t* __restrict__ ray_offsets, const int32_t* __restrict__ counts,
                                                             int n_rays, int mult_mode, float* __restrict__ rgb_out, uchar4* __restrict__ rgba8_out) {
  const int lane = lane_id();
  const int r = blockIdx.x * 4 + (static_cast<int>(threadIdx.x) >> 6);
  if (r >= n_rays) return;
  const int o = ray_offsets[r], c = counts[r];
  float al[2], col[2][3];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k = lane + 64 * u;
    al[u] = 0.f;
    col[u][0] = col[u][1] = col[u][2] = 0.f;
    if (k < c) {
      const float4 v = raw[o + k];
      float a0 = sigmoidf_dev(v.w);
      const float wv = sample_w[o + k];
      if (mult_mode == 1) a0 = __fmul_rn(a0, wv);
      al[u] = a0;
      const float m = (mult_mode == 2) ? wv : 1.0f;
      col[u][0] = sigmoidf_dev(v.x) * m;
      col[u][1] = sigmoidf_dev(v.y) * m;
      col[u][2] = sigmoidf_dev(v.z) * m;
    }
  }
  const float f0 = __fadd_rn(__fsub_rn(1.0f, al[0]), 1e-10f), f1 = __fadd_rn(__fsub_rn(1.0f, al[1]), 1e-10f);
  const float p0 = wave_incl_prod_f32(lane + 0 < c ? f0 : 1.0f, lane);
  const float tot0 = __shfl(p0, 63);
  const float p1 = wave_incl_prod_f32(lane + 64 < c ? f1 : 1.0f, lane);
  float e0 = __shfl_up(p0, 1), e1 = __shfl_up(p1, 1);      // exclusive products
  if (lane == 0) {
    e0 = 1.0f;
    e1 = 1.0f;
  }
  const float w0 = al[0] * e0, w1 = al[1] * (tot0 * e1);
  float cr = w0 * col[0][0] + w1 * col[1][0], cg = w0 * col[0][1] + w1 * col[1][1], cb = w0 * col[0][2] + w1 * col[1][2];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    cr += __shfl_xor(cr, off);
    cg += __shfl_xor(cg, off);
    cb += __shfl_xor(cb, off);
  }
  if (lane == 0) {
    if (rgb_out) {
      rgb_out[3 * static_cast<size_t>(r) + 0] = cr;
      rgb_out[3 * static_cast<size_t>(r) + 1] = cg;
      rgb_out[3 * static_cast<size_t>(r) + 2] = cb;
    }
    if (rgba8_out) {
      uchar4 px;
      px.x = static_cast<unsigned char>(fminf(fmaxf(cr, 0.f), 1.f) * 255.0f);
      px.y = static_cast<unsigned char>(fminf(fmaxf(cg, 0.f), 1.f) * 255.0f);
      px.z = static_cast<unsigned char>(fminf(fmaxf(cb, 0.f), 1.f) * 255.0f);
      px.w = 255;
      rgba8_out[r] = px;
    }
  }
}

// multi-GPU: gathered [world][rays_local_max] uchar4 (rank-major) -> row-major image
__global__ __launch_bounds__(256) void assemble_strips_kernel(const uchar4* __restrict__ gathered, uchar4* __restrict__ image, int w, int h,
                                                              int strip_rows, int world, int rays_local_max) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w * h) return;
  const int row = i / w, col = i - row * w;
  const int strip = row / strip_rows;
  const int rank = strip % world, sl = strip / world;
  const int local = (sl * strip_rows + (row - strip * strip_rows)) * w + col;
  image[i] = gathered[static_cast<size_t>(rank) * rays_local_max + local];
}

}  // namespace adanerf
