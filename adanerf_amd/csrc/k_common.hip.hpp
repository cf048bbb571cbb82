// Types, kernel-argument structs and small device helpers shared by every stage: ray generation (A1/A2), positional
// encoding (sin_or_cos, pe_eval), sample position / normalisation (A5), wave reductions.
// Device code only (gfx950, wave64); part of kernels.hip.hpp.
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
  uint32_t n_bias;           // floats in `bias` (the run-time-shaped kernels copy the whole table to LDS once per workgroup)
  float out_scale[2];        // shading nets: factors that take the kernel's alpha / rgb outputs back to the network's own scale (exact
                             // powers of two; 1 unless the bf16 packing scaled the layers: pack.hpp PackedNet::out_exp)
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

// rayMarchNormalization (src/nerf_raymarch_common.py:195-244)
enum { kNormNone = 0, kNormInverseSqrtDistCentered = 1, kNormCentered = 2, kNormMaxDepth = 3, kNormMaxDepthCentered = 4, kNormLogCentered = 5,
       kNormInverseDistCentered = 6 };

struct ShadeParams {
  float center[3];                     // view_cell_center, or rayMarchNormalizationCenter when the config sets three values
  float max_depth;
  float sqrt_max_depth;
  int32_t normalize;                   // kNorm*
  int32_t unit_dir;                    // 1: PE(dir/|dir|) (NDC), 0: PE(dir) as received
  float log_max_depth_p1;              // math.log(max_depth + 1): kNormLogCentered
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
// polynomial part: r in [-pi/4, pi/4], n = quadrant (+ 1 for the cosine)
__device__ __forceinline__ float sin_or_cos_poly(float r, int n) {
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
// |a| < 1e5 (the caller knows): no branch
__device__ __forceinline__ float sin_or_cos_small(float a, int h) {
  const float j = __builtin_rintf(a * 0.636619747f);             // a * 2/pi
  float r = __builtin_fmaf(j, -1.57079601e+00f, a);              // pi/2 = 1.57079601 + 3.13916473e-7 + 5.39030253e-15
  r = __builtin_fmaf(j, -3.13916473e-07f, r);
  r = __builtin_fmaf(j, -5.39030253e-15f, r);
  return sin_or_cos_poly(r, static_cast<int>(j) + h);
}
__device__ __forceinline__ float sin_or_cos(float a, int h) {
  if (__builtin_expect(fabsf(a) < 1.0e5f, 1)) return sin_or_cos_small(a, h);
  // rare: the same reduction in fp64 (two-term pi/2), exact to ~1e-16 while the quotient fits a double's integers
  // (|a| < ~1e15).  Beyond that the argument's own fp32 spacing spans > 1e7 periods and the value carries no
  // information: the reduced argument is clamped so the result stays in [-1, 1], but it is not libm's value.
  // inf/NaN -> NaN like libm.
  const double ad = static_cast<double>(a);
  const double k = __builtin_rint(ad * 0.6366197723675814);
  double rd = __builtin_fma(k, -1.5707963267948966, ad);
  rd = __builtin_fma(k, -6.123233995736766e-17, rd);
  rd = __builtin_fmin(__builtin_fmax(rd, -0.7853981633974483), 0.7853981633974483);   // NaN stays NaN: see below
  const float r = (a != a || fabsf(a) == INFINITY) ? __builtin_nanf("") : static_cast<float>(rd);
  return sin_or_cos_poly(r, static_cast<int>(k - 4.0 * __builtin_floor(k * 0.25)) + h);
}

// PE slots of lane-half h (layout.hpp): slot q < 3F -> h ? cos : sin of 2^(q/3) * x[q%3];
// then two identity slots.  ACCURATE: libm-grade sin_or_cos (fp32 parity path);
// !ACCURATE: one v_sin_f32 per slot (cos = sin shifted by a quarter revolution).
template <int F, bool ACCURATE>
__device__ __forceinline__ void pe_eval(const float x[3], int h, float* out) {
  // ACCURATE: one wave-uniform range test for all 3 F arguments instead of a branch around every evaluation (round 6: 42 exec-mask branches per
  // ray block of the sampling kernels); the rare wave with an argument beyond the single-precision reduction (or a NaN) takes the general form.
  bool small = true;
  if (ACCURATE) {
    const float amax = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fabsf(x[2])) * static_cast<float>(1 << (F > 0 ? F - 1 : 0));
    small = __builtin_amdgcn_ballot_w64(!(amax < 1.0e5f)) == 0ull;
  }
  if (ACCURATE && small) {
#pragma unroll
    for (int q = 0; q < 3 * F; ++q) out[q] = sin_or_cos_small(x[q % 3] * static_cast<float>(1 << (q / 3)), h);
  } else {
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
  if (sp.normalize == kNormNone) return;
  if (sp.normalize == kNormMaxDepth) {      // x / max_depth: also what a config WITHOUT the key gets (src/features.py:319-324)
    x[0] = x[0] / sp.max_depth;
    x[1] = x[1] / sp.max_depth;
    x[2] = x[2] / sp.max_depth;
    return;
  }
  const float l[3] = {x[0] - sp.center[0], x[1] - sp.center[1], x[2] - sp.center[2]};
  if (sp.normalize == kNormCentered) {
    x[0] = l[0];
    x[1] = l[1];
    x[2] = l[2];
  } else if (sp.normalize == kNormMaxDepthCentered) {
    x[0] = l[0] / sp.max_depth;
    x[1] = l[1] / sp.max_depth;
    x[2] = l[2] / sp.max_depth;
  } else {
    const float n2 = __fadd_rn(__fadd_rn(__fmul_rn(l[0], l[0]), __fmul_rn(l[1], l[1])), __fmul_rn(l[2], l[2]));
    if (sp.normalize == kNormInverseSqrtDistCentered) {
      const float local = sqrtf(sqrtf(n2));
      const float den = __fmul_rn(sp.sqrt_max_depth, local);
      x[0] = l[0] / den;
      x[1] = l[1] / den;
      x[2] = l[2] / den;
    } else if (sp.normalize == kNormInverseDistCentered) {      // localized * (1 - 1 / (1 + |localized|))
      const float local = sqrtf(n2);
      const float f = __fsub_rn(1.0f, 1.0f / __fadd_rn(1.0f, local));
      x[0] = __fmul_rn(l[0], f);
      x[1] = __fmul_rn(l[1], f);
      x[2] = __fmul_rn(l[2], f);
    } else {      // kNormLogCentered: localized * (log(local + 1) / log(max_depth + 1) / local), local <= 0 -> 0.001 (LogTransform.from_world
                  // clamps in place, util/depth_transformations.py:21-27, so the divisor sees the clamped value too)
      float local = sqrtf(n2);
      local = local <= 0.f ? 0.001f : local;
      const float f = (logf(__fadd_rn(local, 1.0f)) / sp.log_max_depth_p1) / local;
      x[0] = __fmul_rn(l[0], f);
      x[1] = __fmul_rn(l[1], f);
      x[2] = __fmul_rn(l[2], f);
    }
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

// ---- wave64 float reductions / scans on DPP --------------------------------------------------------------------------------------
// The float butterflies of the one-wave-per-ray compositing kernels used to go through __shfl_xor / __shfl_up (ds_bpermute_b32).
// Next to OTHER kernels on the same GPU (several contexts: strip shards on one device, two frames in flight) composite_wave_kernel
// then dropped one lane's contribution from the first of its three sums on about one ray in 10^4 -- same inputs, every kernel alone
// reproducible (tools/probes/dense_stage_isolation.py, profiles/r03_dense_shard_flake.md).  These forms stay inside the VALU:
// in-row DPP steps, then readlane across the four rows.  Fixed association, so results are reproducible by construction.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float old, float v) {      // lanes without a source (or outside ROW_MASK) keep `old`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}

// sum over the wave, every lane gets it: ((row sums by 4 butterflies) r0 + r1) + (r2 + r3)
__device__ __forceinline__ float wave_sum_dpp_f32(float v) {
  v += dpp_f32<0xB1>(0.f, v);       // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(0.f, v);       // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(0.f, v);      // row_half_mirror
  v += dpp_f32<0x140>(0.f, v);      // row_mirror
  const int x = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 48));
  return (r0 + r1) + (r2 + r3);
}

// inclusive scans over the wave (lane i: op of lanes 0..i): four row_shr steps inside each row of 16, then row_bcast:15 into rows 1 / 3
// and row_bcast:31 into rows 2 / 3 (the gfx9 scan of LLVM's AMDGPUAtomicOptimizer)
__device__ __forceinline__ float wave_incl_prod_dpp_f32(float v) {
  v *= dpp_f32<0x111>(1.f, v);
  v *= dpp_f32<0x112>(1.f, v);
  v *= dpp_f32<0x114>(1.f, v);
  v *= dpp_f32<0x118>(1.f, v);
  v *= dpp_f32<0x142, 0xA>(1.f, v);
  v *= dpp_f32<0x143, 0xC>(1.f, v);
  return v;
}
__device__ __forceinline__ float wave_incl_sum_dpp_f32(float v) {
  v += dpp_f32<0x111>(0.f, v);
  v += dpp_f32<0x112>(0.f, v);
  v += dpp_f32<0x114>(0.f, v);
  v += dpp_f32<0x118>(0.f, v);
  v += dpp_f32<0x142, 0xA>(0.f, v);
  v += dpp_f32<0x143, 0xC>(0.f, v);
  return v;
}
// lane i gets lane i - 1's value, lane 0 gets `first` (wave_shr:1)
__device__ __forceinline__ float wave_shift_up1_f32(float first, float v) { return dpp_f32<0x138>(first, v); }
__device__ __forceinline__ float wave_last_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// The same forms for the integer exchanges of the selection / compaction kernels (segment totals, prefix sums of counts): since round 4
// NO kernel of the library goes through ds_bpermute (__shfl*) any more -- tests/test_host_cpu.py greps for it.  profiles/r03_dense_shard_flake.md
// is still without a root cause; staying in the VALU removes the only instruction the failing kernel had that its fixed form has not.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_i32(int old, int v) {      // lanes without a source (or outside ROW_MASK) keep `old`
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false);
}
// sum over the wave, every lane gets it
__device__ __forceinline__ int wave_sum_dpp_i32(int v) {
  v += dpp_i32<0xB1>(0, v);        // quad_perm [1,0,3,2]
  v += dpp_i32<0x4E>(0, v);        // quad_perm [2,3,0,1]
  v += dpp_i32<0x141>(0, v);       // row_half_mirror
  v += dpp_i32<0x140>(0, v);       // row_mirror
  return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
// inclusive prefix sum over the wave (lane i: lanes 0..i); SEG = 32: two independent scans over lanes 0..31 and 32..63
template <int SEG = 64>
__device__ __forceinline__ int wave_incl_sum_dpp_i32(int v) {
  static_assert(SEG == 32 || SEG == 64, "segment = half a wave or the wave");
  v += dpp_i32<0x111>(0, v);       // row_shr:1
  v += dpp_i32<0x112>(0, v);       // row_shr:2
  v += dpp_i32<0x114>(0, v);       // row_shr:4
  v += dpp_i32<0x118>(0, v);       // row_shr:8
  v += dpp_i32<0x142, 0xA>(0, v);  // row_bcast:15 into rows 1 and 3
  if (SEG == 64) v += dpp_i32<0x143, 0xC>(0, v);      // row_bcast:31 into rows 2 and 3
  return v;
}

__device__ __forceinline__ float sigmoidf_dev(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ int mbcnt64(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0));
}

}  // namespace adanerf
