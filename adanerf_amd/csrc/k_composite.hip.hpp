// A7: compositing kernels and the strip de-interleave of the multi-GPU exchange.
// Device code only (gfx950, wave64); part of kernels.hip.hpp.
#pragma once
#include "k_common.hip.hpp"

namespace adanerf {

// ------------------------------------------------------------------------------------------
// A7: compositing (src/nerf_raymarch_common.py:91-144)
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// Optional secondary outputs of the compositing step (src/nerf_raymarch_common.py:137-139 / 60-62): per ray
// depth_map = sum w z (z = world depth of the sample as the sampler placed it) and acc_map = sum w.
struct AuxOut {
  float* depth;               // [R] or null
  float* acc;                 // [R] or null
  const uint32_t* sample_key; // adaptive path: z = ztab[key & 127]; null with dense != 0: the bin is the sample index & 127
  int32_t dense;
  const float* ztab;
};

// disp_map of src/nerf_raymarch_common.py:61 / :138: 1 / max(1e-10, depth_map / sum(weights)), from the two maps above
__global__ __launch_bounds__(256) void disp_map_kernel(const float* __restrict__ depth, const float* __restrict__ acc, int n, float* __restrict__ disp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float q = depth[i] / acc[i];
  disp[i] = 1.0f / ((q != q) ? q : fmaxf(1e-10f, q));      // torch.max propagates the NaN of an empty ray (0 / 0); fmaxf would not
}

// one sample of the front-to-back recurrence (src/nerf_raymarch_common.py:91-144): every product and sum rounds
// to fp32 where the reference's does; returns the sample's weight
__device__ __forceinline__ float composite_step(const float4 v, float wv, int mult_mode, float& cr, float& cg, float& cb, float& T) {
  float al = sigmoidf(v.w);
  if (mult_mode == 1) al = __fmul_rn(al, wv);
  float wt = __fmul_rn(al, T);
  if (mult_mode == 2) wt = __fmul_rn(wt, wv);
  cr = __fadd_rn(cr, __fmul_rn(wt, sigmoidf(v.x)));
  cg = __fadd_rn(cg, __fmul_rn(wt, sigmoidf(v.y)));
  cb = __fadd_rn(cb, __fmul_rn(wt, sigmoidf(v.z)));
  T = __fmul_rn(T, __fadd_rn(__fsub_rn(1.0f, al), 1e-10f));
  return wt;
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
                                                       uchar4* __restrict__ rgba8_out, AuxOut aux) {
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
  if (aux.depth || aux.acc) {                                   // the rarely used path keeps the plain loads
    float dm = 0.f, am = 0.f;
    for (int k = 0; k < c; ++k) {
      const float wt = composite_step(raw[o + k], sample_w[o + k], mult_mode, cr, cg, cb, T);
      dm = __fadd_rn(dm, __fmul_rn(wt, aux.ztab[(aux.sample_key ? aux.sample_key[o + k] : static_cast<uint32_t>(o + k)) & 127u]));
      am = __fadd_rn(am, wt);
    }
    if (aux.depth) aux.depth[r] = dm;
    if (aux.acc) aux.acc[r] = am;
  } else if (staged && ol >= 0 && ol + c <= n) {
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
// transmittance = exclusive product scan of (1 - alpha + 1e-10) across the wave, colour = wave sum (DPP forms of k_common.hip.hpp).
// Coalesced 16-byte loads instead of one thread striding through 2 KiB per ray.
__global__ __launch_bounds__(256) void composite_wave_kernel(const float4* __restrict__ raw, const float* __restrict__ sample_w,
                                                             const int32_t* __restrict__ ray_offsets, const int32_t* __restrict__ counts,
                                                             int n_rays, int mult_mode, float* __restrict__ rgb_out, uchar4* __restrict__ rgba8_out,
                                                             AuxOut aux) {
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
  const float p0 = wave_incl_prod_dpp_f32(lane + 0 < c ? f0 : 1.0f);
  const float tot0 = wave_last_f32(p0);
  const float p1 = wave_incl_prod_dpp_f32(lane + 64 < c ? f1 : 1.0f);
  const float e0 = wave_shift_up1_f32(1.0f, p0), e1 = wave_shift_up1_f32(1.0f, p1);      // exclusive products
  const float w0 = al[0] * e0, w1 = al[1] * (tot0 * e1);
  float cr = w0 * col[0][0] + w1 * col[1][0], cg = w0 * col[0][1] + w1 * col[1][1], cb = w0 * col[0][2] + w1 * col[1][2];
  if (aux.depth || aux.acc) {      // weights of the two samples of this lane (mult_mode 2 scales the weight, not alpha)
    float q0 = w0, q1 = w1;
    if (mult_mode == 2) {
      q0 *= (lane < c) ? sample_w[o + lane] : 0.f;
      q1 *= (lane + 64 < c) ? sample_w[o + lane + 64] : 0.f;
    }
    float dm = q0 * ((lane < c) ? aux.ztab[(aux.sample_key ? aux.sample_key[o + lane] : static_cast<uint32_t>(o + lane)) & 127u] : 0.f) +
               q1 * ((lane + 64 < c) ? aux.ztab[(aux.sample_key ? aux.sample_key[o + lane + 64] : static_cast<uint32_t>(o + lane + 64)) & 127u] : 0.f);
    float am = q0 + q1;
    dm = wave_sum_dpp_f32(dm);
    am = wave_sum_dpp_f32(am);
    if (lane == 0) {
      if (aux.depth) aux.depth[r] = dm;
      if (aux.acc) aux.acc[r] = am;
    }
  }
  cr = wave_sum_dpp_f32(cr);
  cg = wave_sum_dpp_f32(cg);
  cb = wave_sum_dpp_f32(cb);
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
