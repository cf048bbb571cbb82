// vanilla NeRF with hierarchical sampling (inFeatures [RayMarchFromPoses, RayMarchFromCoarse]) -- the three
// small kernels around the two shading-net passes: camera rays, uniform coarse samples, and the fine sampler
// (coarse weights -> pdf over the interval mid-points -> inverse CDF -> merge with the coarse depths).
// Reference: src/features.py:380-480 (RayMarchFromPoses over LinearlySpacedZNearZFar, src/nerf_raymarch_common.py:295-331),
// src/features.py:640-672 (RayMarchFromCoarse.batch), nerf_raw2outputs / nerf_sample_pdf (src/nerf_raymarch_common.py:19-68,
// 160-192); viewer: updateRayMarchCoarse / updateRayMarchFromCoarse (adanerf_real_time_viewer/include/cuda/adanerf_cuda_kernels.cuh:60-74).
// Device code only (gfx950, wave64); part of kernels.hip.hpp.
#pragma once
#include "k_common.hip.hpp"

namespace adanerf {

constexpr int kMaxCoarse = 128;      // coarse samples per ray (their depths are a table, like the 128 bins)
constexpr int kFineRaysPerBlock = 64;

// Rays as RayMarchFromPoses makes them without a SpherePosDir in front (src/features.py:417-428): origin = the camera
// position, direction = R d (not normalised).  [n,8] = (origin.xyz, 0, dir.xyz, 0)
__global__ __launch_bounds__(256) void camera_rays_kernel(RayGenParams g, int first_ray, int n_rays, float* __restrict__ rays_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rays) return;
  int col, row;
  ray_pixel(g, first_ray + i, &col, &row);
  float nds[3], p[3];
  gen_ray(g, col, row, nds, p);
  float4* r = reinterpret_cast<float4*>(rays_out + static_cast<size_t>(i) * 8);
  float ro[3] = {g.pos[0], g.pos[1], g.pos[2]}, rd[3] = {nds[0], nds[1], nds[2]};
  if (g.use_ndc) ndc_ray(g, g.pos, nds, ro, rd);      // src/features.py:429-431: everything downstream works on the NDC ray
  r[0] = make_float4(ro[0], ro[1], ro[2], 0.f);
  r[1] = make_float4(rd[0], rd[1], rd[2], 0.f);
}

// n samples per ray at table depths: key = ray << 7 | k (the shading kernel reads the depth table by k)
__global__ __launch_bounds__(256) void uniform_sample_kernel(int n_rays, int n, int32_t* __restrict__ ray_offsets, int32_t* __restrict__ counts,
                                                             uint32_t* __restrict__ sample_key, int32_t* __restrict__ total) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i == 0) *total = n_rays * n;
  if (i >= static_cast<int64_t>(n_rays) * n) return;
  const int r = static_cast<int>(i / n), k = static_cast<int>(i - static_cast<int64_t>(r) * n);
  sample_key[i] = (static_cast<uint32_t>(r) << 7) | static_cast<uint32_t>(k);
  if (k == 0) {
    ray_offsets[r] = r * n;
    counts[r] = n;
  }
}

// RayMarchFromCoarse.batch for one ray per thread, in the reference's order of operations:
//   weights_k = alpha_k prod_{j<k} (1 - alpha_j + 1e-10), alpha_k = 1 - exp(-relu(raw_a) (z_{k+1} - z_k) |d|), last interval 1e10
//   pdf over weights[1 .. nc-2] + 1e-5, cdf over the nc-1 mid-points, u = linspace(0, 1, nf), searchsorted(right) + lerp,
//   then the nc coarse and nf new depths merged in ascending order.
// torch's CPU cumprod / cumsum accumulate in double and store float; so does this kernel (one thread, sequential).
__global__ __launch_bounds__(kFineRaysPerBlock) void fine_sample_kernel(const float4* __restrict__ raw_coarse, const float* __restrict__ ztab,
                                                                        const float* __restrict__ rays, int n_rays, int nc, int nf,
                                                                        int32_t* __restrict__ ray_offsets, int32_t* __restrict__ counts,
                                                                        uint32_t* __restrict__ sample_key, float* __restrict__ sample_z,
                                                                        int32_t* __restrict__ total) {
  __shared__ float cdf_s[kFineRaysPerBlock][kMaxCoarse + 1];      // [thread][nc - 1 entries]; +1: odd stride, no bank conflicts
  __shared__ float zc[kMaxCoarse];
  const int r = blockIdx.x * kFineRaysPerBlock + threadIdx.x;
  for (int k = threadIdx.x; k < nc; k += kFineRaysPerBlock) zc[k] = ztab[k];
  if (blockIdx.x == 0 && threadIdx.x == 0) *total = n_rays * (nc + nf);
  __syncthreads();
  if (r >= n_rays) return;
  float* cdf = cdf_s[threadIdx.x];
  const float4 d4 = reinterpret_cast<const float4*>(rays + static_cast<size_t>(r) * 8)[1];
  const float dn = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d4.x, d4.x), __fmul_rn(d4.y, d4.y)), __fmul_rn(d4.z, d4.z)));
  // pass 1: weights of the interior samples into cdf[0 .. nc-3] (unnormalised), their sum
  const float4* raw = raw_coarse + static_cast<size_t>(r) * nc;
  double T = 1.0;
  float wsum = 0.f;     // torch.sum(weights, -1): float accumulation is what the vectorised CPU sum does for 62 values within 1 ulp
  {
    double s = 0.0;
    for (int k = 0; k < nc; ++k) {
      const float dist = __fmul_rn((k + 1 < nc) ? __fsub_rn(zc[k + 1], zc[k]) : 1e10f, dn);
      const float al = __fsub_rn(1.0f, expf(-__fmul_rn(fmaxf(raw[k].w, 0.f), dist)));
      const float wt = __fmul_rn(al, static_cast<float>(T));
      T *= static_cast<double>(__fadd_rn(__fsub_rn(1.0f, al), 1e-10f));
      if (k >= 1 && k + 1 < nc) {
        const float wk = __fadd_rn(wt, 1e-5f);
        cdf[k - 1] = wk;
        s += static_cast<double>(wk);
      }
    }
    wsum = static_cast<float>(s);
  }
  // pass 2: cdf[0] = 0, cdf[j] = cumsum(pdf)[j-1]  (nc - 1 entries, over the nc - 1 mid-points)
  {
    double acc = 0.0;
    float prev = 0.f;
    for (int j = 0; j < nc - 2; ++j) {
      const float pdf = cdf[j] / wsum;
      cdf[j] = prev;
      acc += static_cast<double>(pdf);
      prev = static_cast<float>(acc);
    }
    cdf[nc - 2] = prev;
  }
  const int nb = nc - 1;                       // number of bins edges = cdf entries
  // pass 3: walk the nf new depths (ascending in u) and the nc coarse depths together
  const size_t o = static_cast<size_t>(r) * (nc + nf);
  int ic = 0, out = 0;
  for (int j = 0; j < nf; ++j) {
    // torch.linspace(0, 1, nf): step = 1 / (nf - 1); lower half from the start, upper half from the end
    float u;
    if (nf == 1) u = 0.f;
    else {
      const float step = 1.0f / static_cast<float>(nf - 1);
      u = (j < nf / 2) ? __fmul_rn(step, static_cast<float>(j)) : __fsub_rn(1.0f, __fmul_rn(step, static_cast<float>(nf - 1 - j)));
    }
    int lo = 0, hi = nb;                       // searchsorted(cdf, u, right=True)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= u) lo = mid + 1;
      else hi = mid;
    }
    const int below = max(lo - 1, 0), above = min(lo, nb - 1);
    const float c0 = cdf[below], c1 = cdf[above];
    float denom = __fsub_rn(c1, c0);
    denom = denom < 1e-5f ? 1.0f : denom;
    const float t = __fsub_rn(u, c0) / denom;
    const float b0 = __fmul_rn(0.5f, __fadd_rn(zc[below + 1], zc[below])), b1 = __fmul_rn(0.5f, __fadd_rn(zc[above + 1], zc[above]));
    const float zf = __fadd_rn(b0, __fmul_rn(t, __fsub_rn(b1, b0)));
    while (ic < nc && zc[ic] <= zf) sample_z[o + out++] = zc[ic++];
    sample_z[o + out++] = zf;
  }
  while (ic < nc) sample_z[o + out++] = zc[ic++];
  for (int k = 0; k < nc + nf; ++k) sample_key[o + k] = static_cast<uint32_t>(r) << 7;
  ray_offsets[r] = r * (nc + nf);
  counts[r] = nc + nf;
}

}  // namespace adanerf
