// Launchers of the two kernels that run on the fp32 MFMA engine (v_mfma_f32_32x32x2_f32).  They live in their own
// translation unit (launch_f32.hip) because the rest of the library is compiled with -mllvm -amdgpu-mfma-vgpr-form=1
// (MFMA accumulators in architectural VGPRs: good for the 16-bit kernels' VALU epilogues, 1.48 -> 1.35 ms in the
// split-precision sampling kernel) and that option slows these two down (fp32 sampling 5.3 -> 9.6 ms, fp32 shading
// 60 -> 67 ms): their accumulators belong in AGPRs.
#pragma once
#include <hip/hip_runtime.h>

namespace adanerf {
struct SampleArgs;
struct ShadeArgs;
struct GenericTopo;

// sample_mlp_kernel<10,4> (full = true) or <2,2>; grid = ceil(n_rays / 128) workgroups of 256 threads
hipError_t launch_sample_mlp_f32(const SampleArgs& a, bool full, unsigned grid, hipStream_t stream);
// persistent grid of shade_mlp32_kernel<10,4> for a device with `compute_units` CUs
hipError_t shade_mlp_f32_grid(int compute_units, int* grid);
hipError_t launch_shade_mlp_f32(const ShadeArgs& a, int grid, hipStream_t stream);

// Generic-topology kernels (k_generic_f32.hip.hpp): width 64 / 128 / 256, run-time depth / skip / raySampleInput.
// enc: slot layout of the positional encodings the network was packed with -- kEnc10_4, kEnc2_2 (sampling nets only) or
// kEncMax (the catch-all kMaxBands-band layout: any posEncArgs).  hipErrorInvalidValue for a width / layout without an instantiation.
enum { kEnc10_4 = 0, kEnc2_2 = 1, kEncMax = 2 };
hipError_t launch_sample_mlp_gen(const SampleArgs& a, const GenericTopo& t, int enc, int width, unsigned grid, hipStream_t stream);
hipError_t shade_mlp_gen_grid(int compute_units, int enc, int width, int* grid);
hipError_t launch_shade_mlp_gen(const ShadeArgs& a, const GenericTopo& t, int enc, int width, int grid, hipStream_t stream);

// Measurement hook (k_probe.hip.hpp; include/adanerf_hip.h adanerf_probe_mfma): register-only v_mfma_f32_32x32x16_{bf16,f16} loops on every
// CU for ~target_ms, operands 0 zero / 1 constant / 2 random / 3 relu-like, two accumulator chains per wave and one wave per SIMD (the
// shading kernel's form).  Synchronises the stream.
hipError_t probe_mfma_rate(int operands, bool f16, double target_ms, int compute_units, hipStream_t stream, double* tflops, double* mhz);
}  // namespace adanerf
