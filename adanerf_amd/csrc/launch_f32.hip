// See launch_f32.hpp: the fp32-MFMA kernels, compiled WITHOUT -amdgpu-mfma-vgpr-form.
#include "launch_f32.hpp"

#include "k_mlp16.hip.hpp"     // shade_mlp32_kernel (shares ShadeArgs / load_sample with the 16-bit shading kernel)
#include "k_mlp_f32.hip.hpp"   // sample_mlp_kernel
#include "k_generic_f32.hip.hpp"
#include "k_probe.hip.hpp"

namespace adanerf {

hipError_t probe_mfma_rate(int operands, bool f16, double target_ms, int compute_units, hipStream_t stream, double* tflops, double* mhz) {
  using namespace probe;
  float* sink = nullptr;
  uint64_t* clocks = nullptr;
  hipError_t rc = hipMalloc(reinterpret_cast<void**>(&sink), 64);
  if (rc != hipSuccess) return rc;
  if ((rc = hipMalloc(reinterpret_cast<void**>(&clocks), 64)) != hipSuccess) {
    (void)hipFree(sink);
    return rc;
  }
#define ADN_PROBE(F16v, MODEv) rc = mfma_rate<2, F16v, MODEv>(compute_units, target_ms, stream, sink, clocks, tflops, mhz)
#define ADN_PROBE_M(F16v)                          \
  do {                                             \
    if (operands == kZero) ADN_PROBE(F16v, kZero); \
    else if (operands == kConstant) ADN_PROBE(F16v, kConstant); \
    else if (operands == kRandom) ADN_PROBE(F16v, kRandom);     \
    else ADN_PROBE(F16v, kRelu);                   \
  } while (0)
  if (f16) ADN_PROBE_M(true);
  else ADN_PROBE_M(false);
#undef ADN_PROBE_M
#undef ADN_PROBE
  (void)hipFree(sink);
  (void)hipFree(clocks);
  return rc;
}

hipError_t launch_sample_mlp_f32(const SampleArgs& a, bool full, unsigned grid, hipStream_t stream) {
  if (full) hipLaunchKernelGGL((sample_mlp_kernel<10, 4>), dim3(grid), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((sample_mlp_kernel<2, 2>), dim3(grid), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t shade_mlp_f32_grid(int compute_units, int* grid) {
  int per_cu = 0;
  const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, shade_mlp32_kernel<10, 4>, 256, 0);
  if (e != hipSuccess) return e;
  *grid = (per_cu < 1 ? 1 : per_cu) * compute_units;
  return hipSuccess;
}

hipError_t launch_shade_mlp_f32(const ShadeArgs& a, int grid, hipStream_t stream) {
  hipLaunchKernelGGL((shade_mlp32_kernel<10, 4>), dim3(grid), dim3(256), 0, stream, a);
  return hipGetLastError();
}


hipError_t launch_sample_mlp_gen(const SampleArgs& a, const GenericTopo& t, int enc, int width, unsigned grid, hipStream_t stream) {
#define ADN_GEN_S(W)                                                                                                             \
  if (width == W) {                                                                                                              \
    if (enc == kEnc10_4) hipLaunchKernelGGL((sample_mlp_gen_kernel<10, 4, W>), dim3(grid), dim3(256), 0, stream, a, t);          \
    else if (enc == kEnc2_2) hipLaunchKernelGGL((sample_mlp_gen_kernel<2, 2, W>), dim3(grid), dim3(256), 0, stream, a, t);       \
    else hipLaunchKernelGGL((sample_mlp_gen_kernel<kMaxBands, kMaxBands, W>), dim3(grid), dim3(256), 0, stream, a, t);           \
    return hipGetLastError();                                                                                                    \
  }
  ADN_GEN_S(64) ADN_GEN_S(128) ADN_GEN_S(256)
#undef ADN_GEN_S
  return hipErrorInvalidValue;
}

hipError_t shade_mlp_gen_grid(int compute_units, int enc, int width, int* grid) {
  int per_cu = 0;
  hipError_t e = hipErrorInvalidValue;
#define ADN_GEN_G(W)                                                                                                             \
  if (width == W) e = enc == kEnc10_4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, shade_mlp32_gen_kernel<10, 4, W>, 256, 0)   \
                                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, shade_mlp32_gen_kernel<kMaxBands, kMaxBands, W>, 256, 0);
  ADN_GEN_G(64) ADN_GEN_G(128) ADN_GEN_G(256)
#undef ADN_GEN_G
  if (e != hipSuccess) return e;
  *grid = (per_cu < 1 ? 1 : per_cu) * compute_units;
  return hipSuccess;
}

hipError_t launch_shade_mlp_gen(const ShadeArgs& a, const GenericTopo& t, int enc, int width, int grid, hipStream_t stream) {
#define ADN_GEN_L(W)                                                                                                             \
  if (width == W) {                                                                                                              \
    if (enc == kEnc10_4) hipLaunchKernelGGL((shade_mlp32_gen_kernel<10, 4, W>), dim3(grid), dim3(256), 0, stream, a, t);         \
    else hipLaunchKernelGGL((shade_mlp32_gen_kernel<kMaxBands, kMaxBands, W>), dim3(grid), dim3(256), 0, stream, a, t);          \
    return hipGetLastError();                                                                                                    \
  }
  ADN_GEN_L(64) ADN_GEN_L(128) ADN_GEN_L(256)
#undef ADN_GEN_L
  return hipErrorInvalidValue;
}

}  // namespace adanerf
