// See launch_f32.hpp: the fp32-MFMA kernels, compiled WITHOUT -amdgpu-mfma-vgpr-form.
#include "launch_f32.hpp"

#include "k_mlp16.hip.hpp"     // shade_mlp32_kernel (shares ShadeArgs / load_sample with the 16-bit shading kernel)
#include "k_mlp_f32.hip.hpp"   // sample_mlp_kernel

namespace adanerf {

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

}  // namespace adanerf
