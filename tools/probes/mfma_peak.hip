// Sustained dense MFMA rate of this MI355X: register-only loops of v_mfma_f32_32x32x16_bf16 (no memory, no LDS),
// W waves per SIMD, 4 independent accumulators per wave.  Prints achieved TFLOP/s for a ~0.5 s run per configuration,
// so the figure includes whatever clock the part settles at under a pure matrix load.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS, bool F16>
__global__ __launch_bounds__(256) void burn(int iters, float* sink) {
  bf16x8 a, b;
  f16x8 ah, bh;
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)(0.001f * (threadIdx.x + i));
    b[i] = (__bf16)(0.002f * (threadIdx.x - i));
    ah[i] = (_Float16)(0.001f * (threadIdx.x + i));
    bh[i] = (_Float16)(0.002f * (threadIdx.x - i));
  }
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c)
        acc[c] = F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[c], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][7];
  if (s == 12345.678f) sink[0] = s;
}

template <int CHAINS, bool F16>
void run(int waves_per_simd, int cus, float* sink) {
  const int blocks = cus * waves_per_simd;          // 256 threads = 4 waves = one per SIMD
  int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((burn<CHAINS, F16>), dim3(blocks), dim3(256), 0, 0, 200, sink);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((burn<CHAINS, F16>), dim3(blocks), dim3(256), 0, 0, iters, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 32 * 32 * 16 * 8.0 * CHAINS * iters * 4.0 * blocks;
    if (rep == 1)
      printf("%s chains %d waves/SIMD %d: %.1f ms, %.0f TFLOP/s\n", F16 ? "f16 " : "bf16", CHAINS, waves_per_simd, ms, flop / ms * 1e-9);
    if (rep == 0) iters = static_cast<int>(iters * 500.0 / (ms > 1 ? ms : 1));   // aim at ~0.5 s
  }
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  float* sink;
  hipMalloc(&sink, 64);
  printf("%s, %d CUs, nominal clock %d MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
  run<1, false>(1, p.multiProcessorCount, sink);
  run<4, false>(1, p.multiProcessorCount, sink);
  run<1, false>(2, p.multiProcessorCount, sink);
  run<4, false>(2, p.multiProcessorCount, sink);
  run<4, true>(2, p.multiProcessorCount, sink);
  return 0;
}
