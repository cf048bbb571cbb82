// Prints the table behind adanerf_amd/csrc/k_probe.hip.hpp: sustained dense 16-bit MFMA rate of this MI355X for zero / constant / random
// operands, a ~2 ms burst and a ~0.5 s run each, with the effective clock and the fraction of the pipe's issue slots used
// (= TFLOP/s / (CUs x 4 SIMDs x 1024 FLOP/cycle x clock): 1.0 means an MFMA retires every 32 cycles on every SIMD, whatever the clock).
//   hipcc --offload-arch=gfx950 -O3 -I adanerf_amd/csrc tools/probes/mfma_peak.hip -o adanerf_amd/bin/mfma_peak && adanerf_amd/bin/mfma_peak
#include <cstdio>

#include "k_probe.hip.hpp"

using namespace adanerf::probe;

template <int CHAINS, bool F16, int MODE>
void run(int waves_per_simd, int cus, float* sink, uint64_t* d_clocks, double target_ms) {
  double tf = 0, mhz = 0, ms = 0;
  if (mfma_rate<CHAINS, F16, MODE>(cus * waves_per_simd, target_ms, 0, sink, d_clocks, &tf, &mhz, &ms) != hipSuccess) {
    printf("launch failed\n");
    return;
  }
  printf("%-4s %-8s chains %d waves/SIMD %d %9.1f ms  %7.0f TFLOP/s  %5.0f MHz  issue-slot use %.3f  (of 2500: %.3f)\n", F16 ? "f16" : "bf16",
         MODE == kZero ? "zero" : MODE == kConstant ? "constant" : MODE == kRandom ? "random" : "relu", CHAINS, waves_per_simd, ms, tf, mhz,
         mhz > 0 ? tf * 1e12 / (cus * 4.0 * 1024.0 * mhz * 1e6) : 0.0, tf / 2500.0);
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
  float* sink;
  uint64_t* d_clocks;
  if (hipMalloc(&sink, 64) != hipSuccess || hipMalloc(&d_clocks, 64) != hipSuccess) return 1;
  const int cus = p.multiProcessorCount;
  printf("%s, %d CUs, nominal clock %d MHz; peak at nominal = %d x 4 x 1024 FLOP/cycle x clock = %.0f TFLOP/s\n", p.gcnArchName, cus, p.clockRate / 1000,
         cus, cus * 4.0 * 1024.0 * (p.clockRate * 1e3) * 1e-12);
  for (double target : {2.0, 500.0}) {
    printf("-- %s (~%.0f ms per launch)\n", target < 10 ? "burst" : "sustained", target);
    run<4, false, kZero>(1, cus, sink, d_clocks, target);
    run<4, false, kConstant>(1, cus, sink, d_clocks, target);
    run<4, false, kRandom>(1, cus, sink, d_clocks, target);
    run<4, true, kRandom>(1, cus, sink, d_clocks, target);
    run<4, false, kRelu>(1, cus, sink, d_clocks, target);
    run<2, false, kRelu>(1, cus, sink, d_clocks, target);        // the shading kernel's form: two chains alternating, one wave per SIMD
    run<2, false, kRandom>(1, cus, sink, d_clocks, target);
    run<4, false, kRandom>(2, cus, sink, d_clocks, target);
  }
  return 0;
}
