// What dense 16-bit MFMA rate does THIS MI355X sustain, and at which clock?  Register-only loops of v_mfma_f32_32x32x16_{bf16,f16}
// (no memory, no LDS; one wave per SIMD x 4 independent accumulator chains, or two waves per SIMD), for three kinds of operands and two
// durations, with the effective shader clock read beside every figure:
//   zero      all operand bits 0                                  -- the least switching activity the data path can have
//   constant  small non-zero values that never change             -- what round 3's version of this probe ran
//   random    full-range random 16-bit values, a fresh pair of operand registers every MFMA (8 pairs cycled): every operand
//             latch and most multiplier inputs toggle between consecutive MFMAs -- what a real network does to the pipe
//   burst     ~2 ms      sustained  ~0.5 s
// The clock is measured inside the kernel: s_memtime (shader cycles) against s_memrealtime (constant 100 MHz), first wave of block 0.
// Per line: TFLOP/s, effective MHz, and the fraction of the pipe's issue slots used = TFLOP/s / (CUs x 4 SIMDs x 1024 FLOP/cycle x clock)
// -- 1.0 means an MFMA retires every 32 cycles on every SIMD, whatever the clock.
// The guide's "2495 TF measured" and this kernel's shading loop at 2.0 GHz are then the same pipe at different clocks (DVFS: the part
// holds a power budget, MI355X_MICROARCH.md "DVFS give-back").
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o adanerf_amd/bin/mfma_peak && adanerf_amd/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

enum { kZero = 0, kConstant = 1, kRandom = 2 };

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// two random 16-bit floats with exponents in a sane range (no inf / NaN / subnormals: sign random, exponent field ~ bias +- 3, mantissa random)
__device__ __forceinline__ uint32_t rnd_pair(uint32_t seed, bool f16) {
  const uint32_t r = mix(seed);
  auto one = [&](uint32_t b) -> uint32_t {
    if (f16) return (b & 0x8000u) | ((12u + ((b >> 10) & 7u)) << 10) | (b & 0x3ffu);      // fp16: 5-bit exponent, bias 15
    return (b & 0x8000u) | ((124u + ((b >> 7) & 7u)) << 7) | (b & 0x7fu);                 // bf16: 8-bit exponent, bias 127
  };
  return one(r & 0xffffu) | (one(r >> 16) << 16);
}

template <int CHAINS, bool F16, int MODE>
__global__ __launch_bounds__(256) void burn(int iters, float* sink, uint64_t* clocks) {
  constexpr int NOP = MODE == kRandom ? 8 : 1;
  u32x4 a[NOP], b[NOP];
  for (int k = 0; k < NOP; ++k)
    for (int i = 0; i < 4; ++i) {
      if (MODE == kZero) a[k][i] = b[k][i] = 0u;
      else if (MODE == kConstant) {
        // the values of round 3's probe: 0.001 (t + i), 0.002 (t - i) as 16-bit floats
        const float v0 = 0.001f * (threadIdx.x + 2 * i), v1 = 0.001f * (threadIdx.x + 2 * i + 1);
        const float w0 = 0.002f * (static_cast<float>(threadIdx.x) - 2 * i), w1 = 0.002f * (static_cast<float>(threadIdx.x) - 2 * i - 1);
        if (F16) {
          a[k][i] = static_cast<uint32_t>(__builtin_bit_cast(uint16_t, static_cast<_Float16>(v0))) | (static_cast<uint32_t>(__builtin_bit_cast(uint16_t, static_cast<_Float16>(v1))) << 16);
          b[k][i] = static_cast<uint32_t>(__builtin_bit_cast(uint16_t, static_cast<_Float16>(w0))) | (static_cast<uint32_t>(__builtin_bit_cast(uint16_t, static_cast<_Float16>(w1))) << 16);
        } else {
          a[k][i] = (__builtin_bit_cast(uint32_t, v0) >> 16) | (__builtin_bit_cast(uint32_t, v1) & 0xffff0000u);
          b[k][i] = (__builtin_bit_cast(uint32_t, w0) >> 16) | (__builtin_bit_cast(uint32_t, w1) & 0xffff0000u);
        }
      } else {
        a[k][i] = rnd_pair(0x1234567u + threadIdx.x * 64u + blockIdx.x * 16384u + k * 8u + i, F16);
        b[k][i] = rnd_pair(0x89abcdeu + threadIdx.x * 64u + blockIdx.x * 16384u + k * 8u + i, F16);
      }
    }
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  uint64_t t0 = 0, r0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    t0 = __builtin_readcyclecounter();      // s_memtime: shader clock
    r0 = wall_clock64();                    // s_memrealtime: 100 MHz
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        const int k = (u * CHAINS + c) % NOP;
        if (F16) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[k]), __builtin_bit_cast(f16x8, b[k]), acc[c], 0, 0, 0);
        else acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[k]), __builtin_bit_cast(bf16x8, b[k]), acc[c], 0, 0, 0);
      }
    if (MODE == kRandom && (it & 255) == 255) {      // keep the accumulators finite over millions of steps
      for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] *= 1.0e-6f;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clocks[0] = __builtin_readcyclecounter() - t0;
    clocks[1] = wall_clock64() - r0;
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][7];
  if (s == 12345.678f) sink[0] = s;
}

template <int CHAINS, bool F16, int MODE>
void run(int waves_per_simd, int cus, float* sink, uint64_t* d_clocks, double target_ms) {
  const int blocks = cus * waves_per_simd;          // 256 threads = 4 waves = one per SIMD
  int iters = 400;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((burn<CHAINS, F16, MODE>), dim3(blocks), dim3(256), 0, 0, 50, sink, d_clocks);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((burn<CHAINS, F16, MODE>), dim3(blocks), dim3(256), 0, 0, iters, sink, d_clocks);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    uint64_t ck[2] = {0, 0};
    hipMemcpy(ck, d_clocks, sizeof(ck), hipMemcpyDeviceToHost);
    const double flop = 2.0 * 32 * 32 * 16 * 8.0 * CHAINS * iters * 4.0 * blocks;
    const double mhz = ck[1] ? 100.0 * static_cast<double>(ck[0]) / static_cast<double>(ck[1]) : 0.0;
    const double tf = flop / ms * 1e-9;
    if (rep == 1)
      printf("%-4s %-8s chains %d waves/SIMD %d %9.1f ms  %7.0f TFLOP/s  %5.0f MHz  issue-slot use %.3f  (of 2500: %.3f)\n", F16 ? "f16" : "bf16",
             MODE == kZero ? "zero" : MODE == kConstant ? "constant" : "random", CHAINS, waves_per_simd, ms, tf, mhz,
             mhz > 0 ? tf * 1e12 / (cus * 4.0 * 1024.0 * mhz * 1e6) : 0.0, tf / 2500.0);
    if (rep == 0) iters = static_cast<int>(iters * target_ms / (ms > 0.01 ? ms : 0.01)) + 1;
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  float* sink;
  uint64_t* d_clocks;
  hipMalloc(&sink, 64);
  hipMalloc(&d_clocks, 64);
  const int cus = p.multiProcessorCount;
  printf("%s, %d CUs, nominal clock %d MHz; peak at nominal = %d x 4 x 1024 FLOP/cycle x clock = %.0f TFLOP/s\n", p.gcnArchName, cus, p.clockRate / 1000,
         cus, cus * 4.0 * 1024.0 * (p.clockRate * 1e3) * 1e-12);
  for (double target : {2.0, 500.0}) {
    printf("-- %s (~%.0f ms per launch)\n", target < 10 ? "burst" : "sustained", target);
    run<4, false, kZero>(1, cus, sink, d_clocks, target);
    run<4, false, kConstant>(1, cus, sink, d_clocks, target);
    run<4, false, kRandom>(1, cus, sink, d_clocks, target);
    run<4, true, kRandom>(1, cus, sink, d_clocks, target);
    run<2, false, kRandom>(1, cus, sink, d_clocks, target);      // the shading kernel's form: two chains alternating, one wave per SIMD
    run<4, false, kRandom>(2, cus, sink, d_clocks, target);
  }
  return 0;
}
