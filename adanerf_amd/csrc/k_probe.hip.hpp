// Measurement kernel, not part of the render path: what dense 16-bit MFMA rate does THIS device sustain, and at which clock?
// Register-only loops of v_mfma_f32_32x32x16_{bf16,f16} (no memory, no LDS; CHAINS independent accumulator chains per wave) for three
// kinds of operands:
//   zero      all operand bits 0                                  -- the least switching activity the data path can have
//   constant  small non-zero values that never change
//   random    full-range random 16-bit values, a fresh pair of operand registers every MFMA (8 pairs cycled): every operand latch and
//             most multiplier inputs toggle between consecutive MFMAs -- the most a network can do to the pipe
//   relu      as random, but the B operand looks like a hidden layer after ReLU: about half of its elements are 0, the rest
//             positive -- what the 256 -> 256 layers of this path feed the pipe (A = random weights)
// The effective shader clock is measured inside the kernel: s_memtime (shader cycles) against s_memrealtime (constant 100 MHz), first
// wave of block 0.  The part holds a power budget (MI355X_MICROARCH.md "DVFS give-back"): the same instruction stream runs at
// ~2.4 GHz on zeros and ~1.7 GHz on random data, so "fraction of 2.5 PFLOP/s" and "fraction of what the silicon sustains on live data" are
// different numbers; bench.py reports both (adanerf_probe_mfma), tools/probes/mfma_peak.hip prints the whole table.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace adanerf {
namespace probe {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

enum { kZero = 0, kConstant = 1, kRandom = 2, kRelu = 3 };

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// two random 16-bit floats with exponents in a sane range (no inf / NaN / subnormals: sign random, exponent field ~ bias +- 3, mantissa random)
__device__ __forceinline__ uint32_t rnd_pair(uint32_t seed, bool f16, bool relu = false) {
  const uint32_t r = mix(seed);
  const uint32_t z = mix(seed ^ 0x5bd1e995u);
  auto one = [&](uint32_t b, bool zero) -> uint32_t {
    if (relu && zero) return 0u;
    const uint32_t sign = relu ? 0u : (b & 0x8000u);
    if (f16) return sign | ((12u + ((b >> 10) & 7u)) << 10) | (b & 0x3ffu);      // fp16: 5-bit exponent, bias 15
    return sign | ((124u + ((b >> 7) & 7u)) << 7) | (b & 0x7fu);                 // bf16: 8-bit exponent, bias 127
  };
  return one(r & 0xffffu, (z & 1u) != 0u) | (one(r >> 16, (z & 2u) != 0u) << 16);
}

template <int CHAINS, bool F16, int MODE>
__global__ __launch_bounds__(256) void mfma_burn(int iters, float* sink, uint64_t* clocks) {
  constexpr int NOP = (MODE == kRandom || MODE == kRelu) ? 8 : 1;
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
        b[k][i] = rnd_pair(0x89abcdeu + threadIdx.x * 64u + blockIdx.x * 16384u + k * 8u + i, F16, MODE == kRelu);
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
    if ((MODE == kRandom || MODE == kRelu) && (it & 255) == 255) {      // keep the accumulators finite over millions of steps
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


// One measurement: `blocks` workgroups of 4 waves (one per SIMD), ~target_ms long (a short calibration launch first).  Returns the
// achieved TFLOP/s and the effective clock in MHz.  Synchronises the stream.
template <int CHAINS, bool F16, int MODE>
inline hipError_t mfma_rate(int blocks, double target_ms, hipStream_t stream, float* sink, uint64_t* d_clocks, double* tflops, double* mhz, double* ms_out = nullptr) {
  hipEvent_t e0, e1;
  hipError_t rc;
  if ((rc = hipEventCreate(&e0)) != hipSuccess) return rc;
  if ((rc = hipEventCreate(&e1)) != hipSuccess) return rc;
  int iters = 400;
  hipLaunchKernelGGL((mfma_burn<CHAINS, F16, MODE>), dim3(blocks), dim3(256), 0, stream, 50, sink, d_clocks);
  for (int rep = 0; rep < 2 && rc == hipSuccess; ++rep) {
    (void)hipEventRecord(e0, stream);
    hipLaunchKernelGGL((mfma_burn<CHAINS, F16, MODE>), dim3(blocks), dim3(256), 0, stream, iters, sink, d_clocks);
    (void)hipEventRecord(e1, stream);
    if ((rc = hipEventSynchronize(e1)) != hipSuccess) break;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    uint64_t ck[2] = {0, 0};
    if ((rc = hipMemcpy(ck, d_clocks, sizeof(ck), hipMemcpyDeviceToHost)) != hipSuccess) break;
    if (rep == 1) {
      const double flop = 2.0 * 32 * 32 * 16 * 8.0 * CHAINS * iters * 4.0 * blocks;
      *tflops = flop / ms * 1e-9;
      *mhz = ck[1] ? 100.0 * static_cast<double>(ck[0]) / static_cast<double>(ck[1]) : 0.0;
      if (ms_out) *ms_out = ms;
    } else {
      iters = static_cast<int>(iters * target_ms / (ms > 0.01 ? ms : 0.01)) + 1;
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

}  // namespace probe
}  // namespace adanerf
