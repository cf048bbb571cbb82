// Probe (not part of the library): what does the ORDER of dependent MFMAs cost on gfx950 with one wave per SIMD?
// The split-precision sampling engine issues, per k-step, one MFMA into the main accumulator (A) and two into the cross-term
// accumulator (X).  hipcc's schedule of round 5 puts same-accumulator MFMAs next to each other with one or a few other
// instructions in between.  This probe measures shader cycles per v_mfma_f32_32x32x16_f16 (s_memtime, one wave per SIMD,
// every CU busy) for:
//   pattern 0  A A A ...            one chain
//   pattern 1  A B A B ...          two chains, strictly alternating
//   pattern 2  A X X A X X ...      the engine's source order (two chains, 1 : 2)
//   pattern 3  A X Y A X Y ...      three chains round-robin (cross term split into two accumulators)
//   pattern 4  X A X X A X ...      1 : 2 with the main MFMA between the two cross MFMAs of a k-step
//   pattern 5  A f X X ff ...       1 : 2, the two cross MFMAs strictly back to back, all fillers elsewhere (F after A, 2F after X X)
//   pattern 6  A ff X X f ...       the same with 2F after A, F after X X
// each with F = 0 .. 5 independent VALU fillers between consecutive MFMAs (the order is pinned with sched_barrier).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_chain_spacing.hip -o adanerf_amd/bin/mfma_chain_spacing
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// KIND of filler: 0 v_fma_f32 (VALU), 1 s_add_i32 (SALU), 2 s_waitcnt lgkmcnt(15) (always satisfied), 3 s_nop 0,
// 4 v_cvt_pk_f16_f32, 5 v_fma_mix_f32, 6 alternating VALU / SALU
template <int F, int KIND = 0>
__device__ __forceinline__ void fillers(float& x, float& y, int& sx) {
#pragma unroll
  for (int i = 0; i < F; ++i) {
    if (KIND == 0 || (KIND == 6 && !(i & 1))) {
      if (i & 2) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
      else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(y));
    } else if (KIND == 1 || KIND == 6) asm volatile("s_add_i32 %0, %0, 3" : "+s"(sx));
    else if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(15)");
    else if (KIND == 3) asm volatile("s_nop 0");
    else if (KIND == 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(x) : "v"(y));
    else if (KIND == 5) asm volatile("v_fma_mix_f32 %0, %1, %1, %1" : "=v"(x) : "v"(y));
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ void mf(f32x16& c, const u32x4& a, const u32x4& b) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
}

template <int PATTERN, int F, int KIND>
__global__ __launch_bounds__(256) void spacing(int iters, float* sink, uint64_t* cycles) {
  u32x4 a[4], b[4];
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 4; ++i) {
      a[k][i] = 0x2c003400u + ((threadIdx.x * 131u + k * 17u + i) & 0x3ffu);
      b[k][i] = (threadIdx.x + k + i) & 1 ? 0u : 0x30002e00u + ((threadIdx.x * 37u + k * 5u + i) & 0x3ffu);
    }
  f32x16 A, X, Y;
  for (int r = 0; r < 16; ++r) A[r] = X[r] = Y[r] = 0.f;
  float fx = threadIdx.x * 1e-9f, fy = 1e-9f;
  int sx = iters;
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {      // 16 "k-steps" of 3 MFMAs
      const u32x4 &ah = a[s & 3], &al = a[(s + 1) & 3], &bh = b[s & 3], &bl = b[(s + 2) & 3];
      if (PATTERN == 0) { mf(A, ah, bh); fillers<F, KIND>(fx, fy, sx); mf(A, ah, bl); fillers<F, KIND>(fx, fy, sx); mf(A, al, bh); fillers<F, KIND>(fx, fy, sx); }
      if (PATTERN == 1) {
        if (s & 1) { mf(X, ah, bh); fillers<F, KIND>(fx, fy, sx); mf(A, ah, bl); fillers<F, KIND>(fx, fy, sx); mf(X, al, bh); fillers<F, KIND>(fx, fy, sx); }
        else { mf(A, ah, bh); fillers<F, KIND>(fx, fy, sx); mf(X, ah, bl); fillers<F, KIND>(fx, fy, sx); mf(A, al, bh); fillers<F, KIND>(fx, fy, sx); }
      }
      if (PATTERN == 2) { mf(A, ah, bh); fillers<F, KIND>(fx, fy, sx); mf(X, ah, bl); fillers<F, KIND>(fx, fy, sx); mf(X, al, bh); fillers<F, KIND>(fx, fy, sx); }
      if (PATTERN == 3) { mf(A, ah, bh); fillers<F, KIND>(fx, fy, sx); mf(X, ah, bl); fillers<F, KIND>(fx, fy, sx); mf(Y, al, bh); fillers<F, KIND>(fx, fy, sx); }
      if (PATTERN == 4) { mf(X, ah, bl); fillers<F, KIND>(fx, fy, sx); mf(A, ah, bh); fillers<F, KIND>(fx, fy, sx); mf(X, al, bh); fillers<F, KIND>(fx, fy, sx); }
      if (PATTERN == 5) { mf(A, ah, bh); fillers<F, KIND>(fx, fy, sx); mf(X, ah, bl); mf(X, al, bh); fillers<2 * F, KIND>(fx, fy, sx); }
      if (PATTERN == 6) { mf(A, ah, bh); fillers<2 * F, KIND>(fx, fy, sx); mf(X, ah, bl); mf(X, al, bh); fillers<F, KIND>(fx, fy, sx); }
    }
    if ((it & 63) == 63)
      for (int r = 0; r < 16; ++r) { A[r] *= 1e-6f; X[r] *= 1e-6f; Y[r] *= 1e-6f; }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  float s = A[0] + X[3] + Y[5] + fx + fy + sx;
  if (s == 12345.678f) sink[0] = s;
}

template <int PATTERN, int F, int KIND = 0>
static void run(int blocks, float* sink, uint64_t* d_cycles) {
  const int iters = 2000;
  hipLaunchKernelGGL((spacing<PATTERN, F, KIND>), dim3(blocks), dim3(256), 0, 0, 50, sink, d_cycles);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((spacing<PATTERN, F, KIND>), dim3(blocks), dim3(256), 0, 0, iters, sink, d_cycles);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> c(blocks * 4);
  hipMemcpy(c.data(), d_cycles, c.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : c) sum += static_cast<double>(v);
  const double per = sum / c.size() / (iters * 48.0);
  const double tflops = 2.0 * 32 * 32 * 16 * 48.0 * iters * 4.0 * blocks / ms * 1e-9;
  static const char* names[] = {"A A A (one chain)", "A B A B (two chains)", "A X X (engine, round 5)", "A X Y (three chains)", "X A X (1:2, A between)", "A f XX ff (XX back to back)", "A ff XX f (XX back to back)"};
  static const char* kinds[] = {"v_fma_f32", "s_add_i32", "s_waitcnt (satisfied)", "s_nop 0", "v_cvt_pk_f16_f32", "v_fma_mix_f32", "VALU / SALU alternating"};
  printf("pattern %d %-26s fillers %d x %-24s: %6.2f cycles / MFMA   %7.1f TFLOP/s  (%.2f ms)\n", PATTERN, names[PATTERN], F, kinds[KIND], per, tflops, ms);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int PATTERN>
static void run_all(int blocks, float* sink, uint64_t* d_cycles) {
  run<PATTERN, 0>(blocks, sink, d_cycles);
  run<PATTERN, 1>(blocks, sink, d_cycles);
  run<PATTERN, 2>(blocks, sink, d_cycles);
  run<PATTERN, 3>(blocks, sink, d_cycles);
  run<PATTERN, 4>(blocks, sink, d_cycles);
  run<PATTERN, 5>(blocks, sink, d_cycles);
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
  const int blocks = p.multiProcessorCount;
  float* sink; uint64_t* cyc;
  hipMalloc(&sink, 64); hipMalloc(&cyc, blocks * 4 * 8);
  printf("# %s, %d CUs, one 4-wave workgroup per CU (one wave per SIMD); v_mfma_f32_32x32x16_f16, ReLU-like B operands\n", p.gcnArchName, blocks);
  run_all<0>(blocks, sink, cyc);
  run_all<1>(blocks, sink, cyc);
  run_all<2>(blocks, sink, cyc);
  run_all<3>(blocks, sink, cyc);
  run_all<4>(blocks, sink, cyc);
  run_all<5>(blocks, sink, cyc);
  run_all<6>(blocks, sink, cyc);
  // which instructions take a filler slot beside the MFMAs (pattern 2, the engine's order)
  run<2, 4, 1>(blocks, sink, cyc); run<2, 6, 1>(blocks, sink, cyc); run<2, 8, 1>(blocks, sink, cyc);
  run<2, 4, 2>(blocks, sink, cyc); run<2, 6, 2>(blocks, sink, cyc); run<2, 8, 2>(blocks, sink, cyc);
  run<2, 4, 3>(blocks, sink, cyc); run<2, 6, 3>(blocks, sink, cyc); run<2, 8, 3>(blocks, sink, cyc);
  run<2, 3, 4>(blocks, sink, cyc); run<2, 4, 4>(blocks, sink, cyc); run<2, 5, 4>(blocks, sink, cyc);
  run<2, 3, 5>(blocks, sink, cyc); run<2, 4, 5>(blocks, sink, cyc); run<2, 5, 5>(blocks, sink, cyc);
  run<2, 4, 6>(blocks, sink, cyc); run<2, 6, 6>(blocks, sink, cyc); run<2, 8, 6>(blocks, sink, cyc);
  run<2, 4, 0>(blocks, sink, cyc);      // again, last: the clock of the first rows is still ramping
  return 0;
}
