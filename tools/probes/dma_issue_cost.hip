// Probe (not part of the library): what does ISSUING the LDS-DMA weight copies cost a wave that is alone on its SIMD and busy with MFMAs?
// The split-precision sampling engine copies 1 KiB per wave and 6 MFMAs (4 pieces per 24-MFMA chunk); round 6's timing ablation prices the
// issue of those copies at 0.11 of the kernel's 1.31 ms.  Every wave here runs the engine's MFMA stream (A X X, F VALU fillers per MFMA)
// and copies the same bytes per MFMA from an L2-resident stream into its own LDS region (no barrier: issue cost only), in several forms:
//   0  no copies
//   1  4 x buffer_load_dwordx4 ... lds back to back, M0 and the scalar offset re-computed per piece (the round-5 ring)
//   2  4 x dwordx4 back to back, ONE M0 write, pieces addressed by the instruction's immediate offset (tools/probes/lds_dma_offset.hip)
//   3  4 x dwordx4, one M0 write, one piece behind every 6th MFMA
//   (4  8 x dwordx2 does not exist: LDS-DMA sizes are 1, 2, 4, 12 or 16 bytes per lane)
//   5  16 x dword (256 B per instruction), two behind every 3rd MFMA
//   6  as 3, but the piece is issued right BEHIND an MFMA pair (two MFMAs back to back, then the copy)
//   7  as 3 with an s_barrier at the head of every chunk, as the ring has it: the four waves of a workgroup then reach every piece
//      position in the same cycles and their copies queue up behind each other in the CU's one address unit
//   8  as 7, and wave w idles 8 w issue cycles (w x s_nop 7) behind the barrier: the waves' copies no longer coincide
//   9  as 7, 16 w issue cycles
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/dma_issue_cost.hip -o adanerf_amd/bin/dma_issue_cost
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

constexpr int kStreamBytes = 1856 * 1024;      // the packed (hi, lo') sampling network

template <int F>
__device__ __forceinline__ void fillers(float& x, float& y) {
#pragma unroll
  for (int i = 0; i < F; ++i) {
    if (i & 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
    else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(y));
    __builtin_amdgcn_sched_barrier(0);
  }
}
__device__ __forceinline__ void mf(f32x16& c, const u32x4& a, const u32x4& b) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
}

#define PIECE16(dst, voff, soff, IMM)                                                                                      \
  do {                                                                                                                     \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)static_cast<uintptr_t>(dst), 16, voff, soff, IMM, 0);         \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
  } while (0)
#define PIECE4(dst, voff, soff, IMM)                                                                                       \
  do {                                                                                                                     \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)static_cast<uintptr_t>(dst), 4, voff, soff, IMM, 0);          \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
  } while (0)

template <int FORM, int F>
__global__ __launch_bounds__(256) void dma_cost(const char* stream, int iters, float* sink, uint64_t* cycles) {
  __shared__ __attribute__((aligned(16))) char lds[4 * 8 * 4096];      // per wave: 8 slots of 4 KiB
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  u32x4 a[4], b[4];
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 4; ++i) {
      a[k][i] = 0x2c003400u + ((threadIdx.x * 131u + k * 17u + i) & 0x3ffu);
      b[k][i] = (threadIdx.x + k + i) & 1 ? 0u : 0x30002e00u + ((threadIdx.x * 37u + k * 5u + i) & 0x3ffu);
    }
  f32x16 A, X;
  for (int r = 0; r < 16; ++r) A[r] = X[r] = 0.f;
  float fx = threadIdx.x * 1e-9f, fy = 1e-9f;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(stream), 0, kStreamBytes, 0x00020000);
  const uint32_t lds_wave = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds)) + wave * 8 * 4096;
  uint32_t goff = wave * 4096, slot = 0;
  const int v16 = lane * 16, v4 = lane * 4;
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const uint32_t dst = lds_wave + slot * 4096;
#pragma unroll
    for (int s = 0; s < 8; ++s) {      // one chunk: 8 k-steps of 3 MFMAs
      const u32x4 &ah = a[s & 3], &al = a[(s + 1) & 3], &bh = b[s & 3], &bl = b[(s + 2) & 3];
      if ((FORM == 7 || FORM == 8 || FORM == 9) && s == 0) {
        asm volatile("s_barrier" ::: "memory");
        // wave-dependent idle time without C++ control flow (a loop here makes hipcc shuffle the accumulators between register files)
        if (FORM == 8)
          asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 7\n\ts_cmp_eq_u32 %0, 1\n\ts_cbranch_scc1 1f\n\ts_nop 7\n\t"
                       "s_cmp_eq_u32 %0, 2\n\ts_cbranch_scc1 1f\n\ts_nop 7\n1:" ::"s"(wave) : "scc");
        if (FORM == 9)
          asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 15\n\ts_cmp_eq_u32 %0, 1\n\ts_cbranch_scc1 1f\n\ts_nop 15\n\t"
                       "s_cmp_eq_u32 %0, 2\n\ts_cbranch_scc1 1f\n\ts_nop 15\n1:" ::"s"(wave) : "scc");
        __builtin_amdgcn_sched_barrier(0);
      }
      mf(A, ah, bh);
      if (FORM == 1 && s == 0) {
        PIECE16(dst, v16, goff, 0);
        PIECE16(dst + 1024, v16, goff + 1024, 0);
        PIECE16(dst + 2048, v16, goff + 2048, 0);
        PIECE16(dst + 3072, v16, goff + 3072, 0);
      }
      if (FORM == 2 && s == 0) {
        PIECE16(dst, v16, goff, 0);
        PIECE16(dst, v16, goff, 1024);
        PIECE16(dst, v16, goff, 2048);
        PIECE16(dst, v16, goff, 3072);
      }
      if (FORM == 3 || FORM == 7 || FORM == 8 || FORM == 9) {
        if (s == 0) PIECE16(dst, v16, goff, 0);
        if (s == 2) PIECE16(dst, v16, goff, 1024);
        if (s == 4) PIECE16(dst, v16, goff, 2048);
        if (s == 6) PIECE16(dst, v16, goff, 3072);
      }
      if (FORM == 5) {
        if (s == 0) { PIECE4(dst, v4, goff, 0); PIECE4(dst, v4, goff, 256); }
        if (s == 1) { PIECE4(dst, v4, goff, 512); PIECE4(dst, v4, goff, 768); }
        if (s == 2) { PIECE4(dst, v4, goff, 1024); PIECE4(dst, v4, goff, 1280); }
        if (s == 3) { PIECE4(dst, v4, goff, 1536); PIECE4(dst, v4, goff, 1792); }
        if (s == 4) { PIECE4(dst, v4, goff, 2048); PIECE4(dst, v4, goff, 2304); }
        if (s == 5) { PIECE4(dst, v4, goff, 2560); PIECE4(dst, v4, goff, 2816); }
        if (s == 6) { PIECE4(dst, v4, goff, 3072); PIECE4(dst, v4, goff, 3328); }
        if (s == 7) { PIECE4(dst, v4, goff, 3584); PIECE4(dst, v4, goff, 3840); }
      }
      fillers<F>(fx, fy);
      mf(X, ah, bl);
      if (FORM != 6) fillers<F>(fx, fy);
      mf(X, al, bh);
      if (FORM == 6) {
        if (s == 0) PIECE16(dst, v16, goff, 0);
        if (s == 2) PIECE16(dst, v16, goff, 1024);
        if (s == 4) PIECE16(dst, v16, goff, 2048);
        if (s == 6) PIECE16(dst, v16, goff, 3072);
      }
      if (FORM == 6) fillers<2 * F>(fx, fy);
      else fillers<F>(fx, fy);
    }
    goff += 16384;
    if (goff >= kStreamBytes) goff = wave * 4096;
    slot = (slot + 1) & 7;
    if ((it & 63) == 63) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int r = 0; r < 16; ++r) { A[r] *= 1e-6f; X[r] *= 1e-6f; }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) cycles[blockIdx.x * 4 + wave] = t1 - t0;
  float s = A[0] + X[3] + fx + fy;
  if (s == 12345.678f) sink[0] = s;
}

template <int FORM, int F>
static void run(int blocks, const char* stream, float* sink, uint64_t* d_cycles) {
  const int iters = 4000;
  hipLaunchKernelGGL((dma_cost<FORM, F>), dim3(blocks), dim3(256), 0, 0, stream, 100, sink, d_cycles);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((dma_cost<FORM, F>), dim3(blocks), dim3(256), 0, 0, stream, iters, sink, d_cycles);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> c(blocks * 4);
  hipMemcpy(c.data(), d_cycles, c.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : c) sum += static_cast<double>(v);
  const double per = sum / c.size() / (iters * 24.0);
  const double tflops = 2.0 * 32 * 32 * 16 * 24.0 * iters * 4.0 * blocks / ms * 1e-9;
  static const char* names[] = {"no copies", "4 x dwordx4 burst, M0 + offset per piece (round 5)", "4 x dwordx4 burst, one M0, immediate offsets",
                                "4 x dwordx4, one per 6 MFMAs", "8 x dwordx2, one per 3 MFMAs", "16 x dword, two per 3 MFMAs", "4 x dwordx4 behind an MFMA pair", "as 3 + s_barrier per chunk", "as 7 + wave w idles 8 w issue cycles", "as 7 + wave w idles 16 w issue cycles"};
  printf("form %d %-52s %d VALU / MFMA: %6.2f cycles / MFMA  %7.1f TFLOP/s  (%.2f ms)\n", FORM, names[FORM], F, per, tflops, ms);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int F>
static void run_all(int blocks, const char* stream, float* sink, uint64_t* cyc) {
  run<0, F>(blocks, stream, sink, cyc);
  run<1, F>(blocks, stream, sink, cyc);
  run<2, F>(blocks, stream, sink, cyc);
  run<3, F>(blocks, stream, sink, cyc);
  run<5, F>(blocks, stream, sink, cyc);
  run<6, F>(blocks, stream, sink, cyc);
  run<7, F>(blocks, stream, sink, cyc);
  run<8, F>(blocks, stream, sink, cyc);
  run<9, F>(blocks, stream, sink, cyc);
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
  const int blocks = p.multiProcessorCount;
  float* sink; uint64_t* cyc; char* stream;
  hipMalloc(&sink, 64); hipMalloc(&cyc, blocks * 4 * 8); hipMalloc(&stream, kStreamBytes);
  hipMemset(stream, 0x11, kStreamBytes);
  printf("# %s, %d CUs, one 4-wave workgroup per CU; v_mfma_f32_32x32x16_f16 in the engine's order, 1 KiB copied per wave and 6 MFMAs\n", p.gcnArchName, blocks);
  run<0, 0>(blocks, stream, sink, cyc);      // warm-up of the clock
  run_all<0>(blocks, stream, sink, cyc);
  run_all<2>(blocks, stream, sink, cyc);
  run_all<3>(blocks, stream, sink, cyc);
  return 0;
}
