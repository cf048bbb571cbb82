// Prints the table behind adanerf_amd/csrc/k_probe.hip.hpp: sustained dense 16-bit MFMA rate of this MI355X for zero / constant / random
// operands, a ~2 ms burst and a ~0.5 s run each, with the effective clock and the fraction of the pipe's issue slots used
// (= TFLOP/s / (CUs x 4 SIMDs x 1024 FLOP/cycle x clock): 1.0 means an MFMA retires every 32 cycles on every SIMD, whatever the clock).
//   hipcc --offload-arch=gfx950 -O3 -I adanerf_amd/csrc tools/probes/mfma_peak.hip -o adanerf_amd/bin/mfma_peak && adanerf_amd/bin/mfma_peak
#include <cstdio>

#include "k_probe.hip.hpp"

using namespace adanerf::probe;

// The shading kernel's operand delivery around the same MFMA stream (round 4, DESIGN 3.1 "pinned ceiling"): one wave per SIMD, two
// accumulator chains (= two 32-sample blocks), B operands ReLU-like in registers, and
//   LDS    every A fragment (1 KiB per wave) read from LDS with ds_read_b128, one read per TWO MFMAs, requested 4 fragments ahead;
//   DMA    + every wave copies 1 KiB global -> LDS (buffer_load ... lds, L2-resident source) per 8 MFMAs -- the weight ring's refill rate
//          (1184 KiB per 256-sample tile = 2368 MFMAs per wave);
//   VALU   + one v_cvt_pk_bf16_f32 and one v_pk_max_i16 per two MFMAs -- the conversions of a 256-wide layer's outputs.
// No barriers, no waits other than the data's own: what is left when scheduling is perfect.
enum { kLds = 1, kDma = 2, kValu = 4 };

template <int WHAT>
__global__ __launch_bounds__(256) void mfma_burn_dataflow(int iters, const uint32_t* __restrict__ gsrc, float* sink, uint64_t* clocks) {
  __shared__ __attribute__((aligned(1024))) uint32_t lds[16384];      // 64 KiB: 32 KiB read as fragments, 32 KiB DMA target
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = rnd_pair(0x2468aceu + i * 7u + blockIdx.x * 131u, false);
  __syncthreads();
  u32x4 b0[8], b1[8];
  for (int k = 0; k < 8; ++k)
    for (int i = 0; i < 4; ++i) {
      b0[k][i] = rnd_pair(0x89abcdeu + threadIdx.x * 64u + blockIdx.x * 16384u + k * 8u + i, false, true);
      b1[k][i] = rnd_pair(0x13579bdu + threadIdx.x * 64u + blockIdx.x * 16384u + k * 8u + i, false, true);
    }
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  typedef const __attribute__((address_space(3))) u32x4* lds_rd;
  const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds));
  auto frag = [&](int f) -> u32x4 {      // fragment f of this wave's 8 KiB window: lane-linear, conflict-free
    if (!(WHAT & kLds)) return b0[f & 7];
    return *((lds_rd)(uintptr_t)(base + wave * 8192 + (f & 7) * 1024 + lane * 16));
  };
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(gsrc), 0, 0x7fffffff, 0x00020000);
#endif
  uint64_t t0 = 0, r0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    t0 = __builtin_readcyclecounter();
    r0 = wall_clock64();
  }
  u32x4 fr[4];
  for (int i = 0; i < 4; ++i) fr[i] = frag(i);
  uint32_t packed = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {      // 8 fragments -> 16 MFMAs
      const u32x4 a = fr[u & 3];
      fr[u & 3] = frag(u + 4);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b0[u]), acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b1[u]), acc1, 0, 0, 0);
      if (WHAT & kValu) {
        uint32_t p;
        if (WHAT & 8)      // kClamp (round 5): one clamped conversion instead of cvt + max
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(p) : "v"(__builtin_bit_cast(float, b0[u][0])), "v"(__builtin_bit_cast(float, b1[u][1])));
        else
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0" : "=v"(p) : "v"(__builtin_bit_cast(float, b0[u][0])), "v"(__builtin_bit_cast(float, b1[u][1])));
        packed ^= p;
      }
#if defined(__HIP_DEVICE_COMPILE__)
      if ((WHAT & kDma) && (u == 3 || u == 7)) {      // 1 KiB per 8 MFMAs per wave
        const uint32_t dst = base + 32768 + wave * 8192 + (u >> 2) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)static_cast<uintptr_t>(dst), 16, lane * 16,
                                                 static_cast<int>(((it * 2 + (u >> 2)) & 1023) * 1024), 0, 0);
      }
#endif
    }
    if ((it & 127) == 127) {
      for (int r = 0; r < 16; ++r) {
        acc0[r] *= 1.0e-6f;
        acc1[r] *= 1.0e-6f;
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clocks[0] = __builtin_readcyclecounter() - t0;
    clocks[1] = wall_clock64() - r0;
  }
  const float s_ = acc0[0] + acc0[7] + acc1[3] + __builtin_bit_cast(float, packed);
  if (s_ == 12345.678f) sink[0] = s_ + lds[threadIdx.x + 8192];
}


// Round 5: the same stream with the LDS-DMA refill issued by ANOTHER wave.  Consumers = waves 0-3 (one per SIMD): MFMAs +
// fragment reads (+ conversions); loaders = LOADERS further waves of the same workgroup (4: one beside each consumer on its SIMD; 1: one per
// CU) that issue the consumers' 1 KiB pieces at the same rate -- paced by a progress word each consumer publishes in LDS once per 16 MFMAs
// (one ds_write_b32), polled by the loader with s_sleep between polls, at most AHEAD iterations in front, <= 4 pieces in flight.
// kClamp: the conversions as ONE instruction per two MFMAs (v_cvt_pk_bf16_f32 ... clamp instead of cvt + v_pk_max_i16).
// NOTE what this row can and cannot stand for: every wave of a dispatch gets the SAME register allocation (one granulated VGPR count in the
// kernel descriptor), so a loader wave beside a 484-register MFMA wave is not launchable -- the probe's consumers need ~130 registers.
enum { kClamp = 8 };
template <int WHAT, int LOADERS>
__global__ __launch_bounds__(256 + 64 * LOADERS) void mfma_burn_dataflow_pc(int iters, const uint32_t* __restrict__ gsrc, float* sink, uint64_t* clocks) {
  __shared__ __attribute__((aligned(1024))) uint32_t lds[16384 + 64];      // 32 KiB fragments, 32 KiB DMA target, progress words
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = rnd_pair(0x2468aceu + i * 7u + blockIdx.x * 131u, false);
  if (threadIdx.x < 64) lds[16384 + threadIdx.x] = 0u;
  __syncthreads();
  typedef const __attribute__((address_space(3))) u32x4* lds_rd;
  const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds));
  const uint32_t prog = base + 65536;
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(gsrc), 0, 0x7fffffff, 0x00020000);
#endif
  if (wave >= 4) {      // ---- loader ----
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int AHEAD = 2;
    const int first = (LOADERS == 4) ? wave - 4 : 0, count = (LOADERS == 4) ? 1 : 4;
    for (int it = 0; it < iters; ++it) {
      // wait until every consumer this loader serves has reached iteration it - AHEAD
      while (true) {
        int behind = 0;
        for (int c = first; c < first + count; ++c) {
          uint32_t p;
          asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(p) : "v"(prog + c * 4) : "memory");
          behind |= (static_cast<int>(__builtin_amdgcn_readfirstlane(p)) + AHEAD < it);
        }
        if (!behind) break;
        __builtin_amdgcn_s_sleep(2);
      }
      for (int c = first; c < first + count; ++c)
        for (int half = 0; half < 2; ++half) {
          const uint32_t dst = base + 32768 + c * 8192 + half * 1024;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)static_cast<uintptr_t>(dst), 16, lane * 16,
                                                   static_cast<int>(((it * 2 + half) & 1023) * 1024), 0, 0);
        }
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    return;
  }
  // ---- consumer ----
  u32x4 b0[8], b1[8];
  for (int k = 0; k < 8; ++k)
    for (int i = 0; i < 4; ++i) {
      b0[k][i] = rnd_pair(0x89abcdeu + threadIdx.x * 64u + blockIdx.x * 16384u + k * 8u + i, false, true);
      b1[k][i] = rnd_pair(0x13579bdu + threadIdx.x * 64u + blockIdx.x * 16384u + k * 8u + i, false, true);
    }
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  auto frag = [&](int f) -> u32x4 { return *((lds_rd)(uintptr_t)(base + wave * 8192 + (f & 7) * 1024 + lane * 16)); };
  uint64_t t0 = 0, r0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    t0 = __builtin_readcyclecounter();
    r0 = wall_clock64();
  }
  u32x4 fr[4];
  for (int i = 0; i < 4; ++i) fr[i] = frag(i);
  uint32_t packed = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const u32x4 a = fr[u & 3];
      fr[u & 3] = frag(u + 4);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b0[u]), acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b1[u]), acc1, 0, 0, 0);
      if (WHAT & kValu) {
        uint32_t p;
        if (WHAT & kClamp)
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(p) : "v"(__builtin_bit_cast(float, b0[u][0])), "v"(__builtin_bit_cast(float, b1[u][1])));
        else
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0" : "=v"(p) : "v"(__builtin_bit_cast(float, b0[u][0])), "v"(__builtin_bit_cast(float, b1[u][1])));
        packed ^= p;
      }
    }
    if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(prog + wave * 4), "v"(it + 1) : "memory");
    if ((it & 127) == 127) {
      for (int r = 0; r < 16; ++r) {
        acc0[r] *= 1.0e-6f;
        acc1[r] *= 1.0e-6f;
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clocks[0] = __builtin_readcyclecounter() - t0;
    clocks[1] = wall_clock64() - r0;
  }
  const float s_ = acc0[0] + acc0[7] + acc1[3] + __builtin_bit_cast(float, packed);
  if (s_ == 12345.678f) sink[0] = s_ + lds[threadIdx.x + 8192];
}

template <int WHAT, int LOADERS>
void run_dataflow_pc(int cus, const uint32_t* gsrc, float* sink, uint64_t* d_clocks, double target_ms) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  int iters = 200;
  const dim3 block(256 + 64 * LOADERS);
  hipLaunchKernelGGL((mfma_burn_dataflow_pc<WHAT, LOADERS>), dim3(cus), block, 0, 0, 50, gsrc, sink, d_clocks);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((mfma_burn_dataflow_pc<WHAT, LOADERS>), dim3(cus), block, 0, 0, iters, gsrc, sink, d_clocks);
    (void)hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) {
      printf("producer/consumer dataflow launch failed\n");
      return;
    }
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep == 0) iters = static_cast<int>(iters * target_ms / (ms > 0.01 ? ms : 0.01)) + 1;
  }
  uint64_t ck[2] = {0, 0};
  (void)hipMemcpy(ck, d_clocks, sizeof(ck), hipMemcpyDeviceToHost);
  const double tf = 2.0 * 32 * 32 * 16 * 16.0 * iters * 4.0 * cus / ms * 1e-9, mhz = ck[1] ? 100.0 * ck[0] / static_cast<double>(ck[1]) : 0.0;
  char what[96];
  snprintf(what, sizeof(what), "LDS%s, DMA by %s", (WHAT & kValu) ? ((WHAT & kClamp) ? " + cvt-clamp" : " + conversions") : "",
           LOADERS == 4 ? "a 2nd wave per SIMD" : "1 loader wave per CU");
  printf("bf16 relu + %-42s %9.1f ms  %7.0f TFLOP/s  %5.0f MHz  issue-slot use %.3f  (of 2500: %.3f)\n", what, ms, tf, mhz,
         mhz > 0 ? tf * 1e12 / (cus * 4.0 * 1024.0 * mhz * 1e6) : 0.0, tf / 2500.0);
}


// Round 5: the dataflow of the RUN-TIME-SHAPED kernels (k_generic16.hip.hpp, TileStage) around the same ReLU-like stream.
// Per output tile of 32 features: s_waitcnt vmcnt(0) + s_barrier, the copy of the NEXT tile's KS fragments (global -> LDS, every wave each
// fourth KiB) into the other of two buffers, the tile's bias block (4 ds_read_b128 per sample block), KS k-steps of NB MFMAs (each fragment
// read from LDS feeds NB sample blocks, requested up to 4 ahead), then the tile's conversions (8 NB x (v_cvt_pk + v_pk_max)).  KS = W / 16 is what
// a hidden layer of width W has: 4 / 8 / 16 for 64 / 128 / 256, i.e. 8 / 16 / 32 MFMAs per wave between two barriers at NB = 2.
// OCC workgroups of 4 waves per CU as the kernels run (2 / 2 / 1).  RESIDENT: the same without wait, barrier and copy -- every fragment already
// in LDS (what a network that fits would see).
template <int KS, int NB, int OCC, bool RESIDENT>
__global__ __launch_bounds__(256, OCC) void tile_stage_dataflow(int tiles, const uint32_t* __restrict__ gsrc, float* sink, uint64_t* clocks) {
  __shared__ __attribute__((aligned(1024))) uint32_t lds[2 * KS * 256 + 1024];      // two tile buffers + a 4 KiB "bias table"
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  for (int i = threadIdx.x; i < 2 * KS * 256 + 1024; i += 256) lds[i] = rnd_pair(0x2468aceu + i * 7u + blockIdx.x * 131u, false);
  __syncthreads();
  typedef const __attribute__((address_space(3))) u32x4* lds_rd;
  const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds));
  const uint32_t bias_at = base + 2 * KS * 1024;
  u32x4 b[NB][KS];
  for (int nb = 0; nb < NB; ++nb)
    for (int k = 0; k < KS; ++k)
      for (int i = 0; i < 4; ++i) b[nb][k][i] = rnd_pair(0x89abcdeu + threadIdx.x * 64u + blockIdx.x * 16384u + (nb * KS + k) * 8u + i, false, true);
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(gsrc), 0, 0x7fffffff, 0x00020000);
#endif
  uint64_t t0 = 0, r0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    t0 = __builtin_readcyclecounter();
    r0 = wall_clock64();
  }
  constexpr int D = KS < 4 ? KS : 4;
  uint32_t buf = 0, packed = 0;
  float keepf = 0.f;
  for (int t = 0; t < tiles; ++t) {
    if (!RESIDENT) {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#if defined(__HIP_DEVICE_COMPILE__)
      for (int i = wave; i < KS; i += 4)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)static_cast<uintptr_t>(base + (buf ^ 1u) * KS * 1024 + i * 1024), 16,
                                                 lane * 16, static_cast<int>(((t * KS + i) & 1023) * 1024), 0, 0);
#endif
    }
    const uint32_t rd = base + buf * KS * 1024 + lane * 16;
    buf ^= 1u;
    u32x4 fr[D];
#pragma unroll
    for (int i = 0; i < D; ++i) fr[i] = *((lds_rd)(uintptr_t)(rd + i * 1024));
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const u32x4 v = *((lds_rd)(uintptr_t)(bias_at + (lane >> 5) * 64 + g * 16 + ((t & 15) * 128)));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[nb][4 * g + e] = __builtin_bit_cast(float, v[e]) * 1.0e-9f;
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4 a = fr[s % D];
      if (s + D < KS) fr[s % D] = *((lds_rd)(uintptr_t)(rd + (s + D) * 1024));
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b[nb][s]), acc[nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        uint32_t p;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0" : "=v"(p) : "v"(acc[nb][2 * q]), "v"(acc[nb][2 * q + 1]));
        packed ^= p;
      }
    keepf += acc[0][3];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clocks[0] = __builtin_readcyclecounter() - t0;
    clocks[1] = wall_clock64() - r0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (keepf + __builtin_bit_cast(float, packed) == 12345.678f) sink[0] = keepf + lds[threadIdx.x];
}

template <int KS, int NB, int OCC, bool RESIDENT>
void run_tile_stage(int cus, const uint32_t* gsrc, float* sink, uint64_t* d_clocks, double target_ms) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  int tiles = 2000;
  hipLaunchKernelGGL((tile_stage_dataflow<KS, NB, OCC, RESIDENT>), dim3(cus * OCC), dim3(256), 0, 0, 200, gsrc, sink, d_clocks);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((tile_stage_dataflow<KS, NB, OCC, RESIDENT>), dim3(cus * OCC), dim3(256), 0, 0, tiles, gsrc, sink, d_clocks);
    (void)hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) {
      printf("tile-stage dataflow launch failed\n");
      return;
    }
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep == 0) tiles = static_cast<int>(tiles * target_ms / (ms > 0.01 ? ms : 0.01)) + 1;
  }
  uint64_t ck[2] = {0, 0};
  (void)hipMemcpy(ck, d_clocks, sizeof(ck), hipMemcpyDeviceToHost);
  const double tf = 2.0 * 32 * 32 * 16 * static_cast<double>(KS) * NB * tiles * 4.0 * cus * OCC / ms * 1e-9, mhz = ck[1] ? 100.0 * ck[0] / static_cast<double>(ck[1]) : 0.0;
  printf("width %3d: %2d MFMAs per wave per tile, %d workgroup(s) per CU, %-34s %8.1f ms  %7.0f TFLOP/s  %5.0f MHz  (of 2500: %.3f)\n", KS * 16, KS * NB, OCC,
         RESIDENT ? "fragments resident in LDS" : "wait + barrier + copy per tile", ms, tf, mhz, tf / 2500.0);
}

template <int WHAT>
void run_dataflow(int cus, const uint32_t* gsrc, float* sink, uint64_t* d_clocks, double target_ms) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  int iters = 200;
  hipLaunchKernelGGL((mfma_burn_dataflow<WHAT>), dim3(cus), dim3(256), 0, 0, 50, gsrc, sink, d_clocks);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((mfma_burn_dataflow<WHAT>), dim3(cus), dim3(256), 0, 0, iters, gsrc, sink, d_clocks);
    (void)hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) {
      printf("dataflow launch failed\n");
      return;
    }
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep == 0) iters = static_cast<int>(iters * target_ms / (ms > 0.01 ? ms : 0.01)) + 1;
  }
  uint64_t ck[2] = {0, 0};
  (void)hipMemcpy(ck, d_clocks, sizeof(ck), hipMemcpyDeviceToHost);
  const double tf = 2.0 * 32 * 32 * 16 * 16.0 * iters * 4.0 * cus / ms * 1e-9, mhz = ck[1] ? 100.0 * ck[0] / static_cast<double>(ck[1]) : 0.0;
  printf("bf16 relu + %-22s 2 chains, 1 wave/SIMD %9.1f ms  %7.0f TFLOP/s  %5.0f MHz  issue-slot use %.3f  (of 2500: %.3f)\n",
         WHAT == kLds ? "LDS fragments" : WHAT == (kLds | kDma) ? "LDS + DMA refill" : WHAT == (kLds | kDma | kValu) ? "LDS + DMA + conversions" : WHAT == (kLds | kDma | kValu | 8) ? "LDS + DMA + cvt-clamp" : "registers only",
         ms, tf, mhz, mhz > 0 ? tf * 1e12 / (cus * 4.0 * 1024.0 * mhz * 1e6) : 0.0, tf / 2500.0);
}

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
  // the shading kernel's dataflow around the ReLU-like stream (sustained only)
  uint32_t* gsrc;
  if (hipMalloc(&gsrc, 1024 * 1024 + 4096) != hipSuccess || hipMemset(gsrc, 0x3c, 1024 * 1024 + 4096) != hipSuccess) return 1;
  printf("-- the shading kernel's operand delivery around the ReLU-like stream (~500 ms per launch)\n");
  run_dataflow<0>(cus, gsrc, sink, d_clocks, 500.0);
  run_dataflow<kLds>(cus, gsrc, sink, d_clocks, 500.0);
  run_dataflow<kLds | kDma>(cus, gsrc, sink, d_clocks, 500.0);
  run_dataflow<kLds | kDma | kValu>(cus, gsrc, sink, d_clocks, 500.0);
  printf("-- round 5: the refill issued by another wave of the workgroup; conversions as one clamped cvt (~500 ms per launch)\n");
  run_dataflow_pc<kLds | kDma, 4>(cus, gsrc, sink, d_clocks, 500.0);
  run_dataflow_pc<kLds | kDma, 1>(cus, gsrc, sink, d_clocks, 500.0);
  run_dataflow_pc<kLds | kDma | kValu, 4>(cus, gsrc, sink, d_clocks, 500.0);
  run_dataflow_pc<kLds | kDma | kValu, 1>(cus, gsrc, sink, d_clocks, 500.0);
  run_dataflow_pc<kLds | kDma | kValu | kClamp, 4>(cus, gsrc, sink, d_clocks, 500.0);
  run_dataflow<kLds | kDma | kValu | kClamp>(cus, gsrc, sink, d_clocks, 500.0);
  printf("-- round 5: the run-time-shaped kernels' per-tile staging (k_generic16.hip.hpp) around the ReLU-like stream, two sample blocks per wave (~300 ms per launch)\n");
  run_tile_stage<4, 2, 2, false>(cus, gsrc, sink, d_clocks, 300.0);
  run_tile_stage<4, 2, 2, true>(cus, gsrc, sink, d_clocks, 300.0);
  run_tile_stage<8, 2, 2, false>(cus, gsrc, sink, d_clocks, 300.0);
  run_tile_stage<8, 2, 2, true>(cus, gsrc, sink, d_clocks, 300.0);
  run_tile_stage<16, 2, 1, false>(cus, gsrc, sink, d_clocks, 300.0);
  run_tile_stage<16, 2, 1, true>(cus, gsrc, sink, d_clocks, 300.0);
  run_tile_stage<16, 2, 2, false>(cus, gsrc, sink, d_clocks, 300.0);
  return 0;
}
