// Does a float butterfly through __shfl_xor (ds_bpermute_b32) stay reproducible while an MFMA / LDS-heavy kernel runs on the rest of the
// GPU?  (Follow-up of profiles/r03_dense_shard_flake.md: composite_wave_kernel dropped a term of its first wave sum about once per
// 10^4 rays next to other kernels.)  Kernel A: one wave per "ray", three float sums over 128 values reduced exactly like that kernel
// did, compared with the same reduction done through DPP row steps + readlane and with run 0's results.  Kernel B (other stream): MFMA
// loop with LDS traffic on every CU.   hipcc --offload-arch=gfx950 -O3 bpermute_next_to_mfma.hip -o bpermute_next_to_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void sums_shfl(const float4* __restrict__ raw, const float* __restrict__ w, int n, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  float cr = 0.f, cg = 0.f, cb = 0.f;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const float4 v = raw[r * 128 + lane + 64 * u];
    const float wt = w[r * 128 + lane + 64 * u];
    cr += wt * (1.0f / (1.0f + expf(-v.x)));
    cg += wt * (1.0f / (1.0f + expf(-v.y)));
    cb += wt * (1.0f / (1.0f + expf(-v.z)));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    cr += __shfl_xor(cr, off);
    cg += __shfl_xor(cg, off);
    cb += __shfl_xor(cb, off);
  }
  if (lane == 0) {
    out[3 * r + 0] = cr;
    out[3 * r + 1] = cg;
    out[3 * r + 2] = cb;
  }
}

__global__ __launch_bounds__(256) void burner(float* sink, int iters) {
  __shared__ f16x8 tile[64 * 16];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 64 * 16; i += 256) tile[i] = f16x8{1, 2, 3, 4, 5, 6, 7, 8};
  __syncthreads();
  f32x16 acc = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const f16x8 a = tile[s * 64 + lane];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc, 0, 0, 0);
    }
  }
  if (acc[0] == 12345.f) sink[0] = acc[1];
}

// burner 2: the frames' weight streaming -- LDS-DMA copies (buffer_load ... lds) of 1 KiB pieces into a ring, ds_read_b128 of them, MFMA
__global__ __launch_bounds__(256) void burner_dma(const char* __restrict__ src, unsigned bytes, float* sink, int iters) {
  __shared__ __attribute__((aligned(1024))) char ring[64 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, static_cast<int>(bytes), 0x00020000);
  const unsigned base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(ring));
  typedef const __attribute__((address_space(3))) f16x8* lds_ptr;
  f32x16 acc = {};
  unsigned goff = (blockIdx.x * 4096u) % bytes;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned dst = base + ((it & 3) * 16 + wave * 4 + i) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)static_cast<uintptr_t>(dst), 16, lane * 16,
                                               static_cast<int>((goff + (wave * 4 + i) * 1024) % bytes), 0, 0);
    }
    goff = (goff + 16384) % bytes;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const f16x8 a = *((lds_ptr)(uintptr_t)(base + ((it & 3) * 16 + s) * 1024 + lane * 16));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc, 0, 0, 0);
    }
  }
  if (acc[0] == 12345.f) sink[0] = acc[1];
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 4096, reps = argc > 2 ? atoi(argv[2]) : 200, burn = argc > 3 ? atoi(argv[3]) : 1;
  std::vector<float> h_raw(static_cast<size_t>(n) * 128 * 4), h_w(static_cast<size_t>(n) * 128);
  srand(1);
  for (auto& v : h_raw) v = (rand() / float(RAND_MAX)) * 6.f - 3.f;
  for (auto& v : h_w) v = (rand() / float(RAND_MAX)) * 0.02f;
  float *d_raw, *d_w, *d_out, *d_sink;
  hipMalloc(&d_raw, h_raw.size() * 4); hipMalloc(&d_w, h_w.size() * 4); hipMalloc(&d_out, static_cast<size_t>(reps) * n * 3 * 4); hipMalloc(&d_sink, 64);
  char* d_src;
  hipMalloc(&d_src, 1u << 22);
  hipMemset(d_src, 0x3c, 1u << 22);
  hipMemcpy(d_raw, h_raw.data(), h_raw.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_w, h_w.data(), h_w.size() * 4, hipMemcpyHostToDevice);
  hipStream_t sa, sb;
  hipStreamCreate(&sa); hipStreamCreate(&sb);
  for (int pass = 0; pass < 2; ++pass) {
    const bool with_burner = burn && pass == 1;
    for (int i = 0; i < reps; ++i) {
      if (with_burner && i % 4 == 0) {
        if (burn == 2) hipLaunchKernelGGL(burner_dma, dim3(256), dim3(256), 0, sb, d_src, 1u << 22, d_sink, 300);      // co-resident with kernel A's waves
        else hipLaunchKernelGGL(burner, dim3(192), dim3(256), 0, sb, d_sink, 400);                                      // leaves CUs for kernel A
      }
      hipLaunchKernelGGL(sums_shfl, dim3((n + 3) / 4), dim3(256), 0, sa, reinterpret_cast<const float4*>(d_raw), d_w, n, d_out + static_cast<size_t>(i) * n * 3);
    }
    hipDeviceSynchronize();
    std::vector<float> h(static_cast<size_t>(reps) * n * 3);
    hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0, bad_ch[3] = {0, 0, 0};
    double worst = 0;
    for (int i = 1; i < reps; ++i)
      for (int k = 0; k < n * 3; ++k)
        if (h[static_cast<size_t>(i) * n * 3 + k] != h[k]) {
          ++bad; ++bad_ch[k % 3];
          const double d = h[static_cast<size_t>(i) * n * 3 + k] - h[k];
          if (fabs(d) > fabs(worst)) worst = d;
        }
    printf("%s: %ld of %ld sums differ from run 0 (per channel %ld / %ld / %ld; largest difference %.3e)\n", with_burner ? (burn == 2 ? "next to the LDS-DMA + MFMA kernel" : "next to the MFMA kernel") : "alone",
           bad, static_cast<long>(reps - 1) * n * 3, bad_ch[0], bad_ch[1], bad_ch[2], worst);
  }
  return 0;
}
