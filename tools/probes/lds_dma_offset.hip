// Probe (not part of the library): where does `buffer_load_dwordx4 ... offen offset:IMM lds` put its data on gfx950?
// Question behind it: does the instruction's immediate offset move BOTH the global address and the LDS destination (then the
// pieces of one ring chunk need one M0 write and no per-piece scalar address arithmetic), or the global address only?
// One wave copies 1 KiB with M0 = 4096, voffset = lane * 16, soffset = 8192 and an immediate offset of 0 / 1024 / 3072; the host
// prints where the bytes landed and which global words they are.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/lds_dma_offset.hip -o adanerf_amd/bin/lds_dma_offset
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

template <int IMM>
__global__ __launch_bounds__(64) void probe(const uint32_t* g, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[8192];      // 32 KiB
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = 0xffffffffu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(g), 0, 1 << 20, 0x00020000);
  const uint32_t dst = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds)) + 4096;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)static_cast<uintptr_t>(dst), 16,
                                           static_cast<int>(threadIdx.x * 16), 8192, IMM, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 8192; i += 64) out[i] = lds[i];
}

template <int IMM>
static void run(const uint32_t* g, uint32_t* out) {
  hipLaunchKernelGGL((probe<IMM>), dim3(1), dim3(64), 0, 0, g, out);
  std::vector<uint32_t> h(8192);
  hipMemcpy(h.data(), out, 8192 * 4, hipMemcpyDeviceToHost);
  int first = -1, last = -1;
  for (int i = 0; i < 8192; ++i)
    if (h[i] != 0xffffffffu) { if (first < 0) first = i; last = i; }
  if (first < 0) { printf("imm %4d: nothing landed\n", IMM); return; }
  printf("imm %4d: LDS bytes [%d, %d) relative to the array (M0 pointed at +4096) hold global bytes starting at %u (expected %d if the offset applies to the global side)\n",
         IMM, first * 4, (last + 1) * 4, h[first] * 4, 8192 + IMM);
}

int main() {
  uint32_t *g, *out;
  hipMalloc(&g, 1 << 20); hipMalloc(&out, 8192 * 4);
  std::vector<uint32_t> h(1 << 18);
  for (size_t i = 0; i < h.size(); ++i) h[i] = static_cast<uint32_t>(i);
  hipMemcpy(g, h.data(), 1 << 20, hipMemcpyHostToDevice);
  run<0>(g, out); run<1024>(g, out); run<3072>(g, out);
  return 0;
}
