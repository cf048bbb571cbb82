// What does the clamp modifier do on v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 on gfx950?  (Round 5: the bf16 shading kernels use it as ReLU + conversion in one
// instruction, k_mlp16.hip.hpp Bf16::kClampRelu.)  Prints input pairs and the clamped conversions: negative -> 0, above 1 -> 1, in between unchanged.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/cvt_clamp_probe.hip -o /tmp/cvt_clamp && /tmp/cvt_clamp        (profiles/r05_cvt_clamp_probe.log)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__global__ void k(const float* in, uint32_t* out) {
  float a = in[threadIdx.x * 2], b = in[threadIdx.x * 2 + 1];
  uint32_t p, q;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(p) : "v"(a), "v"(b));
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2 clamp" : "=v"(q) : "v"(a), "v"(b));
  out[threadIdx.x * 2] = p;
  out[threadIdx.x * 2 + 1] = q;
}
int main() {
  float h[16] = {-1.f, 0.5f, 2.0f, 0.75f, -0.0f, 1.0f, 1e-3f, -1e-3f, 0.999f, 1.001f, 3e38f, -3e38f, 0.f, 0.33333f, 100.f, 0.125f};
  float* d; uint32_t* o; uint32_t r[16];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, d, o);
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  for (int i = 0; i < 8; ++i) {
    uint32_t lo = r[2*i] << 16, hi = r[2*i] & 0xffff0000u; float fl, fh; memcpy(&fl, &lo, 4); memcpy(&fh, &hi, 4);
    printf("in (%g, %g) -> bf16 clamp (%g, %g)  f16 bits %08x\n", h[2*i], h[2*i+1], fl, fh, r[2*i+1]);
  }
  return 0;
}
