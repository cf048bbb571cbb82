// Test helper (built by adanerf_amd.build into adanerf_amd/bin/sincos_probe): evaluates the device sin_or_cos() of
// csrc/kernels.hip.hpp on the fp32 values in <in.bin> and writes [sin..., cos...] to <out.bin>.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../adanerf_amd/csrc/kernels.hip.hpp"

__global__ void probe(const float* in, int n, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = adanerf::sin_or_cos(in[i], 0);
  out[n + i] = adanerf::sin_or_cos(in[i], 1);
}

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  fseek(f, 0, SEEK_END);
  const long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  const int n = static_cast<int>(bytes / 4);
  std::vector<float> h(n), o(2 * static_cast<size_t>(n));
  if (fread(h.data(), 4, n, f) != static_cast<size_t>(n)) return 3;
  fclose(f);
  float *d_in, *d_out;
  if (hipMalloc(&d_in, bytes) != hipSuccess || hipMalloc(&d_out, 2 * bytes) != hipSuccess) return 4;
  if (hipMemcpy(d_in, h.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return 4;
  hipLaunchKernelGGL(probe, dim3((n + 255) / 256), dim3(256), 0, 0, d_in, n, d_out);
  if (hipMemcpy(o.data(), d_out, 2 * bytes, hipMemcpyDeviceToHost) != hipSuccess) return 4;
  f = fopen(argv[2], "wb");
  fwrite(o.data(), 4, o.size(), f);
  fclose(f);
  return 0;
}
