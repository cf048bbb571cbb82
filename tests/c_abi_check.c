/* Compiled as C99 by tests/test_host_cpu.py: proves include/adanerf_hip.h is a plain-C header and that the
 * shared library links and behaves through the C ABI alone (no C++ types, no exceptions across the line). */
#include <stdio.h>
#include <string.h>

#include "../include/adanerf_hip.h"

int main(int argc, char** argv) {
  adanerf_options opt;
  adanerf_info info;
  adanerf_ctx* ctx = NULL;
  int rc;
  memset(&opt, 0, sizeof(opt));
  opt.width = 64;
  opt.height = 48;
  opt.threshold = -1.0f;
  opt.shard_world = 1;
  rc = adanerf_host_parse_model("/definitely/not/a/model/dir/", &opt, &info);
  printf("parse rc=%d msg=%s\n", rc, adanerf_last_error(NULL));
  if (rc != ADANERF_EIO) return 1;
  rc = adanerf_create(NULL, &opt, &ctx);
  if (rc != ADANERF_EINVAL || ctx != NULL) return 2;
  if (argc > 1) {
    rc = adanerf_host_parse_model(argv[1], &opt, &info);
    printf("model rc=%d rays=%d n_in0=%d N=%d thr=%.3f sampler=%d\n", rc, info.rays_local, info.n_in0, info.num_samples,
           (double)info.threshold, info.sampler_mode);
    if (rc != ADANERF_OK || info.rays_local != 64 * 48 || info.abi_version != ADANERF_ABI_VERSION) return 3;
  }
  if (adanerf_destroy(NULL) != ADANERF_OK) return 4;
  {
    /* the handshake a binding does after loading the library: version and struct sizes as the LIBRARY sees them */
    int32_t sizes[3] = {0, 0, 0};
    if (adanerf_abi_version() != ADANERF_ABI_VERSION) return 5;
    if (adanerf_struct_sizes(sizes) != ADANERF_OK || adanerf_struct_sizes(NULL) != ADANERF_EINVAL) return 6;
    if (sizes[0] != (int32_t)sizeof(adanerf_options) || sizes[1] != (int32_t)sizeof(adanerf_info) || sizes[2] != (int32_t)sizeof(adanerf_stats))
      return 7;
  }
  printf("sizeof options=%u info=%u stats=%u\n", (unsigned)sizeof(adanerf_options), (unsigned)sizeof(adanerf_info),
         (unsigned)sizeof(adanerf_stats));
  return 0;
}
