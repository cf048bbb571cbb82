// TEST INFRASTRUCTURE (tools/host_abi_sanitize.sh builds it; tests/test_host_cpu.py runs it): the library's HOST-ONLY entry points --
// adanerf_host_parse_model (setup_model: every check adanerf_create makes on a model directory), adanerf_host_depth_table,
// adanerf_host_pack_weights -- called through the C ABI on randomly damaged model directories, with the library's own host code
// (adanerf_hip.hip and launch_f32.hip compiled --cuda-host-only, format.cpp, pack.cpp) built with -fsanitize=address,undefined.
// An audit build: it contains no device code (an empty fat-binary stub satisfies the linker), launches nothing and is never shipped.
//   host_abi_fuzz <model_dir> <work_dir> <iterations> <seed>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <random>
#include <string>
#include <vector>
#include "adanerf_hip.h"
static const char* kFiles[4] = {"config.ini", "dataset_info.txt", "model0.onnx", "model1.onnx"};
static std::vector<unsigned char> slurp(const std::string& p) { std::ifstream f(p, std::ios::binary); return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); }
static void spit(const std::string& p, const std::vector<unsigned char>& b) { std::ofstream f(p, std::ios::binary | std::ios::trunc); f.write((const char*)b.data(), (std::streamsize)b.size()); }
static int run(const std::string& dir, std::mt19937_64& rng) {
  adanerf_options o; std::memset(&o, 0, sizeof(o));
  o.width = 8 + (int)(rng() % 64); o.height = 8 + (int)(rng() % 64); o.threshold = (rng() % 3) ? -1.0f : (float)(rng() % 100) / 100.f; o.shard_world = 1; o.strip_rows = 8;
  o.num_samples = (rng() % 3) ? 0 : (int)(rng() % 140) - 4; o.precision = (int)(rng() % 3); o.sampling_mode = (int)(rng() % 4);
  adanerf_info info; float zt[128];
  int ok = 0;
  ok += adanerf_host_parse_model(dir.c_str(), &o, &info) == 0;
  ok += adanerf_host_depth_table(dir.c_str(), &o, zt) == 0;
  for (int net = 0; net < 2; ++net) for (int prec = 0; prec < 5; ++prec) { size_t wb = 0, bf = 0; int nl = 0; ok += adanerf_host_pack_weights(dir.c_str(), net, prec, nullptr, &wb, nullptr, &bf, nullptr, &nl) == 0; }
  return ok;
}
int main(int argc, char** argv) {
  std::string src = argv[1], work = argv[2]; int iters = atoi(argv[3]); std::mt19937_64 rng(atoll(argv[4]));
  std::vector<unsigned char> orig[4];
  for (int f = 0; f < 4; ++f) { orig[f] = slurp(src + "/" + kFiles[f]); spit(work + "/" + kFiles[f], orig[f]); }
  printf("untouched: %d calls ok\n", run(work, rng));
  long okc = 0;
  for (int it = 0; it < iters; ++it) {
    int f = (int)(rng() % 8); if (f >= 4) f = f % 2;      // mostly the text files: setup_model's logic
    std::vector<unsigned char> m = orig[f];
    for (int k = 1 + (int)(rng() % 3); k > 0 && !m.empty(); --k) {
      size_t a = rng() % m.size();
      switch (rng() % 6) {
        case 0: m.resize(a); break;
        case 1: m[a] ^= (unsigned char)(1u << (rng() % 8)); break;
        case 2: m.erase(m.begin() + (long)a, m.begin() + (long)std::min(m.size(), a + 1 + rng() % 40)); break;
        case 3: m.insert(m.begin() + (long)a, 1 + rng() % 8, (unsigned char)"[],-=0e9.\n"[rng() % 10]); break;
        case 4: for (size_t t = 0; t < 64; ++t) { size_t b = rng() % m.size(); if (m[b] >= '0' && m[b] <= '9') { static const char* sub[] = {"-", "1e39", "99999999999", "nan", "", "-0", "4294967296", "0", "1024", "129", "-1"}; std::string s = sub[rng() % 11]; m.erase(m.begin() + (long)b); m.insert(m.begin() + (long)b, s.begin(), s.end()); break; } } break;
        default: { // drop or duplicate a list entry: remove text between two commas
          size_t c1 = a; while (c1 < m.size() && m[c1] != ',') ++c1; size_t c2 = c1 + 1; while (c2 < m.size() && m[c2] != ',' && m[c2] != ']' && m[c2] != '\n') ++c2;
          if (c2 < m.size()) m.erase(m.begin() + (long)c1, m.begin() + (long)c2); } break;
      }
    }
    spit(work + "/" + kFiles[f], m);
    okc += run(work, rng);
    spit(work + "/" + kFiles[f], orig[f]);
  }
  printf("%d mutated directories, %ld calls succeeded, 0 faults\n", iters, okc);
  return 0;
}
