// TEST INFRASTRUCTURE (tests/test_host_cpu.py builds it with g++ -fsanitize=address,undefined): the host-side loader of the product --
// adanerf_amd/csrc/format.cpp (config.ini / dataset_info.txt / ONNX-initializer reader) and pack.cpp (MFMA fragment packing) -- on a model
// directory whose files are mutated at random: truncated, bytes flipped, runs overwritten, varints stretched, list entries dropped.  A mutated
// directory may load or be refused with a message; what it must never do is read or write out of bounds, overflow a signed integer, throw
// through the loader or ask for an absurd allocation.  The untouched directory must load and pack in every element type.
//   host_sanitize_fuzz <model_dir> <work_dir> <iterations> <seed>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <iterator>
#include <random>
#include <string>
#include <utility>
#include <vector>

#include "format.hpp"
#include "pack.hpp"

using namespace adanerf;

namespace {
const char* kFiles[4] = {"config.ini", "dataset_info.txt", "model0.onnx", "model1.onnx"};

std::vector<uint8_t> slurp(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
void spit(const std::string& p, const std::vector<uint8_t>& b) {
  std::ofstream f(p, std::ios::binary | std::ios::trunc);
  f.write(reinterpret_cast<const char*>(b.data()), static_cast<std::streamsize>(b.size()));
}

// the calls adanerf_create / adanerf_host_pack_weights make on a model directory (adanerf_hip.hip: setup_model, create); returns the
// number of networks packed (0 = refused)
int load_and_pack(const std::string& dir, std::string* why) {
  Config cfg;
  std::string err;
  if (!cfg.load(dir, &err)) return *why = err, 0;
  if (cfg.posEncArgs.size() != 2) return *why = "posEncArgs", 0;
  NetShape sh{static_cast<int>(cfg.posEncArgs[0][0]), static_cast<int>(cfg.posEncArgs[0][1]), static_cast<int>(cfg.posEncArgs[1][0]),
              static_cast<int>(cfg.posEncArgs[1][1]), cfg.raySampleInput.empty() ? 0 : cfg.raySampleInput[0]};
  // no range checks here: adanerf_host_pack_weights hands the packer whatever config.ini says, and the packer must refuse it itself
  if (!((sh.fp0 == 10 && sh.fd0 == 4) || (sh.fp0 == 2 && sh.fd0 == 2))) sh.lp0 = sh.ld0 = 16;
  if (!(sh.fp1 == 10 && sh.fd1 == 4)) sh.lp1 = sh.ld1 = 16;
  TensorMap n0, n1;
  if (!read_onnx_initializers(join_path(dir, "model0.onnx"), &n0, &err)) return *why = err, 0;
  if (!read_onnx_initializers(join_path(dir, "model1.onnx"), &n1, &err)) return *why = err, 0;
  int packed = 0;
  for (Elem e : {Elem::F32, Elem::F16, Elem::F16_SPLIT}) {
    PackedNet p;
    if (sh.ray_samples > 0 && e != Elem::F32) continue;
    if (pack_sampling_net(n0, sh, e, &p, &err)) ++packed; else *why = err;
  }
  for (Elem e : {Elem::F32, Elem::BF16, Elem::F16}) {
    PackedNet p;
    if (pack_shading_net(n1, sh, e, &p, &err)) ++packed; else *why = err;
  }
  return packed;
}

void put_varint(std::vector<uint8_t>& b, size_t at, uint64_t v) {      // overwrites bytes at `at` with a (possibly over-long) varint
  for (int k = 0; k < 10 && at < b.size(); ++k, ++at) {
    b[at] = static_cast<uint8_t>((v & 0x7F) | (k < 9 ? 0x80 : 0));
    v >>= 7;
  }
}

void mutate(std::vector<uint8_t>& b, std::mt19937_64& rng, bool text) {
  if (b.empty()) return;
  auto at = [&](size_t n) { return static_cast<size_t>(rng() % n); };
  switch (rng() % (text ? 6 : 7)) {
    case 0: b.resize(at(b.size())); break;                                                        // truncated
    case 1: for (int k = 1 + static_cast<int>(rng() % 8); k > 0; --k) b[at(b.size())] ^= static_cast<uint8_t>(1u << (rng() % 8)); break;
    case 2: { size_t a = at(b.size()), n = 1 + at(64); for (size_t i = a; i < b.size() && i < a + n; ++i) b[i] = static_cast<uint8_t>(rng()); } break;
    case 3: { size_t a = at(b.size()), n = 1 + at(256); b.erase(b.begin() + static_cast<long>(a), b.begin() + static_cast<long>(std::min(b.size(), a + n))); } break;
    case 4: { size_t a = at(b.size()); b.insert(b.begin() + static_cast<long>(a), 1 + at(32), static_cast<uint8_t>(text ? "[],-=0e9.\n"[rng() % 10] : rng())); } break;
    case 5:
      if (text) {      // a digit becomes something else a number parser meets: sign, exponent, a run of nines
        for (size_t tries = 0; tries < 64; ++tries) {
          size_t a = at(b.size());
          if (b[a] >= '0' && b[a] <= '9') {
            static const char* sub[] = {"-", "1e39", "99999999999", "nan", "", "-0", "4294967296"};
            const std::string s = sub[rng() % 7];
            b.erase(b.begin() + static_cast<long>(a));
            b.insert(b.begin() + static_cast<long>(a), s.begin(), s.end());
            break;
          }
        }
      } else {
        // the first bytes hold ModelProto's header and the graph's length prefix; inside, every initializer starts with its dims
        static const uint64_t vals[] = {0, 1, 0x7FFFFFFF, 0x80000000ull, 0xFFFFFFFFull, 0x100000000ull, 0x3FFFFFFFFFFFFFFFull, 0x4000000000000000ull, ~0ull};
        put_varint(b, at(std::min<size_t>(b.size(), rng() % 2 ? 64 : b.size())), vals[rng() % 9]);
      }
      break;
    default: { size_t a = at(b.size()); put_varint(b, a, rng() >> (rng() % 64)); } break;      // a random varint anywhere
  }
}
}  // namespace

// crafted initializers: dims whose product wraps, negative and over-long dims, lengths beyond the file
void put_field(std::vector<uint8_t>& b, uint32_t num, uint64_t varint) {
  b.push_back(static_cast<uint8_t>(num << 3));
  do { b.push_back(static_cast<uint8_t>((varint & 0x7F) | (varint > 0x7F ? 0x80 : 0))); varint >>= 7; } while (varint);
}
void put_bytes(std::vector<uint8_t>& b, uint32_t num, const std::vector<uint8_t>& payload, uint64_t claimed_len = ~0ull) {
  b.push_back(static_cast<uint8_t>((num << 3) | 2));
  uint64_t n = claimed_len == ~0ull ? payload.size() : claimed_len;
  do { b.push_back(static_cast<uint8_t>((n & 0x7F) | (n > 0x7F ? 0x80 : 0))); n >>= 7; } while (n);
  b.insert(b.end(), payload.begin(), payload.end());
}
std::vector<uint8_t> tensor(const std::string& name, const std::vector<uint64_t>& dims, size_t raw_floats, bool packed_dims = false, uint64_t claimed_raw_len = ~0ull);
// a whole network of fp32 initializers [rows, cols] + biases [rows], values a small ramp
std::vector<uint8_t> net_model(const std::vector<std::pair<std::string, std::pair<int, int>>>& layers) {
  std::vector<uint8_t> g, m;
  for (auto& l : layers) {
    const uint64_t r = static_cast<uint64_t>(l.second.first), c = static_cast<uint64_t>(l.second.second);
    put_bytes(g, 5, tensor(l.first + ".weight", {r, c}, r * c));
    put_bytes(g, 5, tensor(l.first + ".bias", {r}, r));
  }
  put_bytes(m, 7, g);
  return m;
}
std::vector<uint8_t> crafted_model(const std::vector<uint64_t>& dims, size_t raw_floats, bool packed_dims, uint64_t claimed_raw_len = ~0ull) {
  std::vector<uint8_t> g, m;
  put_bytes(g, 5, tensor("layers.0.weight", dims, raw_floats, packed_dims, claimed_raw_len));
  put_bytes(m, 7, g);
  return m;
}
std::vector<uint8_t> tensor(const std::string& name, const std::vector<uint64_t>& dims, size_t raw_floats, bool packed_dims, uint64_t claimed_raw_len) {
  std::vector<uint8_t> t;
  if (packed_dims) {
    std::vector<uint8_t> pk;
    for (uint64_t d : dims) { do { pk.push_back(static_cast<uint8_t>((d & 0x7F) | (d > 0x7F ? 0x80 : 0))); d >>= 7; } while (d); }
    put_bytes(t, 1, pk);
  } else {
    for (uint64_t d : dims) put_field(t, 1, d);
  }
  put_field(t, 2, 1);
  put_bytes(t, 8, std::vector<uint8_t>(name.begin(), name.end()));
  std::vector<uint8_t> raw(raw_floats * 4, 0);
  for (size_t i = 0; i < raw_floats; ++i) {
    const float v = 0.01f * static_cast<float>(static_cast<int>(i % 17) - 8);
    std::memcpy(raw.data() + 4 * i, &v, 4);
  }
  put_bytes(t, 9, raw, claimed_raw_len);
  return t;
}

int main(int argc, char** argv) {
  if (argc < 5) return std::fprintf(stderr, "usage: %s model_dir work_dir iterations seed\n", argv[0]), 2;
  const std::string src = argv[1], work = argv[2];
  const int iters = std::atoi(argv[3]);
  std::mt19937_64 rng(static_cast<uint64_t>(std::atoll(argv[4])));
  std::string why;
  const int full = load_and_pack(src, &why);
  if (full != 6) return std::fprintf(stderr, "the untouched directory packed %d of 6 networks: %s\n", full, why.c_str()), 1;
  std::vector<uint8_t> orig[4];
  for (int f = 0; f < 4; ++f) {
    orig[f] = slurp(join_path(src, kFiles[f]));
    spit(join_path(work, kFiles[f]), orig[f]);
  }
  {      // crafted model0.onnx files: each must be refused (or load as a harmless tensor), never throw or fault
    const uint64_t M = 0x80000000ull;
    const std::vector<std::vector<uint8_t>> crafted = {
        crafted_model({M, M}, 0, false),                      // (int)2^31 x (int)2^31 = 2^62 elements, x 4 bytes wraps to 0 = the empty raw_data
        crafted_model({M, M}, 0, true),
        crafted_model({0xFFFFFFFFull, 4}, 0, false),          // -1 x 4
        crafted_model({0x100000002ull, 3}, 6, false),         // 2^32 + 2 truncates to 2
        crafted_model({0x40000000ull, 0x40000000ull, 4}, 0, false),
        crafted_model({2, 3}, 6, false, 1ull << 40),          // raw_data claims a terabyte
        crafted_model({~0ull, ~0ull, ~0ull}, 0, true),
        crafted_model({}, 1, false),                          // a scalar
        crafted_model({0, M}, 0, false),
    };
    for (size_t k = 0; k < crafted.size(); ++k) {
      spit(join_path(work, "model0.onnx"), crafted[k]);
      try {
        TensorMap tm;
        std::string err;
        const bool ok = read_onnx_initializers(join_path(work, "model0.onnx"), &tm, &err);
        for (auto& kv : tm) {
          size_t n = 1;
          for (int d : kv.second.dims) {
            if (d < 0) return std::fprintf(stderr, "crafted %zu: a negative dimension came through\n", k), 1;
            n *= static_cast<size_t>(d);
          }
          if (n != kv.second.data.size()) return std::fprintf(stderr, "crafted %zu: dims say %zu elements, data holds %zu\n", k, n, kv.second.data.size()), 1;
        }
        load_and_pack(work, &why);
        (void)ok;
      } catch (const std::exception& e) {
        return std::fprintf(stderr, "crafted %zu: the loader threw %s\n", k, e.what()), 1;
      }
    }
    spit(join_path(work, "model0.onnx"), orig[2]);
  }
  int swept = 0, swept_ok = 0;
  {      // well-formed files of odd topologies: every width / depth / output count around the limits the packer states, layers that disagree
    const int n_in = 90, n_pos = 63, n_dir = 27;      // the 10-4 encoding of the untouched config
    for (int depth : {1, 2, 3, 8, 9})
      for (int W : {0, 1, 2, 31, 64, 65, 255, 256, 257, 300})
        for (int bins : {0, 1, 127, 128, 129}) {
          if (depth > 3 && W > 2 && W < 255) continue;
          if (bins != 128 && !(W == 64 || W == 256)) continue;
          std::vector<std::pair<std::string, std::pair<int, int>>> l0, l1;
          for (int i = 0; i < depth; ++i)
            l0.push_back({"layers." + std::to_string(i), {i == depth - 1 ? bins : W, i == 0 ? n_in : W}});
          for (int i = 0; i < depth; ++i)
            l1.push_back({"pts_linears." + std::to_string(i), {W, i == 0 ? n_pos : (i == depth / 2 + 1 ? W + n_pos : W)}});
          l1.push_back({"feature_linear", {W, W}});
          l1.push_back({"alpha_linear", {1, W}});
          l1.push_back({"views_linears.0", {W / 2, W + n_dir}});
          l1.push_back({"rgb_linear", {bins == 129 ? 4 : 3, W / 2}});      // one family with a wrong colour head
          if (bins == 127 && !l1.empty()) l1[0].second.second += 1;        // ... and one with a wrong input width
          spit(join_path(work, "model0.onnx"), net_model(l0));
          spit(join_path(work, "model1.onnx"), net_model(l1));
          try {
            ++swept;
            swept_ok += load_and_pack(work, &why) > 0;
          } catch (const std::exception& e) {
            return std::fprintf(stderr, "topology depth %d width %d bins %d: the loader threw %s\n", depth, W, bins, e.what()), 1;
          }
        }
    spit(join_path(work, "model0.onnx"), orig[2]);
    spit(join_path(work, "model1.onnx"), orig[3]);
  }
  int loaded = 0, refused = 0;
  for (int it = 0; it < iters; ++it) {
    const int f = static_cast<int>(rng() % 4);
    std::vector<uint8_t> m = orig[f];
    for (int k = 1 + static_cast<int>(rng() % 3); k > 0; --k) mutate(m, rng, f < 2);
    spit(join_path(work, kFiles[f]), m);
    try {
      (load_and_pack(work, &why) > 0 ? loaded : refused)++;
    } catch (const std::exception& e) {
      std::fprintf(stderr, "iteration %d (%s): the loader threw %s\n", it, kFiles[f], e.what());
      spit(join_path(work, std::string("crash_") + kFiles[f]), m);
      return 1;
    }
    spit(join_path(work, kFiles[f]), orig[f]);
  }
  std::printf("9 crafted initializers refused or harmless; %d odd topologies: %d packed; %d mutated directories: %d loaded, %d refused; 0 faults\n", swept, swept_ok, iters, loaded, refused);
  return 0;
}
