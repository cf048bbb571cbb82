#include "format.hpp"

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace adanerf {

std::string join_path(const std::string& dir, const std::string& file) {
  if (dir.empty()) return file;
  char last = dir[dir.size() - 1];
  if (last == '/' || last == '\\') return dir + file;
  return dir + "/" + file;
}

// ---- key = value files ------------------------------------------------------------------

static std::string strip_ws(const std::string& s) {
  std::string o;
  o.reserve(s.size());
  for (char c : s)
    if (!std::isspace(static_cast<unsigned char>(c))) o.push_back(c);
  return o;
}

// "[a,b,c]" -> {"a","b","c"}; a bare scalar -> {"scalar"}; "[]" -> {}
static std::vector<std::string> split_list(const std::string& value) {
  std::string s = value;
  size_t a = s.find('['), b = s.find(']');
  if (a != std::string::npos && b != std::string::npos && b > a) s = s.substr(a + 1, b - a - 1);
  std::vector<std::string> out;
  if (s.empty()) return out;
  size_t pos = 0;
  while (true) {
    size_t c = s.find(',', pos);
    if (c == std::string::npos) {
      out.push_back(s.substr(pos));
      break;
    }
    out.push_back(s.substr(pos, c - pos));
    pos = c + 1;
  }
  return out;
}

static std::vector<float> to_floats(const std::vector<std::string>& v) {
  std::vector<float> o;
  for (auto& s : v) o.push_back(s.empty() ? 0.f : static_cast<float>(std::atof(s.c_str())));
  return o;
}
static std::vector<int> to_ints(const std::vector<std::string>& v) {
  std::vector<int> o;
  for (auto& s : v) {      // out-of-range text saturates (atoi's behaviour there is undefined)
    const long x = s.empty() ? 0 : std::strtol(s.c_str(), nullptr, 10);
    o.push_back(static_cast<int>(std::max(-2147483647L - 1, std::min(2147483647L, x))));
  }
  return o;
}
// a frequency-band count of posEncArgs: the callers convert it to int, so anything that is no small non-negative number (nan, 1e39, a
// damaged file) becomes -1, which every range check downstream refuses
static float band_count(const std::string& s) {
  const double v = std::atof(s.c_str());
  return (v >= 0.0 && v <= 1024.0) ? static_cast<float>(v) : -1.0f;
}

// Keys consumed: the set of adanerf_real_time_viewer/src/config.cpp:206-266.  Unknown keys are
// ignored.  A key that appears twice takes the later value (the reference appends to its vectors,
// config.cpp:41-49; the shipped model directories contain no duplicates).
void Config::store(std::string key, std::string value) {
  key = strip_ws(key);
  value = strip_ws(value);
  if (key.empty() || key[0] == ';' || key[0] == '#' || key[0] == '[') return;
  auto L = split_list(value);
  if (key == "posEncArgs") {
    posEncArgs.clear();
    for (auto& item : L) {
      std::vector<float> tf;
      if (item == "none") {   // config.cpp:142-146
        tf = {4.0f, 0.0f};
      } else {
        size_t d = item.find('-');
        if (d == std::string::npos) {
          tf = {band_count(item), 0.f};
        } else {
          tf = {band_count(item.substr(0, d)), band_count(item.substr(d + 1))};
        }
      }
      posEncArgs.push_back(tf);
    }
  } else if (key == "posEnc") posEnc = L;
  else if (key == "inFeatures") inFeatures = L;
  else if (key == "outFeatures") outFeatures = L;
  else if (key == "rayMarchSampler") rayMarchSampler = L;
  else if (key == "rayMarchNormalization") rayMarchNormalization = L;
  else if (key == "rayMarchNormalizationCenter") rayMarchNormalizationCenter = to_floats(L);
  else if (key == "activation") activation = L;
  else if (key == "losses") losses = L;
  else if (key == "numRaymarchSamples") numRaymarchSamples = to_ints(L);
  else if (key == "rayMarchSamplingStep") rayMarchSamplingStep = to_floats(L);
  else if (key == "rayMarchSamplingNoise") rayMarchSamplingNoise = to_floats(L);
  else if (key == "zNear") zNear = to_floats(L);
  else if (key == "zFar") zFar = to_floats(L);
  else if (key == "adaptiveSamplingThreshold") adaptiveSamplingThreshold = static_cast<float>(std::atof(value.c_str()));
  else if (key == "depth_range") depthRange = to_floats(L);
  else if (key == "view_cell_size") viewcellSize = to_floats(L);
  else if (key == "view_cell_center") viewcellCenter = to_floats(L);
  else if (key == "fov") fov = std::atof(value.c_str());
  else if (key == "max_depth") max_depth = static_cast<float>(std::atof(value.c_str()));
  else if (key == "raySampleInput") raySampleInput = to_ints(L);
  else if (key == "multiDepthFeatures") multiDepthFeatures = to_ints(L);
  else if (key == "depthTransform") depthTransform = value;
  else if (key == "accumulationMult") accumulationMult = value;
  else if (key == "useNDC") useNDC = (value == "True" || value == "true" || value == "1");
}

static bool load_kv(Config* c, const std::string& path, std::string* err) {
  std::ifstream f(path);
  if (!f) {
    if (err) *err = "couldn't open " + path;
    return false;
  }
  std::string line;
  while (std::getline(f, line)) {
    size_t eq = line.find('=');
    if (eq == std::string::npos) continue;
    c->store(line.substr(0, eq), line.substr(eq + 1));
  }
  return true;
}

bool Config::load(const std::string& dir, std::string* err) {
  if (!load_kv(this, join_path(dir, "config.ini"), err)) return false;
  if (!load_kv(this, join_path(dir, "dataset_info.txt"), err)) return false;
  return true;
}

// ---- ONNX initializers (hand-rolled protobuf wire reader) -----------------------------------
// ModelProto.graph = field 7; GraphProto.initializer = field 5; TensorProto: dims = 1 (varint,
// possibly packed), data_type = 2 (1 == FLOAT), float_data = 4, name = 8, raw_data = 9.
// Written by torch.onnx.export in the reference's src/export.py:78-83.

namespace {
struct Span {
  const uint8_t* p;
  size_t n;
};
struct Field {
  uint32_t num;
  uint32_t wire;
  uint64_t varint;
  Span bytes;
};

bool read_varint(const uint8_t* p, size_t n, size_t* i, uint64_t* out) {
  uint64_t r = 0;
  int shift = 0;
  while (*i < n && shift < 64) {
    uint8_t c = p[(*i)++];
    r |= static_cast<uint64_t>(c & 0x7F) << shift;
    shift += 7;
    if (c < 0x80) {
      *out = r;
      return true;
    }
  }
  return false;
}

// iterates the fields of one message; returns false on malformed input
template <typename Fn>
bool for_each_field(Span s, Fn fn) {
  size_t i = 0;
  while (i < s.n) {
    uint64_t key;
    if (!read_varint(s.p, s.n, &i, &key)) return false;
    Field f{};
    f.num = static_cast<uint32_t>(key >> 3);
    f.wire = static_cast<uint32_t>(key & 7);
    switch (f.wire) {
      case 0:
        if (!read_varint(s.p, s.n, &i, &f.varint)) return false;
        break;
      case 2: {
        uint64_t len;
        if (!read_varint(s.p, s.n, &i, &len)) return false;
        if (len > s.n - i) return false;
        f.bytes = Span{s.p + i, static_cast<size_t>(len)};
        i += len;
        break;
      }
      case 5:
        if (s.n - i < 4) return false;
        f.bytes = Span{s.p + i, 4};
        i += 4;
        break;
      case 1:
        if (s.n - i < 8) return false;
        f.bytes = Span{s.p + i, 8};
        i += 8;
        break;
      default:
        return false;
    }
    fn(f);
  }
  return true;
}
}  // namespace

bool read_onnx_initializers(const std::string& path, TensorMap* out, std::string* err) {
  std::ifstream f(path, std::ios::binary);
  if (!f) {
    if (err) *err = "couldn't open " + path;
    return false;
  }
  std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  bool ok = true;
  ok &= for_each_field(Span{buf.data(), buf.size()}, [&](const Field& f1) {
    if (f1.num != 7 || f1.wire != 2) return;
    ok &= for_each_field(f1.bytes, [&](const Field& f2) {
      if (f2.num != 5 || f2.wire != 2) return;
      Tensor t;
      std::string name;
      int dtype = 1;
      Span raw{nullptr, 0};
      std::vector<float> fdata;
      bool bad_dims = false;      // a dimension that is no non-negative int (a hostile or damaged file): the tensor is refused below
      auto push_dim = [&](uint64_t d) {
        if (d > 0x7FFFFFFFull) bad_dims = true;
        else t.dims.push_back(static_cast<int>(d));
      };
      ok &= for_each_field(f2.bytes, [&](const Field& f3) {
        if (f3.num == 1) {
          if (f3.wire == 0) {
            push_dim(f3.varint);
          } else if (f3.wire == 2) {
            size_t j = 0;
            uint64_t d;
            while (j < f3.bytes.n && read_varint(f3.bytes.p, f3.bytes.n, &j, &d)) push_dim(d);
          }
        } else if (f3.num == 2 && f3.wire == 0) {
          dtype = static_cast<int>(f3.varint);
        } else if (f3.num == 8 && f3.wire == 2) {
          name.assign(reinterpret_cast<const char*>(f3.bytes.p), f3.bytes.n);
        } else if (f3.num == 9 && f3.wire == 2) {
          raw = f3.bytes;
        } else if (f3.num == 4) {
          size_t cnt = f3.bytes.n / 4;
          for (size_t k = 0; k < cnt; ++k) {
            float v;
            std::memcpy(&v, f3.bytes.p + 4 * k, 4);
            fdata.push_back(v);
          }
        }
      });
      if (dtype != 1 || name.empty()) return;
      // element count, bounded by what the file can hold (no product that wraps, no allocation a few crafted bytes can ask for)
      size_t count = 1;
      for (int d : t.dims) {
        if (bad_dims || (d != 0 && count > buf.size() / static_cast<size_t>(d))) {
          ok = false;
          return;
        }
        count *= static_cast<size_t>(d);
      }
      if (bad_dims) {
        ok = false;
        return;
      }
      if (raw.p) {
        if (raw.n != count * 4) {
          ok = false;
          return;
        }
        t.data.resize(count);
        if (raw.n) std::memcpy(t.data.data(), raw.p, raw.n);   // little-endian fp32, host is little-endian
      } else {
        if (fdata.size() != count) {
          ok = false;
          return;
        }
        t.data = fdata;
      }
      (*out)[name] = std::move(t);
    });
  });
  if (!ok) {
    if (err) *err = "malformed ONNX protobuf: " + path;
    return false;
  }
  if (out->empty()) {
    if (err) *err = "no fp32 initializers found in " + path;
    return false;
  }
  return true;
}

}  // namespace adanerf
