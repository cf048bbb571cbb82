// Weight pre-packing into MFMA A-fragment order (host C++).  See layout.hpp for the slot maps.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "format.hpp"

namespace adanerf {

// F16_SPLIT: every fragment is emitted twice, back to back: hi = fp16(w) and lo' = fp16((w - hi) * 2^11)
// (the split-precision path of the sampling net: w ~= hi + lo' * 2^-11 to 22 bits)
enum class Elem { F32, BF16, F16, F16_SPLIT };
constexpr float kSplitScale = 2048.0f;

// One packed network: every layer's A fragments back to back plus the per-tile bias blocks.
//   fp32 engine   : float  w[layer][m][s4][lane][4]   (slot q = 4 s4 + e)
//   16-bit engine : uint16 w[layer][m][s ][lane][8]   (slot q = 8 s  + e)
//   split engine  : uint16 w[layer][m][s ][part][lane][8], part 0 = hi, 1 = lo' 
//   bias          : float  b[layer][m][h][16]         (feature 32m + 8(r>>2) + 4h + (r&3))
struct PackedNet {
  Elem elem = Elem::F32;
  std::vector<uint8_t> weights;      // 16-byte fragments
  std::vector<float> bias;
  std::vector<uint32_t> w_off;       // per layer, in 16-byte units
  std::vector<uint32_t> b_off;       // per layer, in floats
  std::vector<int> slots;            // per layer: input slots per lane-half
  std::vector<int> mtiles;           // per layer: 32-row output tiles
};

struct NetShape {
  int fp0 = 10, fd0 = 4;   // posEncArgs[0]  (oracle net input encoding)
  int fp1 = 10, fd1 = 4;   // posEncArgs[1]  (shading net input encoding)
};

// layers.{0..7}.{weight,bias}: [dir PE | pos PE] -> 256 x7 -> 128  (src/models.py:18-82,183-195)
bool pack_sampling_net(const TensorMap& net0, const NetShape& shape, Elem elem, PackedNet* out, std::string* err);

// pts_linears.{0..7}, feature_linear(+alpha_linear as row 256), views_linears.0, rgb_linear
// (src/models.py:199-277).  Layer order in the blob: 0..7, feature+alpha, views, rgb.
bool pack_shading_net(const TensorMap& net1, const NetShape& shape, Elem elem, PackedNet* out, std::string* err);

uint16_t f32_to_bf16(float f);
uint16_t f32_to_f16(float f);
float f16_to_f32(uint16_t h);

}  // namespace adanerf
