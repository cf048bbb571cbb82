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

// Topology of an exported network, read off its initializers (the reference's BaseNet / NeRF classes,
// src/models.py:18-82, 199-250; the viewer takes it from the ONNX graph as well, not from config.ini).
struct NetTopology {
  int depth = 8;           // Linear layers of the trunk (sampling net: all of them)
  int width = 256;         // hidden width the kernels run: 64 / 128 / 256 (the network's own width padded with zero units, pack.cpp pad_width)
  int real_width = 256;    // the network's own hidden width W
  int skip = -1;           // NeRF trunk: the first entry of the reference's `skips` (layer skip + 1 takes cat([input_pts, h])); -1 = none
  int cat_mask = 0;        // ... all of them: bit l set <=> layer l takes cat([input_pts, h]) (one bit per skip: the NeRF class accepts a list)
  int ray_samples = 0;     // sampling net: raySampleInput points in layer 0's input
  int bins = 128;          // sampling net: outputs = depth cells (multiDepthFeatures[0]); fewer than 128 run padded with kAbsentBin rows
  bool is_default(bool shading) const { return depth == 8 && width == 256 && ray_samples == 0 && cat_mask == (shading ? (1 << 5) : 0); }
};
constexpr float kAbsentBin = -1.0e30f;      // bias of the output rows a sampling net with fewer than 128 depth cells does not have
constexpr int kMaxDepth = 8;    // w_off / b_off tables hold depth + 3 entries (kMaxLayers = 12)

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
  NetTopology topo;
  uint32_t rsi_w_off = 0;            // fp32, ray_samples > 0: 16-byte offset of layer 0's raySampleInput fragments, [a][s4][m][lane][4]
  // bf16 shading nets are packed SCALED (pack.cpp scale_layer): every ReLU layer's weights and bias carry a power of two chosen so that no
  // activation can exceed 1 -- the 16-bit kernels then do ReLU + conversion with one clamped v_cvt_pk_bf16_f32 (k_mlp16.hip.hpp Bf16).
  // out_exp: the packed network's alpha / rgb outputs x 2^out_exp = the network's own (NetParams::out_scale).
  bool relu_scaled = false;
  int out_exp[2] = {0, 0};
};

// |value| bounds the scaled packing assumes for the identity slots of the two encodings (everything else in them is a sin / cos): sample
// positions after the config's normalisation, and ray directions (unit vectors).  Generous on purpose -- a power of two here only moves
// exponents: 2^12 for positions (InverseSqrtDistCentered keeps them below ~1.5; un-normalised scenes are world coordinates), 4 for directions.
constexpr double kPosIdentityBound = 4096.0;
constexpr double kDirIdentityBound = 4.0;

struct NetShape {
  int fp0 = 10, fd0 = 4;   // posEncArgs[0]  (oracle net input encoding)
  int fp1 = 10, fd1 = 4;   // posEncArgs[1]  (shading net input encoding)
  int ray_samples = 0;     // raySampleInput[0]: extra encoded points along the ray in the oracle net's input
  // layout bands per encoding (layout.hpp pe_col): 0 = the encoding's own band count (kernels instantiated for it), else the
  // catch-all kMaxBands layout of the run-time-shaped kernels
  int lp0 = 0, ld0 = 0, lp1 = 0, ld1 = 0;
};


// layers.{0..D-1}.{weight,bias}: [dir PE | pos PE | raySampleInput points] -> W x (D-1) -> 128  (src/models.py:18-82,183-195).
// Every shipped config is 8 x 256 without extra points; other depths (2..8) / widths (any W <= 256, run zero-padded to 64 / 128 / 256) pack
// for Elem::F32 and as split pairs; the raySampleInput input for Elem::F32 only (the generic fp32-MFMA kernels).
bool pack_sampling_net(const TensorMap& net0, const NetShape& shape, Elem elem, PackedNet* out, std::string* err);

// pts_linears.{0..D-1}, feature_linear(+alpha_linear as row W), views_linears.0, rgb_linear
// (src/models.py:199-277).  Layer order in the blob: 0..D-1, feature+alpha, views, rgb.  Any topology (any W <= 256, run zero-padded to
// 64 / 128 / 256; D in 1..8, skips anywhere) and encoding layout packs for every element type (16-bit: k_generic16.hip.hpp).
bool pack_shading_net(const TensorMap& net1, const NetShape& shape, Elem elem, PackedNet* out, std::string* err, bool scale_bf16 = true);
// (scale_bf16 = false: the bf16 blob WITHOUT the per-layer powers of two -- no kernel consumes it; the CPU test replays it next to the scaled one)

uint16_t f32_to_bf16(float f);
uint16_t f32_to_f16(float f);
float f16_to_f32(uint16_t h);

}  // namespace adanerf
