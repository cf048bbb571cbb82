#include "pack.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "layout.hpp"

namespace adanerf {

uint16_t f32_to_bf16(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return static_cast<uint16_t>((u >> 16) | 0x40);  // NaN
  u += 0x7FFFu + ((u >> 16) & 1u);   // round to nearest even
  return static_cast<uint16_t>(u >> 16);
}

uint16_t f32_to_f16(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t mant = x & 0x007FFFFFu;
  int32_t exp = static_cast<int32_t>((x >> 23) & 0xFF);
  if (exp == 0xFF) return static_cast<uint16_t>(sign | 0x7C00u | (mant ? 0x200u : 0));
  int32_t e = exp - 127 + 15;
  if (e >= 31) return static_cast<uint16_t>(sign | 0x7C00u);   // overflow -> inf
  if (e <= 0) {
    if (e < -10) return static_cast<uint16_t>(sign);
    mant |= 0x00800000u;
    uint32_t shift = static_cast<uint32_t>(14 - e);
    uint32_t half = mant >> shift;
    uint32_t rem = mant & ((1u << shift) - 1u);
    uint32_t mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1u))) half++;
    return static_cast<uint16_t>(sign | half);
  }
  uint32_t half = static_cast<uint32_t>(e << 10) | (mant >> 13);
  uint32_t rem = mant & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) half++;   // may carry into the exponent: correct
  return static_cast<uint16_t>(sign | half);
}

float f16_to_f32(uint16_t h) {
  uint32_t sign = static_cast<uint32_t>(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1Fu, mant = h & 0x3FFu;
  uint32_t u;
  if (exp == 0) {
    if (mant == 0) u = sign;
    else {   // subnormal: normalise
      int e = -1;
      do {
        mant <<= 1;
        e++;
      } while (!(mant & 0x400u));
      u = sign | static_cast<uint32_t>(127 - 15 - e) << 23 | ((mant & 0x3FFu) << 13);
    }
  } else if (exp == 31) u = sign | 0x7F800000u | (mant << 13);
  else u = sign | (exp + 127 - 15) << 23 | (mant << 13);
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

namespace {

// dense "virtual" layer: rows padded to 32, arbitrary source columns
struct VLayer {
  int rows = 0, cols = 0;          // cols = number of source columns
  std::vector<float> w;            // [rows][cols]
  std::vector<float> b;            // [rows]
  std::vector<int> col_h0, col_h1; // per slot: source column for lane-half 0 / 1 (-1 = zero)
};

const Tensor* find(const TensorMap& m, const std::string& k, std::string* err) {
  auto it = m.find(k);
  if (it == m.end()) {
    if (err) *err = "missing initializer " + k;
    return nullptr;
  }
  return &it->second;
}

bool set_rows(VLayer* L, int row0, const Tensor* W, const Tensor* B, int expect_cols, std::string* err,
              const std::string& name) {
  if (W->dims.size() != 2 || W->cols() != expect_cols || static_cast<int>(B->data.size()) != W->rows()) {
    if (err) *err = "unexpected shape for " + name + " (expected [*, " + std::to_string(expect_cols) + "])";
    return false;
  }
  for (int r = 0; r < W->rows(); ++r) {
    std::memcpy(&L->w[static_cast<size_t>(row0 + r) * L->cols], &W->data[static_cast<size_t>(r) * expect_cols],
                sizeof(float) * expect_cols);
    L->b[row0 + r] = B->data[r];
  }
  return true;
}

void init_layer(VLayer* L, int rows_padded, int cols) {
  L->rows = rows_padded;
  L->cols = cols;
  L->w.assign(static_cast<size_t>(rows_padded) * cols, 0.f);
  L->b.assign(rows_padded, 0.f);
}

void add_pe_slots(VLayer* L, int F, int col_base, int FL = 0) {
  if (FL <= 0) FL = F;
  int n = pe_slots(FL);
  for (int q = 0; q < n; ++q) {
    int c0 = pe_col(F, q, 0, FL), c1 = pe_col(F, q, 1, FL);
    L->col_h0.push_back(c0 < 0 ? -1 : col_base + c0);
    L->col_h1.push_back(c1 < 0 ? -1 : col_base + c1);
  }
}

// n_features: the (padded) width the kernels run; n_real: the network's own width -- features beyond it are padding (zero column)
void add_act_slots(VLayer* L, int n_features, int col_base, int n_real = -1) {
  if (n_real < 0) n_real = n_features;
  int n = n_features / 2;   // slots per lane-half
  for (int q = 0; q < n; ++q) {
    const int f0 = act_feature(q, 0), f1 = act_feature(q, 1);
    L->col_h0.push_back(f0 < n_real ? col_base + f0 : -1);
    L->col_h1.push_back(f1 < n_real ? col_base + f1 : -1);
  }
}

// The kernels are instantiated for hidden widths 64 / 128 / 256.  A network of any other width W <= 256 (the reference's layerWidth is
// free, src/models.py:18-82, 199-250) runs as the next of those with zero rows / zero bias appended to every hidden layer: a padded unit
// is relu(0) = 0 (the feature layer has no activation: 0 as well) and feeds zero columns -- the results are those of the W-wide network
// bit for bit in fp32, and to the engine's usual accuracy in 16 bits (the extra products are exact zeros).
int pad_width(int w) { return w <= 64 ? 64 : (w <= 128 ? 128 : (w <= 256 ? 256 : 0)); }

// ---- scaled packing of bf16 shading nets (PackedNet::relu_scaled) --------------------------------------------------------------------
// Source column c of a layer holds x_c 2^-ce[c] with |x_c| <= cb[c] (true scale).  The layer's outputs are bounded row by row,
// |y_r| <= sum_c |w_rc| cb[c] + |b_r|, with 2^-5 of slack for the bf16 rounding of the weights and of the inputs and the fp32 summation.
// choose: the output exponent e becomes the smallest with bound 2^-e <= 1 (ReLU layers: the kernel's clamped conversion must never
// clamp); else it is given (layers without an activation keep their input's exponent).  Weights and bias are rescaled in place by exact
// powers of two -- w_rc 2^(ce[c] - e), b_r 2^-e -- so the packed layer computes y 2^-e with the roundings of the unscaled one.
// Returns a negative value when no exponent within +-kMaxScaleExp brings the bound under 1 (or the bound is not finite): the clamped conversion would
// then cut real activations, so the caller refuses the bf16 packing of this network (never a silent clamp).
constexpr int kMaxScaleExp = 100;
double scale_layer(VLayer* L, const std::vector<double>& cb, const std::vector<int>& ce, bool choose, int* e) {
  double bound = 0.0;
  for (int r = 0; r < L->rows; ++r) {
    double s = std::fabs(static_cast<double>(L->b[r]));
    for (int c = 0; c < L->cols; ++c) s += std::fabs(static_cast<double>(L->w[static_cast<size_t>(r) * L->cols + c])) * cb[c];
    bound = std::max(bound, s);
  }
  bound *= 1.03125;
  if (!(bound < 1e300)) return -1.0;      // inf / NaN weights
  if (choose) {
    *e = bound > 0.0 ? static_cast<int>(std::ceil(std::log2(bound))) : 0;
    if (*e > kMaxScaleExp) return -1.0;      // the bound is loose by construction (products of L1 row norms): beyond this range it proves nothing any more
    if (*e < -kMaxScaleExp) *e = -kMaxScaleExp;      // a (near-)zero layer: scaling UP less than possible is always safe
  }
  for (int r = 0; r < L->rows; ++r) {
    L->b[r] = std::ldexp(L->b[r], -*e);
    for (int c = 0; c < L->cols; ++c) {
      float& w = L->w[static_cast<size_t>(r) * L->cols + c];
      w = std::ldexp(w, ce[c] - *e);
    }
  }
  return bound;
}

// bounds / exponents of the source columns of an encoding: 3 identity columns, then sin / cos
void pe_col_scale(int n_cols, double identity_bound, std::vector<double>* cb, std::vector<int>* ce) {
  for (int c = 0; c < n_cols; ++c) {
    cb->push_back(c < 3 ? identity_bound : 1.0);
    ce->push_back(0);
  }
}

void emit(const VLayer& L, Elem elem, PackedNet* out) {
  const int G = (elem == Elem::F32) ? 4 : 8;          // slots per 16-byte fragment element group
  const int QS = static_cast<int>(L.col_h0.size());
  const int steps = QS / G;
  const int MT = L.rows / 32;
  const int parts = (elem == Elem::F16_SPLIT) ? 2 : 1;
  out->w_off.push_back(static_cast<uint32_t>(out->weights.size() / 16));
  out->b_off.push_back(static_cast<uint32_t>(out->bias.size()));
  out->slots.push_back(QS);
  out->mtiles.push_back(MT);
  size_t base = out->weights.size();
  out->weights.resize(base + static_cast<size_t>(MT) * steps * parts * 64 * 16);
  uint8_t* dst = out->weights.data() + base;
  for (int m = 0; m < MT; ++m)
    for (int s = 0; s < steps; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        int i = lane & 31, h = lane >> 5;
        int row = 32 * m + i;
        for (int e = 0; e < G; ++e) {
          int q = G * s + e;
          int col = h ? L.col_h1[q] : L.col_h0[q];
          float v = (col >= 0) ? L.w[static_cast<size_t>(row) * L.cols + col] : 0.f;
          size_t frag = (static_cast<size_t>(m) * steps + s) * parts * 64 + lane;
          if (elem == Elem::F32) {
            std::memcpy(dst + frag * 16 + 4 * e, &v, 4);
          } else if (elem == Elem::F16_SPLIT) {
            uint16_t hi = f32_to_f16(v);
            float lo = (v - f16_to_f32(hi)) * kSplitScale;
            uint16_t lv = f32_to_f16(lo);
            std::memcpy(dst + frag * 16 + 2 * e, &hi, 2);
            std::memcpy(dst + (frag + 64) * 16 + 2 * e, &lv, 2);
          } else {
            uint16_t hv = (elem == Elem::BF16) ? f32_to_bf16(v) : f32_to_f16(v);
            std::memcpy(dst + frag * 16 + 2 * e, &hv, 2);
          }
        }
      }
  for (int m = 0; m < MT; ++m)
    for (int h = 0; h < 2; ++h)
      for (int r = 0; r < 16; ++r) out->bias.push_back(L.b[32 * m + 8 * (r >> 2) + 4 * h + (r & 3)]);
}

}  // namespace

namespace {

int count_layers(const TensorMap& m, const std::string& prefix) {
  int n = 0;
  while (m.count(prefix + std::to_string(n) + ".weight")) ++n;
  return n;
}

const std::string kScaleRefused = "shading net (bf16): the activation bound of a layer leaves the range the scaled packing can prove anything in (2^100): "
                                  "use fp16 / fp32 shading for this network; layer ";
bool fail(std::string* err, const std::string& msg) {
  if (err) *err = msg;
  return false;
}

// raySampleInput part of the sampling net's first layer: K-major fragments [a][s4][m][lane][4] (fp32 engine only), so the
// kernel can walk the A extra points in a run-time loop with all MT accumulators live.
void emit_ray_samples(const VLayer& L, int A, int fp, int col_base, PackedNet* out, int FL = 0) {
  if (FL <= 0) FL = fp;
  const int QP = pe_slots(FL), MT = L.rows / 32, n_pt = 3 + 6 * fp;
  out->rsi_w_off = static_cast<uint32_t>(out->weights.size() / 16);
  const size_t base = out->weights.size();
  out->weights.resize(base + static_cast<size_t>(A) * (QP / 4) * MT * 64 * 16);
  uint8_t* dst = out->weights.data() + base;
  for (int a = 0; a < A; ++a)
    for (int s4 = 0; s4 < QP / 4; ++s4)
      for (int m = 0; m < MT; ++m)
        for (int lane = 0; lane < 64; ++lane) {
          const int row = 32 * m + (lane & 31), h = lane >> 5;
          const size_t frag = ((static_cast<size_t>(a) * (QP / 4) + s4) * MT + m) * 64 + lane;
          for (int e = 0; e < 4; ++e) {
            const int c = pe_col(fp, 4 * s4 + e, h, FL);
            const float v = c < 0 ? 0.f : L.w[static_cast<size_t>(row) * L.cols + col_base + a * n_pt + c];
            std::memcpy(dst + frag * 16 + 4 * e, &v, 4);
          }
        }
}

}  // namespace

// what every caller's NetShape must satisfy before any arithmetic on it (adanerf_create checks the same in setup_model; the host-only entry
// point adanerf_host_pack_weights builds its shape straight from config.ini): band counts the slot layouts exist for, raySampleInput in range
static bool shape_ok(const NetShape& sh, std::string* err) {
  for (int f : {sh.fp0, sh.fd0, sh.fp1, sh.fd1})
    if (f < 0 || f > kMaxBands) return fail(err, "posEncArgs: 0.." + std::to_string(kMaxBands) + " frequency bands are supported");
  if (sh.ray_samples < 0 || sh.ray_samples > 1024) return fail(err, "raySampleInput[0] must be in 0..1024");
  return true;
}

bool pack_sampling_net(const TensorMap& net0, const NetShape& sh, Elem elem, PackedNet* out, std::string* err) {
  *out = PackedNet();
  out->elem = elem;
  if (!shape_ok(sh, err)) return false;
  const int n_dir = 3 + 6 * sh.fd0, n_pos = 3 + 6 * sh.fp0;
  const int n_in = n_dir + n_pos + sh.ray_samples * n_pos;
  NetTopology& T = out->topo;
  T.depth = count_layers(net0, "layers.");
  if (T.depth < 2 || T.depth > kMaxDepth) return fail(err, "sampling net: " + std::to_string(T.depth) + " layers (2.." + std::to_string(kMaxDepth) + " supported)");
  const int n_bins = find(net0, "layers." + std::to_string(T.depth - 1) + ".weight", err)->rows();      // multiDepthFeatures[0]
  T.bins = n_bins;
  if (n_bins < 1 || n_bins > kBins) return fail(err, "sampling net: " + std::to_string(n_bins) + " outputs (multiDepthFeatures 1 .. 128 supported)");
  const int Wr = find(net0, "layers.0.weight", err)->rows();     // exists: depth >= 2.  The network's own width ...
  T.width = pad_width(Wr);                                       // ... and the width it runs at
  T.real_width = Wr;
  T.ray_samples = sh.ray_samples;
  if (Wr < 1 || T.width == 0) return fail(err, "sampling net: width " + std::to_string(Wr) + " (1 .. 256 supported)");
  // plain 16-bit fragments: the ring-streamed kernel of the 8 x 256 / 10-4 or 2-2 net only; the (hi, lo') split pairs pack for any
  // topology and layout without raySampleInput (k_generic16.hip.hpp); raySampleInput is fp32 only (K-major block, emit_ray_samples)
  const bool special = T.is_default(false) && !sh.lp0 && !sh.ld0;
  if (!special && elem != Elem::F32 && !(elem == Elem::F16_SPLIT && sh.ray_samples == 0))
    return fail(err, "sampling net: only the 8 x 256 topology without raySampleInput and with a 10-4 or 2-2 encoding runs on the plain 16-bit engine "
                     "(the split-precision packing exists for every topology without raySampleInput)");
  for (int i = 0; i < T.depth; ++i) {
    const Tensor* W = find(net0, "layers." + std::to_string(i) + ".weight", err);
    const Tensor* B = find(net0, "layers." + std::to_string(i) + ".bias", err);
    if (!W || !B) return false;
    const int n_out = (i == T.depth - 1) ? n_bins : Wr;
    const int k = (i == 0) ? n_in : Wr;           // BaseNet skip specs (src/models.py:44-66) are not used by any config: plain chain
    if (W->rows() != n_out) {
      if (err) *err = "layers." + std::to_string(i) + ".weight: expected " + std::to_string(n_out) + " rows";
      return false;
    }
    VLayer L;
    init_layer(&L, (i == T.depth - 1) ? kBins : T.width, k);      // hidden layers: rows padded to the width the kernels run
    if (!set_rows(&L, 0, W, B, k, err, "layers." + std::to_string(i))) return false;
    // multiDepthFeatures = D < 128: the selection kernels work on 128 values per ray; the bins the network does not have get zero
    // weights and a bias no threshold, arg-max or softmax ever picks (-1e30: exp() of it is 0, a sigmoid of it 0)
    if (i == T.depth - 1)
      for (int r = n_bins; r < kBins; ++r) L.b[r] = kAbsentBin;
    if (i == 0) {
      add_pe_slots(&L, sh.fd0, 0, sh.ld0);        // [dir PE | pos PE]  (src/features.py:868-874)
      add_pe_slots(&L, sh.fp0, n_dir, sh.lp0);
    } else {
      add_act_slots(&L, T.width, 0, Wr);
    }
    emit(L, elem, out);
    if (i == 0 && sh.ray_samples > 0) emit_ray_samples(L, sh.ray_samples, sh.fp0, n_dir + n_pos, out, sh.lp0);
  }
  return true;
}

bool pack_shading_net(const TensorMap& net1, const NetShape& sh, Elem elem, PackedNet* out, std::string* err, bool scale_bf16) {
  *out = PackedNet();
  out->elem = elem;
  if (!shape_ok(sh, err)) return false;
  const int n_pos = 3 + 6 * sh.fp1, n_dir = 3 + 6 * sh.fd1;
  NetTopology& T = out->topo;
  T.depth = count_layers(net1, "pts_linears.");
  if (T.depth < 1 || T.depth > kMaxDepth) return fail(err, "shading net: " + std::to_string(T.depth) + " trunk layers (1.." + std::to_string(kMaxDepth) + " supported)");
  const int Wr = find(net1, "pts_linears.0.weight", err)->rows();   // exists: depth >= 1.  The network's own width W ...
  T.width = pad_width(Wr);                                          // ... and the width it runs at (pad_width)
  T.real_width = Wr;
  if (Wr < 2 || T.width == 0) return fail(err, "shading net: width " + std::to_string(Wr) + " (2 .. 256 supported)");
  const int Wd = T.width, Wh = Wr / 2;      // views_linears.0 has W // 2 rows (src/models.py:236)
  // bf16: scaled packing (scale_layer) -- h_bound / h_exp describe the trunk's current activations
  const bool scaled = elem == Elem::BF16 && scale_bf16;
  out->relu_scaled = scaled;
  double h_bound = 0.0, f_bound = 0.0, v_bound = 0.0;
  int h_exp = 0, v_exp = 0;
  T.skip = -1;
  for (int i = 1; i < T.depth; ++i) {
    const Tensor* W = find(net1, "pts_linears." + std::to_string(i) + ".weight", err);
    if (!W) return false;
    if (W->cols() == Wr + n_pos) {          // cat([input_pts, h]) in front of layer i  <=>  i - 1 in skips (src/models.py:226-228)
      if (T.skip < 0) T.skip = i - 1;
      T.cat_mask |= 1 << i;
    }
  }
  for (int i = 0; i < T.depth; ++i) {
    const std::string nm = "pts_linears." + std::to_string(i);
    const Tensor* W = find(net1, nm + ".weight", err);
    const Tensor* B = find(net1, nm + ".bias", err);
    if (!W || !B) return false;
    const bool cat = i > 0 && ((T.cat_mask >> i) & 1);
    const int k = (i == 0) ? n_pos : (cat ? n_pos + Wr : Wr);
    if (W->rows() != Wr) {
      if (err) *err = nm + ".weight: expected " + std::to_string(Wr) + " rows";
      return false;
    }
    VLayer L;
    init_layer(&L, Wd, k);
    if (!set_rows(&L, 0, W, B, k, err, nm)) return false;
    if (i == 0) {
      add_pe_slots(&L, sh.fp1, 0, sh.lp1);
    } else if (cat) {                      // cat([input_pts, h])  (src/models.py:260-261)
      add_pe_slots(&L, sh.fp1, 0, sh.lp1);
      add_act_slots(&L, Wd, n_pos, Wr);
    } else {
      add_act_slots(&L, Wd, 0, Wr);
    }
    if (scaled) {
      std::vector<double> cb;
      std::vector<int> ce;
      if (i == 0 || cat) pe_col_scale(n_pos, kPosIdentityBound, &cb, &ce);
      if (i > 0) {
        cb.insert(cb.end(), Wr, h_bound);
        ce.insert(ce.end(), Wr, h_exp);
      }
      h_bound = scale_layer(&L, cb, ce, true, &h_exp);
      if (h_bound < 0.0) return fail(err, kScaleRefused + ("pts_linears." + std::to_string(i)));
    }
    emit(L, elem, out);
  }
  {   // feature_linear rows 0..W-1, alpha_linear as row W (tile W/32, row 0)
    const Tensor* WF = find(net1, "feature_linear.weight", err);
    const Tensor* BF = find(net1, "feature_linear.bias", err);
    const Tensor* WA = find(net1, "alpha_linear.weight", err);
    const Tensor* BA = find(net1, "alpha_linear.bias", err);
    if (!WF || !BF || !WA || !BA) return false;
    VLayer L;
    init_layer(&L, Wd + 32, Wr);
    if (!set_rows(&L, 0, WF, BF, Wr, err, "feature_linear")) return false;
    if (WF->rows() != Wr || WA->rows() != 1) {
      if (err) *err = "feature_linear/alpha_linear: unexpected row count";
      return false;
    }
    if (!set_rows(&L, Wd, WA, BA, Wr, err, "alpha_linear")) return false;      // the alpha row sits behind the PADDED feature rows
    add_act_slots(&L, Wd, 0, Wr);
    if (scaled) {      // no activation: the layer keeps its input's exponent; alpha leaves the kernel times 2^h_exp
      const std::vector<double> cb(Wr, h_bound);
      const std::vector<int> ce(Wr, h_exp);
      int e = h_exp;
      f_bound = scale_layer(&L, cb, ce, false, &e);
      if (f_bound < 0.0) return fail(err, kScaleRefused + std::string("feature_linear"));
      out->out_exp[0] = h_exp;
    }
    emit(L, elem, out);
  }
  {   // views_linears.0 on cat([feature, input_views])  (src/models.py:266-270)
    const Tensor* W = find(net1, "views_linears.0.weight", err);
    const Tensor* B = find(net1, "views_linears.0.bias", err);
    if (!W || !B) return false;
    if (W->rows() != Wh) {
      if (err) *err = "views_linears.0.weight: expected " + std::to_string(Wh) + " rows";
      return false;
    }
    VLayer L;
    init_layer(&L, Wd / 2, Wr + n_dir);
    if (!set_rows(&L, 0, W, B, Wr + n_dir, err, "views_linears.0")) return false;
    add_act_slots(&L, Wd, 0, Wr);
    add_pe_slots(&L, sh.fd1, Wr, sh.ld1);
    if (scaled) {      // cat([feature (the trunk's exponent), dir encoding (unscaled)]) -> ReLU
      std::vector<double> cb(Wr, f_bound);
      std::vector<int> ce(Wr, h_exp);
      pe_col_scale(n_dir, kDirIdentityBound, &cb, &ce);
      v_bound = scale_layer(&L, cb, ce, true, &v_exp);
      if (v_bound < 0.0) return fail(err, kScaleRefused + std::string("views_linears.0"));
    }
    emit(L, elem, out);
  }
  {   // rgb_linear W/2 -> 3 (tile 0 rows 0..2)
    const Tensor* W = find(net1, "rgb_linear.weight", err);
    const Tensor* B = find(net1, "rgb_linear.bias", err);
    if (!W || !B) return false;
    if (W->rows() != 3) {
      if (err) *err = "rgb_linear.weight: expected 3 rows";
      return false;
    }
    VLayer L;
    init_layer(&L, 32, Wh);
    if (!set_rows(&L, 0, W, B, Wh, err, "rgb_linear")) return false;
    add_act_slots(&L, Wd / 2, 0, Wh);
    if (scaled) {      // no activation: rgb leaves the kernel times 2^v_exp
      const std::vector<double> cb(Wh, v_bound);
      const std::vector<int> ce(Wh, v_exp);
      int e = v_exp;
      scale_layer(&L, cb, ce, false, &e);
      out->out_exp[1] = v_exp;
    }
    emit(L, elem, out);
  }
  return true;
}

}  // namespace adanerf
