// C ABI of libadanerf_hip.so (see include/adanerf_hip.h).  Host side: model loading, weight
// packing/upload, buffer ownership, per-frame launch sequence on the context's own HIP stream.
#include "../../include/adanerf_hip.h"

#include <hip/hip_runtime.h>

#include <sys/stat.h>

#include <algorithm>
#include <type_traits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "format.hpp"
#include "kernels.hip.hpp"
#include "launch_f32.hpp"
#include "k_generic_f32.hip.hpp"   // GenericTopo (the kernels themselves live in launch_f32.hip)
#include "k_generic16.hip.hpp"     // the 16-bit form of the run-time-shaped shading kernel
#include "pack.hpp"

using namespace adanerf;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct PackedDev {
  DevBuf w, b;
  NetParams params{};
};

// What bounds the sample positions the bf16 shading path's scaled layers may see (pack.hpp kPosIdentityBound): kept so that adanerf_set_camera can
// check a pose outside the view cell (a free-fly viewer) instead of letting the clamped conversion cut activations silently.
struct PosBound {
  bool active = false;      // bf16 shading and not NDC
  int normalize = 0;
  double cmax = 0, zmax = 0, off = 0, M = 1, rad = 0;
  double center[3] = {0, 0, 0};
  // largest |encoded position| for a camera `dcam` away from the view-cell centre: the ray starts on (or, outside the cell, within dcam + rad of) the
  // cell's sphere and runs zmax further along a unit direction
  double at(double dcam) const {
    const double reach = std::max(2.0 * rad, 2.0 * dcam + rad);
    const double world = cmax + reach + zmax, local = reach + zmax + off;
    if (normalize == kNormMaxDepth) return world / M;
    if (normalize == kNormCentered) return local;
    if (normalize == kNormMaxDepthCentered) return local / M;
    if (normalize == kNormInverseSqrtDistCentered) return std::sqrt(local / M);
    if (normalize != kNormNone) return local;      // InverseDistCentered (<= |l|), LogCentered (<= |l| for max_depth >= e - 1 ... kept loose)
    return world;
  }
};

}  // namespace

struct adanerf_ctx {
  adanerf_options opt{};
  adanerf_info info{};
  PosBound pos_bound;
  Config cfg;
  std::string err;
  hipStream_t stream = nullptr;       // stream in use
  hipStream_t own_stream = nullptr;   // the context's own stream
  std::vector<hipEvent_t> events;     // pool; 5 per batch
  size_t events_used = 0;
  bool profiling = false;
  int prof_frames = 0;
  std::vector<int32_t*> pinned_totals;   // one pinned int32 per recorded batch
  // always-on monitor of the guarded selection (ADANERF_SAMPLING_GUARDED): the device's running {largest difference seen, violations,
  // largest pair error seen, audit mismatches, audited rays} copied to pinned memory after every frame and looked at before the next
  // one -- a violated band (or a failed audit) widens it (poll_guard)
  int32_t* guard_host = nullptr;
  hipEvent_t guard_ev = nullptr;
  bool guard_ev_pending = false;
  int guard_viol_seen = 0;
  int guard_mism_seen = 0;
  int guard_widened = 0;
  uint32_t guard_frame = 0;              // frames rendered in guarded mode: the audit's rotating phase
  int debug_guard = 0;                   // $ADANERF_DEBUG_GUARD at create (measurement knobs, bit 0: no whole-row monitor; bits 1 / 2: which plain-fp16 sampling kernel)
  int sample16x2_grid = 0;
  std::string model_dir;
  uint64_t model0_hash = 0;              // FNV-1a 64 of model0.onnx: key of the calibration record
  adanerf_stats folded{};                // profiling record folded out of a full event pool (see adanerf_render)

  RayGenParams rg{};
  ShadeParams sp{};
  int mult_mode = 1;   // 0 none, 1 alpha, 2 weights
  int transform = 0;   // kOracle*: sampler's transform of the raw oracle outputs (losses[0])
  int fp0 = 10, fd0 = 4, fp1 = 10, fd1 = 4;
  int ray_samples = 0;            // raySampleInput[0]
  NetTopology topo0, topo1;       // read off the ONNX initializers
  bool generic0 = false, generic1 = false;   // not the 8 x 256 (/ skip 4) topology or not a 10-4 (2-2) encoding: the run-time-shaped fp32 kernels (k_generic_f32.hip.hpp)
  int enc0 = kEnc10_4, enc1 = kEnc10_4;      // slot layout of the two networks' encodings (launch_f32.hpp)
  GenericTopo gen0{}, gen1{};
  DevBuf rsi_z;                   // [ray_samples] world depths of the raySampleInput points
  int shade_gen_grid = 0;
  int shade_gen16_grid[2][2] = {{0, 0}, {0, 0}};      // [fine | coarse net][bf16 | fp16]: resident workgroups of the run-time-shaped 16-bit kernel

  PackedDev net0;                 // sampling net, fp32 fragments (exact engine)
  PackedDev net0_split;           // sampling net, fp16 hi/lo' fragment pairs (split-precision engine)
  PackedDev net0_f16;             // sampling net, plain fp16 fragments (ADANERF_SAMPLING_FP16, packed on first use)
  TensorMap net0_host;
  int sample16_grid = 0;
  int sampling_mode = 0;          // 0: split-precision fp16x3 (default), 1: exact fp32 MFMA, 2: plain fp16
  DevBuf overflow;                // int32 counter: rays whose oracle values came out non-finite
  int sample_grid = 0;
  PackedDev net1[3];              // shading net per precision (packed lazily)
  TensorMap net1_host;
  DevBuf ztab;
  // vanilla NeRF (ADANERF_SAMPLER_COARSE_FINE): model0.onnx is a NeRF net as well, evaluated at n_coarse uniform depths
  bool coarse_fine = false;
  int n_coarse = 0;
  PackedDev netc[3];              // coarse net per precision (packed lazily)
  NetTopology topoc;
  bool genericc = false;
  GenericTopo genc{};
  ShadeParams spc{};              // its sample positions: uniform depth table, rayMarchNormalization[0]
  DevBuf ztab_coarse, raw_coarse, key_coarse;
  int shade_gen_grid_c = 0;

  // per-batch buffers
  int cap_rays = 0, cap_nmax = 0;
  DevBuf rays, oracle, ray_offsets, ray_counts, selbin, selw, block_total, block_offset, total;
  DevBuf sample_key, sample_w, raw, sample_z;
  DevBuf guard_mask, refine_list, guard_probe;   // ADANERF_SAMPLING_GUARDED: undecided bit per ray (one word per 32), ids of the
                                                 // rays to re-evaluate, first-pass top value of each undecided ray (monitor)
  float guard_eps = 0.f;             // the band in raw-output units; 0: not calibrated yet
  float guard_eps_pair = 0.f;        // bound on the error of a (kept - candidate) difference, raw-output units; 0: 2 x guard_eps
  int guard_audit_period = 0;        // 0: no audit; else a power of two <= 32
  DepthMap dm{};
  int shade_grid[3] = {0, 0, 0};
  int device = 0;                 // HIP device ordinal this context lives on
  float* aux_depth = nullptr;        // adanerf_set_aux_outputs: caller-owned [rays_local] buffers filled by adanerf_render
  float* aux_acc = nullptr;
  float* aux_disp = nullptr;         // adanerf_set_disp_output
  DevBuf disp_scratch;               // [2, batch] depth / accumulation of the batch when the caller asked for disparity only
  hipEvent_t peer_event = nullptr;   // adanerf_gather_to: orders the destination stream behind the copy
  uint64_t peer_tried = 0;           // bit d: peer access to device d has been requested once
};

namespace {

#define HIP_TRY(ctx, expr)                                                                  \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                       \
      return ADANERF_EDEVICE;                                                               \
    }                                                                                       \
  } while (0)

// Every entry point that touches the device first makes the context's device current: one host thread may drive
// several contexts on different GPUs (adanerf_amd/host --gpus N), and a launch on another device's stream fails.
#define BIND(ctx) HIP_TRY(ctx, hipSetDevice((ctx)->device))

int fail(adanerf_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  else g_create_error = msg;
  return code;
}

int dev_alloc(adanerf_ctx* c, DevBuf* b, size_t bytes) {
  if (b->bytes >= bytes && b->p) return ADANERF_OK;
  if (b->p) HIP_TRY(c, hipFree(b->p));
  b->p = nullptr;
  b->bytes = 0;
  if (bytes == 0) return ADANERF_OK;
  HIP_TRY(c, hipMalloc(&b->p, bytes));
  b->bytes = bytes;
  return ADANERF_OK;
}

void dev_free(DevBuf* b) {
  if (b->p) (void)hipFree(b->p);
  b->p = nullptr;
  b->bytes = 0;
}

int upload_net(adanerf_ctx* c, const PackedNet& pn, PackedDev* d) {
  int rc;
  if ((rc = dev_alloc(c, &d->w, pn.weights.size()))) return rc;
  if ((rc = dev_alloc(c, &d->b, pn.bias.size() * sizeof(float)))) return rc;
  HIP_TRY(c, hipMemcpy(d->w.p, pn.weights.data(), pn.weights.size(), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(d->b.p, pn.bias.data(), pn.bias.size() * sizeof(float), hipMemcpyHostToDevice));
  d->params.w = reinterpret_cast<const u32x4*>(d->w.p);
  d->params.bias = reinterpret_cast<const float*>(d->b.p);
  for (size_t i = 0; i < pn.w_off.size() && i < kMaxLayers; ++i) {
    d->params.w_off[i] = pn.w_off[i];
    d->params.b_off[i] = pn.b_off[i];
  }
  d->params.n_bias = static_cast<uint32_t>(pn.bias.size());
  d->params.out_scale[0] = std::ldexp(1.0f, pn.out_exp[0]);      // 1 unless the bf16 packing scaled the layers (pack.cpp scale_layer)
  d->params.out_scale[1] = std::ldexp(1.0f, pn.out_exp[1]);
  return ADANERF_OK;
}


struct ModelSetup {
  PosBound pos_bound;
  Config cfg;
  adanerf_info info{};
  RayGenParams rg{};
  ShadeParams sp{};
  int mult_mode = 1;
  int transform = 0;
  int fp0 = 10, fd0 = 4, fp1 = 10, fd1 = 4;
  int ray_samples = 0;
  std::vector<float> rsi_z;
  std::vector<float> ztab;
  int bins = 128;                 // multiDepthFeatures: depth cells of the adaptive sampler (src/nerf_raymarch_common.py:675-677, 726-727)
  DepthMap dm{};
  bool coarse_fine = false;
  int n_coarse = 0;
  std::vector<float> ztab_coarse;
  int normalize0 = 0;      // kNorm* of the coarse pass
};

// slot layout of an encoding pair: the specialised kernels exist for 10-4 (both networks) and 2-2 (sampling network); any other
// pair is packed into the catch-all kMaxBands layout and runs on the run-time-shaped fp32 kernels
int enc_layout(int fp, int fd, bool sampling) {
  if (fp == 10 && fd == 4) return kEnc10_4;
  if (sampling && fp == 2 && fd == 2) return kEnc2_2;
  return kEncMax;
}
NetShape shape_of(int fp0, int fd0, int fp1, int fd1, int ray_samples, bool net0_is_sampling = true) {
  NetShape sh{fp0, fd0, fp1, fd1, ray_samples};
  if (enc_layout(fp0, fd0, net0_is_sampling) == kEncMax) sh.lp0 = sh.ld0 = kMaxBands;
  if (enc_layout(fp1, fd1, false) == kEncMax) sh.lp1 = sh.ld1 = kMaxBands;
  return sh;
}

Elem elem_of(int prec) {
  return prec == ADANERF_PREC_BF16 ? Elem::BF16 : (prec == ADANERF_PREC_FP16 ? Elem::F16 : (prec == 3 ? Elem::F16_SPLIT : Elem::F32));
}

// rows of the image owned by `rank` under round-robin strips
int rows_of_rank(int h, int strip_rows, int world, int rank) {
  int n_strips = (h + strip_rows - 1) / strip_rows;
  int rows = 0;
  for (int s = rank; s < n_strips; s += world) rows += std::min(strip_rows, h - s * strip_rows);
  return rows;
}

bool contains(const std::string& s, const char* sub) { return s.find(sub) != std::string::npos; }

// Host-only: parse + validate the model directory and derive every per-context constant.
int setup_model_unguarded(const char* model_dir, const adanerf_options* opt, ModelSetup* ms, std::string* err);
// No exception leaves the C ABI: whatever a damaged or hostile model directory makes the loader throw (std::bad_alloc, std::length_error)
// comes back as a status + message (tests/host_sanitize_fuzz.cpp runs the loader itself under ASan / UBSan on mutated directories).
int setup_model(const char* model_dir, const adanerf_options* opt, ModelSetup* ms, std::string* err) {
  try {
    return setup_model_unguarded(model_dir, opt, ms, err);
  } catch (const std::exception& e) {
    *err = std::string("model directory: ") + e.what();
    return ADANERF_EIO;
  }
}
int setup_model_unguarded(const char* model_dir, const adanerf_options* opt, ModelSetup* ms, std::string* err) {
  auto bad = [&](int code, const std::string& msg) {
    *err = msg;
    return code;
  };
  if (opt->width <= 0 || opt->height <= 0) return bad(ADANERF_EINVAL, "width/height must be positive");
  if (!ms->cfg.load(model_dir, err)) return ADANERF_EIO;
  const Config& cf = ms->cfg;

  // ---- validate the configuration against the supported (north-star) path ----
  const bool coarse_fine = cf.inFeatures.size() == 2 && cf.inFeatures[0] == "RayMarchFromPoses" && cf.inFeatures[1] == "RayMarchFromCoarse";
  ms->coarse_fine = coarse_fine;
  if (!coarse_fine && (cf.inFeatures.size() != 2 || cf.inFeatures[0] != "SpherePosDir" || cf.inFeatures[1] != "RayMarchFromPoses"))
    return bad(ADANERF_EUNSUPPORTED, "inFeatures must be [SpherePosDir, RayMarchFromPoses] or [RayMarchFromPoses, RayMarchFromCoarse]");
  if (cf.posEnc.size() != 2 || cf.posEnc[0] != "nerf" || cf.posEnc[1] != "nerf" || cf.posEncArgs.size() != 2)
    return bad(ADANERF_EUNSUPPORTED, "posEnc must be [nerf, nerf] with two posEncArgs entries");
  const bool pdf_mode = !coarse_fine && cf.rayMarchSampler.size() == 2 && cf.rayMarchSampler[1] == "FromClassifiedDepth";
  if (coarse_fine) {
    // RayMarchFromPoses without an oracle in front draws its depths from rayMarchSampler[0] (src/features.py:431-436)
    if (cf.rayMarchSampler.empty() || cf.rayMarchSampler[0] != "LinearlySpacedZNearZFar")
      return bad(ADANERF_EUNSUPPORTED, "coarse/fine: rayMarchSampler[0] must be LinearlySpacedZNearZFar");
    if (cf.numRaymarchSamples.size() != 2) return bad(ADANERF_EIO, "coarse/fine: numRaymarchSamples must be [Nc, Nf]");
  } else if (cf.rayMarchSampler.size() != 2 || (!pdf_mode && !contains(cf.rayMarchSampler[1], "FromClassifiedDepthAdaptive")))
    return bad(ADANERF_EUNSUPPORTED, "rayMarchSampler[1] must be FromClassifiedDepthAdaptive[NoDepthRange] or FromClassifiedDepth");
  // The transform every sampler applies to the raw oracle outputs follows losses[0] (src/nerf_raymarch_common.py:624-630,
  // 686-690, 782-788).  A model directory without a losses key (the trimmed 19-key config.ini) is an AdaNeRF export
  // (NeRFWeightMultiplicationLoss: no transform) -- except under FromClassifiedDepth, where the viewer's samplePDF
  // (base_cuda_kernels.cu:296-372) and every DONeRF config apply the sigmoid.
  {
    const std::string l0 = cf.losses.empty() ? std::string(pdf_mode ? "BCEWithLogitsLoss" : "NeRFWeightMultiplicationLoss") : cf.losses[0];
    ms->transform = l0 == "BCEWithLogitsLoss" ? kOracleSigmoid : ((l0 == "CrossEntropyLoss" || l0 == "CrossEntropyLossWeighted") ? kOracleSoftmax : kOracleRaw);
  }
  // raySampleInput[0] = A > 0: A extra encoded points along the ray in the oracle net's input (src/features.py:876-888);
  // the shading net takes no such input on this path (RayMarchFromPoses ignores the key)
  ms->ray_samples = cf.raySampleInput.empty() ? 0 : cf.raySampleInput[0];
  if (ms->ray_samples < 0 || ms->ray_samples > 1024) return bad(ADANERF_EUNSUPPORTED, "raySampleInput[0] must be in 0..1024");
  if (cf.viewcellCenter.size() != 3 || cf.viewcellSize.size() != 3 || cf.depthRange.size() != 2 || cf.fov <= 0.0)
    return bad(ADANERF_EIO, "dataset_info.txt: view_cell_center/view_cell_size/depth_range/fov missing or malformed");
  if (cf.numRaymarchSamples.empty()) return bad(ADANERF_EIO, "config.ini: numRaymarchSamples missing");
  ms->fp0 = static_cast<int>(cf.posEncArgs[0][0]);
  ms->fd0 = static_cast<int>(cf.posEncArgs[0][1]);
  ms->fp1 = static_cast<int>(cf.posEncArgs[1][0]);
  ms->fd1 = static_cast<int>(cf.posEncArgs[1][1]);
  // any F_pos-F_dir the reference's "nerf" encoding accepts (src/util/feature_encoding.py:54-73; viewer config.cpp:142-146) up to
  // kMaxBands bands: 10-4 (both nets) and 2-2 (sampling net) run on the specialised kernels, every other pair on the
  // run-time-shaped fp32 kernels with the catch-all slot layout
  for (int f : {ms->fp0, ms->fd0, ms->fp1, ms->fd1})
    if (f < 0 || f > kMaxBands) return bad(ADANERF_EUNSUPPORTED, "posEncArgs: 0.." + std::to_string(kMaxBands) + " frequency bands are supported");
  const bool ndc = cf.useNDC;
  const bool no_range = !coarse_fine && contains(cf.rayMarchSampler[1], "NoDepthRange");
  if (!pdf_mode && !coarse_fine && ndc != no_range) return bad(ADANERF_EUNSUPPORTED, "useNDC requires the NoDepthRange sampler and vice versa");
  // every function nerf_get_normalization_function knows (src/nerf_raymarch_common.py:195-244); a config WITHOUT the key gets
  // normalization_max_depth (src/features.py:319-324)
  auto norm_code = [](const std::string& n) {
    return n == "None" ? kNormNone : n == "InverseSqrtDistCentered" ? kNormInverseSqrtDistCentered : n == "Centered" ? kNormCentered
         : n == "MaxDepth" ? kNormMaxDepth : n == "MaxDepthCentered" ? kNormMaxDepthCentered : n == "LogCentered" ? kNormLogCentered
         : n == "InverseDistCentered" ? kNormInverseDistCentered : -1;
  };
  const size_t norm_idx = 1;
  if (!cf.rayMarchNormalization.empty() && cf.rayMarchNormalization.size() <= norm_idx)
    return bad(ADANERF_EIO, "rayMarchNormalization needs one entry per network");
  const std::string norm = cf.rayMarchNormalization.empty() ? std::string("MaxDepth") : cf.rayMarchNormalization[norm_idx];
  if (norm_code(norm) < 0)
    return bad(ADANERF_EUNSUPPORTED, "rayMarchNormalization[1] = " + norm + ": None, Centered, MaxDepth, MaxDepthCentered, LogCentered, InverseDistCentered or InverseSqrtDistCentered");
  if (!cf.rayMarchNormalizationCenter.empty() && cf.rayMarchNormalizationCenter.size() != 3)
    return bad(ADANERF_EIO, "rayMarchNormalizationCenter must hold three values (or none)");
  if (cf.depthTransform != "log" && cf.depthTransform != "linear")
    return bad(ADANERF_EUNSUPPORTED, "depthTransform must be log or linear");
  if (coarse_fine) {
    const std::string norm0 = cf.rayMarchNormalization.empty() ? std::string("MaxDepth") : cf.rayMarchNormalization[0];
    if (norm_code(norm0) < 0) return bad(ADANERF_EUNSUPPORTED, "rayMarchNormalization[0] = " + norm0 + " is not a normalisation the reference knows");
    ms->normalize0 = norm_code(norm0);
  }
  if (cf.accumulationMult == "alpha") ms->mult_mode = 1;
  else if (cf.accumulationMult == "weights") ms->mult_mode = 2;
  else ms->mult_mode = 0;
  if (coarse_fine) ms->mult_mode = 0;
  if (!pdf_mode && !coarse_fine && !cf.losses.empty()) {
    // losses[0] drives two things on the adaptive path (src/nerf_raymarch_common.py:686-690, src/features.py:503):
    // the transform applied to the oracle outputs before the threshold test (sigmoid / softmax for the BCE / CE losses)
    // and whether the kept oracle values reach compositing at all (only under NeRFWeightMultiplicationLoss).
    if (cf.losses[0] != "NeRFWeightMultiplicationLoss") ms->mult_mode = 0;   // no oracle weights in compositing
  }

  int n_max = opt->num_samples > 0 ? opt->num_samples : cf.numRaymarchSamples.back();
  float thr = opt->threshold >= 0.f ? opt->threshold : cf.adaptiveSamplingThreshold;
  if (pdf_mode || coarse_fine) thr = 1.0f;   // unused by the inverse-CDF samplers; any positive value keeps the bin-centre depth table
  if (coarse_fine) {
    // numRaymarchSamples = [Nc, Nf] (options.num_samples overrides Nf); every ray carries Nc + Nf samples through model1
    ms->n_coarse = cf.numRaymarchSamples[0];
    if (ms->n_coarse < 3 || ms->n_coarse > kMaxCoarse) return bad(ADANERF_EINVAL, "coarse/fine: numRaymarchSamples[0] must be in 3..128");
    if (n_max < 1 || ms->n_coarse + n_max > 1024) return bad(ADANERF_EINVAL, "coarse/fine: numRaymarchSamples[1] must be >= 1 and Nc + Nf <= 1024");
    n_max += ms->n_coarse;
  }
  if (thr < 0.f) return bad(ADANERF_EUNSUPPORTED, "adaptiveSamplingThreshold < 0 is unsupported on the adaptive path (as in the reference)");
  // multiDepthFeatures = [D0, D1]: D0 outputs of the sampling network, D1 depth cells of the sampler (cell_size = 1 / D1); the
  // reference needs them equal (it indexes cells by output position).  D < 128 runs on 128-wide rows padded with absent bins
  // (pack.cpp); only the adaptive sampler with a threshold takes it -- dense mode and the inverse-CDF sampler walk all 128 bins.
  ms->bins = cf.multiDepthFeatures.empty() ? kBins : cf.multiDepthFeatures.back();
  if (!cf.multiDepthFeatures.empty() && cf.multiDepthFeatures.front() != cf.multiDepthFeatures.back() && !coarse_fine)
    return bad(ADANERF_EUNSUPPORTED, "multiDepthFeatures entries differ: the sampler's cells are the sampling network's outputs");
  if (ms->bins < 1 || ms->bins > kBins) return bad(ADANERF_EUNSUPPORTED, "multiDepthFeatures must be in 1..128");
  if (ms->bins != kBins && (pdf_mode || coarse_fine || thr == 0.f))
    return bad(ADANERF_EUNSUPPORTED, "multiDepthFeatures != 128 is supported with the adaptive sampler and a threshold > 0 only");
  if (ms->bins != kBins && n_max > ms->bins) return bad(ADANERF_EINVAL, "numRaymarchSamples exceeds multiDepthFeatures");
  if (thr == 0.f && n_max != kBins) return bad(ADANERF_EUNSUPPORTED, "adaptiveSamplingThreshold == 0 (dense) requires numRaymarchSamples == 128");
  if (!coarse_fine && (n_max < 1 || n_max > kBins)) return bad(ADANERF_EINVAL, "numRaymarchSamples must be in 1..128");
  if (opt->precision < 0 || opt->precision > 2) return bad(ADANERF_EINVAL, "precision must be ADANERF_PREC_{BF16,FP16,FP32}");
  if (opt->sampling_mode < 0 || opt->sampling_mode > 3) return bad(ADANERF_EINVAL, "sampling_mode must be ADANERF_SAMPLING_{SPLIT_FP16,FP32,FP16,GUARDED}");
  if (!(opt->guard_eps <= 1.0f)) return bad(ADANERF_EINVAL, "guard_eps must be <= 1 (<= 0 selects the default)");
  if (!(opt->guard_eps_pair <= 2.0f)) return bad(ADANERF_EINVAL, "guard_eps_pair must be <= 2 (<= 0 selects the default)");
  {
    const int ap = opt->guard_audit_period;
    if (ap > 32 || (ap > 0 && (ap & (ap - 1)) != 0)) return bad(ADANERF_EINVAL, "guard_audit_period must be a power of two <= 32 (0: default, < 0: off)");
  }

  // ---- info / ray generation constants (A1: src/util/raygeneration.py:10-26, float64) ----
  const int w = opt->width, h = opt->height;
  adanerf_info& I = ms->info;
  I.abi_version = ADANERF_ABI_VERSION;
  I.width = w;
  I.height = h;
  I.compute_units = 0;
  const int world = opt->shard_world > 0 ? opt->shard_world : 1;
  const int rank = opt->shard_rank;
  if (rank < 0 || rank >= world) return bad(ADANERF_EINVAL, "shard_rank out of range");
  const int strip_rows = opt->strip_rows > 0 ? opt->strip_rows : 8;
  I.rays_local = rows_of_rank(h, strip_rows, world, rank) * w;
  I.rays_local_max = rows_of_rank(h, strip_rows, world, 0) * w;
  const int R = I.rays_local;
  if (static_cast<int64_t>(w) * h >= (1ll << 25)) return bad(ADANERF_EINVAL, "width*height must be < 2^25");
  I.batch_rays = (opt->batch_rays <= 0) ? std::max(R, 1) : std::min(opt->batch_rays, std::max(R, 1));
  // sample offsets, keys and totals are int32 on the device
  if (static_cast<int64_t>(I.batch_rays) * n_max > 0x7fffffffll)
    return bad(ADANERF_EINVAL, "batch_rays * num_samples exceeds 2^31 - 1; use a smaller batch (-bs)");
  I.n_in0 = (ms->ray_samples * 3 + 3) * (2 * ms->fp0 + 1) + 3 + 6 * ms->fd0;     // src/features.py:738-740
  if (coarse_fine) I.n_in0 = 6 + 6 * (ms->fp0 + ms->fd0);                          // src/features.py:622
  I.n_in1 = 6 + 6 * (ms->fp1 + ms->fd1);
  I.num_samples = n_max;
  I.threshold = thr;
  I.dense = thr == 0.f;
  I.use_ndc = ndc;
  I.sampler_mode = coarse_fine ? ADANERF_SAMPLER_COARSE_FINE : (pdf_mode ? ADANERF_SAMPLER_PDF : ADANERF_SAMPLER_ADAPTIVE);
  I.num_samples_coarse = ms->n_coarse;
  I.precision = opt->precision;
  I.fov = static_cast<float>(cf.fov);
  const double fov = cf.fov;
  const double focal = 0.5 * w / std::tan(0.5 * fov);   // src/datasets.py:182
  I.focal = static_cast<float>(focal);
  const double x_dist = std::tan(fov / 2) * focal;
  const double y_dist = x_dist * (static_cast<double>(h) / w);
  const double x_pp = x_dist / (w / 2.0), y_pp = y_dist / (h / 2.0);
  RayGenParams& g = ms->rg;
  g.start_x = -(x_dist - x_pp / 2);
  g.x_pp = x_pp;
  g.start_y = -(y_dist - y_pp / 2);
  g.y_pp = y_pp;
  g.focal = focal;
  g.w = w;
  g.h = h;
  g.strip_rows = strip_rows;
  g.world = world;
  g.rank = rank;
  g.use_ndc = ndc;
  double r2 = 0;
  for (int i = 0; i < 3; ++i) {
    g.center[i] = cf.viewcellCenter[i];
    I.view_cell_center[i] = cf.viewcellCenter[i];
    I.view_cell_size[i] = cf.viewcellSize[i];
    r2 += (static_cast<double>(cf.viewcellSize[i]) / 2.0) * (static_cast<double>(cf.viewcellSize[i]) / 2.0);
  }
  // radius = ||view_cell_size / 2||_2 (src/features.py:761); the reference squares the float64 norm
  const double rad = std::sqrt(r2);
  g.rad2 = static_cast<float>(rad * rad);
  I.view_cell_radius = static_cast<float>(rad);
  g.ndc_sw = static_cast<float>(-1.0 / (w / (2.0 * focal)));
  g.ndc_sh = static_cast<float>(-1.0 / (h / (2.0 * focal)));
  const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  std::memcpy(g.rot, ident, sizeof(ident));
  for (int i = 0; i < 3; ++i) g.pos[i] = g.center[i];
  I.depth_range[0] = cf.depthRange[0];
  I.depth_range[1] = cf.depthRange[1];
  I.max_depth = cf.max_depth;

  ShadeParams& sp = ms->sp;
  for (int i = 0; i < 3; ++i) sp.center[i] = cf.rayMarchNormalizationCenter.size() == 3 ? cf.rayMarchNormalizationCenter[i] : cf.viewcellCenter[i];
  sp.max_depth = cf.max_depth;
  sp.sqrt_max_depth = static_cast<float>(std::sqrt(static_cast<double>(cf.max_depth)));   // math.sqrt(max_depth)
  sp.log_max_depth_p1 = static_cast<float>(std::log(static_cast<double>(cf.max_depth) + 1.0));      // math.log(max_v + 1)
  sp.normalize = norm_code(norm);
  sp.unit_dir = ndc;
  sp.ztab = nullptr;

  ms->dm.d0 = cf.depthRange[0];
  ms->dm.d1 = cf.depthRange[1];
  ms->dm.log_transform = cf.depthTransform == "log";
  // world depths of the raySampleInput points: to_world(linspace(step/2, 1 - step/2, A)), always through the depth range
  for (int a = 0; a < ms->ray_samples; ++a) {
    const float step2 = static_cast<float>(0.5 / ms->ray_samples), end = static_cast<float>(1.0 - 0.5 / ms->ray_samples);
    // torch.linspace(start, end, A): start + a * (end - start) / (A - 1) in fp32 (symmetric form for the upper half)
    const int A = ms->ray_samples;
    const float inc = A > 1 ? (end - step2) / static_cast<float>(A - 1) : 0.f;
    const float t = (a < A / 2) ? step2 + inc * static_cast<float>(a) : end - inc * static_cast<float>(A - 1 - a);
    const float d0 = cf.depthRange[0], d1 = cf.depthRange[1];
    ms->rsi_z.push_back(ms->dm.log_transform ? powf(static_cast<float>(static_cast<double>(d1) - d0 + 1.0), t) - 1.0f + d0 : t * (d1 - d0) + d0);
  }
  // ---- depth table: world depth of each of the 128 bins (A4/A5) ----
  ms->ztab.resize(kBins);
  const float znear = cf.zNear.empty() ? 0.001f : cf.zNear.back();
  const float zfar = cf.zFar.empty() ? 1.0f : cf.zFar.back();
  const float d0 = cf.depthRange[0], d1 = cf.depthRange[1];
  for (int k = 0; k < kBins; ++k) {
    float t;
    if (thr == 0.f) {
      // src/nerf_raymarch_common.py:708-720: t = linspace(0,1,N+1)[:-1] + .5/N; z = near(1-t) + far t
      float u = static_cast<float>(k) * (1.0f / kBins) + 0.5f / kBins;
      t = znear * (1.0f - u) + zfar * u;
    } else {
      t = (static_cast<float>(k) + 0.5f) * (1.0f / static_cast<float>(ms->bins));   // (k + .5) * cell_size, cell_size = 1 / multiDepthFeatures, :726-741
    }
    float z;
    if (ndc) z = t;                                            // ...NoDepthRange: :796-851
    else if (cf.depthTransform == "log")                       // util/depth_transformations.py:37-48
      z = powf(static_cast<float>(static_cast<double>(d1) - d0 + 1.0), t) - 1.0f + d0;
    else z = t * (d1 - d0) + d0;                               // :57-58
    ms->ztab[k] = z;
  }
  // coarse/fine: LinearlySpacedZNearZFar.generate (src/nerf_raymarch_common.py:310-325): t = linspace(0,1,Nc+1)[:-1] + 0.5/Nc,
  // near (1-t) + far t with zNear[0] / zFar[0], then depth_transform.to_world over the depth range
  for (int k = 0; k < ms->n_coarse; ++k) {
    const int A = ms->n_coarse + 1;
    const float inc = 1.0f / static_cast<float>(A - 1);
    const float lin = (k < A / 2) ? inc * static_cast<float>(k) : 1.0f - inc * static_cast<float>(A - 1 - k);      // torch.linspace, fp32
    const float t = lin + static_cast<float>(0.5 / ms->n_coarse);
    const float zn = cf.zNear.empty() ? 0.001f : cf.zNear.front(), zf = cf.zFar.empty() ? 1.0f : cf.zFar.front();
    const float zw = zn * (1.0f - t) + zf * t;
    ms->ztab_coarse.push_back(cf.depthTransform == "log" ? powf(static_cast<float>(static_cast<double>(d1) - d0 + 1.0), zw) - 1.0f + d0
                                                         : zw * (d1 - d0) + d0);
  }
  // bf16 shading nets are packed scaled (pack.cpp scale_layer): every ReLU layer carries a power of two that keeps its activations <= 1 for
  // encoding inputs whose identity slots stay below kPosIdentityBound (positions) -- a scene whose sample positions can exceed it is refused here
  // rather than clamped silently.  Positions: camera inside the view cell, samples up to the far end of the depth range along a unit ray.
  if (opt->precision == ADANERF_PREC_BF16) {
    double zmax = std::max<double>(std::fabs(cf.depthRange[1]), std::fabs(cf.max_depth));
    for (float z : ms->ztab) zmax = std::max<double>(zmax, std::fabs(z));
    for (float z : ms->ztab_coarse) zmax = std::max<double>(zmax, std::fabs(z));
    double cmax = 0.0, off = 0.0;
    for (int i = 0; i < 3; ++i) {
      cmax = std::max<double>(cmax, std::fabs(cf.viewcellCenter[i]));
      off = std::max<double>(off, std::fabs(static_cast<double>(sp.center[i]) - cf.viewcellCenter[i]));
    }
    PosBound& pb = ms->pos_bound;
    pb.active = !ndc;
    pb.normalize = sp.normalize;
    pb.cmax = cmax;
    pb.zmax = zmax;
    pb.off = off;
    pb.M = std::max<double>(cf.max_depth, 1e-30);
    pb.rad = rad;
    for (int i = 0; i < 3; ++i) pb.center[i] = cf.viewcellCenter[i];
    double bound = pb.at(0.5 * rad);      // a camera inside the view cell
    if (ndc) bound = 64.0;      // NDC cube [-1, 1]^3 for rays inside the frustum (positions o' + t d', t in [0, 1])
    if (!(bound <= kPosIdentityBound)) {
      char msg[256];
      std::snprintf(msg, sizeof(msg), "sample positions of this scene can reach %.3g after rayMarchNormalization: beyond the %.0f the bf16 shading path's "
                    "layer scaling assumes (pack.hpp kPosIdentityBound) -- use precision fp16 or fp32", bound, kPosIdentityBound);
      return bad(ADANERF_EUNSUPPORTED, msg);
    }
  }
  return ADANERF_OK;
}

int ensure_net1(adanerf_ctx* c, int prec) {
  if (prec < 0 || prec > 2) return fail(c, ADANERF_EINVAL, "precision must be ADANERF_PREC_{BF16,FP16,FP32}");
  if (c->net1[prec].w.p) return ADANERF_OK;
  PackedNet pn;
  std::string err;
  const NetShape sh = shape_of(c->fp0, c->fd0, c->fp1, c->fd1, c->ray_samples, !c->coarse_fine);
  if (!pack_shading_net(c->net1_host, sh, elem_of(prec), &pn, &err)) return fail(c, ADANERF_EIO, "model1.onnx: " + err);
  return upload_net(c, pn, &c->net1[prec]);
}


// coarse/fine: model0.onnx is a NeRF net with the encoding posEncArgs[0]
int ensure_netc(adanerf_ctx* c, int prec) {
  if (prec < 0 || prec > 2) return fail(c, ADANERF_EINVAL, "precision must be ADANERF_PREC_{BF16,FP16,FP32}");
  if (c->netc[prec].w.p) return ADANERF_OK;
  PackedNet pn;
  std::string err;
  const NetShape sh = shape_of(c->fp0, c->fd0, c->fp0, c->fd0, 0, false);
  if (!pack_shading_net(c->net0_host, sh, elem_of(prec), &pn, &err)) return fail(c, ADANERF_EIO, "model0.onnx: " + err);
  return upload_net(c, pn, &c->netc[prec]);
}

int ensure_batch_buffers(adanerf_ctx* c, int n_rays, int n_max) {
  if (n_rays <= c->cap_rays && n_max <= c->cap_nmax) return ADANERF_OK;
  n_rays = std::max(n_rays, c->cap_rays);
  n_max = std::max(n_max, c->cap_nmax);
  const size_t R = static_cast<size_t>(n_rays), S = R * static_cast<size_t>(n_max);
  const size_t nblk = (R + 31) / 32;     // segment totals: per 64 rays (select_kernel) or per 32 (pair selection)
  int rc;
  if ((rc = dev_alloc(c, &c->rays, R * 8 * sizeof(float)))) return rc;
  if ((rc = dev_alloc(c, &c->ray_offsets, R * sizeof(int32_t)))) return rc;
  if ((rc = dev_alloc(c, &c->ray_counts, R * sizeof(int32_t)))) return rc;
  if (c->coarse_fine) {      // no oracle values, no selection scratch; the coarse pass has its own keys and raw outputs
    const size_t Sc = R * static_cast<size_t>(c->n_coarse);
    if ((rc = dev_alloc(c, &c->key_coarse, Sc * sizeof(uint32_t)))) return rc;
    if ((rc = dev_alloc(c, &c->raw_coarse, Sc * 4 * sizeof(float)))) return rc;
    if ((rc = dev_alloc(c, &c->sample_z, S * sizeof(float)))) return rc;
  } else {
    if ((rc = dev_alloc(c, &c->oracle, R * kBins * sizeof(float)))) return rc;
    if ((rc = dev_alloc(c, &c->selbin, S))) return rc;
    if ((rc = dev_alloc(c, &c->selw, S * sizeof(float)))) return rc;
    if (c->sampling_mode == ADANERF_SAMPLING_GUARDED) {
      if ((rc = dev_alloc(c, &c->guard_mask, nblk * sizeof(uint32_t)))) return rc;
      if ((rc = dev_alloc(c, &c->refine_list, R * sizeof(int32_t)))) return rc;
      if ((rc = dev_alloc(c, &c->guard_probe, R * 2 * sizeof(float)))) return rc;
    }
  }
  if ((rc = dev_alloc(c, &c->block_total, nblk * sizeof(int32_t)))) return rc;
  if ((rc = dev_alloc(c, &c->block_offset, nblk * sizeof(int32_t)))) return rc;
  if (!c->total.p) {
    if ((rc = dev_alloc(c, &c->total, 64))) return rc;      // [0] samples of the batch, [4] rays re-evaluated by the guarded selection,
                                                            // [8..12] its monitor (SelectOut::guard_seen): largest error seen (float bits), rays
                                                            // beyond a bound, largest pair error seen, audit mismatches, audited rays
    HIP_TRY(c, hipMemset(c->total.p, 0, 64));
  }
  if ((rc = dev_alloc(c, &c->sample_key, S * sizeof(uint32_t)))) return rc;
  if ((rc = dev_alloc(c, &c->sample_w, S * sizeof(float)))) return rc;
  if ((rc = dev_alloc(c, &c->raw, S * 4 * sizeof(float)))) return rc;
  if (c->info.sampler_mode == ADANERF_SAMPLER_PDF && (rc = dev_alloc(c, &c->sample_z, S * sizeof(float)))) return rc;
  c->cap_rays = n_rays;
  c->cap_nmax = n_max;
  return ADANERF_OK;
}

// Scratch of the selection / compaction stage only.  The stage entry point adanerf_compact must not re-allocate the
// per-batch buffers: a caller may hold pointers from adanerf_get_buffer (e.g. pass the context's own oracle buffer).
int ensure_compact_scratch(adanerf_ctx* c, int n_rays, int n_max) {
  const size_t R = static_cast<size_t>(std::max(n_rays, 1)), S = R * static_cast<size_t>(n_max);
  const size_t nblk = (R + 31) / 32;
  int rc;
  if (c->selbin.bytes < S || c->selw.bytes < S * sizeof(float) || c->block_total.bytes < nblk * sizeof(int32_t) ||
      c->block_offset.bytes < nblk * sizeof(int32_t)) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // a previous launch may still use the old scratch
    if ((rc = dev_alloc(c, &c->selbin, S))) return rc;
    if ((rc = dev_alloc(c, &c->selw, S * sizeof(float)))) return rc;
    if ((rc = dev_alloc(c, &c->block_total, nblk * sizeof(int32_t)))) return rc;
    if ((rc = dev_alloc(c, &c->block_offset, nblk * sizeof(int32_t)))) return rc;
  }
  return ADANERF_OK;
}

// ---- launches ------------------------------------------------------------------------------

template <typename K>
int occupancy_grid(adanerf_ctx* c, K kernel, int threads, int* out) {
  int per_cu = 0;
  HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0));
  if (per_cu < 1) per_cu = 1;
  *out = per_cu * c->info.compute_units;
  return ADANERF_OK;
}

int calibrate_guard(adanerf_ctx* c, int n_poses, uint32_t seed, bool install, float* max_diff, float* max_pair);
int ensure_guard_band(adanerf_ctx* c);

// a sampling net of another topology / layout on the split-precision run-time-shaped kernel (else: the fp32 one)
bool generic_split_sampling(const adanerf_ctx* c) {
  return c->generic0 && c->sampling_mode != ADANERF_SAMPLING_FP32 && c->ray_samples == 0 && c->net0_split.w.p != nullptr;
}

// sel != nullptr: the adaptive selection runs in the kernel's epilogue (k_select_pair.hip.hpp) and d_oracle may be null
int launch_sample_mlp(adanerf_ctx* c, int first_ray, int n_rays, float* d_oracle, float* d_rays, const SelectOut* sel = nullptr) {
  if (n_rays <= 0) return ADANERF_OK;
  SampleArgs a{};
  if (sel) {
    a.sel = *sel;
    a.fused_select = 1;
  }
  a.g = c->rg;
  a.net = c->net0.params;
  a.net16 = c->net0_split.params;
  a.overflow_flag = reinterpret_cast<int32_t*>(c->overflow.p);
  a.first_ray = first_ray;
  a.n_rays = n_rays;
  a.oracle_out = d_oracle;
  a.rays_out = d_rays;
  dim3 grid((n_rays + 127) / 128), block(256);
  const bool full = c->fp0 == 10 && c->fd0 == 4;
  if (c->generic0) {      // any other topology / encoding layout / raySampleInput: run-time-shaped kernels, no fused selection
    if (generic_split_sampling(c)) {
      // split-precision engine (fp32-class accuracy at 3 / 16 of the fp32-MFMA cycle count), weight tiles staged through LDS
      const uint32_t bias_cap = c->topo0.width == 64 ? gen_bias_cap<64>() : c->topo0.width == 128 ? gen_bias_cap<128>() : gen_bias_cap<256>();
      if (tune::kGenericStaged && a.net16.n_bias > bias_cap)      // cannot happen for depth <= 8: the kernel keeps the whole table in LDS
        return fail(c, ADANERF_EUNSUPPORTED, "sampling network: bias table exceeds the kernel's LDS capacity");
#define ADN_GENS(FPv, FDv, Wv) hipLaunchKernelGGL((sample_mlp16x3_gen_kernel<FPv, FDv, Wv, tune::kGenericStaged>), grid, block, 0, c->stream, a, c->gen0)
#define ADN_GENS_W(FPv, FDv)                                   \
  do {                                                         \
    if (c->topo0.width == 64) ADN_GENS(FPv, FDv, 64);          \
    else if (c->topo0.width == 128) ADN_GENS(FPv, FDv, 128);   \
    else ADN_GENS(FPv, FDv, 256);                              \
  } while (0)
      if (c->enc0 == kEnc10_4) ADN_GENS_W(10, 4);
      else if (c->enc0 == kEnc2_2) ADN_GENS_W(2, 2);
      else ADN_GENS_W(kMaxBands, kMaxBands);
#undef ADN_GENS_W
#undef ADN_GENS
      HIP_TRY(c, hipGetLastError());
      return ADANERF_OK;
    }
    if (sel) return fail(c, ADANERF_EINVAL, "fused selection is not available on the fp32 run-time-shaped sampling kernel");
    HIP_TRY(c, launch_sample_mlp_gen(a, c->gen0, c->enc0, c->topo0.width, grid.x, c->stream));      // exact fp32 MFMA (and raySampleInput)
    return ADANERF_OK;
  }
  // Guarded two-precision selection: plain fp16 for every ray, then the split engine on the rays the guard band flagged.
  // Only where the selection is fused; otherwise this mode is the split-precision engine.
  const bool guarded = c->sampling_mode == ADANERF_SAMPLING_GUARDED && sel != nullptr;
  if (guarded) {
    if (!(c->guard_eps > 0.f)) {      // first guarded frame of a context created without a band: the model's record, or measure one
      int rc = ensure_guard_band(c);
      if (rc) return rc;
    }
    a.sel.guard_mask = reinterpret_cast<uint32_t*>(c->guard_mask.p);
    a.sel.guard_eps = guard_band_of(c->transform, c->guard_eps);
    a.sel.guard_pair = guard_pair_of(c->transform, c->guard_eps, c->guard_eps_pair);
    a.sel.audit_period = c->guard_audit_period;
    a.sel.audit_phase = c->guard_audit_period > 0 ? static_cast<int32_t>(c->guard_frame & static_cast<uint32_t>(c->guard_audit_period - 1)) : 0;
    a.sel.guard_probe = reinterpret_cast<float*>(c->guard_probe.p);
    a.sel.guard_rows = reinterpret_cast<float*>(c->oracle.p);      // free on this path: the selection is fused, nobody else writes the oracle buffer
    if (c->debug_guard & 1) a.sel.guard_rows = nullptr;      // measurement knob (profiles/r04_guard_monitor_cost.md): no whole-row monitor
    a.sel.guard_seen = reinterpret_cast<uint32_t*>(c->total.p) + 8;
  }
  if (c->sampling_mode == 1) {
    HIP_TRY(c, launch_sample_mlp_f32(a, full, grid.x, c->stream));
  } else if (c->sampling_mode == 2 || guarded) {
    if (!c->net0_f16.w.p) {
      PackedNet pn;
      std::string err;
      const NetShape sh = shape_of(c->fp0, c->fd0, c->fp1, c->fd1, c->ray_samples);
      if (!pack_sampling_net(c->net0_host, sh, Elem::F16, &pn, &err)) return fail(c, ADANERF_EIO, "model0.onnx: " + err);
      int rc = upload_net(c, pn, &c->net0_f16);
      if (rc) return rc;
    }
    a.net16 = c->net0_f16.params;
    if (!c->sample16_grid) {
      int rc = full ? occupancy_grid(c, sample_mlp16_kernel<10, 4>, 512, &c->sample16_grid)
                    : occupancy_grid(c, sample_mlp16_kernel<2, 2>, 512, &c->sample16_grid);
      if (rc) return rc;
    }
    // large batches with the fused selection: two ray blocks per wave (sample_mlp16x2_kernel, bit-identical outputs); $ADANERF_DEBUG_GUARD bit 1 keeps
    // the 8-wave kernel, bit 2 takes the two-block kernel whatever the batch size (tests)
    const bool two_blocks = sel != nullptr && a.fused_select && !a.oracle_out && !(c->debug_guard & 2) && (n_rays >= kSample16x2MinRays || (c->debug_guard & 4));
    if (two_blocks) {
      if (!c->sample16x2_grid) {
        int rc = full ? occupancy_grid(c, sample_mlp16x2_kernel<10, 4>, 256, &c->sample16x2_grid)
                      : occupancy_grid(c, sample_mlp16x2_kernel<2, 2>, 256, &c->sample16x2_grid);
        if (rc) return rc;
      }
      dim3 g2(std::min<unsigned>((n_rays + 255) / 256, static_cast<unsigned>(c->sample16x2_grid))), b2(256);
      if (full) hipLaunchKernelGGL((sample_mlp16x2_kernel<10, 4>), g2, b2, 0, c->stream, a);
      else hipLaunchKernelGGL((sample_mlp16x2_kernel<2, 2>), g2, b2, 0, c->stream, a);
    } else {
    // wave tiles are dealt out evenly over the grid (sample_mlp16_kernel): every workgroup of a small batch gets a share
    dim3 g16(std::min<unsigned>((n_rays + 255) / 256, static_cast<unsigned>(c->sample16_grid))), b16(512);
    if (full) hipLaunchKernelGGL((sample_mlp16_kernel<10, 4>), g16, b16, 0, c->stream, a);
    else hipLaunchKernelGGL((sample_mlp16_kernel<2, 2>), g16, b16, 0, c->stream, a);
    }
    if (guarded) {
      // undecided rays -> ascending list (+ count at total[4]) -> split engine over the list, rows overwritten in place
      const int n_words = (n_rays + 31) / 32;
      int32_t* n_list = reinterpret_cast<int32_t*>(c->total.p) + 4;
      if (!c->sample_grid) {
        int rc = full ? occupancy_grid(c, sample_mlp16x3_kernel<10, 4>, 256, &c->sample_grid)
                      : occupancy_grid(c, sample_mlp16x3_kernel<2, 2>, 256, &c->sample_grid);
        if (rc) return rc;
      }
      const dim3 rgrid(std::min<unsigned>(grid.x, static_cast<unsigned>(c->sample_grid)));
      // ADANERF_FLAG_GUARD_AUDIT_FILL: the audit only fills the last round of the refinement pass (rgrid tiles of 128 rays per round)
      const int cap_round = (c->opt.flags & ADANERF_FLAG_GUARD_AUDIT_FILL) ? static_cast<int>(rgrid.x) * 128 : 0;
      const int cycle = a.sel.audit_period > 0 ? static_cast<int>((c->guard_frame / static_cast<uint32_t>(a.sel.audit_period)) & 0x7fffffffu) : 0;
      hipLaunchKernelGGL(refine_list_kernel, dim3((n_words + 255) / 256), dim3(256), 0, c->stream, reinterpret_cast<const uint32_t*>(c->guard_mask.p),
                         n_words, n_rays, a.sel.audit_period, a.sel.audit_phase, cap_round, cycle, reinterpret_cast<int32_t*>(c->refine_list.p), n_list);
      SampleArgs r = a;
      r.net16 = c->net0_split.params;
      r.rays_out = nullptr;              // the first pass wrote the ray records
      r.sel.guard_mask = nullptr;
      // the second pass only monitors, in RAW units: the bound on single values, and -- where pass 1 used a measured one -- on differences
      r.sel.guard_band = c->guard_eps;
      r.sel.guard_band_pair = c->transform == kOracleRaw ? a.sel.guard_pair : 0.f;
      r.sel.guard_eps = 0.f;
      r.sel.guard_pair = 0.f;
      r.sel.audit_period = 0;
      r.sel.refine_list = reinterpret_cast<const int32_t*>(c->refine_list.p);
      r.ray_list = r.sel.refine_list;
      r.n_list = n_list;
      if (full) hipLaunchKernelGGL((sample_mlp16x3_kernel<10, 4>), rgrid, block, 0, c->stream, r);
      else hipLaunchKernelGGL((sample_mlp16x3_kernel<2, 2>), rgrid, block, 0, c->stream, r);
    }
  } else {
    if (!c->sample_grid) {
      int rc = full ? occupancy_grid(c, sample_mlp16x3_kernel<10, 4>, 256, &c->sample_grid)
                    : occupancy_grid(c, sample_mlp16x3_kernel<2, 2>, 256, &c->sample_grid);
      if (rc) return rc;
    }
    grid.x = std::min<unsigned>(grid.x, static_cast<unsigned>(c->sample_grid));
    if (full) hipLaunchKernelGGL((sample_mlp16x3_kernel<10, 4>), grid, block, 0, c->stream, a);
    else hipLaunchKernelGGL((sample_mlp16x3_kernel<2, 2>), grid, block, 0, c->stream, a);
  }
  HIP_TRY(c, hipGetLastError());
  return ADANERF_OK;
}

// above this many segment totals (one per 32 or 64 rays) they are scanned by their own kernel
constexpr int kInlineScanMaxBlocks = 65536;

SelectOut select_out(adanerf_ctx* c, int n_max, float thr, int32_t* d_cnt) {
  SelectOut so{};
  so.counts = d_cnt;
  so.selbin = reinterpret_cast<uint8_t*>(c->selbin.p);
  so.selw = reinterpret_cast<float*>(c->selw.p);
  so.seg_total = reinterpret_cast<int32_t*>(c->block_total.p);
  so.n_max = n_max;
  so.thr = thr;
  so.transform = c->transform;
  return so;
}

// the selection (counts, selbin, selw, segment totals per 2^seg_shift rays) is in place: offsets + compacted arrays
int launch_expand(adanerf_ctx* c, int n_rays, int n_max, int seg_shift, int32_t* d_off, const int32_t* d_cnt, uint32_t* d_key, float* d_w,
                  int32_t* d_total) {
  const int nblk = (n_rays + (1 << seg_shift) - 1) >> seg_shift;
  const int32_t* bt = reinterpret_cast<const int32_t*>(c->block_total.p);
  int32_t* bo = reinterpret_cast<int32_t*>(c->block_offset.p);
  const dim3 egrid((n_rays + 255) / 256);
  if (nblk <= kInlineScanMaxBlocks) {
    hipLaunchKernelGGL(expand_kernel<true>, egrid, dim3(256), 0, c->stream, d_cnt, reinterpret_cast<const uint8_t*>(c->selbin.p),
                       reinterpret_cast<const float*>(c->selw.p), bo, bt, nblk, n_rays, n_max, seg_shift, d_off, d_key, d_w, d_total);
  } else {
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, c->stream, bt, nblk, bo, d_total);
    hipLaunchKernelGGL(expand_kernel<false>, egrid, dim3(256), 0, c->stream, d_cnt, reinterpret_cast<const uint8_t*>(c->selbin.p),
                       reinterpret_cast<const float*>(c->selw.p), bo, bt, nblk, n_rays, n_max, seg_shift, d_off, d_key, d_w, d_total);
  }
  HIP_TRY(c, hipGetLastError());
  return ADANERF_OK;
}

// the pair selection keeps its sorted candidate lists in registers: n_max <= kPairMaxN; ADANERF_FLAG_WAVE_SELECT forces the
// wave-per-ray select_kernel (any n_max)
bool use_pair_select(const adanerf_ctx* c, int n_max) { return n_max <= kPairMaxN && !(c->opt.flags & ADANERF_FLAG_WAVE_SELECT); }

int launch_compact(adanerf_ctx* c, const float* d_oracle, int n_rays, int n_max, float thr, int32_t* d_off, int32_t* d_cnt,
                   uint32_t* d_key, float* d_w, int32_t* d_total) {
  if (n_rays <= 0) return ADANERF_OK;
  if (thr == 0.0f) {
    const size_t n = static_cast<size_t>(n_rays) * kBins;
    dim3 grid(static_cast<unsigned>((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(dense_expand_kernel, grid, block, 0, c->stream, d_oracle, n_rays, d_off, d_cnt, d_key, d_w, d_total);
    HIP_TRY(c, hipGetLastError());
    return ADANERF_OK;
  }
  if (use_pair_select(c, n_max)) {
    hipLaunchKernelGGL(select_rows_kernel, dim3((n_rays + 127) / 128), dim3(256), 0, c->stream, d_oracle, n_rays, select_out(c, n_max, thr, d_cnt),
                       static_cast<const int32_t*>(nullptr));
    return launch_expand(c, n_rays, n_max, kPairSegShift, d_off, d_cnt, d_key, d_w, d_total);
  }
  const int nblk = (n_rays + kSelRaysPerBlock - 1) / kSelRaysPerBlock;
  hipLaunchKernelGGL(select_kernel, dim3(nblk), dim3(256), 0, c->stream, d_oracle, n_rays, n_max, thr, c->transform, d_cnt,
                     reinterpret_cast<uint8_t*>(c->selbin.p), reinterpret_cast<float*>(c->selw.p),
                     reinterpret_cast<int32_t*>(c->block_total.p));
  return launch_expand(c, n_rays, n_max, kSelSegShift, d_off, d_cnt, d_key, d_w, d_total);
}

// ---- calibration record of the guarded selection (include/adanerf_hip.h: adanerf_guard_calibration_file) ----

uint64_t fnv1a64_file(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return 0;
  uint64_t h = 0xcbf29ce484222325ull;
  unsigned char buf[1 << 16];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0)
    for (size_t i = 0; i < n; ++i) h = (h ^ buf[i]) * 0x100000001b3ull;
  std::fclose(f);
  return h ? h : 1;
}

std::string hex64(uint64_t v) {
  char b[17];
  std::snprintf(b, sizeof(b), "%016llx", static_cast<unsigned long long>(v));
  return b;
}

std::string guard_record_dir(const adanerf_ctx* c) {
  const char* env = std::getenv("ADANERF_GUARD_CACHE_DIR");
  if (env && *env) return join_path(env, hex64(c->model0_hash));
  return c->model_dir;
}

std::string guard_record_path(const adanerf_ctx* c) {
  uint32_t tb;
  const float thr = c->info.threshold;
  std::memcpy(&tb, &thr, sizeof(tb));
  char name[64];
  std::snprintf(name, sizeof(name), "guard_band.n%d.t%08x.cal", c->info.num_samples, tb);
  return join_path(guard_record_dir(c), name);
}

struct GuardRecord {
  int poses = 0;
  uint32_t seed = 0;
  float max_diff = 0.f, max_pair = 0.f;
};

// what a record must agree with to be this context's
std::string guard_record_key(const adanerf_ctx* c) {
  char b[160];
  std::snprintf(b, sizeof(b), "%s|enc %d-%d|transform %d|engine %d|n %d|thr %.9g", hex64(c->model0_hash).c_str(), c->fp0, c->fd0, c->transform,
                kGuardEngineRev, c->info.num_samples, static_cast<double>(c->info.threshold));
  return b;
}

bool read_guard_record(const adanerf_ctx* c, GuardRecord* r) {
  FILE* f = std::fopen(guard_record_path(c).c_str(), "r");
  if (!f) return false;
  char line[512];
  std::string key;
  bool have[4] = {false, false, false, false};
  while (std::fgets(line, sizeof(line), f)) {
    std::string l(line);
    const size_t eq = l.find('=');
    if (l.empty() || l[0] == '#' || eq == std::string::npos) continue;
    auto trim = [](std::string t) {
      const char* ws = " \t\r\n";
      const size_t a0 = t.find_first_not_of(ws), a1 = t.find_last_not_of(ws);
      return a0 == std::string::npos ? std::string() : t.substr(a0, a1 - a0 + 1);
    };
    const std::string k = trim(l.substr(0, eq)), v = trim(l.substr(eq + 1));
    if (k == "key") key = v;
    else if (k == "poses") { r->poses = std::atoi(v.c_str()); have[0] = true; }
    else if (k == "seed") { r->seed = static_cast<uint32_t>(std::strtoul(v.c_str(), nullptr, 10)); have[1] = true; }
    else if (k == "max_diff") { r->max_diff = std::strtof(v.c_str(), nullptr); have[2] = true; }
    else if (k == "max_pair_diff") { r->max_pair = std::strtof(v.c_str(), nullptr); have[3] = true; }
  }
  std::fclose(f);
  return key == guard_record_key(c) && have[0] && have[1] && have[2] && have[3] && r->poses >= 1 && r->max_diff > 0.f &&
         r->max_diff < INFINITY && r->max_pair >= 0.f && r->max_pair < INFINITY;
}

// best effort: a read-only model directory simply keeps being calibrated at start-up (or use ADANERF_GUARD_CACHE_DIR)
void write_guard_record(const adanerf_ctx* c, const GuardRecord& r) {
  const std::string dir = guard_record_dir(c), path = guard_record_path(c), tmp = path + ".tmp";
  if (std::getenv("ADANERF_GUARD_CACHE_DIR")) {
    (void)mkdir(std::getenv("ADANERF_GUARD_CACHE_DIR"), 0777);
    (void)mkdir(dir.c_str(), 0777);
  }
  FILE* f = std::fopen(tmp.c_str(), "w");
  if (!f) return;
  std::fprintf(f,
               "# libadanerf_hip: measured error bounds of the plain-fp16 sampling pass against the split-precision engine\n"
               "# (ADANERF_SAMPLING_GUARDED).  Delete this file to have them measured again.\n"
               "key = %s\nposes = %d\nseed = %u\nmax_diff = %.9g\nmax_pair_diff = %.9g\n",
               guard_record_key(c).c_str(), r.poses, r.seed, static_cast<double>(r.max_diff), static_cast<double>(r.max_pair));
  const bool ok = std::fclose(f) == 0;
  if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) (void)std::remove(tmp.c_str());
}

void install_guard_band(adanerf_ctx* c, float max_diff, float max_pair, int poses, int source) {
  c->guard_eps = std::max(ADANERF_GUARD_CALIB_MARGIN * max_diff, ADANERF_GUARD_EPS_MIN);
  // a measured pair bound applies to untransformed outputs only (guard_pair_of); never below the floor, never above what the
  // single-value bound implies
  c->guard_eps_pair = (c->transform == kOracleRaw && max_pair > 0.f)
                          ? std::min(std::max(ADANERF_GUARD_CALIB_MARGIN * max_pair, ADANERF_GUARD_EPS_MIN), 2.0f * c->guard_eps)
                          : 2.0f * c->guard_eps;
  c->info.guard_eps = c->guard_eps;
  c->info.guard_eps_pair = c->guard_eps_pair;
  c->info.guard_calib_source = source;
  c->info.guard_calib_poses = poses;
}

// Largest |plain-fp16 - split-precision| raw output of the sampling network over n_poses x 64 x 64 calibration rays, and the largest
// error of a (kept - candidate) difference under the single-value bound that gives (include/adanerf_hip.h: adanerf_calibrate_guard).
// Two sweeps over the same seeded poses: the second statistic needs the first.  Temporarily replaces the ray generator's image and camera.
int calibrate_guard(adanerf_ctx* c, int n_poses, uint32_t seed, bool install, float* max_diff, float* max_pair) {
  if (c->coarse_fine || c->generic0) return fail(c, ADANERF_EUNSUPPORTED, "guard calibration needs an 8 x 256 sampling network");
  if (n_poses < 1 || n_poses > 4096) return fail(c, ADANERF_EINVAL, "n_poses must be in 1..4096");
  constexpr int CW = 64, CH = 64, CR = CW * CH;
  const RayGenParams saved = c->rg;
  const int saved_mode = c->sampling_mode;
  RayGenParams g = saved;
  // same field of view on a 64 x 64 image (setup_model's arithmetic): x_dist = tan(fov / 2) focal, y_dist = x_dist h / w
  const double x_dist = -saved.start_x + saved.x_pp / 2, y_dist = -saved.start_y + saved.y_pp / 2;
  g.w = CW;
  g.h = CH;
  g.x_pp = x_dist / (CW / 2.0);
  g.y_pp = y_dist / (CH / 2.0);
  g.start_x = -(x_dist - g.x_pp / 2);
  g.start_y = -(y_dist - g.y_pp / 2);
  g.strip_rows = CH;
  g.world = 1;
  g.rank = 0;
  DevBuf a_buf, b_buf, acc;
  int rc = ADANERF_OK;
  auto done = [&](int code) {
    (void)hipStreamSynchronize(c->stream);
    dev_free(&a_buf);
    dev_free(&b_buf);
    dev_free(&acc);
    c->rg = saved;
    c->sampling_mode = saved_mode;
    return code;
  };
  if ((rc = dev_alloc(c, &a_buf, static_cast<size_t>(CR) * kBins * sizeof(float)))) return done(rc);
  if ((rc = dev_alloc(c, &b_buf, static_cast<size_t>(CR) * kBins * sizeof(float)))) return done(rc);
  if ((rc = dev_alloc(c, &acc, 64))) return done(rc);
  if (hipMemsetAsync(acc.p, 0, 64, c->stream) != hipSuccess) return done(fail(c, ADANERF_EDEVICE, "hipMemsetAsync failed"));
  const bool want_pairs = c->transform == kOracleRaw && c->info.threshold > 0.f && c->info.num_samples <= kPairMaxN;
  float d = 0.f, dp = 0.f;
  bool nonfinite = false;
  for (int sweep = 0; sweep < (want_pairs ? 2 : 1) && !nonfinite; ++sweep) {
    const float two_eps = sweep == 0 ? 0.f : 2.0f * std::max(ADANERF_GUARD_CALIB_MARGIN * d, ADANERF_GUARD_EPS_MIN);
    uint64_t st = 0x9E3779B97F4A7C15ull ^ (static_cast<uint64_t>(seed) << 17);
    auto rnd = [&]() {      // xorshift64*, uniform in [0, 1)
      st ^= st >> 12;
      st ^= st << 25;
      st ^= st >> 27;
      return static_cast<double>((st * 0x2545F4914F6CDD1Dull) >> 11) * (1.0 / 9007199254740992.0);
    };
    for (int k = 0; k < n_poses; ++k) {
      for (int i = 0; i < 3; ++i) g.pos[i] = g.center[i] + static_cast<float>((rnd() - 0.5) * 0.9 * c->info.view_cell_size[i]);
      // orientation: yaw about z, then pitch (the viewer's camera: z up, looking along -z of the camera frame)
      const double yaw = 2.0 * M_PI * rnd(), pitch = (rnd() - 0.5) * (c->info.use_ndc ? 0.2 : 1.4);
      const double cy = std::cos(yaw), sy = std::sin(yaw), cp = std::cos(pitch), sp = std::sin(pitch);
      double fwd[3], right[3], up[3];
      if (c->info.use_ndc) {      // forward-facing scenes look down -z
        fwd[0] = sp * cy; fwd[1] = sp * sy; fwd[2] = -cp;
        right[0] = 1; right[1] = 0; right[2] = 0;
      } else {
        fwd[0] = cp * cy; fwd[1] = cp * sy; fwd[2] = sp;
        right[0] = sy; right[1] = -cy; right[2] = 0;
      }
      // re-orthogonalise: right = normalize(right - (right . fwd) fwd), up = right x fwd
      const double rf = right[0] * fwd[0] + right[1] * fwd[1] + right[2] * fwd[2];
      double n = 0;
      for (int i = 0; i < 3; ++i) { right[i] -= rf * fwd[i]; n += right[i] * right[i]; }
      n = std::sqrt(n);
      for (int i = 0; i < 3; ++i) right[i] /= n;
      up[0] = right[1] * fwd[2] - right[2] * fwd[1];
      up[1] = right[2] * fwd[0] - right[0] * fwd[2];
      up[2] = right[0] * fwd[1] - right[1] * fwd[0];
      for (int i = 0; i < 3; ++i) {      // columns of c2w: camera x = right, y = up, -z = forward
        g.rot[3 * i + 0] = static_cast<float>(right[i]);
        g.rot[3 * i + 1] = static_cast<float>(up[i]);
        g.rot[3 * i + 2] = static_cast<float>(-fwd[i]);
      }
      c->rg = g;
      c->sampling_mode = ADANERF_SAMPLING_SPLIT_FP16;
      if ((rc = launch_sample_mlp(c, 0, CR, reinterpret_cast<float*>(a_buf.p), nullptr))) return done(rc);
      c->sampling_mode = ADANERF_SAMPLING_FP16;
      if ((rc = launch_sample_mlp(c, 0, CR, reinterpret_cast<float*>(b_buf.p), nullptr))) return done(rc);
      hipLaunchKernelGGL(guard_stats_kernel, dim3((CR + 3) / 4), dim3(256), 0, c->stream, reinterpret_cast<const float*>(b_buf.p),
                         reinterpret_cast<const float*>(a_buf.p), CR, c->info.num_samples, c->info.threshold, two_eps, reinterpret_cast<uint32_t*>(acc.p));
    }
    uint32_t res[3] = {0, 0, 0};
    if (hipMemcpyAsync(res, acc.p, sizeof(res), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
      return done(fail(c, ADANERF_EDEVICE, "guard calibration: read-back failed"));
    std::memcpy(&d, &res[0], sizeof(d));
    std::memcpy(&dp, &res[2], sizeof(dp));
    nonfinite = res[1] != 0;
  }
  if (max_diff) *max_diff = nonfinite ? INFINITY : d;
  if (max_pair) *max_pair = (nonfinite || !want_pairs) ? 0.f : dp;
  if (install) {
    if (nonfinite) {
      c->guard_eps = ADANERF_GUARD_EPS_DEFAULT;
      c->guard_eps_pair = 2.0f * c->guard_eps;
      c->info.guard_eps = c->guard_eps;
      c->info.guard_eps_pair = c->guard_eps_pair;
      c->info.guard_calib_source = ADANERF_GUARD_FROM_CALIBRATION;
      c->info.guard_calib_poses = n_poses;
    } else {
      install_guard_band(c, d, want_pairs ? dp : 0.f, n_poses, ADANERF_GUARD_FROM_CALIBRATION);
      GuardRecord old;      // a record from more poses than this measurement stays
      if (!(c->opt.flags & ADANERF_FLAG_NO_GUARD_CACHE) && !(read_guard_record(c, &old) && old.poses > n_poses)) {
        GuardRecord r;
        r.poses = n_poses;
        r.seed = seed;
        r.max_diff = d;
        r.max_pair = want_pairs ? dp : 0.f;
        write_guard_record(c, r);
      }
    }
  }
  return done(ADANERF_OK);
}

// first guarded frame of a context created without a band: the model's calibration record if it is current, else a measurement
int ensure_guard_band(adanerf_ctx* c) {
  GuardRecord r;
  if (!(c->opt.flags & ADANERF_FLAG_NO_GUARD_CACHE) && read_guard_record(c, &r) && r.poses >= ADANERF_GUARD_CALIB_POSES) {
    install_guard_band(c, r.max_diff, r.max_pair, r.poses, ADANERF_GUARD_FROM_RECORD);
    return ADANERF_OK;
  }
  float d = 0.f, dp = 0.f;
  return calibrate_guard(c, ADANERF_GUARD_CALIB_POSES, 1u, true, &d, &dp);
}

constexpr int kShadeWaves = 8;   // one 8-wave workgroup per CU (two independent 4-wave workgroups measured 4.2-5.7 ms vs 3.8)

int launch_shade_mlp(adanerf_ctx* c, const float* d_rays, const uint32_t* d_key, const int32_t* d_total, int max_samples, int prec,
                     float* d_raw, const float* d_z = nullptr, bool coarse = false) {
  if (max_samples <= 0) return ADANERF_OK;
  // coarse: the first network of the vanilla-NeRF mode (model0.onnx as a NeRF net, uniform depth table)
  const bool generic = coarse ? c->genericc : c->generic1;
  const NetTopology& topo = coarse ? c->topoc : c->topo1;
  const GenericTopo& gen = coarse ? c->genc : c->gen1;
  int& gen_grid = coarse ? c->shade_gen_grid_c : c->shade_gen_grid;
  int rc = coarse ? ensure_netc(c, prec) : ensure_net1(c, prec);
  if (rc) return rc;
  ShadeArgs a{};
  a.sp = coarse ? c->spc : c->sp;
  a.net = coarse ? c->netc[prec].params : c->net1[prec].params;
  a.rays = d_rays;
  a.sample_key = d_key;
  a.sample_z = d_z;
  a.total = d_total;
  a.max_samples = max_samples;
  a.raw_out = d_raw;
  if (generic && prec != ADANERF_PREC_FP32) {
    // any topology / encoding layout on the 16-bit MFMA pipe (k_generic16.hip.hpp)
    const int enc = coarse ? enc_layout(c->fp0, c->fd0, false) : c->enc1;
    // staged: one copy of every weight tile per workgroup through LDS, NB 32-sample blocks per wave; else fragments straight from L2.
    // Persistent grid = the workgroups the chosen instantiation really keeps resident (registers and LDS differ per width / layout).
    const bool st = tune::kGenericStaged;
    const bool enc104 = enc == kEnc10_4;      // else the catch-all 16-band layout
    const int nb_wg = !st ? 1 : topo.width == 64 ? gen_blocks<64>() : topo.width == 128 ? (enc104 ? gen_blocks<128, 10>() : gen_blocks<128, kMaxBands>())
                                                                                         : gen_blocks<256>();
    const int per_wg = 128 * nb_wg;
    const int tiles = (max_samples + per_wg - 1) / per_wg;
    const uint32_t bias_cap = topo.width == 64 ? gen_bias_cap<64>() : topo.width == 128 ? gen_bias_cap<128>() : gen_bias_cap<256>();
    if (st && a.net.n_bias > bias_cap)      // cannot happen for depth <= 8 (pack.hpp kMaxDepth): the kernel keeps the whole table in LDS
      return fail(c, ADANERF_EUNSUPPORTED, "shading network: bias table exceeds the kernel's LDS capacity");
    int& grid16 = c->shade_gen16_grid[coarse ? 1 : 0][prec == ADANERF_PREC_BF16 ? 0 : 1];
#define ADN_GEN16_GO(KERNEL)                                                                                             \
  do {                                                                                                                   \
    if (!grid16) {                                                                                                       \
      int per_cu = 0;                                                                                                    \
      HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, KERNEL, 256, 0));                                 \
      grid16 = (per_cu < 1 ? 1 : per_cu) * c->info.compute_units;                                                        \
    }                                                                                                                    \
    hipLaunchKernelGGL(KERNEL, dim3(std::min(tiles, grid16)), dim3(256), 0, c->stream, a, gen);                          \
  } while (0)
#define ADN_GEN16(ET, FPv, FDv, Wv)                                                                                      \
  do {                                                                                                                   \
    if constexpr (tune::kGenericStaged)                                                                                  \
      ADN_GEN16_GO((shade_mlp16_gen_staged_kernel<ET, FPv, FDv, Wv, gen_blocks<Wv, FPv>(), gen_occupancy<Wv, FPv>()>));       \
    else ADN_GEN16_GO((shade_mlp16_gen_kernel<ET, FPv, FDv, Wv>));                                                       \
  } while (0)
#define ADN_GEN16_W(ET, FPv, FDv)                                  \
  do {                                                             \
    if (topo.width == 64) ADN_GEN16(ET, FPv, FDv, 64);             \
    else if (topo.width == 128) ADN_GEN16(ET, FPv, FDv, 128);      \
    else ADN_GEN16(ET, FPv, FDv, 256);                             \
  } while (0)
    if (prec == ADANERF_PREC_BF16) {
      if (enc == kEnc10_4) ADN_GEN16_W(Bf16, 10, 4);
      else ADN_GEN16_W(Bf16, kMaxBands, kMaxBands);
    } else {
      if (enc == kEnc10_4) ADN_GEN16_W(Fp16, 10, 4);
      else ADN_GEN16_W(Fp16, kMaxBands, kMaxBands);
    }
#undef ADN_GEN16_W
#undef ADN_GEN16
#undef ADN_GEN16_GO
    HIP_TRY(c, hipGetLastError());
  } else if (generic) {
    const int enc = coarse ? enc_layout(c->fp0, c->fd0, false) : c->enc1;
    if (!gen_grid) HIP_TRY(c, shade_mlp_gen_grid(c->info.compute_units, enc, topo.width, &gen_grid));
    const int tiles = (max_samples + 127) / 128;
    HIP_TRY(c, launch_shade_mlp_gen(a, gen, enc, topo.width, std::min(tiles, gen_grid), c->stream));
  } else if (prec == ADANERF_PREC_FP32) {
    if (!c->shade_grid[2]) HIP_TRY(c, shade_mlp_f32_grid(c->info.compute_units, &c->shade_grid[2]));
    const int tiles = (max_samples + 127) / 128;
    HIP_TRY(c, launch_shade_mlp_f32(a, std::min(tiles, c->shade_grid[2]), c->stream));
  } else {
    const int tile = kShadeWaves * 32;
    const int tiles = (max_samples + tile - 1) / tile;
    if (tune::kShadeBlocks == 2) {      // two sample blocks per wave, 4 waves x 64 samples (same 256-sample tile)
      const int cu = c->info.compute_units;
      if (prec == ADANERF_PREC_BF16) hipLaunchKernelGGL((shade_mlp16x2_kernel<Bf16, 10, 4>), dim3(std::min(tiles, cu)), dim3(256), 0, c->stream, a);
      else hipLaunchKernelGGL((shade_mlp16x2_kernel<Fp16, 10, 4>), dim3(std::min(tiles, cu)), dim3(256), 0, c->stream, a);
    } else if (prec == ADANERF_PREC_BF16) {
      if (!c->shade_grid[0] && (rc = occupancy_grid(c, shade_mlp16_kernel<Bf16, 10, 4, kShadeWaves>, kShadeWaves * 64, &c->shade_grid[0]))) return rc;
      hipLaunchKernelGGL((shade_mlp16_kernel<Bf16, 10, 4, kShadeWaves>), dim3(std::min(tiles, c->shade_grid[0])), dim3(kShadeWaves * 64), 0,
                         c->stream, a);
    } else {
      if (!c->shade_grid[1] && (rc = occupancy_grid(c, shade_mlp16_kernel<Fp16, 10, 4, kShadeWaves>, kShadeWaves * 64, &c->shade_grid[1]))) return rc;
      hipLaunchKernelGGL((shade_mlp16_kernel<Fp16, 10, 4, kShadeWaves>), dim3(std::min(tiles, c->shade_grid[1])), dim3(kShadeWaves * 64), 0,
                         c->stream, a);
    }
  }
  HIP_TRY(c, hipGetLastError());
  return ADANERF_OK;
}

int launch_sample_pdf(adanerf_ctx* c, const float* d_oracle, int n_rays, int n, int32_t* d_off, int32_t* d_cnt, uint32_t* d_key, float* d_w,
                      float* d_z, int32_t* d_total) {
  if (n_rays <= 0) return ADANERF_OK;
  const int grid = std::min((n_rays + 3) / 4, c->info.compute_units * 8);
  hipLaunchKernelGGL(pdf_sample_kernel, dim3(grid), dim3(256), 0, c->stream, d_oracle, n_rays, n, c->transform, c->dm, d_off, d_cnt, d_key, d_w, d_z, d_total);
  HIP_TRY(c, hipGetLastError());
  return ADANERF_OK;
}

int launch_camera_rays(adanerf_ctx* c, int first_ray, int n_rays, float* d_rays) {
  if (n_rays <= 0) return ADANERF_OK;
  hipLaunchKernelGGL(camera_rays_kernel, dim3((n_rays + 255) / 256), dim3(256), 0, c->stream, c->rg, first_ray, n_rays, d_rays);
  HIP_TRY(c, hipGetLastError());
  return ADANERF_OK;
}

int launch_sample_uniform(adanerf_ctx* c, int n_rays, int n, int32_t* d_off, int32_t* d_cnt, uint32_t* d_key, int32_t* d_total) {
  if (n_rays <= 0) return ADANERF_OK;
  const int64_t s = static_cast<int64_t>(n_rays) * n;
  hipLaunchKernelGGL(uniform_sample_kernel, dim3(static_cast<unsigned>((s + 255) / 256)), dim3(256), 0, c->stream, n_rays, n, d_off, d_cnt, d_key, d_total);
  HIP_TRY(c, hipGetLastError());
  return ADANERF_OK;
}

int launch_sample_fine(adanerf_ctx* c, const float* d_raw_coarse, const float* d_rays, int n_rays, int32_t* d_off, int32_t* d_cnt,
                       uint32_t* d_key, float* d_z, int32_t* d_total) {
  if (n_rays <= 0) return ADANERF_OK;
  hipLaunchKernelGGL(fine_sample_kernel, dim3((n_rays + kFineRaysPerBlock - 1) / kFineRaysPerBlock), dim3(kFineRaysPerBlock), 0, c->stream,
                     reinterpret_cast<const float4*>(d_raw_coarse), reinterpret_cast<const float*>(c->ztab_coarse.p), d_rays, n_rays, c->n_coarse,
                     c->info.num_samples - c->n_coarse, d_off, d_cnt, d_key, d_z, d_total);
  HIP_TRY(c, hipGetLastError());
  return ADANERF_OK;
}

int launch_composite_classic(adanerf_ctx* c, const float* d_raw, const float* d_z, const float* d_rays, int n_rays, int n, float* d_rgb,
                             void* d_rgba8, float* d_depth = nullptr, float* d_acc = nullptr) {
  if (n_rays <= 0) return ADANERF_OK;
  if (n > 32)      // long rays: one wave per ray, coalesced
    hipLaunchKernelGGL(composite_classic_wave_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, c->stream, reinterpret_cast<const float4*>(d_raw),
                       d_z, d_rays, n_rays, n, d_rgb, reinterpret_cast<uchar4*>(d_rgba8), d_depth, d_acc);
  else
    hipLaunchKernelGGL(composite_classic_kernel, dim3((n_rays + 255) / 256), dim3(256), 0, c->stream, reinterpret_cast<const float4*>(d_raw),
                       d_z, d_rays, n_rays, n, d_rgb, reinterpret_cast<uchar4*>(d_rgba8), d_depth, d_acc);
  HIP_TRY(c, hipGetLastError());
  return ADANERF_OK;
}

int launch_composite(adanerf_ctx* c, const float* d_raw, const float* d_w, const int32_t* d_off, const int32_t* d_cnt, int n_rays,
                     float* d_rgb, void* d_rgba8, const uint32_t* d_key = nullptr, float* d_depth = nullptr, float* d_acc = nullptr) {
  if (n_rays <= 0) return ADANERF_OK;
  AuxOut aux{};
  if ((d_key || c->info.dense) && (d_depth || d_acc)) {
    aux.depth = d_depth;
    aux.acc = d_acc;
    aux.sample_key = d_key;      // null in dense mode: the bin of sample i is i & 127
    aux.ztab = reinterpret_cast<const float*>(c->ztab.p);
  }
  if (c->info.num_samples > 32)   // long rays (dense mode): one wave per ray, coalesced
    hipLaunchKernelGGL(composite_wave_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, c->stream, reinterpret_cast<const float4*>(d_raw), d_w,
                       d_off, d_cnt, n_rays, c->mult_mode, d_rgb, reinterpret_cast<uchar4*>(d_rgba8), aux);
  else {
    // samples of RB consecutive rays staged in LDS (20 B each), RB chosen so that RB * N * 20 B <= 48 KB
    const int N = c->info.num_samples;
    auto run = [&](auto rb_tag) {
      constexpr int RB = decltype(rb_tag)::value;
      const int cap = RB * N;
      hipLaunchKernelGGL(composite_kernel<RB>, dim3((n_rays + RB - 1) / RB), dim3(RB), static_cast<size_t>(cap) * 20, c->stream,
                         reinterpret_cast<const float4*>(d_raw), d_w, d_off, d_cnt, n_rays, c->mult_mode, cap, d_rgb,
                         reinterpret_cast<uchar4*>(d_rgba8), aux);
    };
    if (N <= 9) run(std::integral_constant<int, 256>{});
    else if (N <= 19) run(std::integral_constant<int, 128>{});
    else run(std::integral_constant<int, 64>{});
  }
  HIP_TRY(c, hipGetLastError());
  return ADANERF_OK;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================

extern "C" {

const char* adanerf_last_error(const adanerf_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

static int create_on(adanerf_ctx* c, const char* model_dir, const adanerf_options* opt, adanerf_ctx** out);

int adanerf_create(const char* model_dir, const adanerf_options* opt, adanerf_ctx** out) {
  if (!out) return fail(nullptr, ADANERF_EINVAL, "out == NULL");
  *out = nullptr;
  if (!model_dir || !opt) return fail(nullptr, ADANERF_EINVAL, "model_dir/opt == NULL");
  adanerf_ctx* c = nullptr;
  try {
    c = new adanerf_ctx();
    return create_on(c, model_dir, opt, out);
  } catch (const std::exception& e) {      // see setup_model: the loader's exceptions end here, with the context released
    *out = nullptr;
    adanerf_destroy(c);
    return fail(nullptr, ADANERF_EIO, std::string("adanerf_create: ") + e.what());
  }
}

static int create_on(adanerf_ctx* c, const char* model_dir, const adanerf_options* opt, adanerf_ctx** out) {
  auto bail = [&](int code, const std::string& msg) {
    g_create_error = msg;
    adanerf_destroy(c);
    return code;
  };
  std::string err;
  ModelSetup ms;
  int rc = setup_model(model_dir, opt, &ms, &err);
  if (rc) return bail(rc, err);
  c->opt = *opt;
  c->cfg = ms.cfg;
  c->info = ms.info;
  c->rg = ms.rg;
  c->sp = ms.sp;
  c->mult_mode = ms.mult_mode;
  c->transform = ms.transform;
  c->pos_bound = ms.pos_bound;
  c->dm = ms.dm;
  c->fp0 = ms.fp0;
  c->fd0 = ms.fd0;
  c->fp1 = ms.fp1;
  c->fd1 = ms.fd1;
  c->coarse_fine = ms.coarse_fine;
  c->n_coarse = ms.n_coarse;

  // ---- weights: parse + pack on the host before touching the device ----
  TensorMap& n0 = c->net0_host;
  if (!read_onnx_initializers(join_path(model_dir, "model0.onnx"), &n0, &err)) return bail(ADANERF_EIO, err);
  if (!read_onnx_initializers(join_path(model_dir, "model1.onnx"), &c->net1_host, &err)) return bail(ADANERF_EIO, err);
  c->ray_samples = ms.ray_samples;
  const NetShape sh = shape_of(c->fp0, c->fd0, c->fp1, c->fd1, c->ray_samples, !c->coarse_fine);
  c->enc0 = enc_layout(c->fp0, c->fd0, !c->coarse_fine);
  c->enc1 = enc_layout(c->fp1, c->fd1, false);
  PackedNet p0, p1;
  PackedNet p0s;
  if (c->coarse_fine) {       // model0.onnx is a NeRF net here (src/models.py:199-277); probe its topology with the fp32 packing
    const NetShape shc = shape_of(c->fp0, c->fd0, c->fp0, c->fd0, 0, false);
    if (!pack_shading_net(n0, shc, Elem::F32, &p0, &err)) return bail(ADANERF_EIO, "model0.onnx: " + err);
    c->topoc = p0.topo;
    c->genericc = !p0.topo.is_default(true) || c->enc0 == kEncMax;
    if (opt->precision != ADANERF_PREC_FP32 && !pack_shading_net(n0, shc, elem_of(opt->precision), &p0, &err))
      return bail(ADANERF_EIO, "model0.onnx: " + err);
  } else {
    if (!pack_sampling_net(n0, sh, Elem::F32, &p0, &err)) return bail(ADANERF_EIO, "model0.onnx: " + err);
    if (p0.topo.bins != ms.bins)
      return bail(ADANERF_EIO, "model0.onnx has " + std::to_string(p0.topo.bins) + " outputs, config.ini says multiDepthFeatures = " + std::to_string(ms.bins));
    c->topo0 = p0.topo;
    c->generic0 = !p0.topo.is_default(false) || c->enc0 == kEncMax;
    // the split-precision packing: the ring-streamed kernel's for the 8 x 256 net, the run-time-shaped kernel's otherwise (not with raySampleInput)
    if ((!c->generic0 || c->ray_samples == 0) && !pack_sampling_net(n0, sh, Elem::F16_SPLIT, &p0s, &err)) return bail(ADANERF_EIO, "model0.onnx: " + err);
  }
  c->sampling_mode = opt->sampling_mode;
  c->model_dir = model_dir;
  if (const char* dbg = std::getenv("ADANERF_DEBUG_GUARD")) c->debug_guard = std::atoi(dbg);
  c->guard_eps = opt->guard_eps > 0.f ? opt->guard_eps : 0.f;      // 0: the model's calibration record, or calibrated before the first guarded frame
  c->guard_eps_pair = (c->guard_eps > 0.f && opt->guard_eps_pair > 0.f) ? std::min(opt->guard_eps_pair, 2.0f * c->guard_eps) : 2.0f * c->guard_eps;
  c->guard_audit_period = opt->guard_audit_period == 0 ? ADANERF_GUARD_AUDIT_PERIOD : std::max(opt->guard_audit_period, 0);
  c->info.guard_eps = c->guard_eps;
  c->info.guard_eps_pair = c->guard_eps_pair;
  c->info.guard_audit_period = c->sampling_mode == ADANERF_SAMPLING_GUARDED ? c->guard_audit_period : 0;
  c->info.guard_calib_source = c->guard_eps > 0.f ? ADANERF_GUARD_FROM_OPTIONS : ADANERF_GUARD_FROM_NONE;
  if (c->sampling_mode == ADANERF_SAMPLING_GUARDED) c->model0_hash = fnv1a64_file(join_path(model_dir, "model0.onnx"));
  {   // topology of the shading net first (fp32 packing accepts every supported topology)
    PackedNet probe;
    if (!pack_shading_net(c->net1_host, sh, Elem::F32, &probe, &err)) return bail(ADANERF_EIO, "model1.onnx: " + err);
    c->topo1 = probe.topo;
    c->generic1 = !probe.topo.is_default(true) || c->enc1 == kEncMax;
    if (opt->precision == ADANERF_PREC_FP32) p1 = std::move(probe);
    else if (!pack_shading_net(c->net1_host, sh, elem_of(opt->precision), &p1, &err)) return bail(ADANERF_EIO, "model1.onnx: " + err);
  }
  // the run-time-shaped kernels are instantiated for these widths (k_generic_f32.hip.hpp); the packer pads any width <= 256 up to one of them
  auto width_ok = [](int w) { return w == 64 || w == 128 || w == 256; };
  if (c->genericc && !width_ok(c->topoc.width)) return bail(ADANERF_EUNSUPPORTED, "model0.onnx: layer width " + std::to_string(c->topoc.width) + " (64, 128 or 256 supported)");
  if (c->generic0 && !width_ok(c->topo0.width)) return bail(ADANERF_EUNSUPPORTED, "model0.onnx: layer width " + std::to_string(c->topo0.width) + " (64, 128 or 256 supported)");
  if (c->generic1 && !width_ok(c->topo1.width)) return bail(ADANERF_EUNSUPPORTED, "model1.onnx: layer width " + std::to_string(c->topo1.width) + " (64, 128 or 256 supported)");

  // ---- device ----
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return bail(ADANERF_EDEVICE, "no HIP device available: libadanerf_hip has no CPU fallback");
  if (hipSetDevice(opt->device_id) != hipSuccess) return bail(ADANERF_EDEVICE, "hipSetDevice failed");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, opt->device_id) != hipSuccess) return bail(ADANERF_EDEVICE, "hipGetDeviceProperties failed");
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    return bail(ADANERF_EDEVICE, std::string("device is ") + prop.gcnArchName + "; this library is built for gfx950 (MI355X) only");
  c->info.compute_units = prop.multiProcessorCount;
  c->device = opt->device_id;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) return bail(ADANERF_EDEVICE, "hipStreamCreate failed");
  c->stream = c->own_stream;

  rc = dev_alloc(c, &c->ztab, kBins * sizeof(float));
  if (rc) return bail(rc, c->err);
  if (hipMemcpy(c->ztab.p, ms.ztab.data(), kBins * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
    return bail(ADANERF_EDEVICE, "ztab upload failed");
  c->sp.ztab = reinterpret_cast<const float*>(c->ztab.p);
  if (c->coarse_fine) {
    if ((rc = upload_net(c, p0, &c->netc[opt->precision]))) return bail(rc, c->err);
    if ((rc = dev_alloc(c, &c->ztab_coarse, kMaxCoarse * sizeof(float)))) return bail(rc, c->err);
    if (hipMemcpy(c->ztab_coarse.p, ms.ztab_coarse.data(), ms.ztab_coarse.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
      return bail(ADANERF_EDEVICE, "coarse depth table upload failed");
    c->spc = c->sp;
    c->spc.normalize = ms.normalize0;
    c->sp.unit_dir = 0;      // the fine pass encodes rays_d as RayMarchFromPoses handed it over: un-normalised under NDC (src/features.py:654-668)
    c->spc.ztab = reinterpret_cast<const float*>(c->ztab_coarse.p);
    c->genc = GenericTopo{c->topoc.depth, c->topoc.cat_mask, 0, 0, nullptr, 0.f};
  } else {
    if ((rc = upload_net(c, p0, &c->net0))) return bail(rc, c->err);
    if ((!c->generic0 || c->ray_samples == 0) && (rc = upload_net(c, p0s, &c->net0_split))) return bail(rc, c->err);
  }
  if (c->ray_samples > 0) {
    if ((rc = dev_alloc(c, &c->rsi_z, ms.rsi_z.size() * sizeof(float)))) return bail(rc, c->err);
    if (hipMemcpy(c->rsi_z.p, ms.rsi_z.data(), ms.rsi_z.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
      return bail(ADANERF_EDEVICE, "raySampleInput depth table upload failed");
  }
  c->gen0 = GenericTopo{c->topo0.depth, 0, c->ray_samples, p0.rsi_w_off, reinterpret_cast<const float*>(c->rsi_z.p), c->cfg.depthRange[1]};
  c->gen1 = GenericTopo{c->topo1.depth, c->topo1.cat_mask, 0, 0, nullptr, 0.f};
  if ((rc = dev_alloc(c, &c->overflow, 64))) return bail(rc, c->err);
  if (hipMemset(c->overflow.p, 0, 64) != hipSuccess) return bail(ADANERF_EDEVICE, "hipMemset failed");
  if ((rc = upload_net(c, p1, &c->net1[opt->precision]))) return bail(rc, c->err);
  if ((rc = ensure_batch_buffers(c, c->info.batch_rays, c->info.num_samples))) return bail(rc, c->err);
  *out = c;
  return ADANERF_OK;
}

int adanerf_host_parse_model(const char* model_dir, const adanerf_options* opt, adanerf_info* info) {
  if (!model_dir || !opt || !info) return fail(nullptr, ADANERF_EINVAL, "NULL argument");
  ModelSetup ms;
  std::string err;
  int rc = setup_model(model_dir, opt, &ms, &err);
  if (rc) return fail(nullptr, rc, err);
  *info = ms.info;
  return ADANERF_OK;
}

int adanerf_host_depth_table(const char* model_dir, const adanerf_options* opt, float* ztab128) {
  if (!model_dir || !opt || !ztab128) return fail(nullptr, ADANERF_EINVAL, "NULL argument");
  ModelSetup ms;
  std::string err;
  int rc = setup_model(model_dir, opt, &ms, &err);
  if (rc) return fail(nullptr, rc, err);
  std::memcpy(ztab128, ms.ztab.data(), kBins * sizeof(float));
  return ADANERF_OK;
}

int adanerf_host_pack_weights(const char* model_dir, int32_t net, int32_t precision, void* weights_out, size_t* weights_bytes,
                              float* bias_out, size_t* bias_floats, int32_t* layer_out, int32_t* n_layers) try {
  if (!model_dir || !weights_bytes || !bias_floats || !n_layers) return fail(nullptr, ADANERF_EINVAL, "NULL argument");
  // precision 4 (shading nets only): bf16 WITHOUT the scaled packing -- for the CPU test that replays both blobs; no kernel consumes it
  const bool unscaled_bf16 = precision == 4 && net == 1;
  if (unscaled_bf16) precision = ADANERF_PREC_BF16;
  if (net < 0 || net > 1 || precision < 0 || precision > 3 || (precision == 3 && net != 0))
    return fail(nullptr, ADANERF_EINVAL, "net/precision out of range");
  Config cfg;
  std::string err;
  if (!cfg.load(model_dir, &err)) return fail(nullptr, ADANERF_EIO, err);
  if (cfg.posEncArgs.size() != 2) return fail(nullptr, ADANERF_EIO, "posEncArgs missing");
  const bool cfm = cfg.inFeatures.size() == 2 && cfg.inFeatures[0] == "RayMarchFromPoses" && cfg.inFeatures[1] == "RayMarchFromCoarse";
  const NetShape sh = shape_of(static_cast<int>(cfg.posEncArgs[0][0]), static_cast<int>(cfg.posEncArgs[0][1]), static_cast<int>(cfg.posEncArgs[1][0]),
                               static_cast<int>(cfg.posEncArgs[1][1]), cfg.raySampleInput.empty() ? 0 : cfg.raySampleInput[0], !cfm);
  TensorMap tm;
  if (!read_onnx_initializers(join_path(model_dir, net == 0 ? "model0.onnx" : "model1.onnx"), &tm, &err)) return fail(nullptr, ADANERF_EIO, err);
  PackedNet pn;
  // a coarse/fine directory holds two NeRF nets: model0.onnx packs like a shading net with the encoding posEncArgs[0]
  bool ok;
  if (net == 0 && cfm) {
    if (precision == 3) return fail(nullptr, ADANERF_EINVAL, "coarse/fine model: net 0 is a NeRF net (precision 0..2)");
    const NetShape shc = shape_of(sh.fp0, sh.fd0, sh.fp0, sh.fd0, 0, false);
    ok = pack_shading_net(tm, shc, elem_of(precision), &pn, &err);
  } else {
    ok = net == 0 ? pack_sampling_net(tm, sh, elem_of(precision), &pn, &err) : pack_shading_net(tm, sh, elem_of(precision), &pn, &err, !unscaled_bf16);
  }
  if (!ok) return fail(nullptr, ADANERF_EIO, err);
  if (weights_out) {
    if (*weights_bytes < pn.weights.size()) return fail(nullptr, ADANERF_EINVAL, "weights_out too small");
    std::memcpy(weights_out, pn.weights.data(), pn.weights.size());
  }
  if (bias_out) {
    if (*bias_floats < pn.bias.size()) return fail(nullptr, ADANERF_EINVAL, "bias_out too small");
    std::memcpy(bias_out, pn.bias.data(), pn.bias.size() * sizeof(float));
  }
  const bool rsi = net == 0 && pn.topo.ray_samples > 0;     // one more record: the raySampleInput block of layer 0
  const bool scaled = pn.relu_scaled;                        // one more record: the output exponents of a scaled (bf16) shading net
  const int32_t n_rec = static_cast<int32_t>(pn.w_off.size()) + (rsi ? 1 : 0) + (scaled ? 1 : 0);
  if (layer_out) {
    if (*n_layers < n_rec) return fail(nullptr, ADANERF_EINVAL, "layer_out too small");
    for (size_t i = 0; i < pn.w_off.size(); ++i) {
      layer_out[4 * i + 0] = static_cast<int32_t>(pn.w_off[i]);
      layer_out[4 * i + 1] = static_cast<int32_t>(pn.b_off[i]);
      layer_out[4 * i + 2] = pn.slots[i];
      layer_out[4 * i + 3] = pn.mtiles[i];
    }
    if (rsi) {
      const size_t i = pn.w_off.size();
      layer_out[4 * i + 0] = static_cast<int32_t>(pn.rsi_w_off);
      layer_out[4 * i + 1] = pn.topo.ray_samples;
      layer_out[4 * i + 2] = pe_slots(sh.lp0 ? sh.lp0 : sh.fp0);
      layer_out[4 * i + 3] = pn.mtiles[0];
    }
    if (scaled) {      // {alpha exponent, rgb exponent, 0, -1}: outputs of the packed network x 2^exponent = the network's own
      const size_t i = pn.w_off.size();
      layer_out[4 * i + 0] = pn.out_exp[0];
      layer_out[4 * i + 1] = pn.out_exp[1];
      layer_out[4 * i + 2] = 0;
      layer_out[4 * i + 3] = -1;
    }
  }
  *weights_bytes = pn.weights.size();
  *bias_floats = pn.bias.size();
  *n_layers = n_rec;
  return ADANERF_OK;
} catch (const std::exception& e) {
  return fail(nullptr, ADANERF_EIO, std::string("adanerf_host_pack_weights: ") + e.what());
}

int adanerf_destroy(adanerf_ctx* c) {
  if (!c) return ADANERF_OK;
  (void)hipSetDevice(c->device);
  if (c->peer_event) (void)hipEventDestroy(c->peer_event);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
  for (int32_t* p : c->pinned_totals) (void)hipHostFree(p);
  if (c->guard_host) (void)hipHostFree(c->guard_host);
  if (c->guard_ev) (void)hipEventDestroy(c->guard_ev);
  DevBuf* bufs[] = {&c->net0_split.w, &c->net0_split.b, &c->net0_f16.w, &c->net0_f16.b, &c->overflow, &c->net0.w, &c->net0.b, &c->net1[0].w, &c->net1[0].b, &c->net1[1].w, &c->net1[1].b, &c->net1[2].w, &c->net1[2].b,
                    &c->ztab, &c->rays, &c->oracle, &c->ray_offsets, &c->ray_counts, &c->selbin, &c->selw, &c->block_total,
                    &c->block_offset, &c->total, &c->sample_key, &c->sample_w, &c->raw, &c->sample_z, &c->rsi_z,
                    &c->netc[0].w, &c->netc[0].b, &c->netc[1].w, &c->netc[1].b, &c->netc[2].w, &c->netc[2].b, &c->ztab_coarse, &c->raw_coarse, &c->key_coarse,
                    &c->guard_mask, &c->refine_list, &c->guard_probe, &c->disp_scratch};
  for (DevBuf* b : bufs) dev_free(b);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return ADANERF_OK;
}

int adanerf_get_info(const adanerf_ctx* c, adanerf_info* info) {
  if (!c || !info) return ADANERF_EINVAL;
  *info = c->info;
  return ADANERF_OK;
}

int adanerf_set_camera(adanerf_ctx* c, const float pos[3], const float rot[9]) {
  if (!c) return ADANERF_EINVAL;
  if (!pos || !rot) return fail(c, ADANERF_EINVAL, "pos/rot == NULL");
  if (c->pos_bound.active) {
    // bf16 shading: the layers were scaled for positions below kPosIdentityBound with the camera inside the view cell (setup_model); a pose far
    // outside it (a free-fly viewer) is refused rather than rendered with activations cut by the clamped conversion
    double d2 = 0.0;
    for (int i = 0; i < 3; ++i) d2 += (pos[i] - c->pos_bound.center[i]) * (pos[i] - c->pos_bound.center[i]);
    const double bound = c->pos_bound.at(std::sqrt(d2));
    if (!(bound <= kPosIdentityBound)) {
      char msg[256];
      std::snprintf(msg, sizeof(msg), "camera %.3g away from the view-cell centre: sample positions can reach %.3g, beyond the %.0f the bf16 shading path's "
                    "layer scaling assumes -- create the context with precision fp16 or fp32 for such poses", std::sqrt(d2), bound, kPosIdentityBound);
      return fail(c, ADANERF_EUNSUPPORTED, msg);
    }
  }
  std::memcpy(c->rg.pos, pos, 3 * sizeof(float));
  std::memcpy(c->rg.rot, rot, 9 * sizeof(float));
  return ADANERF_OK;
}

int adanerf_sync(adanerf_ctx* c) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ADANERF_OK;
}

int adanerf_ray_features(adanerf_ctx* c, int32_t first_ray, int32_t n_rays, float* d_feat, float* d_rays) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  if (c->coarse_fine) return fail(c, ADANERF_EUNSUPPORTED, "coarse/fine models have no sampling network");
  if (first_ray < 0 || n_rays < 0 || first_ray + n_rays > c->info.rays_local) return fail(c, ADANERF_EINVAL, "ray range outside this context's rays");
  if (n_rays == 0) return ADANERF_OK;
  dim3 grid((n_rays + 255) / 256), block(256);
  const float* rz = reinterpret_cast<const float*>(c->rsi_z.p);
  const float d1 = c->cfg.depthRange[1];
  hipLaunchKernelGGL(ray_features_kernel, grid, block, 0, c->stream, c->rg, first_ray, n_rays, d_feat, d_rays, c->ray_samples, rz, d1, c->fp0, c->fd0);
  HIP_TRY(c, hipGetLastError());
  return ADANERF_OK;
}

int adanerf_sample_mlp(adanerf_ctx* c, int32_t first_ray, int32_t n_rays, float* d_oracle, float* d_rays) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  if (c->coarse_fine) return fail(c, ADANERF_EUNSUPPORTED, "coarse/fine models have no sampling network (adanerf_sample_uniform / adanerf_shade_mlp_coarse)");
  if (first_ray < 0 || n_rays < 0 || first_ray + n_rays > c->info.rays_local) return fail(c, ADANERF_EINVAL, "ray range outside this context's rays");
  return launch_sample_mlp(c, first_ray, n_rays, d_oracle, d_rays);
}

int adanerf_compact(adanerf_ctx* c, const float* d_oracle, int32_t n_rays, int32_t n_max, float thr, int32_t* d_off, int32_t* d_cnt,
                    uint32_t* d_key, float* d_w, int32_t* d_total) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  if (!d_oracle || !d_off || !d_cnt || !d_key || !d_w || !d_total) return fail(c, ADANERF_EINVAL, "NULL buffer");
  if (n_rays < 0 || n_max < 1 || n_max > kBins || thr < 0.f) return fail(c, ADANERF_EINVAL, "n_rays/n_max/thr out of range");
  if (thr == 0.f && n_max != kBins) return fail(c, ADANERF_EINVAL, "dense mode (thr == 0) requires n_max == 128");
  if (static_cast<int64_t>(n_rays) >= (1ll << 25)) return fail(c, ADANERF_EINVAL, "n_rays must be < 2^25 per batch");
  if (static_cast<int64_t>(n_rays) * n_max > 0x7fffffffll) return fail(c, ADANERF_EINVAL, "n_rays * n_max exceeds 2^31 - 1");
  int rc = ensure_compact_scratch(c, n_rays, n_max);
  if (rc) return rc;
  return launch_compact(c, d_oracle, n_rays, n_max, thr, d_off, d_cnt, d_key, d_w, d_total);
}

int adanerf_compact_guarded(adanerf_ctx* c, const float* d_approx, const float* d_exact, int32_t n_rays, int32_t n_max, float thr, float eps,
                            float eps_pair, int32_t audit_period, int32_t audit_phase, int32_t audit_fill_cap, int32_t audit_cycle, int32_t* d_off,
                            int32_t* d_cnt, uint32_t* d_key, float* d_w, int32_t* d_total, int32_t* d_refined, uint32_t* d_monitor) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  if (!d_approx || !d_exact || !d_off || !d_cnt || !d_key || !d_w || !d_total || !d_refined) return fail(c, ADANERF_EINVAL, "NULL buffer");
  if (n_rays < 0 || n_max < 1 || n_max > kPairMaxN || !(thr > 0.f) || !(eps > 0.f)) return fail(c, ADANERF_EINVAL, "n_rays/n_max/thr/eps out of range");
  if (audit_period > 32 || (audit_period > 0 && (audit_period & (audit_period - 1)) != 0)) return fail(c, ADANERF_EINVAL, "audit_period must be a power of two <= 32");
  if (static_cast<int64_t>(n_rays) >= (1ll << 25)) return fail(c, ADANERF_EINVAL, "n_rays must be < 2^25 per batch");
  if (n_rays == 0) return ADANERF_OK;
  int rc = ensure_compact_scratch(c, n_rays, n_max);
  if (rc) return rc;
  const int n_words = (n_rays + 31) / 32;
  if (c->guard_mask.bytes < n_words * sizeof(uint32_t) || c->refine_list.bytes < static_cast<size_t>(n_rays) * sizeof(int32_t) ||
      c->guard_probe.bytes < static_cast<size_t>(n_rays) * 2 * sizeof(float)) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if ((rc = dev_alloc(c, &c->guard_mask, n_words * sizeof(uint32_t)))) return rc;
    if ((rc = dev_alloc(c, &c->refine_list, static_cast<size_t>(n_rays) * sizeof(int32_t)))) return rc;
    if ((rc = dev_alloc(c, &c->guard_probe, static_cast<size_t>(n_rays) * 2 * sizeof(float)))) return rc;
  }
  const int period = audit_period > 0 ? audit_period : 0, phase = period ? (audit_phase & (period - 1)) : 0;
  SelectOut so = select_out(c, n_max, thr, d_cnt);
  so.guard_mask = reinterpret_cast<uint32_t*>(c->guard_mask.p);
  so.guard_eps = guard_band_of(c->transform, eps);
  so.guard_pair = guard_pair_of(c->transform, eps, eps_pair);
  so.audit_period = period;
  so.audit_phase = phase;
  so.guard_probe = reinterpret_cast<float*>(c->guard_probe.p);      // the cut values of the rays pass 2 will look at; their rows are d_approx itself
  const dim3 grid((n_rays + 127) / 128), block(256);
  hipLaunchKernelGGL(select_rows_kernel, grid, block, 0, c->stream, d_approx, n_rays, so, static_cast<const int32_t*>(nullptr));
  hipLaunchKernelGGL(refine_list_kernel, dim3((n_words + 255) / 256), dim3(256), 0, c->stream, reinterpret_cast<const uint32_t*>(c->guard_mask.p), n_words,
                     n_rays, period, phase, audit_fill_cap > 0 ? audit_fill_cap : 0, audit_cycle, reinterpret_cast<int32_t*>(c->refine_list.p), d_refined);
  so.guard_mask = nullptr;
  so.guard_band = eps;
  so.guard_band_pair = c->transform == kOracleRaw ? so.guard_pair : 0.f;
  so.guard_eps = 0.f;
  so.guard_pair = 0.f;
  so.audit_period = 0;
  so.refine_list = reinterpret_cast<const int32_t*>(c->refine_list.p);
  so.guard_rows = d_monitor ? const_cast<float*>(d_approx) : nullptr;      // pass 2 only reads them
  so.guard_seen = d_monitor;
  hipLaunchKernelGGL(select_rows_kernel, grid, block, 0, c->stream, d_exact, n_rays, so, static_cast<const int32_t*>(d_refined));
  return launch_expand(c, n_rays, n_max, kPairSegShift, d_off, d_cnt, d_key, d_w, d_total);
}

int adanerf_calibrate_guard(adanerf_ctx* c, int32_t n_poses, uint32_t seed, int32_t set, float* max_diff, float* max_pair_diff) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  if (!c->model0_hash) c->model0_hash = fnv1a64_file(join_path(c->model_dir, "model0.onnx"));
  return calibrate_guard(c, n_poses, seed, set != 0, max_diff, max_pair_diff);
}

int adanerf_guard_calibration_file(const adanerf_ctx* c, char* buf, size_t buf_bytes) {
  if (!c) return ADANERF_EINVAL;
  adanerf_ctx* m = const_cast<adanerf_ctx*>(c);
  if (!m->model0_hash) m->model0_hash = fnv1a64_file(join_path(c->model_dir, "model0.onnx"));
  const std::string p = guard_record_path(c);
  if (buf && buf_bytes > 0) {
    const size_t n = std::min(p.size(), buf_bytes - 1);
    std::memcpy(buf, p.data(), n);
    buf[n] = 0;
  }
  return static_cast<int>(p.size() + 1);
}

int adanerf_abi_version(void) { return ADANERF_ABI_VERSION; }

int adanerf_struct_sizes(int32_t sizes_out[3]) {
  if (!sizes_out) return ADANERF_EINVAL;
  sizes_out[0] = static_cast<int32_t>(sizeof(adanerf_options));
  sizes_out[1] = static_cast<int32_t>(sizeof(adanerf_info));
  sizes_out[2] = static_cast<int32_t>(sizeof(adanerf_stats));
  return ADANERF_OK;
}

int adanerf_shade_features(adanerf_ctx* c, const float* d_rays, const uint32_t* d_key, int32_t n_samples, float* d_feat) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  if (!d_rays || !d_key || !d_feat || n_samples < 0) return fail(c, ADANERF_EINVAL, "bad argument");
  if (n_samples == 0) return ADANERF_OK;
  ShadeArgs a{};
  a.sp = c->sp;
  a.rays = d_rays;
  a.sample_key = d_key;
  a.max_samples = n_samples;
  hipLaunchKernelGGL(shade_features_kernel, dim3((n_samples + 255) / 256), dim3(256), 0, c->stream, a, d_feat, c->fp1, c->fd1);
  HIP_TRY(c, hipGetLastError());
  return ADANERF_OK;
}

int adanerf_shade_mlp(adanerf_ctx* c, const float* d_rays, const uint32_t* d_key, const int32_t* d_total, int32_t max_samples,
                      int32_t precision, float* d_raw) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  if (!d_rays || !d_key || !d_raw || max_samples < 0) return fail(c, ADANERF_EINVAL, "bad argument");
  return launch_shade_mlp(c, d_rays, d_key, d_total, max_samples, precision < 0 ? c->info.precision : precision, d_raw);
}

int adanerf_shade_mlp_z(adanerf_ctx* c, const float* d_rays, const uint32_t* d_key, const float* d_z, const int32_t* d_total,
                        int32_t max_samples, int32_t precision, float* d_raw) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  if (!d_rays || !d_key || !d_raw || max_samples < 0) return fail(c, ADANERF_EINVAL, "bad argument");
  return launch_shade_mlp(c, d_rays, d_key, d_total, max_samples, precision < 0 ? c->info.precision : precision, d_raw, d_z);
}

int adanerf_sample_uniform(adanerf_ctx* c, int32_t first_ray, int32_t n_rays, float* d_rays, int32_t* d_off, int32_t* d_cnt, uint32_t* d_key,
                           int32_t* d_total) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  if (!c->coarse_fine) return fail(c, ADANERF_EUNSUPPORTED, "adanerf_sample_uniform: not a coarse/fine model");
  if (!d_rays || !d_off || !d_cnt || !d_key || !d_total) return fail(c, ADANERF_EINVAL, "NULL buffer");
  if (first_ray < 0 || n_rays < 0 || first_ray + n_rays > c->info.rays_local) return fail(c, ADANERF_EINVAL, "ray range outside this context's rays");
  int rc = launch_camera_rays(c, first_ray, n_rays, d_rays);
  return rc ? rc : launch_sample_uniform(c, n_rays, c->n_coarse, d_off, d_cnt, d_key, d_total);
}

int adanerf_shade_mlp_coarse(adanerf_ctx* c, const float* d_rays, const uint32_t* d_key, const int32_t* d_total, int32_t max_samples,
                             int32_t precision, float* d_raw) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  if (!c->coarse_fine) return fail(c, ADANERF_EUNSUPPORTED, "adanerf_shade_mlp_coarse: not a coarse/fine model");
  if (!d_rays || !d_key || !d_raw || max_samples < 0) return fail(c, ADANERF_EINVAL, "bad argument");
  return launch_shade_mlp(c, d_rays, d_key, d_total, max_samples, precision < 0 ? c->info.precision : precision, d_raw, nullptr, true);
}

int adanerf_sample_from_coarse(adanerf_ctx* c, const float* d_raw_coarse, const float* d_rays, int32_t n_rays, int32_t* d_off, int32_t* d_cnt,
                               uint32_t* d_key, float* d_z, int32_t* d_total) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  if (!c->coarse_fine) return fail(c, ADANERF_EUNSUPPORTED, "adanerf_sample_from_coarse: not a coarse/fine model");
  if (!d_raw_coarse || !d_rays || !d_off || !d_cnt || !d_key || !d_z || !d_total) return fail(c, ADANERF_EINVAL, "NULL buffer");
  if (n_rays < 0 || static_cast<int64_t>(n_rays) * c->info.num_samples > 0x7fffffffll || n_rays >= (1 << 25))
    return fail(c, ADANERF_EINVAL, "n_rays out of range");
  return launch_sample_fine(c, d_raw_coarse, d_rays, n_rays, d_off, d_cnt, d_key, d_z, d_total);
}

int adanerf_sample_pdf(adanerf_ctx* c, const float* d_oracle, int32_t n_rays, int32_t n, int32_t* d_off, int32_t* d_cnt, uint32_t* d_key,
                       float* d_w, float* d_z, int32_t* d_total) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  if (!d_oracle || !d_off || !d_cnt || !d_key || !d_w || !d_z || !d_total) return fail(c, ADANERF_EINVAL, "NULL buffer");
  if (n_rays < 0 || n < 1 || n > 4096) return fail(c, ADANERF_EINVAL, "n_rays/n out of range");
  if (static_cast<int64_t>(n_rays) * n > 0x7fffffffll || n_rays >= (1 << 25)) return fail(c, ADANERF_EINVAL, "n_rays * n too large");
  return launch_sample_pdf(c, d_oracle, n_rays, n, d_off, d_cnt, d_key, d_w, d_z, d_total);
}

int adanerf_composite_classic(adanerf_ctx* c, const float* d_raw, const float* d_z, const float* d_rays, int32_t n_rays, int32_t n,
                              float* d_rgb, void* d_rgba8) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  if (!d_raw || !d_z || !d_rays || n_rays < 0 || n < 1) return fail(c, ADANERF_EINVAL, "bad argument");
  return launch_composite_classic(c, d_raw, d_z, d_rays, n_rays, n, d_rgb, d_rgba8);
}

int adanerf_composite(adanerf_ctx* c, const float* d_raw, const float* d_w, const int32_t* d_off, const int32_t* d_cnt, int32_t n_rays,
                      float* d_rgb, void* d_rgba8) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  if (!d_raw || !d_w || !d_off || !d_cnt || n_rays < 0) return fail(c, ADANERF_EINVAL, "bad argument");
  return launch_composite(c, d_raw, d_w, d_off, d_cnt, n_rays, d_rgb, d_rgba8);
}

namespace {

constexpr size_t kMaxProfiledBatches = 1 << 16;

int sum_stats(adanerf_ctx* c, adanerf_stats* stats) {
  std::memset(stats, 0, sizeof(*stats));
  const size_t nb = c->events_used / 5;
  for (size_t b = 0; b < nb; ++b) {
    hipEvent_t* ev = &c->events[b * 5];
    float t;
    HIP_TRY(c, hipEventElapsedTime(&t, ev[0], ev[1]));
    stats->ms_sample_mlp += t;
    HIP_TRY(c, hipEventElapsedTime(&t, ev[1], ev[2]));
    stats->ms_compact += t;
    HIP_TRY(c, hipEventElapsedTime(&t, ev[2], ev[3]));
    stats->ms_shade_mlp += t;
    HIP_TRY(c, hipEventElapsedTime(&t, ev[3], ev[4]));
    stats->ms_composite += t;
    HIP_TRY(c, hipEventElapsedTime(&t, ev[0], ev[4]));
    stats->ms_total += t;
    stats->total_samples += c->pinned_totals[b][0];
    if (c->sampling_mode == ADANERF_SAMPLING_GUARDED) {
      stats->rays_refined += c->pinned_totals[b][4];
      float seen, pair;                                // cumulative on the device: the latest batch holds the running values
      std::memcpy(&seen, &c->pinned_totals[b][8], sizeof(seen));
      std::memcpy(&pair, &c->pinned_totals[b][10], sizeof(pair));
      stats->guard_max_seen = std::max(stats->guard_max_seen, seen);
      stats->guard_pair_seen = std::max(stats->guard_pair_seen, pair);
      stats->guard_violations = std::max(stats->guard_violations, c->pinned_totals[b][9]);
      stats->guard_audit_mismatch = std::max(stats->guard_audit_mismatch, c->pinned_totals[b][11]);
      stats->guard_audited = std::max(stats->guard_audited, c->pinned_totals[b][12]);
    }
  }
  int32_t ovf = 0;
  HIP_TRY(c, hipMemcpy(&ovf, c->overflow.p, sizeof(int32_t), hipMemcpyDeviceToHost));
  stats->sampling_overflow = ovf;
  stats->batches = static_cast<int32_t>(nb);
  stats->shade_launches = static_cast<int32_t>(nb);
  stats->sample_launches = static_cast<int32_t>(nb);
  // plus whatever an earlier, full event pool was folded into
  const adanerf_stats& f = c->folded;
  stats->total_samples += f.total_samples;
  stats->rays_refined += f.rays_refined;
  stats->guard_max_seen = std::max(stats->guard_max_seen, f.guard_max_seen);
  stats->guard_violations = std::max(stats->guard_violations, f.guard_violations);
  stats->guard_pair_seen = std::max(stats->guard_pair_seen, f.guard_pair_seen);
  stats->guard_audit_mismatch = std::max(stats->guard_audit_mismatch, f.guard_audit_mismatch);
  stats->guard_audited = std::max(stats->guard_audited, f.guard_audited);
  stats->guard_widened = c->guard_widened;
  stats->ms_total += f.ms_total;
  stats->ms_sample_mlp += f.ms_sample_mlp;
  stats->ms_compact += f.ms_compact;
  stats->ms_shade_mlp += f.ms_shade_mlp;
  stats->ms_composite += f.ms_composite;
  stats->batches += f.batches;
  stats->shade_launches += f.shade_launches;
  stats->sample_launches += f.sample_launches;
  return ADANERF_OK;
}

}  // namespace

int adanerf_copy_result_sampling_network(adanerf_ctx* c, const float* d_oracle, int32_t n_rays, void* d_rgba8) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  if (!d_oracle || !d_rgba8) return fail(c, ADANERF_EINVAL, "NULL buffer");
  if (n_rays < 0) return fail(c, ADANERF_EINVAL, "n_rays out of range");
  if (n_rays == 0) return ADANERF_OK;
  hipLaunchKernelGGL(oracle_view_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, c->stream, d_oracle, n_rays, static_cast<uchar4*>(d_rgba8));
  HIP_TRY(c, hipGetLastError());
  return ADANERF_OK;
}

int adanerf_render_oracle(adanerf_ctx* c, void* d_rgba8) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  if (!d_rgba8) return fail(c, ADANERF_EINVAL, "NULL buffer");
  if (c->coarse_fine) return fail(c, ADANERF_EUNSUPPORTED, "coarse/fine models have no sampling network to view");
  const int R = c->info.rays_local, B = c->info.batch_rays;
  int rc = ensure_batch_buffers(c, std::min(B, std::max(R, 1)), c->info.num_samples);
  if (rc) return rc;
  float* rays = reinterpret_cast<float*>(c->rays.p);
  float* oracle = reinterpret_cast<float*>(c->oracle.p);
  for (int first = 0; first < R; first += B) {
    const int n = std::min(B, R - first);
    if ((rc = launch_sample_mlp(c, first, n, oracle, rays))) return rc;
    if ((rc = adanerf_copy_result_sampling_network(c, oracle, n, static_cast<char*>(d_rgba8) + static_cast<size_t>(first) * 4))) return rc;
  }
  return ADANERF_OK;
}

namespace {

// The band of the guarded selection rests on two measured assumptions (|fp16 output - split output| <= guard_eps on every raw
// output; the error of a (kept - candidate) difference <= guard_eps_pair) that the second pass re-measures on the whole row of every
// re-evaluated ray, and on an audit of the outcome (decided rays re-evaluated anyway).  All of it in RAW-output units -- the units
// guard_eps is in (round 3 compared transformed values with a raw band).  If a frame saw a bound violated, every later frame runs
// with ADANERF_GUARD_CALIB_MARGIN x the largest errors seen so far; an audit mismatch without a violated bound (the rule itself would
// be wrong) falls back to the doubled single-value bound with no pair bound.  adanerf_info follows, adanerf_stats.guard_widened
// counts.  Non-blocking: looks only at a copy that has already arrived.
void poll_guard(adanerf_ctx* c) {
  if (!c->guard_ev_pending) return;
  const hipError_t q = hipEventQuery(c->guard_ev);
  (void)hipGetLastError();      // hipErrorNotReady is not a failure of this context
  if (q != hipSuccess) return;
  c->guard_ev_pending = false;
  float seen, pair;
  std::memcpy(&seen, &c->guard_host[0], sizeof(seen));
  std::memcpy(&pair, &c->guard_host[2], sizeof(pair));
  const int viol = c->guard_host[1], mism = c->guard_host[3];
  bool widened = false;
  if (viol > c->guard_viol_seen) {
    c->guard_viol_seen = viol;
    if (seen > c->guard_eps) {
      // the bound on single values was exceeded: widen it, and drop the measured pair bound -- it was measured over the candidates
      // the narrower band defined; 2 x the new bound is what holds by construction
      c->guard_eps = ADANERF_GUARD_CALIB_MARGIN * seen;
      c->guard_eps_pair = 2.0f * c->guard_eps;
      widened = true;
    } else if (pair > c->guard_eps_pair) {
      c->guard_eps_pair = ADANERF_GUARD_CALIB_MARGIN * pair;
      widened = true;
    }
  }
  if (mism > c->guard_mism_seen) {
    c->guard_mism_seen = mism;
    if (!widened) {
      c->guard_eps *= 2.0f;
      c->guard_eps_pair = 2.0f * c->guard_eps;
      widened = true;
    }
  }
  if (widened) {
    c->guard_eps_pair = std::min(std::max(c->guard_eps_pair, ADANERF_GUARD_EPS_MIN), 2.0f * c->guard_eps);
    c->info.guard_eps = c->guard_eps;
    c->info.guard_eps_pair = c->guard_eps_pair;
    c->info.guard_calib_source = ADANERF_GUARD_FROM_MONITOR;
    ++c->guard_widened;
  }
}

}  // namespace

int adanerf_render(adanerf_ctx* c, void* d_rgba8, float* d_rgb, adanerf_stats* stats) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  poll_guard(c);
  const int R = c->info.rays_local, B = c->info.batch_rays, N = c->info.num_samples;
  const float thr = c->info.threshold;
  const int n_batches = R > 0 ? (R + B - 1) / B : 0;
  int rc = ensure_batch_buffers(c, std::min(B, std::max(R, 1)), N);
  if (rc) return rc;
  if (stats && c->profiling && c->events_used) {   // a synchronous call starts a fresh record
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->events_used = 0;
    c->prof_frames = 0;
    c->folded = adanerf_stats{};
  }
  // a single frame of more batches than the event pool may hold is rendered without per-stage timing (stats then carry the
  // sample totals of nothing and zero times) rather than growing the pool without bound
  const bool record = (stats || c->profiling) && static_cast<size_t>(n_batches) <= kMaxProfiledBatches;
  if (record && c->events_used && c->events_used / 5 + n_batches > kMaxProfiledBatches) {
    // the event pool is full: fold what it holds into the running record (one synchronisation per 65 536 batches)
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    adanerf_stats part;
    if ((rc = sum_stats(c, &part))) return rc;
    c->folded = part;
    c->events_used = 0;
  }
  if (record) {
    const size_t need = c->events_used + static_cast<size_t>(n_batches) * 5;
    while (c->events.size() < need) {
      hipEvent_t e;
      HIP_TRY(c, hipEventCreate(&e));
      c->events.push_back(e);
    }
    while (c->pinned_totals.size() < need / 5) {
      int32_t* p = nullptr;
      HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&p), 16 * sizeof(int32_t), hipHostMallocDefault));   // image of the `total` block
      std::memset(p, 0, 16 * sizeof(int32_t));
      c->pinned_totals.push_back(p);
    }
  }
  float* rays = reinterpret_cast<float*>(c->rays.p);
  float* oracle = reinterpret_cast<float*>(c->oracle.p);
  int32_t* off = reinterpret_cast<int32_t*>(c->ray_offsets.p);
  const bool cfm = c->coarse_fine;
  int32_t* cnt = reinterpret_cast<int32_t*>(c->ray_counts.p);
  uint32_t* key = reinterpret_cast<uint32_t*>(c->sample_key.p);
  float* sw = reinterpret_cast<float*>(c->sample_w.p);
  float* raw = reinterpret_cast<float*>(c->raw.p);
  int32_t* total = reinterpret_cast<int32_t*>(c->total.p);
  for (int b = 0; b < n_batches; ++b) {
    const int first = b * B, n = std::min(B, R - first);
    hipEvent_t* ev = record ? &c->events[c->events_used] : nullptr;
    if (ev) HIP_TRY(c, hipEventRecord(ev[0], c->stream));
    const bool pdf = c->info.sampler_mode == ADANERF_SAMPLER_PDF || cfm;      // explicit sample depths + classic compositing
    if (cfm) {
      // vanilla NeRF: camera rays, Nc uniform samples, coarse network ("sample_mlp" in the statistics: the first network)
      uint32_t* keyc = reinterpret_cast<uint32_t*>(c->key_coarse.p);
      float* rawc = reinterpret_cast<float*>(c->raw_coarse.p);
      if ((rc = launch_camera_rays(c, first, n, rays))) return rc;
      if ((rc = launch_sample_uniform(c, n, c->n_coarse, off, cnt, keyc, total))) return rc;
      if ((rc = launch_shade_mlp(c, rays, keyc, total, n * c->n_coarse, c->info.precision, rawc, nullptr, true))) return rc;
      if (ev) HIP_TRY(c, hipEventRecord(ev[1], c->stream));
      // weights -> pdf -> Nf more depths, merged with the Nc coarse ones ("compact")
      if ((rc = launch_sample_fine(c, rawc, rays, n, off, cnt, key, reinterpret_cast<float*>(c->sample_z.p), total))) return rc;
    }
    // adaptive selection in the epilogue of the sampling kernel: the [R,128] oracle values never reach HBM
    const bool fused = !pdf && thr > 0.f && use_pair_select(c, N) && c->sampling_mode != 1 && (!c->generic0 || generic_split_sampling(c)) &&
                       !(c->opt.flags & ADANERF_FLAG_KEEP_ORACLE);
    if (cfm) {
      rc = ADANERF_OK;
    } else if (fused) {
      const SelectOut so = select_out(c, N, thr, cnt);
      rc = launch_sample_mlp(c, first, n, nullptr, rays, &so);
    } else {
      rc = launch_sample_mlp(c, first, n, oracle, rays);
    }
    if (rc) return rc;
    if (ev && !cfm) HIP_TRY(c, hipEventRecord(ev[1], c->stream));
    float* sz = reinterpret_cast<float*>(c->sample_z.p);
    if (cfm) rc = ADANERF_OK;
    else if (pdf) rc = launch_sample_pdf(c, oracle, n, N, off, cnt, key, sw, sz, total);
    else if (fused) rc = launch_expand(c, n, N, kPairSegShift, off, cnt, key, sw, total);
    else if (thr == 0.f) {      // dense: implicit keys, the oracle buffer is the weight array (dense_offsets_kernel)
      hipLaunchKernelGGL(dense_offsets_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, off, cnt, total);
      HIP_TRY(c, hipGetLastError());
      rc = ADANERF_OK;
    } else rc = launch_compact(c, oracle, n, N, thr, off, cnt, key, sw, total);
    if (rc) return rc;
    const bool dense_implicit = !cfm && !pdf && !fused && thr == 0.f;
    const uint32_t* key_s = dense_implicit ? nullptr : key;
    const float* sw_s = dense_implicit ? oracle : sw;
    if (ev) HIP_TRY(c, hipEventRecord(ev[2], c->stream));
    const int64_t max_s = static_cast<int64_t>(n) * N;   // <= INT32_MAX: checked by setup_model
    if ((rc = launch_shade_mlp(c, rays, key_s, total, static_cast<int>(max_s), c->info.precision, raw, pdf ? sz : nullptr))) return rc;
    if (ev) HIP_TRY(c, hipEventRecord(ev[3], c->stream));
    float* rgb_b = d_rgb ? d_rgb + static_cast<size_t>(first) * 3 : nullptr;
    void* rgba_b = d_rgba8 ? static_cast<char*>(d_rgba8) + static_cast<size_t>(first) * 4 : nullptr;
    float* depth_b = c->aux_depth ? c->aux_depth + first : nullptr;
    float* acc_b = c->aux_acc ? c->aux_acc + first : nullptr;
    if (c->aux_disp && (!depth_b || !acc_b)) {      // disparity needs both maps of the batch
      if ((rc = dev_alloc(c, &c->disp_scratch, static_cast<size_t>(B) * 2 * sizeof(float)))) return rc;
      if (!depth_b) depth_b = reinterpret_cast<float*>(c->disp_scratch.p);
      if (!acc_b) acc_b = reinterpret_cast<float*>(c->disp_scratch.p) + B;
    }
    if (pdf) rc = launch_composite_classic(c, raw, sz, rays, n, N, rgb_b, rgba_b, depth_b, acc_b);
    else rc = launch_composite(c, raw, sw_s, off, cnt, n, rgb_b, rgba_b, key_s, depth_b, acc_b);
    if (rc) return rc;
    if (c->aux_disp && n > 0) {
      hipLaunchKernelGGL(disp_map_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, depth_b, acc_b, n, c->aux_disp + first);
      HIP_TRY(c, hipGetLastError());
    }
    if (ev) {
      HIP_TRY(c, hipEventRecord(ev[4], c->stream));
      HIP_TRY(c, hipMemcpyAsync(c->pinned_totals[c->events_used / 5], total, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
      c->events_used += 5;
    }
  }
  if (c->sampling_mode == ADANERF_SAMPLING_GUARDED && c->guard_mask.p && n_batches > 0) ++c->guard_frame;      // the audit moves on
  if (c->sampling_mode == ADANERF_SAMPLING_GUARDED && c->guard_mask.p && n_batches > 0 && !c->guard_ev_pending) {
    if (!c->guard_host) {
      HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->guard_host), 8 * sizeof(int32_t)));
      std::memset(c->guard_host, 0, 8 * sizeof(int32_t));
      HIP_TRY(c, hipEventCreateWithFlags(&c->guard_ev, hipEventDisableTiming));
    }
    HIP_TRY(c, hipMemcpyAsync(c->guard_host, total + 8, 5 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipEventRecord(c->guard_ev, c->stream));
    c->guard_ev_pending = true;
  }
  if (record) c->prof_frames++;
  if (stats) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    poll_guard(c);
    if ((rc = sum_stats(c, stats))) return rc;
    stats->rays = R;
    if (n_batches > 0 && record) {   // wall span of this frame, first launch -> last kernel end
      const size_t e0 = c->events_used - static_cast<size_t>(n_batches) * 5;
      HIP_TRY(c, hipEventElapsedTime(&stats->ms_total, c->events[e0], c->events[c->events_used - 1]));
    }
    c->events_used = 0;
    c->prof_frames = 0;
    c->folded = adanerf_stats{};
  }
  return ADANERF_OK;
}

int adanerf_set_aux_outputs(adanerf_ctx* c, float* d_depth_map, float* d_acc_map) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  c->aux_depth = d_depth_map;
  c->aux_acc = d_acc_map;
  return ADANERF_OK;
}

int adanerf_set_disp_output(adanerf_ctx* c, float* d_disp_map) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  c->aux_disp = d_disp_map;
  return ADANERF_OK;
}

int adanerf_set_stream(adanerf_ctx* c, void* hip_stream) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->own_stream;
  return ADANERF_OK;
}

int adanerf_set_profiling(adanerf_ctx* c, int32_t enabled) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->profiling = enabled != 0;
  c->events_used = 0;
  c->prof_frames = 0;
  c->folded = adanerf_stats{};
  return ADANERF_OK;
}

int adanerf_collect_stats(adanerf_ctx* c, adanerf_stats* stats, int32_t* frames) {
  if (!c || !stats) return ADANERF_EINVAL;
  BIND(c);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  int rc = sum_stats(c, stats);
  if (rc) return rc;
  stats->rays = c->info.rays_local;
  if (frames) *frames = c->prof_frames;
  c->events_used = 0;
  c->prof_frames = 0;
  c->folded = adanerf_stats{};
  return ADANERF_OK;
}

int adanerf_assemble_strips(adanerf_ctx* c, const void* d_gathered, void* d_image) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  if (!d_gathered || !d_image) return fail(c, ADANERF_EINVAL, "NULL buffer");
  const int n = c->info.width * c->info.height;
  hipLaunchKernelGGL(assemble_strips_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, reinterpret_cast<const uchar4*>(d_gathered),
                     reinterpret_cast<uchar4*>(d_image), c->info.width, c->info.height, c->rg.strip_rows, c->rg.world, c->info.rays_local_max);
  HIP_TRY(c, hipGetLastError());
  return ADANERF_OK;
}

int adanerf_gather_to(adanerf_ctx* dst, void* d_dst, adanerf_ctx* src, const void* d_src, size_t bytes) {
  if (!dst || !src) return ADANERF_EINVAL;
  if (!d_dst || !d_src) return fail(dst, ADANERF_EINVAL, "NULL buffer");
  if (bytes == 0) return ADANERF_OK;
  adanerf_ctx* c = src;
  BIND(src);
  if (src->device != dst->device) {
    if (dst->device < 64 && !((src->peer_tried >> dst->device) & 1)) {
      src->peer_tried |= 1ull << dst->device;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, src->device, dst->device) == hipSuccess && can)
        (void)hipDeviceEnablePeerAccess(dst->device, 0);      // direct xGMI stores; "already enabled" is fine
      (void)hipGetLastError();                                // without peer access the copy is staged by the runtime
    }
    HIP_TRY(c, hipMemcpyPeerAsync(d_dst, dst->device, d_src, src->device, bytes, src->stream));
  } else {
    HIP_TRY(c, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, src->stream));
  }
  if (src == dst || src->stream == dst->stream) return ADANERF_OK;
  if (!src->peer_event) HIP_TRY(c, hipEventCreateWithFlags(&src->peer_event, hipEventDisableTiming));
  HIP_TRY(c, hipEventRecord(src->peer_event, src->stream));
  c = dst;
  BIND(dst);
  HIP_TRY(c, hipStreamWaitEvent(dst->stream, src->peer_event, 0));
  return ADANERF_OK;
}

int adanerf_probe_mfma(adanerf_ctx* c, int32_t operands, int32_t f16, float target_ms, float* tflops, float* clock_mhz) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  if (operands < 0 || operands > 3 || !(target_ms > 0.f) || target_ms > 5000.f || !tflops || !clock_mhz)
    return fail(c, ADANERF_EINVAL, "operands must be 0..3, target_ms in (0, 5000], outputs non-NULL");
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  double tf = 0, mhz = 0;
  HIP_TRY(c, probe_mfma_rate(operands, f16 != 0, target_ms, c->info.compute_units, c->stream, &tf, &mhz));
  *tflops = static_cast<float>(tf);
  *clock_mhz = static_cast<float>(mhz);
  return ADANERF_OK;
}

int adanerf_malloc(adanerf_ctx* c, size_t bytes, void** d_out) {
  if (!c || !d_out) return ADANERF_EINVAL;
  BIND(c);
  HIP_TRY(c, hipMalloc(d_out, bytes ? bytes : 1));
  return ADANERF_OK;
}
int adanerf_free(adanerf_ctx* c, void* d_ptr) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipFree(d_ptr));
  return ADANERF_OK;
}
int adanerf_memcpy_h2d(adanerf_ctx* c, void* d_dst, const void* src, size_t bytes) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  HIP_TRY(c, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ADANERF_OK;
}
int adanerf_memcpy_d2h(adanerf_ctx* c, void* dst, const void* d_src, size_t bytes) {
  if (!c) return ADANERF_EINVAL;
  BIND(c);
  HIP_TRY(c, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ADANERF_OK;
}

int adanerf_get_buffer(adanerf_ctx* c, int32_t which, void** d_out, size_t* bytes_out) {
  if (!c || !d_out) return ADANERF_EINVAL;
  DevBuf* b = nullptr;
  switch (which) {
    case ADANERF_BUF_RAYS: b = &c->rays; break;
    case ADANERF_BUF_ORACLE: b = &c->oracle; break;
    case ADANERF_BUF_RAY_OFFSETS: b = &c->ray_offsets; break;
    case ADANERF_BUF_RAY_COUNTS: b = &c->ray_counts; break;
    case ADANERF_BUF_SAMPLE_KEY: b = &c->sample_key; break;
    case ADANERF_BUF_SAMPLE_W: b = &c->sample_w; break;
    case ADANERF_BUF_RAW: b = &c->raw; break;
    case ADANERF_BUF_TOTAL: b = &c->total; break;
    case ADANERF_BUF_SAMPLE_Z: b = &c->sample_z; break;
    case ADANERF_BUF_RAW_COARSE: b = &c->raw_coarse; break;
    default: return fail(c, ADANERF_EINVAL, "unknown buffer id");
  }
  *d_out = b->p;
  if (bytes_out) *bytes_out = b->bytes;
  return ADANERF_OK;
}

}  // extern "C"
