/*
 * adanerf_hip.h -- C ABI of libadanerf_hip.so, the MI355X (gfx950) AdaNeRF inference renderer.
 *
 * Drop-in boundary: these entry points are what the reference viewer's host pipeline binds in
 * place of its CUDA launchers + TensorRT contexts.  Reference interface replaced (paths relative
 * to /root/reference/adanerf_real_time_viewer):
 *
 *   adanerf_create / adanerf_destroy     NeuralRenderer::init (src/neuralrenderer.cpp:59-139):
 *                                        Config::load (src/config.cpp:270-344), Encoding::load,
 *                                        RayMarchFromPoses::create (src/featureset.cpp:67-140),
 *                                        ImageGenerator::load/initEngine (src/imagegenerator.cpp:84-226)
 *   adanerf_set_camera                   Camera::UpdateFeatureRot/getPosition (src/camera.cpp:143-201)
 *   adanerf_render                       ImageGenerator::inference, 2-context adaptive branch
 *                                        (src/imagegenerator.cpp:282-394) as called from
 *                                        NeuralRenderer::render (src/neuralrenderer.cpp:146-182)
 *   adanerf_ray_features                 updateSpherePosDirBatchedUnrolledEnc
 *                                        (include/cuda/adanerf_cuda_kernels.cuh:41-45)
 *   adanerf_sample_mlp                   contexts[0]->executeV2 (src/imagegenerator.cpp:308-312)
 *   adanerf_compact                      updateRayMarchFromPosesAdaptive, selection half
 *                                        (include/cuda/adanerf_cuda_kernels.cuh:52-58; kernels
 *                                        src/cuda/adaptive_cuda_kernels.cu:229-607)
 *   adanerf_shade_features               updateRayMarchFromPosesAdaptive, feature half
 *                                        (rayMarchFromPosesAdaptive[NDC], adaptive_cuda_kernels.cu:609-739)
 *   adanerf_shade_mlp                    contexts[1]->executeV2 (src/imagegenerator.cpp:336-344)
 *   adanerf_composite                    copyResultRaymarchAdaptiveMultDepth
 *                                        (include/cuda/adanerf_cuda_kernels.cuh:30-32)
 *
 * Numerics follow the reference's PyTorch path (src/evaluate.py over nerf_raymarch_common.py /
 * features.py / models.py) wherever it disagrees with the viewer.
 *
 * Conventions
 *   - every function returns 0 on success, a negative ADANERF_E* code on failure; the message is
 *     available from adanerf_last_error().  Nothing throws or aborts across this boundary.
 *   - one context = one device = one non-default HIP stream.  A context is not thread-safe;
 *     distinct contexts are independent.
 *   - pointers named d_* are DEVICE pointers on the context's device; all others are host.
 *   - the library owns every buffer behind the context; the caller owns output buffers it passes.
 *   - calls enqueue on the context's stream; adanerf_sync() (or stats != NULL) waits.
 */
#ifndef ADANERF_HIP_H
#define ADANERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADANERF_ABI_VERSION 4   /* 2: adanerf_info.view_cell_size, .num_samples_coarse appended
                                   3: ADANERF_SAMPLING_GUARDED, adanerf_options.guard_eps, adanerf_stats.rays_refined (carved out of
                                      reserved words), adanerf_info.guard_eps APPENDED (adanerf_info grew by 4 bytes)
                                   4: audited guard band: adanerf_options.guard_eps_pair / .guard_audit_period (carved out of the
                                      reserved words: size unchanged); adanerf_info and adanerf_stats GROW (guard fields + reserved
                                      words for later versions); adanerf_compact_guarded / adanerf_calibrate_guard take more
                                      arguments.  adanerf_get_info / adanerf_render write the whole struct: a caller checks
                                      adanerf_abi_version() == ADANERF_ABI_VERSION (and may compare adanerf_struct_sizes with its
                                      own sizeof) once after loading the library -- there is no per-call size argument */

enum {
  ADANERF_OK = 0,
  ADANERF_EINVAL = -1,    /* bad argument */
  ADANERF_EIO = -2,       /* model directory / file format problem */
  ADANERF_EDEVICE = -3,   /* HIP runtime error */
  ADANERF_EUNSUPPORTED = -4 /* config outside the supported north-star path */
};

/* arithmetic of the sampling MLP (its outputs drive the bit-exact sample selection) */
enum {
  ADANERF_SAMPLING_SPLIT_FP16 = 0, /* default: x = hi + 2^-11 lo' (two fp16), 3 x v_mfma_f32_32x32x16_f16 per term,
                                      fp32 accumulate; 22-bit operands, measured error <= fp32 sgemm's */
  ADANERF_SAMPLING_FP32 = 1,       /* v_mfma_f32_32x32x2_f32: bitwise an fp32 fma chain */
  ADANERF_SAMPLING_FP16 = 2,       /* opt-in speed mode: plain fp16 operands, one MFMA per term, fp32 accumulate -- what the
                                      reference viewer's TensorRT engine does (imagegenerator.cpp:155-156).  Raw outputs
                                      are then within ~3e-3 of fp32 and the selected bins differ from the fp32 PyTorch
                                      path on 0.3-1.5 % of rays (tools/probes/sampling_agreement.py); 3x faster. */
  ADANERF_SAMPLING_GUARDED = 3     /* two-precision selection with the results of ADANERF_SAMPLING_SPLIT_FP16: every ray goes
                                      through the plain-fp16 engine; a ray whose selection could change under a perturbation
                                      of guard_eps of its raw outputs (a value within the band around the threshold, a
                                      not-kept value closer than guard_eps_pair to the N-th largest, a tie, a non-finite value) is
                                      re-evaluated by the split-precision engine, which overwrites its row.  Selections are
                                      those of the split engine as long as |fp16 output - split output| <= guard_eps and the
                                      error of a (kept - candidate) difference <= guard_eps_pair hold.  Both assumptions are
                                      MEASURED on every re-evaluated ray (whole row: adanerf_stats.guard_max_seen /
                                      .guard_pair_seen / .guard_violations) and the OUTCOME is audited: every frame a rotating
                                      1 / guard_audit_period of ALL rays -- the ones the band declared decided included -- goes
                                      through the split engine as well and its selection is compared with the one in place
                                      (adanerf_stats.guard_audited / .guard_audit_mismatch); a violated bound or a mismatch widens
                                      the band for the frames that follow.  adanerf_stats.rays_refined counts the re-evaluated
                                      rays.  Applies where the selection is fused into the sampling kernel (adaptive sampler,
                                      N <= 16, threshold > 0, 8 x 256 net); everywhere else this mode runs the split-precision
                                      engine alone. */
};
/* Sampling nets of another topology / encoding layout (run-time-shaped kernels): ADANERF_SAMPLING_FP32 runs the exact fp32
 * kernel, every other mode the split-precision one (same arithmetic as ADANERF_SAMPLING_SPLIT_FP16; no plain-fp16 pass);
 * with raySampleInput always the fp32 kernel. */

/* sample placement (config.ini rayMarchSampler[1]) */
enum {
  ADANERF_SAMPLER_ADAPTIVE = 0, /* FromClassifiedDepthAdaptive[NoDepthRange]: AdaNeRF top-N / threshold selection */
  ADANERF_SAMPLER_PDF = 1,      /* FromClassifiedDepth: DONeRF inverse-CDF sampling of sigmoid(oracle), fixed N samples
                                   per ray, classic sigma/delta compositing */
  ADANERF_SAMPLER_COARSE_FINE = 2 /* inFeatures [RayMarchFromPoses, RayMarchFromCoarse]: vanilla NeRF.  model0.onnx is a NeRF net
                                   too; numRaymarchSamples = [Nc, Nf]: Nc uniform depths, Nf more from the coarse weights,
                                   Nc + Nf samples through model1, classic compositing */
};

/* arithmetic of the shading MLP's MFMA path */
enum {
  ADANERF_PREC_BF16 = 0,  /* v_mfma_f32_32x32x16_bf16, fp32 accumulate */
  ADANERF_PREC_FP16 = 1,  /* v_mfma_f32_32x32x16_f16,  fp32 accumulate */
  ADANERF_PREC_FP32 = 2   /* v_mfma_f32_32x32x2_f32 (exact fp32; parity mode) */
};

/* adanerf_options.flags */
enum {
  ADANERF_FLAG_KEEP_ORACLE = 1, /* adanerf_render writes the raw oracle values [batch,128] to ADANERF_BUF_ORACLE and selects from
                                   there in a separate launch (debug / the reference's data flow); by default the selection runs
                                   in the sampling kernel's epilogue and that buffer is not written */
  ADANERF_FLAG_WAVE_SELECT = 2, /* selection by the wave-per-ray kernel (the only one for numRaymarchSamples > 16) even where the
                                   lane-pair selection applies; implies the separate launch */
  ADANERF_FLAG_NO_GUARD_CACHE = 4, /* ADANERF_SAMPLING_GUARDED: neither read nor write the calibration record next to the model
                                   (adanerf_guard_calibration_file); the band is measured at the first guarded frame */
  ADANERF_FLAG_GUARD_AUDIT_FILL = 8 /* ADANERF_SAMPLING_GUARDED (both shipped hosts set it): audit only as many decided rays per frame
                                   as the last, partly filled round of the refinement pass has room for (its grid x 128 rays per
                                   round; the undecided rays fix the number of rounds), so that the audit does not cost a round of
                                   its own (the 800 x 800 frame: 8 rounds instead of 9; an 80 000-ray share of it: 1 instead of 2)
                                   -- unless that room is less than a quarter of the frame's 1 / period quota, in which case it does
                                   take one more round.  The audited window moves through the frame's audit candidates from cycle
                                   to cycle, so every ray is audited at least once in 4 x period frames
                                   (adanerf_stats.guard_audited counts).  Without the flag every frame audits its full quota. */
};

typedef struct adanerf_ctx adanerf_ctx;

typedef struct adanerf_options {
  int32_t width;            /* image width  (Settings::width,  -s w h) */
  int32_t height;           /* image height (Settings::height)         */
  int32_t batch_rays;       /* <=0: whole frame; else min(batch_rays, rays) like Settings::batch_size (-bs) */
  int32_t device_id;        /* HIP device ordinal */
  int32_t precision;        /* ADANERF_PREC_* for the shading MLP */
  int32_t num_samples;      /* >0 overrides numRaymarchSamples from config.ini */
  float   threshold;        /* >=0 overrides adaptiveSamplingThreshold; <0 keeps config value */
  int32_t shard_rank;       /* image-strip shard of this context (multi-GPU); 0 */
  int32_t shard_world;      /* number of shards; 1 */
  int32_t strip_rows;       /* rows per strip for round-robin strip sharding; <=0 -> 8 */
  int32_t sampling_mode;    /* ADANERF_SAMPLING_* */
  int32_t flags;            /* ADANERF_FLAG_* (0 = defaults) */
  float   guard_eps;        /* ADANERF_SAMPLING_GUARDED: the band, in units of the raw network outputs; <= 0 -> the model's
                               calibration record if there is a current one, else calibrated for the loaded model at the first
                               guarded frame (adanerf_calibrate_guard, ADANERF_GUARD_CALIB_POSES poses) and recorded */
  float   guard_eps_pair;   /* ... the bound on the error of a (kept - candidate) difference; <= 0 -> calibrated like guard_eps when
                               that is, else 2 x guard_eps (what the bound on single values implies) */
  int32_t guard_audit_period; /* ... audit 1 / period of all rays per frame: 0 -> ADANERF_GUARD_AUDIT_PERIOD, < 0 -> no audit,
                               else a power of two <= 32 */
  int32_t reserved[1];
} adanerf_options;

/* Band of the guarded selection when calibration is impossible (non-finite outputs): about 2x the largest difference between
 * the plain-fp16 and the split-precision engine measured on the shipped networks (4.7e-3). */
#define ADANERF_GUARD_EPS_DEFAULT 1.0e-2f
/* The calibrated band is ADANERF_GUARD_CALIB_MARGIN x the largest difference found on the calibration rays, at least
 * ADANERF_GUARD_EPS_MIN. */
#define ADANERF_GUARD_CALIB_MARGIN 2.0f
#define ADANERF_GUARD_EPS_MIN 1.0e-3f
#define ADANERF_GUARD_CALIB_POSES 64     /* poses of the automatic calibration (x 4096 rays each) */
#define ADANERF_GUARD_AUDIT_PERIOD 16    /* default audit rate: every ray is audited once in 16 frames */

/* adanerf_info.guard_calib_source */
enum {
  ADANERF_GUARD_FROM_NONE = 0,        /* not calibrated yet */
  ADANERF_GUARD_FROM_OPTIONS = 1,     /* adanerf_options.guard_eps */
  ADANERF_GUARD_FROM_RECORD = 2,      /* the calibration record next to the model (adanerf_guard_calibration_file) */
  ADANERF_GUARD_FROM_CALIBRATION = 3, /* measured by this context (and recorded, if the directory is writable) */
  ADANERF_GUARD_FROM_MONITOR = 4      /* widened after a frame that violated a bound or failed the audit */
};

typedef struct adanerf_info {
  int32_t abi_version;
  int32_t width, height;
  int32_t rays_local;       /* rays rendered by this context (all of w*h when shard_world==1) */
  int32_t rays_local_max;   /* max over shards (gather payload size in rays) */
  int32_t batch_rays;
  int32_t n_in0, n_in1;     /* sampling / shading net input widths (90/90, or 30/90 for 2-2 NDC oracle) */
  int32_t num_samples;      /* N */
  float   threshold;
  int32_t dense;            /* threshold == 0: all 128 bins, no selection */
  int32_t use_ndc;
  int32_t precision;        /* the shading engine that runs: options.precision (every topology / encoding has a 16-bit and an
                               fp32 form since ABI 3) */
  int32_t compute_units;
  float   fov, focal;
  float   view_cell_center[3];
  float   view_cell_radius;
  float   depth_range[2];
  float   max_depth;
  int32_t sampler_mode;     /* ADANERF_SAMPLER_* (from rayMarchSampler[1]) */
  float   view_cell_size[3];/* dataset_info.txt view_cell_size (the viewer's camera speed: max(size / 2), camera.cpp:47) */
  int32_t num_samples_coarse; /* ADANERF_SAMPLER_COARSE_FINE: Nc (num_samples is then Nc + Nf); else 0 */
  float   guard_eps;          /* ADANERF_SAMPLING_GUARDED: the band in use (0 until it has been calibrated) */
  float   guard_eps_pair;     /* ... the bound on (kept - candidate) differences in use */
  int32_t guard_audit_period; /* ... 0: no audit */
  int32_t guard_calib_source; /* ... ADANERF_GUARD_FROM_* */
  int32_t guard_calib_poses;  /* ... poses behind a calibrated band (record or own calibration), else 0 */
  int32_t reserved[8];
} adanerf_info;

/* per-frame statistics: the fields the reference logs every 100 frames
 * (src/imagegenerator.cpp:379-393): Inference 1/2, fc1, fc2, rm, avg samples ppx */
typedef struct adanerf_stats {
  int64_t total_samples;      /* S summed over batches */
  int32_t rays;               /* rays rendered */
  int32_t batches;
  float   ms_total;           /* first launch -> last kernel end, HIP events on the context's stream */
  float   ms_sample_mlp;      /* ray gen + oracle PE + sampling MLP  ("fc1" + "Inference 1") */
  float   ms_compact;         /* selection + scan + expand           ("fc2", selection half) */
  float   ms_shade_mlp;       /* fused PE + shading MLP              ("fc2" feature half + "Inference 2") */
  float   ms_composite;       /* compositing                          ("rm") */
  int32_t shade_launches;     /* kernel launches behind ms_shade_mlp  */
  int32_t sample_launches;    /* kernel launches behind ms_sample_mlp */
  int32_t sampling_overflow;  /* rays (since create) whose oracle values were non-finite: an activation left the
                                 fp16 range of the split-precision engine -> use ADANERF_SAMPLING_FP32 */
  int32_t rays_refined;       /* ADANERF_SAMPLING_GUARDED: rays re-evaluated by the split-precision engine, summed like
                                 total_samples */
  float   guard_max_seen;     /* ... largest |fp16 - split| raw output seen since create, over ALL 128 outputs of every
                                 re-evaluated ray (undecided or audited): the assumption behind guard_eps, measured every frame */
  int32_t guard_violations;   /* ... re-evaluated rays (since create) where a measured error exceeded its bound: if this is
                                 not 0 the band was too narrow for this model; the library widens it for the frames that
                                 follow (guard_widened), frames before that may hold rays selected by the fp16 engine */
  int32_t guard_widened;      /* ... times (since create) the library widened the band after a frame reported violations or an audit
                                 mismatch: every later frame runs with ADANERF_GUARD_CALIB_MARGIN x the largest errors seen
                                 (adanerf_info.guard_eps / .guard_eps_pair are the bounds in force) */
  float   guard_pair_seen;    /* ... largest error of a (kept - candidate) difference seen since create (the assumption behind
                                 guard_eps_pair; 0 where the sampler transforms its values) */
  int32_t guard_audited;      /* ... rays (since create) that pass 1 had DECIDED and the audit re-evaluated anyway */
  int32_t guard_audit_mismatch; /* ... of those, rays whose exact selection differs from the one pass 1 left in place: must be 0
                                 while no bound is violated; not 0 -> the band is widened (guard_widened) */
  int32_t reserved[5];
} adanerf_stats;

/* ---- lifecycle ---------------------------------------------------------------------------- */

/* model_dir holds config.ini, dataset_info.txt, model0.onnx, model1.onnx (the format written by the
 * reference's src/export.py).  A trailing path separator is optional. */
int adanerf_create(const char* model_dir, const adanerf_options* opt, adanerf_ctx** out);
int adanerf_destroy(adanerf_ctx* ctx);
int adanerf_get_info(const adanerf_ctx* ctx, adanerf_info* info);
/* message of the last failing call on ctx; ctx == NULL reads the thread's last create() failure */
const char* adanerf_last_error(const adanerf_ctx* ctx);
/* ADANERF_ABI_VERSION the library was built with, and sizeof(adanerf_options / adanerf_info / adanerf_stats) as it sees them:
 * the handshake a binding does once after dlopen (structs are written whole, see ADANERF_ABI_VERSION). */
int adanerf_abi_version(void);
int adanerf_struct_sizes(int32_t sizes_out[3]);

/* ---- per frame ---------------------------------------------------------------------------- */

/* pos: camera position (world).  rot_c2w: row-major 3x3 camera-to-world rotation; camera looks
 * along -z, +y up (the convention of src/util/raygeneration.py:24-25).
 * ADANERF_PREC_BF16 contexts: the shading network's layers are scaled for sample positions the scene can produce with the camera in or
 * near its view cell (adanerf_create refuses a scene beyond that range).  A pose so far outside the cell that positions could leave the range
 * returns ADANERF_EUNSUPPORTED and leaves the previous camera in place -- render such poses with an fp16 / fp32 context.  NDC scenes
 * (useNDC) assume rays inside the frustum of the recorded cameras (positions in the NDC cube); that is not checked per pose. */
int adanerf_set_camera(adanerf_ctx* ctx, const float pos[3], const float rot_c2w[9]);

/* Renders this context's rays.  d_rgba8_out: [rays_local] uchar4 (A=255), row-major over the shard's
 * rows -- for shard_world==1 that is the whole image, pixel (x,y) at y*w+x.  d_rgb_f32_out: optional
 * [rays_local,3] fp32 unclamped colour (parity output).  Either may be NULL.  stats != NULL makes
 * the call synchronous and fills the per-stage timings. */
int adanerf_render(adanerf_ctx* ctx, void* d_rgba8_out, float* d_rgb_f32_out, adanerf_stats* stats);

/* Secondary outputs of the compositing step (reference: adaptive_raw2outputs / nerf_raw2outputs return them next to the
 * colour, src/nerf_raymarch_common.py:137-139 and :60-62; the viewer has no counterpart): until reset with NULLs, every
 * adanerf_render also fills
 *   d_depth_map [rays_local] fp32   sum_k w_k z_k, z = world depth of the sample as the sampler placed it (NDC depth for
 *                                   useNDC models).  The reference's NeRFOutputDepth is this value under NDC and
 *                                   depth_transform.from_world(.) of it otherwise (src/features.py:571-577); its disp_map:
 *                                   adanerf_set_disp_output
 *   d_acc_map   [rays_local] fp32   sum_k w_k (accumulated opacity)
 * with w_k the compositing weight of sample k (after accumulationMult).  Either may be NULL.  Caller-owned buffers. */
int adanerf_set_aux_outputs(adanerf_ctx* ctx, float* d_depth_map, float* d_acc_map);

/* The third secondary output of the reference's compositing (disp_map, src/nerf_raymarch_common.py:61 and :138):
 * d_disp_map [rays_local] fp32 = 1 / max(1e-10, depth_map / acc_map), filled by every adanerf_render until reset with NULL
 * (independent of adanerf_set_aux_outputs; a ray with acc_map == 0 gives NaN exactly as the reference's 0 / 0 does). */
int adanerf_set_disp_output(adanerf_ctx* ctx, float* d_disp_map);

/* De-interleaves the gathered shard payloads ([shard_world][rays_local_max] uchar4, rank-major)
 * into the full row-major image [h*w] uchar4. */
int adanerf_assemble_strips(adanerf_ctx* ctx, const void* d_gathered, void* d_image_out);

/* Single-process multi-GPU exchange (the counterpart of the RCCL gather for a host that owns all N contexts, e.g. the
 * `adanerf` CLI with --gpus N): copies `bytes` from d_src (a buffer of src's device, produced on src's stream) to d_dst
 * on dst's device -- hipMemcpyPeerAsync over xGMI, peer access enabled on first use -- ordered behind src's stream, and
 * makes dst's stream wait for it.  No host synchronisation.  src == dst or same device: a device-to-device copy. */
int adanerf_gather_to(adanerf_ctx* dst, void* d_dst, adanerf_ctx* src, const void* d_src, size_t bytes);

int adanerf_sync(adanerf_ctx* ctx);

/* Makes the context enqueue on a caller-owned HIP stream (hipStream_t, e.g. PyTorch's current stream)
 * instead of its own; NULL restores the context's stream.  The caller keeps the stream alive. */
int adanerf_set_stream(adanerf_ctx* ctx, void* hip_stream);

/* Asynchronous per-stage timing: with profiling on, every adanerf_render() records HIP events on the
 * context's stream without synchronising; adanerf_collect_stats() synchronises once and returns the
 * SUM over all frames rendered since the previous collect (stats->batches counts batches,
 * shade_launches / sample_launches the kernel launches behind the summed durations). */
int adanerf_set_profiling(adanerf_ctx* ctx, int32_t enabled);
int adanerf_collect_stats(adanerf_ctx* ctx, adanerf_stats* stats, int32_t* frames);

/* ---- stage-level entry points (mirror the reference launchers; used by the parity tests) ---- */

/* Oracle-net input features [n_rays, n_in0] fp32 = [PE(dir/|dir|) | PE(p)] and the per-ray record
 * d_rays_out [n_rays, 8] fp32 = (origin.xyz, 0, dir.xyz, 0) handed to the shading stage (sphere-exit
 * point and un-normalised world direction; NDC-space origin/direction when useNDC).  first_ray
 * indexes this context's local ray list.  Either output may be NULL. */
int adanerf_ray_features(adanerf_ctx* ctx, int32_t first_ray, int32_t n_rays,
                         float* d_features_out, float* d_rays_out);

/* Fused ray generation + PE + sampling MLP -> raw oracle values [n_rays,128] fp32 (+ ray records).  A model with
 * multiDepthFeatures = D < 128 fills bins D..127 with -1e30 ("absent": never selected). */
int adanerf_sample_mlp(adanerf_ctx* ctx, int32_t first_ray, int32_t n_rays,
                       float* d_oracle_out, float* d_rays_out);

/* Adaptive selection + deterministic compaction of arbitrary oracle values.
 *   d_oracle       [n_rays,128] fp32
 *   n_max, thr     N and threshold (thr > 0)
 *   d_ray_offsets  [n_rays] int32  exclusive prefix sum of counts (ray-major)
 *   d_ray_counts   [n_rays] int32  1..n_max
 *   d_sample_key   [>= n_rays*n_max] uint32  (local_ray << 7) | bin, ray-major, bins ascending
 *   d_sample_w     [>= n_rays*n_max] fp32    oracle value of the kept bin
 *   d_total        [1] int32       S
 * thr == 0 selects the dense mode (all 128 bins; n_max must be 128). */
int adanerf_compact(adanerf_ctx* ctx, const float* d_oracle, int32_t n_rays, int32_t n_max, float thr,
                    int32_t* d_ray_offsets, int32_t* d_ray_counts, uint32_t* d_sample_key,
                    float* d_sample_w, int32_t* d_total);

/* The guarded two-precision selection (ADANERF_SAMPLING_GUARDED) on caller-provided values, for tests: selects from
 * d_oracle_approx [n_rays,128] with the guard band (eps, eps_pair <= 0 -> 2 eps), lists the undecided rays plus the rays audited at
 * (audit_period, audit_phase) (audit_period <= 0: none; audit_fill_cap > 0: only as many of them as fill the last round of
 * audit_fill_cap rays, the window chosen by audit_cycle -- ADANERF_FLAG_GUARD_AUDIT_FILL), re-selects the undecided ones from d_oracle_exact [n_rays,128], compares
 * the audited decided ones, then compacts.  If the two bounds hold on every row the outputs equal adanerf_compact(d_oracle_exact, ...)
 * except that d_sample_w holds the approximate values on the rays that were not re-selected.  d_refined [1] int32: how many rays
 * were looked at again.  d_monitor (may be NULL) [5] uint32, ADDED to / maximised into (the caller zeroes it): [0] float bits of
 * the largest |approx - exact| over the re-evaluated rows, [1] rows that exceeded a bound, [2] float bits of the largest
 * (kept - candidate) difference error, [3] audited rows whose approximate selection differs from the exact one, [4] audited rows.
 * n_max <= 16, thr > 0. */
int adanerf_compact_guarded(adanerf_ctx* ctx, const float* d_oracle_approx, const float* d_oracle_exact, int32_t n_rays,
                            int32_t n_max, float thr, float eps, float eps_pair, int32_t audit_period, int32_t audit_phase,
                            int32_t audit_fill_cap, int32_t audit_cycle, int32_t* d_ray_offsets, int32_t* d_ray_counts, uint32_t* d_sample_key, float* d_sample_w,
                            int32_t* d_total, int32_t* d_refined, uint32_t* d_monitor);

/* Calibrates the band of ADANERF_SAMPLING_GUARDED for the loaded model: n_poses cameras drawn inside the view cell
 * (positions uniform in 90 % of it, any orientation; seeded), 64 x 64 rays covering the field of view each, through both the
 * plain-fp16 and the split-precision sampling network; *max_diff = the largest |difference| over all raw outputs, *max_pair_diff
 * (may be NULL) = the largest error of a (kept - candidate) difference under the context's N and threshold with the single-value
 * bound that max_diff gives (0 where the sampler transforms its values).
 * set != 0 also installs max(ADANERF_GUARD_CALIB_MARGIN * max_diff, ADANERF_GUARD_EPS_MIN) and ADANERF_GUARD_CALIB_MARGIN *
 * max_pair_diff as the context's bounds -- what a context created with guard_eps <= 0 does by itself before its first guarded
 * frame (ADANERF_GUARD_CALIB_POSES poses, seed 1) unless the model directory holds a current calibration record of at least that
 * many poses -- and writes the record (see adanerf_guard_calibration_file; a record measured over more poses is left alone).
 * Synchronous.  8 x 256 sampling networks only. */
int adanerf_calibrate_guard(adanerf_ctx* ctx, int32_t n_poses, uint32_t seed, int32_t set, float* max_diff, float* max_pair_diff);

/* Path of the calibration record of this context's (model, N, threshold): <model_dir>/guard_band.n<N>.t<threshold bits>.cal, or
 * under $ADANERF_GUARD_CACHE_DIR/<model hash>/ when that variable is set (read-only model directories).  A text file (key = value):
 * hash of model0.onnx, the encoding, the sampler transform, the revision of the fp16 engine, poses, seed, the measured maxima.  Read
 * at the first guarded frame of a context created with guard_eps <= 0; a record of another model / engine revision is ignored
 * and overwritten.  Delete it to force a new measurement.  buf may be NULL to query the length (return value: bytes needed
 * including the terminator, or a negative ADANERF_E* code). */
int adanerf_guard_calibration_file(const adanerf_ctx* ctx, char* buf, size_t buf_bytes);

/* Explicit shading-net input features [n_samples, n_in1] fp32 = [PE(x^) | PE(dir)] (parity/debug;
 * the render path never materialises them). */
int adanerf_shade_features(adanerf_ctx* ctx, const float* d_rays, const uint32_t* d_sample_key,
                           int32_t n_samples, float* d_features_out);

/* Fused PE + shading MLP over compacted samples -> raw [rgb, alpha] fp32 [n_samples,4].
 * d_total (device int32) bounds the work without a host sync; max_samples bounds the launch.
 * precision: ADANERF_PREC_*, or -1 for the context's. */
int adanerf_shade_mlp(adanerf_ctx* ctx, const float* d_rays, const uint32_t* d_sample_key,
                      const int32_t* d_total, int32_t max_samples, int32_t precision, float* d_raw_out);

/* As adanerf_shade_mlp with an explicit world depth per sample (d_sample_z [S] fp32, may be NULL -> the bin
 * centre of sample_key's bin): the inverse-CDF sampler places samples anywhere inside a bin. */
int adanerf_shade_mlp_z(adanerf_ctx* ctx, const float* d_rays, const uint32_t* d_sample_key, const float* d_sample_z,
                        const int32_t* d_total, int32_t max_samples, int32_t precision, float* d_raw_out);

/* Debug view of the sampling network (reference: copyResultSamplingNetwork(output, surf, batch_size, batch_offset,
 * width, 128), include/cuda/adanerf_cuda_kernels.cuh:20-21 -> samplesToImage, src/cuda/base_cuda_kernels.cu:487-528;
 * the viewer's 'O' key, src/inputhandler.cpp:76): per ray the three bins with the largest raw outputs, largest
 * first (equal values: lower bin first), written as RGBA8 ((0.5 + bin) / 128 * 255, truncated; A = 255).
 * d_rgba8 [n_rays] uchar4. */
int adanerf_copy_result_sampling_network(adanerf_ctx* ctx, const float* d_oracle, int32_t n_rays, void* d_rgba8);

/* Whole-frame version of the above for this rank's rays (ImageGenerator::inference with render_oracle set,
 * src/imagegenerator.cpp:316-317): sampling MLP, then the view; the shading half is skipped. */
int adanerf_render_oracle(adanerf_ctx* ctx, void* d_rgba8);

/* DONeRF sampler (reference: updateRayMarchFromPoses / samplePDF, include/cuda/adanerf_cuda_kernels.cuh:47-52,
 * src/cuda/base_cuda_kernels.cu:296-372; PyTorch: FromClassifiedDepth + nerf_sample_pdf): n samples per ray by
 * inverting the CDF of sigmoid(oracle) + 1e-5 at u = k/(n+1).  Outputs as adanerf_compact plus d_sample_z
 * [n_rays*n] world depths; counts are all n; sample_w is zero-filled. */
int adanerf_sample_pdf(adanerf_ctx* ctx, const float* d_oracle, int32_t n_rays, int32_t n, int32_t* d_ray_offsets,
                       int32_t* d_ray_counts, uint32_t* d_sample_key, float* d_sample_w, float* d_sample_z,
                       int32_t* d_total);

/* Vanilla NeRF, coarse pass (reference: updateRayMarchCoarse, include/cuda/adanerf_cuda_kernels.cuh:60-66; PyTorch:
 * RayMarchFromPoses over LinearlySpacedZNearZFar, src/features.py:380-480, src/nerf_raymarch_common.py:295-331): rays of
 * [batch_offset, batch_offset + n_rays) from the camera position (d_rays [n,8]) and Nc samples per ray at the model's uniform
 * depth table (d_sample_key = ray << 7 | k, counts all Nc).  ADANERF_SAMPLER_COARSE_FINE contexts only. */
int adanerf_sample_uniform(adanerf_ctx* ctx, int32_t batch_offset, int32_t n_rays, float* d_rays, int32_t* d_ray_offsets,
                           int32_t* d_ray_counts, uint32_t* d_sample_key, int32_t* d_total);

/* The coarse network (model0.onnx, a NeRF net in this mode) on samples made by adanerf_sample_uniform: as adanerf_shade_mlp. */
int adanerf_shade_mlp_coarse(adanerf_ctx* ctx, const float* d_rays, const uint32_t* d_sample_key, const int32_t* d_total,
                             int32_t max_samples, int32_t precision, float* d_raw);

/* Vanilla NeRF, fine sampler (reference: updateRayMarchFromCoarse incl. nerf_raw_2_output_weights, adanerf_cuda_kernels.cuh:68-74,
 * src/cuda/coarse_cuda_kernels.cu:29-383; PyTorch: RayMarchFromCoarse.batch, src/features.py:640-672): the coarse raw outputs
 * [n_rays*Nc,4] -> nerf_raw2outputs weights -> nerf_sample_pdf over the interval mid-points with weights[1:-1], u =
 * linspace(0,1,Nf) -> merged with the coarse depths, ascending.  d_sample_z [n_rays*(Nc+Nf)] world depths for
 * adanerf_shade_mlp_z; counts are all Nc + Nf. */
int adanerf_sample_from_coarse(adanerf_ctx* ctx, const float* d_raw_coarse, const float* d_rays, int32_t n_rays, int32_t* d_ray_offsets,
                               int32_t* d_ray_counts, uint32_t* d_sample_key, float* d_sample_z, int32_t* d_total);

/* Classic NeRF compositing over a fixed n samples per ray (reference: copyResultRaymarch / nerf_raw_2_output,
 * adanerf_cuda_kernels.cuh:23-24; PyTorch nerf_raw2outputs): alpha = 1 - exp(-relu(raw.a) * dz * |dir|). */
int adanerf_composite_classic(adanerf_ctx* ctx, const float* d_raw, const float* d_sample_z, const float* d_rays,
                              int32_t n_rays, int32_t n, float* d_rgb_out, void* d_rgba8_out);

/* Per-ray front-to-back compositing: c = sigmoid(raw.rgb), a = sigmoid(raw.a) * w. */
int adanerf_composite(adanerf_ctx* ctx, const float* d_raw, const float* d_sample_w,
                      const int32_t* d_ray_offsets, const int32_t* d_ray_counts, int32_t n_rays,
                      float* d_rgb_out, void* d_rgba8_out);

/* ---- measurement hook (bench.py's roofline) ---- */

/* What dense 16-bit MFMA rate does the context's device sustain on live operands, and at which clock?  Runs register-only loops of
 * v_mfma_f32_32x32x16_bf16 (f16 != 0: _f16) on every CU for about target_ms milliseconds -- two accumulator chains per wave, one wave
 * per SIMD, the shading kernel's form -- and returns the achieved TFLOP/s and the effective shader clock (MHz, s_memtime against
 * s_memrealtime).  operands: 0 all-zero bits, 1 constant small values, 2 random values that change with every MFMA, 3 as 2 with a
 * post-ReLU-like B operand (half of its elements 0, the rest positive).  The part runs to a power budget: the same stream reaches
 * ~2.4 GHz on zeros and ~1.7 GHz on random operands, so a kernel's fraction of the nominal 2.5 PFLOP/s and its fraction of what
 * the silicon sustains on its kind of data differ (profiles/r04_mfma_peak_operands_clock.log).  Synchronous; nothing else should
 * run on the device meanwhile. */
int adanerf_probe_mfma(adanerf_ctx* ctx, int32_t operands, int32_t f16, float target_ms, float* tflops, float* clock_mhz);

/* ---- device memory plumbing for callers without a HIP runtime of their own ---- */
int adanerf_malloc(adanerf_ctx* ctx, size_t bytes, void** d_out);
int adanerf_free(adanerf_ctx* ctx, void* d_ptr);
int adanerf_memcpy_h2d(adanerf_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int adanerf_memcpy_d2h(adanerf_ctx* ctx, void* dst, const void* d_src, size_t bytes);

/* Internal buffers of the last adanerf_render batch (device pointers, valid until the next call). */
enum {
  ADANERF_BUF_RAYS = 0,        /* [batch,8] fp32 */
  ADANERF_BUF_ORACLE = 1,      /* [batch,128] fp32 */
  ADANERF_BUF_RAY_OFFSETS = 2, /* [batch] int32 */
  ADANERF_BUF_RAY_COUNTS = 3,  /* [batch] int32 */
  ADANERF_BUF_SAMPLE_KEY = 4,  /* [S] uint32; not written in dense mode (threshold 0): sample i is ray i >> 7, bin i & 127 */
  ADANERF_BUF_SAMPLE_W = 5,    /* [S] fp32; not written in dense mode: the kept values are ADANERF_BUF_ORACLE itself */
  ADANERF_BUF_RAW = 6,         /* [S,4] fp32 */
  ADANERF_BUF_TOTAL = 7,       /* [1] int32 */
  ADANERF_BUF_SAMPLE_Z = 8,    /* [S] fp32 (ADANERF_SAMPLER_PDF / _COARSE_FINE) */
  ADANERF_BUF_RAW_COARSE = 9   /* [batch*Nc,4] fp32 (ADANERF_SAMPLER_COARSE_FINE) */
};
int adanerf_get_buffer(adanerf_ctx* ctx, int32_t which, void** d_out, size_t* bytes_out);

/* ---- host-only inspection (no device needed; used by the CPU test-suite) ---- */

/* Parses and validates config.ini / dataset_info.txt exactly as adanerf_create does and fills
 * `info` (compute_units = 0).  Fails with the same codes/messages as adanerf_create. */
int adanerf_host_parse_model(const char* model_dir, const adanerf_options* opt, adanerf_info* info);

/* Packs net 0 (sampling) or net 1 (shading) of model_dir into MFMA A-fragment order for
 * `precision` (ADANERF_PREC_*, or 3 = the sampling net's fp16 hi/lo' split pairs).  Two-call pattern: pass NULL outputs to query sizes.
 *   weights_out  packed 16-byte fragments        (*weights_bytes)
 *   bias_out     packed bias blocks, fp32        (*bias_floats)
 *   layer_out    per layer {w_off (16-B units), b_off (floats), slots per lane-half, 32-row tiles}
 *                as int32[4] each                (*n_layers); a sampling net with raySampleInput = A > 0 has one more
 *                record {w_off of layer 0's K-major block for the A extra points, A, slots per point, tiles}; a shading net in
 *                ADANERF_PREC_BF16 has one more record {alpha exponent, rgb exponent, 0, -1}: it is packed SCALED (every ReLU
 *                layer's weights and bias carry a power of two that keeps its activations <= 1, so the kernels' ReLU is a clamped
 *                conversion), and its alpha / rgb outputs x 2^exponent are the network's own.
 *                (precision 4, net 1 only: the bf16 blob without that scaling, for tests -- no kernel consumes it.)
 * The shading net packs in every precision for every topology.  A sampling net other than 8 x 256 with a 10-4 / 2-2 encoding
 * packs for ADANERF_PREC_FP32 (run-time-shaped fp32 kernel) and, without raySampleInput, as split pairs (3: run-time-shaped
 * split-precision kernel); the plain 16-bit precisions -- and the split pairs with raySampleInput -- return ADANERF_EIO with a message. */
int adanerf_host_pack_weights(const char* model_dir, int32_t net, int32_t precision, void* weights_out,
                              size_t* weights_bytes, float* bias_out, size_t* bias_floats, int32_t* layer_out,
                              int32_t* n_layers);

/* 128-entry world-depth table the shading stage indexes by bin (and dense t-values) for model_dir. */
int adanerf_host_depth_table(const char* model_dir, const adanerf_options* opt, float* ztab128);

#ifdef __cplusplus
}
#endif
#endif /* ADANERF_HIP_H */
