#include "neuralrenderer.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

NeuralRenderer::~NeuralRenderer() {
  for (size_t i = 0; i < peers.size(); ++i) {
    if (peers[i]) {
      if (i + 1 < d_payload.size() && d_payload[i + 1]) adanerf_free(peers[i], d_payload[i + 1]);
      adanerf_destroy(peers[i]);
    }
  }
  if (ctx) {
    if (d_gathered) adanerf_free(ctx, d_gathered);
    if (d_frame) adanerf_free(ctx, d_frame);
    adanerf_destroy(ctx);
  }
}

int NeuralRenderer::batchesPerFrame() const {
  const long long rays = static_cast<long long>(info_.width) * info_.height, b = info_.batch_rays > 0 ? info_.batch_rays : rays;
  return static_cast<int>((rays + b - 1) / (b > 0 ? b : 1));
}

static adanerf_options options_of(const Settings& settings) {
  adanerf_options opt;
  std::memset(&opt, 0, sizeof(opt));
  opt.width = static_cast<int32_t>(settings.width);
  opt.height = static_cast<int32_t>(settings.height);
  opt.batch_rays = static_cast<int32_t>(settings.batch_size);
  opt.num_samples = settings.num_samples;
  opt.threshold = settings.threshold;
  opt.sampling_mode = (settings.sampling == "split" || settings.sampling == "auto") ? ADANERF_SAMPLING_SPLIT_FP16 : (settings.sampling == "fp32" ? ADANERF_SAMPLING_FP32
                      : (settings.sampling == "fp16" ? ADANERF_SAMPLING_FP16 : ADANERF_SAMPLING_GUARDED));
  opt.shard_world = 1;
  return opt;
}

bool NeuralRenderer::initHostOnly() {
  const adanerf_options opt = options_of(settings);
  if (adanerf_host_parse_model(settings.model_path.c_str(), &opt, &info_) != ADANERF_OK) {
    err = adanerf_last_error(nullptr);
    return false;
  }
  camera.setPosition(info_.view_cell_center);
  camera.setViewCell(info_.view_cell_size);
  render_oracle = settings.render_oracle;
  return true;
}

// --sampling auto: the default rule measured on the workload at hand -- a few frames of this model / frame size / threshold in the
// split mode (exact by construction) and in the guarded mode (the same frames while its measured band holds), camera at the view-cell centre;
// guarded only if it is at least 8 % faster here.  One context at a time, on GPU 0.
static std::string measure_sampling_mode(const adanerf_options& base, const Settings& settings, const Camera& camera) {
  double fps[2] = {0.0, 0.0};
  const int modes[2] = {ADANERF_SAMPLING_SPLIT_FP16, ADANERF_SAMPLING_GUARDED};
  for (int k = 0; k < 2; ++k) {
    adanerf_options opt = base;
    opt.sampling_mode = modes[k];
    adanerf_ctx* c = nullptr;
    if (adanerf_create(settings.model_path.c_str(), &opt, &c) != ADANERF_OK) break;      // no guarded mode for this model: split
    adanerf_info info;
    void* frame = nullptr;
    float rot[9];
    camera.getRotMatrix(rot);
    bool ok = adanerf_get_info(c, &info) == ADANERF_OK && adanerf_malloc(c, static_cast<size_t>(info.rays_local_max) * 4, &frame) == ADANERF_OK &&
              adanerf_set_camera(c, info.view_cell_center, rot) == ADANERF_OK;
    const int warm = 2, frames = 6;
    for (int i = 0; ok && i < warm; ++i) ok = adanerf_render(c, frame, nullptr, nullptr) == ADANERF_OK;
    ok = ok && adanerf_sync(c) == ADANERF_OK;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; ok && i < frames; ++i) ok = adanerf_render(c, frame, nullptr, nullptr) == ADANERF_OK;
    ok = ok && adanerf_sync(c) == ADANERF_OK;
    if (ok) fps[k] = frames / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (frame) adanerf_free(c, frame);
    adanerf_destroy(c);
    if (!ok) break;
  }
  const bool guarded = fps[0] > 0.0 && fps[1] >= 1.08 * fps[0];
  std::cout << "--sampling auto: split " << fps[0] << " frames/s, guarded " << fps[1] << " frames/s -> " << (guarded ? "guarded" : "split")
            << " (guarded only if >= 8 % faster on this workload)" << std::endl;
  return guarded ? "guarded" : "split";
}

bool NeuralRenderer::init() {
  std::cout << "Model Path: " << settings.model_path << std::endl;
  adanerf_options opt;
  std::memset(&opt, 0, sizeof(opt));
  opt.width = static_cast<int32_t>(settings.width);
  opt.height = static_cast<int32_t>(settings.height);
  opt.batch_rays = static_cast<int32_t>(settings.batch_size);
  opt.device_id = 0;
  opt.precision = settings.precision == "fp32" ? ADANERF_PREC_FP32 : (settings.precision == "fp16" ? ADANERF_PREC_FP16 : ADANERF_PREC_BF16);
  opt.num_samples = settings.num_samples;
  opt.threshold = settings.threshold;
  opt.shard_world = 1;
  opt.strip_rows = 8;
  opt.flags |= ADANERF_FLAG_GUARD_AUDIT_FILL;
  if (settings.sampling == "auto") settings.sampling = settings.precision == "fp32" ? std::string("split") : measure_sampling_mode(opt, settings, camera);
  opt.sampling_mode = settings.sampling == "split" ? ADANERF_SAMPLING_SPLIT_FP16 : (settings.sampling == "fp32" ? ADANERF_SAMPLING_FP32
                      : (settings.sampling == "fp16" ? ADANERF_SAMPLING_FP16 : ADANERF_SAMPLING_GUARDED));
  // --gpus N: rows are cut into strips, strip s belongs to (virtual) rank s % world; largest strip height <= 8 rows that
  // gives every rank the same number of strips.  --sub-shares P: every GPU hosts P virtual ranks (rank / P = its GPU), i.e. renders its
  // share of the frame as P concurrent sub-shares on P contexts / streams -- one sub-share's kernels take the CUs the other's tails
  // leave idle, the frame's latency does not grow (bench.py --sub-shares); default 2 on several GPUs when the rows split evenly
  auto even_split = [&](int ranks) {
    for (int sr = 8; sr >= 1; --sr)
      if (opt.height % sr == 0 && (opt.height / sr) % ranks == 0) return true;
    return false;
  };
  const int parts = settings.sub_shares > 0 ? settings.sub_shares : (settings.gpus > 1 && even_split(2 * settings.gpus) ? 2 : 1);
  const int world = settings.gpus * parts;
  if (world > 1 && settings.render_oracle) {
    err = "--oracle renders one context's rays; use it with --gpus 1";
    return false;
  }
  int strip_rows = 8;
  for (int sr = 8; sr >= 1; --sr)
    if (opt.height % sr == 0 && (opt.height / sr) % world == 0) {
      strip_rows = sr;
      break;
    }
  opt.shard_world = world;
  // the guarded selection's audit fills the last round of its refinement pass instead of adding one (adanerf_hip.h); as in renderer.py
  opt.flags |= ADANERF_FLAG_GUARD_AUDIT_FILL;
  opt.strip_rows = strip_rows;
  for (int rank = 0; rank < world; ++rank) {
    opt.shard_rank = rank;
    opt.device_id = settings.same_device ? 0 : rank / parts;
    adanerf_ctx* c = nullptr;
    if (adanerf_create(settings.model_path.c_str(), &opt, &c) != ADANERF_OK) {
      err = adanerf_last_error(nullptr);
      return false;
    }
    if (rank == 0) ctx = c;
    else peers.push_back(c);
  }
  adanerf_get_info(ctx, &info_);
  if (adanerf_malloc(ctx, static_cast<size_t>(info_.width) * info_.height * 4, &d_frame) != ADANERF_OK) {
    err = adanerf_last_error(ctx);
    return false;
  }
  if (world > 1) {
    const size_t payload = static_cast<size_t>(info_.rays_local_max) * 4;
    if (adanerf_malloc(ctx, payload * world, &d_gathered) != ADANERF_OK) {
      err = adanerf_last_error(ctx);
      return false;
    }
    d_payload.assign(world, nullptr);
    d_payload[0] = d_gathered;     // rank 0 renders straight into its slot
    for (int rank = 1; rank < world; ++rank)
      if (adanerf_malloc(peers[rank - 1], payload, &d_payload[rank]) != ADANERF_OK) {
        err = adanerf_last_error(peers[rank - 1]);
        return false;
      }
  }
  camera.setPosition(info_.view_cell_center);   // Camera::init: pos = view-cell centre (camera.cpp:49)
  camera.setViewCell(info_.view_cell_size);
  if (world > 1 && adanerf_set_profiling(ctx, 1) != ADANERF_OK) {
    err = adanerf_last_error(ctx);
    return false;
  }
  render_oracle = settings.render_oracle;
  return true;
}

bool NeuralRenderer::render() {
  float rot[9];
  camera.getRotMatrix(rot);
  if (adanerf_set_camera(ctx, camera.getPosition(), rot) != ADANERF_OK) {
    err = adanerf_last_error(ctx);
    return false;
  }
  if (render_oracle) {   // imagegenerator.cpp:316-317: sampling network only, top-3 bins as RGB
    if (adanerf_render_oracle(ctx, d_frame) != ADANERF_OK || adanerf_sync(ctx) != ADANERF_OK) {
      err = adanerf_last_error(ctx);
      return false;
    }
    sample_count++;
    return settings.write_images ? writeImageToFile() : true;
  }
  adanerf_stats st;
  std::memset(&st, 0, sizeof(st));
  if (!peers.empty()) {
    // every GPU renders its strips (launches are asynchronous: the GPUs run concurrently), each payload is copied to
    // the display GPU over xGMI behind its render, and rank 0 de-interleaves; one host sync at the end of the frame.
    // Rank 0 is enqueued first so that its stream does not wait for the peers before it starts.
    const size_t payload = static_cast<size_t>(info_.rays_local_max) * 4;
    if (adanerf_render(ctx, d_payload[0], nullptr, nullptr) != ADANERF_OK) {
      err = adanerf_last_error(ctx);
      return false;
    }
    for (size_t i = 0; i < peers.size(); ++i) {
      if (adanerf_set_camera(peers[i], camera.getPosition(), rot) != ADANERF_OK ||
          adanerf_render(peers[i], d_payload[i + 1], nullptr, nullptr) != ADANERF_OK ||
          adanerf_gather_to(ctx, static_cast<char*>(d_gathered) + (i + 1) * payload, peers[i], d_payload[i + 1], payload) != ADANERF_OK) {
        err = adanerf_last_error(peers[i]);
        return false;
      }
    }
    if (adanerf_assemble_strips(ctx, d_gathered, d_frame) != ADANERF_OK || adanerf_sync(ctx) != ADANERF_OK) {
      err = adanerf_last_error(ctx);
      return false;
    }
    sample_count++;
    if (sample_count % logging_interval == 0) {
      // rank 0's share, accumulated on its stream by the profiling API (no per-frame sync); samples scaled to the frame
      int32_t frames = 0;
      if (adanerf_collect_stats(ctx, &st, &frames) == ADANERF_OK && frames > 0) {
        const double f = frames, world = static_cast<double>(peers.size() + 1);
        std::cout << "Inference 1:" << st.ms_sample_mlp / f << ", 2:" << st.ms_shade_mlp / f << " | fc1: 0"
                  << ", fc2: " << st.ms_compact / f << ", rm: " << st.ms_composite / f
                  << ", avg samples ppx: " << st.total_samples * world / f / settings.total_size
                  << " (total: " << static_cast<long long>(st.total_samples * world / f) << ")"
                  << ", frames: " << sample_count << ", gpus: " << peers.size() + 1 << " (stage times: GPU 0's share)" << std::endl;
      }
    }
    return settings.write_images ? writeImageToFile() : true;
  }
  if (adanerf_render(ctx, d_frame, nullptr, &st) != ADANERF_OK) {
    err = adanerf_last_error(ctx);
    return false;
  }
  if (st.guard_widened > guard_widened_seen) {      // the guarded selection's monitor saw its band violated and widened it (adanerf_hip.h)
    guard_widened_seen = st.guard_widened;
    adanerf_info now;
    if (adanerf_get_info(ctx, &now) == ADANERF_OK)
      std::cout << "guarded sampling: " << st.guard_violations << " re-evaluated rays exceeded a measured bound (largest output error " << st.guard_max_seen
                << ", largest pair error " << st.guard_pair_seen << "), audit: " << st.guard_audit_mismatch << " of " << st.guard_audited
                << " decided rays selected differently; bounds widened to " << now.guard_eps << " / " << now.guard_eps_pair
                << " for the following frames" << std::endl;
  }
  s_inference1 += st.ms_sample_mlp;
  s_inference2 += st.ms_shade_mlp;
  s_fc2 += st.ms_compact;
  s_rm += st.ms_composite;
  s_total += st.ms_total;
  s_num_total_samples += st.total_samples;
  sample_count++;
  if (sample_count % logging_interval == 0) {
    // same fields as the reference's log line; fc1 (ray/oracle features) is fused into "Inference 1"
    std::cout << "Inference 1:" << s_inference1 / logging_interval << ", 2:" << s_inference2 / logging_interval << " | fc1: 0"
              << ", fc2: " << s_fc2 / logging_interval << ", rm: " << s_rm / logging_interval
              << ", avg samples ppx: " << s_num_total_samples / static_cast<double>(logging_interval) / settings.total_size
              << " (total: " << s_num_total_samples / logging_interval << ")"
              << ", frames: " << sample_count << ", frame ms: " << s_total / logging_interval << std::endl;
    s_inference1 = s_inference2 = s_fc2 = s_rm = s_total = 0;
    s_num_total_samples = 0;
  }
  if (settings.write_images) return writeImageToFile();
  return true;
}

bool NeuralRenderer::writeImageToFile() {
  const int w = static_cast<int>(settings.width), h = static_cast<int>(settings.height);
  std::vector<unsigned char> image(static_cast<size_t>(w) * h * 4);
  if (adanerf_memcpy_d2h(ctx, image.data(), d_frame, image.size()) != ADANERF_OK) {
    err = adanerf_last_error(ctx);
    return false;
  }
  std::string path = settings.model_path;
  if (!path.empty() && path.back() != '/') path += '/';
  path += "out.bmp";
  std::ofstream fout(path, std::ios::binary);
  if (!fout) {
    err = "cannot write " + path;
    return false;
  }
  // 24-bit BMP, bottom-up rows, BGR, rows padded to 4 bytes
  const int row_bytes = (w * 3 + 3) & ~3;
  const uint32_t data_size = static_cast<uint32_t>(row_bytes) * h, off = 54, file_size = off + data_size;
  unsigned char hdr[54] = {'B', 'M'};
  auto put32 = [&](int at, uint32_t v) {
    hdr[at] = v & 255;
    hdr[at + 1] = (v >> 8) & 255;
    hdr[at + 2] = (v >> 16) & 255;
    hdr[at + 3] = (v >> 24) & 255;
  };
  put32(2, file_size);
  put32(10, off);
  put32(14, 40);
  put32(18, static_cast<uint32_t>(w));
  put32(22, static_cast<uint32_t>(h));
  hdr[26] = 1;
  hdr[28] = 24;
  put32(34, data_size);
  fout.write(reinterpret_cast<char*>(hdr), 54);
  std::vector<unsigned char> row(row_bytes, 0);
  for (int y = h - 1; y >= 0; --y) {
    for (int x = 0; x < w; ++x) {
      const unsigned char* px = &image[(static_cast<size_t>(y) * w + x) * 4];
      row[3 * x + 0] = px[2];
      row[3 * x + 1] = px[1];
      row[3 * x + 2] = px[0];
    }
    fout.write(reinterpret_cast<char*>(row.data()), row_bytes);
  }
  return true;
}
