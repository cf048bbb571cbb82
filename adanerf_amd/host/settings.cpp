#include "settings.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

const char* Settings::usage() {
  return "Usage: adanerf [modelPath] [-s|--size W H] [-ws|--windowSize W H] [-bs|--batchSize N]\n"
         "               [-nb|--numberOfBatches N] [-w|--writeImages] [-d|--debug]\n"
         "               [--frames N] [--precision bf16|fp16|fp32] [--sampling auto|guarded|split|fp32|fp16]\n"
         "               [--yaw DEG] [--pitch DEG]\n"
         "               [--samples N] [--threshold T] [--oracle]\n"
         "               [--script FILE] [--log-camera] [--dry-run]     input replay: one line of events per frame\n"
         "               [--gpus N] [--same-device] [--sub-shares P]\n";
}

bool Settings::init(int argc, char** argv, std::string* err) {
  bool bs_used = false, ws_used = false, model_set = false;
  int batch_arg = -1;
  unsigned int n_batches = 1;
  auto need = [&](int i, int n) {
    if (i + n >= argc) {
      *err = std::string("missing value after ") + argv[i];
      return false;
    }
    return true;
  };
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "-s" || a == "--size") {
      if (!need(i, 2)) return false;
      width = static_cast<unsigned>(std::atoi(argv[i + 1]));
      height = static_cast<unsigned>(std::atoi(argv[i + 2]));
      i += 2;
    } else if (a == "-ws" || a == "--windowSize") {
      if (!need(i, 2)) return false;
      window_width = static_cast<unsigned>(std::atoi(argv[i + 1]));
      window_height = static_cast<unsigned>(std::atoi(argv[i + 2]));
      ws_used = true;
      i += 2;
    } else if (a == "-bs" || a == "--batchSize") {
      if (!need(i, 1)) return false;
      batch_arg = std::atoi(argv[++i]);
      bs_used = true;
    } else if (a == "-nb" || a == "--numberOfBatches") {
      if (!need(i, 1)) return false;
      n_batches = static_cast<unsigned>(std::max(1, std::atoi(argv[++i])));
    } else if (a == "-w" || a == "--writeImages") {
      write_images = true;
    } else if (a == "-d" || a == "--debug") {
      is_debug = true;
    } else if (a == "--frames") {
      if (!need(i, 1)) return false;
      frames = std::atoi(argv[++i]);
    } else if (a == "--precision") {
      if (!need(i, 1)) return false;
      precision = argv[++i];
    } else if (a == "--sampling") {
      if (!need(i, 1)) return false;
      sampling = argv[++i];
      if (sampling != "auto" && sampling != "guarded" && sampling != "split" && sampling != "fp32" && sampling != "fp16") {
        *err = "--sampling must be auto, guarded, split, fp32 or fp16";
        return false;
      }
    } else if (a == "--yaw") {
      if (!need(i, 1)) return false;
      yaw = static_cast<float>(std::atof(argv[++i]));
    } else if (a == "--pitch") {
      if (!need(i, 1)) return false;
      pitch = static_cast<float>(std::atof(argv[++i]));
    } else if (a == "--samples") {
      if (!need(i, 1)) return false;
      num_samples = std::atoi(argv[++i]);
    } else if (a == "--threshold") {
      if (!need(i, 1)) return false;
      threshold = static_cast<float>(std::atof(argv[++i]));
    } else if (a == "--gpus") {
      if (!need(i, 1)) return false;
      gpus = std::max(1, std::atoi(argv[++i]));
    } else if (a == "--same-device") {
      same_device = true;
    } else if (a == "--sub-shares") {
      if (!need(i, 1)) return false;
      sub_shares = std::max(1, std::min(4, std::atoi(argv[++i])));
    } else if (a == "--oracle") {
      render_oracle = true;
    } else if (a == "--script") {
      if (!need(i, 1)) return false;
      script = argv[++i];
    } else if (a == "--log-camera") {
      log_camera = true;
    } else if (a == "--dry-run") {
      dry_run = true;
    } else if (a == "-h" || a == "--help") {
      *err = usage();
      return false;
    } else if (!a.empty() && a[0] != '-' && !model_set) {
      model_path = a;
      model_set = true;
    } else {
      *err = "unknown argument " + a;
      return false;
    }
  }
  if (width == 0 || height == 0) {
    *err = "size must be positive";
    return false;
  }
  total_size = width * height;
  if (sampling.empty()) sampling = "split";      // exact by construction; guarded / fp16 are opt-in (the default rule)
  if (!ws_used) {
    window_width = width;
    window_height = height;
  }
  // settings.cpp:38-46
  batch_size = static_cast<unsigned>(std::ceil(total_size / static_cast<float>(n_batches)));
  if (bs_used) batch_size = batch_arg <= 0 ? total_size : std::min(static_cast<unsigned>(batch_arg), total_size);
  return true;
}
