// Command-line settings of the headless `adanerf` host.  Same arguments and batch-size semantics as the
// reference viewer (adanerf_real_time_viewer/src/main.cpp:19-50, src/settings.cpp:15-47).
#pragma once
#include <string>

class Settings {
 public:
  std::string model_path = "sample/";
  unsigned int width = 800, height = 800, total_size = 640000;
  unsigned int window_width = 800, window_height = 800;   // accepted, unused (headless)
  unsigned int batch_size = 640000;
  bool write_images = false;
  bool is_debug = false;        // -d: accepted, headless is always "debug" (no GL interop)
  // headless extras (not in the reference)
  int frames = 100;             // --frames: frames to render before exiting
  std::string precision = "bf16";
  std::string sampling;               // --sampling auto|guarded|split|fp32|fp16: arithmetic of the sampling network (ADANERF_SAMPLING_*; auto = the default rule measured on this workload at start-up); the
                                      // viewer itself has one (TensorRT kFP16 = "fp16"); "guarded" keeps the exact engine's selections while its
                                      // measured band holds.  Not given: split (exact by construction on every ray; DESIGN 1 has the rule)
  float yaw = -80.f, pitch = 0.f;   // Camera::init defaults (camera.cpp:90-91)
  int num_samples = 0;
  float threshold = -1.f;
  int gpus = 1;                 // --gpus N: one context per GPU in this process, strips exchanged with peer copies over xGMI
  bool same_device = false;     // --same-device: all N contexts on device 0 (exercises the N-GPU path on a 1-GPU box)
  int sub_shares = 0;           // --sub-shares P: every GPU renders its share of a frame as P concurrent sub-shares (contexts / streams);
                                // 0: 2 with --gpus N > 1 when the rows split evenly over 2 N, else 1
  std::string script;           // --script FILE: replay input events, one line per frame (inputhandler.h)
  bool log_camera = false;      // --log-camera: print position / yaw / pitch / view per frame
  bool dry_run = false;         // --dry-run: replay the script without a device (no rendering)
  bool render_oracle = false;   // --oracle: the viewer's 'O' key (inputhandler.cpp:76), sampling-network debug view

  // returns false and fills err on a malformed command line
  bool init(int argc, char** argv, std::string* err);
  static const char* usage();
};
