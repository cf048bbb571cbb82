// Headless NeuralRenderer: the viewer's init()/render() pair (include/neuralrenderer.h:53-55) over the
// C ABI of libadanerf_hip.so.  Owns the device framebuffer the viewer's BufferManager would own.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/adanerf_hip.h"
#include "camera.h"
#include "settings.h"

class NeuralRenderer {
 public:
  NeuralRenderer(Settings& settings, Camera& camera) : settings(settings), camera(camera) {}
  ~NeuralRenderer();

  bool init();
  bool initHostOnly();               // --dry-run: parse the model directory on the host only (no device), for input replay
  bool render();                     // one frame; logs every 100 frames like imagegenerator.cpp:379-393
  void switchRenderOracle() { render_oracle = !render_oracle; }   // neuralrenderer.h: the 'O' key toggle
  bool renderingOracle() const { return render_oracle; }
  int batchesPerFrame() const;       // ceil(rays / batch_rays)
  bool writeImageToFile();           // out.bmp in the model directory (neuralrenderer.cpp:184-222)
  const adanerf_info& info() const { return info_; }
  const std::string& error() const { return err; }

 private:
  Settings& settings;
  Camera& camera;
  adanerf_ctx* ctx = nullptr;        // rank 0: display GPU, owns the frame
  std::vector<adanerf_ctx*> peers;   // ranks 1 .. N-1 (--gpus N), one per GPU, same process
  std::vector<void*> d_payload;      // per rank: uchar4 [rays_local_max] on that rank's GPU
  void* d_gathered = nullptr;        // rank 0: uchar4 [N][rays_local_max]
  adanerf_info info_{};
  void* d_frame = nullptr;           // uchar4 [h*w]
  std::string err;
  bool render_oracle = false;
  // 100-frame running sums
  int logging_interval = 100, sample_count = 0;
  double s_inference1 = 0, s_inference2 = 0, s_fc2 = 0, s_rm = 0, s_total = 0;
  int guard_widened_seen = 0;        // adanerf_stats.guard_widened already reported on the console
  long long s_num_total_samples = 0;
};
