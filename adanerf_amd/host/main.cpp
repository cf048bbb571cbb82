// Headless `adanerf`: the reference viewer's entry point (adanerf_real_time_viewer/src/main.cpp:15-96)
// without the window -- parse the same arguments, load the model directory, render frames, log every
// 100 frames, optionally write out.bmp.  Pure host C++ over the C ABI (no HIP in this file).
#include <chrono>
#include <iostream>

#include "camera.h"
#include "neuralrenderer.h"
#include "settings.h"

int main(int argc, char* argv[]) {
  Settings settings;
  std::string err;
  if (!settings.init(argc, argv, &err)) {
    std::cout << "Argument parsing failed!" << std::endl << err << std::endl << Settings::usage();
    return -1;
  }
  Camera camera;
  camera.yaw = settings.yaw;
  camera.pitch = settings.pitch;
  NeuralRenderer neural_renderer(settings, camera);
  if (!neural_renderer.init()) {
    std::cout << "NeuralRenderer failed to initialize: " << neural_renderer.error() << std::endl;
    return -1;
  }
  std::cout << "Starting" << std::endl;
  auto t0 = std::chrono::steady_clock::now();
  for (int f = 0; f < settings.frames; ++f) {
    if (!neural_renderer.render()) {
      std::cout << "render failed: " << neural_renderer.error() << std::endl;
      return -1;
    }
  }
  double ms = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() / 1000.0;
  std::cout << "NeuralRenderer iter: " << ms / std::max(1, settings.frames) << " [ms] over " << settings.frames << " frames" << std::endl;
  return 0;
}
