// Headless `adanerf`: the reference viewer's entry point (adanerf_real_time_viewer/src/main.cpp:15-96)
// without the window -- parse the same arguments, load the model directory, render frames, log every
// 100 frames, optionally write out.bmp.  Pure host C++ over the C ABI (no HIP in this file).
#include <chrono>
#include <iostream>

#include <cstdio>
#include <fstream>
#include <vector>

#include "camera.h"
#include "inputhandler.h"
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
  if (!(settings.dry_run ? neural_renderer.initHostOnly() : neural_renderer.init())) {
    std::cout << "NeuralRenderer failed to initialize: " << neural_renderer.error() << std::endl;
    return -1;
  }
  InputHandler input(neural_renderer, camera);
  std::vector<std::string> script;
  if (!settings.script.empty()) {       // --script: one line of input events per frame (inputhandler.h)
    std::ifstream f(settings.script);
    if (!f) {
      std::cout << "cannot read " << settings.script << std::endl;
      return -1;
    }
    for (std::string line; std::getline(f, line);) script.push_back(line);
    settings.frames = static_cast<int>(script.size());
  }
  std::cout << "Starting" << std::endl;
  auto t0 = std::chrono::steady_clock::now();
  for (int f = 0; f < settings.frames && !input.quitRequested(); ++f) {
    if (!script.empty() && !input.replay(script[f].c_str())) {
      std::cout << "malformed script line " << f + 1 << ": " << script[f] << std::endl;
      return -1;
    }
    if (input.quitRequested()) break;
    for (int b = 0; b < neural_renderer.batchesPerFrame(); ++b) camera.step();     // Camera::UpdateFeaturesBatch, once per batch
    if (settings.log_camera)
      std::printf("camera %d pos %.9g %.9g %.9g yaw %.9g pitch %.9g view %s\n", f, camera.pos[0], camera.pos[1], camera.pos[2],
                  camera.yaw, camera.pitch, neural_renderer.renderingOracle() ? "oracle" : "image");
    if (!settings.dry_run && !neural_renderer.render()) {
      std::cout << "render failed: " << neural_renderer.error() << std::endl;
      return -1;
    }
  }
  double ms = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() / 1000.0;
  std::cout << "NeuralRenderer iter: " << ms / std::max(1, settings.frames) << " [ms] over " << settings.frames << " frames" << std::endl;
  return 0;
}
