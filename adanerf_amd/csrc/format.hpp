// Exported model-directory format: config.ini, dataset_info.txt, model{0,1}.onnx.
// Host C++ (no device code).  Mirrors the role of the reference viewer's Config
// (adanerf_real_time_viewer/include/config.h:10-63, src/config.cpp:200-344) and of the ONNX
// parser TensorRT ran for it (src/imagegenerator.cpp:84-201) -- only graph.initializer is needed
// because the two topologies on the north-star path are fixed.
#pragma once
#include <map>
#include <string>
#include <vector>

namespace adanerf {

struct Tensor {
  std::vector<int> dims;
  std::vector<float> data;
  int rows() const { return dims.size() > 0 ? dims[0] : 1; }
  int cols() const { return dims.size() > 1 ? dims[1] : 1; }
};
using TensorMap = std::map<std::string, Tensor>;

// Config: same member names as the reference's Config class so host code reads alike.
struct Config {
  std::vector<std::vector<float>> posEncArgs;
  std::vector<std::string> posEnc, inFeatures, outFeatures;
  std::vector<int> numRaymarchSamples;
  std::vector<std::string> rayMarchSampler, rayMarchNormalization, activation;
  std::vector<std::string> losses;   // training config key; only losses[0] matters (oracle output transform)
  std::vector<float> rayMarchSamplingStep, rayMarchSamplingNoise;
  std::vector<float> rayMarchNormalizationCenter;   // training config key (src/features.py:317, 460-467): three floats replace view_cell_center in the normalisation
  std::vector<int> raySampleInput, multiDepthFeatures;
  std::string depthTransform = "linear";
  std::vector<float> zNear, zFar;
  float adaptiveSamplingThreshold = -1.0f;
  std::string accumulationMult;
  bool useNDC = false;
  double fov = 0.0;      // float64 like the PyTorch path's camera_angle_x (src/datasets.py:181-182): the pixel-ray table is float64 arithmetic on it
  float max_depth = 0.f;
  std::vector<float> viewcellCenter, viewcellSize, depthRange;

  // load(dir): config.ini then dataset_info.txt; returns false + message on failure
  bool load(const std::string& dir, std::string* err);
  void store(std::string key, std::string value);
};

bool read_onnx_initializers(const std::string& path, TensorMap* out, std::string* err);

std::string join_path(const std::string& dir, const std::string& file);

}  // namespace adanerf
