#include "camera.h"

#include <cmath>

void Camera::setPosition(const float p[3]) {
  pos[0] = p[0];
  pos[1] = p[1];
  pos[2] = p[2];
}

void Camera::MouseDrag(float dx, float dy) {
  const float sensitivity = 0.15f;
  yaw -= dx * sensitivity;
  pitch -= dy * sensitivity;
  if (pitch > 89.0f) pitch = 89.0f;
  if (pitch < -89.0f) pitch = -89.0f;
}

void Camera::getRotMatrix(float r[9]) const {
  const double deg = 3.14159265358979323846 / 180.0;
  const double y = yaw * deg, p = pitch * deg;
  double f[3] = {std::cos(y) * std::cos(p), std::sin(y) * std::cos(p), std::sin(p)};
  double n = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
  for (double& v : f) v /= n;
  double rt[3] = {f[1] * 1.0 - f[2] * 0.0, f[2] * 0.0 - f[0] * 1.0, 0.0};   // f x (0,0,1)
  n = std::sqrt(rt[0] * rt[0] + rt[1] * rt[1] + rt[2] * rt[2]);
  for (double& v : rt) v /= n;
  double up[3] = {rt[1] * f[2] - rt[2] * f[1], rt[2] * f[0] - rt[0] * f[2], rt[0] * f[1] - rt[1] * f[0]};   // rt x f
  // camera looks along -z, +y up: columns (right, up, -forward), row-major
  for (int i = 0; i < 3; ++i) {
    r[3 * i + 0] = static_cast<float>(rt[i]);
    r[3 * i + 1] = static_cast<float>(up[i]);
    r[3 * i + 2] = static_cast<float>(-f[i]);
  }
}
