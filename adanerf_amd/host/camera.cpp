#include "camera.h"

#include <cmath>

void Camera::setPosition(const float p[3]) {
  pos[0] = p[0];
  pos[1] = p[1];
  pos[2] = p[2];
}

void Camera::setViewCell(const float size[3]) {
  speed_mult = std::fmax(size[0] / 2, std::fmax(size[1] / 2, size[2] / 2));
}

void Camera::MovementKeyPressed(unsigned char key) {
  if (key == 'w') move_fwd = 1;
  if (key == 's') move_fwd = -1;
  if (key == 'a') move_right = -1;
  if (key == 'd') move_right = 1;
  if (key == 'q') move_up = 1;
  if (key == 'e') move_up = -1;
}

void Camera::MovementKeyReleased(unsigned char key) {
  if (key == 'w' || key == 's') move_fwd = 0;
  if (key == 'a' || key == 'd') move_right = 0;
  if (key == 'q' || key == 'e') move_up = 0;
}

void Camera::basis(float dir[3], float right[3], float up[3]) const {
  const float deg = 0.01745329251994329576923690768489f;      // glm::radians
  const float y = yaw * deg, p = pitch * deg;
  float f[3] = {std::cos(y) * std::cos(p), std::sin(y) * std::cos(p), std::sin(p)};
  const float n = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
  for (int i = 0; i < 3; ++i) dir[i] = f[i] / n;
  right[0] = dir[1];          // dir x (0,0,1), not normalised (camera.cpp:151)
  right[1] = -dir[0];
  right[2] = 0.f;
  up[0] = right[1] * dir[2] - right[2] * dir[1];      // right x dir
  up[1] = right[2] * dir[0] - right[0] * dir[2];
  up[2] = right[0] * dir[1] - right[1] * dir[0];
}

bool Camera::step() {
  if (!moving()) return false;
  float dir[3], right[3], up[3];
  basis(dir, right, up);
  const float speed = 0.005f * speed_mult;
  for (int i = 0; i < 3; ++i) {
    if (move_fwd) pos[i] += dir[i] * speed * static_cast<float>(move_fwd);
    if (move_right) pos[i] += right[i] * speed * static_cast<float>(move_right);
    if (move_up) pos[i] += up[i] * speed * static_cast<float>(move_up);
  }
  return true;
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
