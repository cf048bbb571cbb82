// Free-fly camera of the viewer reduced to what the renderer consumes: position + camera-to-world
// rotation from yaw/pitch (adanerf_real_time_viewer/src/camera.cpp:143-158: dir from yaw/pitch, z-up world,
// right = dir x (0,0,1), up = right x dir, rotation = mat3(lookAt(pos, pos+dir, up))).
#pragma once

class Camera {
 public:
  float pos[3] = {0, 0, 0};
  float yaw = -80.f, pitch = 0.f;   // degrees

  void setPosition(const float p[3]);
  void MouseDrag(float dx, float dy);             // camera.cpp:128-141 (sensitivity 0.15, pitch clamp +-89)
  void getRotMatrix(float rot_c2w_rowmajor[9]) const;
  const float* getPosition() const { return pos; }
};
