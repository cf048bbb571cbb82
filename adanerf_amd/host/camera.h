// Free-fly camera of the viewer reduced to what the renderer consumes: position + camera-to-world
// rotation from yaw/pitch (adanerf_real_time_viewer/src/camera.cpp:143-158: dir from yaw/pitch, z-up world,
// right = dir x (0,0,1), up = right x dir, rotation = mat3(lookAt(pos, pos+dir, up))), the W/A/S/D/Q/E movement state
// (camera.cpp:97-126) and the per-batch position step of Camera::UpdateFeaturesBatch (camera.cpp:160-185).
#pragma once

class Camera {
 public:
  float pos[3] = {0, 0, 0};
  float yaw = -80.f, pitch = 0.f;   // degrees
  float speed_mult = 1.f;           // max(view_cell_size / 2) (camera.cpp:47)

  void setPosition(const float p[3]);
  void setViewCell(const float size[3]);          // speed_mult
  void MovementKeyPressed(unsigned char key);     // 'w' 's' 'a' 'd' 'q' 'e'
  void MovementKeyReleased(unsigned char key);
  void MouseDrag(float dx, float dy);             // camera.cpp:128-141 (sensitivity 0.15, pitch clamp +-89)
  // One UpdateFeaturesBatch worth of movement: pos += (dir * fwd + right * rgt + up * upw) * 0.005 * speed_mult with the
  // reference's un-normalised right = dir x (0,0,1) and up = right x dir.  The viewer runs this once per BATCH, so a frame
  // rendered in n batches moves n steps (and tears: later batches see a later position); the headless host applies the
  // n steps in front of the frame.  Returns true if the position changed.
  bool step();
  void getRotMatrix(float rot_c2w_rowmajor[9]) const;
  const float* getPosition() const { return pos; }
  bool moving() const { return move_fwd || move_right || move_up; }

 private:
  int move_fwd = 0, move_right = 0, move_up = 0;
  void basis(float dir[3], float right[3], float up[3]) const;
};
