// The viewer's input handling (adanerf_real_time_viewer/src/inputhandler.cpp:19-110) without a window system: the same
// event entry points, fed by whoever has events -- here the `--script` replay of the headless host (one line of events per
// frame), in a windowed host the platform's callbacks.
#pragma once
#include "camera.h"

class NeuralRenderer;

enum class Key { W, A, S, D, Q, E, O, F, BACKSPACE, ESCAPE, OTHER };
enum class Button { LEFT, RIGHT, MIDDLE };

class InputHandler {
 public:
  InputHandler(NeuralRenderer& renderer, Camera& camera) : renderer(renderer), camera(camera) {}
  void keyDown(Key key);                       // W A S D Q E: movement starts
  void keyUp(Key key);                         // movement stops; O toggles the sampling-network view; ESC asks to quit
  void buttonDown(Button button, int x, int y);
  void buttonUp(Button button, int x, int y);
  void mouseMove(int x, int y);                // left button held: Camera::MouseDrag(delta)
  void mouseWheel(int) {}
  bool quitRequested() const { return quit; }
  // one script line = the events in front of one frame: "+w" "-w" (key down / up: w a s d q e o f esc), "b+ x y" / "b- x y"
  // (left button), "m x y" (mouse position); returns false on a malformed line
  bool replay(const char* line);

 private:
  NeuralRenderer& renderer;
  Camera& camera;
  bool mouse_left = false, quit = false;
  int last_x = 0, last_y = 0;
};
