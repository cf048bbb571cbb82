#include "inputhandler.h"

#include <cstring>
#include <sstream>
#include <string>

#include "neuralrenderer.h"

static unsigned char movement_char(Key k) {
  switch (k) {
    case Key::W: return 'w';
    case Key::A: return 'a';
    case Key::S: return 's';
    case Key::D: return 'd';
    case Key::Q: return 'q';
    case Key::E: return 'e';
    default: return 0;
  }
}

void InputHandler::keyDown(Key key) {
  if (const unsigned char c = movement_char(key)) camera.MovementKeyPressed(c);
}

void InputHandler::keyUp(Key key) {
  if (const unsigned char c = movement_char(key)) camera.MovementKeyReleased(c);
  else if (key == Key::ESCAPE) quit = true;                 // GL::platform::quit()
  else if (key == Key::O) renderer.switchRenderOracle();
}

void InputHandler::buttonDown(Button button, int x, int y) {
  if (button == Button::LEFT) mouse_left = true;
  last_x = x;
  last_y = y;
}

void InputHandler::buttonUp(Button button, int, int) {
  if (button == Button::LEFT) mouse_left = false;
}

void InputHandler::mouseMove(int x, int y) {
  if (mouse_left && (x != last_x || y != last_y)) camera.MouseDrag(static_cast<float>(x - last_x), static_cast<float>(y - last_y));
  last_x = x;
  last_y = y;
}

static bool key_of(const std::string& s, Key* k) {
  static const struct { const char* name; Key key; } names[] = {{"w", Key::W}, {"a", Key::A}, {"s", Key::S}, {"d", Key::D},
      {"q", Key::Q}, {"e", Key::E}, {"o", Key::O}, {"f", Key::F}, {"backspace", Key::BACKSPACE}, {"esc", Key::ESCAPE}};
  for (const auto& n : names)
    if (s == n.name) {
      *k = n.key;
      return true;
    }
  return false;
}

bool InputHandler::replay(const char* line) {
  std::istringstream in(line);
  std::string tok;
  while (in >> tok) {
    if (tok[0] == '#') break;
    if (tok == "b+" || tok == "b-" || tok == "m") {
      int x, y;
      if (!(in >> x >> y)) return false;
      if (tok == "b+") buttonDown(Button::LEFT, x, y);
      else if (tok == "b-") buttonUp(Button::LEFT, x, y);
      else mouseMove(x, y);
    } else if ((tok[0] == '+' || tok[0] == '-') && tok.size() > 1) {
      Key k;
      if (!key_of(tok.substr(1), &k)) return false;
      if (tok[0] == '+') keyDown(k);
      else keyUp(k);
    } else {
      return false;
    }
  }
  return true;
}
