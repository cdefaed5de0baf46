/*
 * ref_main_camera_pre.h -- prelude force-included in front of the REFERENCE's camera code from src/main.cpp.
 *
 * TEST INFRASTRUCTURE ONLY.  oracle/Makefile pipes the text of /root/reference/src/main.cpp from the line
 * "// --- CAMERA CONTROLLER ---" to the end of `struct PathController` (main.cpp:125-220) into g++ from where it
 * lies -- no copy of it is written anywhere -- followed by oracle/ref_main_camera_post.inc (the extern "C" wrappers,
 * which must sit in the same translation unit because the structs are defined in the piped text).  That range
 * uses nothing of GLFW / GLAD / OpenGL: CameraController::getCUDAStateFrom (:141-167) and
 * PathController::getInterpolatedState / start / update (:176-212) need <cmath>, <algorithm>, the window size
 * macros of config.h, CameraState of raymarcher.h and camera_paths.h -- exactly what main.cpp itself includes
 * for them (main.cpp:14-20).  Everything else of main.cpp (window, GL, recorder, input) is not compiled.
 */
#include <algorithm>
#include <cmath>
#include <cuda_runtime.h>
#include "config.h"
#include "raymarcher.h"
#include "camera_paths.h"
