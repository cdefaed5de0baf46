/* ref_stb.c -- the reference's own image decoder as a checker library (test infrastructure, build container only).
 *
 * The reference decodes its sky with stbi_load(filename, &w, &h, &channels, 4) (src/main.cpp:240) from the stb_image it
 * vendors (include/stb_image.h, v2.30), compiled into main.cpp by `#define STB_IMAGE_IMPLEMENTATION` (src/main.cpp:1-2)
 * with no other configuration macro.  This translation unit does exactly that with the header WHERE IT LIES
 * (-I/root/reference/include, oracle/Makefile: _ref/libref_stb.so); nothing of it is copied.  It pins SURVEY.md row f1's
 * loader: tests/golden/make_golden.py::make_sky decodes the reference's assets with it and commits digests, crops and the
 * difference to this repository's PIL-based loader; tools/sky_to_raw.py uses it to ship a reference-exact raw sky. */
#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h"

unsigned char* ref_stbi_load_rgba(const char* filename, int* width, int* height, int* channels_in_file) {
    return stbi_load(filename, width, height, channels_in_file, 4);
}
void ref_stbi_free(void* data) { stbi_image_free(data); }
const char* ref_stbi_failure_reason(void) { return stbi_failure_reason(); }
