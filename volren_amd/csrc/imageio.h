// imageio.h -- the two image formats the offline path touches: Radiance .hdr in (cppgl Texture2D(path),
// environment.cpp:9) and 8-bit PNG out (cppgl Texture2D::save_ldr, main.cpp:554).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace vr {

// Radiance RGBE (flat or new-style RLE) -> float RGB, rows in file order (top row first).
// value = mantissa * 2^(e-136).  Throws std::runtime_error.
void load_hdr(const std::string& path, std::vector<float>& rgb, int& w, int& h);

// RGBA8 PNG, rows given top first.
void save_png_rgba8(const std::string& path, const uint8_t* rgba, int w, int h);

// float RGBA framebuffer with row 0 at the bottom -> 8-bit (x*255 + .5 clamped), flipped to top-first
void framebuffer_to_rgba8(const float* fb, int w, int h, std::vector<uint8_t>& out);

}  // namespace vr
