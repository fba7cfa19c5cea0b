// PNG boundary I/O of the cube tool: RGBA8 in, RGBA8 out.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

// Loads an 8-bit non-interlaced PNG of any colour type as tightly packed RGBA8
// (what stbi_load(..., STBI_rgb_alpha) hands the reference, cube.cpp:458).
bool png_read_rgba8(const std::string& path, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba,
                    std::string& err);

// Writes tightly packed RGBA8 rows (stbi_write_png(path, w, h, 4, data, w*4), cube.cpp:141,322).
bool png_write_rgba8(const std::string& path, uint32_t width, uint32_t height, const uint8_t* rgba, std::string& err);
