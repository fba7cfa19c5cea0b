// PNG boundary I/O of the cube tool: RGBA8 in, RGBA8 out.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

// Loads an 8-bit non-interlaced PNG of any colour type as tightly packed RGBA8
// (what stbi_load(..., STBI_rgb_alpha) hands the reference, cube.cpp:458).
bool png_read_rgba8(const std::string& path, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba,
                    std::string& err);

// Loads an input image of any format the tool reads -- recognised by its contents as stbi_load recognises
// it (cube.cpp:458): PNG (above), BMP, binary PGM/PPM, Radiance HDR, TGA -- as tightly packed RGBA8,
// converted by stb_image's rules (host/image_formats.cpp); JPEG, GIF, PSD and PIC are refused by name.
bool image_read_rgba8(const std::string& path, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba,
                      std::string& err);

// Writes tightly packed RGBA8 rows (stbi_write_png(path, w, h, 4, data, w*4), cube.cpp:141,322).
bool png_write_rgba8(const std::string& path, uint32_t width, uint32_t height, const uint8_t* rgba, std::string& err);

// Writes an RGBA8 image whose scanlines are already filtered: `rows` holds height rows of
// 1 + 4*width bytes (PNG filter type, filtered bytes), e.g. from ibl_bake_png_rows.
bool png_write_filtered_rows(const std::string& path, uint32_t width, uint32_t height, const uint8_t* rows, size_t n_bytes,
                             std::string& err);
