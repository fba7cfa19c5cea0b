// C entry points of the PNG boundary code for the Python-side tools (vulkan_renderer_amd.pngio):
// the same reader and writer the cube tool uses, instead of per-byte Python loops.
#include "png_io.hpp"

#include <cstdlib>
#include <cstring>
#include <exception>

// Reads `path` as RGBA8.  On success returns 0 and a malloc'd buffer of width*height*4 bytes in
// *pixels (release it with ibl_png_free); on failure returns 1 and copies a message into err.
namespace {
int read_with(bool any_format, const char* path, uint32_t* width, uint32_t* height, uint8_t** pixels, char* err, size_t err_len) {
    // (nothing may be thrown across the C boundary: a failed allocation becomes a message too)
    auto say = [&](const char* message) {
        if (err && err_len) { std::strncpy(err, message, err_len - 1); err[err_len - 1] = 0; }
        return 1;
    };
    try {
        std::vector<uint8_t> rgba;
        std::string message;
        uint32_t w = 0, h = 0;
        if (!(any_format ? image_read_rgba8(path, w, h, rgba, message) : png_read_rgba8(path, w, h, rgba, message)))
            return say(message.c_str());
        uint8_t* out = (uint8_t*)std::malloc(rgba.size() ? rgba.size() : 1);
        if (!out) return say("out of memory");
        std::memcpy(out, rgba.data(), rgba.size());
        *width = w; *height = h; *pixels = out;
        return 0;
    } catch (const std::exception& e) {
        return say(e.what());
    }
}
}  // namespace

extern "C" {

int ibl_png_read_rgba8(const char* path, uint32_t* width, uint32_t* height, uint8_t** pixels, char* err, size_t err_len) {
    return read_with(false, path, width, height, pixels, err, err_len);
}

// The same for any input format the cube tool reads (PNG, BMP, PGM/PPM, HDR, TGA: image_read_rgba8).
int ibl_image_read_rgba8(const char* path, uint32_t* width, uint32_t* height, uint8_t** pixels, char* err, size_t err_len) {
    return read_with(true, path, width, height, pixels, err, err_len);
}

void ibl_png_free(uint8_t* pixels) { std::free(pixels); }

int ibl_png_write_rgba8(const char* path, uint32_t width, uint32_t height, const uint8_t* rgba, char* err, size_t err_len) {
    std::string message;
    try {
        if (png_write_rgba8(path, width, height, rgba, message)) return 0;
    } catch (const std::exception& e) {
        message = e.what();
    }
    if (err && err_len) { std::strncpy(err, message.c_str(), err_len - 1); err[err_len - 1] = 0; }
    return 1;
}

}  // extern "C"
