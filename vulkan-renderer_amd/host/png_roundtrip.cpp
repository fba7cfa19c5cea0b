// Test helper for png_io: png_roundtrip in.png out.png loads a PNG as RGBA8 and
// writes it back.  Prints "W H" on success.
#include "png_io.hpp"

#include <cstdio>

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: png_roundtrip in.png out.png\n"); return 2; }
    uint32_t w = 0, h = 0;
    std::vector<uint8_t> px;
    std::string err;
    if (!png_read_rgba8(argv[1], w, h, px, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    if (!png_write_rgba8(argv[2], w, h, px.data(), err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    printf("%u %u\n", w, h);
    return 0;
}
