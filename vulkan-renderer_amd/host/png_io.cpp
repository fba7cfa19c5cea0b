// Minimal PNG reader/writer for the cube tool's boundary format: 8-bit,
// non-interlaced images, returned/accepted as tightly packed RGBA8 rows.
// The reference goes through stb_image/stb_image_write for the same job
// (cube.cpp:141,322,458); this is an independent implementation on top of zlib.
#include "png_io.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

const unsigned char kSignature[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};

uint32_t be32(const unsigned char* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

void put_be32(std::vector<unsigned char>& v, uint32_t x) {
    v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16));
    v.push_back((unsigned char)(x >> 8));  v.push_back((unsigned char)x);
}

int paeth(int a, int b, int c) {
    int p = a + b - c;
    int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

void write_chunk(FILE* f, const char type[4], const std::vector<unsigned char>& data) {
    std::vector<unsigned char> head;
    put_be32(head, (uint32_t)data.size());
    fwrite(head.data(), 1, 4, f);
    fwrite(type, 1, 4, f);
    if (!data.empty()) fwrite(data.data(), 1, data.size(), f);
    uLong crc = crc32(0L, (const Bytef*)type, 4);
    if (!data.empty()) crc = crc32(crc, data.data(), (uInt)data.size());
    std::vector<unsigned char> tail;
    put_be32(tail, (uint32_t)crc);
    fwrite(tail.data(), 1, 4, f);
}

}  // namespace

bool png_read_rgba8(const std::string& path, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba,
                    std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    std::vector<unsigned char> file;
    unsigned char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
    fclose(f);
    if (file.size() < 8 || memcmp(file.data(), kSignature, 8) != 0) { err = "not a PNG file: " + path; return false; }

    uint32_t w = 0, h = 0;
    int depth = 0, color = 0, interlace = 0;
    std::vector<unsigned char> idat, palette, trns;
    size_t pos = 8;
    bool seen_end = false;
    while (pos + 12 <= file.size() && !seen_end) {
        uint32_t len = be32(&file[pos]);
        const unsigned char* type = &file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) { err = "truncated PNG chunk"; return false; }
        const unsigned char* data = &file[pos + 8];
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) { err = "bad IHDR"; return false; }
            w = be32(data); h = be32(data + 4);
            depth = data[8]; color = data[9]; interlace = data[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            palette.assign(data, data + len);
        } else if (!memcmp(type, "tRNS", 4)) {
            trns.assign(data, data + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            seen_end = true;
        }
        pos += 12 + (size_t)len;
    }
    if (w == 0 || h == 0) { err = "PNG has no IHDR"; return false; }
    if (depth != 8 || interlace != 0) { err = "only 8-bit non-interlaced PNGs are supported"; return false; }
    int channels;
    switch (color) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: err = "unsupported PNG colour type"; return false;
    }
    const size_t stride = (size_t)w * channels;
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf raw_len = (uLongf)raw.size();
    int zrc = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
    if (zrc != Z_OK || raw_len != raw.size()) { err = "PNG image data is corrupt"; return false; }

    // Undo the per-row filters in place.
    std::vector<unsigned char> img(stride * h);
    for (uint32_t y = 0; y < h; y++) {
        const unsigned char* src = &raw[(stride + 1) * y];
        unsigned char* cur = &img[stride * y];
        const unsigned char* up = y ? &img[stride * (y - 1)] : nullptr;
        int ft = src[0];
        for (size_t i = 0; i < stride; i++) {
            int a = i >= (size_t)channels ? cur[i - channels] : 0;
            int b = up ? up[i] : 0;
            int c = (up && i >= (size_t)channels) ? up[i - channels] : 0;
            int x = src[1 + i];
            switch (ft) {
                case 0: break;
                case 1: x += a; break;
                case 2: x += b; break;
                case 3: x += (a + b) / 2; break;
                case 4: x += paeth(a, b, c); break;
                default: err = "bad PNG filter type"; return false;
            }
            cur[i] = (unsigned char)x;
        }
    }

    rgba.resize((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const unsigned char* p = &img[i * channels];
        uint8_t* o = &rgba[i * 4];
        switch (color) {
            case 0: o[0] = o[1] = o[2] = p[0]; o[3] = 255; break;
            case 2: o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = 255; break;
            case 3: {
                size_t k = p[0];
                if (k * 3 + 2 >= palette.size()) { err = "palette index out of range"; return false; }
                o[0] = palette[k * 3]; o[1] = palette[k * 3 + 1]; o[2] = palette[k * 3 + 2];
                o[3] = k < trns.size() ? trns[k] : 255;
                break;
            }
            case 4: o[0] = o[1] = o[2] = p[0]; o[3] = p[1]; break;
            default: o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = p[3]; break;
        }
    }
    width = w;
    height = h;
    return true;
}

bool png_write_rgba8(const std::string& path, uint32_t width, uint32_t height, const uint8_t* rgba, std::string& err) {
    const size_t stride = (size_t)width * 4;
    std::vector<unsigned char> raw((stride + 1) * height);
    for (uint32_t y = 0; y < height; y++) {
        unsigned char* dst = &raw[(stride + 1) * y];
        const uint8_t* cur = rgba + stride * y;
        const uint8_t* up = y ? rgba + stride * (y - 1) : nullptr;
        dst[0] = up ? 2 : 0;   // "up" filter: neighbouring rows of an environment are similar
        for (size_t i = 0; i < stride; i++) dst[1 + i] = (unsigned char)(cur[i] - (up ? up[i] : 0));
    }
    uLongf bound = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(bound);
    if (compress2(z.data(), &bound, raw.data(), (uLong)raw.size(), 6) != Z_OK) { err = "zlib compress failed"; return false; }
    z.resize(bound);

    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { err = "cannot create " + path; return false; }
    fwrite(kSignature, 1, 8, f);
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, width);
    put_be32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    write_chunk(f, "IHDR", ihdr);
    write_chunk(f, "IDAT", z);
    write_chunk(f, "IEND", {});
    bool ok = fclose(f) == 0;
    if (!ok) err = "write to " + path + " failed";
    return ok;
}
