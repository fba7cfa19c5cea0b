// The cube tool's input images other than PNG.  The reference loads its input with
// stbi_load(path, &w, &h, &n, STBI_rgb_alpha) (cube.cpp:458), which recognises a file by its
// contents, not its name, and converts whatever it finds to RGBA8.  This file does the same for
// the formats with simple containers -- BMP, binary PGM/PPM, Radiance HDR, TGA -- by stb_image's
// conversion rules (ext/stb_image.h: stbi__bmp_load :5528, stbi__tga_load :5868, stbi__hdr_load
// :7155 with stbi__hdr_to_ldr :1883, stbi__pnm_load :7503, the order of the tests in
// stbi__load_main :1136); JPEG, GIF, PSD and PIC are recognised and refused by name.
#include "png_io.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

constexpr uint32_t kMaxDimension = 1u << 24;      // stb_image's STBI_MAX_DIMENSIONS

struct Bytes {
    const std::vector<unsigned char>& d;
    size_t pos = 0;
    bool short_read = false;
    explicit Bytes(const std::vector<unsigned char>& data) : d(data) {}
    // (a read past the end gives zeros, as stb_image's does, and is remembered)
    uint32_t u8() { if (pos < d.size()) return d[pos++]; short_read = true; return 0; }
    uint32_t le16() { uint32_t a = u8(); return a | (u8() << 8); }
    uint32_t le32() { uint32_t a = le16(); return a | (le16() << 16); }
    void skip(long n) {
        if (n <= 0) return;
        if ((size_t)n > d.size() - pos) { pos = d.size(); short_read = true; } else pos += (size_t)n;
    }
    bool eof() const { return pos >= d.size(); }
};

bool fits(uint32_t w, uint32_t h, std::string& err) {
    if (w == 0 || h == 0) { err = "image has no pixels"; return false; }
    if (w > kMaxDimension || h > kMaxDimension || (uint64_t)w * h * 4 > (1ull << 31)) { err = "image too large"; return false; }
    return true;
}

// ---- BMP -------------------------------------------------------------------------------------

int high_bit(uint32_t z) { int n = -1; while (z) { n++; z >>= 1; } return n; }
int bit_count(uint32_t z) { int n = 0; while (z) { n += z & 1; z >>= 1; } return n; }

// A field of `bits` bits whose top bit sits `shift` above bit 7, widened to 8 bits by repeating it.
uint32_t widen_field(uint32_t v, int shift, int bits) {
    static const uint32_t mul[9] = {0, 0xff, 0x55, 0x49, 0x11, 0x21, 0x41, 0x81, 0x01};
    static const uint32_t shr[9] = {0, 0, 0, 1, 0, 2, 4, 6, 0};
    v = shift < 0 ? v << -shift : v >> shift;
    v >>= 8 - bits;
    return (v * mul[bits]) >> shr[bits];
}

bool is_bmp(const std::vector<unsigned char>& f) {
    if (f.size() < 18 || f[0] != 'B' || f[1] != 'M') return false;
    uint32_t hsz = f[14] | f[15] << 8 | f[16] << 16 | (uint32_t)f[17] << 24;
    return hsz == 12 || hsz == 40 || hsz == 56 || hsz == 108 || hsz == 124;
}

bool read_bmp(const std::vector<unsigned char>& file, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba,
              std::string& err) {
    Bytes s(file);
    s.skip(10);
    const long offset = (long)(int32_t)s.le32();
    const uint32_t hsz = s.le32();
    long header_read = 14;          // what the masks of a 40-byte header add to the header's size
    uint32_t mr = 0, mg = 0, mb = 0, ma = 0, all_a = 255;
    int32_t w, h;
    if (offset < 0) { err = "bad BMP"; return false; }
    if (hsz == 12) { w = (int32_t)s.le16(); h = (int32_t)s.le16(); }
    else { w = (int32_t)s.le32(); h = (int32_t)s.le32(); }
    if (s.le16() != 1) { err = "bad BMP"; return false; }
    const uint32_t bpp = s.le16();
    if (hsz != 12) {
        const uint32_t compress = s.le32();
        if (compress == 1 || compress == 2) { err = "BMP type not supported: RLE"; return false; }
        if (compress >= 4) { err = "BMP type not supported: unsupported compression"; return false; }
        if (compress == 3 && bpp != 16 && bpp != 32) { err = "bad BMP"; return false; }
        s.skip(20);
        auto default_masks = [&]() {
            if (bpp == 16) { mr = 31u << 10; mg = 31u << 5; mb = 31u; }
            else if (bpp == 32) { mr = 0xffu << 16; mg = 0xffu << 8; mb = 0xffu; ma = 0xffu << 24; all_a = 0; }
            else mr = mg = mb = ma = 0;
        };
        if (hsz == 40 || hsz == 56) {
            if (hsz == 56) s.skip(16);
            if (bpp == 16 || bpp == 32) {
                if (compress == 0) default_masks();
                else {
                    mr = s.le32(); mg = s.le32(); mb = s.le32();
                    header_read += 12;
                    if (mr == mg && mg == mb) { err = "bad BMP"; return false; }
                }
            }
        } else {
            mr = s.le32(); mg = s.le32(); mb = s.le32(); ma = s.le32();
            if (compress != 3) default_masks();
            s.skip(4 + 48 + (hsz == 124 ? 16 : 0));
        }
    }
    const bool bottom_up = h > 0;
    if (h < 0) h = -h;
    if (w <= 0 || !fits((uint32_t)w, (uint32_t)h, err)) { if (err.empty()) err = "bad BMP"; return false; }

    long psize = 0;
    if (hsz == 12) { if (bpp < 24) psize = (offset - header_read - 24) / 3; }
    else if (bpp < 16) psize = (offset - header_read - (long)hsz) >> 2;
    if (psize == 0) {
        if ((long)s.pos > 1024 || offset < (long)s.pos || offset - (long)s.pos > 1024) { err = "Corrupt BMP"; return false; }
        s.skip(offset - (long)s.pos);
    }
    rgba.assign((size_t)w * h * 4, 0);
    size_t z = 0;
    if (bpp < 16) {
        if (psize <= 0 || psize > 256) { err = "Corrupt BMP"; return false; }
        unsigned char pal[256][3];
        for (long i = 0; i < psize; i++) {
            pal[i][2] = (unsigned char)s.u8(); pal[i][1] = (unsigned char)s.u8(); pal[i][0] = (unsigned char)s.u8();
            if (hsz != 12) s.u8();
        }
        for (long i = psize; i < 256; i++) pal[i][0] = pal[i][1] = pal[i][2] = 0;
        s.skip(offset - header_read - (long)hsz - psize * (hsz == 12 ? 3 : 4));
        uint32_t row_bytes;
        if (bpp == 1) row_bytes = ((uint32_t)w + 7) >> 3;
        else if (bpp == 4) row_bytes = ((uint32_t)w + 1) >> 1;
        else if (bpp == 8) row_bytes = (uint32_t)w;
        else { err = "Corrupt BMP"; return false; }
        const uint32_t pad = (0u - row_bytes) & 3u;
        for (int32_t j = 0; j < h; j++) {
            uint32_t v = 0;
            for (int32_t i = 0; i < w; i++) {
                uint32_t index;
                if (bpp == 1) { if ((i & 7) == 0) v = s.u8(); index = (v >> (7 - (i & 7))) & 1u; }
                else if (bpp == 4) { if ((i & 1) == 0) v = s.u8(); index = (i & 1) ? (v & 15u) : (v >> 4); }
                else index = s.u8();
                rgba[z++] = pal[index][0]; rgba[z++] = pal[index][1]; rgba[z++] = pal[index][2]; rgba[z++] = 255;
            }
            s.skip(pad);
        }
    } else {
        s.skip(offset - header_read - (long)hsz);
        uint32_t row_bytes = bpp == 24 ? 3u * (uint32_t)w : bpp == 16 ? 2u * (uint32_t)w : 0u;
        const uint32_t pad = (0u - row_bytes) & 3u;
        int easy = 0;
        if (bpp == 24) easy = 1;
        else if (bpp == 32 && mb == 0xffu && mg == 0xff00u && mr == 0x00ff0000u && ma == 0xff000000u) easy = 2;
        else if (bpp != 16 && bpp != 32) { err = "Corrupt BMP"; return false; }
        int rs = 0, gs = 0, bs = 0, as = 0, rc = 0, gc = 0, bc = 0, ac = 0;
        if (!easy) {
            if (!mr || !mg || !mb) { err = "Corrupt BMP"; return false; }
            rs = high_bit(mr) - 7; rc = bit_count(mr);
            gs = high_bit(mg) - 7; gc = bit_count(mg);
            bs = high_bit(mb) - 7; bc = bit_count(mb);
            as = high_bit(ma) - 7; ac = bit_count(ma);
            if (rc > 8 || gc > 8 || bc > 8 || ac > 8) { err = "Corrupt BMP"; return false; }
        }
        for (int32_t j = 0; j < h; j++) {
            for (int32_t i = 0; i < w; i++) {
                uint32_t a;
                if (easy) {
                    rgba[z + 2] = (uint8_t)s.u8(); rgba[z + 1] = (uint8_t)s.u8(); rgba[z] = (uint8_t)s.u8();
                    a = easy == 2 ? s.u8() : 255u;
                } else {
                    const uint32_t v = bpp == 16 ? s.le16() : s.le32();
                    rgba[z] = (uint8_t)widen_field(v & mr, rs, rc);
                    rgba[z + 1] = (uint8_t)widen_field(v & mg, gs, gc);
                    rgba[z + 2] = (uint8_t)widen_field(v & mb, bs, bc);
                    a = ma ? widen_field(v & ma, as, ac) : 255u;
                }
                all_a |= a;
                rgba[z + 3] = (uint8_t)a;
                z += 4;
            }
            s.skip(pad);
        }
    }
    if (all_a == 0)             // an alpha channel of zeros throughout means "no alpha"
        for (size_t i = 3; i < rgba.size(); i += 4) rgba[i] = 255;
    if (bottom_up) {
        const size_t row = (size_t)w * 4;
        for (int32_t j = 0; j < h / 2; j++)
            for (size_t i = 0; i < row; i++) std::swap(rgba[(size_t)j * row + i], rgba[(size_t)(h - 1 - j) * row + i]);
    }
    width = (uint32_t)w; height = (uint32_t)h;
    return true;
}

// ---- binary PGM / PPM ------------------------------------------------------------------------

bool is_pnm(const std::vector<unsigned char>& f) { return f.size() > 2 && f[0] == 'P' && (f[1] == '5' || f[1] == '6'); }

bool read_pnm(const std::vector<unsigned char>& file, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba,
              std::string& err) {
    Bytes s(file);
    s.skip(1);
    const uint32_t comp = s.u8() == '6' ? 3u : 1u;
    char c = (char)s.u8();
    auto is_space = [](char ch) { return ch == ' ' || ch == '\t' || ch == '\n' || ch == '\v' || ch == '\f' || ch == '\r'; };
    auto skip_space = [&]() {
        for (;;) {
            while (!s.eof() && is_space(c)) c = (char)s.u8();
            if (s.eof() || c != '#') break;
            while (!s.eof() && c != '\n' && c != '\r') c = (char)s.u8();
        }
    };
    bool overflow = false;
    auto integer = [&]() {
        uint64_t v = 0;
        while (!s.eof() && c >= '0' && c <= '9') {
            v = v * 10 + (uint64_t)(c - '0');
            c = (char)s.u8();
            if (v > 2147483647ull) { overflow = true; return 0u; }
        }
        return (uint32_t)v;
    };
    skip_space();
    const uint32_t w = integer();
    skip_space();
    const uint32_t h = integer();
    skip_space();
    const uint32_t maxv = integer();      // (the one whitespace byte behind it is already in c)
    if (overflow || w == 0 || h == 0) { err = "PPM image header had zero or overflowing width"; return false; }
    if (maxv > 65535) { err = "PPM image supports only 8-bit and 16-bit images"; return false; }
    if (!fits(w, h, err)) return false;
    const uint32_t bytes = maxv > 255 ? 2u : 1u;
    const size_t need = (size_t)w * h * comp * bytes;
    if (file.size() - s.pos < need) { err = "PNM file truncated"; return false; }
    const unsigned char* p = &file[s.pos];
    rgba.resize((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        // 16-bit samples: stb_image reads the big-endian pairs as they lie into native 16-bit words and keeps a word's
        // upper byte -- on the little-endian machines the reference runs on, the SECOND byte of the pair
        uint8_t v[3];
        for (uint32_t k = 0; k < comp; k++) v[k] = bytes == 2 ? p[(i * comp + k) * 2 + 1] : p[i * comp + k];
        rgba[4 * i] = v[0];
        rgba[4 * i + 1] = comp == 3 ? v[1] : v[0];
        rgba[4 * i + 2] = comp == 3 ? v[2] : v[0];
        rgba[4 * i + 3] = 255;
    }
    width = w; height = h;
    return true;
}

// ---- Radiance HDR ----------------------------------------------------------------------------

bool is_hdr(const std::vector<unsigned char>& f) {
    auto starts = [&](const char* sig) { size_t n = strlen(sig); return f.size() >= n && memcmp(f.data(), sig, n) == 0; };
    return starts("#?RADIANCE\n") || starts("#?RGBE\n");
}

// One RGBE pixel as the 8-bit texel stb_image makes of it: float = mantissa * 2^(e - 136), then
// pow(v, 1/2.2) * 255 + 0.5, clamped and truncated; alpha 255.
void hdr_texel(const unsigned char* rgbe, uint8_t* out) {
    static const float gamma_i = 1.0f / 2.2f;
    for (int k = 0; k < 3; k++) {
        float v = 0.0f;
        if (rgbe[3] != 0) v = (float)rgbe[k] * (float)std::ldexp(1.0f, (int)rgbe[3] - 136);
        float z = (float)std::pow((double)(v * 1.0f), (double)gamma_i) * 255 + 0.5f;
        if (z < 0) z = 0;
        if (z > 255) z = 255;
        out[k] = (uint8_t)(int)z;
    }
    out[3] = 255;
}

bool read_hdr(const std::vector<unsigned char>& file, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba,
              std::string& err) {
    Bytes s(file);
    auto line = [&]() {
        std::string t;
        while (!s.eof()) {
            char c = (char)s.u8();
            if (c == '\n') break;
            if (t.size() < 1022) t.push_back(c);
        }
        return t;
    };
    line();                                   // the signature
    bool valid = false;
    for (;;) {
        std::string t = line();
        if (t.empty()) break;
        if (t == "FORMAT=32-bit_rle_rgbe") valid = true;
        if (s.eof()) break;
    }
    if (!valid) { err = "Unsupported HDR format"; return false; }
    std::string dims = line();
    if (dims.compare(0, 3, "-Y ") != 0) { err = "Unsupported HDR format"; return false; }
    char* rest = nullptr;
    const long h = strtol(dims.c_str() + 3, &rest, 10);
    while (*rest == ' ') rest++;
    if (strncmp(rest, "+X ", 3) != 0) { err = "Unsupported HDR format"; return false; }
    const long w = strtol(rest + 3, nullptr, 10);
    if (w <= 0 || h <= 0 || !fits((uint32_t)w, (uint32_t)h, err)) { if (err.empty()) err = "Corrupt HDR image"; return false; }
    rgba.assign((size_t)w * h * 4, 0);
    auto flat_from = [&](size_t first) {      // uncompressed pixels from texel `first` on
        for (size_t i = first; i < (size_t)w * h; i++) {
            unsigned char px[4];
            for (int k = 0; k < 4; k++) px[k] = (unsigned char)s.u8();
            hdr_texel(px, &rgba[4 * i]);
        }
    };
    if (w < 8 || w >= 32768) {
        flat_from(0);
    } else {
        std::vector<unsigned char> scan((size_t)w * 4);
        for (long j = 0; j < h; j++) {
            const uint32_t c1 = s.u8(), c2 = s.u8(), hi = s.u8();
            if (c1 != 2 || c2 != 2 || (hi & 0x80u)) {
                // not run-length coded: these four bytes are the first pixel of a flat image
                if (j != 0) { err = "corrupt HDR"; return false; }
                unsigned char px[4] = {(unsigned char)c1, (unsigned char)c2, (unsigned char)hi, (unsigned char)s.u8()};
                hdr_texel(px, &rgba[0]);
                flat_from(1);
                break;
            }
            if ((long)((hi << 8) | s.u8()) != w) { err = "corrupt HDR"; return false; }
            for (int k = 0; k < 4; k++) {
                long i = 0;
                while (i < w) {
                    uint32_t count = s.u8();
                    const bool run = count > 128;
                    if (run) count -= 128;
                    if (count == 0 || (long)count > w - i) { err = "bad RLE data in HDR"; return false; }
                    if (run) { const unsigned char v = (unsigned char)s.u8(); while (count--) scan[(size_t)(i++) * 4 + k] = v; }
                    else while (count--) scan[(size_t)(i++) * 4 + k] = (unsigned char)s.u8();
                }
            }
            for (long i = 0; i < w; i++) hdr_texel(&scan[(size_t)i * 4], &rgba[((size_t)j * w + i) * 4]);
        }
    }
    width = (uint32_t)w; height = (uint32_t)h;
    return true;
}

// ---- TGA -------------------------------------------------------------------------------------

// Components of a TGA pixel or palette entry of `bits` bits (0: not a format stb_image reads).
uint32_t tga_components(uint32_t bits, bool grey, bool& rgb16) {
    rgb16 = false;
    switch (bits) {
        case 8: return 1;
        case 16: if (grey) return 2; [[fallthrough]];
        case 15: rgb16 = true; return 3;
        case 24: case 32: return bits / 8;
        default: return 0;
    }
}

bool is_tga(const std::vector<unsigned char>& f) {
    if (f.size() < 18) return false;
    const uint32_t color_type = f[1], type = f[2];
    if (color_type > 1) return false;
    if (color_type == 1) {
        if (type != 1 && type != 9) return false;
        const uint32_t pb = f[7];
        if (pb != 8 && pb != 15 && pb != 16 && pb != 24 && pb != 32) return false;
    } else if (type != 2 && type != 3 && type != 10 && type != 11) return false;
    if ((f[12] | f[13] << 8) < 1 || (f[14] | f[15] << 8) < 1) return false;
    const uint32_t bpp = f[16];
    if (color_type == 1 && bpp != 8 && bpp != 16) return false;
    return bpp == 8 || bpp == 15 || bpp == 16 || bpp == 24 || bpp == 32;
}

void tga_rgb16(uint32_t px, unsigned char* out) {
    out[0] = (unsigned char)((((px >> 10) & 31u) * 255u) / 31u);
    out[1] = (unsigned char)((((px >> 5) & 31u) * 255u) / 31u);
    out[2] = (unsigned char)(((px & 31u) * 255u) / 31u);
}

bool read_tga(const std::vector<unsigned char>& file, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba,
              std::string& err) {
    Bytes s(file);
    const uint32_t id_len = s.u8();
    const bool indexed = s.u8() != 0;
    uint32_t type = s.u8();
    const uint32_t pal_start = s.le16(), pal_len = s.le16(), pal_bits = s.u8();
    s.skip(4);
    const uint32_t w = s.le16(), h = s.le16(), bpp = s.u8(), descriptor = s.u8();
    const bool rle = type >= 8;
    if (rle) type -= 8;
    const bool bottom_up = ((descriptor >> 5) & 1u) == 0;
    bool rgb16 = false;
    const uint32_t comp = indexed ? tga_components(pal_bits, false, rgb16) : tga_components(bpp, type == 3, rgb16);
    if (!comp) { err = "Can't find out TGA pixelformat"; return false; }
    if (!fits(w, h, err)) return false;
    s.skip((long)id_len);
    std::vector<unsigned char> pal;
    if (indexed) {
        if (pal_len == 0) { err = "Corrupt TGA"; return false; }
        s.skip((long)pal_start);
        pal.resize((size_t)pal_len * comp);
        for (uint32_t i = 0; i < pal_len; i++) {
            if (rgb16) tga_rgb16(s.le16(), &pal[(size_t)i * comp]);
            else for (uint32_t k = 0; k < comp; k++) pal[(size_t)i * comp + k] = (unsigned char)s.u8();
        }
        if (s.short_read) { err = "Corrupt TGA"; return false; }
    }
    rgba.assign((size_t)w * h * 4, 0);
    unsigned char raw[4] = {0, 0, 0, 0};
    uint32_t left = 0;
    bool repeating = false;
    for (size_t i = 0; i < (size_t)w * h; i++) {
        bool read = true;
        if (rle) {
            if (left == 0) { const uint32_t cmd = s.u8(); left = 1 + (cmd & 127u); repeating = (cmd >> 7) != 0; }
            else if (repeating) read = false;
        }
        if (read) {
            if (indexed) {
                uint32_t idx = bpp == 8 ? s.u8() : s.le16();
                if (idx >= pal_len) idx = 0;
                for (uint32_t k = 0; k < comp; k++) raw[k] = pal[(size_t)idx * comp + k];
            } else if (rgb16) {
                tga_rgb16(s.le16(), raw);
            } else {
                for (uint32_t k = 0; k < comp; k++) raw[k] = (unsigned char)s.u8();
            }
        }
        if (rle) left--;
        const size_t y = i / w, x = i % w;
        uint8_t* out = &rgba[(((bottom_up ? h - 1 - y : y) * (size_t)w) + x) * 4];
        // (24- and 32-bit pixels lie blue first; 15- and 16-bit ones were unpacked red first)
        if (comp == 1) { out[0] = out[1] = out[2] = raw[0]; out[3] = 255; }
        else if (comp == 2) { out[0] = out[1] = out[2] = raw[0]; out[3] = raw[1]; }
        else if (rgb16) { out[0] = raw[0]; out[1] = raw[1]; out[2] = raw[2]; out[3] = 255; }
        else { out[0] = raw[2]; out[1] = raw[1]; out[2] = raw[0]; out[3] = comp == 4 ? raw[3] : 255; }
    }
    width = w; height = h;
    return true;
}

}  // namespace

bool image_read_rgba8(const std::string& path, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba,
                      std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    std::vector<unsigned char> file;
    unsigned char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
    fclose(f);
    static const unsigned char png_sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (file.size() >= 8 && memcmp(file.data(), png_sig, 8) == 0) return png_read_rgba8(path, width, height, rgba, err);
    // the order of stb_image's tests: explicit signatures first, TGA's loose header check last
    if (is_bmp(file)) return read_bmp(file, width, height, rgba, err);
    if (file.size() >= 6 && (memcmp(file.data(), "GIF87a", 6) == 0 || memcmp(file.data(), "GIF89a", 6) == 0)) {
        err = "GIF input is not supported (PNG, BMP, PGM/PPM, HDR and TGA are): " + path; return false;
    }
    if (file.size() >= 4 && memcmp(file.data(), "8BPS", 4) == 0) {
        err = "PSD input is not supported (PNG, BMP, PGM/PPM, HDR and TGA are): " + path; return false;
    }
    if (file.size() >= 92 && file[0] == 0x53 && file[1] == 0x80 && file[2] == 0xF6 && file[3] == 0x34 && memcmp(&file[88], "PICT", 4) == 0) {
        err = "PIC input is not supported (PNG, BMP, PGM/PPM, HDR and TGA are): " + path; return false;
    }
    if (file.size() >= 3 && file[0] == 0xFF && file[1] == 0xD8 && file[2] == 0xFF) {
        err = "JPEG input is not supported (PNG, BMP, PGM/PPM, HDR and TGA are): " + path; return false;
    }
    if (is_pnm(file)) return read_pnm(file, width, height, rgba, err);
    if (is_hdr(file)) return read_hdr(file, width, height, rgba, err);
    if (is_tga(file)) return read_tga(file, width, height, rgba, err);
    err = "unknown image type: " + path;
    return false;
}
