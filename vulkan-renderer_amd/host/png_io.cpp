// PNG reader/writer for the cube tool's boundary format: reads any PNG bit depth and colour
// type, Adam7 interlacing included, and returns tightly packed RGBA8 rows; writes RGBA8.
// The reference goes through stb_image/stb_image_write for the same job
// (cube.cpp:141,322,458); this is an independent implementation on top of zlib.
#include "png_io.hpp"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

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

// One zlib stream made of independently deflated pieces, so that all host cores work on a
// strip: every piece but the last ends with a sync flush (which byte-aligns it), the last
// with the final block; the stream's Adler-32 is the combination of the pieces' checksums.
// Pieces (512 KB: a 6 MB level already keeps twelve threads busy) do not see each other's
// history, which costs a per cent or two of size; level 4
// is 2.8 times faster than 6 on RGBE strips for 1.5 % more bytes.
bool deflate_parallel(const unsigned char* raw, size_t raw_size, std::vector<unsigned char>& out) {
    const size_t piece = (size_t)512 << 10;
    const size_t n_pieces = raw_size == 0 ? 1 : (raw_size + piece - 1) / piece;
    std::vector<std::vector<unsigned char>> parts(n_pieces);
    std::vector<uLong> adlers(n_pieces, 1L);
    std::atomic<size_t> next(0);
    std::atomic<bool> ok(true);
    auto work = [&]() {
        for (size_t k = next++; k < n_pieces; k = next++) {
            const size_t begin = k * piece, len = std::min(piece, raw_size - begin);
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            if (deflateInit2(&zs, 4, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { ok = false; return; }
            std::vector<unsigned char>& dst = parts[k];
            dst.resize(deflateBound(&zs, (uLong)len) + 16);
            zs.next_in = const_cast<Bytef*>(raw + begin);
            zs.avail_in = (uInt)len;
            zs.next_out = dst.data();
            zs.avail_out = (uInt)dst.size();
            const int rc = deflate(&zs, k + 1 == n_pieces ? Z_FINISH : Z_SYNC_FLUSH);
            if ((k + 1 == n_pieces && rc != Z_STREAM_END) || (k + 1 != n_pieces && (rc != Z_OK || zs.avail_in != 0))) ok = false;
            dst.resize(dst.size() - zs.avail_out);
            deflateEnd(&zs);
            adlers[k] = adler32(adler32(0L, Z_NULL, 0), raw + begin, (uInt)len);
        }
    };
    unsigned n_threads = std::thread::hardware_concurrency();
    if (n_threads == 0) n_threads = 1;
    if (n_threads > 16) n_threads = 16;
    if (n_threads > n_pieces) n_threads = (unsigned)n_pieces;
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_threads; t++) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    if (!ok) return false;

    out.clear();
    out.push_back(0x78);
    out.push_back(0x9c);       // zlib header: deflate, 32 KiB window, default level, no dictionary
    uLong adler = adler32(0L, Z_NULL, 0);
    for (size_t k = 0; k < n_pieces; k++) {
        out.insert(out.end(), parts[k].begin(), parts[k].end());
        const size_t begin = k * piece, len = std::min(piece, raw_size - begin);
        adler = adler32_combine(adler, adlers[k], (z_off_t)len);
    }
    put_be32(out, (uint32_t)adler);
    return true;
}

// Undoes the per-row filters of one (sub-)image: rows of 1 + stride bytes in, stride bytes out.
// bpp: bytes per complete pixel, at least 1 (the filter's "left" distance).
bool unfilter_rows_range(const unsigned char* raw, size_t stride, uint32_t y0, uint32_t y1, size_t bpp, unsigned char* img,
                         std::string& err) {
    for (uint32_t y = y0; y < y1; y++) {
        const unsigned char* src = raw + (stride + 1) * y + 1;
        unsigned char* cur = img + stride * y;
        const unsigned char* up = y ? img + stride * (y - 1) : nullptr;
        const int ft = src[-1];
        switch (ft) {       // one loop per filter type: this runs over every byte of a strip that can hold hundreds of megabytes
            case 0:
                memcpy(cur, src, stride);
                break;
            case 1:
                for (size_t i = 0; i < bpp && i < stride; i++) cur[i] = src[i];
                for (size_t i = bpp; i < stride; i++) cur[i] = (unsigned char)(src[i] + cur[i - bpp]);
                break;
            case 2:
                if (!up) memcpy(cur, src, stride);
                else for (size_t i = 0; i < stride; i++) cur[i] = (unsigned char)(src[i] + up[i]);
                break;
            case 3:
                for (size_t i = 0; i < stride; i++) {
                    const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0;
                    cur[i] = (unsigned char)(src[i] + (a + b) / 2);
                }
                break;
            case 4:
                for (size_t i = 0; i < stride; i++) {
                    const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0;
                    const int c = (up && i >= bpp) ? up[i - bpp] : 0;
                    cur[i] = (unsigned char)(src[i] + paeth(a, b, c));
                }
                break;
            default:
                err = "bad PNG filter type";
                return false;
        }
    }
    return true;
}

bool unfilter_rows(const unsigned char* raw, size_t stride, uint32_t h, size_t bpp, unsigned char* img, std::string& err) {
    return unfilter_rows_range(raw, stride, 0, h, bpp, img, err);
}

// Everything the fast path below leaves out: 1/2/4/16-bit samples, Adam7 interlacing, colour keys.
// Conversions to RGBA8 follow stb_image (the reader behind cube.cpp:458's STBI_rgb_alpha load):
// 16-bit samples keep their high byte, low-depth grey is scaled to 0..255, a tRNS colour key makes
// the matching pixels transparent.
bool decode_general(const std::vector<unsigned char>& idat, uint32_t w, uint32_t h, int depth, int color, int interlace,
                    int channels, const std::vector<unsigned char>& palette, const std::vector<unsigned char>& trns,
                    std::vector<uint8_t>& rgba, std::string& err) {
    struct Pass { uint32_t x0, y0, dx, dy; };
    static const Pass adam7[7] = { {0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2} };
    static const Pass whole = {0, 0, 1, 1};
    const Pass* passes = interlace ? adam7 : &whole;
    const int n_passes = interlace ? 7 : 1;
    const size_t bits = (size_t)depth * channels;
    const size_t bpp = bits >= 8 ? bits / 8 : 1;
    size_t total = 0;
    for (int p = 0; p < n_passes; p++) {
        const uint32_t pw = w > passes[p].x0 ? (w - passes[p].x0 + passes[p].dx - 1) / passes[p].dx : 0;
        const uint32_t ph = h > passes[p].y0 ? (h - passes[p].y0 + passes[p].dy - 1) / passes[p].dy : 0;
        if (pw && ph) total += (1 + ((size_t)pw * bits + 7) / 8) * ph;
    }
    std::vector<unsigned char> raw(total);
    uLongf raw_len = (uLongf)raw.size();
    int zrc = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
    if (zrc != Z_OK || raw_len != raw.size()) { err = "PNG image data is corrupt"; return false; }

    std::vector<uint16_t> px((size_t)w * h * channels);      // the samples as stored, before any conversion
    size_t at = 0;
    std::vector<unsigned char> img;
    for (int p = 0; p < n_passes; p++) {
        const uint32_t pw = w > passes[p].x0 ? (w - passes[p].x0 + passes[p].dx - 1) / passes[p].dx : 0;
        const uint32_t ph = h > passes[p].y0 ? (h - passes[p].y0 + passes[p].dy - 1) / passes[p].dy : 0;
        if (!pw || !ph) continue;
        const size_t stride = ((size_t)pw * bits + 7) / 8;
        img.resize(stride * ph);
        if (!unfilter_rows(&raw[at], stride, ph, bpp, img.data(), err)) return false;
        at += (stride + 1) * ph;
        for (uint32_t py = 0; py < ph; py++) {
            const unsigned char* row = &img[stride * py];
            const uint32_t y = passes[p].y0 + py * passes[p].dy;
            for (uint32_t pxi = 0; pxi < pw; pxi++) {
                const uint32_t x = passes[p].x0 + pxi * passes[p].dx;
                uint16_t* o = &px[((size_t)y * w + x) * channels];
                for (int c = 0; c < channels; c++) {
                    const size_t s = (size_t)pxi * channels + c;      // sample number in the row
                    if (depth == 16) o[c] = (uint16_t)((row[2 * s] << 8) | row[2 * s + 1]);
                    else if (depth == 8) o[c] = row[s];
                    else {                                              // 1, 2, 4 bits, most significant first
                        const size_t bit = s * depth;
                        o[c] = (uint16_t)((row[bit / 8] >> (8 - depth - bit % 8)) & ((1 << depth) - 1));
                    }
                }
            }
        }
    }

    const int scale = depth == 1 ? 0xff : depth == 2 ? 0x55 : depth == 4 ? 0x11 : 1;     // low-depth grey to 0..255
    auto to8 = [&](uint16_t v) -> uint8_t { return depth == 16 ? (uint8_t)(v >> 8) : (uint8_t)(v * (color == 3 ? 1 : scale)); };
    const bool keyed = (color == 0 && trns.size() >= 2) || (color == 2 && trns.size() >= 6);
    uint16_t key[3] = {0, 0, 0};
    for (int c = 0; keyed && c < (color == 0 ? 1 : 3); c++) {
        const uint16_t k16 = (uint16_t)((trns[2 * c] << 8) | trns[2 * c + 1]);
        key[c] = depth == 16 ? k16 : (uint16_t)((k16 & 255) * scale);
    }
    rgba.resize((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const uint16_t* p = &px[i * channels];
        uint8_t* o = &rgba[i * 4];
        switch (color) {
            case 0: {
                o[0] = o[1] = o[2] = to8(p[0]);
                const uint16_t cmp = depth == 16 ? p[0] : (uint16_t)o[0];
                o[3] = (keyed && cmp == key[0]) ? 0 : 255;
                break;
            }
            case 2: {
                o[0] = to8(p[0]); o[1] = to8(p[1]); o[2] = to8(p[2]);
                bool match = keyed;
                for (int c = 0; match && c < 3; c++) match = (depth == 16 ? p[c] : (uint16_t)o[c]) == key[c];
                o[3] = match ? 0 : 255;
                break;
            }
            case 3: {
                const size_t k = p[0];
                if (k * 3 + 2 >= palette.size()) { err = "palette index out of range"; return false; }
                o[0] = palette[k * 3]; o[1] = palette[k * 3 + 1]; o[2] = palette[k * 3 + 2];
                o[3] = k < trns.size() ? trns[k] : 255;
                break;
            }
            case 4: o[0] = o[1] = o[2] = to8(p[0]); o[3] = to8(p[1]); break;
            default: o[0] = to8(p[0]); o[1] = to8(p[1]); o[2] = to8(p[2]); o[3] = to8(p[3]); break;
        }
    }
    return true;
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
    // Every buffer below is sized from w and h.  The largest strip the library takes is 9459 x 56754
    // (2^31 bytes of RGBA8); anything beyond that is refused here, before a product of the two can wrap.
    if ((uint64_t)w * h * 4 > 0x80000000ull) { err = "PNG too large (more than 2^31 bytes of RGBA8)"; return false; }
    int channels;
    switch (color) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: err = "unsupported PNG colour type"; return false;
    }
    const bool depth_ok = color == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                        : color == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                                     : (depth == 8 || depth == 16);
    if (!depth_ok || interlace > 1) { err = "unsupported PNG bit depth or interlace method"; return false; }
    if (depth != 8 || interlace != 0 || ((color == 0 || color == 2) && !trns.empty())) {
        if (!decode_general(idat, w, h, depth, color, interlace, channels, palette, trns, rgba, err)) return false;
        width = w;
        height = h;
        return true;
    }
    const size_t stride = (size_t)w * channels;
    std::vector<unsigned char> raw((stride + 1) * h);
    // RGBA8 input (what the tool's own strips are) is unfiltered straight into the result.
    std::vector<unsigned char> other;
    if (color == 6) rgba.resize(stride * h);
    else other.resize(stride * h);
    unsigned char* const img = color == 6 ? rgba.data() : other.data();
    // Inflate is serial by the format and the longest single step of a bake's start (a 2048 x 2048 strip: 100 MB,
    // 0.3 s).  It hands over what it has every few megabytes, and a second thread undoes the row filters behind it.
    {
        std::atomic<size_t> ready{0};          // bytes of `raw` that are final
        std::atomic<bool> failed{false};
        std::string filter_err;
        bool filter_ok = true;
        std::thread rows([&]() {
            uint32_t y = 0;
            while (y < h) {
                const size_t have = ready.load(std::memory_order_acquire);
                const uint32_t upto = (uint32_t)std::min<size_t>(h, have / (stride + 1));
                if (upto > y) {
                    if (!unfilter_rows_range(raw.data(), stride, y, upto, (size_t)channels, img, filter_err)) { filter_ok = false; return; }
                    y = upto;
                } else if (failed.load(std::memory_order_acquire)) {
                    return;
                } else {
                    std::this_thread::yield();
                }
            }
        });
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        bool ok = inflateInit(&zs) == Z_OK;
        if (ok) {
            size_t in_at = 0, out_at = 0;
            int zrc = Z_OK;
            while (zrc == Z_OK) {
                // (zlib counts in 32 bits: both sides are fed in pieces)
                if (zs.avail_in == 0) {
                    const size_t n = std::min<size_t>(idat.size() - in_at, (size_t)1 << 30);
                    zs.next_in = idat.data() + in_at;
                    zs.avail_in = (uInt)n;
                    in_at += n;
                }
                const size_t room = std::min<size_t>(raw.size() - out_at, (size_t)4 << 20);
                zs.next_out = raw.data() + out_at;
                zs.avail_out = (uInt)room;
                zrc = inflate(&zs, Z_NO_FLUSH);
                out_at += room - zs.avail_out;
                ready.store(out_at, std::memory_order_release);
                if (zrc == Z_OK && room == 0) break;                    // more data than the header announced
                if (zrc == Z_BUF_ERROR && zs.avail_in == 0 && in_at < idat.size()) zrc = Z_OK;
            }
            ok = zrc == Z_STREAM_END && out_at == raw.size();
            inflateEnd(&zs);
        }
        if (!ok) failed.store(true, std::memory_order_release);
        rows.join();
        if (!ok) { err = "PNG image data is corrupt"; return false; }
        if (!filter_ok) { err = filter_err; return false; }
    }

    if (color != 6) rgba.resize((size_t)w * h * 4);
    for (size_t i = 0; color != 6 && i < (size_t)w * h; i++) {
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
    return png_write_filtered_rows(path, width, height, raw.data(), raw.size(), err);
}

bool png_write_filtered_rows(const std::string& path, uint32_t width, uint32_t height, const uint8_t* rows, size_t n_bytes,
                             std::string& err) {
    if (n_bytes != ((size_t)width * 4 + 1) * height) { err = "scanline stream has the wrong size"; return false; }
    std::vector<unsigned char> z;
    if (!deflate_parallel(rows, n_bytes, z)) { err = "zlib compress failed"; return false; }

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
