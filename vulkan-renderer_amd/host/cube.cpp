// bin/cube rebuilt on the MI355X library: same flags, defaults, messages, file
// naming and output geometry as the reference tool's main (src/utilities/cube.cpp:
// 392-517), with every integral computed by libibl_hip.so through its C ABI.
//
//   cube [--lambertian] [--ggx] [--cubemap-size N] [--input-downscale D] file.png
//   cube --generate-ggx-lut > ggx_lut.inl
//
// Differences from the reference, all outside the numerics: errors end the
// process with a message and EXIT_FAILURE instead of an uncaught exception
// (the reference's main has no try/catch), the progress bars are printed when a
// step has finished rather than while it runs, and IBL_LUT_SEED may fix the
// jitter seed that the reference draws from std::random_device (cube.cpp:342).
#include "ibl_hip.h"
#include "png_io.hpp"

#include <charconv>
#include <cstdlib>
#include <filesystem>
#include <iostream>
#include <optional>
#include <random>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

#define PANIC(msg) throw std::runtime_error(msg)

namespace {

void check(int rc) {
    if (rc != IBL_OK) PANIC(std::string(ibl_last_error()));
}

// The finished form of the reference's progress bar (cube.cpp:76-92,220-236).
void print_bar() {
    for (uint32_t percent = 1; percent <= 100; percent++) {
        if (percent == 100) std::cout << "#] Complete!" << std::endl;
        else if (percent % 10 == 0) std::cout << "#";
        else std::cout << "=";
    }
}

// generateGGXLUT's printing, cube.cpp:351-356,386,389.
void generate_ggx_lut() {
    const uint32_t range = 256, samples_sqrt = 256;
    uint64_t seed;
    if (const char* e = std::getenv("IBL_LUT_SEED")) seed = std::strtoull(e, nullptr, 10);
    else { std::random_device rd; seed = rd(); }
    std::vector<float> lut((size_t)range * range * 2);
    check(ibl_brdf_lut(range, range, samples_sqrt, seed, IBL_LUT_JITTER_COUNTER, lut.data()));
    for (uint32_t i = 0; i < range * range; i++) {
        std::cout << (i == 0 ? "{" : " ");
        std::cout << (double)lut[i * 2] << ", " << (double)lut[i * 2 + 1] << "," << std::endl;
    }
    std::cout << "}" << std::endl;
}

int run(int argc, char** argv) {
    bool do_lambertian = false, do_ggx = false;
    uint32_t cubemap_size = 16, input_downscale = 1;
    std::optional<std::string> input_file;

    int i = 1;
    const std::vector<std::string_view> args(argv + 0, argv + argc);
    while (i < argc) {
        std::string_view arg = args[i];
        if (arg == "--lambertian") do_lambertian = true;
        else if (arg == "--ggx") do_ggx = true;
        else if (arg == "--cubemap-size") {
            if (++i >= argc) PANIC("missing argument to --cubemap-size");
            std::string_view s = args[i];
            auto r = std::from_chars(s.data(), s.data() + s.size(), cubemap_size);
            if (r.ec == std::errc::invalid_argument || r.ec == std::errc::result_out_of_range)
                PANIC("invalid argument to --cubemap-size");
        } else if (arg == "--input-downscale") {
            if (++i >= argc) PANIC("missing argument to --input-downscale");
            std::string_view s = args[i];
            auto r = std::from_chars(s.data(), s.data() + s.size(), input_downscale);
            if (r.ec == std::errc::invalid_argument || r.ec == std::errc::result_out_of_range)
                PANIC("invalid argument to --input-downscale");
            if (input_downscale == 0 || (input_downscale & (input_downscale - 1)) != 0)
                PANIC("scale factor for --input-downscale should be a power of 2 >= 1");
        } else if (arg == "--generate-ggx-lut") {
            check(ibl_init(-1));
            generate_ggx_lut();
            return 0;
        } else if (!input_file.has_value()) input_file = std::string(arg);
        else PANIC("invalid command line argument: " + std::string(arg));
        i++;
    }
    if (!input_file.has_value()) PANIC("command line arguments do not have an input file listed");
    std::filesystem::path input_path = input_file.value();

    uint32_t w = 0, h = 0;
    std::vector<uint8_t> pixels;
    std::string err;
    if (!png_read_rgba8(input_path.string(), w, h, pixels, err)) PANIC("failed to load texture image! (" + err + ")");
    h /= 6;
    if (w == 0 || h != w) PANIC("input must be a W x 6W strip of square cube faces");
    if (!do_lambertian && !do_ggx) return 0;

    check(ibl_init(-1));
    const uint32_t mips = do_ggx ? ibl_ggx_mip_levels(w, h, cubemap_size) : 0;
    std::vector<uint8_t> lambertian(do_lambertian ? (size_t)cubemap_size * cubemap_size * 6 * 4 : 0);
    size_t ggx_bytes = 0;
    for (uint32_t m = 1; m <= mips; m++) ggx_bytes += (size_t)(w >> m) * (w >> m) * 6 * 4;
    std::vector<uint8_t> ggx(ggx_bytes);
    check(ibl_bake_rgbe(pixels.data(), w, input_downscale, cubemap_size,
                        do_lambertian ? lambertian.data() : nullptr, (do_ggx && mips) ? ggx.data() : nullptr));

    if (do_lambertian) {
        std::cout << "Integrating Lambertian: [" << std::flush;
        print_bar();
        std::filesystem::path out = input_path;
        out.replace_extension(".lambertian.png");
        if (!png_write_rgba8(out.string(), cubemap_size, cubemap_size * 6, lambertian.data(), err)) PANIC(err);
    }
    size_t off = 0;
    for (uint32_t m = 1; m <= mips; m++) {
        const uint32_t mw = w >> m;
        std::cout << "Integrating GGX (Level " << m << " of " << mips << "): [" << std::flush;
        print_bar();
        std::filesystem::path out = input_path;
        out.replace_extension(".ggx" + std::to_string(m) + ".png");
        if (!png_write_rgba8(out.string(), mw, mw * 6, ggx.data() + off, err)) PANIC(err);
        off += (size_t)mw * mw * 6 * 4;
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    try {
        int rc = run(argc, argv);
        ibl_shutdown();
        return rc;
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;
        return EXIT_FAILURE;
    }
}
