// bin/cube rebuilt on the MI355X library: same flags, defaults, messages, file
// naming and output geometry as the reference tool's main (src/utilities/cube.cpp:
// 392-517), with every integral computed by libibl_hip.so through its C ABI.
//
//   cube [--lambertian] [--ggx] [--cubemap-size N] [--input-downscale D] file.png
//   cube --generate-ggx-lut > ggx_lut.inl
//   cube [same flags] --ggx-samples N [--ggx-lod-bias B] file.png
//                                             GGX levels by N importance samples per texel
//                                             (an addition; the default is the reference's quadrature)
//   cube [same flags] --scene scene.s72        every environment the scene refers to
//   cube --scene scene.s72 --list-environments   only print which files those are
//
// --scene is an addition: it bakes, in one process, the files the renderer's scene
// loader will look for (src/scene_loader.cpp:509-583) -- the radiance cube of each
// ENVIRONMENT object and the captured <scene>.localIBL_<n>.png of each local one.
//
// Differences from the reference, all outside the numerics: errors end the
// process with a message and EXIT_FAILURE instead of an uncaught exception
// (the reference's main has no try/catch), every progress bar is printed, and its
// file written, when the library reports that cube finished (while the GPUs work on
// the next one) rather than growing while the cube is computed, the bake runs on
// every visible GPU (IBL_CUBE_DEVICES=n limits it), and IBL_LUT_SEED may fix the
// jitter seed that the reference draws from std::random_device (cube.cpp:342).
#include "ibl_hip.h"
#include "png_io.hpp"
#include "s72_scene.hpp"

#include <charconv>
#include <chrono>
#include <cstdlib>
#include <filesystem>
#include <future>
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

struct BakeOptions {
    bool lambertian = false, ggx = false;
    uint32_t cubemap_size = 16, input_downscale = 1;
    uint32_t ggx_samples = 0;      // 0: quadrature over all input texels, as the reference
    float ggx_lod_bias = 1.0f;
};

// Stage times on stderr when IBL_CUBE_TIMING is set (diagnostics only).
struct StageClock {
    bool on = std::getenv("IBL_CUBE_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        const auto now = std::chrono::steady_clock::now();
        if (on) std::cerr << "[cube] " << what << ": " << std::chrono::duration<double, std::milli>(now - t).count() << " ms" << std::endl;
        t = now;
    }
};

// What the progress callback needs to print a cube's bar and write its file as soon as the
// library has delivered it (the reference: cube.cpp:131-141,311-322).
struct BakeOutputs {
    std::filesystem::path input_path;
    uint32_t w = 0, mips = 0, cubemap_size = 0;
    std::vector<uint8_t> lambertian, ggx;
    std::string error;

    void cube_done(int cube) {
        std::string err;
        std::filesystem::path out = input_path;
        if (cube == 0) {
            std::cout << "Integrating Lambertian: [" << std::flush;
            print_bar();
            out.replace_extension(".lambertian.png");
            if (!png_write_filtered_rows(out.string(), cubemap_size, cubemap_size * 6, lambertian.data(), lambertian.size(), err))
                error = err;
            return;
        }
        const uint32_t m = (uint32_t)cube, mw = w >> m;
        size_t off = 0;
        for (uint32_t k = 1; k < m; k++) off += (size_t)ibl_png_rows_bytes(w >> k, (w >> k) * 6);
        std::cout << "Integrating GGX (Level " << m << " of " << mips << "): [" << std::flush;
        print_bar();
        out.replace_extension(".ggx" + std::to_string(m) + ".png");
        if (!png_write_filtered_rows(out.string(), mw, mw * 6, ggx.data() + off, (size_t)ibl_png_rows_bytes(mw, mw * 6), err))
            error = err;
    }
};

void on_cube_done(int cube, void* user) { static_cast<BakeOutputs*>(user)->cube_done(cube); }

// Every visible GPU takes a share of each cube (IBL_CUBE_DEVICES limits how many).
void open_gpus() {
    int n = 0;
    if (const char* e = std::getenv("IBL_CUBE_DEVICES")) n = std::atoi(e);
    check(ibl_init_all(n));
}

// The body of the reference's main after argument parsing, cube.cpp:455-516.
void bake_file(const std::filesystem::path& input_path, const BakeOptions& opt) {
    uint32_t w = 0, h = 0;
    std::vector<uint8_t> pixels;
    std::string err;
    StageClock clock;
    // (the GPUs are opened beside the read: 40 ms that a large strip's inflate hides)
    std::future<void> gpus;
    if (opt.lambertian || opt.ggx) gpus = std::async(std::launch::async, open_gpus);
    const bool loaded = image_read_rgba8(input_path.string(), w, h, pixels, err);
    clock.lap("read input image");
    // (what is wrong with the file is reported first, as before; the future waits for the opener on the way out)
    if (!loaded) PANIC("failed to load texture image! (" + err + ")");
    h /= 6;
    if (w == 0 || h != w) PANIC("input must be a W x 6W strip of square cube faces");
    if (!opt.lambertian && !opt.ggx) return;
    gpus.get();
    clock.lap("open the GPU (what the read did not hide)");
    BakeOutputs outs;
    outs.input_path = input_path;
    outs.w = w;
    outs.cubemap_size = opt.cubemap_size;
    outs.mips = opt.ggx ? ibl_ggx_mip_levels(w, h, opt.cubemap_size) : 0;
    // The library hands back PNG scanlines (RGBE-encoded and filtered on the device); only deflate is left to do here.
    outs.lambertian.resize(opt.lambertian ? (size_t)ibl_png_rows_bytes(opt.cubemap_size, opt.cubemap_size * 6) : 0);
    size_t ggx_bytes = 0;
    for (uint32_t m = 1; m <= outs.mips; m++) ggx_bytes += (size_t)ibl_png_rows_bytes(w >> m, (w >> m) * 6);
    outs.ggx.resize(ggx_bytes);
    check(ibl_set_progress(on_cube_done, &outs));
    const int rc = ibl_bake_png_rows(pixels.data(), w, opt.input_downscale, opt.cubemap_size, opt.ggx_samples, opt.ggx_lod_bias,
                                     opt.lambertian ? outs.lambertian.data() : nullptr,
                                     (opt.ggx && outs.mips) ? outs.ggx.data() : nullptr);
    ibl_set_progress(nullptr, nullptr);
    check(rc);
    if (!outs.error.empty()) PANIC(outs.error);
    clock.lap("bake on the GPU with deflate and PNG writes of finished cubes beside it");
}

// Every environment of a scene, in the loader's numbering.
void bake_scene(const std::filesystem::path& scene_file, const BakeOptions& opt, bool list_only) {
    const std::vector<SceneEnvironment> envs = scene_environments(scene_file);
    for (size_t n = 0; n < envs.size(); n++) {
        const SceneEnvironment& e = envs[n];
        std::cout << "Environment " << n << " \"" << e.name << "\": ";
        if (e.file.empty()) { std::cout << "no radiance or local description" << std::endl; continue; }
        const bool present = std::filesystem::exists(e.file);
        std::cout << e.file.string() << (e.local ? " (local)" : "") << (present ? "" : " (missing)") << std::endl;
        if (list_only || !present) continue;   // the loader also skips environments whose file is absent (:559)
        bake_file(e.file, opt);
    }
}

int run(int argc, char** argv) {
    BakeOptions opt;
    std::optional<std::string> input_file, scene_file;
    bool list_only = false;

    int i = 1;
    const std::vector<std::string_view> args(argv + 0, argv + argc);
    while (i < argc) {
        std::string_view arg = args[i];
        if (arg == "--lambertian") opt.lambertian = true;
        else if (arg == "--ggx") opt.ggx = true;
        else if (arg == "--cubemap-size") {
            if (++i >= argc) PANIC("missing argument to --cubemap-size");
            std::string_view s = args[i];
            auto r = std::from_chars(s.data(), s.data() + s.size(), opt.cubemap_size);
            if (r.ec == std::errc::invalid_argument || r.ec == std::errc::result_out_of_range)
                PANIC("invalid argument to --cubemap-size");
        } else if (arg == "--input-downscale") {
            if (++i >= argc) PANIC("missing argument to --input-downscale");
            std::string_view s = args[i];
            auto r = std::from_chars(s.data(), s.data() + s.size(), opt.input_downscale);
            if (r.ec == std::errc::invalid_argument || r.ec == std::errc::result_out_of_range)
                PANIC("invalid argument to --input-downscale");
            if (opt.input_downscale == 0 || (opt.input_downscale & (opt.input_downscale - 1)) != 0)
                PANIC("scale factor for --input-downscale should be a power of 2 >= 1");
        } else if (arg == "--ggx-samples") {
            if (++i >= argc) PANIC("missing argument to --ggx-samples");
            std::string_view s = args[i];
            auto r = std::from_chars(s.data(), s.data() + s.size(), opt.ggx_samples);
            if (r.ec == std::errc::invalid_argument || r.ec == std::errc::result_out_of_range || opt.ggx_samples == 0)
                PANIC("invalid argument to --ggx-samples");
        } else if (arg == "--ggx-lod-bias") {
            if (++i >= argc) PANIC("missing argument to --ggx-lod-bias");
            char* end = nullptr;
            const std::string s(args[i]);
            opt.ggx_lod_bias = std::strtof(s.c_str(), &end);
            if (end == s.c_str() || *end != '\0') PANIC("invalid argument to --ggx-lod-bias");
        } else if (arg == "--generate-ggx-lut") {
            open_gpus();                 // the table's cells are split over every GPU, like the cubes
            generate_ggx_lut();
            return 0;
        } else if (arg == "--scene") {
            if (++i >= argc) PANIC("missing argument to --scene");
            scene_file = std::string(args[i]);
        } else if (arg == "--list-environments") list_only = true;
        else if (!input_file.has_value()) input_file = std::string(arg);
        else PANIC("invalid command line argument: " + std::string(arg));
        i++;
    }
    if (scene_file.has_value()) {
        if (input_file.has_value()) PANIC("invalid command line argument: " + input_file.value());
        bake_scene(scene_file.value(), opt, list_only);
        return 0;
    }
    if (!input_file.has_value()) PANIC("command line arguments do not have an input file listed");
    bake_file(input_file.value(), opt);
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
