// Host-visible launch plans and launchers of the IBL kernels (internal to the
// library; the public surface is include/ibl_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace ibl {

struct LambertianPlan {
    int p;                 // output texels per thread
    uint32_t slice_len;    // input texels per slice (multiple of the LDS tile)
    uint32_t n_slices;
    uint32_t out_blocks;
    size_t partial_bytes;  // [slice][3][out_count] double
    size_t prep_bytes_a;   // float4 per input texel
    size_t prep_bytes_b;   // float2 per input texel
};
LambertianPlan lambertian_plan(uint32_t n_in, uint32_t out_count);
hipError_t lambertian_launch(const LambertianPlan& pl, const float* d_env, uint32_t in_w, uint32_t in_h,
                             uint32_t out_size, uint32_t out_begin, uint32_t out_count, float* d_out,
                             void* d_a, void* d_b, double* d_partial, hipStream_t stream,
                             hipEvent_t ev0, hipEvent_t ev1);

// Mirror-octet form of the same integral (lambertian_octets.hip); needs even,
// mirror-exact face sizes on both sides.
bool cube_is_mirror_exact(uint32_t w);
struct LambertianOctPlan {
    uint32_t n_oct_in;     // 3 * (in_w/2)^2
    uint32_t slice_len;    // input octets per slice
    uint32_t n_slices;
    uint32_t reduce_groups; // > 1: the slices are summed in this many groups first
    uint32_t out_blocks;
    size_t partial_bytes;  // [slice (+ group)][24][oct_count] double
    size_t rec_bytes;      // 7 float4 per input octet
};
LambertianOctPlan lambertian_oct_plan(uint32_t in_w, uint32_t out_size, uint32_t oct_count);
hipError_t lambertian_oct_launch(const LambertianOctPlan& pl, const float* d_env, uint32_t in_w, uint32_t out_size,
                                 uint32_t oct_begin, uint32_t oct_count, uint32_t out_begin, uint32_t out_count,
                                 float* d_out, void* d_rec, double* d_partial, hipStream_t stream,
                                 hipEvent_t ev0, hipEvent_t ev1);

struct GgxLevel {          // cube.cpp:161-175
    float roughness, alpha2, cutoff;
    uint32_t patch;        // patch edge in texels
};
GgxLevel ggx_level_params(uint32_t mip, uint32_t mips, uint32_t in_w, uint32_t in_h);
uint32_t ggx_mip_levels(uint32_t top_w, uint32_t top_h, uint32_t min_size);
// Denominator value below which a pair's GGX denominator is recomputed with the
// reference's exact float sequence (0: never).  Only narrow lobes need it.
float ggx_core_threshold(float alpha2);

struct GgxPlan {
    GgxLevel lv;
    int px, py;                    // output texels per thread = px*py; tile = 16px x 16py
    uint32_t tiles_x, tiles_y;     // tiles per face
    uint32_t n_patches;            // 6 * (w/patch) * (h/patch)
    uint32_t n_slices;
    size_t partial_bytes;          // [slice][out_count][4] double
    size_t prep_bytes_a;           // float4 per input texel (unit direction, |w|^-3)
    size_t prep_bytes_b;           // float4 per input texel (radiance * |w|^-3)
    size_t prep_bytes_pn;          // float4 per patch (unit centre direction)
};
GgxPlan ggx_plan(uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h, uint32_t mip, uint32_t mips,
                 uint32_t out_count);
hipError_t ggx_launch(const GgxPlan& pl, const float* d_env, uint32_t in_w, uint32_t in_h,
                      uint32_t out_w, uint32_t out_h, uint32_t out_begin, uint32_t out_count, float* d_out,
                      void* d_a, void* d_b, void* d_pn, double* d_partial, unsigned long long* d_counts,
                      hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);

// Mirror-octet form of the GGX level (ggx_octets.hip).
struct GgxOctPlan {
    GgxLevel lv;
    uint32_t n_po;            // patch octets: 3 * (pw/2)^2
    uint32_t tw;              // edge of the output tile used for octet numbering
    uint32_t n_slices;
    uint32_t reduce_groups;   // > 1: the slices are summed in this many groups first
    uint32_t out_units;       // 64-octet units this call covers
    size_t partial_bytes;     // [slice (+ group)][25][out_units*64] double
    size_t rec_bytes;         // 7 float4 per input texel octet
    size_t pn_bytes;          // float4 per patch octet
};
bool ggx_octets_eligible(uint32_t in_w, uint32_t out_w, uint32_t patch);
// The call covers the 64-octet units first_block, first_block + block_stride, ... (tile-major octet order)
GgxOctPlan ggx_oct_plan(uint32_t in_w, uint32_t out_w, uint32_t mip, uint32_t mips, uint32_t first_block,
                        uint32_t block_stride);
hipError_t ggx_oct_launch(const GgxOctPlan& pl, const float* d_env, uint32_t in_w, uint32_t out_w,
                          uint32_t first_block, uint32_t block_stride, uint32_t out_begin, uint32_t out_count,
                          float* d_out, void* d_rec, void* d_pn, double* d_partial, unsigned long long* d_counts,
                          hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);

// Sampled GGX mode (ggx_sampled.hip): box pyramid of the environment and the level kernel.
uint32_t pyramid_levels(uint32_t w);
uint64_t pyramid_offset(uint32_t w, uint32_t level);     // in float4 texels
hipError_t pyramid_launch(const float* d_env, uint32_t w, float* d_pyramid, hipStream_t stream);
hipError_t ggx_sampled_launch(const float* d_pyramid, uint32_t in_w, uint32_t out_w, uint32_t mip, uint32_t mips,
                              uint32_t n_samples, float lod_bias, uint32_t out_begin, uint32_t out_count, float* d_out,
                              hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);

hipError_t brdf_lut_launch(uint32_t n_rough, uint32_t n_cos, uint32_t samples_sqrt, uint64_t seed, int mode,
                           uint32_t cell_begin, uint32_t cell_count, float* d_out, hipStream_t stream,
                           hipEvent_t ev0, hipEvent_t ev1);

hipError_t rgbe_decode_launch(const uint8_t* d_rgbe, float* d_rgba, uint32_t n, hipStream_t stream);
hipError_t rgbe_encode_launch(const float* d_rgba, uint8_t* d_rgbe, uint32_t n, hipStream_t stream);
hipError_t texel_directions_launch(uint32_t w, uint32_t h, int normalized, float* d_out, hipStream_t stream);
hipError_t downscale_launch(const float* d_env, uint32_t w, uint32_t h, uint32_t d, float* d_out, hipStream_t stream);
// png_rows.hip: rows of (1 + 4w) bytes, filter type first
hipError_t png_filter_rows_launch(const uint8_t* d_rgba8, uint32_t w, uint32_t h, uint8_t* d_rows, hipStream_t stream);

}  // namespace ibl
