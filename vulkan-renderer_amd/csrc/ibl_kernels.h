// Host-visible launch plans and launchers of the IBL kernels (internal to the
// library; the public surface is include/ibl_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace ibl {

// Tuning and diagnostic knobs.  ibl_init reads them from the environment once
// (IBL_NO_OCTETS, IBL_LAMB_P, IBL_LAMB_OCT_SLICES, IBL_LAMB_OCT_RUN, IBL_LAMB_OCT_AFFINE,
// IBL_LAMB_SPLIT, IBL_GGX_WAVES, IBL_GGX_XCD_SLICES, IBL_GGX_CORE, IBL_GGX_MATRIX, IBL_GGX_RSQ); no entry point looks at the
// environment afterwards.  ibl_set_tuning / ibl_get_tuning reach all of them by name.
struct Tuning {
    bool no_octets = false;        // general kernels only
    int lamb_p = 0;                // general Lambertian kernel: output texels per thread (0: automatic)
    uint32_t lamb_oct_slices = 0;  // octet Lambertian kernels: maximum slice count (0: 341, and 430 from 1024 x 1024 faces on)
    int lamb_oct_run = -1;         // octet Lambertian kernel: 1 / 0 force / forbid the run form (-1: automatic)
    int lamb_oct_affine = -1;      // octet Lambertian kernel: 1 / 0 force / forbid dealing slots to the XCDs (-1: automatic)
    double ggx_waves = 0.0;        // octet GGX kernel: waves per level to aim for (0: automatic)
    int ggx_xcd_slices = -1;       // octet GGX kernel: 1 / 0 force / forbid dealing runs to the XCDs (-1: automatic)
    int bake_rehearse_shards = 0;  // host-buffer calls and bakes: > 1 makes one device play that many shards in turn (tests)
    int bake_rehearse_concurrent = 0; // ... 1: all at once, each on a stand-in state with its own workspaces and stream
    float ggx_core = 2.0f;         // multiple of alpha^2 below which the reference's own denominator is repeated
    int ggx_early = 1;             // octet GGX kernel, wide lobes: drop records no lane of a wave accepts (0: evaluate all)
    int lamb_split = 1;            // octet Lambertian: even/odd split of the clamped cosine first (0: exact weights only)
    int lamb_hex = 1;              // ... over hexadecets (sign flips and the exchange of x and y) where the sizes allow (lambertian_hex.hip)
    int lamb_o48 = 1;              // ... over the orbits of all 48 symmetries of the cube where lamb_hex applies (lambertian_o48.hip; 0: hexadecets)
    int ggx_matrix = 1;            // octet GGX kernel: a walk's sums on the matrix pipe as bf16 pieces (0: vector pipe only)
    int ggx_hex = 1;               // ... through the 16-fold symmetry where the sizes allow (ggx_hex.hip; 0: octets)
    int ggx_wg_waves = 0;          // waves (neighbouring units) per workgroup of ggx_hex_pairs: 0 = by input size, 1-4
    int ggx_rsq = 1;               // ... its weights at narrow lobes through one v_rsq_f32: 1 for inputs from 1024 x 1024, 2 always, 0 never
    float ggx_few = 16.0f;         // ... patches whose radiance amounts to fewer equal texels than this go to the vector pipe (0: none do)
    double ggx_sums_mib = 0.0;     // octet GGX kernel: memory a level's sums may take, in MiB (0: 4096); tests use it to reach the plans of huge levels on small ones
    float lamb_split_limit = 16.0f; // ... blocks whose texels cancel by more than this factor are redone with exact weights
};
Tuning& tuning();

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
    uint32_t n_runs;       // runs of 8 consecutive slices: the first stage of the summation order
    bool run;              // a workgroup walks a whole run (large calls) or one slice (small calls)
    bool split;            // the split form runs first, the exact kernels only on the blocks it marks
    uint32_t out_blocks;   // 256-octet output blocks the call covers
    size_t partial_bytes;  // [run][24][out_blocks*256] double + (slice form) [slice][24][out_blocks*256] float
    size_t rec_bytes;      // 7 float4 per input octet
};
size_t lambertian_oct_partial_bytes(const LambertianOctPlan& pl);
// lambertian_split.hip: the workspace header (marks of the output blocks, first moment and its parts)
// and the sums of the split form; the pass itself.
size_t lambertian_split_header_bytes(uint32_t n_oct_in, uint32_t out_blocks);
size_t lambertian_split_sums_bytes(const LambertianOctPlan& pl);
hipError_t lambertian_split_launch(const LambertianOctPlan& pl, const float* d_env, uint32_t in_w, uint32_t out_size,
                                   const uint32_t* d_block_list, uint32_t oct_begin, uint32_t oct_end,
                                   uint32_t out_begin, uint32_t out_count, float* d_out, void* d_rec, void* d_work,
                                   unsigned long long* d_counts, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);
hipError_t lambertian_oct_exact_pass(const LambertianOctPlan& pl, const float* d_env, uint32_t in_w, uint32_t out_size,
                                     const uint32_t* d_block_list, uint32_t oct_begin, uint32_t oct_end,
                                     uint32_t out_begin, uint32_t out_count, float* d_out, void* d_rec, double* d_partial,
                                     const uint32_t* d_flags, unsigned long long* counts,
                                     hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);
// lambertian_hex.hip: the split form over hexadecets (the 8 sign flips and the exchange of x and y), with
// lambertian_octets.hip's kernels behind it for the octets it marks.  Output units of 64 hexadecets as in
// ggx_hex.hip (the same unit lists and the same shard partition).
struct LambertianHexPlan {
    LambertianOctPlan oct;   // the second pass: the whole cube's octet blocks
    uint32_t n_rec;          // input records: 64 per unit of the input cube
    uint32_t slice_len, n_slices, n_runs;      // n_runs: groups of slices whose sums a workgroup adds
    uint32_t n_units;        // output units the call covers
    size_t rec_bytes;        // the larger of the two passes' records
    size_t work_bytes;       // header + the larger of the two passes' sums
};
bool lambertian_hex_eligible(uint32_t in_w, uint32_t out_size);
LambertianHexPlan lambertian_hex_plan(uint32_t in_w, uint32_t out_size, uint32_t n_units);
hipError_t lambertian_hex_launch(const LambertianHexPlan& hp, const float* d_env, uint32_t in_w, uint32_t out_size,
                                 const uint32_t* d_units, uint32_t out_begin, uint32_t out_count, float* d_out,
                                 void* d_rec, void* d_work, unsigned long long* d_counts, hipStream_t stream,
                                 hipEvent_t ev0, hipEvent_t ev1, hipEvent_t ev2, hipEvent_t ev3);      // d_units null: every unit of the cube
// lambertian_o48.hip: the split form over the orbits of the cube's whole symmetry group (48 texels: the sign
// flips and all six axis permutations), with lambertian_octets.hip's kernels behind it for the octets it marks.
// Output units of 64 orbits, numbered along the lower triangle of the +Z face's first quadrant.
struct LambertianO48Plan {
    LambertianOctPlan oct;   // the second pass: the whole cube's octet blocks
    uint32_t n_rec;          // input records: one per orbit of the input cube
    uint32_t slice_len, n_slices, n_groups;    // n_groups: groups of slices whose sums a workgroup adds
    uint32_t n_chunks, chunk_len;              // a slice's records are summed in FP32 over chunks of this length
    uint32_t n_units;        // output units the call covers
    size_t recg_offset;      // the weight vectors behind the records
    size_t rec_bytes;        // the larger of the two passes' records
    size_t work_bytes;       // header + the larger of the two passes' sums
};
bool lambertian_o48_eligible(uint32_t in_w, uint32_t out_size);
uint32_t lambertian_o48_units_total(uint32_t out_size);
uint32_t lambertian_o48_units_in_range_count(uint32_t out_w, uint32_t begin, uint32_t count);
hipError_t lambertian_o48_units_in_range_launch(uint32_t out_w, uint32_t begin, uint32_t count, uint32_t* d_list, hipStream_t stream);
LambertianO48Plan lambertian_o48_plan(uint32_t in_w, uint32_t out_size, uint32_t n_units);
hipError_t lambertian_o48_launch(const LambertianO48Plan& op, const float* d_env, uint32_t in_w, uint32_t out_size,
                                 const uint32_t* d_units, uint32_t out_begin, uint32_t out_count, float* d_out,
                                 void* d_rec, void* d_work, unsigned long long* d_counts, hipStream_t stream,
                                 hipEvent_t ev0, hipEvent_t ev1, hipEvent_t ev2, hipEvent_t ev3);      // d_units null: every unit of the cube
uint32_t lambertian_oct_blocks_total(uint32_t out_size);
uint32_t lambertian_oct_tile_edge(uint32_t out_size);
constexpr uint32_t LAMBERTIAN_OCT_BLOCK = 256;   // output octets per workgroup
// Lists of units (blocks of `unit` consecutive octets of the tile-major order, tile edge tw) for calls that
// cover part of a cube: the units with a texel in the (face, y, x) range [begin, begin + count), ascending
// (the host counts them, the device writes the list), or first, first + stride, ...
uint32_t octet_units_in_range_count(uint32_t unit, uint32_t out_w, uint32_t tw, uint32_t begin, uint32_t count);
hipError_t octet_units_in_range_launch(uint32_t unit, uint32_t out_w, uint32_t tw, uint32_t begin, uint32_t count,
                                       uint32_t* d_list, hipStream_t stream);
hipError_t octet_units_strided_launch(uint32_t first, uint32_t stride, uint32_t n, uint32_t* d_list, hipStream_t stream);
LambertianOctPlan lambertian_oct_plan(uint32_t in_w, uint32_t out_blocks);
// Output block b of the call is block d_block_list[b] of the tile-major octet order, or, without a
// list, the 256 octets from oct_begin + 256*b; octets at or past oct_end do not exist.
// ev0/ev1 bracket the call's first pair kernel (the split form's, or the only one); with the split form
// in front, ev2/ev3 bracket the exact-weights pair kernel behind it (a few microseconds when it finds
// nothing marked, about five times the split kernel's time per marked block when it does).
hipError_t lambertian_oct_launch(const LambertianOctPlan& pl, const float* d_env, uint32_t in_w, uint32_t out_size,
                                 const uint32_t* d_block_list, uint32_t oct_begin, uint32_t oct_end,
                                 uint32_t out_begin, uint32_t out_count,
                                 float* d_out, void* d_rec, double* d_partial, unsigned long long* d_counts,
                                 hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1,
                                 hipEvent_t ev2 = nullptr, hipEvent_t ev3 = nullptr);

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
constexpr uint32_t GGX_OCT_UNIT = 64;   // output octets per wave: the unit of sharding
struct GgxOctPlan {
    bool hex;                 // the plan of the 16-fold kernel (ggx_hex.hip): units of 64 hexadecets, patch hexadecets, 50 sums
    GgxLevel lv;
    uint32_t n_po;            // patch octets: 3 * (pw/2)^2
    uint32_t tw;              // edge of the output tile used for octet numbering
    uint32_t n_runs;          // partial sums per output octet; a wave walks one run
    uint32_t passes;          // launches the runs are spread over (each reuses the same buffer of sums)
    uint32_t runs_per_pass;
    uint32_t per_group;       // consecutive runs per group of the two-stage sum
    uint32_t reduce_groups;   // > 1: the runs are summed in this many groups first
    bool accumulate;          // every launch adds its runs into ONE set of FP64 sums (levels whose runs do not fit 8 at a time)
    uint32_t out_units;       // 64-octet units this call covers
    size_t partial_bytes;     // [group][25][out_units*64] double (two-stage and accumulated sums) + [run of one pass][25][out_units*64] float
    bool matrix;              // the level's sums go through the matrix pipe (records in their matrix form)
    bool marks;               // ... except on patch octets whose radiance sits in a few texels (vector pipe, FP32 products)
    size_t mx_bytes;          // the matrix form's records: 144 bytes per input texel octet (0 without the matrix form)
    size_t rec_bytes;         // mx_bytes + 7 float4 per input texel octet (vector form; with marks: room for both)
    size_t pn_bytes;          // float4 per patch octet + one for the count of marked ones
};
void ggx_plan_runs(GgxOctPlan& pl, uint32_t lanes_level, uint32_t units_level, uint32_t members, uint32_t sums,
                   size_t rec_each, uint32_t n_units, double max_waves = 1048576.0);
bool ggx_octets_eligible(uint32_t in_w, uint32_t out_w, uint32_t patch);
uint32_t ggx_oct_tile_edge(uint32_t out_w);
// The call covers the n_units 64-octet units d_units[0], d_units[1], ... (tile-major octet order).
GgxOctPlan ggx_oct_plan(uint32_t in_w, uint32_t out_w, uint32_t mip, uint32_t mips, uint32_t n_units);
hipError_t ggx_oct_launch(const GgxOctPlan& pl, const float* d_env, uint32_t in_w, uint32_t out_w,
                          const uint32_t* d_units, uint32_t out_begin, uint32_t out_count,
                          float* d_out, void* d_rec, void* d_pn, double* d_partial, unsigned long long* d_counts,
                          hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);

// The same level through the cube's 16-fold symmetry (ggx_hex.hip: sign flips and the exchange of x and
// y): hexadecets of texels in place of octets, matrix form only.  Needs out_w % 16 == 0 on top of what the
// octet kernel needs.  The plan is a GgxOctPlan with hex set; units are 8 x 8 tiles of canonical positions
// (the +-X faces' first, then the lower triangle of the +-Z faces').
bool ggx_hex_eligible(uint32_t in_w, uint32_t out_w, uint32_t patch);
uint32_t ggx_hex_units_total(uint32_t out_w);
uint32_t ggx_hex_units_in_range_count(uint32_t out_w, uint32_t begin, uint32_t count);
hipError_t ggx_hex_units_in_range_launch(uint32_t out_w, uint32_t begin, uint32_t count, uint32_t* d_list, hipStream_t stream);
GgxOctPlan ggx_hex_plan(uint32_t in_w, uint32_t out_w, uint32_t mip, uint32_t mips, uint32_t n_units);
hipError_t ggx_hex_launch(const GgxOctPlan& pl, const float* d_env, uint32_t in_w, uint32_t out_w,
                          const uint32_t* d_units, uint32_t out_begin, uint32_t out_count,
                          float* d_out, void* d_rec, void* d_pn, double* d_partial, unsigned long long* d_counts,
                          hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);
// The partition of a cube's output texels into shards (sharding.hip).
enum ShardPartition { SHARD_RANGE = 0, SHARD_LAMB_OCT = 1, SHARD_GGX_OCT = 2, SHARD_GGX_HEX = 3, SHARD_LAMB_O48 = 4 };
struct ShardMap {
    uint32_t partition;   // ShardPartition
    uint32_t out_w;       // face edge of the cube
    uint32_t tw;          // tile edge of the tile-major octet order (octet partitions)
    uint32_t n_shards;
};
uint32_t shard_slots(const ShardMap& m);                                      // slots per shard, padding included
void shard_indices(const ShardMap& m, uint32_t shard, uint32_t* indices);   // host: texel of every slot, 0xffffffff = padding
// elem_bytes 16 (float4 texels) or 4 (RGBE texels)
hipError_t shard_pack_launch(const ShardMap& m, uint32_t shard, uint32_t elem_bytes, const void* d_full, void* d_packed,
                             hipStream_t stream);
hipError_t shard_unpack_launch(const ShardMap& m, uint32_t elem_bytes, const void* d_packed_all, void* d_full,
                               hipStream_t stream);

// Mirror-local re-projection of one cube level (mirror_reproject.hip).
hipError_t mirror_reproject_launch(const float* d_src, uint32_t w, const float cam[3], const float bmin[3],
                                   const float bmax[3], float* d_out, hipStream_t stream);

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
