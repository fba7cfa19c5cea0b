/*
 * ibl_hip.h -- C ABI of the MI355X (gfx950) image-based-lighting precompute library.
 *
 * The reference tool (src/utilities/cube.cpp) has no plugin or FFI interface: its
 * seams are three free functions, two codec functions and a box filter inside
 * main().  Each entry point below replaces one of those seams and says which.
 *
 * Conventions
 *   - Every function returns 0 on success and a non-zero IBL_E* code otherwise;
 *     nothing throws across this boundary.  ibl_last_error() returns the message
 *     of the calling thread's last failure (the reference would PANIC,
 *     include/util.hpp:25).
 *   - A cube map is a face-major vertical strip of RGBA float32 texels,
 *     index ((face*H + y)*W + x)*4, faces +X,-X,+Y,-Y,+Z,-Z
 *     (cube.cpp:16-29,96).  Faces are square (the reference strides input rows by
 *     the height, cube.cpp:104,264, so nothing else is meaningful).
 *   - "*_dev" functions take DEVICE pointers and enqueue on `stream` (a hipStream_t;
 *     NULL = the HIP default stream).  They use per-device workspaces the library
 *     owns, so calls on one device must not overlap in time: issue them on one
 *     stream, or order streams with events.  The workspaces only grow: a call that
 *     finds one too small waits for the device, frees it and allocates a larger one
 *     (not allowed under stream capture); every other call only enqueues kernels and
 *     does not synchronise.  ibl_reserve sizes the workspaces ahead of time.
 *   - The library keeps one state (workspaces, timer) per initialised device and
 *     every call works on the state of the calling thread's current HIP device
 *     (hipSetDevice / hipGetDevice).  The states have no locks, like the
 *     single-threaded tool the library serves (cube.cpp:392-517): at most one
 *     thread at a time calls in for a given device.
 *   - The functions without the suffix take HOST pointers, copy in, run, copy out
 *     and block; they are what a C++ host that used to call the reference
 *     functions calls instead.
 *   - [out_begin, out_begin+out_count) selects output texels in (face, y, x)
 *     order (cells for the LUT); only those are written.  One process per GPU
 *     shards a bake this way and gathers the slices afterwards.
 */
#ifndef IBL_HIP_H
#define IBL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    IBL_OK = 0,
    IBL_EINVAL = 1,     /* bad size / range / null pointer               */
    IBL_ENODEV = 2,     /* no usable gfx950 device, or ibl_init not called */
    IBL_EHIP = 3,       /* a HIP runtime call failed (message has details) */
    IBL_ENOMEM = 4
};

/* BRDF-table jitter modes (cube.cpp:342-344,366 draws from std::mt19937 seeded by
 * std::random_device; the seed is explicit here). */
enum {
    IBL_LUT_JITTER_COUNTER = 1,  /* counter-based hash of (seed, cell, sample)   */
    IBL_LUT_JITTER_MIDPOINT = 2  /* stratum midpoints: deterministic quadrature  */
};

/* ---- life cycle ---------------------------------------------------------- */

/* Initialises HIP device `device` (-1: the current device) for this library and
 * makes it the calling thread's current device.  Fails with IBL_ENODEV when no
 * GPU is present: there is no CPU fallback behind this library.  Tuning knobs are
 * read from the environment by the first init call and never again. */
int ibl_init(int device);
/* Initialises devices 0 .. n_devices-1 (0: every visible device) and leaves device 0
 * current.  With more than one device initialised the ibl_bake_* entry points split
 * every cube over all of them, as integrateLambertian / integrateGGX split their
 * texel loops over the host's cores (cube.cpp:95,240); the bytes they return do not
 * depend on the number of devices. */
int ibl_init_all(int n_devices);
int ibl_initialized_devices(void);
/* Releases every initialised device. */
int ibl_shutdown(void);
const char* ibl_last_error(void);
int ibl_device_count(void);

/* Grows the current device's workspaces to what whole-cube calls of these shapes need:
 * ibl_lambertian_dev from in_w to lambertian_size texels per face edge (0: none), and
 * ibl_ggx_level_dev for every level of the chain of a ggx_top_w face with minimum size
 * ggx_min_size (the tool's --cubemap-size) computed from an in_w input (ggx_top_w 0: none;
 * in_w is ggx_top_w divided by the tool's --input-downscale).  Afterwards such calls neither
 * allocate nor synchronise, e.g. under HIP graph capture. */
int ibl_reserve(uint32_t in_w, uint32_t lambertian_size, uint32_t ggx_top_w, uint32_t ggx_min_size);

/* ---- level geometry (host arithmetic only) ------------------------------- */

/* cube.cpp:147: floor(log2(max(w,h)) - log2(min_size)). */
uint32_t ibl_ggx_mip_levels(uint32_t top_w, uint32_t top_h, uint32_t min_size);
/* cube.cpp:161-170,175: roughness, alpha^2, cutoff and patch size of one level. */
int ibl_ggx_level_params(uint32_t mip_level, uint32_t mip_levels, uint32_t in_w, uint32_t in_h,
                         float* roughness, float* alpha2, float* cutoff, uint32_t* patch_size);

/* How the mirror-octet kernel would run one whole GGX level (host arithmetic only, no device
 * needed): the runs the level's input is dealt to (one partial sum per output texel and run),
 * the launches they are spread over, and the bytes of workspace for the sums (never more than
 * 4 GiB) and for the per-level records.  IBL_EINVAL for sizes the general kernel serves. */
int ibl_ggx_level_plan(uint32_t in_w, uint32_t out_w, uint32_t mip_level, uint32_t mip_levels, uint32_t* runs,
                       uint32_t* passes, uint64_t* sums_bytes, uint64_t* records_bytes);

/* ---- device-resident entry points ---------------------------------------- */

/* integrateLambertian's texel loop, cube.cpp:95-129 (no encode, no file).  On face sizes
 * with exact mirror symmetry (every power of two) the clamped cosine of cube.cpp:112-115 is
 * evaluated as the sum of its even and odd halves, and the octets of texels where those halves
 * cancel by more than a factor 16 are recomputed pair by pair with the reference's own weights
 * (see ibl_last_exact_octets): a texel is within ~2e-6 of the reference's double-accumulated
 * sum either way, and exact zeros stay exact.  Where both faces are multiples of 16 wide and at
 * least 32 the split form works on the orbits of the cube's whole symmetry group -- the 8 sign
 * flips and the 6 axis permutations: 48 texels share six sets of weights, and a Fourier transform
 * over the group leaves 64 products per channel for the 2304 texel pairs of two orbits (knob
 * "lamb_o48", 1 by default).  A permuted member's dot product adds the reference's three products
 * in another order, so single cosines may differ from the reference's in the last bit; with
 * "lamb_o48" 0 the split form keeps to hexadecets -- the sign flips and the exchange of x and y,
 * which leave the reference's dot products unchanged to the bit (knob "lamb_hex", 1 by default;
 * 0: octets everywhere). */
int ibl_lambertian_dev(const float* d_env, uint32_t in_w, uint32_t in_h, uint32_t out_size,
                       uint32_t out_begin, uint32_t out_count, float* d_out, void* stream);

/* The same integral for shard `shard` of `n_shards`: a library-chosen partition of
 * the output texels that keeps mirror-image texels (+-x, +-y, +-z; with the 16-fold kernels
 * also their images under the exchange of x and y, with the 48-fold Lambertian kernel under
 * every axis permutation) on one shard, because the kernel evaluates
 * them together.  Writes that shard's texels into
 * their places in the full-size d_out and touches nothing else; the shards are
 * disjoint and cover the cube, so zero-filled buffers summed over the shards give
 * the whole result.  A texel's value does not depend on n_shards. */
int ibl_lambertian_shard_dev(const float* d_env, uint32_t in_w, uint32_t in_h, uint32_t out_size,
                             uint32_t shard, uint32_t n_shards, float* d_out, void* stream);

/* ibl_lambertian_dev over the whole cube for callers that cannot allocate: code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart
 * in the reference.  The call allocates and frees nothing: the kernels of
 * ibl_lambertian_dev work in a 64 MiB file-scope device array of the library instead of
 * its grow-only workspaces (every GPU the library is initialised on holds that array until
 * the process ends).  Sizes whose records and partial sums do not fit the array are
 * rejected with IBL_EINVAL (512 -> 32, 256 -> 64 and 128 -> 64 texels per face edge fit),
 * and so are face sizes the mirror-octet kernel does not serve (both must be even and
 * exactly mirror-symmetric, as every power of two is).  The result has the bits of
 * ibl_lambertian_dev.  All calls on one GPU share that GPU's array: calls that overlap in
 * time on one GPU race, whatever their streams. */
int ibl_lambertian_static_dev(const float* d_env, uint32_t in_w, uint32_t in_h, uint32_t out_size,
                              float* d_out, void* stream);

/* One iteration of integrateGGX's level loop, cube.cpp:152-309 (no encode, no
 * file).  Rejects min(in_w,in_h) < 16, where the reference divides by a zero
 * patch size (cube.cpp:175,182).
 * Arithmetic: the weights of cube.cpp:274-290 are evaluated in FP32 as the reference does; the
 * sums of cube.cpp:292-295 are FP32 over at most 256 terms, then FP32 over a run of those, then
 * FP64.  On mirror-symmetric face sizes with an input of 64 x 64 and more the products
 * weight x radiance of those first sums are formed on the matrix pipe from two bf16 pieces of each
 * factor (16 significant bits; knob "ggx_matrix", 1 by default, 0 keeps them in FP32 on the vector
 * pipe).  The 2^-18 cut of a term is random and averages out over the texels under a lobe; where it
 * could not -- patches whose radiance amounts to fewer than 16 equal texels in some channel (knob
 * "ggx_few"; a sun of a few texels, one lamp on black) -- the patch is walked by a second launch
 * that adds into the same sums: with three bf16 pieces of each factor and six partial products
 * (products exact to 2^-25) at the levels the 16-fold kernel serves, with FP32 products on the
 * vector pipe at the others.  That kernel evaluates a weight once for the 16 images of a texel pair
 * under the cube's sign flips and the exchange of x and y, which leave the reference's cosine
 * unchanged to the bit; it serves outputs of 32 x 32 and more whose width is a multiple of 16 (knob
 * "ggx_hex", 1 by default; 0: the 8 sign flips only, as at the smaller levels).  Measured against
 * the FP32 form: <= 1e-6, 99.99 % of the RGBE texels identical, lone lights included.  On inputs of
 * 1024 x 1024 and more, at the levels whose acceptance window ends above the horizon
 * (cutoff > 1/2), that form takes sqrt(NoH^2) / denom^2 of cube.cpp:280-281 as one reciprocal
 * square root (knob "ggx_rsq": 1 this rule, 0 never, 2 at every such level; the two forms agree
 * to 5e-7). */
int ibl_ggx_level_dev(const float* d_env, uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h,
                      uint32_t mip_level, uint32_t mip_levels,
                      uint32_t out_begin, uint32_t out_count, float* d_out, void* stream);

/* The same level for shard `shard` of `n_shards`; semantics as
 * ibl_lambertian_shard_dev. */
int ibl_ggx_level_shard_dev(const float* d_env, uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h,
                            uint32_t mip_level, uint32_t mip_levels,
                            uint32_t shard, uint32_t n_shards, float* d_out, void* stream);

/* ---- shards: the exchange step between GPUs ---------------------------------- */

/* The reference splits the texel loops of integrateLambertian / integrateGGX over the host's
 * cores (cube.cpp:95,240) and every thread stores into the one shared cube.  Between GPUs the
 * store becomes an exchange: every GPU packs the texels of its shard, one all-gather moves the
 * packed shards (each texel crosses the links once), and every GPU (or only the one that
 * encodes) scatters them into its cube.  The functions below describe the partition that
 * ibl_lambertian_shard_dev (kind IBL_SHARD_LAMBERTIAN; mip arguments ignored) and
 * ibl_ggx_level_shard_dev (IBL_SHARD_GGX) use for a cube of face edge out_w computed from an
 * input of face edge in_w; it depends on which kernel family serves those sizes. */
enum { IBL_SHARD_LAMBERTIAN = 0, IBL_SHARD_GGX = 1 };
/* Texel slots of one packed shard, the same for every shard (short shards are padded); 0 for
 * invalid arguments.  Host arithmetic only. */
uint32_t ibl_shard_texels(int kind, uint32_t in_w, uint32_t out_w, uint32_t mip_level, uint32_t mip_levels,
                          uint32_t n_shards);
/* The (face, y, x) texel index of every slot of shard `shard`, 0xffffffff for padding.
 * Host arithmetic only (needs no GPU). */
int ibl_shard_indices(int kind, uint32_t in_w, uint32_t out_w, uint32_t mip_level, uint32_t mip_levels,
                      uint32_t shard, uint32_t n_shards, uint32_t* indices);
/* d_packed[slot] = d_cube[texel of the slot] (zero for padding); texel_bytes is 16 (RGBA
 * float32) or 4 (RGBE). */
int ibl_shard_pack_dev(int kind, uint32_t in_w, uint32_t out_w, uint32_t mip_level, uint32_t mip_levels,
                       uint32_t shard, uint32_t n_shards, uint32_t texel_bytes, const void* d_cube, void* d_packed,
                       void* stream);
/* The inverse for all shards at once: d_packed_all holds the n_shards packed shards back to
 * back (what an all-gather delivers); every texel of the cube is written. */
int ibl_shard_unpack_dev(int kind, uint32_t in_w, uint32_t out_w, uint32_t mip_level, uint32_t mip_levels,
                         uint32_t n_shards, uint32_t texel_bytes, const void* d_packed_all, void* d_cube, void* stream);

/* generateGGXLUT's cell loop, cube.cpp:346-385 (no printing): d_out is
 * [n_rough][n_cos][2] float32 (slope, bias). */
int ibl_brdf_lut_dev(uint32_t n_rough, uint32_t n_cos, uint32_t samples_sqrt, uint64_t seed, int jitter_mode,
                     uint32_t cell_begin, uint32_t cell_count, float* d_out, void* stream);

/* convertRGBEtoRGB / convertRGBtoRGBE, src/util.cpp:93-145, n texels. */
int ibl_rgbe_decode_dev(const uint8_t* d_rgbe, float* d_rgba, uint32_t n, void* stream);
int ibl_rgbe_encode_dev(const float* d_rgba, uint8_t* d_rgbe, uint32_t n, void* stream);

/* The box filter of main, cube.cpp:482-501: (w x 6h) strip -> (w/d x 6h/d). */
int ibl_downscale_dev(const float* d_env, uint32_t w, uint32_t h, uint32_t d, float* d_out, void* stream);

/* getNormalFrom followed by linear::normalize, src/utilities/cube.cpp:31-50 with
 * include/linear.hpp:114-122: the unit direction of every texel of a w x h x 6 cube,
 * as both integrals compute it (cube.cpp:97,197,242).  d_out receives w*h*6 float4
 * (x, y, z, 0) in (face, y, x) order.  normalized == 0 stores the unnormalised
 * getNormalFrom vector instead (cube.cpp:103,187). */
int ibl_texel_directions_dev(uint32_t w, uint32_t h, int normalized, float* d_out, void* stream);

/* ---- sampled GGX mode (no counterpart in the reference) -------------------- */

/* The reference integrates every GGX level over all input texels (cube.cpp:240-313).
 * This mode estimates the same normalised integral, including the reference's cutoff
 * window (cube.cpp:284-290), with n_samples GGX-distributed half vectors per output
 * texel (Hammersley points) and filtered reads from a box pyramid of the environment.
 * Results approach the quadrature as n_samples grows (lod_bias 0); they are never
 * byte-identical with it.  For callers that trade accuracy for time (previews, large
 * environments). */

/* Number of levels and of float4 texels of the box pyramid of a w x w x 6 cube: sizes
 * w, w/2, ... down to an odd size or 1; level k+1 is the 2x2 box mean of level k (the
 * filter of cube.cpp:482-501 with factor 2).  Every face of every level is stored with
 * a one-texel apron ((S+2) x (S+2) texels for an S x S face) that holds, for each tap
 * position just outside the face, the texel of the neighbouring face its direction
 * falls on; the layout is the library's own, fill it with ibl_env_pyramid_dev. */
uint32_t ibl_env_pyramid_levels(uint32_t w);
uint64_t ibl_env_pyramid_texels(uint32_t w);

/* Fills d_pyramid (ibl_env_pyramid_texels(w) float4, levels back to back, level 0 first). */
int ibl_env_pyramid_dev(const float* d_env, uint32_t w, float* d_pyramid, void* stream);

/* One level (size out_w, roughness mip_level / mip_levels as in cube.cpp:161-170) for the
 * output texels [out_begin, out_begin + out_count); lod_bias is added to the pyramid level
 * chosen for a sample (1.0 is the usual choice for small n_samples). */
int ibl_ggx_level_sampled_dev(const float* d_pyramid, uint32_t in_w, uint32_t out_w, uint32_t mip_level,
                              uint32_t mip_levels, uint32_t n_samples, float lod_bias, uint32_t out_begin,
                              uint32_t out_count, float* d_out, void* stream);

/* PNG scanline filtering of a w x h RGBA8 image, the step of the tool's PNG writer
 * between the RGBE encoder and deflate (cube.cpp:141,322 through stb_image_write,
 * ext/stb_image_write.h:1092-1174): every row is filtered with the PNG filter type
 * (none, sub, up, average, Paeth) whose output has the least sum of absolute values
 * read as signed bytes, the lowest type winning ties.  d_rows receives
 * ibl_png_rows_bytes(w, h) bytes: per row the filter type, then 4*w filtered bytes --
 * the stream an IDAT chunk holds before compression. */
uint64_t ibl_png_rows_bytes(uint32_t w, uint32_t h);
int ibl_png_filter_rows_dev(const uint8_t* d_rgba8, uint32_t w, uint32_t h, uint8_t* d_rows, void* stream);

/* ---- mirror-local re-projection -------------------------------------------- */

/* The viewer re-projects the pre-filtered chain of a mirror-local environment every frame: a
 * raster pass per (mip level, face) draws the environment's bounding mesh, back faces only, from
 * the camera reflected across the mirror plane and looks the chain up in the direction of each
 * fragment's position (src/main.cpp:523-604, shaders/mirrorLocal.vert:78-87,
 * shaders/mirrorLocal.frag:17-19).  ibl_mirror_reproject_dev is that pass for one level as a
 * per-texel computation, for a bounding mesh that is the axis-aligned box [box_min, box_max] of
 * the environment's local space: d_src and d_out are w x w x 6 RGBA float32 cubes of the level;
 * `camera` is the reflected camera position in that space (ibl_mirror_reflect_camera).  Texels
 * whose ray misses the box get the pass's clear colour (0, 0, 0, 1).  The lookup is bilinear
 * inside a face; bit parity with the hardware rasteriser and cube sampler is not claimed.  The
 * caller loops over the levels like main.cpp:533. */
int ibl_mirror_reproject_dev(const float* d_src, uint32_t w, const float* camera, const float* box_min,
                             const float* box_max, float* d_out, void* stream);
/* mirrorLocal.vert:81-82 (host arithmetic): transform is the environment's column-major 4x4
 * world-to-local matrix, camera the world-space camera position; the local position is
 * reflected across the plane z = -mirror_dist. */
int ibl_mirror_reflect_camera(const float* transform, const float* camera, float mirror_dist, float* out);

/* ---- host-buffer entry points -------------------------------------------- */

/* The reference's functions on caller-owned host buffers, blocking: integrateLambertian's and
 * integrateGGX's texel loops (cube.cpp:95-133, 240-313) and generateGGXLUT's cell loop
 * (cube.cpp:346-385), minus encoding, files and printing.  As the reference's one call uses every
 * core of the host (cube.cpp:95,240), these use every initialised device: after ibl_init_all(n)
 * the output is split into n shards (the partition of the *_shard_dev entry points; ranges of cells
 * for the table), one host thread per device copies the input to its device, computes its shard
 * and puts its texels into `out`.  The floats returned do not depend on the number of devices.
 * ibl_rgbe_*, ibl_downscale and ibl_png_filter_rows are streaming passes and stay on the current
 * device. */
int ibl_lambertian(const float* env, uint32_t in_w, uint32_t in_h, uint32_t out_size, float* out);
int ibl_ggx_level(const float* env, uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h,
                  uint32_t mip_level, uint32_t mip_levels, float* out);
/* Sampled mode on host buffers: builds the pyramid of env, then the level. */
int ibl_ggx_level_sampled(const float* env, uint32_t in_w, uint32_t out_w, uint32_t mip_level, uint32_t mip_levels,
                          uint32_t n_samples, float lod_bias, float* out);
int ibl_brdf_lut(uint32_t n_rough, uint32_t n_cos, uint32_t samples_sqrt, uint64_t seed, int jitter_mode,
                 float* out_slope_bias);
int ibl_rgbe_decode(const uint8_t* rgbe, float* rgba, uint32_t n);
int ibl_rgbe_encode(const float* rgba, uint8_t* rgbe, uint32_t n);
int ibl_downscale(const float* env, uint32_t w, uint32_t h, uint32_t d, float* out);

/* Progress of the ibl_bake_* calls: fn(cube, user) is called, from a thread of the library,
 * each time a cube's bytes have arrived in the caller's buffer -- cube 0 is the Lambertian
 * cube, cube m > 0 GGX level m -- where the reference draws its progress bars
 * (cube.cpp:76-92,131-132,220-236,311-312).  Calls never overlap.  On one device they come in
 * that order, each while the device is already baking the next cube, so whatever the callback
 * does with a finished cube (compress it, write its file) costs the bake no time; on several
 * devices the thread that delivers a cube's last shard reports it, in the order the cubes
 * complete.  NULL switches it off. */
typedef void (*ibl_progress_fn)(int cube, void* user);
int ibl_set_progress(ibl_progress_fn fn, void* user);

/* The whole bake of main, cube.cpp:457-514, with the input resident on the device
 * across all steps: RGBE strip in, RGBE strips out.  rgbe_in is w x 6w x 4 bytes.
 * lambertian_out (may be NULL) receives cubemap_size^2*6*4 bytes; ggx_out (may be
 * NULL) receives the levels 1..mip_levels back to back, level m being
 * (w>>m)^2*6*4 bytes.  mip sizes derive from the pre-downscale width
 * (cube.cpp:473-474,513). */
int ibl_bake_rgbe(const uint8_t* rgbe_in, uint32_t w, uint32_t input_downscale, uint32_t cubemap_size,
                  uint8_t* lambertian_out, uint8_t* ggx_out);
/* With several devices initialised (ibl_init_all) the three bake calls run one host thread per
 * device: every device decodes the strip, computes its shard of every cube (the partition of
 * the *_shard_dev entry points), packs its texels and hands them to the host, where they are
 * put in place; the thread that brings a cube's last shard finishes the cube (PNG scanlines: it
 * filters the assembled texels on its own device) and reports it.  The bytes returned do not
 * depend on the number of devices.  A machine with one GPU rehearses this path with the knobs
 * "bake_rehearse_shards" = n (the device plays n shards in turn) and "bake_rehearse_concurrent" = 1
 * (all n at once, every thread with workspaces and a stream of its own); the host-buffer entry
 * points above follow the same knobs. */

/* The same bake with the GGX levels computed by the sampled mode above (ggx_samples > 0;
 * 0 selects the quadrature, i.e. ibl_bake_rgbe).  No counterpart in the reference. */
int ibl_bake_rgbe_sampled(const uint8_t* rgbe_in, uint32_t w, uint32_t input_downscale, uint32_t cubemap_size,
                          uint32_t ggx_samples, float lod_bias, uint8_t* lambertian_out, uint8_t* ggx_out);

/* The same bake, handing back PNG scanline streams instead of pixels (the filter step of
 * ibl_png_filter_rows_dev runs on the device after the encoder): lambertian_rows receives
 * ibl_png_rows_bytes(cubemap_size, 6*cubemap_size) bytes, ggx_rows the levels back to back,
 * level m being ibl_png_rows_bytes(w>>m, 6*(w>>m)) bytes.  The caller deflates and wraps
 * them in IHDR/IDAT/IEND. */
int ibl_bake_png_rows(const uint8_t* rgbe_in, uint32_t w, uint32_t input_downscale, uint32_t cubemap_size,
                      uint32_t ggx_samples, float lod_bias, uint8_t* lambertian_rows, uint8_t* ggx_rows);
int ibl_png_filter_rows(const uint8_t* rgba8, uint32_t w, uint32_t h, uint8_t* rows);

/* ---- measurement ---------------------------------------------------------- */

/* 1: every integral runs on the general kernels (any face size); 0 (default): the
 * mirror-octet kernels wherever the face sizes allow.  Both are held to the same
 * parity bars; the switch exists so that tests can cover both. */
int ibl_set_kernel_mode(int general_only);
/* Launch-shape knobs of the kernels (slice counts, grid orders; the names are the
 * library's own, the tools under tools/ use them).  They change speed only, except
 * the slice counts, which fix the order of the sums and with it the last bits, and
 * "lamb_split" / "lamb_hex" / "lamb_o48" / "ggx_matrix" / "ggx_hex" / "ggx_few" / "ggx_rsq", which choose between two forms of the arithmetic (see
 * ibl_lambertian_dev and ibl_ggx_level_dev). */
int ibl_set_tuning(const char* name, double value);
/* The current value of a knob ("general_only" reads back ibl_set_kernel_mode). */
int ibl_get_tuning(const char* name, double* value);

/* When enabled, every *_dev call brackets its dominant kernel with HIP events on
 * the stream it was launched on.  A Lambertian call on the mirror-octet kernels has two: the
 * split-form pair kernel and, behind it, the exact-weights pair kernel for the octets the first
 * marked; ibl_last_kernel_ms / ibl_kernel_ms_back give the sum of the two. */
int ibl_set_timing(int enabled);
/* Waits for the last bracketed kernel and returns its duration in ms. */
int ibl_last_kernel_ms(float* ms);
/* The same for an earlier timed call on this device: calls_back 0 is the last one, 1 the one
 * before, ... (the library keeps the event pairs of the last 64 timed calls; a call that had
 * nothing to launch records none and does not count).  Lets a caller queue a series of calls
 * and read their kernel times afterwards, without a wait between them. */
int ibl_kernel_ms_back(uint32_t calls_back, float* ms);
/* Work done by the last lambertian/ggx *_dev call as the kernels counted it
 * (GGX counts need ibl_set_counting(1); they cost a few percent). */
int ibl_set_counting(int enabled);
int ibl_last_counts(uint64_t* patch_tests, uint64_t* tested, uint64_t* accepted);
/* Mirror-octet kernels, counting on: what the wavefronts of the last Lambertian or GGX call
 * executed.  wave_records: records walked, summed over wavefronts (a record is one input texel
 * octet, evaluated by the 64 lanes of a wavefront for 64 output octets); wave_patterns: sign
 * patterns evaluated for them (the Lambertian kernel evaluates 4 to 7 per record, skipping the
 * ones a whole wavefront agrees are negative; the GGX kernel walks a record once per pattern).
 * With the kernels' instruction counts per record and per pattern this prices the arithmetic
 * actually executed.  0 for the general kernels. */
int ibl_last_wave_work(uint64_t* wave_records, uint64_t* wave_patterns);
/* Mirror-octet GGX kernel, counting on: the records of the patch octets its wavefronts walked,
 * each counted once however many sign patterns were live there (wave_records of
 * ibl_last_wave_work counts one per live pattern).  The quotient is the number of patterns a
 * record is evaluated under at that level. */
int ibl_last_wave_visits(uint64_t* wave_visits);
/* Mirror-octet Lambertian, counting on.  The integral (cube.cpp:95-129) is evaluated with the
 * clamped cosine split into its even and odd halves, a quarter of the arithmetic; where a texel
 * comes out as the small difference of much larger halves (a texel in deep shadow of a very
 * bright light: cancellation above the knob "lamb_split_limit", 16), its octet (the texel and
 * its 7 mirror images) is recomputed with the reference's own per-pair weights.  This returns
 * how many octets of the last call were.  ("lamb_split" 0 computes everything with those
 * weights.) */
int ibl_last_exact_octets(uint64_t* octets);

#ifdef __cplusplus
}
#endif
#endif /* IBL_HIP_H */
