// C ABI of the IBL precompute library (declared in include/ibl_hip.h).
// Host-side only: argument checking, the per-device workspace, stream and event
// plumbing.  There is no CPU implementation behind any entry point: without a
// gfx950 device ibl_init fails and every other call returns IBL_ENODEV.
#include "ibl_hip.h"
#include "ibl_kernels.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(IBL_EHIP, "%s: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

struct Buffer {
    void* ptr = nullptr;
    size_t cap = 0;
};

constexpr int TIMING_RING = 64;

struct State {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;   // host-buffer entry points run on the default stream
    // Timing: a ring of event pairs, one per timed call, so that a caller can queue many calls
    // and read their kernel times afterwards without waiting in between.  ev0 / ev1 are the pair
    // the next timed call records; timed_calls counts the calls that have recorded one.
    hipEvent_t ring0[TIMING_RING] = {}, ring1[TIMING_RING] = {};
    // A second pair per call for calls with two pair kernels (the Lambertian call's exact-weights pass
    // behind its split pass); ring_two says whether the call of a slot recorded it.
    hipEvent_t ring2[TIMING_RING] = {}, ring3[TIMING_RING] = {};
    bool ring_two[TIMING_RING] = {};
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    uint64_t timed_calls = 0;
    bool timing = false, counting = false, timed_valid = false;
    Buffer a, b, pn, partial, counts, stage_in, stage_out, units, png;
    void* static_scratch = nullptr; // this device's copy of the file-scope scratch array (ibl_lambertian_static_dev)
    uint64_t last_counts[7] = {0, 0, 0, 0, 0, 0, 0};   // patch tests, texel tests, accepted, wave-records, wave-patterns, exact blocks
    int counts_pending = 0;          // 1: a GGX call's counters wait on the device, 2: a Lambertian call's
};

// One state per device.  A call works on the state of the calling thread's current HIP
// device (hipGetDevice); S points at it from check_ready() to the end of the call.
constexpr int MAX_DEVICES = 64;
State g_states[MAX_DEVICES];
thread_local State* S = nullptr;
// A worker thread of a rehearsed multi-device call (one GPU playing several at once) works on a
// temporary state of its own -- own workspaces, own stream -- instead of its device's.
thread_local State* S_stand_in = nullptr;
ibl::Tuning g_tuning;
bool g_tuning_read = false;

// Grow-only device allocation.  The old block is freed only after the device has
// drained, because a previous asynchronous call may still be using it.
int ensure(Buffer& buf, size_t bytes) {
    if (bytes <= buf.cap) return IBL_OK;
    HIP_TRY(hipDeviceSynchronize());
    if (buf.ptr) HIP_TRY(hipFree(buf.ptr));
    buf.ptr = nullptr;
    buf.cap = 0;
    size_t want = bytes + bytes / 8;
    hipError_t e = hipMalloc(&buf.ptr, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(IBL_ENOMEM, "hipMalloc(%zu bytes): %s", want, hipGetErrorString(e));
    }
    buf.cap = want;
    return IBL_OK;
}

void release(Buffer& buf) {
    if (buf.ptr) (void)hipFree(buf.ptr);
    buf.ptr = nullptr;
    buf.cap = 0;
}

int check_ready() {
    if (S_stand_in) { S = S_stand_in; return IBL_OK; }
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEVICES || !g_states[d].ready) {
        (void)hipGetLastError();
        S = nullptr;
        return fail(IBL_ENODEV, "ibl_init has not succeeded for the current HIP device (%d) in this process", d);
    }
    S = &g_states[d];
    return IBL_OK;
}

hipStream_t pick(void* stream) { return (hipStream_t)stream; }

bool is_pow2(uint32_t v) { return v && !(v & (v - 1)); }

// The mirror-octet kernels need even, exactly mirror-symmetric face grids on both
// sides.  ibl_set_kernel_mode(1) (or IBL_NO_OCTETS=1 when ibl_init runs) forces the
// general kernels; the tests use it to cover both.
bool use_octets(uint32_t in_w, uint32_t out_w) {
    if (g_tuning.no_octets) return false;
    return ibl::cube_is_mirror_exact(in_w) && ibl::cube_is_mirror_exact(out_w);
}

// The split Lambertian form over hexadecets (lambertian_hex.hip) serves the sizes.
bool use_lamb_hex(uint32_t in_w, uint32_t out_w) {
    return use_octets(in_w, out_w) && g_tuning.lamb_split != 0 && g_tuning.lamb_hex != 0
        && ibl::lambertian_hex_eligible(in_w, out_w);
}

// ... over the orbits of the cube's whole symmetry group (lambertian_o48.hip).
bool use_lamb_o48(uint32_t in_w, uint32_t out_w) {
    return use_lamb_hex(in_w, out_w) && g_tuning.lamb_o48 != 0 && ibl::lambertian_o48_eligible(in_w, out_w);
}

void read_tuning_once() {
    if (g_tuning_read) return;
    g_tuning_read = true;
    if (const char* e = getenv("IBL_NO_OCTETS")) g_tuning.no_octets = e[0] == '1';
    if (const char* e = getenv("IBL_LAMB_P")) g_tuning.lamb_p = atoi(e);
    if (const char* e = getenv("IBL_LAMB_OCT_SLICES")) g_tuning.lamb_oct_slices = (uint32_t)atoi(e);
    if (const char* e = getenv("IBL_LAMB_OCT_RUN")) g_tuning.lamb_oct_run = atoi(e);
    if (const char* e = getenv("IBL_LAMB_OCT_AFFINE")) g_tuning.lamb_oct_affine = atoi(e);
    if (const char* e = getenv("IBL_LAMB_SPLIT")) g_tuning.lamb_split = atoi(e);
    if (const char* e = getenv("IBL_LAMB_HEX")) g_tuning.lamb_hex = atoi(e);
    if (const char* e = getenv("IBL_LAMB_O48")) g_tuning.lamb_o48 = atoi(e);
    if (const char* e = getenv("IBL_GGX_WAVES")) g_tuning.ggx_waves = atof(e);
    if (const char* e = getenv("IBL_GGX_XCD_SLICES")) g_tuning.ggx_xcd_slices = atoi(e);
    if (const char* e = getenv("IBL_GGX_CORE")) g_tuning.ggx_core = (float)atof(e);
    if (const char* e = getenv("IBL_GGX_MATRIX")) g_tuning.ggx_matrix = atoi(e);
    if (const char* e = getenv("IBL_GGX_RSQ")) g_tuning.ggx_rsq = atoi(e);
    if (const char* e = getenv("IBL_GGX_HEX")) g_tuning.ggx_hex = atoi(e);
    if (const char* e = getenv("IBL_GGX_FEW")) g_tuning.ggx_few = (float)atof(e);
}

// The 64 MiB that ibl_lambertian_static_dev works in instead of allocated workspaces.  Every
// device the library runs on holds its own copy of this array, at its own address, until the
// process ends; host code reaches it only through the address hipGetSymbolAddress returns
// with that device current (State::static_scratch).
constexpr size_t STATIC_SCRATCH_BYTES = (size_t)64 << 20;
__device__ __align__(256) unsigned char g_static_scratch[STATIC_SCRATCH_BYTES];

int create_events(State& st) {
    for (int k = 0; k < TIMING_RING; k++) {
        hipError_t e0 = hipEventCreate(&st.ring0[k]);
        hipError_t e1 = e0 == hipSuccess ? hipEventCreate(&st.ring1[k]) : e0;
        if (e1 == hipSuccess) e1 = hipEventCreate(&st.ring2[k]);
        if (e1 == hipSuccess) e1 = hipEventCreate(&st.ring3[k]);
        if (e1 != hipSuccess) {
            for (int j = 0; j <= k; j++) {
                for (hipEvent_t* ring : {st.ring0, st.ring1, st.ring2, st.ring3}) {
                    if (ring[j]) (void)hipEventDestroy(ring[j]);
                    ring[j] = nullptr;
                }
            }
            return fail(IBL_EHIP, "hipEventCreate: %s", hipGetErrorString(e1));
        }
    }
    st.ev0 = st.ring0[0];
    st.ev1 = st.ring1[0];
    st.ev2 = st.ring2[0];
    st.ev3 = st.ring3[0];
    st.timed_calls = 0;
    return IBL_OK;
}

int init_device(int device) {
    State& st = g_states[device];
    if (st.ready) return IBL_OK;
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(IBL_ENODEV, "device %d is %s; the kernels are built for gfx950 only", device, prop.gcnArchName);
    void* scratch = nullptr;
    HIP_TRY(hipGetSymbolAddress(&scratch, HIP_SYMBOL(g_static_scratch)));
    if (int rc = create_events(st)) return rc;
    st.static_scratch = scratch;
    st.device = device;
    st.ready = true;
    return IBL_OK;
}

void drop_state(State& st) {
    release(st.a); release(st.b); release(st.pn); release(st.partial); release(st.counts);
    release(st.stage_in); release(st.stage_out); release(st.units); release(st.png);
    for (int k = 0; k < TIMING_RING; k++)
        for (hipEvent_t* ring : {st.ring0, st.ring1, st.ring2, st.ring3})
            if (ring[k]) (void)hipEventDestroy(ring[k]);
    if (st.stream) (void)hipStreamDestroy(st.stream);
    st = State();
}

void shutdown_device(int device) {
    State& st = g_states[device];
    if (!st.ready) return;
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    drop_state(st);
}

// The device counters of the counting kernel variants, zeroed on the stream (null when counting is off).
int counts_buffer(hipStream_t s, unsigned long long** d_counts) {
    *d_counts = nullptr;
    if (!S->counting) return IBL_OK;
    if (int rc = ensure(S->counts, 8 * sizeof(unsigned long long))) return rc;
    *d_counts = (unsigned long long*)S->counts.ptr;
    HIP_TRY(hipMemsetAsync(*d_counts, 0, 8 * sizeof(unsigned long long), s));
    return IBL_OK;
}

// Device list of the call's output units when the (face, y, x) range leaves part of the cube
// out; a null list with every unit of the cube when it does not.
int units_for_range(uint32_t unit, uint32_t tw, uint32_t out_w, uint32_t begin, uint32_t count, hipStream_t s,
                    const uint32_t** d_list, uint32_t* n_units) {
    const uint32_t n_oct = 3 * (out_w / 2) * (out_w / 2), total = (n_oct + unit - 1) / unit;
    *d_list = nullptr;
    *n_units = total;
    if (begin == 0 && count == 6 * out_w * out_w) return IBL_OK;
    *n_units = ibl::octet_units_in_range_count(unit, out_w, tw, begin, count);
    if (*n_units == 0) return IBL_OK;
    if (int rc = ensure(S->units, (size_t)total * sizeof(uint32_t))) return rc;
    HIP_TRY(ibl::octet_units_in_range_launch(unit, out_w, tw, begin, count, (uint32_t*)S->units.ptr, s));
    *d_list = (const uint32_t*)S->units.ptr;
    return IBL_OK;
}

int check_cube(uint32_t w, uint32_t h, const char* what) {
    if (w == 0 || h == 0) return fail(IBL_EINVAL, "%s: empty cube face %ux%u", what, w, h);
    if (w != h) return fail(IBL_EINVAL, "%s: cube faces must be square, got %ux%u", what, w, h);
    if ((uint64_t)w * h * 6 > 0x7fffffffull / 4) return fail(IBL_EINVAL, "%s: cube %ux%u too large", what, w, h);
    return IBL_OK;
}

// A *_dev call that has recorded its event pair: the next timed call gets the next pair.
// two: the call also recorded the second pair (ev2, ev3).
void finish_timed_call(bool two = false) {
    S->timed_valid = S->timing;
    if (!S->timing) return;
    S->ring_two[S->timed_calls % TIMING_RING] = two;
    S->timed_calls++;
    S->ev0 = S->ring0[S->timed_calls % TIMING_RING];
    S->ev1 = S->ring1[S->timed_calls % TIMING_RING];
    S->ev2 = S->ring2[S->timed_calls % TIMING_RING];
    S->ev3 = S->ring3[S->timed_calls % TIMING_RING];
}

// The kernel time of the timed call in a slot of the ring: its pair kernel, plus its second one's.
int slot_ms(uint64_t slot, float* ms) {
    HIP_TRY(hipEventSynchronize(S->ring1[slot]));
    HIP_TRY(hipEventElapsedTime(ms, S->ring0[slot], S->ring1[slot]));
    if (S->ring_two[slot]) {
        float more = 0.0f;
        HIP_TRY(hipEventSynchronize(S->ring3[slot]));
        HIP_TRY(hipEventElapsedTime(&more, S->ring2[slot], S->ring3[slot]));
        *ms += more;
    }
    return IBL_OK;
}

int check_range(uint32_t begin, uint32_t count, uint64_t total, const char* what) {
    if (begin > total || count > total - begin)
        return fail(IBL_EINVAL, "%s: range [%u, %u+%u) exceeds %llu", what, begin, begin, count,
                    (unsigned long long)total);
    return IBL_OK;
}

}  // namespace

namespace ibl {
Tuning& tuning() { return g_tuning; }
}  // namespace ibl

extern "C" {

const char* ibl_last_error(void) { return g_err; }

int ibl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int ibl_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        (void)hipGetLastError();
        return fail(IBL_ENODEV, "no HIP device visible (%s); this library has no CPU path",
                    e == hipSuccess ? "count is 0" : hipGetErrorString(e));
    }
    read_tuning_once();
    if (device >= n || device >= MAX_DEVICES) return fail(IBL_EINVAL, "device %d out of range (have %d)", device, n);
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (int rc = init_device(device)) return rc;
    HIP_TRY(hipSetDevice(device));
    return IBL_OK;
}

int ibl_init_all(int n_devices) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        (void)hipGetLastError();
        return fail(IBL_ENODEV, "no HIP device visible (%s); this library has no CPU path",
                    e == hipSuccess ? "count is 0" : hipGetErrorString(e));
    }
    read_tuning_once();
    if (n_devices < 0 || n_devices > n || n_devices > MAX_DEVICES)
        return fail(IBL_EINVAL, "%d devices asked for, %d visible", n_devices, n);
    if (n_devices == 0) n_devices = n < MAX_DEVICES ? n : MAX_DEVICES;
    for (int d = 0; d < n_devices; d++)
        if (int rc = init_device(d)) return rc;
    HIP_TRY(hipSetDevice(0));
    return IBL_OK;
}

int ibl_initialized_devices(void) {
    int n = 0;
    for (int d = 0; d < MAX_DEVICES; d++) n += g_states[d].ready ? 1 : 0;
    return n;
}

int ibl_shutdown(void) {
    int cur = -1;
    (void)hipGetDevice(&cur);
    bool any = false;
    for (int d = 0; d < MAX_DEVICES; d++) { any = any || g_states[d].ready; shutdown_device(d); }
    if (any && cur >= 0) (void)hipSetDevice(cur);
    (void)hipGetLastError();
    return IBL_OK;
}

int ibl_set_tuning(const char* name, double value) {
    read_tuning_once();
    if (!name) return fail(IBL_EINVAL, "set_tuning: null name");
    if (!strcmp(name, "lamb_p")) g_tuning.lamb_p = (int)value;
    else if (!strcmp(name, "lamb_oct_slices")) g_tuning.lamb_oct_slices = (uint32_t)value;
    else if (!strcmp(name, "lamb_oct_run")) g_tuning.lamb_oct_run = (int)value;
    else if (!strcmp(name, "lamb_oct_affine")) g_tuning.lamb_oct_affine = (int)value;
    else if (!strcmp(name, "lamb_split")) g_tuning.lamb_split = (int)value;
    else if (!strcmp(name, "lamb_hex")) g_tuning.lamb_hex = (int)value;
    else if (!strcmp(name, "lamb_o48")) g_tuning.lamb_o48 = (int)value;
    else if (!strcmp(name, "lamb_split_limit")) g_tuning.lamb_split_limit = (float)value;
    else if (!strcmp(name, "ggx_waves")) g_tuning.ggx_waves = value;
    else if (!strcmp(name, "ggx_xcd_slices")) g_tuning.ggx_xcd_slices = (int)value;
    else if (!strcmp(name, "ggx_core")) g_tuning.ggx_core = (float)value;
    else if (!strcmp(name, "ggx_early")) g_tuning.ggx_early = (int)value;
    else if (!strcmp(name, "ggx_matrix")) g_tuning.ggx_matrix = (int)value;
    else if (!strcmp(name, "ggx_rsq")) g_tuning.ggx_rsq = (int)value;
    else if (!strcmp(name, "ggx_hex")) g_tuning.ggx_hex = (int)value;
    else if (!strcmp(name, "ggx_wg_waves")) g_tuning.ggx_wg_waves = (int)value;
    else if (!strcmp(name, "ggx_few")) g_tuning.ggx_few = (float)value;
    else if (!strcmp(name, "ggx_sums_mib")) g_tuning.ggx_sums_mib = value;
    else if (!strcmp(name, "bake_rehearse_shards")) g_tuning.bake_rehearse_shards = (int)value;
    else if (!strcmp(name, "bake_rehearse_concurrent")) g_tuning.bake_rehearse_concurrent = (int)value;
    else return fail(IBL_EINVAL, "set_tuning: unknown knob %s", name);
    return IBL_OK;
}

int ibl_get_tuning(const char* name, double* value) {
    read_tuning_once();
    if (!name || !value) return fail(IBL_EINVAL, "get_tuning: null pointer");
    if (!strcmp(name, "lamb_p")) *value = g_tuning.lamb_p;
    else if (!strcmp(name, "lamb_oct_slices")) *value = g_tuning.lamb_oct_slices;
    else if (!strcmp(name, "lamb_oct_run")) *value = g_tuning.lamb_oct_run;
    else if (!strcmp(name, "lamb_oct_affine")) *value = g_tuning.lamb_oct_affine;
    else if (!strcmp(name, "lamb_split")) *value = g_tuning.lamb_split;
    else if (!strcmp(name, "lamb_hex")) *value = g_tuning.lamb_hex;
    else if (!strcmp(name, "lamb_o48")) *value = g_tuning.lamb_o48;
    else if (!strcmp(name, "lamb_split_limit")) *value = g_tuning.lamb_split_limit;
    else if (!strcmp(name, "ggx_waves")) *value = g_tuning.ggx_waves;
    else if (!strcmp(name, "ggx_xcd_slices")) *value = g_tuning.ggx_xcd_slices;
    else if (!strcmp(name, "ggx_core")) *value = g_tuning.ggx_core;
    else if (!strcmp(name, "ggx_early")) *value = g_tuning.ggx_early;
    else if (!strcmp(name, "ggx_matrix")) *value = g_tuning.ggx_matrix;
    else if (!strcmp(name, "ggx_rsq")) *value = g_tuning.ggx_rsq;
    else if (!strcmp(name, "ggx_hex")) *value = g_tuning.ggx_hex;
    else if (!strcmp(name, "ggx_wg_waves")) *value = g_tuning.ggx_wg_waves;
    else if (!strcmp(name, "ggx_few")) *value = g_tuning.ggx_few;
    else if (!strcmp(name, "ggx_sums_mib")) *value = g_tuning.ggx_sums_mib;
    else if (!strcmp(name, "bake_rehearse_shards")) *value = g_tuning.bake_rehearse_shards;
    else if (!strcmp(name, "bake_rehearse_concurrent")) *value = g_tuning.bake_rehearse_concurrent;
    else if (!strcmp(name, "general_only")) *value = g_tuning.no_octets ? 1.0 : 0.0;
    else return fail(IBL_EINVAL, "get_tuning: unknown knob %s", name);
    return IBL_OK;
}

int ibl_set_kernel_mode(int general_only) {
    read_tuning_once();
    g_tuning.no_octets = general_only != 0;
    return IBL_OK;
}

uint32_t ibl_ggx_mip_levels(uint32_t top_w, uint32_t top_h, uint32_t min_size) {
    if (top_w == 0 || top_h == 0 || min_size == 0) return 0;
    uint32_t m = top_w > top_h ? top_w : top_h;
    if (m < min_size) return 0;
    return ibl::ggx_mip_levels(top_w, top_h, min_size);
}

int ibl_ggx_level_params(uint32_t mip_level, uint32_t mip_levels, uint32_t in_w, uint32_t in_h,
                         float* roughness, float* alpha2, float* cutoff, uint32_t* patch_size) {
    if (mip_levels == 0 || mip_level == 0 || mip_level > mip_levels)
        return fail(IBL_EINVAL, "ggx level %u of %u is out of range", mip_level, mip_levels);
    ibl::GgxLevel lv = ibl::ggx_level_params(mip_level, mip_levels, in_w, in_h);
    if (roughness) *roughness = lv.roughness;
    if (alpha2) *alpha2 = lv.alpha2;
    if (cutoff) *cutoff = lv.cutoff;
    if (patch_size) *patch_size = lv.patch;
    return IBL_OK;
}

int ibl_ggx_level_plan(uint32_t in_w, uint32_t out_w, uint32_t mip_level, uint32_t mip_levels, uint32_t* runs,
                       uint32_t* passes, uint64_t* sums_bytes, uint64_t* records_bytes) {
    read_tuning_once();
    if (mip_levels == 0 || mip_level == 0 || mip_level > mip_levels)
        return fail(IBL_EINVAL, "ggx level %u of %u is out of range", mip_level, mip_levels);
    if (in_w < 16 || out_w == 0) return fail(IBL_EINVAL, "ggx_level_plan: faces %u -> %u", in_w, out_w);
    const ibl::GgxLevel lv = ibl::ggx_level_params(mip_level, mip_levels, in_w, in_w);
    if (g_tuning.no_octets || !ibl::ggx_octets_eligible(in_w, out_w, lv.patch))
        return fail(IBL_EINVAL, "ggx_level_plan: faces %u -> %u are served by the general kernel", in_w, out_w);
    const uint32_t n_oct = 3 * (out_w / 2) * (out_w / 2);
    const bool hex = g_tuning.ggx_hex != 0 && g_tuning.ggx_matrix != 0 && ibl::ggx_hex_eligible(in_w, out_w, lv.patch);
    const ibl::GgxOctPlan pl = hex ? ibl::ggx_hex_plan(in_w, out_w, mip_level, mip_levels, ibl::ggx_hex_units_total(out_w))
                                   : ibl::ggx_oct_plan(in_w, out_w, mip_level, mip_levels, (n_oct + ibl::GGX_OCT_UNIT - 1) / ibl::GGX_OCT_UNIT);
    if (runs) *runs = pl.n_runs;
    if (passes) *passes = pl.passes;
    if (sums_bytes) *sums_bytes = pl.partial_bytes;
    if (records_bytes) *records_bytes = pl.rec_bytes + pl.pn_bytes;
    return IBL_OK;
}

int ibl_set_timing(int enabled) {
    if (int rc = check_ready()) return rc;
    S->timing = enabled != 0;
    S->timed_valid = false;
    return IBL_OK;
}

int ibl_set_counting(int enabled) {
    if (int rc = check_ready()) return rc;
    S->counting = enabled != 0;
    return IBL_OK;
}

int ibl_last_kernel_ms(float* ms) {
    if (int rc = check_ready()) return rc;
    if (!ms) return fail(IBL_EINVAL, "ms is null");
    if (!S->timed_valid) return fail(IBL_EINVAL, "no timed kernel: call ibl_set_timing(1) before a *_dev call");
    return slot_ms((S->timed_calls - 1) % TIMING_RING, ms);
}

int ibl_kernel_ms_back(uint32_t calls_back, float* ms) {
    if (int rc = check_ready()) return rc;
    if (!ms) return fail(IBL_EINVAL, "ms is null");
    if (calls_back >= (uint32_t)TIMING_RING || calls_back >= S->timed_calls)
        return fail(IBL_EINVAL, "kernel_ms_back: %u calls back, but %llu timed calls so far and %d are kept",
                    calls_back, (unsigned long long)S->timed_calls, TIMING_RING);
    return slot_ms((S->timed_calls - 1 - calls_back) % TIMING_RING, ms);
}

int ibl_last_counts(uint64_t* patch_tests, uint64_t* tested, uint64_t* accepted) {
    if (int rc = check_ready()) return rc;
    if (S->counts_pending) {
        unsigned long long h[7];
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(h, S->counts.ptr, sizeof(h), hipMemcpyDeviceToHost));
        if (S->counts_pending == 1) { S->last_counts[1] = h[1]; S->last_counts[2] = h[2]; }   // GGX: texel tests
        S->last_counts[3] = h[3];
        S->last_counts[4] = h[4];
        S->last_counts[5] = h[5];
        S->last_counts[6] = h[6];
        S->counts_pending = 0;
    }
    if (patch_tests) *patch_tests = S->last_counts[0];
    if (tested) *tested = S->last_counts[1];
    if (accepted) *accepted = S->last_counts[2];
    return IBL_OK;
}

int ibl_last_wave_work(uint64_t* wave_records, uint64_t* wave_patterns) {
    if (int rc = ibl_last_counts(nullptr, nullptr, nullptr)) return rc;
    if (wave_records) *wave_records = S->last_counts[3];
    if (wave_patterns) *wave_patterns = S->last_counts[4];
    return IBL_OK;
}

int ibl_last_wave_visits(uint64_t* wave_visits) {
    if (int rc = ibl_last_counts(nullptr, nullptr, nullptr)) return rc;
    if (wave_visits) *wave_visits = S->last_counts[6];
    return IBL_OK;
}

int ibl_last_exact_octets(uint64_t* octets) {
    if (int rc = ibl_last_counts(nullptr, nullptr, nullptr)) return rc;
    if (octets) *octets = S->last_counts[5];
    return IBL_OK;
}

// ---- device-resident entry points ---------------------------------------------

namespace {
// The hexadecet form over the n_units units listed in S->units.
int lambertian_run_hex(const float* d_env, uint32_t in_w, uint32_t out_size, uint32_t n_units,
                       uint32_t out_begin, uint32_t out_count, float* d_out, hipStream_t s, bool every_unit = false) {
    const ibl::LambertianHexPlan hp = ibl::lambertian_hex_plan(in_w, out_size, n_units);
    if (int rc = ensure(S->a, hp.rec_bytes)) return rc;
    if (int rc = ensure(S->partial, hp.work_bytes)) return rc;
    unsigned long long* d_counts = nullptr;
    if (int rc = counts_buffer(s, &d_counts)) return rc;
    HIP_TRY(ibl::lambertian_hex_launch(hp, d_env, in_w, out_size, every_unit ? nullptr : (const uint32_t*)S->units.ptr, out_begin, out_count, d_out,
                                       S->a.ptr, S->partial.ptr, d_counts, s, S->timing ? S->ev0 : nullptr,
                                       S->timing ? S->ev1 : nullptr, S->timing ? S->ev2 : nullptr,
                                       S->timing ? S->ev3 : nullptr));
    S->counts_pending = S->counting ? 2 : 0;
    finish_timed_call(true);
    return IBL_OK;
}
// The 48-fold form over the n_units units listed in S->units.
int lambertian_run_o48(const float* d_env, uint32_t in_w, uint32_t out_size, uint32_t n_units,
                       uint32_t out_begin, uint32_t out_count, float* d_out, hipStream_t s, bool every_unit = false) {
    const ibl::LambertianO48Plan op = ibl::lambertian_o48_plan(in_w, out_size, n_units);
    if (int rc = ensure(S->a, op.rec_bytes)) return rc;
    if (int rc = ensure(S->partial, op.work_bytes)) return rc;
    unsigned long long* d_counts = nullptr;
    if (int rc = counts_buffer(s, &d_counts)) return rc;
    HIP_TRY(ibl::lambertian_o48_launch(op, d_env, in_w, out_size, every_unit ? nullptr : (const uint32_t*)S->units.ptr, out_begin, out_count, d_out,
                                       S->a.ptr, S->partial.ptr, d_counts, s, S->timing ? S->ev0 : nullptr,
                                       S->timing ? S->ev1 : nullptr, S->timing ? S->ev2 : nullptr,
                                       S->timing ? S->ev3 : nullptr));
    S->counts_pending = S->counting ? 2 : 0;
    finish_timed_call(true);
    return IBL_OK;
}
}  // namespace

int ibl_lambertian_dev(const float* d_env, uint32_t in_w, uint32_t in_h, uint32_t out_size,
                       uint32_t out_begin, uint32_t out_count, float* d_out, void* stream) {
    if (int rc = check_ready()) return rc;
    if (!d_env || !d_out) return fail(IBL_EINVAL, "lambertian: null pointer");
    if (int rc = check_cube(in_w, in_h, "lambertian input")) return rc;
    if (int rc = check_cube(out_size, out_size, "lambertian output")) return rc;
    const uint64_t n_out = (uint64_t)out_size * out_size * 6;
    if (int rc = check_range(out_begin, out_count, n_out, "lambertian")) return rc;
    const uint32_t n_in = in_w * in_h * 6;
    S->last_counts[0] = 0;
    S->last_counts[3] = S->last_counts[4] = S->last_counts[5] = 0;
    S->last_counts[1] = (uint64_t)out_count * n_in;
    S->last_counts[2] = 0;
    S->counts_pending = 0;
    if (out_count == 0) { S->timed_valid = false; return IBL_OK; }
    hipStream_t s = pick(stream);
    if (use_lamb_o48(in_w, out_size)) {
        // the units of 64 orbits that hold a texel of the range; the other members are computed and not stored
        const uint32_t total = ibl::lambertian_o48_units_total(out_size);
        if (out_begin == 0 && out_count == n_out)          // the whole cube: every unit, no list
            return lambertian_run_o48(d_env, in_w, out_size, total, out_begin, out_count, d_out, s, true);
        if (int rc = ensure(S->units, (size_t)total * sizeof(uint32_t))) return rc;
        const uint32_t n_units = ibl::lambertian_o48_units_in_range_count(out_size, out_begin, out_count);
        HIP_TRY(ibl::lambertian_o48_units_in_range_launch(out_size, out_begin, out_count, (uint32_t*)S->units.ptr, s));
        if (n_units == 0) { S->timed_valid = false; return IBL_OK; }
        return lambertian_run_o48(d_env, in_w, out_size, n_units, out_begin, out_count, d_out, s);
    }
    if (use_lamb_hex(in_w, out_size)) {
        // the units of 64 hexadecets that hold a texel of the range; the other members are computed and not stored
        const uint32_t total = ibl::ggx_hex_units_total(out_size);
        if (out_begin == 0 && out_count == n_out)          // the whole cube: every unit, no list
            return lambertian_run_hex(d_env, in_w, out_size, total, out_begin, out_count, d_out, s, true);
        if (int rc = ensure(S->units, (size_t)total * sizeof(uint32_t))) return rc;
        const uint32_t n_units = ibl::ggx_hex_units_in_range_count(out_size, out_begin, out_count);
        HIP_TRY(ibl::ggx_hex_units_in_range_launch(out_size, out_begin, out_count, (uint32_t*)S->units.ptr, s));
        if (n_units == 0) { S->timed_valid = false; return IBL_OK; }
        return lambertian_run_hex(d_env, in_w, out_size, n_units, out_begin, out_count, d_out, s);
    }
    if (use_octets(in_w, out_size)) {
        // mirror-octet kernel over the 256-octet blocks that hold a texel of the range; the
        // other members of those octets are computed and not stored
        const uint32_t n_oct = 3 * (out_size / 2) * (out_size / 2);
        const uint32_t* d_list = nullptr;
        uint32_t n_blocks = 0;
        if (int rc = units_for_range(ibl::LAMBERTIAN_OCT_BLOCK, ibl::lambertian_oct_tile_edge(out_size), out_size,
                                     out_begin, out_count, s, &d_list, &n_blocks)) return rc;
        if (n_blocks == 0) { S->timed_valid = false; return IBL_OK; }
        ibl::LambertianOctPlan pl = ibl::lambertian_oct_plan(in_w, n_blocks);
        if (int rc = ensure(S->a, pl.rec_bytes)) return rc;
        if (int rc = ensure(S->partial, pl.partial_bytes)) return rc;
        unsigned long long* d_counts = nullptr;
        if (int rc = counts_buffer(s, &d_counts)) return rc;
        HIP_TRY(ibl::lambertian_oct_launch(pl, d_env, in_w, out_size, d_list, 0, n_oct, out_begin, out_count, d_out,
                                           S->a.ptr, (double*)S->partial.ptr, d_counts, s, S->timing ? S->ev0 : nullptr,
                                           S->timing ? S->ev1 : nullptr, S->timing ? S->ev2 : nullptr,
                                           S->timing ? S->ev3 : nullptr));
        S->counts_pending = S->counting ? 2 : 0;
        finish_timed_call(pl.split);
        return IBL_OK;
    } else {
        ibl::LambertianPlan pl = ibl::lambertian_plan(n_in, out_count);
        if (int rc = ensure(S->a, pl.prep_bytes_a)) return rc;
        if (int rc = ensure(S->b, pl.prep_bytes_b)) return rc;
        if (int rc = ensure(S->partial, pl.partial_bytes)) return rc;
        HIP_TRY(ibl::lambertian_launch(pl, d_env, in_w, in_h, out_size, out_begin, out_count, d_out, S->a.ptr, S->b.ptr,
                                       (double*)S->partial.ptr, s, S->timing ? S->ev0 : nullptr,
                                       S->timing ? S->ev1 : nullptr));
    }
    finish_timed_call();
    return IBL_OK;
}

int ibl_lambertian_shard_dev(const float* d_env, uint32_t in_w, uint32_t in_h, uint32_t out_size,
                             uint32_t shard, uint32_t n_shards, float* d_out, void* stream) {
    if (int rc = check_ready()) return rc;
    if (!d_env || !d_out) return fail(IBL_EINVAL, "lambertian: null pointer");
    if (int rc = check_cube(in_w, in_h, "lambertian input")) return rc;
    if (int rc = check_cube(out_size, out_size, "lambertian output")) return rc;
    if (n_shards == 0 || shard >= n_shards) return fail(IBL_EINVAL, "lambertian: shard %u of %u", shard, n_shards);
    const uint32_t n_out = out_size * out_size * 6;
    if (!use_octets(in_w, out_size)) {
        // no mirror symmetry to keep together: shards are plain (face, y, x) ranges
        const uint32_t per = (n_out + n_shards - 1) / n_shards;
        const uint32_t begin = shard * per < n_out ? shard * per : n_out;
        const uint32_t count = per < n_out - begin ? per : n_out - begin;
        return ibl_lambertian_dev(d_env, in_w, in_h, out_size, begin, count, d_out, stream);
    }
    if (use_lamb_o48(in_w, out_size)) {
        // every n_shards-th unit of 64 orbits
        const uint32_t units = ibl::lambertian_o48_units_total(out_size);
        const uint32_t mine = shard < units ? (units - shard + n_shards - 1) / n_shards : 0;
        S->last_counts[0] = 0;
        S->last_counts[3] = S->last_counts[4] = S->last_counts[5] = 0;
        S->last_counts[1] = (uint64_t)mine * 64 * 48 * in_w * in_h * 6;
        S->last_counts[2] = 0;
        S->counts_pending = 0;
        if (mine == 0) { S->timed_valid = false; return IBL_OK; }
        if (n_shards == 1) return lambertian_run_o48(d_env, in_w, out_size, mine, 0, n_out, d_out, pick(stream), true);
        if (int rc = ensure(S->units, (size_t)mine * sizeof(uint32_t))) return rc;
        HIP_TRY(ibl::octet_units_strided_launch(shard, n_shards, mine, (uint32_t*)S->units.ptr, pick(stream)));
        return lambertian_run_o48(d_env, in_w, out_size, mine, 0, n_out, d_out, pick(stream));
    }
    if (use_lamb_hex(in_w, out_size)) {
        // every n_shards-th unit of 64 hexadecets (the partition of the 16-fold GGX kernel)
        const uint32_t units = ibl::ggx_hex_units_total(out_size);
        const uint32_t mine = shard < units ? (units - shard + n_shards - 1) / n_shards : 0;
        S->last_counts[0] = 0;
        S->last_counts[3] = S->last_counts[4] = S->last_counts[5] = 0;
        S->last_counts[1] = (uint64_t)mine * 64 * 16 * in_w * in_h * 6;
        S->last_counts[2] = 0;
        S->counts_pending = 0;
        if (mine == 0) { S->timed_valid = false; return IBL_OK; }
        if (n_shards == 1) return lambertian_run_hex(d_env, in_w, out_size, mine, 0, n_out, d_out, pick(stream), true);
        if (int rc = ensure(S->units, (size_t)mine * sizeof(uint32_t))) return rc;
        HIP_TRY(ibl::octet_units_strided_launch(shard, n_shards, mine, (uint32_t*)S->units.ptr, pick(stream)));
        return lambertian_run_hex(d_env, in_w, out_size, mine, 0, n_out, d_out, pick(stream));
    }
    const uint32_t n_oct = 3 * (out_size / 2) * (out_size / 2);
    const uint32_t per = (n_oct + n_shards - 1) / n_shards;
    const uint32_t begin = shard * per < n_oct ? shard * per : n_oct;
    const uint32_t count = per < n_oct - begin ? per : n_oct - begin;
    S->last_counts[0] = 0;
    S->last_counts[3] = S->last_counts[4] = S->last_counts[5] = 0;
    S->last_counts[1] = (uint64_t)count * 8 * in_w * in_h * 6;
    S->last_counts[2] = 0;
    S->counts_pending = 0;
    if (count == 0) { S->timed_valid = false; return IBL_OK; }
    ibl::LambertianOctPlan pl = ibl::lambertian_oct_plan(in_w, (count + ibl::LAMBERTIAN_OCT_BLOCK - 1) / ibl::LAMBERTIAN_OCT_BLOCK);
    if (int rc = ensure(S->a, pl.rec_bytes)) return rc;
    if (int rc = ensure(S->partial, pl.partial_bytes)) return rc;
    unsigned long long* d_counts = nullptr;
    if (int rc = counts_buffer(pick(stream), &d_counts)) return rc;
    HIP_TRY(ibl::lambertian_oct_launch(pl, d_env, in_w, out_size, nullptr, begin, begin + count, 0, n_out, d_out,
                                       S->a.ptr, (double*)S->partial.ptr, d_counts, pick(stream),
                                       S->timing ? S->ev0 : nullptr, S->timing ? S->ev1 : nullptr,
                                       S->timing ? S->ev2 : nullptr, S->timing ? S->ev3 : nullptr));
    S->counts_pending = S->counting ? 2 : 0;
    finish_timed_call(pl.split);
    return IBL_OK;
}

int ibl_lambertian_static_dev(const float* d_env, uint32_t in_w, uint32_t in_h, uint32_t out_size, float* d_out,
                              void* stream) {
    if (int rc = check_ready()) return rc;
    if (!d_env || !d_out) return fail(IBL_EINVAL, "lambertian_static: null pointer");
    if (int rc = check_cube(in_w, in_h, "lambertian_static input")) return rc;
    if (int rc = check_cube(out_size, out_size, "lambertian_static output")) return rc;
    if (!use_octets(in_w, out_size))
        return fail(IBL_EINVAL, "lambertian_static: face sizes %u -> %u are not served by the mirror-octet kernel "
                    "(both must be even and exactly mirror-symmetric, as every power of two is)", in_w, out_size);
    const uint32_t n_oct = 3 * (out_size / 2) * (out_size / 2);
    const uint32_t n_out = out_size * out_size * 6;
    if (use_lamb_o48(in_w, out_size)) {
        // the passes ibl_lambertian_dev runs for these sizes, over every unit of the cube (no list)
        const ibl::LambertianO48Plan op = ibl::lambertian_o48_plan(in_w, out_size, ibl::lambertian_o48_units_total(out_size));
        const size_t rec_bytes = (op.rec_bytes + 255) & ~(size_t)255;
        if (rec_bytes + op.work_bytes > STATIC_SCRATCH_BYTES)
            return fail(IBL_EINVAL, "lambertian_static: %u x %u/face -> %u x %u/face needs %zu bytes of scratch, the "
                        "static array holds %zu (e.g. 512 -> 32, 256 -> 64, 128 -> 64 fit); use ibl_lambertian_dev",
                        in_w, in_w, out_size, out_size, rec_bytes + op.work_bytes, STATIC_SCRATCH_BYTES);
        S->last_counts[0] = 0;
        S->last_counts[3] = S->last_counts[4] = S->last_counts[5] = 0;
        S->last_counts[1] = (uint64_t)n_out * in_w * in_h * 6;
        S->last_counts[2] = 0;
        S->counts_pending = 0;
        unsigned char* base = (unsigned char*)S->static_scratch;
        HIP_TRY(ibl::lambertian_o48_launch(op, d_env, in_w, out_size, nullptr, 0, n_out, d_out, base, base + rec_bytes, nullptr,
                                           pick(stream), S->timing ? S->ev0 : nullptr, S->timing ? S->ev1 : nullptr,
                                           S->timing ? S->ev2 : nullptr, S->timing ? S->ev3 : nullptr));
        finish_timed_call(true);
        return IBL_OK;
    }
    if (use_lamb_hex(in_w, out_size)) {
        // the passes ibl_lambertian_dev runs for these sizes, over every unit of the cube (no list)
        const ibl::LambertianHexPlan hp = ibl::lambertian_hex_plan(in_w, out_size, ibl::ggx_hex_units_total(out_size));
        const size_t rec_bytes = (hp.rec_bytes + 255) & ~(size_t)255;
        if (rec_bytes + hp.work_bytes > STATIC_SCRATCH_BYTES)
            return fail(IBL_EINVAL, "lambertian_static: %u x %u/face -> %u x %u/face needs %zu bytes of scratch, the "
                        "static array holds %zu (e.g. 512 -> 32, 256 -> 64, 128 -> 64 fit); use ibl_lambertian_dev",
                        in_w, in_w, out_size, out_size, rec_bytes + hp.work_bytes, STATIC_SCRATCH_BYTES);
        S->last_counts[0] = 0;
        S->last_counts[3] = S->last_counts[4] = S->last_counts[5] = 0;
        S->last_counts[1] = (uint64_t)n_out * in_w * in_h * 6;
        S->last_counts[2] = 0;
        S->counts_pending = 0;
        unsigned char* base = (unsigned char*)S->static_scratch;
        HIP_TRY(ibl::lambertian_hex_launch(hp, d_env, in_w, out_size, nullptr, 0, n_out, d_out, base, base + rec_bytes, nullptr,
                                           pick(stream), S->timing ? S->ev0 : nullptr, S->timing ? S->ev1 : nullptr,
                                           S->timing ? S->ev2 : nullptr, S->timing ? S->ev3 : nullptr));
        finish_timed_call(true);
        return IBL_OK;
    }
    ibl::LambertianOctPlan pl = ibl::lambertian_oct_plan(in_w, ibl::lambertian_oct_blocks_total(out_size));
    // The same two passes as ibl_lambertian_dev; the exact-weights pass in its run form, which
    // stores an eighth of the partial sums and gives the same bits as the slice form.
    pl.run = pl.n_slices > 1;
    pl.partial_bytes = ibl::lambertian_oct_partial_bytes(pl);
    const size_t rec_bytes = (pl.rec_bytes + 255) & ~(size_t)255;
    if (rec_bytes + pl.partial_bytes > STATIC_SCRATCH_BYTES)
        return fail(IBL_EINVAL, "lambertian_static: %u x %u/face -> %u x %u/face needs %zu bytes of scratch, the "
                    "static array holds %zu (e.g. 512 -> 32, 256 -> 64, 128 -> 64 fit); use ibl_lambertian_dev",
                    in_w, in_w, out_size, out_size, rec_bytes + pl.partial_bytes, STATIC_SCRATCH_BYTES);
    S->last_counts[0] = 0;
    S->last_counts[3] = S->last_counts[4] = S->last_counts[5] = 0;
    S->last_counts[1] = (uint64_t)n_out * in_w * in_h * 6;
    S->last_counts[2] = 0;
    S->counts_pending = 0;
    unsigned char* base = (unsigned char*)S->static_scratch;
    HIP_TRY(ibl::lambertian_oct_launch(pl, d_env, in_w, out_size, nullptr, 0, n_oct, 0, n_out, d_out, base,
                                       (double*)(base + rec_bytes), nullptr, pick(stream), S->timing ? S->ev0 : nullptr,
                                       S->timing ? S->ev1 : nullptr, S->timing ? S->ev2 : nullptr,
                                       S->timing ? S->ev3 : nullptr));
    finish_timed_call(pl.split);
    return IBL_OK;
}

namespace {

int ggx_check(const float* d_env, uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h,
              uint32_t mip_level, uint32_t mip_levels, const float* d_out, uint32_t* patch) {
    if (!d_env || !d_out) return fail(IBL_EINVAL, "ggx: null pointer");
    if (int rc = check_cube(in_w, in_h, "ggx input")) return rc;
    if (int rc = check_cube(out_w, out_h, "ggx output")) return rc;
    if (in_w < 16)
        return fail(IBL_EINVAL, "ggx: input faces must be at least 16x16 (patch size would be 0), got %u", in_w);
    if (mip_levels == 0 || mip_level == 0 || mip_level > mip_levels)
        return fail(IBL_EINVAL, "ggx: level %u of %u is out of range", mip_level, mip_levels);
    ibl::GgxLevel lv = ibl::ggx_level_params(mip_level, mip_levels, in_w, in_h);
    // in_w need not be a multiple of the patch size.  The library then defines the result itself: a
    // patch grid of (in_w / patch)^2 per face, the last in_w % patch columns and rows of a face left
    // out of the quadrature (general kernel).  The reference is undefined for such widths: its patch
    // table is indexed with (face*H/ps + y)*W/ps + x, divisions left to right (cube.cpp:182,196,253),
    // which reads outside the table.
    *patch = lv.patch;
    return IBL_OK;
}

bool use_ggx_octets(uint32_t in_w, uint32_t out_w, uint32_t patch) {
    if (g_tuning.no_octets) return false;
    return ibl::ggx_octets_eligible(in_w, out_w, patch);
}

// The 16-fold kernel serves the level: mirror-exact sizes, output faces a multiple of 16 wide, the matrix form
// switched on (it has no vector form of its own).
bool use_ggx_hex(uint32_t in_w, uint32_t out_w, uint32_t patch) {
    if (g_tuning.no_octets || g_tuning.ggx_hex == 0 || g_tuning.ggx_matrix == 0) return false;
    return ibl::ggx_hex_eligible(in_w, out_w, patch);
}

int ggx_run_octets(const float* d_env, uint32_t in_w, uint32_t out_w, uint32_t mip_level, uint32_t mip_levels,
                   uint32_t n_units, uint32_t out_begin, uint32_t out_count, float* d_out, hipStream_t s);

// The 16-fold kernel over the n_units units of 64 hexadecets listed in S->units.  When counting, the octet
// kernel's counting variant runs first over the octet units listed behind them (n_oct_units; its counts of
// texel tests are the reference's), and the records this kernel walks replace that variant's.
int ggx_run_hex(const float* d_env, uint32_t in_w, uint32_t out_w, uint32_t mip_level, uint32_t mip_levels,
                uint32_t n_units, uint32_t out_begin, uint32_t out_count, float* d_out, hipStream_t s) {
    ibl::GgxOctPlan pl = ibl::ggx_hex_plan(in_w, out_w, mip_level, mip_levels, n_units);
    if (pl.out_units == 0) { S->timed_valid = false; return IBL_OK; }
    if (int rc = ensure(S->a, pl.rec_bytes)) return rc;
    if (int rc = ensure(S->pn, pl.pn_bytes)) return rc;
    if (int rc = ensure(S->partial, pl.partial_bytes)) return rc;
    unsigned long long* d_counts = nullptr;
    if (S->counting) {
        d_counts = (unsigned long long*)S->counts.ptr;          // (the counting pass before this call made and zeroed it)
        HIP_TRY(hipMemsetAsync(d_counts + 3, 0, 2 * sizeof(unsigned long long), s));
    }
    HIP_TRY(ibl::ggx_hex_launch(pl, d_env, in_w, out_w, (const uint32_t*)S->units.ptr, out_begin, out_count, d_out,
                                S->a.ptr, S->pn.ptr, (double*)S->partial.ptr, d_counts, s, S->timing ? S->ev0 : nullptr,
                                S->timing ? S->ev1 : nullptr));
    finish_timed_call();
    S->counts_pending = S->counting ? 1 : 0;
    return IBL_OK;
}

// Octet kernel over the n_units 64-octet units listed in S->units, writing the member texels
// inside [out_begin, out_begin+out_count).
int ggx_run_octets(const float* d_env, uint32_t in_w, uint32_t out_w, uint32_t mip_level, uint32_t mip_levels,
                   uint32_t n_units, uint32_t out_begin, uint32_t out_count, float* d_out, hipStream_t s) {
    ibl::GgxOctPlan pl = ibl::ggx_oct_plan(in_w, out_w, mip_level, mip_levels, n_units);
    if (pl.out_units == 0) { S->timed_valid = false; return IBL_OK; }
    if (int rc = ensure(S->a, pl.rec_bytes)) return rc;
    if (int rc = ensure(S->pn, pl.pn_bytes)) return rc;
    if (int rc = ensure(S->partial, pl.partial_bytes)) return rc;
    unsigned long long* d_counts = nullptr;
    if (int rc = counts_buffer(s, &d_counts)) return rc;
    HIP_TRY(ibl::ggx_oct_launch(pl, d_env, in_w, out_w, (const uint32_t*)S->units.ptr, out_begin, out_count, d_out,
                                S->a.ptr, S->pn.ptr, (double*)S->partial.ptr, d_counts, s, S->timing ? S->ev0 : nullptr,
                                S->timing ? S->ev1 : nullptr));
    finish_timed_call();
    S->counts_pending = S->counting ? 1 : 0;
    return IBL_OK;
}

}  // namespace

int ibl_ggx_level_dev(const float* d_env, uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h,
                      uint32_t mip_level, uint32_t mip_levels,
                      uint32_t out_begin, uint32_t out_count, float* d_out, void* stream) {
    if (int rc = check_ready()) return rc;
    uint32_t patch = 0;
    if (int rc = ggx_check(d_env, in_w, in_h, out_w, out_h, mip_level, mip_levels, d_out, &patch)) return rc;
    const uint64_t n_out = (uint64_t)out_w * out_h * 6;
    if (int rc = check_range(out_begin, out_count, n_out, "ggx")) return rc;
    const uint32_t n_patches = 6 * (in_w / patch) * (in_h / patch);
    S->last_counts[0] = (uint64_t)out_count * n_patches;
    S->last_counts[3] = S->last_counts[4] = S->last_counts[5] = 0;
    S->last_counts[1] = S->last_counts[2] = 0;
    S->counts_pending = 0;
    if (out_count == 0) { S->timed_valid = false; return IBL_OK; }
    hipStream_t s = pick(stream);
    const bool hex = use_ggx_hex(in_w, out_w, patch);
    if (use_ggx_octets(in_w, out_w, patch) && (!hex || S->counting)) {
        // the 64-octet units that hold a texel of the range; the other members of their octets
        // are computed and not stored
        // (counting with the 16-fold kernel: the octet kernel's counting variant first, whose counts of
        // texel tests are the reference's; the 16-fold kernel then overwrites the texels and counts the
        // records it walks itself)
        const uint32_t n_oct = 3 * (out_w / 2) * (out_w / 2), total = (n_oct + ibl::GGX_OCT_UNIT - 1) / ibl::GGX_OCT_UNIT;
        uint32_t n_units = total;
        if (int rc = ensure(S->units, (size_t)total * sizeof(uint32_t))) return rc;
        if (out_begin == 0 && out_count == n_out) {
            HIP_TRY(ibl::octet_units_strided_launch(0, 1, total, (uint32_t*)S->units.ptr, s));
        } else {
            const uint32_t tw = ibl::ggx_oct_tile_edge(out_w);
            n_units = ibl::octet_units_in_range_count(ibl::GGX_OCT_UNIT, out_w, tw, out_begin, out_count);
            HIP_TRY(ibl::octet_units_in_range_launch(ibl::GGX_OCT_UNIT, out_w, tw, out_begin, out_count,
                                                     (uint32_t*)S->units.ptr, s));
        }
        const int rc = ggx_run_octets(d_env, in_w, out_w, mip_level, mip_levels, n_units, out_begin, out_count, d_out, s);
        if (rc || !hex) return rc;
    }
    if (hex) {
        const uint32_t total = ibl::ggx_hex_units_total(out_w);
        uint32_t n_units = total;
        if (int rc = ensure(S->units, (size_t)total * sizeof(uint32_t))) return rc;
        if (out_begin == 0 && out_count == n_out) {
            HIP_TRY(ibl::octet_units_strided_launch(0, 1, total, (uint32_t*)S->units.ptr, s));
        } else {
            n_units = ibl::ggx_hex_units_in_range_count(out_w, out_begin, out_count);
            HIP_TRY(ibl::ggx_hex_units_in_range_launch(out_w, out_begin, out_count, (uint32_t*)S->units.ptr, s));
        }
        return ggx_run_hex(d_env, in_w, out_w, mip_level, mip_levels, n_units, out_begin, out_count, d_out, s);
    }
    ibl::GgxPlan pl = ibl::ggx_plan(in_w, in_h, out_w, out_h, mip_level, mip_levels, out_count);
    if (int rc = ensure(S->a, pl.prep_bytes_a)) return rc;
    if (int rc = ensure(S->b, pl.prep_bytes_b)) return rc;
    if (int rc = ensure(S->pn, pl.prep_bytes_pn)) return rc;
    if (int rc = ensure(S->partial, pl.partial_bytes)) return rc;
    unsigned long long* d_counts = nullptr;
    if (int rc = counts_buffer(s, &d_counts)) return rc;
    HIP_TRY(ibl::ggx_launch(pl, d_env, in_w, in_h, out_w, out_h, out_begin, out_count, d_out, S->a.ptr, S->b.ptr,
                            S->pn.ptr, (double*)S->partial.ptr, d_counts, s, S->timing ? S->ev0 : nullptr,
                            S->timing ? S->ev1 : nullptr));
    finish_timed_call();
    S->counts_pending = S->counting ? 1 : 0;
    return IBL_OK;
}

int ibl_ggx_level_shard_dev(const float* d_env, uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h,
                            uint32_t mip_level, uint32_t mip_levels,
                            uint32_t shard, uint32_t n_shards, float* d_out, void* stream) {
    if (int rc = check_ready()) return rc;
    uint32_t patch = 0;
    if (int rc = ggx_check(d_env, in_w, in_h, out_w, out_h, mip_level, mip_levels, d_out, &patch)) return rc;
    if (n_shards == 0 || shard >= n_shards) return fail(IBL_EINVAL, "ggx: shard %u of %u", shard, n_shards);
    const uint32_t n_out = out_w * out_h * 6;
    if (!use_ggx_octets(in_w, out_w, patch)) {
        const uint32_t per = (n_out + n_shards - 1) / n_shards;
        const uint32_t begin = shard * per < n_out ? shard * per : n_out;
        const uint32_t count = per < n_out - begin ? per : n_out - begin;
        return ibl_ggx_level_dev(d_env, in_w, in_h, out_w, out_h, mip_level, mip_levels, begin, count, d_out, stream);
    }
    // GGX cost per output tile varies over the cube (patches near a face centre subtend
    // more solid angle), so shards take every n_shards-th unit of 64 octets (one wave's worth).
    const uint32_t n_patches = 6 * (in_w / patch) * (in_h / patch);
    const uint32_t n_oct = 3 * (out_w / 2) * (out_w / 2);
    const uint32_t n_units = (n_oct + 63) / 64;
    const uint32_t mine = shard < n_units ? (n_units - shard + n_shards - 1) / n_shards : 0;
    const bool hex = use_ggx_hex(in_w, out_w, patch);
    S->last_counts[0] = (uint64_t)mine * 64 * 8 * n_patches;
    S->last_counts[3] = S->last_counts[4] = S->last_counts[5] = 0;
    S->last_counts[1] = S->last_counts[2] = 0;
    S->counts_pending = 0;
    if (!hex || S->counting) {
        // (counting with the 16-fold kernel: the counting variant of the octet kernel on the octet shard first --
        // its texels are other texels than the hexadecet shard's, the shards' counts add up to the level's all the same)
        if (mine == 0) {
            S->timed_valid = false;
            if (!hex) return IBL_OK;
            unsigned long long* zeroed = nullptr;          // (no octet unit to count on: the counters start at 0)
            if (int rc = counts_buffer(pick(stream), &zeroed)) return rc;
        } else {
            if (int rc = ensure(S->units, (size_t)mine * sizeof(uint32_t))) return rc;
            HIP_TRY(ibl::octet_units_strided_launch(shard, n_shards, mine, (uint32_t*)S->units.ptr, pick(stream)));
            const int rc = ggx_run_octets(d_env, in_w, out_w, mip_level, mip_levels, mine, 0, n_out, d_out, pick(stream));
            if (rc || !hex) return rc;
        }
    }
    // the same with units of 64 hexadecets
    const uint32_t hex_units = ibl::ggx_hex_units_total(out_w);
    const uint32_t hex_mine = shard < hex_units ? (hex_units - shard + n_shards - 1) / n_shards : 0;
    if (hex_mine == 0) { S->timed_valid = false; return IBL_OK; }
    if (int rc = ensure(S->units, (size_t)hex_mine * sizeof(uint32_t))) return rc;
    HIP_TRY(ibl::octet_units_strided_launch(shard, n_shards, hex_mine, (uint32_t*)S->units.ptr, pick(stream)));
    return ggx_run_hex(d_env, in_w, out_w, mip_level, mip_levels, hex_mine, 0, n_out, d_out, pick(stream));
}

// Sizes the current device's workspaces for whole-cube Lambertian calls in_w -> lambertian_size
// (0: none) and for every level of the GGX chain of a ggx_top_w face with minimum size
// ggx_min_size computed from an in_w input (ggx_top_w 0: none), so that later *_dev calls of
// those shapes find them large enough and neither synchronise nor allocate.
int ibl_reserve(uint32_t in_w, uint32_t lambertian_size, uint32_t ggx_top_w, uint32_t ggx_min_size) {
    if (int rc = check_ready()) return rc;
    if (int rc = check_cube(in_w, in_w, "reserve input")) return rc;
    const uint32_t n_in = in_w * in_w * 6;
    if (lambertian_size) {
        if (int rc = check_cube(lambertian_size, lambertian_size, "reserve lambertian output")) return rc;
        if (use_lamb_o48(in_w, lambertian_size)) {
            const uint32_t units = ibl::lambertian_o48_units_total(lambertian_size);
            const ibl::LambertianO48Plan op = ibl::lambertian_o48_plan(in_w, lambertian_size, units);
            if (int rc = ensure(S->a, op.rec_bytes)) return rc;
            if (int rc = ensure(S->partial, op.work_bytes)) return rc;
            if (int rc = ensure(S->units, (size_t)units * sizeof(uint32_t))) return rc;
        } else if (use_lamb_hex(in_w, lambertian_size)) {
            const uint32_t units = ibl::ggx_hex_units_total(lambertian_size);
            const ibl::LambertianHexPlan hp = ibl::lambertian_hex_plan(in_w, lambertian_size, units);
            if (int rc = ensure(S->a, hp.rec_bytes)) return rc;
            if (int rc = ensure(S->partial, hp.work_bytes)) return rc;
            if (int rc = ensure(S->units, (size_t)units * sizeof(uint32_t))) return rc;
        } else if (use_octets(in_w, lambertian_size)) {
            ibl::LambertianOctPlan pl = ibl::lambertian_oct_plan(in_w, ibl::lambertian_oct_blocks_total(lambertian_size));
            if (int rc = ensure(S->a, pl.rec_bytes)) return rc;
            if (int rc = ensure(S->partial, pl.partial_bytes)) return rc;
            if (int rc = ensure(S->units, (size_t)pl.out_blocks * sizeof(uint32_t))) return rc;
        } else {
            ibl::LambertianPlan pl = ibl::lambertian_plan(n_in, lambertian_size * lambertian_size * 6);
            if (int rc = ensure(S->a, pl.prep_bytes_a)) return rc;
            if (int rc = ensure(S->b, pl.prep_bytes_b)) return rc;
            if (int rc = ensure(S->partial, pl.partial_bytes)) return rc;
        }
    }
    if (ggx_top_w) {
        if (in_w < 16) return fail(IBL_EINVAL, "ggx: input faces must be at least 16x16 (patch size would be 0), got %u", in_w);
        const uint32_t mips = ibl_ggx_mip_levels(ggx_top_w, ggx_top_w, ggx_min_size);
        for (uint32_t m = 1; m <= mips; m++) {
            const uint32_t ow = ggx_top_w >> m;
            if (ow == 0) break;
            const ibl::GgxLevel lv = ibl::ggx_level_params(m, mips, in_w, in_w);
            if (use_ggx_hex(in_w, ow, lv.patch) && !S->counting) {
                const uint32_t units = ibl::ggx_hex_units_total(ow);
                ibl::GgxOctPlan pl = ibl::ggx_hex_plan(in_w, ow, m, mips, units);
                if (int rc = ensure(S->a, pl.rec_bytes)) return rc;
                if (int rc = ensure(S->pn, pl.pn_bytes)) return rc;
                if (int rc = ensure(S->partial, pl.partial_bytes)) return rc;
                if (int rc = ensure(S->units, (size_t)units * sizeof(uint32_t))) return rc;
            } else if (!g_tuning.no_octets && ibl::ggx_octets_eligible(in_w, ow, lv.patch)) {
                const uint32_t n_oct = 3 * (ow / 2) * (ow / 2), units = (n_oct + ibl::GGX_OCT_UNIT - 1) / ibl::GGX_OCT_UNIT;
                ibl::GgxOctPlan pl = ibl::ggx_oct_plan(in_w, ow, m, mips, units);
                if (int rc = ensure(S->a, pl.rec_bytes)) return rc;
                if (int rc = ensure(S->pn, pl.pn_bytes)) return rc;
                if (int rc = ensure(S->partial, pl.partial_bytes)) return rc;
                if (int rc = ensure(S->units, (size_t)units * sizeof(uint32_t))) return rc;
            } else {
                ibl::GgxPlan pl = ibl::ggx_plan(in_w, in_w, ow, ow, m, mips, ow * ow * 6);
                if (int rc = ensure(S->a, pl.prep_bytes_a)) return rc;
                if (int rc = ensure(S->b, pl.prep_bytes_b)) return rc;
                if (int rc = ensure(S->pn, pl.prep_bytes_pn)) return rc;
                if (int rc = ensure(S->partial, pl.partial_bytes)) return rc;
            }
        }
    }
    if (S->counting) if (int rc = ensure(S->counts, 8 * sizeof(unsigned long long))) return rc;
    return IBL_OK;
}

// ---- sharding -------------------------------------------------------------------

namespace {

// The partition the shard entry points use for this cube (it follows the kernel family that
// serves the sizes).
int shard_map(int kind, uint32_t in_w, uint32_t out_w, uint32_t mip_level, uint32_t mip_levels, uint32_t n_shards,
              ibl::ShardMap* m) {
    read_tuning_once();
    if (kind != IBL_SHARD_LAMBERTIAN && kind != IBL_SHARD_GGX) return fail(IBL_EINVAL, "shard: unknown kind %d", kind);
    if (int rc = check_cube(in_w, in_w, "shard input")) return rc;
    if (int rc = check_cube(out_w, out_w, "shard output")) return rc;
    if (n_shards == 0) return fail(IBL_EINVAL, "shard: no shards");
    m->out_w = out_w;
    m->n_shards = n_shards;
    m->partition = ibl::SHARD_RANGE;
    m->tw = 1;
    if (kind == IBL_SHARD_LAMBERTIAN) {
        if (use_lamb_o48(in_w, out_w)) m->partition = ibl::SHARD_LAMB_O48;
        else if (use_lamb_hex(in_w, out_w)) m->partition = ibl::SHARD_GGX_HEX;
        else if (use_octets(in_w, out_w)) { m->partition = ibl::SHARD_LAMB_OCT; m->tw = ibl::lambertian_oct_tile_edge(out_w); }
    } else {
        if (in_w < 16) return fail(IBL_EINVAL, "ggx: input faces must be at least 16x16 (patch size would be 0), got %u", in_w);
        if (mip_levels == 0 || mip_level == 0 || mip_level > mip_levels)
            return fail(IBL_EINVAL, "ggx: level %u of %u is out of range", mip_level, mip_levels);
        const ibl::GgxLevel lv = ibl::ggx_level_params(mip_level, mip_levels, in_w, in_w);
        if (use_ggx_hex(in_w, out_w, lv.patch)) {
            m->partition = ibl::SHARD_GGX_HEX;
        } else if (!g_tuning.no_octets && ibl::ggx_octets_eligible(in_w, out_w, lv.patch)) {
            m->partition = ibl::SHARD_GGX_OCT;
            m->tw = ibl::ggx_oct_tile_edge(out_w);
        }
    }
    return IBL_OK;
}

}  // namespace

uint32_t ibl_shard_texels(int kind, uint32_t in_w, uint32_t out_w, uint32_t mip_level, uint32_t mip_levels,
                          uint32_t n_shards) {
    ibl::ShardMap m;
    if (shard_map(kind, in_w, out_w, mip_level, mip_levels, n_shards, &m)) return 0;
    return ibl::shard_slots(m);
}

int ibl_shard_indices(int kind, uint32_t in_w, uint32_t out_w, uint32_t mip_level, uint32_t mip_levels,
                      uint32_t shard, uint32_t n_shards, uint32_t* indices) {
    ibl::ShardMap m;
    if (int rc = shard_map(kind, in_w, out_w, mip_level, mip_levels, n_shards, &m)) return rc;
    if (shard >= n_shards) return fail(IBL_EINVAL, "shard %u of %u", shard, n_shards);
    if (!indices) return fail(IBL_EINVAL, "shard_indices: null pointer");
    ibl::shard_indices(m, shard, indices);
    return IBL_OK;
}

int ibl_shard_pack_dev(int kind, uint32_t in_w, uint32_t out_w, uint32_t mip_level, uint32_t mip_levels,
                       uint32_t shard, uint32_t n_shards, uint32_t texel_bytes, const void* d_cube, void* d_packed,
                       void* stream) {
    if (int rc = check_ready()) return rc;
    ibl::ShardMap m;
    if (int rc = shard_map(kind, in_w, out_w, mip_level, mip_levels, n_shards, &m)) return rc;
    if (shard >= n_shards) return fail(IBL_EINVAL, "shard %u of %u", shard, n_shards);
    if (texel_bytes != 4 && texel_bytes != 16) return fail(IBL_EINVAL, "shard_pack: texels of %u bytes (4 or 16)", texel_bytes);
    if (!d_cube || !d_packed) return fail(IBL_EINVAL, "shard_pack: null pointer");
    HIP_TRY(ibl::shard_pack_launch(m, shard, texel_bytes, d_cube, d_packed, pick(stream)));
    return IBL_OK;
}

int ibl_shard_unpack_dev(int kind, uint32_t in_w, uint32_t out_w, uint32_t mip_level, uint32_t mip_levels,
                         uint32_t n_shards, uint32_t texel_bytes, const void* d_packed_all, void* d_cube, void* stream) {
    if (int rc = check_ready()) return rc;
    ibl::ShardMap m;
    if (int rc = shard_map(kind, in_w, out_w, mip_level, mip_levels, n_shards, &m)) return rc;
    if (texel_bytes != 4 && texel_bytes != 16) return fail(IBL_EINVAL, "shard_unpack: texels of %u bytes (4 or 16)", texel_bytes);
    if (!d_packed_all || !d_cube) return fail(IBL_EINVAL, "shard_unpack: null pointer");
    HIP_TRY(ibl::shard_unpack_launch(m, texel_bytes, d_packed_all, d_cube, pick(stream)));
    return IBL_OK;
}

int ibl_brdf_lut_dev(uint32_t n_rough, uint32_t n_cos, uint32_t samples_sqrt, uint64_t seed, int jitter_mode,
                     uint32_t cell_begin, uint32_t cell_count, float* d_out, void* stream) {
    if (int rc = check_ready()) return rc;
    if (!d_out) return fail(IBL_EINVAL, "brdf_lut: null pointer");
    if (n_rough == 0 || n_cos == 0 || (uint64_t)n_rough * n_cos > (1u << 26))
        return fail(IBL_EINVAL, "brdf_lut: table %ux%u out of range", n_rough, n_cos);
    if (samples_sqrt == 0 || samples_sqrt > 65535) return fail(IBL_EINVAL, "brdf_lut: samples_sqrt %u out of range", samples_sqrt);
    if (jitter_mode != IBL_LUT_JITTER_COUNTER && jitter_mode != IBL_LUT_JITTER_MIDPOINT)
        return fail(IBL_EINVAL, "brdf_lut: unknown jitter mode %d", jitter_mode);
    if (int rc = check_range(cell_begin, cell_count, (uint64_t)n_rough * n_cos, "brdf_lut")) return rc;
    S->last_counts[0] = 0;
    S->last_counts[3] = S->last_counts[4] = S->last_counts[5] = 0;
    S->last_counts[1] = (uint64_t)cell_count * samples_sqrt * samples_sqrt;
    S->last_counts[2] = 0;
    S->counts_pending = 0;
    hipStream_t s = pick(stream);
    HIP_TRY(ibl::brdf_lut_launch(n_rough, n_cos, samples_sqrt, seed, jitter_mode, cell_begin, cell_count, d_out, s,
                                 S->timing ? S->ev0 : nullptr, S->timing ? S->ev1 : nullptr));
    finish_timed_call();
    return IBL_OK;
}

int ibl_rgbe_decode_dev(const uint8_t* d_rgbe, float* d_rgba, uint32_t n, void* stream) {
    if (int rc = check_ready()) return rc;
    if (n && (!d_rgbe || !d_rgba)) return fail(IBL_EINVAL, "rgbe_decode: null pointer");
    HIP_TRY(ibl::rgbe_decode_launch(d_rgbe, d_rgba, n, pick(stream)));
    return IBL_OK;
}

int ibl_rgbe_encode_dev(const float* d_rgba, uint8_t* d_rgbe, uint32_t n, void* stream) {
    if (int rc = check_ready()) return rc;
    if (n && (!d_rgba || !d_rgbe)) return fail(IBL_EINVAL, "rgbe_encode: null pointer");
    HIP_TRY(ibl::rgbe_encode_launch(d_rgba, d_rgbe, n, pick(stream)));
    return IBL_OK;
}

uint64_t ibl_png_rows_bytes(uint32_t w, uint32_t h) { return ((uint64_t)w * 4 + 1) * h; }

int ibl_png_filter_rows_dev(const uint8_t* d_rgba8, uint32_t w, uint32_t h, uint8_t* d_rows, void* stream) {
    if (int rc = check_ready()) return rc;
    if (w == 0 || h == 0) return IBL_OK;
    if (!d_rgba8 || !d_rows) return fail(IBL_EINVAL, "png_filter_rows: null pointer");
    if ((uint64_t)w * h > 0x7fffffffull / 4) return fail(IBL_EINVAL, "png_filter_rows: image %ux%u too large", w, h);
    HIP_TRY(ibl::png_filter_rows_launch(d_rgba8, w, h, d_rows, pick(stream)));
    return IBL_OK;
}

int ibl_downscale_dev(const float* d_env, uint32_t w, uint32_t h, uint32_t d, float* d_out, void* stream) {
    if (int rc = check_ready()) return rc;
    if (!d_env || !d_out) return fail(IBL_EINVAL, "downscale: null pointer");
    if (int rc = check_cube(w, h, "downscale input")) return rc;
    // cube.cpp:433-435
    if (!is_pow2(d)) return fail(IBL_EINVAL, "scale factor for --input-downscale should be a power of 2 >= 1");
    if (d > w || w % d != 0) return fail(IBL_EINVAL, "downscale: factor %u does not divide the face width %u", d, w);
    HIP_TRY(ibl::downscale_launch(d_env, w, h, d, d_out, pick(stream)));
    return IBL_OK;
}

uint32_t ibl_env_pyramid_levels(uint32_t w) { return w ? ibl::pyramid_levels(w) : 0; }

uint64_t ibl_env_pyramid_texels(uint32_t w) { return w ? ibl::pyramid_offset(w, ibl::pyramid_levels(w)) : 0; }

int ibl_env_pyramid_dev(const float* d_env, uint32_t w, float* d_pyramid, void* stream) {
    if (int rc = check_ready()) return rc;
    if (!d_env || !d_pyramid) return fail(IBL_EINVAL, "env_pyramid: null pointer");
    if (int rc = check_cube(w, w, "env_pyramid input")) return rc;
    if (w > 8192) return fail(IBL_EINVAL, "env_pyramid: face width %u is above 8192", w);
    HIP_TRY(ibl::pyramid_launch(d_env, w, d_pyramid, pick(stream)));
    return IBL_OK;
}

int ibl_ggx_level_sampled_dev(const float* d_pyramid, uint32_t in_w, uint32_t out_w, uint32_t mip_level,
                              uint32_t mip_levels, uint32_t n_samples, float lod_bias, uint32_t out_begin,
                              uint32_t out_count, float* d_out, void* stream) {
    if (int rc = check_ready()) return rc;
    if (!d_pyramid || !d_out) return fail(IBL_EINVAL, "ggx_sampled: null pointer");
    if (int rc = check_cube(in_w, in_w, "ggx_sampled input")) return rc;
    if (int rc = check_cube(out_w, out_w, "ggx_sampled output")) return rc;
    if (in_w > 8192) return fail(IBL_EINVAL, "ggx_sampled: face width %u is above 8192", in_w);
    if (mip_levels == 0 || mip_level == 0 || mip_level > mip_levels)
        return fail(IBL_EINVAL, "ggx_sampled: level %u of %u is out of range", mip_level, mip_levels);
    if (n_samples == 0 || n_samples > (1u << 24))
        return fail(IBL_EINVAL, "ggx_sampled: %u samples per texel is out of range (1..2^24)", n_samples);
    if (!(lod_bias > -64.0f && lod_bias < 64.0f)) return fail(IBL_EINVAL, "ggx_sampled: lod bias out of range");
    if (int rc = check_range(out_begin, out_count, (uint64_t)out_w * out_w * 6, "ggx_sampled")) return rc;
    S->last_counts[0] = 0;
    S->last_counts[3] = S->last_counts[4] = S->last_counts[5] = 0;
    S->last_counts[1] = S->last_counts[2] = (uint64_t)out_count * n_samples;
    S->counts_pending = 0;
    if (out_count == 0) { S->timed_valid = false; return IBL_OK; }
    HIP_TRY(ibl::ggx_sampled_launch(d_pyramid, in_w, out_w, mip_level, mip_levels, n_samples, lod_bias, out_begin,
                                    out_count, d_out, pick(stream), S->timing ? S->ev0 : nullptr,
                                    S->timing ? S->ev1 : nullptr));
    finish_timed_call();
    return IBL_OK;
}

int ibl_texel_directions_dev(uint32_t w, uint32_t h, int normalized, float* d_out, void* stream) {
    if (int rc = check_ready()) return rc;
    if (!d_out) return fail(IBL_EINVAL, "texel_directions: null pointer");
    if (w == 0 || h == 0 || (uint64_t)w * h * 6 > 0xffffffffull)
        return fail(IBL_EINVAL, "texel_directions: bad cube size %u x %u", w, h);
    HIP_TRY(ibl::texel_directions_launch(w, h, normalized, d_out, pick(stream)));
    return IBL_OK;
}

// ---- mirror-local re-projection ---------------------------------------------------

int ibl_mirror_reflect_camera(const float* transform, const float* camera, float mirror_dist, float* out) {
    if (!transform || !camera || !out) return fail(IBL_EINVAL, "mirror_reflect_camera: null pointer");
    float p[3];
    for (int r = 0; r < 3; r++) {
        volatile float v = transform[r] * camera[0];
        v = v + transform[4 + r] * camera[1];
        v = v + transform[8 + r] * camera[2];
        v = v + transform[12 + r];
        p[r] = v;
    }
    out[0] = p[0];
    out[1] = p[1];
    out[2] = -(p[2] + mirror_dist) - mirror_dist;
    return IBL_OK;
}

int ibl_mirror_reproject_dev(const float* d_src, uint32_t w, const float* camera, const float* box_min,
                             const float* box_max, float* d_out, void* stream) {
    if (int rc = check_ready()) return rc;
    if (!d_src || !d_out || !camera || !box_min || !box_max) return fail(IBL_EINVAL, "mirror_reproject: null pointer");
    if (d_src == d_out) return fail(IBL_EINVAL, "mirror_reproject: source and destination must differ");
    if (int rc = check_cube(w, w, "mirror_reproject cube")) return rc;
    for (int k = 0; k < 3; k++) {
        if (!(box_min[k] < box_max[k])) return fail(IBL_EINVAL, "mirror_reproject: empty bounding box on axis %d", k);
        if (!(camera[k] == camera[k])) return fail(IBL_EINVAL, "mirror_reproject: camera position is not a number");
    }
    HIP_TRY(ibl::mirror_reproject_launch(d_src, w, camera, box_min, box_max, d_out, pick(stream)));
    return IBL_OK;
}

// ---- host-buffer entry points ---------------------------------------------------

namespace {
extern "C++" {

// Copies `bytes` from the host into the staging-in buffer.
int stage_in(const void* host, size_t bytes) {
    if (int rc = ensure(S->stage_in, bytes)) return rc;
    HIP_TRY(hipMemcpyAsync(S->stage_in.ptr, host, bytes, hipMemcpyHostToDevice, S->stream));
    return IBL_OK;
}

int stage_out(void* host, size_t bytes) {
    HIP_TRY(hipMemcpyAsync(host, S->stage_out.ptr, bytes, hipMemcpyDeviceToHost, S->stream));
    HIP_TRY(hipStreamSynchronize(S->stream));
    return IBL_OK;
}

// ---- one call on every initialised device ------------------------------------------------
// The reference's integrateLambertian / integrateGGX / generateGGXLUT are one call that uses every
// core of the host (cube.cpp:95,240).  With several devices initialised (ibl_init_all) the
// host-buffer entry points and the bakes do the same with the GPUs: one host thread per device, every
// device works on shard i of n of the output and the shards meet in the caller's host buffer.
// A one-GPU machine rehearses that code with the knob "bake_rehearse_shards" = n: the one device
// plays n shards, in turn on its own state or -- "bake_rehearse_concurrent" = 1 -- at once, every
// thread on a stand-in state of its own (own workspaces, own stream), which is the concurrent shape
// of the real thing.

struct Fan {
    std::vector<int> devices;     // device of every shard
    bool stand_ins = false;       // every shard works on a temporary state of its own
    bool in_turn = false;         // shards run one after the other
};

// The shards a host-buffer call is split into: empty for the plain one-device call.
Fan fan_plan() {
    Fan f;
    for (int d = 0; d < MAX_DEVICES; d++) if (g_states[d].ready) f.devices.push_back(d);
    if (f.devices.size() > 1) return f;
    const int rehearse = g_tuning.bake_rehearse_shards;
    f.devices.clear();
    if (rehearse > 1 && S) {
        f.devices.assign((size_t)rehearse, S->device);
        f.stand_ins = g_tuning.bake_rehearse_concurrent != 0;
        f.in_turn = !f.stand_ins;
    }
    return f;
}

// Runs fn(shard, n_shards) for every shard of the plan, each on its own host thread with its device
// current; returns the first failure (with the device named in the message).
template <typename Fn>
int fan_out(const Fan& fan, Fn fn) {
    const uint32_t n = (uint32_t)fan.devices.size();
    int current = -1;
    (void)hipGetDevice(&current);
    std::vector<int> rcs(n, IBL_OK);
    std::vector<std::string> messages(n);
    std::vector<State> stand_ins(fan.stand_ins ? n : 0);
    std::vector<std::thread> threads;
    for (uint32_t i = 0; i < n; i++) {
        threads.emplace_back([&, i]() {
            int rc = IBL_OK;
            if (hipSetDevice(fan.devices[i]) != hipSuccess) rc = fail(IBL_EHIP, "hipSetDevice(%d) failed", fan.devices[i]);
            if (!rc && fan.stand_ins) {
                State& st = stand_ins[i];
                st.device = fan.devices[i];
                st.static_scratch = g_states[st.device].static_scratch;
                if (hipStreamCreateWithFlags(&st.stream, hipStreamNonBlocking) != hipSuccess)
                    rc = fail(IBL_EHIP, "hipStreamCreate failed");
                if (!rc) rc = create_events(st);
                if (!rc) { st.ready = true; S_stand_in = &st; }
            }
            if (!rc) rc = check_ready();
            if (!rc) rc = fn(i, n);
            if (S) (void)hipStreamSynchronize(S->stream);
            rcs[i] = rc;
            if (rc) messages[i] = g_err;
            if (fan.stand_ins) {
                S_stand_in = nullptr;
                drop_state(stand_ins[i]);
            }
        });
        if (fan.in_turn) threads.back().join();
    }
    for (std::thread& t : threads) if (t.joinable()) t.join();
    if (current >= 0) (void)hipSetDevice(current);
    S_stand_in = nullptr;
    (void)check_ready();
    for (uint32_t i = 0; i < n; i++)
        if (rcs[i]) return fail(rcs[i], "device %d: %s", fan.devices[i], messages[i].c_str());
    return IBL_OK;
}

// This shard's float4 texels of a cube on the device (full-size d_cube) into their places in the
// caller's host cube.  Shards are disjoint, so the threads write different texels.
int deliver_shard(const ibl::ShardMap& map, uint32_t shard, const float* d_cube, float* host_cube) {
    const uint32_t per = ibl::shard_slots(map);
    if (int rc = ensure(S->png, (size_t)per * 16)) return rc;
    HIP_TRY(ibl::shard_pack_launch(map, shard, 16, d_cube, S->png.ptr, S->stream));
    std::vector<float> packed((size_t)per * 4);
    std::vector<uint32_t> slots(per);
    HIP_TRY(hipMemcpyAsync(packed.data(), S->png.ptr, (size_t)per * 16, hipMemcpyDeviceToHost, S->stream));
    ibl::shard_indices(map, shard, slots.data());          // host arithmetic, while the device works
    HIP_TRY(hipStreamSynchronize(S->stream));
    for (uint32_t k = 0; k < per; k++)
        if (slots[k] != 0xffffffffu) memcpy(host_cube + (size_t)slots[k] * 4, packed.data() + (size_t)k * 4, 16);
    return IBL_OK;
}

ibl::ShardMap range_map(uint32_t out_w, uint32_t n_shards) {
    ibl::ShardMap map;
    map.partition = ibl::SHARD_RANGE; map.out_w = out_w; map.tw = 1; map.n_shards = n_shards;
    return map;
}

}  // extern "C++"
}  // namespace

int ibl_lambertian(const float* env, uint32_t in_w, uint32_t in_h, uint32_t out_size, float* out) {
    if (int rc = check_ready()) return rc;
    if (!env || !out) return fail(IBL_EINVAL, "lambertian: null pointer");
    if (int rc = check_cube(in_w, in_h, "lambertian input")) return rc;
    if (int rc = check_cube(out_size, out_size, "lambertian output")) return rc;
    const size_t in_bytes = (size_t)in_w * in_h * 6 * 16, out_bytes = (size_t)out_size * out_size * 6 * 16;
    const Fan fan = fan_plan();
    if (!fan.devices.empty())
        return fan_out(fan, [&](uint32_t shard, uint32_t n) -> int {
            ibl::ShardMap map;
            if (int rc = shard_map(IBL_SHARD_LAMBERTIAN, in_w, out_size, 1, 1, n, &map)) return rc;
            if (int rc = stage_in(env, in_bytes)) return rc;
            if (int rc = ensure(S->stage_out, out_bytes)) return rc;
            if (int rc = ibl_lambertian_shard_dev((const float*)S->stage_in.ptr, in_w, in_h, out_size, shard, n,
                                                  (float*)S->stage_out.ptr, S->stream)) return rc;
            return deliver_shard(map, shard, (const float*)S->stage_out.ptr, out);
        });
    if (int rc = stage_in(env, in_bytes)) return rc;
    if (int rc = ensure(S->stage_out, out_bytes)) return rc;
    if (int rc = ibl_lambertian_dev((const float*)S->stage_in.ptr, in_w, in_h, out_size, 0, out_size * out_size * 6,
                                    (float*)S->stage_out.ptr, S->stream)) return rc;
    return stage_out(out, out_bytes);
}

int ibl_ggx_level(const float* env, uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h,
                  uint32_t mip_level, uint32_t mip_levels, float* out) {
    if (int rc = check_ready()) return rc;
    if (!env || !out) return fail(IBL_EINVAL, "ggx: null pointer");
    if (int rc = check_cube(in_w, in_h, "ggx input")) return rc;
    if (int rc = check_cube(out_w, out_h, "ggx output")) return rc;
    const size_t in_bytes = (size_t)in_w * in_h * 6 * 16, out_bytes = (size_t)out_w * out_h * 6 * 16;
    const Fan fan = fan_plan();
    if (!fan.devices.empty()) {
        uint32_t patch = 0;
        if (int rc = ggx_check(env, in_w, in_h, out_w, out_h, mip_level, mip_levels, out, &patch)) return rc;
        return fan_out(fan, [&](uint32_t shard, uint32_t n) -> int {
            ibl::ShardMap map;
            if (int rc = shard_map(IBL_SHARD_GGX, in_w, out_w, mip_level, mip_levels, n, &map)) return rc;
            if (int rc = stage_in(env, in_bytes)) return rc;
            if (int rc = ensure(S->stage_out, out_bytes)) return rc;
            if (int rc = ibl_ggx_level_shard_dev((const float*)S->stage_in.ptr, in_w, in_h, out_w, out_h, mip_level,
                                                 mip_levels, shard, n, (float*)S->stage_out.ptr, S->stream)) return rc;
            return deliver_shard(map, shard, (const float*)S->stage_out.ptr, out);
        });
    }
    if (int rc = stage_in(env, in_bytes)) return rc;
    if (int rc = ensure(S->stage_out, out_bytes)) return rc;
    if (int rc = ibl_ggx_level_dev((const float*)S->stage_in.ptr, in_w, in_h, out_w, out_h, mip_level, mip_levels, 0,
                                   out_w * out_h * 6, (float*)S->stage_out.ptr, S->stream)) return rc;
    return stage_out(out, out_bytes);
}

int ibl_ggx_level_sampled(const float* env, uint32_t in_w, uint32_t out_w, uint32_t mip_level, uint32_t mip_levels,
                          uint32_t n_samples, float lod_bias, float* out) {
    if (int rc = check_ready()) return rc;
    if (!env || !out) return fail(IBL_EINVAL, "ggx_sampled: null pointer");
    if (int rc = check_cube(in_w, in_w, "ggx_sampled input")) return rc;
    if (int rc = check_cube(out_w, out_w, "ggx_sampled output")) return rc;
    const size_t in_bytes = (size_t)in_w * in_w * 6 * 16, out_bytes = (size_t)out_w * out_w * 6 * 16;
    const uint32_t n_out = out_w * out_w * 6;
    const Fan fan = fan_plan();
    // (sampled mode: shards are plain texel ranges, copied straight into their place)
    auto part = [&](uint32_t shard, uint32_t n) -> int {
        const uint32_t per = (n_out + n - 1) / n;
        const uint32_t begin = shard * per < n_out ? shard * per : n_out;
        const uint32_t count = per < n_out - begin ? per : n_out - begin;
        if (int rc = stage_in(env, in_bytes)) return rc;
        if (int rc = ensure(S->a, (size_t)ibl_env_pyramid_texels(in_w) * 16)) return rc;
        if (int rc = ensure(S->stage_out, out_bytes)) return rc;
        if (int rc = ibl_env_pyramid_dev((const float*)S->stage_in.ptr, in_w, (float*)S->a.ptr, S->stream)) return rc;
        if (int rc = ibl_ggx_level_sampled_dev((const float*)S->a.ptr, in_w, out_w, mip_level, mip_levels, n_samples,
                                               lod_bias, begin, count, (float*)S->stage_out.ptr, S->stream)) return rc;
        HIP_TRY(hipMemcpyAsync(out + (size_t)begin * 4, (const float*)S->stage_out.ptr + (size_t)begin * 4, (size_t)count * 16,
                               hipMemcpyDeviceToHost, S->stream));
        HIP_TRY(hipStreamSynchronize(S->stream));
        return IBL_OK;
    };
    if (!fan.devices.empty()) return fan_out(fan, part);
    return part(0, 1);
}

int ibl_brdf_lut(uint32_t n_rough, uint32_t n_cos, uint32_t samples_sqrt, uint64_t seed, int jitter_mode,
                 float* out_slope_bias) {
    if (int rc = check_ready()) return rc;
    if (!out_slope_bias) return fail(IBL_EINVAL, "brdf_lut: null pointer");
    if (n_rough == 0 || n_cos == 0 || (uint64_t)n_rough * n_cos > (1u << 26))
        return fail(IBL_EINVAL, "brdf_lut: table %ux%u out of range", n_rough, n_cos);
    const uint32_t n_cells = n_rough * n_cos;
    const size_t out_bytes = (size_t)n_cells * 2 * sizeof(float);
    // every cell is independent (cube.cpp:346-388): shards are ranges of cells
    auto part = [&](uint32_t shard, uint32_t n) -> int {
        const uint32_t per = (n_cells + n - 1) / n;
        const uint32_t begin = shard * per < n_cells ? shard * per : n_cells;
        const uint32_t count = per < n_cells - begin ? per : n_cells - begin;
        if (int rc = ensure(S->stage_out, out_bytes)) return rc;
        if (int rc = ibl_brdf_lut_dev(n_rough, n_cos, samples_sqrt, seed, jitter_mode, begin, count,
                                      (float*)S->stage_out.ptr, S->stream)) return rc;
        HIP_TRY(hipMemcpyAsync(out_slope_bias + (size_t)begin * 2, (const float*)S->stage_out.ptr + (size_t)begin * 2,
                               (size_t)count * 2 * sizeof(float), hipMemcpyDeviceToHost, S->stream));
        HIP_TRY(hipStreamSynchronize(S->stream));
        return IBL_OK;
    };
    const Fan fan = fan_plan();
    if (!fan.devices.empty()) return fan_out(fan, part);
    return part(0, 1);
}

int ibl_rgbe_decode(const uint8_t* rgbe, float* rgba, uint32_t n) {
    if (int rc = check_ready()) return rc;
    if (n == 0) return IBL_OK;
    if (!rgbe || !rgba) return fail(IBL_EINVAL, "rgbe_decode: null pointer");
    if (int rc = stage_in(rgbe, (size_t)n * 4)) return rc;
    if (int rc = ensure(S->stage_out, (size_t)n * 16)) return rc;
    if (int rc = ibl_rgbe_decode_dev((const uint8_t*)S->stage_in.ptr, (float*)S->stage_out.ptr, n, S->stream)) return rc;
    return stage_out(rgba, (size_t)n * 16);
}

int ibl_rgbe_encode(const float* rgba, uint8_t* rgbe, uint32_t n) {
    if (int rc = check_ready()) return rc;
    if (n == 0) return IBL_OK;
    if (!rgba || !rgbe) return fail(IBL_EINVAL, "rgbe_encode: null pointer");
    if (int rc = stage_in(rgba, (size_t)n * 16)) return rc;
    if (int rc = ensure(S->stage_out, (size_t)n * 4)) return rc;
    if (int rc = ibl_rgbe_encode_dev((const float*)S->stage_in.ptr, (uint8_t*)S->stage_out.ptr, n, S->stream)) return rc;
    return stage_out(rgbe, (size_t)n * 4);
}

int ibl_downscale(const float* env, uint32_t w, uint32_t h, uint32_t d, float* out) {
    if (int rc = check_ready()) return rc;
    if (!env || !out) return fail(IBL_EINVAL, "downscale: null pointer");
    if (int rc = check_cube(w, h, "downscale input")) return rc;
    if (!is_pow2(d) || d > w || w % d != 0) return fail(IBL_EINVAL, "scale factor for --input-downscale should be a power of 2 >= 1");
    const size_t in_bytes = (size_t)w * h * 6 * 16, out_bytes = (size_t)(w / d) * (h / d) * 6 * 16;
    if (int rc = stage_in(env, in_bytes)) return rc;
    if (int rc = ensure(S->stage_out, out_bytes)) return rc;
    if (int rc = ibl_downscale_dev((const float*)S->stage_in.ptr, w, h, d, (float*)S->stage_out.ptr, S->stream)) return rc;
    return stage_out(out, out_bytes);
}

int ibl_png_filter_rows(const uint8_t* rgba8, uint32_t w, uint32_t h, uint8_t* rows) {
    if (int rc = check_ready()) return rc;
    if (w == 0 || h == 0) return IBL_OK;
    if (!rgba8 || !rows) return fail(IBL_EINVAL, "png_filter_rows: null pointer");
    if ((uint64_t)w * h > 0x7fffffffull / 4) return fail(IBL_EINVAL, "png_filter_rows: image %ux%u too large", w, h);
    const size_t out_bytes = (size_t)ibl_png_rows_bytes(w, h);
    if (int rc = stage_in(rgba8, (size_t)w * h * 4)) return rc;
    if (int rc = ensure(S->stage_out, out_bytes)) return rc;
    if (int rc = ibl_png_filter_rows_dev((const uint8_t*)S->stage_in.ptr, w, h, (uint8_t*)S->stage_out.ptr, S->stream)) return rc;
    return stage_out(rows, out_bytes);
}

namespace {
extern "C++" int bake(const uint8_t* rgbe_in, uint32_t w, uint32_t input_downscale, uint32_t cubemap_size, uint32_t ggx_samples,
                      float lod_bias, bool png_rows, uint8_t* lambertian_out, uint8_t* ggx_out);
}

int ibl_bake_rgbe(const uint8_t* rgbe_in, uint32_t w, uint32_t input_downscale, uint32_t cubemap_size,
                  uint8_t* lambertian_out, uint8_t* ggx_out) {
    return bake(rgbe_in, w, input_downscale, cubemap_size, 0, 0.0f, false, lambertian_out, ggx_out);
}

int ibl_bake_rgbe_sampled(const uint8_t* rgbe_in, uint32_t w, uint32_t input_downscale, uint32_t cubemap_size,
                          uint32_t ggx_samples, float lod_bias, uint8_t* lambertian_out, uint8_t* ggx_out) {
    return bake(rgbe_in, w, input_downscale, cubemap_size, ggx_samples, lod_bias, false, lambertian_out, ggx_out);
}

int ibl_bake_png_rows(const uint8_t* rgbe_in, uint32_t w, uint32_t input_downscale, uint32_t cubemap_size,
                      uint32_t ggx_samples, float lod_bias, uint8_t* lambertian_rows, uint8_t* ggx_rows) {
    return bake(rgbe_in, w, input_downscale, cubemap_size, ggx_samples, lod_bias, true, lambertian_rows, ggx_rows);
}

namespace {
extern "C++" {

ibl_progress_fn g_progress = nullptr;
void* g_progress_user = nullptr;
std::mutex g_progress_mutex;

void report_done(int cube) {
    std::lock_guard<std::mutex> lock(g_progress_mutex);
    if (g_progress) g_progress(cube, g_progress_user);
}

struct BakeJob {
    const uint8_t* rgbe_in;
    uint32_t w, input_downscale, cubemap_size, ggx_samples;
    float lod_bias;
    bool png_rows;
    uint8_t* lambertian_out;
    uint8_t* ggx_out;
    uint32_t mips;      // GGX levels to bake (0: none)
};

// The environment of a bake on the current device: RGBE strip in, decoded and (if asked) box-filtered.
// Device layout of the staging-in block: [rgbe bytes][full float cube][downscaled float cube].
int bake_environment(const BakeJob& job, float** d_env, uint32_t* env_w) {
    const uint32_t w = job.w, n_full = w * w * 6;
    const uint32_t dw = w / job.input_downscale, n_ds = dw * dw * 6;
    const size_t off_full = ((size_t)n_full * 4 + 255) & ~(size_t)255;
    const size_t off_ds = off_full + (size_t)n_full * 16;
    const size_t in_total = off_ds + (job.input_downscale > 1 ? (size_t)n_ds * 16 : 0);
    if (int rc = ensure(S->stage_in, in_total)) return rc;
    char* base = (char*)S->stage_in.ptr;
    HIP_TRY(hipMemcpyAsync(base, job.rgbe_in, (size_t)n_full * 4, hipMemcpyHostToDevice, S->stream));
    float* d_full = (float*)(base + off_full);
    *d_env = d_full;
    *env_w = dw;
    if (int rc = ibl_rgbe_decode_dev((const uint8_t*)base, d_full, n_full, S->stream)) return rc;
    if (job.input_downscale > 1) {
        *d_env = (float*)(base + off_ds);
        if (int rc = ibl_downscale_dev(d_full, w, w, job.input_downscale, *d_env, S->stream)) return rc;
    }
    return IBL_OK;
}

// The cubes of a bake in the order they are baked and reported: the Lambertian cube (report 0), then
// the GGX levels 1 .. mips; where each one's bytes go in the caller's buffers.
struct BakeCube {
    bool lambertian = false;
    uint32_t level = 0;           // GGX level (0 for the Lambertian cube): also the number reported
    uint32_t edge = 0;
    uint8_t* dst = nullptr;       // RGBE texels or PNG scanlines in the caller's buffer
    size_t bytes = 0;
};

std::vector<BakeCube> bake_cubes(const BakeJob& job) {
    std::vector<BakeCube> cubes;
    auto size_of = [&](uint32_t edge) {
        return job.png_rows ? (size_t)ibl_png_rows_bytes(edge, edge * 6) : (size_t)edge * edge * 6 * 4;
    };
    if (job.lambertian_out) {
        BakeCube c;
        c.lambertian = true; c.edge = job.cubemap_size; c.dst = job.lambertian_out; c.bytes = size_of(c.edge);
        cubes.push_back(c);
    }
    size_t off = 0;
    uint32_t mw = job.w / 2;
    for (uint32_t m = 1; m <= job.mips; m++, mw /= 2) {
        BakeCube c;
        c.level = m; c.edge = mw; c.dst = job.ggx_out + off; c.bytes = size_of(mw);
        off += c.bytes;
        cubes.push_back(c);
    }
    return cubes;
}

// The whole bake on the current device.  The cubes go through two sets of output buffers in turn: a
// cube's kernels, its encoding and its copy to the host are queued, and only then is the cube before
// it waited for and reported -- so whatever the progress callback does with a finished cube (bin/cube
// deflates it and writes its file) happens while the device works on the next one.
int bake_single(const BakeJob& job) {
    float* d_env = nullptr;
    uint32_t dw = 0;
    if (int rc = bake_environment(job, &d_env, &dw)) return rc;
    const std::vector<BakeCube> cubes = bake_cubes(job);
    uint32_t n_max = 0;
    for (const BakeCube& c : cubes) if (c.edge * c.edge * 6 > n_max) n_max = c.edge * c.edge * 6;
    if (n_max == 0) return IBL_OK;
    // Staging-out block, twice: [float cube of the largest output][its RGBE bytes][its PNG scanlines]
    const size_t off_bytes = (size_t)n_max * 16;
    const size_t off_rows = off_bytes + (size_t)n_max * 4;
    const size_t set_bytes = (off_rows + (job.png_rows ? (size_t)n_max * 4 + (size_t)n_max : 0) + 255) & ~(size_t)255;
    if (int rc = ensure(S->stage_out, 2 * set_bytes)) return rc;
    if (job.mips && job.ggx_samples) {      // sampled mode: one pyramid of the (downscaled) environment for all levels
        if (int rc = ensure(S->b, (size_t)ibl_env_pyramid_texels(dw) * 16)) return rc;
        if (int rc = ibl_env_pyramid_dev(d_env, dw, (float*)S->b.ptr, S->stream)) return rc;
    }
    hipEvent_t copied[2] = {nullptr, nullptr};
    HIP_TRY(hipEventCreateWithFlags(&copied[0], hipEventDisableTiming));
    if (hipEventCreateWithFlags(&copied[1], hipEventDisableTiming) != hipSuccess) {
        (void)hipEventDestroy(copied[0]);
        return fail(IBL_EHIP, "hipEventCreate failed");
    }
    auto queue = [&](size_t ci) -> int {
        const BakeCube& c = cubes[ci];
        uint8_t* base = (uint8_t*)S->stage_out.ptr + (ci & 1) * set_bytes;
        float* d_f = (float*)base;
        uint8_t* d_e = base + off_bytes;
        uint8_t* d_rows = base + off_rows;
        const uint32_t n = c.edge * c.edge * 6;
        if (c.lambertian) {
            if (int rc = ibl_lambertian_dev(d_env, dw, dw, c.edge, 0, n, d_f, S->stream)) return rc;
        } else if (job.ggx_samples) {
            if (int rc = ibl_ggx_level_sampled_dev((const float*)S->b.ptr, dw, c.edge, c.level, job.mips, job.ggx_samples,
                                                   job.lod_bias, 0, n, d_f, S->stream)) return rc;
        } else if (int rc = ibl_ggx_level_dev(d_env, dw, dw, c.edge, c.edge, c.level, job.mips, 0, n, d_f, S->stream)) return rc;
        if (int rc = ibl_rgbe_encode_dev(d_f, d_e, n, S->stream)) return rc;
        const uint8_t* src = d_e;
        if (job.png_rows) {
            if (int rc = ibl_png_filter_rows_dev(d_e, c.edge, c.edge * 6, d_rows, S->stream)) return rc;
            src = d_rows;
        }
        HIP_TRY(hipMemcpyAsync(c.dst, src, c.bytes, hipMemcpyDeviceToHost, S->stream));
        HIP_TRY(hipEventRecord(copied[ci & 1], S->stream));
        return IBL_OK;
    };
    int rc = IBL_OK;
    for (size_t ci = 0; ci <= cubes.size() && !rc; ci++) {
        if (ci < cubes.size()) rc = queue(ci);
        if (!rc && ci > 0) {
            if (hipEventSynchronize(copied[(ci - 1) & 1]) != hipSuccess) rc = fail(IBL_EHIP, "hipEventSynchronize failed");
            else report_done((int)cubes[ci - 1].level);
        }
    }
    (void)hipStreamSynchronize(S->stream);
    (void)hipEventDestroy(copied[0]);
    (void)hipEventDestroy(copied[1]);
    return rc;
}

// ---- the bake on several devices of one process ----------------------------------------
// One host thread per device (fan_out).  Every device decodes the strip for itself, computes its shard
// of every cube (the partition of the *_shard_dev entry points), encodes it, packs its texels
// (ibl_shard_pack_dev) and copies them to the host, where the thread scatters them into the cube's
// RGBE bytes: shards are disjoint, so the threads write different words.  A texel does not depend on
// the number of devices, so the bytes are those of the one-device bake.  The thread that delivers the
// last shard of a cube finishes it -- in PNG mode it filters the assembled texels on its own device --
// and reports it, so finished cubes reach the caller while the other cubes are still being baked.

struct SharedCube {
    BakeCube cube;
    uint8_t* texels = nullptr;             // the cube's RGBE texels on the host (caller's buffer, or temporary in PNG mode)
    std::atomic<int> waiting{0};           // shards that have not been delivered yet
};

int bake_shard(const BakeJob& job, uint32_t shard, uint32_t n_shards, std::vector<SharedCube>& cubes) {
    float* d_env = nullptr;
    uint32_t dw = 0;
    if (int rc = bake_environment(job, &d_env, &dw)) return rc;
    uint32_t n_max = 0;
    for (const SharedCube& c : cubes) if (c.cube.edge * c.cube.edge * 6 > n_max) n_max = c.cube.edge * c.cube.edge * 6;
    if (n_max == 0) return IBL_OK;
    // Staging-out block: [float cube][its RGBE bytes][this shard's packed RGBE texels]
    const size_t off_bytes = (size_t)n_max * 16;
    const size_t off_packed = off_bytes + (size_t)n_max * 4;
    if (int rc = ensure(S->stage_out, off_packed + (size_t)n_max * 4 + 4096)) return rc;
    float* d_f = (float*)S->stage_out.ptr;
    uint8_t* d_e = (uint8_t*)S->stage_out.ptr + off_bytes;
    uint8_t* d_packed = (uint8_t*)S->stage_out.ptr + off_packed;
    if (job.mips && job.ggx_samples) {
        if (int rc = ensure(S->b, (size_t)ibl_env_pyramid_texels(dw) * 16)) return rc;
        if (int rc = ibl_env_pyramid_dev(d_env, dw, (float*)S->b.ptr, S->stream)) return rc;
    }
    std::vector<uint32_t> packed, slots;
    for (SharedCube& shared : cubes) {
        const BakeCube& c = shared.cube;
        const uint32_t edge = c.edge, n_out = edge * edge * 6, m = c.level;
        ibl::ShardMap map;
        if (c.lambertian) {
            if (int rc = shard_map(IBL_SHARD_LAMBERTIAN, dw, edge, 1, 1, n_shards, &map)) return rc;
            if (int rc = ibl_lambertian_shard_dev(d_env, dw, dw, edge, shard, n_shards, d_f, S->stream)) return rc;
        } else if (job.ggx_samples) {       // sampled mode: plain texel ranges
            map = range_map(edge, n_shards);
            const uint32_t per = (n_out + n_shards - 1) / n_shards;
            const uint32_t begin = shard * per < n_out ? shard * per : n_out;
            const uint32_t count = per < n_out - begin ? per : n_out - begin;
            if (int rc = ibl_ggx_level_sampled_dev((const float*)S->b.ptr, dw, edge, m, job.mips, job.ggx_samples,
                                                   job.lod_bias, begin, count, d_f, S->stream)) return rc;
        } else {
            if (int rc = shard_map(IBL_SHARD_GGX, dw, edge, m, job.mips, n_shards, &map)) return rc;
            if (int rc = ibl_ggx_level_shard_dev(d_env, dw, dw, edge, edge, m, job.mips, shard, n_shards, d_f, S->stream)) return rc;
        }
        if (int rc = ibl_rgbe_encode_dev(d_f, d_e, n_out, S->stream)) return rc;
        const uint32_t per = ibl::shard_slots(map);
        HIP_TRY(ibl::shard_pack_launch(map, shard, 4, d_e, d_packed, S->stream));
        packed.resize(per);
        slots.resize(per);
        HIP_TRY(hipMemcpyAsync(packed.data(), d_packed, (size_t)per * 4, hipMemcpyDeviceToHost, S->stream));
        ibl::shard_indices(map, shard, slots.data());          // host arithmetic, while the device works
        HIP_TRY(hipStreamSynchronize(S->stream));
        uint32_t* dst = (uint32_t*)shared.texels;
        for (uint32_t k = 0; k < per; k++)
            if (slots[k] != 0xffffffffu) dst[slots[k]] = packed[k];
        if (shared.waiting.fetch_sub(1) != 1) continue;
        // the last shard of this cube: every texel is in place (the counter orders the other threads' stores)
        if (job.png_rows) {
            const size_t in_bytes = (size_t)n_out * 4;
            const size_t in_pad = (in_bytes + 255) & ~(size_t)255;
            if (int rc = ensure(S->png, in_pad + c.bytes)) return rc;
            uint8_t* d_in = (uint8_t*)S->png.ptr;
            HIP_TRY(hipMemcpyAsync(d_in, shared.texels, in_bytes, hipMemcpyHostToDevice, S->stream));
            if (int rc = ibl_png_filter_rows_dev(d_in, edge, edge * 6, d_in + in_pad, S->stream)) return rc;
            HIP_TRY(hipMemcpyAsync(c.dst, d_in + in_pad, c.bytes, hipMemcpyDeviceToHost, S->stream));
            HIP_TRY(hipStreamSynchronize(S->stream));
        }
        report_done((int)m);
    }
    return IBL_OK;
}

int bake_multi(const BakeJob& job, const Fan& fan) {
    const uint32_t n = (uint32_t)fan.devices.size();
    const std::vector<BakeCube> list = bake_cubes(job);
    std::vector<SharedCube> cubes(list.size());
    std::vector<std::vector<uint8_t>> temp(cubes.size());          // PNG mode: the RGBE texels wait here for the filter
    for (size_t i = 0; i < cubes.size(); i++) {
        cubes[i].cube = list[i];
        cubes[i].waiting = (int)n;
        cubes[i].texels = list[i].dst;
        if (job.png_rows) {
            temp[i].assign((size_t)list[i].edge * list[i].edge * 6 * 4, 0);
            cubes[i].texels = temp[i].data();
        }
    }
    return fan_out(fan, [&](uint32_t shard, uint32_t n_shards) { return bake_shard(job, shard, n_shards, cubes); });
}

int bake(const uint8_t* rgbe_in, uint32_t w, uint32_t input_downscale, uint32_t cubemap_size, uint32_t ggx_samples,
         float lod_bias, bool png_rows, uint8_t* lambertian_out, uint8_t* ggx_out) {
    if (int rc = check_ready()) return rc;
    if (!rgbe_in) return fail(IBL_EINVAL, "bake: null input");
    if (int rc = check_cube(w, w, "bake input")) return rc;
    if (!is_pow2(input_downscale))
        return fail(IBL_EINVAL, "scale factor for --input-downscale should be a power of 2 >= 1");
    if (input_downscale > w || w % input_downscale != 0)
        return fail(IBL_EINVAL, "bake: downscale %u does not divide the face width %u", input_downscale, w);
    if (cubemap_size == 0) return fail(IBL_EINVAL, "bake: cubemap size 0");
    if (lambertian_out) if (int rc = check_cube(cubemap_size, cubemap_size, "bake lambertian output")) return rc;
    BakeJob job;
    job.rgbe_in = rgbe_in; job.w = w; job.input_downscale = input_downscale; job.cubemap_size = cubemap_size;
    job.ggx_samples = ggx_samples; job.lod_bias = lod_bias; job.png_rows = png_rows;
    job.lambertian_out = lambertian_out; job.ggx_out = ggx_out;
    job.mips = ggx_out ? ibl_ggx_mip_levels(w, w, cubemap_size) : 0;
    if (job.mips && w / input_downscale < 16)
        return fail(IBL_EINVAL, "ggx: input faces must be at least 16x16 (patch size would be 0), got %u", w / input_downscale);
    const Fan fan = fan_plan();
    int rc = fan.devices.empty() ? bake_single(job) : bake_multi(job, fan);
    // the copy from the caller's strip may still be queued when a later step has failed
    if (rc && S) (void)hipStreamSynchronize(S->stream);
    return rc;
}
}  // extern "C++"
}  // namespace

int ibl_set_progress(ibl_progress_fn fn, void* user) {
    std::lock_guard<std::mutex> lock(g_progress_mutex);
    g_progress = fn;
    g_progress_user = user;
    return IBL_OK;
}

}  // extern "C"
