// C ABI of the path tracer core (declarations and reference citations: include/rtiow_hip.h).
// There is no CPU fallback: every device entry point fails with RT_ERR_NO_DEVICE / RT_ERR_HIP
// when the HIP runtime cannot run the kernels.
#include <hip/hip_runtime.h>

#include <limits>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtiow_hip.h"
#include "host_scene.h"
#include "image_decode.h"
#include "rt_device.h"

namespace rt {
hipError_t launchRender(const RenderArgs& a, bool lds, bool count, int blocks, int threads, hipStream_t stream);
hipError_t launchWavefront(const RenderArgs& a, bool lds, bool count, int blocks, int threads, hipStream_t stream);
hipError_t launchTwoPass(const RenderArgs& a, bool count, int blocks, int threads, hipStream_t stream);
hipError_t launchTrace(const DevScene& sc, int n, const float* rays, int* kind, int* id, float* t, float* normal,
                       int* front, float* uv, hipStream_t stream);
hipError_t launchScatter(const DevScene& sc, int n, const int* mat, const float* rayIn, const float* p, const float* nrm,
                         const int* front, const float* uv, const unsigned int* key4, float* atten, float* rayOut,
                         int* alive, unsigned int* counterOut, hipStream_t stream);
hipError_t buildSearchStructureOnDevice(int n, const uint32_t* d_refs, const float4* d_spheres, const float4* d_tris,
                                        float pad_floor, float far_typical, const float cmin[3], const float cmax[3], const float eye[3], hipStream_t stream,
                                        const std::function<bool(int)>& want_copies, float4** entries, uint32_t** out_refs,
                                        int* n_entries, int* stride, float* build_ms);
hipError_t launchReflectance(int n, const float* cosine, const float* refIdx, float* out, hipStream_t stream);
hipError_t launchResolve(const float4* payloads, size_t stride4, int width, int height, int tiles_per_row_local,
                         int world, int only_rank, int mode, int spp, float4* out32, uint32_t* out8, hipStream_t stream);
}  // namespace rt

struct rt_scene {
    rt::Scene sc;
};

struct rt_context {
    int device = 0;
    hipStream_t stream = nullptr;      // the stream in use
    hipStream_t own_stream = nullptr;  // created by rt_create()
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    rt::DevScene dsc{};
    std::vector<void*> scene_allocs;
    bool scene_loaded = false;
    size_t lds_need = 0;
    bool use_lds = false;
    bool lds_refs = true;  // the reference-order arrays are staged too
    int n_cus = 256;
    int width = 0, height = 0, rank = 0, world = 1;
    int tiles_x = 0, tiles_y = 0, tprl = 0, n_local_tiles = 0;
    float4* accum = nullptr;
    bool accum_owned = false;
    unsigned int* queue = nullptr;
    unsigned long long* counters = nullptr;
    std::atomic<int> cancel{0};
    int* cancel_host = nullptr;   // pinned, device-visible copy of the flag the running kernels poll
    int* cancel_dev = nullptr;
    bool cancellable = false;     // set by rt_run() around its render calls
    int force_ref = 0;
    bool huge_coords = false;
    unsigned long long sched_stats[14] = {0};  // from the last rt_count(): {executions, lanes} per operation
    rt::Radiance* colors = nullptr;  // per-sample radiance of the split mode (grown on demand)
    size_t colors_cap = 0;           // in samples
    float4* records = nullptr;       // path records of the two-pass launches, three planes of records_cap (grown on demand)
    size_t records_cap = 0;          // in samples
    int device_build = -1;      // rt_upload_scene() builds the search structure on the device: 1 always, 0 never, -1 for large scenes
    float build_ms = 0.0f;      // device time of the last such build
    void* resolved = nullptr;  // row-major image the read-out calls resolve into (grown on demand, kept)
    size_t resolved_cap = 0;   // in bytes
};

// Scratch of rt_render_noalloc(): the queue counter (first 256 bytes) and the per-sample colour
// buffer of one launch.  One copy per GPU the library runs on, held until the process ends.  The
// name is a host-side stand-in, never a device address: host code uses the per-device pointer
// that rt_create() obtains with hipGetSymbolAddress().
constexpr size_t kScratchQueueBytes = rt::kQueueBytes;
constexpr size_t kScratchColorBytes = size_t(RT_NOALLOC_SCRATCH_MB) << 20;
static __device__ __align__(256) unsigned char scratch[kScratchQueueBytes + kScratchColorBytes];

namespace {

constexpr int kMaxDevices = 64;
std::mutex g_scratchMutex;
unsigned char* g_scratchPtr[kMaxDevices] = {nullptr};  // per device: address of its copy of `scratch`

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
// A failed runtime call also leaves its code behind as the thread's "last error", which a later,
// unrelated hipGetLastError() (the check after a kernel launch) would report: it is taken off here.
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (void)hipGetLastError();                                                          \
            return fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
        }                                                                                     \
    } while (0)

// Largest LDS footprint the staged-scene kernel is allowed to ask for.  Two such workgroups
// still fit the 160 KiB of a CU.
constexpr size_t kMaxLdsScene = 72 * 1024;
constexpr size_t kLdsThreeGroups = 52 * 1024;

int envInt(const char* name, int dflt) {
    const char* v = std::getenv(name);
    return v ? std::atoi(v) : dflt;
}

// `count` elements of T from host memory into a device allocation the context owns
template <class T, class S>
int devUploadRaw(rt_context* ctx, const S* host, size_t count, const T** out) {
    size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, bytes));
    ctx->scene_allocs.push_back(d);
    if (count) HIP_TRY(hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *out = reinterpret_cast<const T*>(d);
    return RT_OK;
}

template <class T>
int devUpload(rt_context* ctx, const std::vector<T>& host, const T** out) {
    return devUploadRaw<T, T>(ctx, host.data(), host.size(), out);
}

// RT_HOST_TIMING=1: the phases of rt_upload_scene() on stderr (host preparation, copies, device build)
struct PhaseClock {
    bool on = envInt("RT_HOST_TIMING", 0) != 0;
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[rt_upload_scene] %-34s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
};

void freeScene(rt_context* ctx) {
    for (void* p : ctx->scene_allocs) (void)hipFree(p);
    ctx->scene_allocs.clear();
    ctx->scene_loaded = false;
}

void freeTarget(rt_context* ctx) {
    if (ctx->accum && ctx->accum_owned) (void)hipFree(ctx->accum);
    ctx->accum = nullptr;
    ctx->accum_owned = false;
}

void clearCancel(rt_context* ctx) {
    ctx->cancel.store(0);
    if (ctx->cancel_host) __atomic_store_n(ctx->cancel_host, 0, __ATOMIC_RELEASE);
}

void tileGeometry(int width, int height, int world, int* tiles_x, int* tiles_y, int* tprl) {
    *tiles_x = (width + rt::kTile - 1) / rt::kTile;
    *tiles_y = (height + rt::kTile - 1) / rt::kTile;
    *tprl = (*tiles_x + world - 1) / world;
}

rt::RenderArgs makeArgs(rt_context* ctx, int first_sample, int n_samples, int max_depth) {
    rt::RenderArgs a{};
    a.sc = ctx->dsc;
    a.accum = ctx->accum;
    a.queue = nullptr;
    a.cancel = ctx->cancellable ? ctx->cancel_dev : nullptr;
    a.counters = ctx->counters;
    a.width = ctx->width;
    a.height = ctx->height;
    a.tiles_x = ctx->tiles_x;
    a.tiles_y = ctx->tiles_y;
    a.tiles_per_row_local = ctx->tprl;
    a.n_local_tiles = ctx->n_local_tiles;
    a.rank = ctx->rank;
    a.world = ctx->world;
    a.first_sample = first_sample;
    a.n_samples = n_samples;
    a.max_depth = max_depth;
    a.lds_bytes = (int)ctx->lds_need;
    a.knob[2] = ctx->lds_refs ? 1 : 0;
    return a;
}

int checkRenderState(rt_context* ctx) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    if (!ctx->scene_loaded) return fail(RT_ERR_STATE, "rt_upload_scene() has not been called");
    if (!ctx->accum) return fail(RT_ERR_STATE, "rt_set_target() has not been called");
    return RT_OK;
}

// Launch shape.  Default: the persistent wavefront loop, three 512-thread workgroups per CU
// whose waves pull chunks of 64 pixel samples from a queue.  RT_KERNEL=simple selects the plain
// one-wave-per-tile kernel (kept for A/B measurements); both produce identical images.
// Where a render call keeps the per-sample colours and the queue counter of its launches.
struct Workspace {
    rt::Radiance* colors;  // null: the context's own buffer, grown on demand
    size_t colors_cap;     // in samples
    unsigned int* queue;
};

int launch(rt_context* ctx, rt::RenderArgs& a, bool count, const Workspace* fixed = nullptr) {
    const int threads = 256;
    const int wavesPerBlock = threads / 64;
    const char* kernel = std::getenv("RT_KERNEL");
    const bool simple = kernel && std::strcmp(kernel, "simple") == 0 && !fixed;
    int blocks;
    if (simple) {
        blocks = (a.n_local_tiles + wavesPerBlock - 1) / wavesPerBlock;
        if (blocks < 1) blocks = 1;
        HIP_TRY(rt::launchRender(a, ctx->use_lds, count, blocks, threads, ctx->stream));
        return RT_OK;
    }
    a.queue = fixed ? fixed->queue : ctx->queue;
    a.knob[0] = envInt("RT_BOX_KEEP", 8);
    a.knob[1] = envInt("RT_SHADE_MIN", 32);
    a.knob[3] = std::max(1, envInt("RT_PRIM_KEEP", 16));
    a.timing = envInt("RT_TIMING", 0);
    // tables in LDS: three 512-thread workgroups per CU = 6 waves per SIMD (two when the staged scene is too
    // big); tables in global memory: eight 256-thread workgroups = 8 waves per SIMD (see renderWavefront)
    const int defThreads = ctx->use_lds ? 512 : 256;
    int wfThreads = envInt("RT_THREADS", defThreads);
    if (wfThreads < 64 || wfThreads > (ctx->use_lds ? WF_MAX_THREADS : WF_GLOBAL_THREADS) || wfThreads % 64 != 0) wfThreads = defThreads;
    const int wfWaves = wfThreads / 64;
    const int perCu = !ctx->use_lds ? 8 * 256 / wfThreads : (ctx->lds_need > kLdsThreeGroups ? 2 : 3);
    const int maxBlocks = ctx->n_cus * envInt("RT_BLOCKS_PER_CU", perCu);

    // The kernel writes one radiance per pixel sample into the colour buffer and a second kernel
    // adds them per pixel in sample order; launches are cut so that the buffer stays within
    // RT_COLORS_MB (default 8192) and the queue index within 32 bits.
    const long long nSlots = (long long)a.n_local_tiles * 64;
    size_t budget = (size_t)std::max(16, envInt("RT_COLORS_MB", 8192)) << 20;
    if (fixed) {
        if ((size_t)nSlots > fixed->colors_cap)
            return fail(RT_ERR_LIMIT, "rt_render_noalloc: this context's share of the frame (" + std::to_string(nSlots) +
                                          " pixel slots) exceeds the fixed scratch of " +
                                          std::to_string(fixed->colors_cap) + " slots");
        budget = std::min(budget, fixed->colors_cap * sizeof(rt::Radiance));
    }
    // Two passes (primaryPass, then the loop over its path records) for scenes whose tables are in LDS: the camera
    // rays of a tile are walked together, a full wave at a time, and the loop's SHADE and NEWPATH become one phase
    // (bench workload 18.7 ms against 19.95 for the single pass).  Costs 48 bytes of records per pixel sample of a
    // launch (RT_RECORDS_MB, default 25600, bounds them: four 1080p x 64 spp launches' worth; a 4K frame of 1024 spp
    // takes 1,160 ms in launches of 64 spp and 1,218 ms in launches of 16).  Never for the fixed-scratch entry point or with the
    // reference-order walk forced.  A second kernel costs a start-up and a tail of its own (a path of 50 bounces takes
    // 0.3-0.4 ms whatever else the GPU does), so small launches stay with the single pass: 1080p x 16 spp 5.55 against
    // 5.73 ms, 1200x675 x 10 spp 2.05 against 2.41; the lines cross at about 45 M samples.  RT_TWO_PASS=1 / 0: always /
    // never.
    constexpr long long kTwoPassMinSamples = 48000000ll;
    const int twoPassKnob = envInt("RT_TWO_PASS", -1);
    const bool twoPass = !fixed && ctx->use_lds && !a.sc.force_ref && a.max_depth >= 1 && twoPassKnob != 0 &&
                         (twoPassKnob > 0 || nSlots * (long long)a.n_samples >= kTwoPassMinSamples);
    // the loop over path records ends its box bursts a little earlier than the single-pass loop (18.71 against 18.80 ms)
    if (twoPass && !std::getenv("RT_BOX_KEEP")) a.knob[0] = 16;
    constexpr size_t kRecordBytes = 48;
    if (twoPass) budget = std::min(budget, ((size_t)std::max(16, envInt("RT_RECORDS_MB", 25600)) << 20) / kRecordBytes * sizeof(rt::Radiance));
    size_t maxSamples = std::max<size_t>(1, budget / ((size_t)nSlots * sizeof(rt::Radiance)));
    maxSamples = std::min<size_t>(maxSamples, (size_t)((1ll << 31) / nSlots));
    maxSamples = std::max<size_t>(1, std::min<size_t>(maxSamples, (size_t)a.n_samples));
    const int first = a.first_sample, total = a.n_samples;
    // The buffers of the largest launch, allocated up front; when the device has not that much memory left the
    // launches are halved until they fit.
    while (!fixed) {
        const size_t need = (size_t)nSlots * maxSamples;
        hipError_t e = hipSuccess;
        if (need > ctx->colors_cap) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (ctx->colors) (void)hipFree(ctx->colors);
            ctx->colors = nullptr;
            ctx->colors_cap = 0;
            e = hipMalloc((void**)&ctx->colors, need * sizeof(rt::Radiance));
            if (e == hipSuccess) ctx->colors_cap = need;
        }
        if (e == hipSuccess && twoPass && need > ctx->records_cap) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (ctx->records) (void)hipFree(ctx->records);
            ctx->records = nullptr;
            ctx->records_cap = 0;
            // + one byte per chunk of 64 for the record counts, read a dword at a time
            e = hipMalloc((void**)&ctx->records, need * kRecordBytes + need / 64 + 16);
            if (e == hipSuccess) ctx->records_cap = need;
        }
        if (e == hipSuccess) break;
        (void)hipGetLastError();
        if (e != hipErrorOutOfMemory || maxSamples == 1) HIP_TRY(e);
        maxSamples = (maxSamples + 1) / 2;
    }
    for (int s0 = 0; s0 < total; s0 += (int)maxSamples) {
        const int ns = std::min((int)maxSamples, total - s0);
        a.first_sample = first + s0;
        a.n_samples = ns;
        a.colors = fixed ? fixed->colors : ctx->colors;
        a.records = ctx->records;
        a.records_cap = (unsigned int)ctx->records_cap;
        a.record_counts = reinterpret_cast<uint8_t*>(ctx->records) + ctx->records_cap * kRecordBytes;
        a.chunk_group = std::max(1, std::min(64, envInt("RT_CHUNK_GROUP", 1)));
        a.n_queues = std::max(1, std::min(rt::kMaxQueues, envInt("RT_QUEUES", 16)));
        int blocks = maxBlocks;
        const long long needed = ((long long)a.n_local_tiles * ns + wfWaves - 1) / wfWaves;
        if (blocks > needed) blocks = (int)needed;
        if (blocks < 1) blocks = 1;
        HIP_TRY(hipMemsetAsync(a.queue, 0, rt::kQueueBytes, ctx->stream));
        if (twoPass)
            HIP_TRY(rt::launchTwoPass(a, count, blocks, wfThreads, ctx->stream));
        else
            HIP_TRY(rt::launchWavefront(a, ctx->use_lds, count, blocks, wfThreads, ctx->stream));
    }
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }
const char* rt_version(void) { return "rtiow-hip 0.1 (gfx950)"; }

// ------------------------------------------------------------------ scene (host)
int rt_scene_create(rt_scene** out) {
    if (!out) return fail(RT_ERR_INVALID, "null output pointer");
    *out = new rt_scene();
    return RT_OK;
}
void rt_scene_destroy(rt_scene* s) { delete s; }

int rt_scene_rng_floats(rt_scene* s, int n, float* out) {
    if (!s || n < 0 || (n && !out)) return fail(RT_ERR_INVALID, "bad arguments");
    for (int i = 0; i < n; i++) out[i] = s->sc.randomFloat();
    return RT_OK;
}
int rt_scene_add_material(rt_scene* s, int kind, float r, float g, float b, float param) {
    if (!s || kind < RT_MAT_LAMBERTIAN || kind > RT_MAT_LIGHT) return fail(RT_ERR_INVALID, "bad material kind");
    return s->sc.addMaterial(kind, r, g, b, param);
}
int rt_scene_add_texture_rgb8(rt_scene* s, int width, int height, const uint8_t* rgb) {
    if (!s || width <= 0 || height <= 0 || !rgb) return fail(RT_ERR_INVALID, "bad texture");
    return s->sc.addTexture(width, height, rgb);
}
// stbi_loadf's 8-bit -> float map, stbi__ldr_to_hdr: (float)(pow(byte / 255.0f, gamma) * scale) with
// the float globals gamma = 2.2f and scale = 1.0f (R/vendor/stb_image/stb_image.h:1553-1554,1840-1861).
// The reference compiles that file as C++, where pow(float, float) is the float overload.
int rt_texture_lut256(float* out256) {
    if (!out256) return fail(RT_ERR_INVALID, "null output pointer");
    for (int i = 0; i < 256; i++) out256[i] = (float)(std::pow(i / 255.0f, 2.2f) * 1.0f);
    return RT_OK;
}
static bool readWholeFile(const char* path, std::vector<uint8_t>* out) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    bool ok = n >= 0;
    if (ok) {
        out->resize((size_t)n);
        ok = n == 0 || std::fread(out->data(), 1, (size_t)n, f) == (size_t)n;
    }
    std::fclose(f);
    return ok;
}

int rt_image_decode(const uint8_t* file_bytes, size_t n_bytes, int* width, int* height, int* channels, uint8_t* pixels,
                    size_t pixels_capacity) {
    if (!file_bytes || !width || !height || !channels) return fail(RT_ERR_INVALID, "bad arguments");
    rt::Image img;
    std::string err;
    if (!rt::decodeImage(file_bytes, n_bytes, &img, &err)) return fail(RT_ERR_INVALID, err);
    *width = img.w;
    *height = img.h;
    *channels = img.channels;
    if (pixels) {
        if (pixels_capacity < img.px.size()) return fail(RT_ERR_LIMIT, "pixel buffer too small");
        std::memcpy(pixels, img.px.data(), img.px.size());
    }
    return RT_OK;
}

int rt_scene_add_texture_file(rt_scene* s, const char* path) {
    if (!s || !path) return fail(RT_ERR_INVALID, "bad arguments");
    std::vector<uint8_t> bytes;
    if (!readWholeFile(path, &bytes)) return fail(RT_ERR_INVALID, std::string("cannot read ") + path);
    rt::Image img;
    std::string err;
    if (!rt::decodeImage(bytes.data(), bytes.size(), &img, &err)) return fail(RT_ERR_INVALID, std::string(path) + ": " + err);
    std::vector<uint8_t> rgb;
    rt::toRgb(img, &rgb);
    return s->sc.addTexture(img.w, img.h, rgb.data());
}

int rt_scene_add_material_pbr(rt_scene* s, int td, int tr, int tn) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    const int nt = (int)s->sc.textures.size();
    if (td < 0 || td >= nt || tr >= nt || tn >= nt) return fail(RT_ERR_INVALID, "texture index out of range");
    return s->sc.addPbr(td, tr < 0 ? -1 : tr, tn < 0 ? -1 : tn);
}
static bool matOk(rt_scene* s, int m) { return s && m >= 0 && m < (int)s->sc.materials.size(); }
int rt_scene_add_sphere(rt_scene* s, float cx, float cy, float cz, float radius, int material) {
    if (!matOk(s, material)) return fail(RT_ERR_INVALID, "material index out of range");
    return s->sc.addSphere(cx, cy, cz, radius, material);
}
int rt_scene_add_triangle(rt_scene* s, const float* pos9, const float* normal9, const float* uv6, int material) {
    if (!matOk(s, material) || !pos9 || !normal9 || !uv6) return fail(RT_ERR_INVALID, "bad triangle arguments");
    return s->sc.addTriangle(pos9, normal9, uv6, material);
}
int rt_scene_add_box(rt_scene* s, const float* p0, const float* p1, int material, const float* model16) {
    if (!matOk(s, material) || !p0 || !p1 || !model16) return fail(RT_ERR_INVALID, "bad box arguments");
    return s->sc.addBox(p0, p1, material, model16);
}
int rt_scene_add_mesh(rt_scene* s, const float* model16, int n_vertices, const float* positions, const float* normals,
                      const float* uvs, int n_triangles, const uint32_t* indices, int material, int max_depth) {
    if (!matOk(s, material) || !model16 || !positions || !indices || n_vertices <= 0 || n_triangles <= 0)
        return fail(RT_ERR_INVALID, "bad mesh arguments");
    int r = s->sc.addMesh(model16, n_vertices, positions, normals, uvs, n_triangles, indices, material, max_depth);
    if (r < 0) return fail(RT_ERR_INVALID, "mesh index out of range");
    return r;
}
int rt_scene_add_mesh_gltf(rt_scene* s, const float* model16, const char* gltf_path, int material, int max_depth) {
    if (!matOk(s, material) || !model16 || !gltf_path) return fail(RT_ERR_INVALID, "bad mesh arguments");
    std::string err;
    int r = s->sc.addMeshGltf(model16, gltf_path, material, max_depth, false, true, &err);
    if (r < 0) return fail(RT_ERR_INVALID, err);
    return r;
}
int rt_scene_add_mesh_file(rt_scene* s, const float* model16, const char* gltf_path, int max_depth, int material,
                           int per_primitive_materials) {
    if (!s || !model16 || !gltf_path) return fail(RT_ERR_INVALID, "bad mesh arguments");
    if (material >= 0 && !matOk(s, material)) return fail(RT_ERR_INVALID, "material index out of range");
    std::string err;
    int r = s->sc.addMeshGltf(model16, gltf_path, material < 0 ? -1 : material, max_depth, per_primitive_materials != 0,
                              false, &err);
    if (r < 0) return fail(RT_ERR_INVALID, err);
    return r;
}
int rt_scene_random_scene(rt_scene* s) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    s->sc.randomScene();
    return RT_OK;
}
int rt_scene_cornell_box(rt_scene* s) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    s->sc.cornellBox();
    return RT_OK;
}
int rt_scene_begin_bvh_node(rt_scene* s, const float* box6) {
    if (!s || !box6) return fail(RT_ERR_INVALID, "bad arguments");
    return s->sc.beginGroup(true, box6);
}
int rt_scene_begin_mesh(rt_scene* s, const float* box6) { return rt_scene_begin_bvh_node(s, box6); }
int rt_scene_begin_list(rt_scene* s) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    return s->sc.beginGroup(false, nullptr);
}
int rt_scene_end_group(rt_scene* s) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    if (s->sc.endGroup() < 0) return fail(RT_ERR_STATE, "rt_scene_end_group() without an open group");
    return RT_OK;
}
int rt_scene_begin_prototype(rt_scene* s) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    const int id = s->sc.beginPrototype();
    if (id < 0) return fail(RT_ERR_STATE, "a prototype or a group is already open");
    return id;
}
int rt_scene_end_prototype(rt_scene* s) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    const int id = s->sc.endPrototype();
    if (id < 0) return fail(RT_ERR_STATE, "rt_scene_end_prototype() without an open prototype, or with a group still open");
    return id;
}
int rt_scene_add_instance(rt_scene* s, int prototype, const float* model16) {
    if (!s || !model16) return fail(RT_ERR_INVALID, "null argument");
    for (int i = 0; i < 16; i++)
        if (!std::isfinite(model16[i])) return fail(RT_ERR_INVALID, "model matrix is not finite");
    const int id = s->sc.addInstance(prototype, model16);
    if (id < 0) return fail(RT_ERR_INVALID, "unknown, unfinished or empty prototype (or one that holds an instance itself)");
    return id;
}
int rt_scene_wrap_bvh(rt_scene* s) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    if (s->sc.groupsOpen()) return fail(RT_ERR_STATE, "a group is still open");
    if (s->sc.world.empty()) return fail(RT_ERR_STATE, "cannot build a BVH over an empty world");
    s->sc.wrapBvh();
    return RT_OK;
}
int rt_scene_set_camera(rt_scene* s, const float* lookfrom3, const float* lookat3, const float* vup3, float vfov,
                        float aspect_ratio, float aperture, float focus_dist) {
    if (!s || !lookfrom3 || !lookat3 || !vup3) return fail(RT_ERR_INVALID, "bad camera arguments");
    s->sc.setCamera(lookfrom3, lookat3, vup3, vfov, aspect_ratio, aperture, focus_dist);
    return RT_OK;
}
int rt_scene_get_camera(const rt_scene* s, float* out22) {
    if (!s || !out22) return fail(RT_ERR_INVALID, "bad arguments");
    std::memcpy(out22, &s->sc.camera, sizeof(float) * 22);
    return RT_OK;
}
int rt_scene_set_camera_members(rt_scene* s, const float* in22) {
    if (!s || !in22) return fail(RT_ERR_INVALID, "bad arguments");
    // the search structure orders its entries by distance from the eye: a moved eye makes it stale
    // (still correct, only slower), and the next flatten() rebuilds it
    if (std::memcmp(in22, s->sc.camera.origin, sizeof(float) * 3) != 0) s->sc.flat.fast_stale = true;
    std::memcpy(&s->sc.camera, in22, sizeof(float) * 22);
    return RT_OK;
}
int rt_scene_set_normal_mapping(rt_scene* s, int on) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    s->sc.normal_mapping = on != 0;
    return RT_OK;
}

int rt_scene_set_background(rt_scene* s, float r, float g, float b) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    s->sc.background[0] = r;
    s->sc.background[1] = g;
    s->sc.background[2] = b;
    return RT_OK;
}

int rt_scene_flatten(rt_scene* s, rt_flat_info* info) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    if (s->sc.groupsOpen()) return fail(RT_ERR_STATE, "a group is still open (rt_scene_end_group() missing)");
    s->sc.flatten();
    if (info) {
        const rt::Scene& sc = s->sc;
        info->n_nodes = (int)(sc.flat.nodes.size() / 8);
        info->n_prim_refs = (int)sc.flat.prim_refs.size();
        info->n_spheres = (int)sc.spheres.size();
        info->n_triangles = (int)sc.triangles.size();
        info->n_materials = (int)sc.materials.size();
        info->n_textures = (int)sc.textures.size();
        info->max_leaf = sc.flat.max_leaf;
        rt::DevScene d{};
        d.n_nodes = info->n_nodes;
        d.n_prim_refs = info->n_prim_refs;
        d.n_fast_nodes = (int)(sc.flat.fast_nodes.size() / 8);
        d.n_fast_refs = (int)sc.flat.fast_refs.size();
        d.n_spheres = info->n_spheres;
        d.n_tris = info->n_triangles;
        d.n_mats = info->n_materials;
        info->fits_lds = rt::ldsSceneBytes(d) <= kMaxLdsScene && rt::ldsWalkEntriesFit(d);
    }
    return RT_OK;
}

int rt_scene_get_flat(const rt_scene* s, uint32_t* nodes8, uint32_t* prim_refs, float* spheres4, int32_t* sphere_mat,
                      float* tri_pos9, float* tri_normal9, float* tri_uv6, int32_t* tri_mat, int32_t* mat_kind,
                      float* mat_rgbp4) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    const rt::Scene& sc = s->sc;
    if (!sc.flat.valid) return fail(RT_ERR_STATE, "rt_scene_flatten() has not been called");
    if (nodes8) std::memcpy(nodes8, sc.flat.nodes.data(), sc.flat.nodes.size() * 4);
    if (prim_refs) std::memcpy(prim_refs, sc.flat.prim_refs.data(), sc.flat.prim_refs.size() * 4);
    for (size_t i = 0; i < sc.spheres.size(); i++) {
        if (spheres4) {
            std::memcpy(&spheres4[4 * i], sc.spheres[i].c, 12);
            spheres4[4 * i + 3] = sc.spheres[i].r;
        }
        if (sphere_mat) sphere_mat[i] = sc.spheres[i].mat;
    }
    for (size_t i = 0; i < sc.triangles.size(); i++) {
        if (tri_pos9) std::memcpy(&tri_pos9[9 * i], sc.triangles[i].pos, 36);
        if (tri_normal9) std::memcpy(&tri_normal9[9 * i], sc.triangles[i].nrm, 36);
        if (tri_uv6) std::memcpy(&tri_uv6[6 * i], sc.triangles[i].uv, 24);
        if (tri_mat) tri_mat[i] = sc.triangles[i].mat;
    }
    for (size_t i = 0; i < sc.materials.size(); i++) {
        if (mat_kind) mat_kind[i] = sc.materials[i].kind;
        if (mat_rgbp4) {
            mat_rgbp4[4 * i] = sc.materials[i].r;
            mat_rgbp4[4 * i + 1] = sc.materials[i].g;
            mat_rgbp4[4 * i + 2] = sc.materials[i].b;
            mat_rgbp4[4 * i + 3] = sc.materials[i].param;
        }
    }
    return RT_OK;
}

int rt_scene_get_search_structure(const rt_scene* s, int* n_entries, uint32_t* entries8, int* n_refs, uint32_t* refs) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    const rt::Scene& sc = s->sc;
    if (!sc.flat.valid) return fail(RT_ERR_STATE, "rt_scene_flatten() has not been called");
    if (n_entries) *n_entries = (int)(sc.flat.fast_nodes.size() / 8);
    if (n_refs) *n_refs = (int)sc.flat.fast_refs.size();
    if (entries8) std::memcpy(entries8, sc.flat.fast_nodes.data(), sc.flat.fast_nodes.size() * 4);
    if (refs) std::memcpy(refs, sc.flat.fast_refs.data(), sc.flat.fast_refs.size() * 4);
    return RT_OK;
}

int rt_scene_get_search_structure_octants(const rt_scene* s, int* stride, uint32_t* entries8) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    const rt::Scene& sc = s->sc;
    if (!sc.flat.valid) return fail(RT_ERR_STATE, "rt_scene_flatten() has not been called");
    if (stride) *stride = sc.flat.fast_oct_stride;
    if (entries8 && !sc.flat.fast_oct.empty()) std::memcpy(entries8, sc.flat.fast_oct.data(), sc.flat.fast_oct.size() * 4);
    return RT_OK;
}

// ------------------------------------------------------------------ device
int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rt_create(int device, rt_context** out) {
    if (!out) return fail(RT_ERR_INVALID, "null output pointer");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(RT_ERR_NO_DEVICE, "no HIP device available; this library has no CPU path");
    if (device < 0 || device >= n) return fail(RT_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    if (device >= kMaxDevices) return fail(RT_ERR_LIMIT, "device index beyond the library's table");
    {
        // this device's copy of the scratch array: looked up once per device, with that device current
        std::lock_guard<std::mutex> lock(g_scratchMutex);
        if (!g_scratchPtr[device]) {
            unsigned char* p = nullptr;
            HIP_TRY(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
            g_scratchPtr[device] = p;
        }
    }
    rt_context* ctx = new rt_context();
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cus = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    ctx->stream = ctx->own_stream;
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->queue, rt::kQueueBytes);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->counters, 256);
    if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->cancel_host, sizeof(int), hipHostMallocMapped);
    if (e == hipSuccess) {
        *ctx->cancel_host = 0;
        e = hipHostGetDevicePointer((void**)&ctx->cancel_dev, ctx->cancel_host, 0);
    }
    if (e != hipSuccess) {
        rt_destroy(ctx);
        return fail(RT_ERR_HIP, std::string("context set-up: ") + hipGetErrorString(e));
    }
    *out = ctx;
    return RT_OK;
}

void rt_destroy(rt_context* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    freeScene(ctx);
    freeTarget(ctx);
    if (ctx->colors) (void)hipFree(ctx->colors);
    if (ctx->records) (void)hipFree(ctx->records);
    if (ctx->resolved) (void)hipFree(ctx->resolved);
    if (ctx->queue) (void)hipFree(ctx->queue);
    if (ctx->counters) (void)hipFree(ctx->counters);
    if (ctx->cancel_host) (void)hipHostFree(ctx->cancel_host);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int rt_set_stream(rt_context* ctx, void* hip_stream) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return RT_OK;
}

int rt_set_device_build(rt_context* ctx, int on) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    ctx->device_build = on < 0 ? -1 : (on != 0 ? 1 : 0);
    return RT_OK;
}

int rt_last_build_ms(rt_context* ctx, float* ms) {
    if (!ctx || !ms) return fail(RT_ERR_INVALID, "bad arguments");
    *ms = ctx->build_ms;
    return RT_OK;
}

int rt_read_search_structure(rt_context* ctx, int* n_entries, uint32_t* entries8, int* n_refs, uint32_t* refs) {
    if (!ctx || !ctx->scene_loaded) return fail(RT_ERR_STATE, "rt_upload_scene() has not been called");
    HIP_TRY(hipSetDevice(ctx->device));
    const rt::DevScene& d = ctx->dsc;
    // of eight direction-ordered copies the first one, without its closing entry and with the links
    // that mean "the walk is over" set to the entry count
    const bool oct = d.fast_oct_stride > 0;
    const int nE = oct ? d.fast_oct_stride - 1 : d.n_fast_nodes;
    if (n_entries) *n_entries = nE;
    if (n_refs) *n_refs = d.n_fast_refs;
    if (entries8) HIP_TRY(hipMemcpyAsync(entries8, d.fast_nodes, (size_t)nE * 32, hipMemcpyDeviceToHost, ctx->stream));
    if (refs) HIP_TRY(hipMemcpyAsync(refs, d.fast_refs, (size_t)d.n_fast_refs * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (entries8 && oct)
        for (int e = 0; e < nE; e++)
            if (entries8[(size_t)e * 8 + 6] > (uint32_t)nE) entries8[(size_t)e * 8 + 6] = (uint32_t)nE;
    return RT_OK;
}

int rt_set_reference_traversal(rt_context* ctx, int on) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    ctx->force_ref = on ? 1 : 0;
    ctx->dsc.force_ref = (ctx->force_ref || ctx->huge_coords) ? 1 : 0;
    return RT_OK;
}

int rt_upload_scene(rt_context* ctx, rt_scene* s) {
    if (!ctx || !s) return fail(RT_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(ctx->device));
    rt::Scene& sc = s->sc;
    if (sc.groupsOpen()) return fail(RT_ERR_STATE, "a group is still open (rt_scene_end_group() missing)");
    if (!sc.instances.empty() && sc.normal_mapping)
        return fail(RT_ERR_INVALID, "normal mapping is not available for scenes with instances");
    // With a device-side build the host prepares the reference-order arrays, the guard tables and the
    // builder's inputs only; a scene flattened earlier for a device build gets its host-built structure
    // now if this context wants one.
    // Left to itself (-1) the context builds on the device from RT_DEVICE_BUILD_MIN primitives up (default 1 M: there the
    // host build takes 0.4 s, at 10 M three seconds, and the device-built tree costs 6 to 8 % more per frame), unless
    // the scene already carries a host-built structure from an earlier flatten.
    ctx->build_ms = 0.0f;
    const bool hostBuiltAlready = sc.flat.valid && !sc.flat.fast_stale && !sc.flat.fast_nodes.empty();
    const size_t autoMin = (size_t)std::max(1, envInt("RT_DEVICE_BUILD_MIN", 1000000));
    const bool wantDevice = ctx->device_build == 1 ||
                            (ctx->device_build < 0 && !hostBuiltAlready && sc.spheres.size() + sc.triangles.size() >= autoMin);
    PhaseClock clock;
    sc.flat.defer_search_build = wantDevice;
    sc.flatten();
    if (!wantDevice && sc.flat.fast_nodes.empty()) sc.buildFastStructure();
    sc.flat.defer_search_build = false;
    clock.lap(wantDevice ? "flatten (inputs of a device build)" : "flatten + host build");
    const bool deviceBuild = wantDevice && !sc.flat.world_refs.empty();
    for (const auto& m : sc.materials)
        if (m.kind == RT_MAT_PBR && (m.tex_d < 0 || m.tex_d >= (int)sc.textures.size()))
            return fail(RT_ERR_INVALID, "PBR material without a diffuse texture");
    freeScene(ctx);

    const int nNodes = (int)(sc.flat.nodes.size() / 8);
    if ((uint32_t)nNodes >= rt::kNodeSkipMask) return fail(RT_ERR_LIMIT, "too many BVH entries");
    // the search structure's fused box test multiplies plane coordinates by reciprocal directions
    // of up to 1e30: beyond 1e7 that could overflow, so such scenes walk the reference-order array
    std::atomic<bool> hugeSeen{false};
    for (const auto& sp : sc.spheres)
        for (int k = 0; k < 3; k++)
            if (!(std::fabs(sp.c[k]) + std::fabs(sp.r) < 1e7f)) hugeSeen = true;
    std::vector<float4> spheres(sc.spheres.size());
    std::vector<int> sphereMat(sc.spheres.size());
    for (size_t i = 0; i < sc.spheres.size(); i++) {
        spheres[i] = make_float4(sc.spheres[i].c[0], sc.spheres[i].c[1], sc.spheres[i].c[2], sc.spheres[i].r);
        sphereMat[i] = sc.spheres[i].mat;
    }
    // (the staging arrays are filled by the threads that first touch their pages)
    const size_t nTri = sc.triangles.size();
    std::unique_ptr<float4[]> triI(new float4[nTri * 3 + 1]), triS(new float4[nTri * 4 + 1]);
    std::unique_ptr<int[]> triMat(new int[nTri + 1]);
    rt::parallelPieces(nTri, [&](size_t first, size_t last) {
      bool huge = false;
      for (size_t i = first; i < last; i++) {
        const rt::HostTriangle& t = sc.triangles[i];
        for (int k = 0; k < 9; k++) huge = huge || !(std::fabs(t.pos[k]) < 1e7f);
        // edge1 = v1 - v0, edge2 = v2 - v0 (R/src/Core/Hittable.h:64-65), computed once here
        triI[3 * i] = make_float4(t.pos[0], t.pos[1], t.pos[2], 0.f);
        triI[3 * i + 1] = make_float4(t.pos[3] - t.pos[0], t.pos[4] - t.pos[1], t.pos[5] - t.pos[2], 0.f);
        triI[3 * i + 2] = make_float4(t.pos[6] - t.pos[0], t.pos[7] - t.pos[1], t.pos[8] - t.pos[2], 0.f);
        triS[4 * i] = make_float4(t.nrm[0], t.nrm[1], t.nrm[2], t.uv[0]);
        triS[4 * i + 1] = make_float4(t.nrm[3], t.nrm[4], t.nrm[5], t.uv[1]);
        triS[4 * i + 2] = make_float4(t.nrm[6], t.nrm[7], t.nrm[8], t.uv[2]);
        triS[4 * i + 3] = make_float4(t.uv[3], t.uv[4], t.uv[5], 0.f);
        triMat[i] = t.mat;
      }
      if (huge) hugeSeen = true;
    });
    const bool hugeCoords = hugeSeen;
    // Tangent frame per triangle, for scenes with normal mapping switched on: the uv-delta construction of
    // R/src/Core/Hittable.h:110-125 with each uv difference paired with the edge to the same vertex
    // (the reference pairs them crosswise, one reason its mapped normal "doesn't work properly yet").
    std::vector<float4> triF;
    if (sc.normal_mapping) {
        triF.resize(sc.triangles.size() * 2);
        rt::parallelPieces(nTri, [&](size_t first, size_t last) {
            for (size_t i = first; i < last; i++) {
                const rt::HostTriangle& t = sc.triangles[i];
                float tan[3], bit[3];
                const bool ok = rt::tangentFrame(t.pos, t.uv, tan, bit);
                triF[2 * i] = make_float4(tan[0], tan[1], tan[2], ok ? 1.0f : 0.0f);
                triF[2 * i + 1] = make_float4(bit[0], bit[1], bit[2], 0.0f);
            }
        });
    }
    std::vector<float4> mats(sc.materials.size() * 2);
    for (size_t i = 0; i < sc.materials.size(); i++) {
        const rt::HostMaterial& m = sc.materials[i];
        mats[2 * i] = make_float4(m.r, m.g, m.b, m.param);
        int k[4] = {m.kind, m.tex_d, m.tex_r, m.tex_n};
        float4 f;
        std::memcpy(&f, k, 16);
        mats[2 * i + 1] = f;
    }
    std::vector<uint32_t> texels;
    std::vector<int4> texDesc(sc.textures.size());
    for (size_t i = 0; i < sc.textures.size(); i++) {
        const rt::HostTexture& t = sc.textures[i];
        texDesc[i] = make_int4((int)texels.size(), t.w, t.h, 0);
        size_t n = (size_t)t.w * t.h, base = texels.size();
        if (base + n > (size_t)0x7FFFFFFF) return fail(RT_ERR_LIMIT, "texture pool beyond 2^31 texels");
        texels.resize(base + n);
        rt::parallelPieces(n, [&](size_t first, size_t last) {
            for (size_t p = first; p < last; p++)
                texels[base + p] = (uint32_t)t.rgb[3 * p] | ((uint32_t)t.rgb[3 * p + 1] << 8) | ((uint32_t)t.rgb[3 * p + 2] << 16);
        });
    }
    std::vector<float> lut(256);  // applied per channel at sampling time
    rt_texture_lut256(lut.data());

    clock.lap("staging of primitives and texels");
    rt::DevScene d{};
    int rc;
    if ((rc = devUploadRaw<float4>(ctx, sc.flat.nodes.data(), sc.flat.nodes.size() / 4, &d.nodes))) return rc;
    if ((rc = devUpload(ctx, sc.flat.prim_refs, &d.prim_refs))) return rc;
    // Scenes whose tables stay in global memory walk the direction-ordered copies of the search
    // structure (same decision as for LDS below, taken here on the counts alone).
    bool octants = false;
    {
        rt::DevScene probe{};
        probe.n_nodes = nNodes;
        probe.n_prim_refs = (int)sc.flat.prim_refs.size();
        probe.n_fast_nodes = (int)(sc.flat.fast_nodes.size() / 8);
        probe.n_fast_refs = (int)sc.flat.fast_refs.size();
        probe.n_spheres = (int)sc.spheres.size();
        probe.n_tris = (int)sc.triangles.size();
        probe.n_mats = (int)sc.materials.size();
        const bool fitsLds = rt::ldsSceneBytes(probe, false) <= kMaxLdsScene && rt::ldsWalkEntriesFit(probe) && envInt("RT_LDS", -1) != 0;
        octants = !deviceBuild && !fitsLds && sc.flat.fast_oct_stride > 0 && envInt("RT_OCTANTS", 1) != 0;
    }
    const std::vector<uint32_t>& fastSrc = octants ? sc.flat.fast_oct : sc.flat.fast_nodes;
    if (!deviceBuild) {
        if ((rc = devUploadRaw<float4>(ctx, fastSrc.data(), fastSrc.size() / 4, &d.fast_nodes))) return rc;
        if ((rc = devUpload(ctx, sc.flat.fast_refs, &d.fast_refs))) return rc;
    }
    d.fast_oct_stride = octants ? sc.flat.fast_oct_stride : 0;
    if ((rc = devUpload(ctx, sc.flat.sphere_ord, &d.sphere_ord))) return rc;
    if ((rc = devUpload(ctx, sc.flat.tri_ord, &d.tri_ord))) return rc;
    if ((rc = devUpload(ctx, sc.flat.prim_guard, &d.prim_guard))) return rc;
    if ((rc = devUpload(ctx, sc.flat.node_outer, &d.node_outer))) return rc;
    if ((rc = devUpload(ctx, spheres, &d.spheres))) return rc;
    if ((rc = devUpload(ctx, sphereMat, &d.sphere_mat))) return rc;
    if ((rc = devUploadRaw<float4>(ctx, triI.get(), nTri * 3, &d.tri_isect))) return rc;
    if ((rc = devUploadRaw<float4>(ctx, triS.get(), nTri * 4, &d.tri_shade))) return rc;
    if ((rc = devUpload(ctx, triF, &d.tri_frame))) return rc;
    d.normal_mapping = sc.normal_mapping ? 1 : 0;
    d.has_pbr = 0;
    d.has_dielectric = 0;
    for (const auto& m : sc.materials) {
        d.has_pbr |= m.kind == 4 ? 1 : 0;
        d.has_dielectric |= m.kind == 2 ? 1 : 0;
    }
    if ((rc = devUploadRaw<int>(ctx, triMat.get(), nTri, &d.tri_mat))) return rc;
    if ((rc = devUpload(ctx, mats, &d.materials))) return rc;
    if ((rc = devUpload(ctx, texels, &d.texels))) return rc;
    if ((rc = devUpload(ctx, texDesc, &d.tex_desc))) return rc;
    if ((rc = devUpload(ctx, lut, &d.lut))) return rc;
    d.n_nodes = nNodes;
    d.n_world_nodes = sc.flat.n_world_nodes;
    d.n_instances = (int)sc.instances.size();
    {
        std::vector<float4> inst(sc.instances.size() * 5);
        for (size_t i = 0; i < sc.instances.size(); i++) {
            const rt::Instance& in = sc.instances[i];
            for (int c = 0; c < 4; c++) inst[5 * i + c] = make_float4(in.inv[4 * c], in.inv[4 * c + 1], in.inv[4 * c + 2], in.inv[4 * c + 3]);
            const int range[4] = {sc.flat.proto_range[2 * in.prototype], sc.flat.proto_range[2 * in.prototype + 1], 0, 0};
            float4 f;
            std::memcpy(&f, range, 16);
            inst[5 * i + 4] = f;
        }
        if ((rc = devUpload(ctx, inst, &d.instances))) return rc;
    }
    d.n_prim_refs = (int)sc.flat.prim_refs.size();
    d.n_fast_nodes = (int)(fastSrc.size() / 8);
    d.n_fast_refs = (int)sc.flat.fast_refs.size();
    clock.lap("copies to the device");
    if (deviceBuild) {
        // the search structure is built where the primitives already are (csrc/device_bvh.hip)
        const uint32_t* dWorld = nullptr;
        if ((rc = devUpload(ctx, sc.flat.world_refs, &dWorld))) return rc;
        float4* entries = nullptr;
        uint32_t* refs = nullptr;
        int nEntries = 0, stride = 0;
        // eight direction-ordered copies under the host builder's rule: the tables will not fit LDS,
        // and there are at most 2 M entries
        auto wantCopies = [&](int nE) {
            rt::DevScene probe{};
            probe.n_fast_nodes = nE;
            probe.n_fast_refs = (int)sc.flat.world_refs.size();
            probe.n_spheres = (int)sc.spheres.size();
            probe.n_tris = (int)sc.triangles.size();
            probe.n_mats = (int)sc.materials.size();
            const bool fitsLds = rt::ldsSceneBytes(probe, false) <= kMaxLdsScene && rt::ldsWalkEntriesFit(probe) && envInt("RT_LDS", -1) != 0;
            return !fitsLds && nE <= (1 << 21) && envInt("RT_OCTANTS", 1) != 0;
        };
        HIP_TRY(rt::buildSearchStructureOnDevice((int)sc.flat.world_refs.size(), dWorld, d.spheres, d.tri_isect,
                                                 sc.flat.pad_floor, sc.flat.far_typical, sc.flat.centroid_min, sc.flat.centroid_max, sc.camera.origin, ctx->stream,
                                                 wantCopies, &entries, &refs, &nEntries, &stride, &ctx->build_ms));
        ctx->scene_allocs.push_back(entries);
        ctx->scene_allocs.push_back(refs);
        d.fast_nodes = entries;
        d.fast_refs = refs;
        d.n_fast_nodes = stride > 0 ? 8 * stride : nEntries;
        d.fast_oct_stride = stride;
        d.n_fast_refs = (int)sc.flat.world_refs.size();
        clock.lap("device build");
    }
    d.force_ref = (ctx->force_ref || hugeCoords || !sc.instances.empty()) ? 1 : 0;
    d.origin_limit = sc.flat.fast_min_pad * 2097152.0f;  // 2^21, see buildFastStructure()
    d.far_spheres = sc.flat.far_spheres;
    d.far_tris = sc.flat.far_tris;
    for (int k = 0; k < 3; k++) d.far_center[k] = sc.flat.far_center[k];
    d.far_radius = sc.flat.far_radius;
    d.far_rmin = sc.flat.far_rmin;
    d.far_typical = sc.flat.far_typical;
    {
        // farRayPad() is non-zero exactly when (|o - centre| + radius) * 1.0001 > far_typical; the kernels test
        // the squared distance against this first
        const double reach = (double)sc.flat.far_typical / 1.0001 - (double)sc.flat.far_radius;
        d.far_trigger2 = (sc.flat.far_spheres + sc.flat.far_tris) == 0 ? std::numeric_limits<float>::infinity()
                                                                        : (float)(reach > 0 ? reach * reach * 0.9999 : 0.0);
    }
    // the kernels address a search-structure entry by a 32-bit byte offset (32 bytes per entry)
    if (d.n_fast_nodes >= (1 << 27)) return fail(RT_ERR_LIMIT, "search structure beyond 2^27 entries");
    ctx->huge_coords = hugeCoords;
    d.n_spheres = (int)sc.spheres.size();
    d.n_tris = (int)sc.triangles.size();
    d.n_mats = (int)sc.materials.size();
    std::memcpy(d.cam, &sc.camera, sizeof(float) * 22);
    std::memcpy(d.bg, sc.background, sizeof(float) * 3);
    ctx->dsc = d;
    // Scene tables in LDS when they fit.  Three workgroups share a CU's 160 KiB, so the preferred
    // footprint is at most kLdsThreeGroups: all tables if that fits, else all but the
    // reference-order arrays (touched once or twice per accepted hit and by the rare re-trace);
    // failing that, up to kMaxLdsScene with two workgroups per CU; else global memory.
    // RT_LDS overrides: 0 = global memory, 1 = all tables, 2 = without the reference arrays.
    const int ldsMode = envInt("RT_LDS", -1);
    const size_t all = rt::ldsSceneBytes(d, true), part = rt::ldsSceneBytes(d, false);
    if (ldsMode == 1 || ldsMode == 2)
        ctx->lds_refs = ldsMode == 1;
    else
        ctx->lds_refs = all <= kLdsThreeGroups || (part > kLdsThreeGroups && all <= kMaxLdsScene);
    ctx->lds_need = ctx->lds_refs ? all : part;
    ctx->use_lds = ctx->lds_need <= kMaxLdsScene && rt::ldsWalkEntriesFit(d) && ldsMode != 0 && sc.instances.empty();
    ctx->scene_loaded = true;
    return RT_OK;
}

size_t rt_accum_bytes(int width, int height, int world) {
    if (width <= 0 || height <= 0 || world <= 0) return 0;
    int tx, ty, tprl;
    tileGeometry(width, height, world, &tx, &ty, &tprl);
    return (size_t)ty * tprl * 64 * sizeof(float4);
}

static int setTarget(rt_context* ctx, int width, int height, int rank, int world, void* external) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    if (width < 2 || height < 2) return fail(RT_ERR_INVALID, "image must be at least 2x2 (the pixel map divides by W-1, H-1)");
    if (world < 1 || rank < 0 || rank >= world) return fail(RT_ERR_INVALID, "bad rank / world");
    if ((long long)width * height > (1ll << 30)) return fail(RT_ERR_LIMIT, "image beyond 2^30 pixels");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    freeTarget(ctx);
    clearCancel(ctx);
    ctx->width = width;
    ctx->height = height;
    ctx->rank = rank;
    ctx->world = world;
    tileGeometry(width, height, world, &ctx->tiles_x, &ctx->tiles_y, &ctx->tprl);
    ctx->n_local_tiles = ctx->tiles_y * ctx->tprl;
    if (external) {
        ctx->accum = reinterpret_cast<float4*>(external);
        ctx->accum_owned = false;
    } else {
        HIP_TRY(hipMalloc((void**)&ctx->accum, rt_accum_bytes(width, height, world)));
        ctx->accum_owned = true;
        HIP_TRY(hipMemsetAsync(ctx->accum, 0, rt_accum_bytes(width, height, world), ctx->stream));
    }
    return RT_OK;
}
int rt_set_target(rt_context* ctx, int width, int height, int rank, int world) {
    return setTarget(ctx, width, height, rank, world, nullptr);
}
int rt_set_target_external(rt_context* ctx, int width, int height, int rank, int world, void* device_accum) {
    if (!device_accum) return fail(RT_ERR_INVALID, "null accumulation buffer");
    return setTarget(ctx, width, height, rank, world, device_accum);
}

int rt_release_target(rt_context* ctx) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    freeTarget(ctx);
    ctx->width = ctx->height = 0;
    ctx->n_local_tiles = 0;
    return RT_OK;
}

int rt_clear(rt_context* ctx) {
    if (!ctx || !ctx->accum) return fail(RT_ERR_STATE, "rt_set_target() has not been called");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemsetAsync(ctx->accum, 0, rt_accum_bytes(ctx->width, ctx->height, ctx->world), ctx->stream));
    return RT_OK;
}

int rt_render(rt_context* ctx, int first_sample, int n_samples, int max_depth) {
    int rc = checkRenderState(ctx);
    if (rc) return rc;
    if (first_sample < 0 || n_samples < 0) return fail(RT_ERR_INVALID, "negative sample range");
    if ((long long)first_sample + n_samples > (1 << 24))
        return fail(RT_ERR_LIMIT, "sample index beyond 2^24 (the RNG key is the sample index as a float)");
    if (n_samples == 0) return RT_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    rt::RenderArgs a = makeArgs(ctx, first_sample, n_samples, max_depth);
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    rc = launch(ctx, a, false);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->timed = true;
    return RT_OK;
}

size_t rt_noalloc_max_pixels(void) { return kScratchColorBytes / sizeof(rt::Radiance); }

int rt_render_noalloc(rt_context* ctx, int first_sample, int n_samples, int max_depth) {
    int rc = checkRenderState(ctx);
    if (rc) return rc;
    if (first_sample < 0 || n_samples < 0) return fail(RT_ERR_INVALID, "negative sample range");
    if ((long long)first_sample + n_samples > (1 << 24))
        return fail(RT_ERR_LIMIT, "sample index beyond 2^24 (the RNG key is the sample index as a float)");
    if (n_samples == 0) return RT_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    unsigned char* base = g_scratchPtr[ctx->device];
    if (!base) return fail(RT_ERR_STATE, "the scratch array of this device was never looked up");
    Workspace ws{reinterpret_cast<rt::Radiance*>(base + kScratchQueueBytes), kScratchColorBytes / sizeof(rt::Radiance),
                 reinterpret_cast<unsigned int*>(base)};
    rt::RenderArgs a = makeArgs(ctx, first_sample, n_samples, max_depth);
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    rc = launch(ctx, a, false, &ws);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->timed = true;
    return RT_OK;
}

int rt_sync(rt_context* ctx) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_cancel(rt_context* ctx) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    ctx->cancel.store(1);
    if (ctx->cancel_host) __atomic_store_n(ctx->cancel_host, 1, __ATOMIC_RELEASE);  // what the running kernels see
    return RT_OK;
}

// Samples every owned pixel holds: the .w of the first accumulation slot (launches are all-or-nothing).
static int samplesAccumulated(rt_context* ctx, int* out) {
    float4 first;
    HIP_TRY(hipMemcpyAsync(&first, ctx->accum, sizeof first, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *out = (int)first.w;
    return RT_OK;
}

int rt_run(rt_context* ctx, int samples_per_pixel, int samples_per_pass, int max_depth, rt_pass_callback on_pass,
           void* user) {
    int rc = checkRenderState(ctx);
    if (rc) return rc;
    if (samples_per_pixel < 0 || samples_per_pass < 1) return fail(RT_ERR_INVALID, "bad sample counts");
    // The flag is not cleared here: a Cancel() that arrived just before Run() still stops it.  It is
    // cleared when Run() returns.
    rc = rt_clear(ctx);
    int done = 0;
    ctx->cancellable = true;
    while (rc == RT_OK && done < samples_per_pixel) {
        if (ctx->cancel.load()) break;
        int n = std::min(samples_per_pass, samples_per_pixel - done);
        rc = rt_render(ctx, done, n, max_depth);
        if (rc == RT_OK) rc = rt_sync(ctx);
        if (rc != RT_OK || ctx->cancel.load()) break;   // a pass the flag may have cut short is settled below
        done += n;
        if (on_pass) on_pass(done, user);
    }
    ctx->cancellable = false;
    const bool cancelled = ctx->cancel.load() != 0;
    clearCancel(ctx);
    if (rc != RT_OK) return rc;
    if (cancelled) {
        // Launches are all-or-nothing (a launch the flag cut short adds nothing), so every pixel holds the
        // same number of samples; when that is more than the last callback reported, report it.
        int have = done;
        if (ctx->n_local_tiles > 0 && (rc = samplesAccumulated(ctx, &have)) != RT_OK) return rc;
        if (have > done && on_pass) on_pass(have, user);
        return fail(RT_ERR_CANCELLED, "render cancelled");
    }
    return RT_OK;
}

// Resolves the tiled accumulation buffer into the context's row-major staging image (kept between
// calls: a progressive display reads once per pass) and copies it to `dst`.
static int readResolved(rt_context* ctx, int mode, int spp, void* dst, size_t bytes) {
    int rc = checkRenderState(ctx);
    if (rc) return rc;
    if (!dst) return fail(RT_ERR_INVALID, "null destination");
    HIP_TRY(hipSetDevice(ctx->device));
    if (bytes > ctx->resolved_cap) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->resolved) (void)hipFree(ctx->resolved);
        ctx->resolved = nullptr;
        ctx->resolved_cap = 0;
        HIP_TRY(hipMalloc(&ctx->resolved, bytes));
        ctx->resolved_cap = bytes;
    }
    void* d = ctx->resolved;
    const int only = ctx->world > 1 ? ctx->rank : -1;
    HIP_TRY(rt::launchResolve(ctx->accum, 0, ctx->width, ctx->height, ctx->tprl, ctx->world, only, mode, spp,
                              reinterpret_cast<float4*>(d), reinterpret_cast<uint32_t*>(d), ctx->stream));
    HIP_TRY(hipMemcpyAsync(dst, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}
int rt_read_accum_rgba32f(rt_context* ctx, float* dst) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    return readResolved(ctx, 0, 1, dst, (size_t)ctx->width * ctx->height * 16);
}
int rt_read_rgba8(rt_context* ctx, int samples_per_pixel, uint8_t* dst) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    if (samples_per_pixel < 1) return fail(RT_ERR_INVALID, "samples_per_pixel must be positive");
    return readResolved(ctx, 1, samples_per_pixel, dst, (size_t)ctx->width * ctx->height * 4);
}

int rt_write_ppm(const char* path, int width, int height, const uint8_t* rgba8) {
    if (!path || !rgba8 || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "bad arguments");
    FILE* f = std::fopen(path, "w");
    if (!f) return fail(RT_ERR_INVALID, std::string("cannot open ") + path);
    std::fprintf(f, "P3\n%d %d\n255\n", width, height);
    std::string line;
    for (int y = height - 1; y >= 0; --y) {  // the buffer's row 0 is the bottom of the image
        line.clear();
        for (int x = 0; x < width; ++x) {
            const uint8_t* p = rgba8 + ((size_t)y * width + x) * 4;
            char buf[16];
            int n = std::snprintf(buf, sizeof buf, "%d %d %d\n", (int)p[0], (int)p[1], (int)p[2]);
            line.append(buf, (size_t)n);
        }
        if (std::fwrite(line.data(), 1, line.size(), f) != line.size()) {
            std::fclose(f);
            return fail(RT_ERR_INVALID, std::string("short write to ") + path);
        }
    }
    if (std::fclose(f) != 0) return fail(RT_ERR_INVALID, std::string("cannot close ") + path);
    return RT_OK;
}

namespace {
uint32_t crc32Of(const uint8_t* p, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}
void putBe32(std::vector<uint8_t>& v, uint32_t x) {
    for (int k = 3; k >= 0; k--) v.push_back((uint8_t)(x >> (8 * k)));
}
void pngChunk(std::vector<uint8_t>& file, const char type[4], const std::vector<uint8_t>& data) {
    putBe32(file, (uint32_t)data.size());
    const size_t at = file.size();
    file.insert(file.end(), type, type + 4);
    file.insert(file.end(), data.begin(), data.end());
    putBe32(file, crc32Of(file.data() + at, file.size() - at));
}
}  // namespace

int rt_write_png(const char* path, int width, int height, const uint8_t* rgba8) {
    if (!path || !rgba8 || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "bad arguments");
    if ((long long)width * height > (1ll << 28)) return fail(RT_ERR_LIMIT, "image too large for the PNG writer");
    // scanlines, top row first, each with filter type 0 in front
    const size_t row = 1 + (size_t)width * 3;
    std::vector<uint8_t> raw(row * (size_t)height);
    for (int y = 0; y < height; y++) {
        uint8_t* out = raw.data() + (size_t)y * row;
        const uint8_t* in = rgba8 + (size_t)(height - 1 - y) * width * 4;  // the buffer's row 0 is the bottom row
        *out++ = 0;
        for (int x = 0; x < width; x++, in += 4, out += 3) {
            out[0] = in[0];
            out[1] = in[1];
            out[2] = in[2];
        }
    }
    // a zlib stream of stored (uncompressed) deflate blocks: any PNG reader takes it, and a frame is written in
    // the time it takes to copy it
    std::vector<uint8_t> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t a1 = 1, a2 = 0;  // Adler-32 of the raw bytes
    for (size_t at = 0; at < raw.size() || at == 0;) {
        const size_t n = std::min<size_t>(65535, raw.size() - at);
        z.push_back(at + n >= raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF));
        z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF));
        z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + (long)at, raw.begin() + (long)(at + n));
        for (size_t i = at; i < at + n; i++) {
            a1 = (a1 + raw[i]) % 65521u;
            a2 = (a2 + a1) % 65521u;
        }
        at += n;
        if (n == 0) break;
    }
    putBe32(z, (a2 << 16) | a1);
    std::vector<uint8_t> file = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    putBe32(ihdr, (uint32_t)width);
    putBe32(ihdr, (uint32_t)height);
    ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});  // 8 bits per channel, RGB, deflate, adaptive filtering, no interlace
    pngChunk(file, "IHDR", ihdr);
    pngChunk(file, "IDAT", z);
    pngChunk(file, "IEND", {});
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(RT_ERR_INVALID, std::string("cannot open ") + path);
    const bool ok = std::fwrite(file.data(), 1, file.size(), f) == file.size();
    if (std::fclose(f) != 0 || !ok) return fail(RT_ERR_INVALID, std::string("short write to ") + path);
    return RT_OK;
}

int rt_resolve_gathered(rt_context* ctx, const void* device_payloads, int world, void* device_rgba32f) {
    if (!ctx || !device_payloads || !device_rgba32f) return fail(RT_ERR_INVALID, "null argument");
    if (world != ctx->world) return fail(RT_ERR_INVALID, "world differs from rt_set_target()");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t stride4 = rt_accum_bytes(ctx->width, ctx->height, world) / sizeof(float4);
    HIP_TRY(rt::launchResolve(reinterpret_cast<const float4*>(device_payloads), stride4, ctx->width, ctx->height,
                              ctx->tprl, world, -1, 0, 1, reinterpret_cast<float4*>(device_rgba32f), nullptr,
                              ctx->stream));
    return RT_OK;
}

int rt_resolve_gathered_rgba8(rt_context* ctx, const void* device_payloads, int world, int samples_per_pixel,
                              void* device_rgba8) {
    if (!ctx || !device_payloads || !device_rgba8) return fail(RT_ERR_INVALID, "null argument");
    if (world != ctx->world) return fail(RT_ERR_INVALID, "world differs from rt_set_target()");
    if (samples_per_pixel < 1) return fail(RT_ERR_INVALID, "samples_per_pixel must be positive");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t stride4 = rt_accum_bytes(ctx->width, ctx->height, world) / sizeof(float4);
    HIP_TRY(rt::launchResolve(reinterpret_cast<const float4*>(device_payloads), stride4, ctx->width, ctx->height,
                              ctx->tprl, world, -1, 1, samples_per_pixel, nullptr,
                              reinterpret_cast<uint32_t*>(device_rgba8), ctx->stream));
    return RT_OK;
}

int rt_trace(rt_context* ctx, int n, const float* rays6, int32_t* prim_kind, int32_t* prim_id, float* t, float* normal3,
             int32_t* front_face, float* uv2) {
    if (!ctx || !ctx->scene_loaded) return fail(RT_ERR_STATE, "rt_upload_scene() has not been called");
    if (n < 0 || (n && (!rays6 || !prim_kind || !prim_id || !t))) return fail(RT_ERR_INVALID, "bad arguments");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t N = (size_t)n;
    char* d = nullptr;
    // rays 24N | kind 4N | id 4N | t 4N | normal 12N | front 4N | uv 8N
    HIP_TRY(hipMalloc((void**)&d, 60 * N));
    float* dRays = (float*)d;
    int* dKind = (int*)(d + 24 * N);
    int* dId = (int*)(d + 28 * N);
    float* dT = (float*)(d + 32 * N);
    float* dN = (float*)(d + 36 * N);
    int* dF = (int*)(d + 48 * N);
    float* dUv = (float*)(d + 52 * N);
    hipError_t e = hipMemcpyAsync(dRays, rays6, 24 * N, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = rt::launchTrace(ctx->dsc, n, dRays, dKind, dId, dT, dN, dF, dUv, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(prim_kind, dKind, 4 * N, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(prim_id, dId, 4 * N, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(t, dT, 4 * N, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && normal3) e = hipMemcpyAsync(normal3, dN, 12 * N, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && front_face) e = hipMemcpyAsync(front_face, dF, 4 * N, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && uv2) e = hipMemcpyAsync(uv2, dUv, 8 * N, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RT_ERR_HIP, std::string("trace: ") + hipGetErrorString(e));
    return RT_OK;
}

int rt_scatter_test(rt_context* ctx, int n, const int32_t* material, const float* ray_in6, const float* p3,
                    const float* normal3, const int32_t* front_face, const float* uv2, const uint32_t* key4,
                    float* atten3, float* ray_out6, int32_t* alive, uint32_t* counter_out) {
    if (!ctx || !ctx->scene_loaded) return fail(RT_ERR_STATE, "rt_upload_scene() has not been called");
    if (n <= 0 || !material || !ray_in6 || !p3 || !normal3 || !front_face || !uv2 || !key4 || !atten3 || !ray_out6 ||
        !alive || !counter_out)
        return fail(RT_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t N = (size_t)n;
    // in: mat 4 | ray 24 | p 12 | n 12 | front 4 | uv 8 | key 16 = 80 ; out: atten 12 | ray 24 | alive 4 | counter 4 = 44
    char* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 124 * N));
    char* q = d;
    auto take = [&](size_t bytes) {
        char* r = q;
        q += bytes * N;
        return r;
    };
    int* dMat = (int*)take(4);
    float* dRay = (float*)take(24);
    float* dP = (float*)take(12);
    float* dN = (float*)take(12);
    int* dF = (int*)take(4);
    float* dUv = (float*)take(8);
    unsigned int* dKey = (unsigned int*)take(16);
    float* dAtt = (float*)take(12);
    float* dOut = (float*)take(24);
    int* dAlive = (int*)take(4);
    unsigned int* dCnt = (unsigned int*)take(4);
    hipError_t e = hipMemcpyAsync(dMat, material, 4 * N, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dRay, ray_in6, 24 * N, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dP, p3, 12 * N, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dN, normal3, 12 * N, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dF, front_face, 4 * N, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dUv, uv2, 8 * N, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dKey, key4, 16 * N, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = rt::launchScatter(ctx->dsc, n, dMat, dRay, dP, dN, dF, dUv, dKey, dAtt, dOut, dAlive, dCnt, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(atten3, dAtt, 12 * N, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ray_out6, dOut, 24 * N, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(alive, dAlive, 4 * N, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(counter_out, dCnt, 4 * N, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RT_ERR_HIP, std::string("scatter test: ") + hipGetErrorString(e));
    return RT_OK;
}

int rt_reflectance_test(rt_context* ctx, int n, const float* cosine, const float* ref_idx, float* out) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    if (n <= 0 || !cosine || !ref_idx || !out) return fail(RT_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t N = (size_t)n;
    float* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 12 * N));
    hipError_t e = hipMemcpyAsync(d, cosine, 4 * N, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + N, ref_idx, 4 * N, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = rt::launchReflectance(n, d, d + N, d + 2 * N, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d + 2 * N, 4 * N, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RT_ERR_HIP, std::string("reflectance test: ") + hipGetErrorString(e));
    return RT_OK;
}

int rt_count(rt_context* ctx, int first_sample, int n_samples, int max_depth, rt_counters* out) {
    int rc = checkRenderState(ctx);
    if (rc) return rc;
    if (!out || first_sample < 0 || n_samples < 1) return fail(RT_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    // counting must not disturb the image: run it on a scratch copy of the accumulation buffer
    const size_t bytes = rt_accum_bytes(ctx->width, ctx->height, ctx->world);
    float4* scratch = nullptr;
    HIP_TRY(hipMalloc((void**)&scratch, bytes));
    hipError_t e = hipMemsetAsync(scratch, 0, bytes, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(ctx->counters, 0, 256, ctx->stream);
    int rcl = RT_OK;
    if (e == hipSuccess) {
        rt::RenderArgs a = makeArgs(ctx, first_sample, n_samples, max_depth);
        a.accum = scratch;
        rcl = launch(ctx, a, true);
    }
    unsigned long long host[32] = {0};  // 13 path counters, 7 x 2 scheduler pairs, fallbacks
    if (e == hipSuccess && rcl == RT_OK)
        e = hipMemcpyAsync(host, ctx->counters, sizeof host, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(scratch);
    if (rcl) return rcl;
    if (e != hipSuccess) return fail(RT_ERR_HIP, std::string("count: ") + hipGetErrorString(e));
    out->samples = host[0];
    out->rays = host[1];
    out->node_tests = host[2];
    out->sphere_tests = host[3];
    out->tri_tests = host[4];
    out->hits = host[5];
    out->scatter_lambertian = host[6];
    out->scatter_metal = host[7];
    out->scatter_dielectric = host[8];
    out->scatter_light = host[9];
    out->scatter_pbr = host[10];
    out->misses = host[11];
    out->depth_cutoffs = host[12];
    out->fallbacks = host[27];
    std::memcpy(ctx->sched_stats, host + 13, sizeof ctx->sched_stats);
    return RT_OK;
}

int rt_sched_stats(rt_context* ctx, uint64_t* out14) {
    if (!ctx || !out14) return fail(RT_ERR_INVALID, "bad arguments");
    for (int i = 0; i < 14; i++) out14[i] = ctx->sched_stats[i];
    return RT_OK;
}

int rt_last_render_ms(rt_context* ctx, float* ms) {
    if (!ctx || !ms) return fail(RT_ERR_INVALID, "bad arguments");
    if (!ctx->timed) return fail(RT_ERR_STATE, "no rt_render() has been issued");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    HIP_TRY(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return RT_OK;
}

void* rt_stream(rt_context* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

}  // extern "C"
