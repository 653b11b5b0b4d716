// rt_group_*: several GPUs of one node driven from ONE host process -- what an application shaped like
// the reference's (one process, one render thread, R/src/RaytracingApplication.cpp:244-276) needs in
// order to use more than one device.  The group owns one context per device and the RCCL communicator
// between them.  The image is cut into 8x8-pixel tiles, device i of n renders the tiles with
// (tx + ty) mod n == i into a compact payload, one all-gather over xGMI lays the n payloads end to end
// on every device, and device 0 resolves them into the row-major frame.  Every pixel is summed on
// exactly one device in sample order, so the frame has the same bits for any n.
//
// RCCL is loaded at run time (librccl.so.1), so that the library itself does not depend on it; the
// group entry points fail with RT_ERR_STATE when it cannot be loaded.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/rtiow_hip.h"

namespace {

typedef void* ncclComm_t;
struct Rccl {
    void* handle = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
constexpr int kNcclFloat = 7;  // ncclFloat32 of rccl.h

Rccl& rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    if (!r.handle) return r;
    r.CommInitAll = (int (*)(ncclComm_t*, int, const int*))dlsym(r.handle, "ncclCommInitAll");
    r.CommDestroy = (int (*)(ncclComm_t))dlsym(r.handle, "ncclCommDestroy");
    r.AllGather = (int (*)(const void*, void*, size_t, int, ncclComm_t, hipStream_t))dlsym(r.handle, "ncclAllGather");
    r.GroupStart = (int (*)())dlsym(r.handle, "ncclGroupStart");
    r.GroupEnd = (int (*)())dlsym(r.handle, "ncclGroupEnd");
    r.GetErrorString = (const char* (*)(int))dlsym(r.handle, "ncclGetErrorString");
    r.ok = r.CommInitAll && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd;
    return r;
}

thread_local std::string g_groupErr;

}  // namespace

struct rt_group {
    std::vector<int> devices;
    std::vector<rt_context*> ctx;
    std::vector<ncclComm_t> comm;
    std::vector<void*> payload;   // per device: its accumulation buffer (rt_accum_bytes)
    std::vector<void*> gathered;  // per device: n payloads end to end
    void* frame = nullptr;        // device 0: the resolved row-major image (grown on demand)
    size_t frame_cap = 0;
    int width = 0, height = 0;
};

// The error text of a failed group call is appended to what rt_last_error() of the underlying call
// left; group-level failures are reported through rt_group_last_error().
extern "C" {

const char* rt_group_last_error(void) { return g_groupErr.c_str(); }

static int gfail(int code, const std::string& msg) {
    g_groupErr = msg;
    return code;
}
#define GROUP_TRY_RT(expr)                                                       \
    do {                                                                         \
        int rc_ = (expr);                                                        \
        if (rc_ < 0) return gfail(rc_, std::string(#expr) + ": " + rt_last_error()); \
    } while (0)
#define GROUP_TRY_HIP(expr)                                                                              \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            (void)hipGetLastError();                                                                     \
            return gfail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                 \
        }                                                                                                \
    } while (0)
#define GROUP_TRY_NCCL(expr)                                                                                   \
    do {                                                                                                       \
        int r_ = (expr);                                                                                       \
        if (r_ != 0)                                                                                           \
            return gfail(RT_ERR_HIP, std::string(#expr) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r_) : "RCCL error")); \
    } while (0)

static void freeTargets(rt_group* g) {
    for (size_t i = 0; i < g->devices.size(); i++) {
        (void)hipSetDevice(g->devices[i]);
        if (i < g->payload.size() && g->payload[i]) (void)hipFree(g->payload[i]);
        if (i < g->gathered.size() && g->gathered[i]) (void)hipFree(g->gathered[i]);
    }
    g->payload.assign(g->devices.size(), nullptr);
    g->gathered.assign(g->devices.size(), nullptr);
}

int rt_group_create(int n_devices, const int* devices, rt_group** out) {
    if (!out || n_devices < 1 || n_devices > 64) return gfail(RT_ERR_INVALID, "bad arguments");
    if (!rccl().ok) return gfail(RT_ERR_STATE, "librccl.so.1 could not be loaded; the group entry points need RCCL");
    rt_group* g = new rt_group();
    for (int i = 0; i < n_devices; i++) g->devices.push_back(devices ? devices[i] : i);
    g->payload.assign((size_t)n_devices, nullptr);
    g->gathered.assign((size_t)n_devices, nullptr);
    for (int i = 0; i < n_devices; i++) {
        rt_context* c = nullptr;
        int rc = rt_create(g->devices[(size_t)i], &c);
        if (rc < 0) {
            g_groupErr = std::string("rt_create: ") + rt_last_error();
            rt_group_destroy(g);
            return rc;
        }
        g->ctx.push_back(c);
    }
    g->comm.assign((size_t)n_devices, nullptr);
    int r = rccl().CommInitAll(g->comm.data(), n_devices, g->devices.data());
    if (r != 0) {
        g_groupErr = std::string("ncclCommInitAll: ") + (rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error");
        g->comm.clear();
        rt_group_destroy(g);
        return RT_ERR_HIP;
    }
    *out = g;
    return RT_OK;
}

void rt_group_destroy(rt_group* g) {
    if (!g) return;
    for (rt_context* c : g->ctx) (void)rt_sync(c);
    for (ncclComm_t c : g->comm)
        if (c) (void)rccl().CommDestroy(c);
    freeTargets(g);
    if (g->frame) {
        (void)hipSetDevice(g->devices[0]);
        (void)hipFree(g->frame);
    }
    for (rt_context* c : g->ctx) rt_destroy(c);
    delete g;
}

int rt_group_size(const rt_group* g) { return g ? (int)g->ctx.size() : 0; }
rt_context* rt_group_context(rt_group* g, int i) { return g && i >= 0 && i < (int)g->ctx.size() ? g->ctx[(size_t)i] : nullptr; }

int rt_group_upload_scene(rt_group* g, rt_scene* s) {
    if (!g || !s) return gfail(RT_ERR_INVALID, "null handle");
    for (rt_context* c : g->ctx) GROUP_TRY_RT(rt_upload_scene(c, s));  // the scene is replicated: it is read-only and small
    return RT_OK;
}

// On failure the group is left WITHOUT a target: every context is detached from its (freed) payload, so a
// later render, clear or read fails with RT_ERR_STATE instead of touching freed device memory.
static int dropTargets(rt_group* g, int rc) {
    const std::string keep = g_groupErr;
    for (rt_context* c : g->ctx) (void)rt_release_target(c);
    freeTargets(g);
    g->width = g->height = 0;
    g_groupErr = keep;
    return rc;
}

int rt_group_set_target(rt_group* g, int width, int height) {
    if (!g) return gfail(RT_ERR_INVALID, "null group");
    const int n = (int)g->ctx.size();
    // everything rt_set_target_external() would refuse is refused here, before the old target is given up
    if (width < 2 || height < 2) return gfail(RT_ERR_INVALID, "image must be at least 2x2 (the pixel map divides by W-1, H-1)");
    if ((long long)width * height > (1ll << 30)) return gfail(RT_ERR_LIMIT, "image beyond 2^30 pixels");
    const size_t bytes = rt_accum_bytes(width, height, n);
    if (bytes == 0) return gfail(RT_ERR_INVALID, "bad image size");
    for (rt_context* c : g->ctx) GROUP_TRY_RT(rt_sync(c));
    for (rt_context* c : g->ctx) (void)rt_release_target(c);  // the contexts let go of the payloads before they are freed
    freeTargets(g);
    g->width = g->height = 0;
    for (int i = 0; i < n; i++) {
        hipError_t e = hipSetDevice(g->devices[(size_t)i]);
        if (e == hipSuccess) e = hipMalloc(&g->payload[(size_t)i], bytes);
        if (e == hipSuccess) e = hipMalloc(&g->gathered[(size_t)i], bytes * (size_t)n);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return dropTargets(g, gfail(RT_ERR_HIP, std::string("target allocation: ") + hipGetErrorString(e)));
        }
        int rc = rt_set_target_external(g->ctx[(size_t)i], width, height, i, n, g->payload[(size_t)i]);
        if (rc == RT_OK) rc = rt_clear(g->ctx[(size_t)i]);
        if (rc < 0) return dropTargets(g, gfail(rc, rt_last_error()));
    }
    g->width = width;
    g->height = height;
    return RT_OK;
}

int rt_group_clear(rt_group* g) {
    if (!g) return gfail(RT_ERR_INVALID, "null group");
    if (g->width == 0) return gfail(RT_ERR_STATE, "rt_group_set_target() has not been called");
    for (rt_context* c : g->ctx) GROUP_TRY_RT(rt_clear(c));
    return RT_OK;
}

int rt_group_render(rt_group* g, int first_sample, int n_samples, int max_depth) {
    if (!g) return gfail(RT_ERR_INVALID, "null group");
    if (g->width == 0) return gfail(RT_ERR_STATE, "rt_group_set_target() has not been called");
    // asynchronous on every device's stream: the devices render their tiles side by side
    for (rt_context* c : g->ctx) GROUP_TRY_RT(rt_render(c, first_sample, n_samples, max_depth));
    return RT_OK;
}

// All-gather of the payloads (one RCCL group call over the n communicators, each on its context's
// stream, so it is ordered after that device's render kernels) and the resolve on device 0.
static int gatherAndResolve(rt_group* g, int mode, int spp, void* dst_host, size_t bytes_out) {
    if (!g || !dst_host) return gfail(RT_ERR_INVALID, "bad arguments");
    if (g->width == 0) return gfail(RT_ERR_STATE, "rt_group_set_target() has not been called");
    const int n = (int)g->ctx.size();
    const size_t bytes = rt_accum_bytes(g->width, g->height, n);
    GROUP_TRY_NCCL(rccl().GroupStart());
    for (int i = 0; i < n; i++) {
        int r = rccl().AllGather(g->payload[(size_t)i], g->gathered[(size_t)i], bytes / sizeof(float), kNcclFloat,
                                 g->comm[(size_t)i], (hipStream_t)rt_stream(g->ctx[(size_t)i]));
        if (r != 0) {
            (void)rccl().GroupEnd();
            return gfail(RT_ERR_HIP, std::string("ncclAllGather: ") + (rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error"));
        }
    }
    GROUP_TRY_NCCL(rccl().GroupEnd());
    GROUP_TRY_HIP(hipSetDevice(g->devices[0]));
    if (bytes_out > g->frame_cap) {
        GROUP_TRY_RT(rt_sync(g->ctx[0]));
        if (g->frame) (void)hipFree(g->frame);
        g->frame = nullptr;
        g->frame_cap = 0;
        GROUP_TRY_HIP(hipMalloc(&g->frame, bytes_out));
        g->frame_cap = bytes_out;
    }
    if (mode == 0)
        GROUP_TRY_RT(rt_resolve_gathered(g->ctx[0], g->gathered[0], n, g->frame));
    else
        GROUP_TRY_RT(rt_resolve_gathered_rgba8(g->ctx[0], g->gathered[0], n, spp, g->frame));
    hipStream_t s0 = (hipStream_t)rt_stream(g->ctx[0]);
    GROUP_TRY_HIP(hipMemcpyAsync(dst_host, g->frame, bytes_out, hipMemcpyDeviceToHost, s0));
    for (rt_context* c : g->ctx) GROUP_TRY_RT(rt_sync(c));
    return RT_OK;
}

int rt_group_read_rgba8(rt_group* g, int samples_per_pixel, uint8_t* dst) {
    if (samples_per_pixel < 1) return gfail(RT_ERR_INVALID, "samples_per_pixel must be positive");
    return gatherAndResolve(g, 1, samples_per_pixel, dst, g ? (size_t)g->width * g->height * 4 : 0);
}

int rt_group_read_accum_rgba32f(rt_group* g, float* dst) {
    return gatherAndResolve(g, 0, 1, dst, g ? (size_t)g->width * g->height * 16 : 0);
}

}  // extern "C"
