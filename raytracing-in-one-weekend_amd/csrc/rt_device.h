// Device-side data layout shared by the host upload code and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

// Where each array of the flattened scene lives in HBM.  Everything is read-only during
// a render and small enough for the cover scene to be staged whole into LDS.
struct DevScene {
    // reference-order arrays (the flatten step's output)
    const float4* nodes;      // 2 x float4 per entry: {min.xyz, skip|flags}, {max.xyz, first_ref}
    const uint32_t* prim_refs;
    // search structure traversed first (same entry format, padded boxes) and its guard data
    const float4* fast_nodes;
    const uint32_t* fast_refs;
    const int* sphere_ord;    // sphere id   -> position in prim_refs (the reference's visit order)
    const int* tri_ord;       // triangle id -> position in prim_refs
    const int* prim_guard;    // position in prim_refs -> innermost box-tested entry of `nodes` above it, or -1
    const int* node_outer;    // entry of `nodes` -> next enclosing entry that must be checked too, or -1
    // primitives and materials
    const float4* spheres;    // {c.xyz, r}
    const int* sphere_mat;
    const float4* tri_isect;  // 3 x float4 per triangle: {v0.xyz, -}, {edge1.xyz, -}, {edge2.xyz, -}
    const float4* tri_shade;  // 4 x float4: {n0.xyz, uv0.x}, {n1.xyz, uv0.y}, {n2.xyz, uv1.x}, {uv1.y, uv2.x, uv2.y, -}
    const int* tri_mat;
    const float4* tri_frame;  // normal mapping only: 2 x float4 per triangle {tangent.xyz, 1 if usable}, {bitangent.xyz, -}
    const float4* materials;  // 2 x float4: {r, g, b, param}, {kind, texD, texR, texN} (ints bit-cast)
    const uint32_t* texels;   // all textures, RGBX8, one dword per texel
    const int4* tex_desc;     // {first texel, width, height, 0}
    const float* lut;         // 256-entry 8-bit -> float map
    int n_nodes, n_prim_refs, n_fast_nodes, n_fast_refs, n_spheres, n_tris, n_mats;
    // Instancing (opt-in): entries [0, n_world_nodes) of `nodes` are the world's, the prototypes' entries follow;
    // an instance is a world entry with bit 29 of its skip word set and the instance's number in its first_ref
    // word.  Five float4 per instance: the inverse model matrix column by column, then {first entry, end entry
    // of its prototype (ints bit-cast), -, -}.  Scenes with instances are rendered along the reference-order entries
    // only (renderWavefront<INST>; the plain kernel and the trace hook walk the two levels too).
    const float4* instances;
    int n_instances, n_world_nodes;
    int fast_oct_stride;      // > 0: fast_nodes holds 8 direction-ordered copies of this many entries each
    int has_pbr;              // 1: some material is a PBRMaterial
    int has_dielectric;       // 1: some material is a Dielectric
    int normal_mapping;       // 1: PBR materials with a normal map shade with the mapped normal (off: as the reference, the map is ignored)
    int force_ref;            // 1: always walk `nodes` in the reference's order (diagnostics)
    // per-ray widening of the search structure's boxes for rays that start far from small primitives
    // (the reference's own hit tests lose accuracy with that distance and report hits the fixed padding
    // does not cover; FlatScene::far_* on the host, farRayPad() in the kernels)
    int far_spheres, far_tris;
    float far_center[3], far_radius, far_rmin, far_typical;
    float far_trigger2;       // squared distance from far_center beyond which a ray origin gets a widening (+inf: never)
    float origin_limit;       // rays starting further than this from the coordinate origin (any axis) walk `nodes`
    float cam[22];            // origin, lowerLeftCorner, horizontal, vertical, u, v, w, lensRadius
    float bg[3];
};

// One path's radiance in the colour buffer: 12 bytes written once and read once per pixel sample.
struct Radiance {
    float r, g, b;
};

struct RenderArgs {
    DevScene sc;
    float4* accum;        // tiled: [local tile][lane] -> {r, g, b, samples}
    // Work queue of the persistent launch shape (may be null): kQueueBytes, cleared before every launch.
    // [1] is set when the launch was cut short; the counters that deal out the work are n_queues dwords at
    // [kQueueFirst + kQueueStride * q], each on a cache line of its own.
    unsigned int* queue;
    int n_queues;
    const int* cancel;    // host-pinned flag the persistent waves read when they fetch a chunk (null: not cancellable)
    unsigned long long* counters;  // path statistics (counting launches only)
    int width, height;
    int tiles_x, tiles_y;      // 8x8-pixel tiles
    int tiles_per_row_local;   // owned tiles per tile row = ceil(tiles_x / world)
    int n_local_tiles;
    int rank, world;
    int first_sample, n_samples, max_depth;
    int lds_bytes;
    int knob[4];  // scheduling thresholds of the wavefront loop (never affect results)
    int timing;  // counting launches collect per-operation clock ticks instead (diagnostic)
    // per-sample radiance of the launch: colors[(sample - first_sample) * n_slots + pixel slot],
    // added to the accumulation buffer in sample order by accumulateColors()
    Radiance* colors;
    // Two-pass launches (primaryPass + the persistent loop): one record per path that goes on after its first ray,
    // three planes of records_cap float4 each:
    //   {origin.xyz, direction.x}, {direction.yz, throughput.xy}, {throughput.z, rng key, rng counter, unit | flag << 31}
    // flag: the record holds the camera ray itself, to be traced again along the reference-order array.
    // The records of chunk c (64 units = one tile x one sample) sit at the front of slots 64 c .. 64 c + 63, and
    // record_counts[c] (one byte per chunk) says how many there are; the loop takes chunks four at a time, so
    // that one dword of counts comes with every queue index.
    float4* records;
    unsigned int records_cap;
    uint8_t* record_counts;
    int chunk_group;  // single-pass launches: chunks handed out per queue access
};

// Megabytes of the fixed colour scratch behind rt_render_noalloc() (one copy per GPU in use).
#ifndef RT_NOALLOC_SCRATCH_MB
#define RT_NOALLOC_SCRATCH_MB 64
#endif

// layout of RenderArgs::queue, in dwords
constexpr int kQueueFirst = 64, kQueueStride = 32, kMaxQueues = 30;
constexpr size_t kQueueBytes = 4096;

constexpr int kTile = 8;  // one wave64 = one 8x8 pixel tile

// workgroups of the persistent loop: at most WF_MAX_THREADS (the kernel's launch bounds), and at most WF_GLOBAL_THREADS when the scene's tables
// are read from global memory (per-lane LDS arrays of those builds are sized for that)
#define WF_GLOBAL_THREADS 256
#ifndef WF_MAX_THREADS
#define WF_MAX_THREADS 512
#endif

// Bytes of LDS the staged scene tables need, each block 16-byte aligned (order = stageScene()).
// with_reference_arrays = false leaves the reference-order entries, their leaf references and the
// guard tables in global memory (they are touched once or twice per accepted hit and by the rare
// re-trace), which lets one more workgroup share a CU's LDS.
// The search structure is budgeted at 40 bytes per entry plus one closing entry, which is how the
// persistent kernel lays it out (the plain kernel stages 32 bytes per entry into the same space).
inline size_t ldsSceneBytes(const DevScene& s, bool with_reference_arrays = true) {
    auto al = [](size_t x) { return (x + 15) & ~size_t(15); };
    size_t n = al(((size_t)s.n_fast_nodes + 1) * 40) + al((size_t)s.n_fast_refs * 4) + al((size_t)s.n_spheres * 16) +
               al((size_t)s.n_tris * 48) + al((size_t)s.n_spheres * 4) + al((size_t)s.n_tris * 4) +
               al((size_t)s.n_mats * 32) + al((size_t)s.n_spheres * 4) + al((size_t)s.n_tris * 4);
    if (with_reference_arrays)
        n += al((size_t)s.n_nodes * 32) + al((size_t)s.n_prim_refs * 4) + al((size_t)s.n_prim_refs * 4) +
             al((size_t)s.n_nodes * 4);
    return n;
}

// The LDS walk entries hold the skip link as a 16-bit LDS address (the entries come first in the kernel's
// dynamic LDS, after at most 4 KiB of static variables) and the leaf run's start as a 16-bit
// index (0xFFFE and 0xFFFF are reserved): scenes beyond either keep their tables in global memory.
inline bool ldsWalkEntriesFit(const DevScene& s) {
    return ((size_t)s.n_fast_nodes + 1) * 40 + 4096 <= 0xFFFF && s.n_fast_refs < 0xFFFE;
}

}  // namespace rt
