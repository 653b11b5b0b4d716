// Host-side scene construction and the BVH flatten step (no device code here).
// R/ = /root/reference/Raytracing-In-A-Weekend/ in the citations below.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <random>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

namespace rt {

struct Box3 {
    float mn[3], mx[3];
};

// One entry of HittableList::objects (R/src/Core/Hittable.h:315): a primitive or a composite.
struct Obj {
    enum Kind : int { SPHERE = 0, TRIANGLE = 1, COMPOSITE = 2, INSTANCE = 3 } kind;
    int id;  // sphere id, triangle id, index into Scene::composites, or index into Scene::instances
};

// A node of a built BVH (R/src/AccelerationStructures/Bvh.h:13-100).
struct BvhNode {
    Box3 box;
    int left = -1, right = -1;  // child node indices when both children are BVH nodes
    std::vector<Obj> leaf;      // otherwise: the objects under this node, in the order Hit() visits them
};

struct Composite {
    enum Kind : int { BVH = 0, MESH = 1, LIST = 2 } kind;
    Box3 box;                // BVH root box, Mesh::boundingBox, or an explicit list box (Box class)
    bool has_box_test;       // BVH and MESH test `box` before descending; a plain list does not
    std::vector<Obj> objs;   // MESH / LIST children in order
    std::vector<BvhNode> nodes;  // BVH only
    int root = -1;
    // nodes[built_first, built_first + built_count) were made by one call of Scene::buildBvh: numbered in pre-order,
    // every leaf non-empty.  0 for trees that arrive node by node through the API.
    size_t built_first = 0, built_count = 0;
};

// Instancing (opt-in; the reference has none: it bakes every placement into world-space triangles,
// R/src/Core/Mesh.cpp:106-208).  A prototype is a list of objects in object space that is not in the world
// itself; an instance places it with a model matrix.  Hit semantics are those of the CPU oracle's Instance
// class: the placement's world-space box, the ray taken into object space (direction not normalised, so hit
// distances are the same numbers in both spaces), the prototype's own Hit, and the normal brought back
// through mat3(transpose(inverse(M))) and normalised.
struct Prototype {
    std::vector<Obj> objs;
};
struct Instance {
    int prototype;
    float model[16], inv[16];  // column-major
    Box3 box;                  // of the prototype's box corners under `model`
};

struct HostMaterial {
    int kind;
    float r, g, b, param;
    int tex_d, tex_r, tex_n;
};
struct HostTexture {
    int w, h;
    std::vector<uint8_t> rgb;
};
struct HostSphere {
    float c[3], r;
    int mat;
};
struct HostTriangle {
    float pos[9], nrm[9], uv[6];
    int mat;
};

struct FlatScene {
    // 8 words per entry: {min.xyz, skip|flags, max.xyz, first_ref}
    std::vector<uint32_t> nodes;
    std::vector<uint32_t> prim_refs;
    int max_leaf = 0;
    bool valid = false;
    bool fast_stale = false;  // the search structure was ordered for another eye position

    // The search structure the kernels traverse first (fast_bvh.cpp): a SAH BVH over all
    // primitives with padded boxes, in the same threaded entry format.  Its answer is accepted
    // only if the guard boxes of the hit primitive pass; otherwise the ray is traced again
    // through `nodes` in the reference's order.
    std::vector<uint32_t> fast_nodes;   // 8 words per entry: {min.x, max.x, min.y, max.y, min.z, max.z, skip, first_ref}
    std::vector<uint32_t> fast_refs;    // leaf references, as `prim_refs`
    // For scenes too large for LDS: the same entries eight times over, each copy with the children of
    // every entry ordered near-to-far for one sign pattern of the ray direction (copy k: bit a of k
    // set = direction component a negative).  A copy has fast_oct_stride entries, the last one a box
    // no ray passes; links are indices into the whole array, "walk over" = 8 * fast_oct_stride.
    // The leaf runs are shared (fast_refs).  Empty for scenes that fit LDS.
    std::vector<uint32_t> fast_oct;
    int fast_oct_stride = 0;
    // Smallest padding any primitive box of the search structure received.  The fused box test's
    // rounding error grows with the ray origin's coordinates (it rounds origin * (1/d) before the
    // subtraction), so rays that start further than fast_min_pad * 2^21 from the coordinate origin
    // are not vouched for by the padding and walk the reference-order array instead.
    float fast_min_pad = 0.0f;
    // The primitive tests themselves lose accuracy with the distance D between ray origin and primitive:
    // Sphere::Hit forms halfB^2 - a c from terms of size D^2, so its discriminant is off by up to about
    // 32 * 2^-24 * D^2 and it reports "hits" for rays that pass a sphere of radius r at up to
    // sqrt(r^2 + 32 * 2^-24 * D^2) from its centre; Triangle::Hit's barycentrics are off by about
    // 64 * 2^-24 * D in position terms.  The reference returns those hits, so the search structure must not
    // cull them.  Every box is padded for the distance far_typical within which the scene's rays normally
    // start (from the eye, or from one primitive to another); a ray that starts further from the
    // primitives concerned (far_spheres / far_tris of them: those the rounding part of the padding does not
    // cover at any distance anyway; far_center / far_radius bound them, far_rmin is the smallest radius)
    // widens every box by the difference (farRayPad() in the kernels).
    int far_spheres = 0, far_tris = 0;
    float far_center[3] = {0, 0, 0}, far_radius = 0.0f, far_rmin = 0.0f, far_typical = 0.0f;
    // Inputs of a device-side build of the search structure (csrc/device_bvh.hip): the primitives that
    // are in the world (leaf references without the last-of-run bit), the floor of the box padding,
    // and the bounds of the padded boxes' centres.  With defer_search_build set, flatten() prepares
    // these (and the guard tables) and leaves fast_nodes / fast_refs empty.
    bool defer_search_build = false;
    std::vector<uint32_t> world_refs;
    float pad_floor = 0.0f;
    float centroid_min[3] = {0, 0, 0}, centroid_max[3] = {0, 0, 0};
    std::vector<int32_t> sphere_ord;    // sphere id   -> position in prim_refs (reference visit order)
    std::vector<int32_t> tri_ord;       // triangle id -> position in prim_refs
    std::vector<int32_t> prim_guard;    // position in prim_refs -> innermost box-tested entry of `nodes` above it, or -1
    std::vector<int32_t> node_outer;    // entry of `nodes` -> nearest enclosing box-tested entry whose box does not contain it, or -1
    // Scenes with instances: entries [0, n_world_nodes) are the world's; the prototypes' entries follow, prototype p
    // at [proto_range[2 p], proto_range[2 p + 1]), and an instance is a world entry with kNodeInstance set whose
    // first_ref word is the instance's number.  Such scenes have no search structure: every ray takes the
    // reference-order walk.
    int n_world_nodes = 0;
    std::vector<int32_t> proto_range;
};

constexpr uint32_t kNodeNoBox = 1u << 30;     // entry without a box test
constexpr uint32_t kNodeInstance = 1u << 29;  // entry that stands for an instance
constexpr uint32_t kNodeSkipMask = kNodeInstance - 1;
constexpr uint32_t kRefLast = 1u << 31;       // last reference of its leaf run
constexpr uint32_t kRefTriangle = 1u << 30;
constexpr uint32_t kRefIdMask = kRefTriangle - 1;

struct Camera22 {
    float origin[3], llc[3], horizontal[3], vertical[3], u[3], v[3], w[3], lens_radius;
};

unsigned buildThreads();  // threads the host-side builders may use

// Starts fn on a new thread; false, and nothing started, if the system will not give one (the callers then do the
// work themselves: a limit on threads slows the builders down, it does not stop them)
template <class F>
bool tryThread(std::thread& t, F fn) {
    try {
        t = std::thread(fn);
        return true;
    } catch (const std::system_error&) {
        return false;
    }
}

// fn(first, last) over [0, n) in contiguous pieces on the builders' threads (small ranges: the caller's thread)
template <class F>
void parallelPieces(size_t n, F fn) {
    const unsigned threads = buildThreads();
    if (n < (1u << 16) || threads < 2) {
        fn((size_t)0, n);
        return;
    }
    const size_t piece = (n + threads - 1) / threads;
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; t++) {
        const size_t a = std::min(n, t * piece), b = std::min(n, a + piece);
        if (a >= b) continue;
        std::thread th;
        if (tryThread(th, [=] { fn(a, b); }))
            pool.push_back(std::move(th));
        else
            fn(a, b);
    }
    fn((size_t)0, std::min(n, piece));
    for (auto& th : pool) th.join();
}

// std::sort's permutation, on several threads.  Which of several objects with equal keys ends up first is part of the
// tree the reference builds (Bvh.h:46 sorts with std::sort, which is not stable), so a faster sort is only usable
// if it leaves every element exactly where std::sort would.  libstdc++'s std::sort is an introsort: while a range
// has more than 16 elements, the median of its second, middle and last element is swapped to the front and the
// rest is partitioned around it from both ends; the right part is handled first (recursively), then the left;
// after 2 log2(n) levels of that a range is heap-sorted instead; one insertion sort over everything finishes.  The
// two parts of a partition never exchange elements again -- no element of the right part is smaller than one of
// the left part, and the final insertion sort only moves an element past strictly larger ones -- so (a) the parts
// can be sorted by different threads and (b) the final pass is the same as a stable insertion sort of each range
// of at most 16 elements the partitioning ends with.  The steps below are those of libstdc++ (GCC 4.7 and later;
// the oracle sorts with the real std::sort, and tests/test_host_flatten.py compares the trees on inputs full of
// equal keys).
template <class T, class Less>
void sortRangeLikeStdSort(T* first, T* last, int depth_limit, Less less, std::atomic<int>& spare_threads) {
    std::vector<std::thread> helpers;
    while (last - first > 16) {
        if (depth_limit == 0) {
            std::partial_sort(first, last, last, less);  // the heap sort std::sort falls back to
            break;
        }
        depth_limit--;
        // median of (first + 1, middle, last - 1) to the front
        T *a = first + 1, *b = first + (last - first) / 2, *c = last - 1;
        if (less(*a, *b)) {
            if (less(*b, *c)) std::swap(*first, *b);
            else if (less(*a, *c)) std::swap(*first, *c);
            else std::swap(*first, *a);
        } else if (less(*a, *c)) std::swap(*first, *a);
        else if (less(*b, *c)) std::swap(*first, *c);
        else std::swap(*first, *b);
        // partition of [first + 1, last) around *first, from both ends, without bounds checks (the median's
        // two companions stop the scans)
        T *lo = first + 1, *hi = last;
        for (;;) {
            while (less(*lo, *first)) ++lo;
            --hi;
            while (less(*first, *hi)) --hi;
            if (!(lo < hi)) break;
            std::swap(*lo, *hi);
            ++lo;
        }
        T* cut = lo;
        std::thread helper;
        const bool wanted = last - cut >= 32768;
        if (wanted && spare_threads.fetch_sub(1) > 0 &&
            tryThread(helper, [=, &spare_threads] { sortRangeLikeStdSort(cut, last, depth_limit, less, spare_threads); })) {
            helpers.push_back(std::move(helper));
        } else {
            if (wanted) spare_threads.fetch_add(1);  // nothing was taken
            sortRangeLikeStdSort(cut, last, depth_limit, less, spare_threads);
        }
        last = cut;
    }
    if (last - first <= 16) {
        for (T* i = first + 1; i < last; ++i) {
            T val = *i;
            T* j = i;
            while (j > first && less(val, *(j - 1))) {
                *j = *(j - 1);
                --j;
            }
            *j = val;
        }
    }
    for (std::thread& h : helpers) {
        h.join();
        spare_threads.fetch_add(1);
    }
}
template <class T, class Less>
void sortLikeStdSort(T* first, T* last, Less less, std::atomic<int>& spare_threads) {
    if (first == last) return;
    int lg = 0;
    for (size_t n = (size_t)(last - first); n > 1; n >>= 1) lg++;
    sortRangeLikeStdSort(first, last, 2 * lg, less, spare_threads);
}

// Tangent and bitangent of a triangle from its positions (9 floats) and texture coordinates (6 floats):
//   T = f (dv2 e1 - dv1 e2),  B = f (-du2 e1 + du1 e2),  f = 1 / (du1 dv2 - du2 dv1),
// e_k = p_k - p_0, (du_k, dv_k) = uv_k - uv_0.  False (and zeros) when the triangle has no texture
// coordinates (uv0.x == -1, the reference's marker) or its uv triangle is degenerate.
bool tangentFrame(const float* pos9, const float* uv6, float* tangent3, float* bitangent3);

class Scene {
public:
    Scene();
    // --- RNG (R/src/Core/RTWeekend.h:26-58) ---
    float randomFloat();
    float randomFloat(float mn, float mx) { return mn + (mx - mn) * randomFloat(); }
    int randomInt(int mn, int mx) { return static_cast<int>(randomFloat((float)mn, (float)(mx + 1))); }
    void randomVec(float out[3]);
    void randomVec(float mn, float mx, float out[3]);

    int addMaterial(int kind, float r, float g, float b, float param);
    int addTexture(int w, int h, const uint8_t* rgb);
    int addPbr(int td, int tr, int tn);
    int addSphere(float cx, float cy, float cz, float r, int mat);
    int addTriangle(const float* pos9, const float* nrm9, const float* uv6, int mat);
    int addBox(const float* p0, const float* p1, int mat, const float* model16);
    int addMesh(const float* model16, int n_vertices, const float* positions, const float* normals, const float* uvs,
                int n_triangles, const uint32_t* indices, int mat, int max_depth, const int* tri_materials = nullptr);
    int addMeshGltf(const float* model16, const char* path, int material, int max_depth, bool per_primitive,
                    bool first_primitive_only, std::string* err);
    int textureFromFile(const std::string& path, std::string* err);
    int textureFromBytes(const std::string& bytes, const std::string& label, std::string* err);
    // Ingestion of a tree the caller has already built (no random numbers are drawn): objects added
    // between beginGroup() and endGroup() become the children, in order, of one composite.
    // boxed = true: the caller's box is tested first, then the children are scanned with the running
    // closest hit -- BVHNode::Hit (left, then right; R/src/AccelerationStructures/Bvh.h:80-88) and
    // Mesh::Hit (R/src/Core/Mesh.h:13-31) alike.  boxed = false: a plain HittableList.
    int beginGroup(bool boxed, const float* box6);
    int endGroup();
    // Instancing: objects added between beginPrototype() and endPrototype() form a prototype (returned: its
    // number) instead of going into the world; addInstance() places one.  -1: misuse (nested definition, an
    // instance of an unknown or unfinished prototype, an empty prototype).
    int beginPrototype();
    int endPrototype();
    int addInstance(int prototype, const float* model16);
    bool groupsOpen() const { return !open_groups_.empty(); }
    void randomScene();
    void cornellBox();
    void wrapBvh();
    void setCamera(const float* lookfrom, const float* lookat, const float* vup, float vfov, float aspect, float aperture,
                   float focus);
    void flatten();
    void buildFastStructure();  // fast_bvh.cpp; called by flatten()

    Box3 objectBox(const Obj& o) const;

    std::vector<HostMaterial> materials;
    std::vector<HostTexture> textures;
    std::vector<HostSphere> spheres;
    std::vector<HostTriangle> triangles;
    std::vector<Composite> composites;
    std::vector<Prototype> prototypes;
    std::vector<Instance> instances;
    std::vector<Obj> world;  // Scene::world.objects
    Camera22 camera{};
    float background[3] = {0, 0, 0};
    bool normal_mapping = false;  // opt-in: PBR normal maps take effect (the reference computes the mapped normal and drops it)
    FlatScene flat;

private:
    std::vector<int> open_groups_;  // composites under construction, innermost last
    int open_prototype_ = -1;       // prototype under construction
    void place(const Obj& o);       // appends to the innermost open group, or to the world
    int newTriangle(const float* pos9, const float* nrm9, const float* uv6, int mat);
    struct BoxedObj {
        Obj obj;
        Box3 box;
    };
    std::vector<BoxedObj> withBoxes(const std::vector<Obj>& objs) const;
    int buildBvh(Composite& c, std::vector<BoxedObj>& objs, size_t start, size_t end, int max_depth);
    void buildBvhAt(Composite& c, std::vector<BoxedObj>& objs, size_t start, size_t end, int max_depth, size_t me,
                    const std::vector<uint8_t>& axes, const std::atomic<size_t>& drawn, std::atomic<int>& spare_threads);
    void emitObjects(const std::vector<Obj>& objs, int parent_entry);
    void emitComposite(const Composite& c);
    void emitBvhNode(const Composite& c, int node);
    bool emitBvhFlat(const Composite& c);
    int newEntry(const Box3* box);
    std::mt19937 gen_;
};

}  // namespace rt
