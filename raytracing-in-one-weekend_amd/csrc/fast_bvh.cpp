// The search structure the kernels traverse first, and the guard data that lets its answer
// stand in for the reference's own traversal.
//
// Why this is allowed.  The reference finds the closest hit by walking ITS tree in ITS order
// (R/src/AccelerationStructures/Bvh.h:80-88).  What that walk returns can be characterised
// without doing it:
//   * A primitive's hit distance does not depend on the tMax it is tested with: Sphere::Hit /
//     Triangle::Hit return the same t for every tMax >= t and miss for tMax < t
//     (R/src/Core/Hittable.h:241-246, 80-82).
//   * AABB::Hit(r, tMin, tMax) (AABB.h:10-24) passes exactly when T1 > Tm and tMax > Tm, where
//     Tm = max(tMin, entry distances) and T1 = min(exit distances) depend on ray and box only.
//   * surroundingBox() nests boxes exactly (fmin/fmax), and the float expressions for the entry
//     and exit distances are monotone in the box planes, so if a node's box passes for some
//     tMax, every enclosing box of the same BVH passes for any tMax' >= tMax.
// So if P is the primitive with the smallest hit distance t_P of ALL primitives (the later one
// in the reference's visit order among exact ties), and the innermost reference box above P
// (plus any enclosing box that does not contain it, e.g. a mesh AABB) passes with tMax = t_P,
// then the reference's walk reaches P, accepts it (tMax >= t_P always holds there) and nothing
// replaces it: the reference returns (P, t_P).  The kernels therefore search a tighter tree
// (surface-area heuristic, boxes padded far beyond rounding error so that no true hit is ever
// culled), check that condition for the winner, and re-trace the ray through the reference
// arrays in the reference's order whenever the check fails, a distance is NaN, or a direction
// component is zero (infinite reciprocal, where the NaN-skipping of the slab test breaks the
// monotonicity argument).
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <limits>
#include <mutex>
#include <thread>

#include "host_scene.h"

namespace rt {

namespace {

struct PrimBox {
    float mn[3], mx[3];
    uint32_t ref;  // kind flag + id, as in prim_refs (without the last-of-run bit)
};

inline float f32(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
inline uint32_t u32(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

inline float area(const float* mn, const float* mx) {
    float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}

// Two phases: a binary tree by the binned surface-area heuristic, then its emission in walk order
// (entry, then its children; `skip` = the entry after the subtree).  The walk needs no binary shape,
// so the emission may leave an inner entry out and hand its children to the entry above.
struct Builder {
    std::vector<PrimBox>& prims;
    std::vector<uint32_t>& nodes;
    std::vector<uint32_t>& refs;
    static constexpr int kBins = 64;
    int kMaxLeaf = 2;
    // An inner entry is left out when its box has more than this share of the area of the entry above:
    // its test would pass for most rays that get there, and costs more than it saves (0 keeps all).
    // Cover scene 29.2 -> 28.6 ms, Cornell + vase 336 -> 301 ms, 1 M triangles 73 -> 69 ms at 0.5.
    float collapse = 0.5f;
    // Which child of a pair is visited first.  A hit in hand lets the walk cull what comes after, so
    // the likelier holder of the closest hit goes first: for camera rays that is the box nearer to the
    // eye, for the rest the larger box.  Cover scene: as partitioned 30.7 ms, smaller first 29.8, larger
    // first 29.2 (28.6 with `collapse`), nearer to the eye first (larger on a tie) 27.9.  A million
    // triangles at 6.9 rays per sample lose 2 % with the eye rule, Cornell + vase is indifferent.
    int order = 4;  // 0 as partitioned, 1 smaller first, 2 larger first, 3 all children of an entry by descending area, 4 nearer to the eye first
    float eye[3] = {0, 0, 0};

    struct TNode {
        float mn[3], mx[3];
        float area;
        int left, right;  // -1: leaf over prims[start, end)
        size_t start, end;
    };
    // the tree is sized up front and its slots are handed out by an atomic counter, so that subtrees
    // (disjoint ranges of `prims`) can be built by different threads; the emission below depends on
    // the tree's shape only, never on which slot a node got
    std::vector<TNode> tree;
    std::atomic<int> nextSlot{0};
    std::atomic<int> spareThreads{0};
    int root = -1;
    std::vector<uint32_t> firstRef;  // per tree node: start of its leaf run in `refs`

    int newEntry(const float* mn, const float* mx) {
        int idx = (int)(nodes.size() / 8);
        // search-structure entry: {min.x, max.x, min.y, max.y, min.z, max.z, skip, first_ref}
        uint32_t w[8] = {u32(mn[0]), u32(mx[0]), u32(mn[1]), u32(mx[1]),
                         u32(mn[2]), u32(mx[2]), (uint32_t)(idx + 1), 0xFFFFFFFFu};
        nodes.insert(nodes.end(), w, w + 8);
        return idx;
    }

    int build(size_t start, size_t end) {
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t i = start; i < end; i++) {
            for (int k = 0; k < 3; k++) {
                mn[k] = std::min(mn[k], prims[i].mn[k]);
                mx[k] = std::max(mx[k], prims[i].mx[k]);
                float c = 0.5f * (prims[i].mn[k] + prims[i].mx[k]);
                cmn[k] = std::min(cmn[k], c);
                cmx[k] = std::max(cmx[k], c);
            }
        }
        const int t = nextSlot.fetch_add(1);
        tree[t] = TNode{{mn[0], mn[1], mn[2]}, {mx[0], mx[1], mx[2]}, area(mn, mx), -1, -1, start, end};
        const size_t n = end - start;
        if (n <= (size_t)kMaxLeaf) return t;
        size_t mid = 0;
        // binned surface-area heuristic over the centroid bounds
        float bestCost = INFINITY;
        int bestAxis = -1, bestBin = -1;
        for (int ax = 0; ax < 3; ax++) {
            const float ext = cmx[ax] - cmn[ax];
            if (!(ext > 0)) continue;
            float bmn[kBins][3], bmx[kBins][3];
            int cnt[kBins] = {0};
            for (int b = 0; b < kBins; b++)
                for (int k = 0; k < 3; k++) {
                    bmn[b][k] = INFINITY;
                    bmx[b][k] = -INFINITY;
                }
            const float scale = kBins / ext;
            for (size_t i = start; i < end; i++) {
                float c = 0.5f * (prims[i].mn[ax] + prims[i].mx[ax]);
                int b = std::min(kBins - 1, std::max(0, (int)((c - cmn[ax]) * scale)));
                cnt[b]++;
                for (int k = 0; k < 3; k++) {
                    bmn[b][k] = std::min(bmn[b][k], prims[i].mn[k]);
                    bmx[b][k] = std::max(bmx[b][k], prims[i].mx[k]);
                }
            }
            float rArea[kBins];
            int rCnt[kBins];
            float amn[3] = {INFINITY, INFINITY, INFINITY}, amx[3] = {-INFINITY, -INFINITY, -INFINITY};
            int c = 0;
            for (int b = kBins - 1; b > 0; b--) {
                for (int k = 0; k < 3; k++) {
                    amn[k] = std::min(amn[k], bmn[b][k]);
                    amx[k] = std::max(amx[k], bmx[b][k]);
                }
                c += cnt[b];
                rArea[b] = c ? area(amn, amx) : 0.0f;
                rCnt[b] = c;
            }
            for (int k = 0; k < 3; k++) {
                amn[k] = INFINITY;
                amx[k] = -INFINITY;
            }
            c = 0;
            for (int b = 0; b < kBins - 1; b++) {
                for (int k = 0; k < 3; k++) {
                    amn[k] = std::min(amn[k], bmn[b][k]);
                    amx[k] = std::max(amx[k], bmx[b][k]);
                }
                c += cnt[b];
                if (c == 0 || rCnt[b + 1] == 0) continue;
                float cost = area(amn, amx) * c + rArea[b + 1] * rCnt[b + 1];
                if (cost < bestCost) {
                    bestCost = cost;
                    bestAxis = ax;
                    bestBin = b;
                }
            }
        }
        if (bestAxis < 0) {
            // all centroids coincide: split by count
            mid = start + n / 2;
        } else {
            const float ext = cmx[bestAxis] - cmn[bestAxis];
            const float scale = kBins / ext;
            auto it = std::partition(prims.begin() + start, prims.begin() + end, [&](const PrimBox& p) {
                float c = 0.5f * (p.mn[bestAxis] + p.mx[bestAxis]);
                int b = std::min(kBins - 1, std::max(0, (int)((c - cmn[bestAxis]) * scale)));
                return b <= bestBin;
            });
            mid = (size_t)(it - prims.begin());
            if (mid == start || mid == end) mid = start + n / 2;
        }
        int l, r;
        std::thread other;
        if (n >= 32768 && spareThreads.fetch_sub(1) > 0 && tryThread(other, [&] { l = build(start, mid); })) {
            r = build(mid, end);
            other.join();
            spareThreads.fetch_add(1);
        } else {
            if (n >= 32768) spareThreads.fetch_add(1);  // nothing was taken
            l = build(start, mid);
            r = build(mid, end);
        }
        tree[t].left = l;
        tree[t].right = r;
        return t;
    }

    void run();
};

// One emission of the tree in walk order (entry, then its children; `skip` = the entry after the
// subtree).  The walk needs no binary shape, so an inner entry may be left out and its children
// handed to the entry above.  The first emission writes the leaf runs and notes where each leaf's
// run starts; further emissions (other child orders over the same tree) point at the same runs.
struct Emission {
    const Builder& b;
    std::vector<uint32_t>& nodes;
    std::vector<uint32_t>* refs;          // where leaf runs are written; null: reuse `firstRef`
    std::vector<uint32_t>& firstRef;      // per tree node: start of its leaf run
    int order;                            // as Builder::order; 5: nearer first along `dir`
    float dir[3] = {0, 0, 0};
    uint32_t endLink = 0;                 // with refs == null: the index that means "walk over"
    uint32_t base = 0;                    // index of this emission's first entry in the final array

    int newEntry(const float* mn, const float* mx) {
        int idx = (int)(nodes.size() / 8);
        // search-structure entry: {min.x, max.x, min.y, max.y, min.z, max.z, skip, first_ref}
        uint32_t w[8] = {u32(mn[0]), u32(mx[0]), u32(mn[1]), u32(mx[1]),
                         u32(mn[2]), u32(mx[2]), (uint32_t)(idx + 1), 0xFFFFFFFFu};
        nodes.insert(nodes.end(), w, w + 8);
        return idx;
    }

    float eyeDistance(const Builder::TNode& n) const {
        float d2 = 0;
        for (int k = 0; k < 3; k++) {
            const float d = std::max(0.0f, std::max(n.mn[k] - b.eye[k], b.eye[k] - n.mx[k]));
            d2 += d * d;
        }
        return d2;
    }

    // how far along `dir` the box starts
    float along(const Builder::TNode& n) const {
        float s = 0;
        for (int k = 0; k < 3; k++) s += dir[k] * (dir[k] > 0 ? n.mn[k] : n.mx[k]);
        return s;
    }

    // the two children of a tree node in visiting order
    void pairOf(const Builder::TNode& n, int& first, int& second) const {
        first = n.left;
        second = n.right;
        const Builder::TNode &L = b.tree[n.left], &R = b.tree[n.right];
        const float aL = L.area, aR = R.area;
        if (order == 4 || order == 5) {
            const float dL = order == 4 ? eyeDistance(L) : along(L), dR = order == 4 ? eyeDistance(R) : along(R);
            if (dL == dR ? aL < aR : dL > dR) std::swap(first, second);
            return;
        }
        if (order == 1 ? aL > aR : (order >= 2 ? aL < aR : false)) std::swap(first, second);
    }

    // the entries that stand directly below a kept entry of area `above`: a child is either kept
    // or replaced, in place, by its own children
    void gather(int t, float above, std::vector<int>& out) const {
        const Builder::TNode& n = b.tree[t];
        if (n.left >= 0 && b.collapse > 0.0f && n.area > b.collapse * above) {
            int a, c;
            pairOf(n, a, c);
            gather(a, above, out);
            gather(c, above, out);
        } else {
            out.push_back(t);
        }
    }

    void emitList(std::vector<int>& list) {
        if (order == 3)
            std::stable_sort(list.begin(), list.end(), [&](int x, int y) { return b.tree[x].area > b.tree[y].area; });
        for (int t : list) emit(t);
    }

    void emit(int t) {
        const Builder::TNode& n = b.tree[t];
        const int e = newEntry(n.mn, n.mx);
        if (n.left < 0) {
            if (refs) {
                firstRef[(size_t)t] = (uint32_t)refs->size();
                for (size_t i = n.start; i < n.end; i++) refs->push_back(b.prims[i].ref);
                refs->back() |= kRefLast;
            }
            nodes[(size_t)e * 8 + 7] = firstRef[(size_t)t];
        } else {
            std::vector<int> kids;
            int a, c;
            pairOf(n, a, c);
            gather(a, n.area, kids);
            gather(c, n.area, kids);
            emitList(kids);
        }
        nodes[(size_t)e * 8 + 6] = (uint32_t)(nodes.size() / 8);
    }

    void run(int root) {
        std::vector<int> top;
        gather(root, b.tree[root].area, top);  // the root itself goes when it is an inner entry
        emitList(top);
    }
};

void Builder::run() {
    tree.resize(2 * prims.size());
    spareThreads = (int)buildThreads() - 1;
    root = build(0, prims.size());
    firstRef.assign(tree.size(), 0xFFFFFFFFu);
    Emission{*this, nodes, &refs, firstRef, order}.run(root);
}

}  // namespace

void Scene::buildFastStructure() {
    FlatScene& f = flat;
    f.fast_stale = false;
    f.fast_nodes.clear();
    f.fast_refs.clear();
    f.fast_oct.clear();
    f.fast_oct_stride = 0;
    f.fast_min_pad = std::numeric_limits<float>::infinity();
    f.sphere_ord.assign(spheres.size(), -1);
    f.tri_ord.assign(triangles.size(), -1);
    const size_t nRefs = f.prim_refs.size();
    const size_t nNodes = f.nodes.size() / 8;
    // RT_HOST_TIMING=1: the stretches of this function on stderr
    const bool timing = std::getenv("RT_HOST_TIMING") && std::atoi(std::getenv("RT_HOST_TIMING")) != 0;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[search structure] %-32s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t0).count());
        t0 = now;
    };

    // ---- reference visit order of every primitive
    for (size_t i = 0; i < nRefs; i++) {
        const uint32_t r = f.prim_refs[i];
        const uint32_t id = r & kRefIdMask;
        // a primitive that is listed twice (possible through the API) keeps its LAST position,
        // the one that wins an exact tie in the reference's walk
        if (r & kRefTriangle) f.tri_ord[id] = (int32_t)i; else f.sphere_ord[id] = (int32_t)i;
    }

    lap("visit order of the primitives");
    // ---- guards: for every leaf run the innermost box-tested entry above it, and for every
    // box-tested entry the nearest enclosing one that does not contain its box
    f.prim_guard.assign(nRefs, -1);
    f.node_outer.assign(nNodes, -1);
    if (!instances.empty()) {
        // scenes with instances have no search structure: their rays walk the reference-order entries, which
        // is where the placements and the prototypes' own entries are
        f.fast_min_pad = 0.0f;
        f.far_spheres = f.far_tris = 0;
        f.world_refs.clear();
        return;
    }
    auto box = [&](int e, float* mn, float* mx) {
        for (int k = 0; k < 3; k++) {
            mn[k] = f32(f.nodes[(size_t)e * 8 + k]);
            mx[k] = f32(f.nodes[(size_t)e * 8 + 4 + k]);
        }
    };
    // (in pieces on the builders' threads: every entry and every leaf run is written by the piece its entry lies in)
    parallelPieces(nNodes, [&](size_t firstEntry, size_t lastEntry) {
    std::vector<int> open;  // stack of box-tested entries whose subtree contains the current entry
    // the stack as the entries before this piece left it: down from the first entry along the skip links
    for (size_t x = 0; x < firstEntry;) {
        const uint32_t w3 = f.nodes[x * 8 + 3];
        const size_t skip = w3 & kNodeSkipMask;
        if (skip > firstEntry) {
            if (!(w3 & kNodeNoBox)) open.push_back((int)x);
            x++;
        } else {
            x = skip;
        }
    }
    for (size_t e = firstEntry; e < lastEntry; e++) {
        while (!open.empty() && (f.nodes[(size_t)open.back() * 8 + 3] & kNodeSkipMask) <= e) open.pop_back();
        const uint32_t w3 = f.nodes[e * 8 + 3];
        const bool hasBox = !(w3 & kNodeNoBox);
        if (hasBox) {
            float mn[3], mx[3];
            box((int)e, mn, mx);
            for (int j = (int)open.size() - 1; j >= 0; j--) {
                float amn[3], amx[3];
                box(open[j], amn, amx);
                bool contains = true;
                for (int k = 0; k < 3; k++) contains = contains && amn[k] <= mn[k] && amx[k] >= mx[k];
                if (!contains) {
                    f.node_outer[e] = open[j];
                    break;
                }
            }
            open.push_back((int)e);
        }
        const int32_t first = (int32_t)f.nodes[e * 8 + 7];
        if (first >= 0) {
            const int guard = open.empty() ? -1 : open.back();
            size_t p = (size_t)first;
            for (;;) {
                f.prim_guard[p] = guard;
                if (f.prim_refs[p] & kRefLast) break;
                p++;
            }
        }
    }
    });

    lap("guard tables");
    // ---- padded primitive boxes.  The padding is what keeps the fused box test conservative.  That
    // test evaluates plane * inv - origin * inv with the second product rounded first, so in position
    // terms its error is about (3 |origin| + 2 |plane|) * 2^-24.  The |plane| part is covered 800
    // times over by the relative term 2e-4 * (the box's largest coordinate) + 1e-4.  The |origin| part
    // is covered for ray origins up to pad * 2^21 from the coordinate origin (a sixth of the padding),
    // so every box is padded by at least extent * 2^-21, where `extent` is the largest coordinate of
    // any primitive: rays start at the camera or on a primitive, and paths do reach the far side of
    // the largest one (a path that gets inside the cover scene's ground sphere bounces around in it,
    // 2000 units from the small spheres, until the depth limit).  For primitives tiny against the
    // scene that floor is capped at a tenth of their own size; rays that start further than
    // fast_min_pad * 2^21 from the coordinate origin (a far camera, or the capped case) never use this
    // structure (fastTestUsable() in the kernels).  The hit-distance arithmetic of the primitives
    // themselves (worst: the sphere discriminant near grazing incidence, whose error grows with the
    // square of the distance to the sphere) is covered by margin, not by a bound.
    const bool deferred = f.defer_search_build;  // the device builds the structure: only its inputs are prepared
    std::vector<PrimBox> prims;
    f.world_refs.clear();
    for (int k = 0; k < 3; k++) {
        f.centroid_min[k] = std::numeric_limits<float>::infinity();
        f.centroid_max[k] = -std::numeric_limits<float>::infinity();
    }
    // (the passes over all primitives below run in pieces on the builders' threads; minima and maxima are
    // exact whatever the order, so the result does not depend on the thread count)
    std::mutex fold;
    float extent = 0.0f;
    for (size_t i = 0; i < spheres.size(); i++)
        if (f.sphere_ord[i] >= 0)
            for (int k = 0; k < 3; k++) extent = std::max(extent, std::fabs(spheres[i].c[k]) + std::fabs(spheres[i].r));
    parallelPieces(triangles.size(), [&](size_t first, size_t last) {
        float e = 0.0f;
        for (size_t i = first; i < last; i++)
            if (f.tri_ord[i] >= 0)
                for (int k = 0; k < 9; k++) e = std::max(e, std::fabs(triangles[i].pos[k]));
        std::lock_guard<std::mutex> g(fold);
        extent = std::max(extent, e);
    });
    const float originFloor = extent * (1.01f / 2097152.0f);
    // The padding of a primitive box has two parts.  `base`: the rounding of the fused box test (above).
    // `far`: the inaccuracy of the reference's own hit test, which grows with the distance D between ray origin
    // and primitive (FlatScene::far_*): it is budgeted for the distance D_typ within which this scene's rays
    // normally start -- the eye and the scene's own primitives -- and rays that start further away widen the
    // boxes themselves by the difference (farRayPad() in the kernels).
    const double kSphereErr = 32.0 / 16777216.0, kTriErr = 64.0 / 16777216.0;
    auto phantomSphere = [&](double r, double D) { return std::sqrt(r * r + kSphereErr * D * D) - r; };
    float minPad = std::numeric_limits<float>::infinity();
    auto primBox = [&](bool tri, size_t i) {
        PrimBox b;
        if (tri) {
            for (int k = 0; k < 3; k++) {
                const float a = triangles[i].pos[k], c = triangles[i].pos[3 + k], d = triangles[i].pos[6 + k];
                b.mn[k] = std::min(a, std::min(c, d));
                b.mx[k] = std::max(a, std::max(c, d));
            }
            b.ref = (uint32_t)i | kRefTriangle;
        } else {
            for (int k = 0; k < 3; k++) {
                b.mn[k] = spheres[i].c[k] - std::fabs(spheres[i].r);
                b.mx[k] = spheres[i].c[k] + std::fabs(spheres[i].r);
            }
            b.ref = (uint32_t)i;
        }
        return b;
    };
    auto basePad = [&](const PrimBox& b) {
        float m = 0.0f, size = 0.0f;
        for (int k = 0; k < 3; k++) {
            m = std::max(m, std::max(std::fabs(b.mn[k]), std::fabs(b.mx[k])));
            size = std::max(size, b.mx[k] - b.mn[k]);
        }
        return std::max(2e-4f * m + 1e-4f, std::min(originFloor, 0.1f * size));
    };
    // pass 1: base paddings, the smallest of them, and the primitives whose base padding does not also cover
    // their own test's inaccuracy at every distance a ray of this scene can start from
    std::vector<uint32_t>& world = f.world_refs;  // the primitives that are in the world: spheres, then triangles
    world.reserve(spheres.size() + triangles.size());
    for (size_t i = 0; i < spheres.size(); i++)
        if (f.sphere_ord[i] >= 0) world.push_back((uint32_t)i);
    for (size_t i = 0; i < triangles.size(); i++)
        if (f.tri_ord[i] >= 0) world.push_back((uint32_t)i | kRefTriangle);
    auto isTri = [&](size_t w) { return (world[w] & kRefTriangle) != 0; };
    auto idOf = [&](size_t w) { return (size_t)(world[w] & kRefIdMask); };
    lap("extent, list of primitives");
    std::vector<float> base(world.size());
    parallelPieces(world.size(), [&](size_t first, size_t last) {
        float m = std::numeric_limits<float>::infinity();
        for (size_t w = first; w < last; w++) {
            base[w] = basePad(primBox(isTri(w), idOf(w)));
            m = std::min(m, base[w]);
        }
        std::lock_guard<std::mutex> g(fold);
        minPad = std::min(minPad, m);
    });
    lap("base paddings");
    f.far_spheres = f.far_tris = 0;
    f.far_rmin = 0.0f;
    double farC[3] = {0, 0, 0}, farR = 0.0, rmin = std::numeric_limits<double>::infinity();
    {
        // the farthest a ray that uses the search structure can start from a primitive: origins are limited
        // to minPad * 2^21 per axis (fastTestUsable), primitives to `extent`
        const double reach = 1.7320508 * ((double)minPad * 2097152.0 + (double)extent) * 1.01;
        // The points of those primitives (sphere centres, triangle vertices): their number, their mean and the
        // largest distance from it.  Sums are formed per block of 65,536 primitives and the blocks' sums added in
        // order, so the mean does not depend on how many threads took part.
        auto isFar = [&](size_t w) {
            const size_t i = idOf(w);
            if (!isTri(w)) return phantomSphere(std::fabs((double)spheres[i].r), reach) > 0.5 * base[w];
            return kTriErr * reach > 0.5 * base[w];
        };
        constexpr size_t kBlock = 65536;
        const size_t nBlocks = (world.size() + kBlock - 1) / kBlock;
        struct BlockSum {
            double c[3] = {0, 0, 0}, rmin = std::numeric_limits<double>::infinity();
            size_t points = 0;
            int spheres = 0, tris = 0;
        };
        std::vector<BlockSum> sums(nBlocks);
        parallelPieces(nBlocks, [&](size_t firstBlock, size_t lastBlock) {
            for (size_t blk = firstBlock; blk < lastBlock; blk++) {
                BlockSum& bs = sums[blk];
                for (size_t w = blk * kBlock; w < std::min(world.size(), (blk + 1) * kBlock); w++) {
                    if (!isFar(w)) continue;  // covered at any distance
                    const size_t i = idOf(w);
                    if (!isTri(w)) {
                        bs.spheres++;
                        bs.rmin = std::min(bs.rmin, std::fabs((double)spheres[i].r));
                        for (int k = 0; k < 3; k++) bs.c[k] += spheres[i].c[k];
                        bs.points++;
                    } else {
                        bs.tris++;
                        for (int v = 0; v < 3; v++)
                            for (int k = 0; k < 3; k++) bs.c[k] += triangles[i].pos[3 * v + k];
                        bs.points += 3;
                    }
                }
            }
        });
        size_t nPoints = 0;
        for (const BlockSum& bs : sums) {
            f.far_spheres += bs.spheres;
            f.far_tris += bs.tris;
            rmin = std::min(rmin, bs.rmin);
            nPoints += bs.points;
            for (int k = 0; k < 3; k++) farC[k] += bs.c[k];
        }
        if (nPoints) {
            for (int k = 0; k < 3; k++) farC[k] /= (double)nPoints;
            std::mutex foldFar;
            parallelPieces(world.size(), [&](size_t first, size_t last) {
                double far2 = 0.0;
                auto reachOf = [&](const float* q) {
                    double d2 = 0;
                    for (int k = 0; k < 3; k++) d2 += ((double)q[k] - farC[k]) * ((double)q[k] - farC[k]);
                    far2 = std::max(far2, d2);
                };
                for (size_t w = first; w < last; w++) {
                    if (!isFar(w)) continue;
                    const size_t i = idOf(w);
                    if (!isTri(w))
                        reachOf(spheres[i].c);
                    else
                        for (int v = 0; v < 3; v++) reachOf(&triangles[i].pos[3 * v]);
                }
                std::lock_guard<std::mutex> g(foldFar);
                farR = std::max(farR, std::sqrt(far2));
            });
        }
    }
    // the distance budgeted for: from the eye to the far side of those primitives, or across them
    double eyeDist2 = 0;
    for (int k = 0; k < 3; k++) eyeDist2 += ((double)camera.origin[k] - farC[k]) * ((double)camera.origin[k] - farC[k]);
    const bool anyFar = f.far_spheres + f.far_tris > 0;
    const double dTyp = anyFar ? std::max(2.0 * farR, std::sqrt(eyeDist2) + farR) * 1.05 + 1e-3 : 0.0;
    for (int k = 0; k < 3; k++) f.far_center[k] = (float)farC[k];
    f.far_radius = (float)(farR * 1.001 + 1e-6);
    f.far_rmin = f.far_spheres ? (float)rmin : 0.0f;
    f.far_typical = (float)dTyp;
    lap("primitives far rays must widen for");
    // pass 2: the padded boxes
    if (!deferred) prims.resize(world.size());
    parallelPieces(world.size(), [&](size_t first, size_t last) {
        float cmn[3], cmx[3];
        for (int k = 0; k < 3; k++) {
            cmn[k] = std::numeric_limits<float>::infinity();
            cmx[k] = -std::numeric_limits<float>::infinity();
        }
        for (size_t w = first; w < last; w++) {
            PrimBox b = primBox(isTri(w), idOf(w));
            for (int k = 0; k < 3; k++) {
                const float c = 0.5f * (b.mn[k] + b.mx[k]);
                cmn[k] = std::min(cmn[k], c);
                cmx[k] = std::max(cmx[k], c);
            }
            const double far = isTri(w) ? kTriErr * dTyp : phantomSphere(std::fabs((double)spheres[idOf(w)].r), dTyp);
            const float d = (float)((double)base[w] + far * 1.01);
            for (int k = 0; k < 3; k++) {
                b.mn[k] -= d;
                b.mx[k] += d;
            }
            if (!deferred) prims[w] = b;
        }
        std::lock_guard<std::mutex> g(fold);
        for (int k = 0; k < 3; k++) {
            f.centroid_min[k] = std::min(f.centroid_min[k], cmn[k]);
            f.centroid_max[k] = std::max(f.centroid_max[k], cmx[k]);
        }
    });
    f.fast_min_pad = minPad;
    f.pad_floor = originFloor;
    lap("padded boxes");
    if (deferred && !f.world_refs.empty()) return;
    if (prims.empty()) {
        // nothing to search: one entry with an inverted box and no leaf run (whatever its test
        // says, the walk ends without a hit)
        const float mn[3] = {1.0f, 1.0f, 1.0f}, mx[3] = {-1.0f, -1.0f, -1.0f};
        Builder{prims, f.fast_nodes, f.fast_refs}.newEntry(mn, mx);
        return;
    }
    Builder bld{prims, f.fast_nodes, f.fast_refs};
    if (const char* e = std::getenv("RT_FAST_LEAF")) bld.kMaxLeaf = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("RT_FAST_ORDER")) bld.order = std::atoi(e);
    if (const char* e = std::getenv("RT_FAST_COLLAPSE")) bld.collapse = (float)std::atof(e);
    for (int k = 0; k < 3; k++) bld.eye[k] = camera.origin[k];
    bld.run();

    // ---- eight direction-ordered copies for scenes whose tables will not fit LDS (the kernels then
    // read them through the caches, where eight times the entries cost nothing per step)
    f.fast_oct.clear();
    f.fast_oct_stride = 0;
    auto al = [](size_t x) { return (x + 15) & ~size_t(15); };
    const size_t nE = f.fast_nodes.size() / 8, nS = spheres.size(), nT = triangles.size();
    const size_t ldsBytes = al((nE + 1) * 40) + al(f.fast_refs.size() * 4) + al(nS * 16) + al(nT * 48) + al(nS * 4) + al(nT * 4) +
                            al(materials.size() * 32) + al(nS * 4) + al(nT * 4);
    const char* octEnv = std::getenv("RT_FAST_OCTANTS");
    // Measured: 1 M triangles 68.7 -> 62.7 ms (35 instead of 48 box tests per ray); 10 M triangles
    // 145 -> 141 ms for 2 s more build time and 3 GB, so beyond 2 M entries the copies are not made;
    // Cornell + vase unchanged (14 instead of 21 box tests per ray, 296 ms either way).
    // (the LDS form of the entries also needs (nE + 1) * 40 and the leaf references to fit 16 bits)
    const bool fitsLds = ldsBytes <= 72 * 1024 && (nE + 1) * 40 + 4096 <= 0xFFFF && f.fast_refs.size() < 0xFFFE;
    const bool wantOct = octEnv ? std::atoi(octEnv) != 0 : (!fitsLds && nE <= (1u << 21));
    if (wantOct && nE * 8 + 8 < 0x7FFFFFFFu) {
        const uint32_t stride = (uint32_t)nE + 1, total = 8 * stride;
        std::vector<uint32_t> copy[8];
        std::atomic<int> next{0};
        auto work = [&] {
            for (int k = next.fetch_add(1); k < 8; k = next.fetch_add(1)) {
                Emission em{bld, copy[k], nullptr, bld.firstRef, 5};
                for (int a = 0; a < 3; a++) em.dir[a] = (k >> a) & 1 ? -1.0f : 1.0f;
                copy[k].reserve((size_t)stride * 8);
                em.run(bld.root);
            }
        };
        std::vector<std::thread> pool;
        const unsigned nThreads = std::min(8u, buildThreads());
        for (unsigned i = 1; i < nThreads; i++) {
            std::thread th;
            if (tryThread(th, work)) pool.push_back(std::move(th));
        }
        work();
        for (auto& t : pool) t.join();
        f.fast_oct.resize((size_t)total * 8);
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t base = k * stride;
            uint32_t* dst = &f.fast_oct[(size_t)base * 8];
            std::memcpy(dst, copy[k].data(), (size_t)nE * 32);
            for (size_t e = 0; e < nE; e++) {
                uint32_t& skip = dst[e * 8 + 6];
                skip = skip >= nE ? total : skip + base;
                // a copy is walked only by rays with its sign pattern, so it stores {entry plane, exit plane}
                // per axis -- min and max swapped where the direction is negative -- and the box test needs no
                // min / max to sort them (copy 0, all signs positive, keeps the plain layout)
                for (int a = 0; a < 3; a++)
                    if ((k >> a) & 1u) std::swap(dst[e * 8 + 2 * a], dst[e * 8 + 2 * a + 1]);
            }
            // the entry after the last one: x planes at +infinity, y and z planes at 0 -- the x distances
            // are both +inf or both -inf, the others finite, so the intervals never meet
            uint32_t* t = dst + (size_t)nE * 8;
            t[0] = t[1] = u32(INFINITY);
            t[2] = t[3] = t[4] = t[5] = 0u;
            t[6] = total;
            t[7] = 0xFFFFFFFFu;
            copy[k] = std::vector<uint32_t>();
        }
        f.fast_oct_stride = (int)stride;
    }
}

}  // namespace rt
