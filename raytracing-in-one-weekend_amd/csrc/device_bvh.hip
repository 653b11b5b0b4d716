// Device-side build of the search structure (the tree the kernels walk first; fast_bvh.cpp builds the
// same thing on the host with a surface-area heuristic).  No counterpart in the reference, whose
// BVHNode constructor (R/src/AccelerationStructures/Bvh.h:18-78) is sequential by its use of one
// random stream and std::sort; that tree is still built on the host, in the reference's order.  What
// is built here only has to be CONSERVATIVE: every primitive's padded box inside the boxes of all the
// entries above it.  Which primitive a ray returns is decided by the exact primitive tests and the
// guard test against the reference's boxes, so any such tree gives the reference's image.
//
// Method: padded primitive boxes -> 63-bit Morton codes of their centres -> radix sort (hipCUB) -> a binary tree
// over the sorted primitives -> entries written in walk order with their skip links.  The tree is built bottom-up
// by locally-ordered clustering (see Ploc below; the default), or top-down by the longest-common-prefix
// construction (Karras, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees", 2012;
// RT_DEVICE_TREE=lbvh), which is a third faster to build and costs a quarter more box tests per ray.  Internal
// nodes whose two children are both primitives become one entry with a run of two.  As with the host builder,
// scenes too large for LDS with at most 2 M entries get the entries eight times over, each copy with every
// node's children ordered near-to-far for one sign pattern of the ray direction; a single copy visits the child
// nearer to the eye first; and inner nodes whose box has more than half the area of the entry above are left
// out (clustering build only: a Morton-split tree has long chains of nearly equal boxes, and judged against
// their parents that rule flattened them into scans of hundreds of siblings, 462 box tests per ray).
// Measured on the vase scenes, host surface-area build = 1: Morton-split 1.55 (1 M triangles) / 1.57 (10 M) of its
// frame time, clustering 1.27 / 1.23, + left-out entries 1.19 / 1.23, + eye order 1.19 / 1.12, + the top of the
// tree (the last 8,192 clusters) built top-down on the host and a search radius of 64: 1.06 / 1.06; build 14 / 23 ms.
//
// Hand-offs between threads of one launch are avoided on purpose: the bottom-up pass runs as one
// launch per tree level (a node is computed in the launch after both its children were), so every
// value a thread reads was written by an EARLIER kernel on the same stream.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <functional>
#include <limits>
#include <string>
#include <vector>

#include "rt_device.h"

namespace rt {

namespace {

constexpr uint32_t kTri = 1u << 30, kLast = 1u << 31, kId = kTri - 1;

struct BuildArgs {
    int n;                      // primitives in the world
    const uint32_t* refs;       // their leaf references (kind flag + id), any order
    const float4* spheres;      // {c.xyz, r}
    const float4* tris;         // 3 x float4: v0, v1 - v0, v2 - v0
    float pad_floor;
    float far_typical;  // distance between ray origin and primitive for which the boxes are padded (see fast_bvh.cpp)
    float cmin[3], cinv[3];     // centre bounds: (c - cmin) * cinv in [0, 1]
    float4* pbox;               // 2 per primitive, input order: {min.xyz, -}, {max.xyz, -}
    uint64_t* keys;
    uint32_t* order;            // value array of the sort: position in `refs`
};

// Same padding rule as the host builder: 2e-4 of the box's largest coordinate + 1e-4, and at least
// the floor that covers far ray origins, capped at a tenth of the primitive's own size.
__device__ __forceinline__ void paddedBox(const BuildArgs& a, int i, float mn[3], float mx[3]) {
    const uint32_t ref = a.refs[i];
    const uint32_t id = ref & kId;
    if (ref & kTri) {
        const float4 v0 = a.tris[3 * (size_t)id], e1 = a.tris[3 * (size_t)id + 1], e2 = a.tris[3 * (size_t)id + 2];
        const float p0[3] = {v0.x, v0.y, v0.z}, p1[3] = {v0.x + e1.x, v0.y + e1.y, v0.z + e1.z},
                    p2[3] = {v0.x + e2.x, v0.y + e2.y, v0.z + e2.z};
        for (int k = 0; k < 3; k++) {
            mn[k] = fminf(p0[k], fminf(p1[k], p2[k]));
            mx[k] = fmaxf(p0[k], fmaxf(p1[k], p2[k]));
        }
    } else {
        const float4 s = a.spheres[id];
        const float c[3] = {s.x, s.y, s.z}, r = fabsf(s.w);
        for (int k = 0; k < 3; k++) {
            mn[k] = c[k] - r;
            mx[k] = c[k] + r;
        }
    }
    float m = 0.0f, size = 0.0f;
    for (int k = 0; k < 3; k++) {
        m = fmaxf(m, fmaxf(fabsf(mn[k]), fabsf(mx[k])));
        size = fmaxf(size, mx[k] - mn[k]);
    }
    // (the host pads by max(own, min(floor, size / 10)); the edges are re-derived here from v0 + edge,
    // which may differ from the stored vertex in the last bit -- 1e3 times less than the padding)
    // plus what the primitive's own hit test can be off by at the distance the scene's rays normally start from
    const float D = a.far_typical;
    const float rr = (ref & kTri) ? 0.0f : 0.5f * size;
    const float far = (ref & kTri) ? (64.0f / 16777216.0f) * D : sqrtf(rr * rr + (32.0f / 16777216.0f) * D * D) - rr;
    const float d = fmaxf(2e-4f * m + 1e-4f, fminf(a.pad_floor, 0.1f * size)) * 1.001f + far * 1.02f;
    for (int k = 0; k < 3; k++) {
        mn[k] -= d;
        mx[k] += d;
    }
}

__device__ __forceinline__ uint64_t spread21(uint32_t v) {  // 21 bits -> every third bit of 63
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void __launch_bounds__(256) primitiveKeys(BuildArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    float mn[3], mx[3];
    paddedBox(a, i, mn, mx);
    a.pbox[2 * (size_t)i] = make_float4(mn[0], mn[1], mn[2], 0.f);
    a.pbox[2 * (size_t)i + 1] = make_float4(mx[0], mx[1], mx[2], 0.f);
    uint64_t key = 0;
    for (int k = 0; k < 3; k++) {
        float t = (0.5f * (mn[k] + mx[k]) - a.cmin[k]) * a.cinv[k];
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        key |= spread21((uint32_t)(t * 2097151.0f)) << k;
    }
    a.keys[i] = key;
    a.order[i] = (uint32_t)i;
}

struct Tree {
    int n;
    const uint64_t* keys;     // sorted
    const uint32_t* order;    // sorted position -> position in `refs`
    int* left;                // per internal node: child >= 0 internal node, < 0 leaf ~child
    int* right;
    int* parent;              // [0, n-1): internal nodes, [n-1, 2n-1): leaves; -1 for the root
    int* first;               // first and last leaf (sorted position) below an internal node
    int* last;
    float4* nbox;             // 2 per internal node
    int* count;               // entries the subtree emits
    int* lcount;              // primitives below the node
    int root;                 // the root's internal-node number
    float* area;              // half the surface area of the node's box (clustering build)
    float* above;             // the same of the nearest ancestor that has an entry
    uint8_t* kept;            // 1: the node has an entry of its own; 0: left out, its children stand in its place
    int* stamp;               // number of the fitLevel launch that computed the node; 0 = not yet
    uint16_t* rfirst;         // bit k < 8: for direction-sign pattern k the right child is visited first; bit 8: for the
                              // single copy, whose children go nearer to the eye first (larger box on a tie)
    int* pending;             // [0]: internal nodes still without a box
};

// Length of the common prefix of the keys at sorted positions i and j (ties broken by position),
// -1 when j is outside the array.
__device__ __forceinline__ int prefix(const Tree& t, int i, int j) {
    if (j < 0 || j >= t.n) return -1;
    const uint64_t a = t.keys[i], b = t.keys[j];
    if (a == b) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)(a ^ b));
}

__global__ void __launch_bounds__(256) hierarchy(Tree t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n - 1) return;
    const int d = prefix(t, i, i + 1) - prefix(t, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = prefix(t, i, i - d);
    int lmax = 2;
    while (prefix(t, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int s = lmax / 2; s >= 1; s /= 2)
        if (prefix(t, i, i + (l + s) * d) > dmin) l += s;
    const int j = i + l * d;
    const int dnode = prefix(t, i, j);
    int s = 0, step = l;
    do {
        step = (step + 1) >> 1;
        if (prefix(t, i, i + (s + step) * d) > dnode) s += step;
    } while (step > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int lc = lo == gamma ? ~gamma : gamma, rc = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    t.left[i] = lc;
    t.right[i] = rc;
    t.first[i] = lo;
    t.last[i] = hi;
    t.parent[lc < 0 ? (t.n - 1) + ~lc : lc] = i;
    t.parent[rc < 0 ? (t.n - 1) + ~rc : rc] = i;
    t.count[i] = 0;
    t.stamp[i] = 0;
    t.rfirst[i] = 0;
    if (i == 0) t.parent[0] = -1;
}

// One level of the bottom-up pass, launch number `launch` (1, 2, ...): an internal node whose children
// were both finished by EARLIER launches (stamp below this launch's number) takes the union of their
// boxes and its subtree's entry count.  A child finished by another thread of this same launch is
// left for the next one, so nothing is read that a concurrent thread may still be writing.
__global__ void __launch_bounds__(256) fitLevel(Tree t, const float4* pbox, int launch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n - 1 || t.stamp[i] != 0) return;
    const int lc = t.left[i], rc = t.right[i];
    const int sl = lc < 0 ? -1 : t.stamp[lc], sr = rc < 0 ? -1 : t.stamp[rc];
    if (sl == 0 || sl >= launch || sr == 0 || sr >= launch) return;
    const int cl = lc < 0 ? 1 : t.count[lc], cr = rc < 0 ? 1 : t.count[rc];
    auto boxOf = [&](int c, float4& mn, float4& mx) {
        if (c < 0) {
            const size_t p = t.order[~c];
            mn = pbox[2 * p];
            mx = pbox[2 * p + 1];
        } else {
            mn = t.nbox[2 * (size_t)c];
            mx = t.nbox[2 * (size_t)c + 1];
        }
    };
    float4 amn, amx, bmn, bmx;
    boxOf(lc, amn, amx);
    boxOf(rc, bmn, bmx);
    const float4 mn = make_float4(fminf(amn.x, bmn.x), fminf(amn.y, bmn.y), fminf(amn.z, bmn.z), 0.f);
    const float4 mx = make_float4(fmaxf(amx.x, bmx.x), fmaxf(amx.y, bmx.y), fmaxf(amx.z, bmx.z), 0.f);
    t.nbox[2 * (size_t)i] = mn;
    t.nbox[2 * (size_t)i + 1] = mx;
    // for every sign pattern of the ray direction: is the right child's box entered first?
    unsigned mask = 0;
    for (int k = 0; k < 8; k++) {
        const float al = ((k & 1) ? -amx.x : amn.x) + ((k & 2) ? -amx.y : amn.y) + ((k & 4) ? -amx.z : amn.z);
        const float ar = ((k & 1) ? -bmx.x : bmn.x) + ((k & 2) ? -bmx.y : bmn.y) + ((k & 4) ? -bmx.z : bmn.z);
        if (ar < al) mask |= 1u << k;
    }
    t.rfirst[i] = (uint16_t)mask;
    // a node over two primitives is ONE entry with a run of two; otherwise itself plus both subtrees
    t.count[i] = (lc < 0 && rc < 0) ? 1 : 1 + cl + cr;
    t.lcount[i] = (lc < 0 ? 1 : t.lcount[lc]) + (rc < 0 ? 1 : t.lcount[rc]);
    t.stamp[i] = launch;
    atomicSub(t.pending, 1);
}

// ---------------------------------------------------------------------------------------
// Agglomerative build by parallel locally-ordered clustering (Meister and Bittner, "Parallel locally-ordered
// clustering for bounding volume hierarchy construction", 2018), the default: the Morton-sorted primitives
// are the first clusters; in every round each cluster looks at its `radius` (64) neighbours to either side in the
// array for the one whose union with it has the smallest surface area, pairs that chose each other are
// merged into a new node that takes the place of the first of the two, and the array is compacted.  The
// comparison key is (area, lower position, higher position), which no two pairs share, so the pair with the
// smallest key of all always chooses each other and every round merges something.  Trees built this way cost
// about as many box tests per ray as the host's surface-area build, where the Morton-split tree above costs
// 1.6 times as many.
// Like the rest of this file, a round is a sequence of launches: nothing is read that a thread of the same
// launch may still be writing (a node's children were created by earlier rounds).
struct Ploc {
    Tree t;
    const float4* pbox;
    const int* cidIn;       // cluster -> child code (>= 0 internal node, < 0 leaf ~sorted position)
    const float4* cboxIn;   // 2 per cluster
    int* cidOut;
    float4* cboxOut;
    int* nn;                // chosen neighbour per cluster
    unsigned long long* flags;    // per cluster: (starts a merge) << 32 | (stays in the array)
    unsigned long long* scanned;  // exclusive sums of `flags`
    const int* stateIn;     // {clusters, internal nodes created so far}
    int* stateOut;
    int radius;
    int round;
    int* roundBase;         // roundBase[r]: internal nodes that existed before round r
    float eye[3];
    float collapse;
};

__device__ __forceinline__ float unionArea(const float4& amn, const float4& amx, const float4& bmn, const float4& bmx) {
    const float dx = fmaxf(amx.x, bmx.x) - fminf(amn.x, bmn.x), dy = fmaxf(amx.y, bmx.y) - fminf(amn.y, bmn.y),
                dz = fmaxf(amx.z, bmx.z) - fminf(amn.z, bmn.z);
    return dx * dy + dy * dz + dz * dx;
}

// A new inner node over the children lc and rc (codes as in Tree::left) with boxes a and b: links both ways, box,
// area, the order its children are visited in for every direction octant and from the eye, and the entry and
// primitive counts below it.  Both children were finished by earlier launches.
__device__ __forceinline__ void makeNode(const Tree& t, int node, int lc, int rc, const float4& amn, const float4& amx, const float4& bmn,
                                         const float4& bmx, const float* eye, float4& mn, float4& mx) {
    mn = make_float4(fminf(amn.x, bmn.x), fminf(amn.y, bmn.y), fminf(amn.z, bmn.z), 0.f);
    mx = make_float4(fmaxf(amx.x, bmx.x), fmaxf(amx.y, bmx.y), fmaxf(amx.z, bmx.z), 0.f);
    t.left[node] = lc;
    t.right[node] = rc;
    t.parent[lc < 0 ? (t.n - 1) + ~lc : lc] = node;
    t.parent[rc < 0 ? (t.n - 1) + ~rc : rc] = node;
    t.parent[node] = -1;  // until a later round gives it one
    t.nbox[2 * (size_t)node] = mn;
    t.nbox[2 * (size_t)node + 1] = mx;
    unsigned mask = 0;  // for every sign pattern of the ray direction: is the right child's box entered first?
    for (int k = 0; k < 8; k++) {
        const float al = ((k & 1) ? -amx.x : amn.x) + ((k & 2) ? -amx.y : amn.y) + ((k & 4) ? -amx.z : amn.z);
        const float ar = ((k & 1) ? -bmx.x : bmn.x) + ((k & 2) ? -bmx.y : bmn.y) + ((k & 4) ? -bmx.z : bmn.z);
        if (ar < al) mask |= 1u << k;
    }
    {
        // the single copy: the child whose box is nearer to the eye first, the larger one on a tie (host builder's rule)
        auto eyeDist2 = [&](const float4& bn, const float4& bx) {
            const float dx = fmaxf(0.0f, fmaxf(bn.x - eye[0], eye[0] - bx.x)), dy = fmaxf(0.0f, fmaxf(bn.y - eye[1], eye[1] - bx.y)),
                        dz = fmaxf(0.0f, fmaxf(bn.z - eye[2], eye[2] - bx.z));
            return dx * dx + dy * dy + dz * dz;
        };
        auto halfArea = [](const float4& bn, const float4& bx) {
            const float dx = bx.x - bn.x, dy = bx.y - bn.y, dz = bx.z - bn.z;
            return dx * dy + dy * dz + dz * dx;
        };
        const float dl = eyeDist2(amn, amx), dr = eyeDist2(bmn, bmx);
        if (dl == dr ? halfArea(amn, amx) < halfArea(bmn, bmx) : dl > dr) mask |= 1u << 8;
    }
    t.rfirst[node] = (uint16_t)mask;
    const int cl = lc < 0 ? 1 : t.count[lc], cr = rc < 0 ? 1 : t.count[rc];
    t.count[node] = (lc < 0 && rc < 0) ? 1 : 1 + cl + cr;
    t.lcount[node] = (lc < 0 ? 1 : t.lcount[lc]) + (rc < 0 ? 1 : t.lcount[rc]);
    {
        const float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z;
        t.area[node] = dx * dy + dy * dz + dz * dx;
    }
}

__global__ void __launch_bounds__(256) plocInit(Tree t, const float4* pbox, int* cid, float4* cbox, int* state) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        state[0] = t.n;
        state[1] = 0;
    }
    if (i >= t.n) return;
    const size_t p = t.order[i];
    cid[i] = ~i;
    cbox[2 * (size_t)i] = pbox[2 * p];
    cbox[2 * (size_t)i + 1] = pbox[2 * p + 1];
    t.parent[(t.n - 1) + i] = -1;
}

__global__ void __launch_bounds__(256) plocNearest(Ploc a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = a.stateIn[0];
    if (i >= m) return;
    const float4 mn = a.cboxIn[2 * (size_t)i], mx = a.cboxIn[2 * (size_t)i + 1];
    const int lo = i - a.radius < 0 ? 0 : i - a.radius, hi = i + a.radius > m - 1 ? m - 1 : i + a.radius;
    float best = __builtin_inff();
    int bj = -1;
    for (int j = lo; j <= hi; j++) {
        if (j == i) continue;
        const float d = unionArea(mn, mx, a.cboxIn[2 * (size_t)j], a.cboxIn[2 * (size_t)j + 1]);
        // key (d, lower position, higher position); j runs upwards, so among equal areas the first j below i has
        // the smallest key of those below, and any j below i beats every j above it
        if (d < best || bj < 0) {
            best = d;
            bj = j;
        }
    }
    a.nn[i] = bj;
}

__global__ void __launch_bounds__(256) plocFlags(Ploc a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = a.stateIn[0];
    if (i >= m) return;
    const int j = a.nn[i];
    const bool mutual = j >= 0 && a.nn[j] == i;
    const unsigned long long leader = mutual && i < j ? 1ull : 0ull, stays = mutual && i > j ? 0ull : 1ull;
    a.flags[i] = (leader << 32) | stays;
}

__global__ void __launch_bounds__(256) plocMerge(Ploc a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = a.stateIn[0];
    if (i >= m) return;
    const Tree& t = a.t;
    const unsigned long long f = a.flags[i], sc = a.scanned[i];
    if (i == m - 1) {
        a.stateOut[0] = (int)(unsigned)(sc & 0xFFFFFFFFull) + (int)(unsigned)(f & 0xFFFFFFFFull);
        a.stateOut[1] = a.stateIn[1] + (int)(unsigned)(sc >> 32) + (int)(unsigned)(f >> 32);
        a.roundBase[a.round] = a.stateIn[1];
        a.roundBase[a.round + 1] = a.stateOut[1];
    }
    if (!(f & 0xFFFFFFFFull)) return;  // merged into its partner's place
    const size_t pos = (size_t)(sc & 0xFFFFFFFFull);
    if (!(f >> 32)) {
        a.cidOut[pos] = a.cidIn[i];
        a.cboxOut[2 * pos] = a.cboxIn[2 * (size_t)i];
        a.cboxOut[2 * pos + 1] = a.cboxIn[2 * (size_t)i + 1];
        return;
    }
    const int j = a.nn[i];
    const int node = a.stateIn[1] + (int)(unsigned)(sc >> 32);
    float4 mn, mx;
    makeNode(t, node, a.cidIn[i], a.cidIn[j], a.cboxIn[2 * (size_t)i], a.cboxIn[2 * (size_t)i + 1], a.cboxIn[2 * (size_t)j],
             a.cboxIn[2 * (size_t)j + 1], a.eye, mn, mx);
    a.cidOut[pos] = node;
    a.cboxOut[2 * pos] = mn;
    a.cboxOut[2 * pos + 1] = mx;
}

// ---- the top of the tree.  Bottom-up clustering is at its weakest where few, large clusters are left: a pair
// is merged because it is the best one in sight, not because the split it makes of everything above is a good
// one, and every ray pays for those entries.  So the rounds stop when a few thousand clusters are left; those
// go to the host, which builds the tree over them top-down by the surface-area heuristic (every split position
// along every axis tried: the clusters are few), and the nodes come back as child pairs, numbered by height, to
// be finished one height per launch like a round's nodes.
__global__ void __launch_bounds__(256) clusterWeights(Tree t, const int* cid, int m, int* weight) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int c = cid[i];
    weight[i] = c < 0 ? 1 : t.lcount[c];
}

__global__ void __launch_bounds__(256) topNodes(Tree t, const float4* pbox, int lo, int hi, const int2* children, float ex, float ey, float ez) {
    const int node = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= hi) return;
    const int2 c = children[node - lo];
    auto boxOf = [&](int code, float4& bn, float4& bx) {
        if (code < 0) {
            const size_t p = t.order[~code];
            bn = pbox[2 * p];
            bx = pbox[2 * p + 1];
        } else {
            bn = t.nbox[2 * (size_t)code];
            bx = t.nbox[2 * (size_t)code + 1];
        }
    };
    float4 amn, amx, bmn, bmx, mn, mx;
    boxOf(c.x, amn, amx);
    boxOf(c.y, bmn, bmx);
    const float eye[3] = {ex, ey, ez};
    makeNode(t, node, c.x, c.y, amn, amx, bmn, bmx, eye, mn, mx);
}

struct TopCluster {
    float mn[3], mx[3];
    int code;    // child code of the cluster (Tree::left)
    int weight;  // primitives below it
};
struct TopNode {
    int left, right;  // >= 0: cluster number, < 0: ~(top node number)
    int height;       // 1 + the larger of the children's (clusters: 0)
};

// Top-down surface-area build over clusters[lo, hi) (reordered in place); returns the child reference of the subtree.
// Cost of a split = area(left) * primitives(left) + area(right) * primitives(right), every position of the
// clusters sorted by centre along every axis; ties go to the earlier axis and position.
struct TopBuilder {
    std::vector<TopCluster>& clusters;
    std::vector<TopNode>& nodes;
    std::vector<float> suffix;
    std::vector<TopCluster> sorted;

    static float halfArea(const float* mn, const float* mx) {
        const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return dx * dy + dy * dz + dz * dx;
    }
    static void grow(float* mn, float* mx, const TopCluster& c) {
        for (int k = 0; k < 3; k++) {
            mn[k] = std::min(mn[k], c.mn[k]);
            mx[k] = std::max(mx[k], c.mx[k]);
        }
    }
    static void sortAlong(std::vector<TopCluster>::iterator first, std::vector<TopCluster>::iterator last, int axis) {
        std::stable_sort(first, last, [axis](const TopCluster& a, const TopCluster& b) { return a.mn[axis] + a.mx[axis] < b.mn[axis] + b.mx[axis]; });
    }
    // sorts clusters[lo, hi) along the best axis and returns the number that go left; equal costs go to the
    // earlier axis and the more even split
    size_t split(size_t lo, size_t hi) {
        const size_t cnt = hi - lo;
        const float inf = std::numeric_limits<float>::infinity();
        double bestCost = std::numeric_limits<double>::infinity();
        size_t bestAt = cnt / 2;
        int bestAxis = -1;
        auto uneven = [cnt](size_t k) { return k > cnt - k ? 2 * k - cnt : cnt - 2 * k; };
        suffix.resize(cnt + 1);
        for (int axis = 0; axis < 3; axis++) {
            sorted.assign(clusters.begin() + (long)lo, clusters.begin() + (long)hi);
            sortAlong(sorted.begin(), sorted.end(), axis);
            float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
            for (size_t k = cnt; k-- > 0;) {
                grow(mn, mx, sorted[k]);
                suffix[k] = halfArea(mn, mx);
            }
            for (int k = 0; k < 3; k++) {
                mn[k] = inf;
                mx[k] = -inf;
            }
            long long wl = 0, wt = 0;
            for (size_t k = 0; k < cnt; k++) wt += sorted[k].weight;
            for (size_t k = 1; k < cnt; k++) {
                grow(mn, mx, sorted[k - 1]);
                wl += sorted[k - 1].weight;
                const double cost = (double)halfArea(mn, mx) * (double)wl + (double)suffix[k] * (double)(wt - wl);
                if (cost < bestCost || (cost == bestCost && axis == bestAxis && uneven(k) < uneven(bestAt))) {
                    bestCost = cost;
                    bestAt = k;
                    bestAxis = axis;
                }
            }
        }
        if (bestAxis >= 0) sortAlong(clusters.begin() + (long)lo, clusters.begin() + (long)hi, bestAxis);
        return bestAt;
    }
    // (below 64 levels of one-sided splits the halves are taken as they lie: bounds the work on degenerate input)
    int build(size_t lo, size_t hi, int depth, int* height) {
        if (hi - lo == 1) {
            *height = 0;
            return (int)lo;
        }
        const size_t k = depth < 64 ? split(lo, hi) : (hi - lo) / 2;
        int hl = 0, hr = 0;
        const int l = build(lo, lo + k, depth + 1, &hl);
        const int r = build(lo + k, hi, depth + 1, &hr);
        *height = 1 + std::max(hl, hr);
        nodes.push_back({l, r, *height});
        return ~(int)(nodes.size() - 1);
    }
};

// Which nodes get an entry (the host builder's rule, fast_bvh.cpp: Emission::gather): a node over two subtrees whose
// box has more than half the area of the entry above it would pass nearly every ray that entry passes, so it is
// left out and its children are tested in its place.  "The entry above" is the nearest ancestor that HAS an entry,
// so the rule cannot flatten a long chain of nearly equal boxes into one scan: after a few levels the area has
// halved.  Parents are decided before children: one launch per round of the build, last round first.
__global__ void __launch_bounds__(256) keptOfRound(Tree t, int lo, int hi, float collapse) {
    const int i = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    const int p = t.parent[i];
    const float a = t.area[i];
    const float abv = p < 0 ? a : (t.kept[p] ? t.area[p] : t.above[p]);
    const bool twig = t.left[i] < 0 && t.right[i] < 0;
    t.above[i] = abv;
    t.kept[i] = (twig || !(a > collapse * abv)) ? 1 : 0;
}
// Entries below and including each node, children before parents: one launch per round, first round first.
__global__ void __launch_bounds__(256) countOfRound(Tree t, int lo, int hi) {
    const int i = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    const int lc = t.left[i], rc = t.right[i];
    if (lc < 0 && rc < 0) {
        t.count[i] = 1;
        return;
    }
    t.count[i] = (t.kept[i] ? 1 : 0) + (lc < 0 ? 1 : t.count[lc]) + (rc < 0 ? 1 : t.count[rc]);
}

struct EmitArgs {
    Tree t;
    const float4* pbox;
    const uint32_t* refs;
    float4* entries;    // 2 per entry: {min.x, max.x, min.y, max.y}, {min.z, max.z, skip, first_ref}
    uint32_t* out_refs; // leaf references in sorted order
    int copies;         // 1, or 8 direction-ordered copies of `stride` entries each
    int n_entries;      // entries of one copy
    int stride;         // n_entries + 1 when copies == 8 (one closing entry per copy)
};

// Entries a child contributes to its parent's subtree.
__device__ __forceinline__ int emitted(const Tree& t, int child) { return child < 0 ? 1 : t.count[child]; }

// Position of a node in walk order for each child order in use (entry, first child's subtree, second
// child's subtree): one for every ancestor, plus the size of the sibling's subtree wherever the node
// lies under the child that is visited second.
__device__ __forceinline__ void walkIndex(const Tree& t, int slot, int copies, int idx[8]) {
    for (int k = 0; k < 8; k++) idx[k] = 0;
    int cur = slot;
    for (int p = t.parent[cur]; p >= 0; p = t.parent[cur]) {
        const int lc = t.left[p], rc = t.right[p];
        const int rslot = rc < 0 ? (t.n - 1) + ~rc : rc;
        const bool isRight = rslot == cur;
        const int sibling = emitted(t, isRight ? lc : rc);
        const unsigned mask = copies == 8 ? t.rfirst[p] : (unsigned)(t.rfirst[p] >> 8);
        const int own = t.kept[p] ? 1 : 0;  // an ancestor that was left out has no entry before its children's
        for (int k = 0; k < copies; k++) {
            const bool rightFirst = (mask >> k) & 1u;
            idx[k] += own + ((isRight != rightFirst) ? sibling : 0);
        }
        cur = p;
    }
}

// Position of a primitive in the leaf references: its rank among the primitives in walk order.  The runs are shared
// by all eight copies, whose walks differ, so with copies they follow the walk that takes left children first; the
// single copy's runs follow its own walk (nearer to the eye first).
__device__ __forceinline__ int leafIndex(const Tree& t, int slot, int copies) {
    int idx = 0, cur = slot;
    for (int p = t.parent[cur]; p >= 0; p = t.parent[cur]) {
        const int lc = t.left[p], rc = t.right[p];
        const int rslot = rc < 0 ? (t.n - 1) + ~rc : rc;
        const bool isRight = rslot == cur;
        const bool rightFirst = copies == 1 && ((t.rfirst[p] >> 8) & 1u);
        if (isRight != rightFirst) {
            const int sib = isRight ? lc : rc;
            idx += sib < 0 ? 1 : t.lcount[sib];
        }
        cur = p;
    }
    return idx;
}

__device__ __forceinline__ void writeEntry(const EmitArgs& e, const int idx[8], float4 mn, float4 mx, int count, int firstRef) {
    const int total = e.copies * e.stride;
    for (int k = 0; k < e.copies; k++) {
        const int base = k * e.stride;
        int skip = idx[k] + count;
        if (e.copies == 8) skip = skip >= e.n_entries ? total : base + skip;  // past the end = the walk is over
        const size_t at = (size_t)(base + idx[k]);
        // copy k serves rays whose direction is negative on the axes of k's set bits: {entry plane, exit plane}
        const bool nx = (k & 1) != 0, ny = (k & 2) != 0, nz = (k & 4) != 0;
        e.entries[2 * at] = make_float4(nx ? mx.x : mn.x, nx ? mn.x : mx.x, ny ? mx.y : mn.y, ny ? mn.y : mx.y);
        e.entries[2 * at + 1] = make_float4(nz ? mx.z : mn.z, nz ? mn.z : mx.z, __int_as_float(skip), __int_as_float(firstRef));
    }
}

__global__ void __launch_bounds__(256) emitEntries(EmitArgs e) {
    const Tree& t = e.t;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int idx[8];
    if (i < t.n - 1 && t.kept[i]) {  // internal node i
        const bool twig = t.left[i] < 0 && t.right[i] < 0;
        walkIndex(t, i, e.copies, idx);
        writeEntry(e, idx, t.nbox[2 * (size_t)i], t.nbox[2 * (size_t)i + 1], t.count[i],
                   twig ? leafIndex(t, (t.n - 1) + ~((e.copies == 1 && ((t.rfirst[i] >> 8) & 1u)) ? t.right[i] : t.left[i]), e.copies) : -1);
    }
    if (i < t.n) {      // leaf at sorted position i
        const size_t p = t.order[i];
        const int par = t.n > 1 ? t.parent[(t.n - 1) + i] : -1;
        const bool underTwig = par >= 0 && t.left[par] < 0 && t.right[par] < 0;
        const bool twigRightFirst = underTwig && e.copies == 1 && ((t.rfirst[par] >> 8) & 1u);
        const bool lastOfRun = !underTwig || ~(twigRightFirst ? t.left[par] : t.right[par]) == i;
        const int at = t.n > 1 ? leafIndex(t, (t.n - 1) + i, e.copies) : 0;
        e.out_refs[at] = e.refs[p] | (lastOfRun ? kLast : 0u);
        if (!underTwig) {  // an entry of its own with a run of one
            if (t.n > 1)
                walkIndex(t, (t.n - 1) + i, e.copies, idx);
            else
                for (int k = 0; k < 8; k++) idx[k] = 0;
            writeEntry(e, idx, e.pbox[2 * p], e.pbox[2 * p + 1], 1, at);
        }
    }
    if (e.copies == 8 && i < 8) {
        // the entry after a copy's last one: x planes at +infinity, y and z planes at 0 -- the x distances
        // are both +inf or both -inf, the others finite, so the intervals never meet and the walk ends
        const size_t at = (size_t)i * e.stride + e.n_entries;
        const float inf = __builtin_inff();
        e.entries[2 * at] = make_float4(inf, inf, 0.f, 0.f);
        e.entries[2 * at + 1] = make_float4(0.f, 0.f, __int_as_float(8 * e.stride), __int_as_float(-1));
    }
}

}  // namespace

// Builds the search structure of `n` primitives on the device.  All buffers are allocated here and
// handed to the caller (who frees them with hipFree): *entries (2 float4 per entry), *out_refs.
// Returns the number of entries through *n_entries.  Blocking (it reads two words back).
// want_copies(entries of one copy) decides whether the eight direction-ordered copies are written; *stride
// comes back 0 (one copy) or entries + 1.
hipError_t buildSearchStructureOnDevice(int n, const uint32_t* d_refs, const float4* d_spheres, const float4* d_tris,
                                        float pad_floor, float far_typical, const float cmin[3], const float cmax[3], const float eye[3], hipStream_t stream,
                                        const std::function<bool(int)>& want_copies, float4** entries, uint32_t** out_refs,
                                        int* n_entries, int* stride, float* build_ms) {
    *stride = 0;
    *entries = nullptr;
    *out_refs = nullptr;
    *n_entries = 0;
    if (n <= 0) return hipErrorInvalidValue;
    hipError_t e = hipSuccess;
    std::vector<void*> temp;
    auto alloc = [&](void** p, size_t bytes) {
        if (e != hipSuccess) return;
        e = hipMalloc(p, bytes ? bytes : 16);
        if (e == hipSuccess) temp.push_back(*p);
    };
    const size_t N = (size_t)n;
    float4 *pbox = nullptr, *nbox = nullptr;
    uint64_t *keysA = nullptr, *keysB = nullptr;
    uint32_t *ordA = nullptr, *ordB = nullptr;
    int *left = nullptr, *right = nullptr, *parent = nullptr, *first = nullptr, *last = nullptr, *count = nullptr, *stamp = nullptr, *pending = nullptr;
    int *lcount = nullptr, *cidA = nullptr, *cidB = nullptr, *nn = nullptr, *state = nullptr;
    float4 *cboxA = nullptr, *cboxB = nullptr;
    unsigned long long *flags = nullptr, *scanned = nullptr;
    float *area = nullptr, *above = nullptr;
    uint8_t* kept = nullptr;
    int* roundBase = nullptr;
    constexpr int kMaxRounds = 4096;
    void* scanTemp = nullptr;
    size_t scanBytes = 0;
    // RT_DEVICE_TREE=lbvh: the Morton-split tree (Karras) instead of the clustering build; RT_PLOC_RADIUS: how far
    // to either side a cluster looks for its partner
    const char* treeEnv = std::getenv("RT_DEVICE_TREE");
    const bool ploc = !(treeEnv && std::string(treeEnv) == "lbvh");
    const char* radiusEnv = std::getenv("RT_PLOC_RADIUS");
    const int radius = std::max(1, std::min(512, radiusEnv ? std::atoi(radiusEnv) : 64));
    uint16_t* rfirst = nullptr;
    void* sortTemp = nullptr;
    size_t sortBytes = 0;
    alloc((void**)&pbox, N * 32);
    alloc((void**)&nbox, N * 32);
    alloc((void**)&keysA, N * 8);
    alloc((void**)&keysB, N * 8);
    alloc((void**)&ordA, N * 4);
    alloc((void**)&ordB, N * 4);
    alloc((void**)&left, N * 4);
    alloc((void**)&right, N * 4);
    alloc((void**)&parent, 2 * N * 4);
    alloc((void**)&first, N * 4);
    alloc((void**)&last, N * 4);
    alloc((void**)&count, N * 4);
    alloc((void**)&lcount, N * 4);
    alloc((void**)&kept, N);
    if (e == hipSuccess) e = hipMemsetAsync(kept, 1, N, stream);  // the Morton-split tree keeps every node
    if (ploc) {
        alloc((void**)&area, N * 4);
        alloc((void**)&above, N * 4);
        alloc((void**)&roundBase, (kMaxRounds + 2) * sizeof(int));
        alloc((void**)&cidA, N * 4);
        alloc((void**)&cidB, N * 4);
        alloc((void**)&cboxA, N * 32);
        alloc((void**)&cboxB, N * 32);
        alloc((void**)&nn, N * 4);
        alloc((void**)&flags, N * 8);
        alloc((void**)&scanned, N * 8);
        alloc((void**)&state, 64);
        if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, flags, scanned, n, stream);
        alloc(&scanTemp, scanBytes);
    }
    alloc((void**)&stamp, N * 4);
    alloc((void**)&rfirst, N * 2);
    alloc((void**)&pending, 256);
    if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(nullptr, sortBytes, keysA, keysB, ordA, ordB, n, 0, 63, stream);
    alloc(&sortTemp, sortBytes);
    float4* outEntries = nullptr;
    uint32_t* outRefs = nullptr;
    if (e == hipSuccess) e = hipMalloc((void**)&outRefs, N * 4);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (e == hipSuccess) e = hipEventCreate(&ev0);
    if (e == hipSuccess) e = hipEventCreate(&ev1);
    if (e == hipSuccess) e = hipEventRecord(ev0, stream);
    const int blocks = (n + 255) / 256;
    if (e == hipSuccess) {
        BuildArgs a{};
        a.n = n;
        a.refs = d_refs;
        a.spheres = d_spheres;
        a.tris = d_tris;
        a.pad_floor = pad_floor;
        a.far_typical = far_typical;
        for (int k = 0; k < 3; k++) {
            a.cmin[k] = cmin[k];
            const float ext = cmax[k] - cmin[k];
            a.cinv[k] = ext > 0.0f ? 1.0f / ext : 0.0f;
        }
        a.pbox = pbox;
        a.keys = keysA;
        a.order = ordA;
        primitiveKeys<<<blocks, 256, 0, stream>>>(a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(sortTemp, sortBytes, keysA, keysB, ordA, ordB, n, 0, 63, stream);
    Tree t{n, keysB, ordB, left, right, parent, first, last, nbox, count, lcount, 0, area, above, kept, stamp, rfirst, pending};
    int total = 1;
    if (e == hipSuccess && n > 1 && ploc) {
        t.root = n - 2;  // nodes are numbered as they are created, and the root is created last
        plocInit<<<blocks, 256, 0, stream>>>(t, pbox, cidA, cboxA, state);
        e = hipGetLastError();
        int m = n;       // clusters left (an upper bound between the read-backs)
        int* stateIn = state;
        int* stateOut = state + 8;
        int rounds = 0;
        // RT_DEVICE_TOP: the clustering stops at this many clusters (and a sixteenth of the primitives at most) and
        // the host builds the tree above them; 0 or 1 = clustering all the way up
        const int topWanted = std::getenv("RT_DEVICE_TOP") ? std::atoi(std::getenv("RT_DEVICE_TOP")) : 8192;
        const int topStop = std::max(1, std::min(topWanted, n / 16));
        for (int round = 0; e == hipSuccess && m > topStop; round++, rounds++) {
            if (round >= kMaxRounds) {
                e = hipErrorUnknown;  // every round merges at least one pair
                break;
            }
            const int grid = (m + 255) / 256;
            Ploc a{t, pbox, cidA, cboxA, cidB, cboxB, nn, flags, scanned, stateIn, stateOut, radius, round, roundBase, {eye[0], eye[1], eye[2]}, 0.0f};
            plocNearest<<<grid, 256, 0, stream>>>(a);
            plocFlags<<<grid, 256, 0, stream>>>(a);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(scanTemp, scanBytes, flags, scanned, m, stream);
            if (e == hipSuccess) {
                plocMerge<<<grid, 256, 0, stream>>>(a);
                e = hipGetLastError();
            }
            std::swap(cidA, cidB);
            std::swap(cboxA, cboxB);
            std::swap(stateIn, stateOut);
            if ((round & 3) == 3 || m <= std::max(4096, 4 * topStop)) {  // how many are left, every fourth round (every round near the end)
                int st[2] = {0, 0};
                if (e == hipSuccess) e = hipMemcpyAsync(st, stateIn, sizeof st, hipMemcpyDeviceToHost, stream);
                if (e == hipSuccess) e = hipStreamSynchronize(stream);
                m = st[0];
            }
        }
        // which nodes have entries (parents first), then the entries per subtree (children first)
        std::vector<int> base((size_t)rounds + 1, 0);
        if (e == hipSuccess && rounds > 0) e = hipMemcpyAsync(base.data(), roundBase, base.size() * sizeof(int), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        std::vector<int2> topChildren;
        if (e == hipSuccess && m > 1) {
            // the top of the tree: the m clusters that are left go to the host (TopBuilder) and come back as nodes
            const int made = n - m;  // every merge took one cluster away
            if (base.back() != made) e = hipErrorUnknown;
            std::vector<int> code((size_t)m), weight((size_t)m);
            std::vector<float4> cbox(2 * (size_t)m);
            if (e == hipSuccess) {
                clusterWeights<<<(m + 255) / 256, 256, 0, stream>>>(t, cidA, m, nn);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(code.data(), cidA, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipMemcpyAsync(weight.data(), nn, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipMemcpyAsync(cbox.data(), cboxA, 2 * (size_t)m * sizeof(float4), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e == hipSuccess) {
                std::vector<TopCluster> clusters((size_t)m);
                for (size_t i = 0; i < (size_t)m; i++) {
                    const float4 bn = cbox[2 * i], bx = cbox[2 * i + 1];
                    clusters[i] = {{bn.x, bn.y, bn.z}, {bx.x, bx.y, bx.z}, code[i], weight[i]};
                }
                std::vector<TopNode> top;
                top.reserve((size_t)m);
                int height = 0;
                TopBuilder{clusters, top, {}, {}}.build(0, (size_t)m, 0, &height);
                // numbered by height, so that a node's children have smaller numbers and the root the last one
                std::vector<int> byHeight(top.size()), number(top.size());
                for (size_t j = 0; j < top.size(); j++) byHeight[j] = (int)j;
                std::stable_sort(byHeight.begin(), byHeight.end(), [&](int x, int y) { return top[(size_t)x].height < top[(size_t)y].height; });
                for (size_t r = 0; r < byHeight.size(); r++) number[(size_t)byHeight[r]] = made + (int)r;
                topChildren.resize(top.size());
                auto codeOf = [&](int child) { return child >= 0 ? clusters[(size_t)child].code : number[(size_t)~child]; };
                for (size_t r = 0; r < byHeight.size(); r++) {
                    const TopNode& tn = top[(size_t)byHeight[r]];
                    topChildren[r] = make_int2(codeOf(tn.left), codeOf(tn.right));
                }
                int2* dChildren = reinterpret_cast<int2*>(flags);  // free again: at least as large as the node count
                e = hipMemcpyAsync(dChildren, topChildren.data(), topChildren.size() * sizeof(int2), hipMemcpyHostToDevice, stream);
                for (size_t r = 0; r < byHeight.size() && e == hipSuccess;) {
                    size_t q = r;
                    while (q < byHeight.size() && top[(size_t)byHeight[q]].height == top[(size_t)byHeight[r]].height) q++;
                    const int lo = made + (int)r, hi = made + (int)q;
                    topNodes<<<(hi - lo + 255) / 256, 256, 0, stream>>>(t, pbox, lo, hi, dChildren + r, eye[0], eye[1], eye[2]);
                    e = hipGetLastError();
                    base.push_back(hi);  // one more "round" for the two passes below
                    r = q;
                }
                if (e == hipSuccess) e = hipStreamSynchronize(stream);
                if (e == hipSuccess && base.back() != n - 1) e = hipErrorUnknown;
            }
        }
        rounds = (int)base.size() - 1;
        // RT_DEVICE_COLLAPSE: the area ratio above which an inner node is left out (0 = keep every node)
        const float collapse = std::getenv("RT_DEVICE_COLLAPSE") ? (float)std::atof(std::getenv("RT_DEVICE_COLLAPSE")) : 0.5f;
        if (collapse > 0.0f) {
            for (int r = rounds - 1; r >= 0 && e == hipSuccess; r--) {
                const int lo = base[(size_t)r], hi = base[(size_t)r + 1];
                if (hi > lo) keptOfRound<<<(hi - lo + 255) / 256, 256, 0, stream>>>(t, lo, hi, collapse);
                e = hipGetLastError();
            }
            for (int r = 0; r < rounds && e == hipSuccess; r++) {
                const int lo = base[(size_t)r], hi = base[(size_t)r + 1];
                if (hi > lo) countOfRound<<<(hi - lo + 255) / 256, 256, 0, stream>>>(t, lo, hi);
                e = hipGetLastError();
            }
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&total, count + t.root, sizeof total, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
    } else if (e == hipSuccess && n > 1) {
        int todo = n - 1;
        e = hipMemcpyAsync(pending, &todo, sizeof todo, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) {
            hierarchy<<<blocks, 256, 0, stream>>>(t);
            e = hipGetLastError();
        }
        // one launch per level; how many nodes are left is read back every eight levels (a Morton tree
        // is at most 63 + 32 levels deep: 63 key bits, then ties split by position)
        for (int launch = 1; e == hipSuccess && todo > 0;) {
            if (launch > 256) {
                e = hipErrorUnknown;  // cannot happen for a well-formed tree
                break;
            }
            for (int k = 0; k < 8 && e == hipSuccess; k++, launch++) {
                fitLevel<<<blocks, 256, 0, stream>>>(t, pbox, launch);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(&todo, pending, sizeof todo, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&total, count, sizeof total, hipMemcpyDeviceToHost, stream);  // count of the root
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
    }
    int copies = 1, strideOut = 0;
    if (e == hipSuccess) {
        if (n > 1 && want_copies && want_copies(total)) {
            copies = 8;
            strideOut = total + 1;
        }
        const size_t nOut = copies == 8 ? (size_t)8 * strideOut : (size_t)total;
        e = hipMalloc((void**)&outEntries, nOut * 32);
    }
    if (e == hipSuccess) {
        EmitArgs ea{t, pbox, d_refs, outEntries, outRefs, copies, total, copies == 8 ? strideOut : total};
        emitEntries<<<blocks, 256, 0, stream>>>(ea);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(ev1, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e == hipSuccess && build_ms) e = hipEventElapsedTime(build_ms, ev0, ev1);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    for (void* p : temp) (void)hipFree(p);
    if (e != hipSuccess) {
        if (outEntries) (void)hipFree(outEntries);
        if (outRefs) (void)hipFree(outRefs);
        return e;
    }
    *entries = outEntries;
    *out_refs = outRefs;
    *n_entries = total;
    *stride = strideOut;
    return hipSuccess;
}

}  // namespace rt
