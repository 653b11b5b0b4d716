// HIP kernels of the path tracer core for gfx950 (wave64).
//
// The unit of work is one pixel sample (one camera path): persistent waves pull units from a
// launch-wide queue, every lane runs a small state machine over its current path, and a second kernel
// adds each pixel's radiances in sample order, which is the order of the reference's `pixelColor +=`
// loop (R/src/Core/Raytracer.cpp:84-90), so no atomics touch the image.  Arithmetic follows the
// reference expression by expression (build with -ffp-contract=off; fp32 divide and sqrt are
// correctly rounded).
// R/ = /root/reference/Raytracing-In-A-Weekend/.
#include "rt_device.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace rt {

constexpr uint32_t kNoBox = 1u << 30;
constexpr uint32_t kNodeInstance = 1u << 29;
constexpr uint32_t kSkipMask = kNodeInstance - 1;
constexpr uint32_t kRefLast = 1u << 31;
constexpr uint32_t kRefTri = 1u << 30;
constexpr uint32_t kRefId = kRefTri - 1;
constexpr uint32_t kNoHit = 0xFFFFFFFFu;
constexpr float kTMin = 0.001f;  // R/src/Core/Raytracer.cpp:10

// What a scene can hold, as far as the code compiled for it goes (template parameter HAS of the render kernels
// and their helpers): code for what a scene does not hold is left out, and with it the registers it would occupy
// in every lane through every operation.
enum : int {
    S_SPHERES = 1,  // Sphere primitives
    S_TRIS = 2,     // Triangle primitives
    S_PBR = 4,      // PBRMaterial (textures)
    S_DIEL = 8,     // Dielectric (its Schlick term runs in double)
    S_ALL = 15,
    S_PLAIN = S_SPHERES | S_DIEL,  // the cover scene's kind: spheres with Lambertian / Metal / Dielectric / DiffuseLight
    S_MESH = S_TRIS | S_PBR        // the Cornell box + vase kind: triangles with Lambertian / Metal / DiffuseLight / PBR
};
template <int HAS>
__device__ __forceinline__ bool isTriangleRef(uint32_t ref) {
    return (HAS & S_TRIS) && (!(HAS & S_SPHERES) || (ref & (1u << 30)));
}

__device__ __forceinline__ uint32_t fbits(float f) { return __float_as_uint(f); }

// R/assets/shaders/raytracer.comp:35-42
__device__ __forceinline__ uint32_t hashu(uint32_t x) {
    x += (x << 10u);
    x ^= (x >> 6u);
    x += (x << 3u);
    x ^= (x >> 11u);
    x += (x << 15u);
    return x;
}

// R/assets/shaders/raytracer.comp:44-58.  The shader hashes
// floatBitsToUint(vec4(x, y, sampleCount, counter)) as v.x ^ hash(v.y) ^ hash(v.z) ^ hash(v.w);
// the first three terms are constant for one pixel-sample and are folded into `key`.
//
// The remaining term hash(floatBits(counter)) depends on the counter alone; every kernel that
// draws random numbers tabulates it for the first kCounterLut counter values in LDS
// (fillCounterLut) so that a draw costs one hash instead of two.
constexpr int kCounterLut = 256;
__shared__ uint32_t g_counterLut[kCounterLut];
__shared__ int g_cancelSeen;  // set by the workgroup's first wave once it has read the caller's cancel flag as up

// stbi_loadf's 8-bit -> float map (DevScene::lut), staged by the kernels that sample textures: a texel fetch is
// followed by three look-ups, which then cost an LDS read each instead of a trip to L1 / L2
__shared__ float g_texLut[256];

// Instancing in the persistent loop: a lane's world-space ray (origin, direction) while the lane walks a prototype
// in object space, one slot per thread (only the build for scenes with instances refers to it)
__shared__ float g_worldRay[6][WF_GLOBAL_THREADS];
// Scenes in global memory: the part of a lane's state that only SHADE and NEWPATH touch -- throughput, unit, depth
// left -- lives here, read and written where it is used, instead of in registers carried through the walk and
// through the scatter code.
__shared__ float g_parked[5][WF_GLOBAL_THREADS];
// ... and what a wave knows about its current chunk (sample, corner pixel; -1: the tile lies outside the image)
__shared__ int g_chunkAt[WF_GLOBAL_THREADS / 64][4];
#ifndef RT_PARK
#define RT_PARK 1
#endif

template <bool TEX = false>
__device__ __forceinline__ void fillCounterLut(const float* texLut = nullptr) {
    for (int i = threadIdx.x; i < kCounterLut; i += blockDim.x) g_counterLut[i] = hashu(fbits((float)i));
    if (TEX)
        for (int i = threadIdx.x; i < 256; i += blockDim.x) g_texLut[i] = texLut[i];
    if (threadIdx.x == 0) g_cancelSeen = 0;
    __syncthreads();
}

struct Rng {
    uint32_t key, counter;
    __device__ __forceinline__ void init(uint32_t px, uint32_t py, int sample) {
        key = fbits((float)px) ^ hashu(fbits((float)py)) ^ hashu(fbits((float)sample));
        counter = 0;
    }
    __device__ __forceinline__ float next() {
        counter += 1;
        const uint32_t inner = counter < (uint32_t)kCounterLut ? g_counterLut[counter] : hashu(fbits((float)counter));
        uint32_t m = hashu(key ^ inner);
        m &= 0x007FFFFFu;
        m |= 0x3F800000u;
        return __uint_as_float(m) - 1.0f;
    }
    // randomFloat(min, max), R/src/Core/RTWeekend.h:33-36
    __device__ __forceinline__ float next(float mn, float mx) { return mn + (mx - mn) * next(); }
    // randomFloat(-1, 1) = -1 + 2 * (f - 1) with f = the mantissa trick's float in [1, 2).  f - 1 is exact, its
    // double is exact, and the sum is a multiple of 2^-22 below 1 in magnitude, hence exact as well: the single
    // rounding of the fused 2 f - 3 returns the same bits, in one instruction instead of three.
    __device__ __forceinline__ float signedFrom(uint32_t inner) const {
        uint32_t m = hashu(key ^ inner);
        m &= 0x007FFFFFu;
        m |= 0x3F800000u;
        return __builtin_fmaf(2.0f, __uint_as_float(m), -3.0f);
    }
    __device__ __forceinline__ uint32_t innerOf(uint32_t c) const { return hashu(fbits((float)c)); }
    // n consecutive draws of randomFloat(-1, 1), first draw first; one look at the counter for all of them
    __device__ __forceinline__ void nextSigned2(float& a, float& b) {
        const uint32_t c = counter;
        counter += 2;
        uint32_t i1, i2;
        if (c + 2 < (uint32_t)kCounterLut) {
            i1 = g_counterLut[c + 1];
            i2 = g_counterLut[c + 2];
        } else {
            i1 = innerOf(c + 1);
            i2 = innerOf(c + 2);
        }
        a = signedFrom(i1);
        b = signedFrom(i2);
    }
    __device__ __forceinline__ void nextSigned3(float& a, float& b, float& cc) {
        const uint32_t c = counter;
        counter += 3;
        uint32_t i1, i2, i3;
        if (c + 3 < (uint32_t)kCounterLut) {
            i1 = g_counterLut[c + 1];
            i2 = g_counterLut[c + 2];
            i3 = g_counterLut[c + 3];
        } else {
            i1 = innerOf(c + 1);
            i2 = innerOf(c + 2);
            i3 = innerOf(c + 3);
        }
        a = signedFrom(i1);
        b = signedFrom(i2);
        cc = signedFrom(i3);
    }
};

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ V3 normalize(V3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
__device__ __forceinline__ V3 reflect(V3 I, V3 N) { return I - N * dot(N, I) * 2.0f; }

// randomVecInUnitSphere, R/src/Core/RTWeekend.cpp:3-12 (z, y, x draw order of the g++ build)
__device__ __forceinline__ V3 randomInUnitSphere(Rng& rng) {
    for (;;) {
        float x, y, z;
        rng.nextSigned3(z, y, x);
        V3 p{x, y, z};
        if (dot(p, p) >= 1.0f) continue;
        return p;
    }
}
// Scene arrays as the traversal sees them: either straight from HBM/L2 or staged in LDS.
struct SceneView {
    const float4* nodes;    // reference-order entries
    const uint32_t* refs;
    const float4* fnodes;   // search structure traversed first
    const uint32_t* frefs;
    const float4* spheres;
    const float4* tris;
    const int* sphere_mat;
    const int* tri_mat;
    const float4* materials;
    const int* sphere_ord;
    const int* tri_ord;
    const int* prim_guard;
    const int* node_outer;
    const char* wnodes;     // LDS only: the search structure as 40-byte walk entries (see stageScene)
};

__device__ __forceinline__ SceneView globalView(const DevScene& sc) {
    return SceneView{sc.nodes,      sc.prim_refs, sc.fast_nodes, sc.fast_refs,  sc.spheres,    sc.tri_isect, sc.sphere_mat,
                     sc.tri_mat,    sc.materials, sc.sphere_ord, sc.tri_ord,    sc.prim_guard, sc.node_outer, nullptr};
}

constexpr int kNumStats = 14;
struct Stats {
    unsigned long long v[kNumStats];
};
enum { C_SAMPLES, C_RAYS, C_NODE, C_SPHERE, C_TRI, C_HITS, C_LAMB, C_METAL, C_DIEL, C_LIGHT, C_PBR, C_MISS, C_CUTOFF, C_FALLBACK };

struct Ray {
    V3 o, d;
};

struct Hit {
    float t;
    uint32_t ref;  // kNoHit or the leaf reference (triangle flag + id)
    float bu, bv;  // barycentrics of the accepted triangle hit
};

// AABB::Hit (R/src/AccelerationStructures/AABB.h:10-24) with tMin = 0.001 and the running
// closest hit as tMax.  `t0 > tMin ? t0 : tMin` keeps tMin when t0 is NaN, which is fmaxf's
// behaviour too (tMin and tMax themselves are never NaN), and the per-axis early exits are
// equivalent to one test at the end because tMin only grows and tMax only shrinks.
__device__ __forceinline__ bool slabPass(const float4& n0, const float4& n1, const Ray& r, float invx, float invy,
                                         float invz, float closest) {
    float t0 = (n0.x - r.o.x) * invx, t1 = (n1.x - r.o.x) * invx;
    if (invx < 0.0f) { float s = t0; t0 = t1; t1 = s; }
    float tmin = fmaxf(t0, kTMin), tmax = fminf(t1, closest);
    t0 = (n0.y - r.o.y) * invy; t1 = (n1.y - r.o.y) * invy;
    if (invy < 0.0f) { float s = t0; t0 = t1; t1 = s; }
    tmin = fmaxf(t0, tmin); tmax = fminf(t1, tmax);
    t0 = (n0.z - r.o.z) * invz; t1 = (n1.z - r.o.z) * invz;
    if (invz < 0.0f) { float s = t0; t0 = t1; t1 = s; }
    tmin = fmaxf(t0, tmin); tmax = fminf(t1, tmax);
    return !(tmax <= tmin);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((aligned(4))) f32x2u {  // two floats at a 4-byte aligned address
    float x, y;
};

// Box test of the search structure.  It only has to be conservative (never cull a box that
// holds an accepted hit): which primitive wins is decided by the exact primitive tests and the
// guard test afterwards, and the boxes are padded far beyond the rounding error of this
// arithmetic.  So it is written for speed: t = plane * (1/d) - o * (1/d) as one fused multiply-add
// per plane (two planes per packed instruction; the entry stores {min, max} pairs per axis), and
// min/max instead of compare-and-swap.  Rays with a zero or denormal direction component never
// get here (they walk the reference-order array).
// `slack`: the ray's own widening of every box in units of distance along the ray (boxSlack): the test passes
// when entry and exit distance are no further apart than that.
template <bool WIDE = true>
__device__ __forceinline__ bool slabPassFast(const float4& a, const float4& b, float invx, float invy, float invz,
                                             float nox, float noy, float noz, float closest, float slack) {
    const f32x2 tx = __builtin_elementwise_fma(f32x2{a.x, a.y}, f32x2{invx, invx}, f32x2{nox, nox});
    const f32x2 ty = __builtin_elementwise_fma(f32x2{a.z, a.w}, f32x2{invy, invy}, f32x2{noy, noy});
    const f32x2 tz = __builtin_elementwise_fma(f32x2{b.x, b.y}, f32x2{invz, invz}, f32x2{noz, noz});
    const float tn = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fmaxf(fminf(tz.x, tz.y), kTMin));
    const float tf = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fminf(fmaxf(tz.x, tz.y), closest));
    return WIDE ? tn <= tf + slack : tn <= tf;
}
// The same test on an entry of a direction-ordered copy, whose planes are stored {entry, exit} per axis for the
// sign pattern of the rays that walk it: nothing to sort.
__device__ __forceinline__ bool slabPassSorted(const float4& a, const float4& b, float invx, float invy, float invz,
                                               float nox, float noy, float noz, float closest) {
    const f32x2 tx = __builtin_elementwise_fma(f32x2{a.x, a.y}, f32x2{invx, invx}, f32x2{nox, nox});
    const f32x2 ty = __builtin_elementwise_fma(f32x2{a.z, a.w}, f32x2{invy, invy}, f32x2{noy, noy});
    const f32x2 tz = __builtin_elementwise_fma(f32x2{b.x, b.y}, f32x2{invz, invz}, f32x2{noz, noz});
    const float tn = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, kTMin));
    const float tf = fminf(fminf(tx.y, ty.y), fminf(tz.y, closest));
    return tn <= tf;
}
// A box widened by e on every side is entered at most e * |1/d| earlier and left at most e * |1/d| later on
// each axis; with the largest of the three factors for all axes the widened test is implied by
// "entry <= exit + 2 e max|1/d|" (the clamps to tMin and to the closest hit only make that easier to meet).
__device__ __forceinline__ float boxSlack(float epad, float invx, float invy, float invz) {
    return 2.0f * epad * fmaxf(fmaxf(fabsf(invx), fabsf(invy)), fabsf(invz));
}

// By how much a ray that starts at `o` widens every box of the search structure.  Sphere::Hit and
// Triangle::Hit (R/src/Core/Hittable.h:228-255, 58-141) lose accuracy with the distance D between ray
// origin and primitive: the sphere's discriminant is a difference of terms of size D^2, so a ray that
// passes a sphere of radius r at up to sqrt(r^2 + 32 * 2^-24 * D^2) from its centre can still be given a
// hit, and the reference returns it; a triangle's barycentrics are off by about 64 * 2^-24 * D.  The
// boxes are padded for D = far_typical (the eye and the scene's own primitives are within that); a ray
// that starts further from the primitives concerned (bounded by far_center / far_radius; far_rmin = the
// smallest radius, which needs most) adds the difference.  Zero nearly always.
// True if farRayPad(o) is not zero (a handful of instructions, for every ray).
__device__ __forceinline__ bool startsFar(const DevScene& sc, const V3& o) {
    const float dx = o.x - sc.far_center[0], dy = o.y - sc.far_center[1], dz = o.z - sc.far_center[2];
    return (dx * dx + dy * dy) + dz * dz > sc.far_trigger2;
}
// (The square roots are the hardware's, good to one unit in the last place: this is a bound with 1e-4 of slack
// in D and 2 % in the result, not part of the image's arithmetic, and a correctly rounded sqrtf() is a dozen
// instructions more per ray.)
__device__ __forceinline__ float farRayPad(const DevScene& sc, const V3& o) {
    if ((sc.far_spheres | sc.far_tris) == 0) return 0.0f;
    const float dx = o.x - sc.far_center[0], dy = o.y - sc.far_center[1], dz = o.z - sc.far_center[2];
    const float D = (__builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz) + sc.far_radius) * 1.0001f;
    if (!(D > sc.far_typical)) return 0.0f;  // also for an origin that is not finite: such rays take the other walk
    const float kS = 32.0f / 16777216.0f, kT = 64.0f / 16777216.0f, r = sc.far_rmin, T = sc.far_typical;
    float e = 0.0f;
    if (sc.far_spheres) e = __builtin_amdgcn_sqrtf(r * r + kS * D * D) - __builtin_amdgcn_sqrtf(r * r + kS * T * T);
    if (sc.far_tris) e = fmaxf(e, kT * (D - T));
    return e * 1.02f + 1e-7f;
}

// Sphere::Hit / Triangle::Hit (R/src/Core/Hittable.h:228-255, 58-141) for one leaf reference:
// true with the hit distance (and a triangle's barycentrics) if the primitive is hit within
// [0.001, tMax].  `tMax < t` rejects, equality accepts, as in the reference.
template <bool COUNT, int HAS = S_ALL>
__device__ __forceinline__ bool primRoot(const SceneView& sv, uint32_t ref, const Ray& r, float a, float tMax, float& tOut,
                                         float& uOut, float& vOut, Stats* st) {
    const uint32_t id = ref & kRefId;
    if (isTriangleRef<HAS>(ref)) {
        // edges are precomputed on the host with the reference's two subtractions
        if (COUNT) st->v[C_TRI]++;
        const float4 q0 = sv.tris[3 * id], q1 = sv.tris[3 * id + 1], q2 = sv.tris[3 * id + 2];
        const V3 v0{q0.x, q0.y, q0.z}, e1{q1.x, q1.y, q1.z}, e2{q2.x, q2.y, q2.z};
        const float EPS = 1e-8f;
        V3 hh = cross(r.d, e2);
        float det = dot(e1, hh);
        if (det > -EPS && det < EPS) return false;
        float f = 1.0f / det;
        V3 s = r.o - v0;
        float u = f * dot(s, hh);
        if (u < 0.0f || u > 1.0f) return false;
        V3 q = cross(s, e1);
        float v = f * dot(r.d, q);
        if (v < 0.0f || u + v > 1.0f) return false;
        float t = f * dot(e2, q);
        if (t < kTMin || tMax < t) return false;
        if (!(t > EPS)) return false;
        tOut = t;
        uOut = u;
        vOut = v;
        return true;
    }
    if (!(HAS & S_SPHERES)) return false;
    if (COUNT) st->v[C_SPHERE]++;
    const float4 s = sv.spheres[id];
    V3 oc{r.o.x - s.x, r.o.y - s.y, r.o.z - s.z};
    float halfB = dot(oc, r.d);
    float c = dot(oc, oc) - s.w * s.w;
    float disc = halfB * halfB - a * c;
    if (disc < 0) return false;
    float sq = sqrtf(disc);
    float root = (-halfB - sq) / a;
    if (root < kTMin || tMax < root) {
        root = (-halfB + sq) / a;
        if (root < kTMin || tMax < root) return false;
    }
    tOut = root;
    return true;
}

// The reference's own update rule: primitives arrive in its visit order and each one is
// tested against the running closest hit.
template <bool COUNT>
__device__ __forceinline__ bool primRef(const SceneView& sv, uint32_t ref, const Ray& r, float a, Hit& h, Stats* st) {
    float t, u = h.bu, v = h.bv;
    if (primRoot<COUNT>(sv, ref, r, a, h.t, t, u, v, st)) {
        h.t = t;
        h.ref = ref & ~kRefLast;
        h.bu = u;
        h.bv = v;
        return true;
    }
    return false;
}

template <int HAS = S_ALL>
__device__ __forceinline__ int refOrder(const SceneView& sv, uint32_t ref) {
    return isTriangleRef<HAS>(ref) ? sv.tri_ord[ref & kRefId] : sv.sphere_ord[ref & kRefId];
}

// Update rule for primitives that arrive in any order: smallest hit distance wins, and of two
// primitives with exactly the same distance the one the reference visits later (the one its
// `tMax < t` test lets overwrite the other).  A NaN distance cannot be ordered: `bad` is set and
// the ray is traced again in the reference's order.
template <bool COUNT>
__device__ __forceinline__ void primAnyOrder(const SceneView& sv, uint32_t ref, const Ray& r, float a, Hit& h, bool& bad,
                                             Stats* st) {
    float t, u = 0.0f, v = 0.0f;
    if (primRoot<COUNT>(sv, ref, r, a, __builtin_inff(), t, u, v, st)) {
        if (t != t) {
            bad = true;
        } else if (h.ref == kNoHit || t < h.t || (t == h.t && refOrder(sv, ref) > refOrder(sv, h.ref))) {
            h.t = t;
            h.ref = ref & ~kRefLast;
            h.bu = u;
            h.bv = v;
        }
    }
}

// Closest hit along the threaded, reference-ordered node array.  Entry i: box test with the
// running closest distance as tMax (BVHNode::Hit passes rec.t of the left child on to the right
// one, R/src/AccelerationStructures/Bvh.h:84-85, which is the running closest hit); on a miss
// jump to `skip`, else test the entry's leaf run and go to i + 1.
template <bool COUNT>
__device__ __forceinline__ Hit traverseRef(const SceneView& sv, int n_nodes, const Ray& r, float invx, float invy, float invz,
                                           float a, Stats* st) {
    Hit h{__builtin_inff(), kNoHit, 0.0f, 0.0f};
    int idx = 0;
    while (idx < n_nodes) {
        const float4 n0 = sv.nodes[2 * idx], n1 = sv.nodes[2 * idx + 1];
        const uint32_t w3 = fbits(n0.w);
        const int first = (int)fbits(n1.w);
        bool pass = true;
        if (!(w3 & kNoBox)) {
            if (COUNT) st->v[C_NODE]++;
            pass = slabPass(n0, n1, r, invx, invy, invz, h.t);
        }
        if (pass) {
            if (first >= 0) {
                int p = first;
                uint32_t ref;
                do {
                    ref = sv.refs[p++];
                    primRef<COUNT>(sv, ref, r, a, h, st);
                } while (!(ref & kRefLast));
            }
            idx = idx + 1;
        } else {
            idx = (int)(w3 & kSkipMask);
        }
    }
    return h;
}

// ------------------------------------------------------------------ instancing (opt-in): plain kernel and trace hook
// The reference has no instances: it bakes every placement into world-space triangles
// (R/src/Core/Mesh.cpp:106-208).  What an instance means here is defined by the CPU oracle's Instance class,
// which these functions follow expression by expression.
// The ray in the object space of an instance: origin as a point, direction as a vector (not normalised, so a hit
// distance is the same number in both spaces).  I = the instance's five float4 (DevScene::instances).
__device__ __forceinline__ Ray toObjectSpace(const float4* I, const Ray& r) {
    const float4 c0 = I[0], c1 = I[1], c2 = I[2], c3 = I[3];
    Ray o;
    o.o = {(c0.x * r.o.x + c1.x * r.o.y) + (c2.x * r.o.z + c3.x * 1.0f), (c0.y * r.o.x + c1.y * r.o.y) + (c2.y * r.o.z + c3.y * 1.0f),
           (c0.z * r.o.x + c1.z * r.o.y) + (c2.z * r.o.z + c3.z * 1.0f)};
    o.d = {(c0.x * r.d.x + c1.x * r.d.y) + c2.x * r.d.z, (c0.y * r.d.x + c1.y * r.d.y) + c2.y * r.d.z,
           (c0.z * r.d.x + c1.z * r.d.y) + c2.z * r.d.z};
    return o;
}

// Entries [begin, end) of the reference-order array (a prototype's), walked like traverseRef() with the running
// closest hit of the caller.  True if a primitive was accepted.
template <bool COUNT>
__device__ __forceinline__ bool walkEntryRange(const SceneView& sv, int begin, int end, const Ray& r, float invx, float invy,
                                               float invz, float a, Hit& h, Stats* st) {
    bool accepted = false;
    int idx = begin;
    while (idx < end) {
        const float4 n0 = sv.nodes[2 * idx], n1 = sv.nodes[2 * idx + 1];
        const uint32_t w3 = fbits(n0.w);
        const int first = (int)fbits(n1.w);
        bool pass = true;
        if (!(w3 & kNoBox)) {
            if (COUNT) st->v[C_NODE]++;
            pass = slabPass(n0, n1, r, invx, invy, invz, h.t);
        }
        if (pass) {
            if (first >= 0) {
                int p = first;
                uint32_t ref;
                do {
                    ref = sv.refs[p++];
                    accepted = primRef<COUNT>(sv, ref, r, a, h, st) || accepted;
                } while (!(ref & kRefLast));
            }
            idx = idx + 1;
        } else {
            idx = (int)(w3 & kSkipMask);
        }
    }
    return accepted;
}

// mWorld.Hit() for a world that holds instances.  instOut: the instance the closest hit was found through, or -1.
template <bool COUNT>
__device__ __forceinline__ Hit traverseInstanced(const DevScene& sc, const SceneView& sv, const Ray& r, float invx, float invy,
                                                 float invz, float a, int& instOut, Stats* st) {
    Hit h{__builtin_inff(), kNoHit, 0.0f, 0.0f};
    instOut = -1;
    int idx = 0;
    while (idx < sc.n_world_nodes) {
        const float4 n0 = sv.nodes[2 * idx], n1 = sv.nodes[2 * idx + 1];
        const uint32_t w3 = fbits(n0.w);
        const int first = (int)fbits(n1.w);
        if (w3 & kNodeInstance) {  // Instance::Hit: its box, then the prototype in object space
            if (COUNT) st->v[C_NODE]++;
            if (slabPass(n0, n1, r, invx, invy, invz, h.t)) {
                const float4* I = sc.instances + 5 * first;
                const Ray ro = toObjectSpace(I, r);
                const float4 range = I[4];
                if (walkEntryRange<COUNT>(sv, (int)fbits(range.x), (int)fbits(range.y), ro, 1.0f / ro.d.x, 1.0f / ro.d.y,
                                          1.0f / ro.d.z, dot(ro.d, ro.d), h, st))
                    instOut = first;
            }
            idx = idx + 1;
            continue;
        }
        bool pass = true;
        if (!(w3 & kNoBox)) {
            if (COUNT) st->v[C_NODE]++;
            pass = slabPass(n0, n1, r, invx, invy, invz, h.t);
        }
        if (pass) {
            if (first >= 0) {
                int p = first;
                uint32_t ref;
                do {
                    ref = sv.refs[p++];
                    if (primRef<COUNT>(sv, ref, r, a, h, st)) instOut = -1;
                } while (!(ref & kRefLast));
            }
            idx = idx + 1;
        } else {
            idx = (int)(w3 & kSkipMask);
        }
    }
    return h;
}

// The same walk over the search structure: tighter tree, padded boxes, conservative box test,
// order-independent update.
template <bool COUNT>
__device__ __forceinline__ Hit traverseFast(const SceneView& sv, int n_nodes, const Ray& r, float invx, float invy, float invz,
                                            float a, float epad, bool& bad, Stats* st) {
    Hit h{__builtin_inff(), kNoHit, 0.0f, 0.0f};
    const float nox = -(r.o.x * invx), noy = -(r.o.y * invy), noz = -(r.o.z * invz);
    const float slack = boxSlack(epad, invx, invy, invz);
    int idx = 0;
    while (idx < n_nodes) {
        const float4 n0 = sv.fnodes[2 * idx], n1 = sv.fnodes[2 * idx + 1];
        const int first = (int)fbits(n1.w);
        if (COUNT) st->v[C_NODE]++;
        if (slabPassFast(n0, n1, invx, invy, invz, nox, noy, noz, h.t, slack)) {
            if (first >= 0) {
                int p = first;
                uint32_t ref;
                do {
                    ref = sv.frefs[p++];
                    primAnyOrder<COUNT>(sv, ref, r, a, h, bad, st);
                } while (!(ref & kRefLast));
            }
            idx = idx + 1;
        } else {
            idx = (int)fbits(n1.z);
        }
    }
    return h;
}

// The condition under which the reference's walk returns exactly the hit found by the search
// structure: the innermost reference box above the winner, and every enclosing box that does
// not contain it, pass the reference's slab test with tMax = the winner's distance.
template <bool COUNT, int HAS = S_ALL>
__device__ __forceinline__ bool guardsPass(const SceneView& sv, const Ray& r, float invx, float invy, float invz, const Hit& h,
                                           Stats* st) {
    int g = sv.prim_guard[refOrder<HAS>(sv, h.ref)];
    while (g >= 0) {
        if (COUNT) st->v[C_NODE]++;
        if (!slabPass(sv.nodes[2 * g], sv.nodes[2 * g + 1], r, invx, invy, invz, h.t)) return false;
        g = sv.node_outer[g];
    }
    return true;
}

// True if the search structure's fused box test may be used for this ray: every component of the
// reciprocal direction finite and small enough that plane * (1/d) cannot overflow (scene coordinates
// are limited to 1e7 on upload), and the origin within the distance from the coordinate origin up to
// which the boxes' padding covers the test's rounding error (which grows with |origin|).  False (also
// for NaN) sends the ray to the reference-order walk.
__device__ __forceinline__ bool fastTestUsable(const V3& o, float originLimit, float invx, float invy, float invz, float nox,
                                               float noy, float noz) {
    const float lim = 1e30f, inf = __builtin_inff();
    return fabsf(invx) < lim && fabsf(invy) < lim && fabsf(invz) < lim && fabsf(nox) < inf && fabsf(noy) < inf &&
           fabsf(noz) < inf && fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) <= originLimit;
}

// mWorld.Hit(r, 0.001f, infinity, rec) (R/src/Core/Raytracer.cpp:10): search structure first,
// the reference's own walk when its answer cannot be vouched for.
// inst: the instance the hit was found through, or -1 (scenes with instances only).
template <bool COUNT>
__device__ __forceinline__ Hit closestHit(const DevScene& sc, const SceneView& sv, const Ray& r, Stats* st, int& inst) {
    const float invx = 1.0f / r.d.x, invy = 1.0f / r.d.y, invz = 1.0f / r.d.z;
    const float a = dot(r.d, r.d);
    inst = -1;
    if (sc.n_instances > 0) return traverseInstanced<COUNT>(sc, sv, r, invx, invy, invz, a, inst, st);
    if (!sc.force_ref &&
        fastTestUsable(r.o, sc.origin_limit, invx, invy, invz, -(r.o.x * invx), -(r.o.y * invy), -(r.o.z * invz))) {
        bool bad = false;
        Hit h = traverseFast<COUNT>(sv, sc.n_fast_nodes, r, invx, invy, invz, a, startsFar(sc, r.o) ? farRayPad(sc, r.o) : 0.0f, bad, st);
        if (!bad && (h.ref == kNoHit || guardsPass<COUNT>(sv, r, invx, invy, invz, h, st))) return h;
        if (COUNT) st->v[C_FALLBACK]++;
    }
    return traverseRef<COUNT>(sv, sc.n_nodes, r, invx, invy, invz, a, st);
}

struct Surface {
    V3 p, n;
    bool front;
    float u, v;
    int mat;
    int tri;  // triangle id, -1 for a sphere
};

// The HitRecord fields of the closest hit, computed once after traversal from (primitive, t):
// same expressions as Sphere::Hit (Hittable.h:248-252) and Triangle::Hit (Hittable.h:86-107).
template <int HAS = S_ALL>
__device__ __forceinline__ Surface surfaceAt(const DevScene& sc, const SceneView& sv, const Ray& r, const Hit& h) {
    Surface s;
    s.p = r.o + h.t * r.d;
    s.u = 0.0f;
    s.v = 0.0f;
    s.front = false;
    const uint32_t id = h.ref & kRefId;
    if (isTriangleRef<HAS>(h.ref)) {
        const float4 q1 = sv.tris[3 * id + 1], q2 = sv.tris[3 * id + 2];
        const float4 a0 = sc.tri_shade[4 * id], a1 = sc.tri_shade[4 * id + 1], a2 = sc.tri_shade[4 * id + 2],
                     a3 = sc.tri_shade[4 * id + 3];
        const V3 n0{a0.x, a0.y, a0.z}, n1{a1.x, a1.y, a1.z}, n2{a2.x, a2.y, a2.z};
        const float u = h.bu, v = h.bv;
        if (n0.x == 0.0f && n0.y == 0.0f && n0.z == 0.0f) {
            V3 e1{q1.x, q1.y, q1.z}, e2{q2.x, q2.y, q2.z};
            V3 outward = normalize(cross(e2, e1));
            s.front = dot(r.d, outward) < 0;
            s.n = s.front ? outward : -outward;
        } else {
            // u weighs vertex 0, v vertex 1, (1-u-v) vertex 2, as the reference does
            s.n = normalize(u * n0 + v * n1 + (1 - u - v) * n2);
        }
        const float uv0x = a0.w, uv0y = a1.w, uv1x = a2.w, uv1y = a3.x, uv2x = a3.y, uv2y = a3.z;
        if (uv0x != -1.0f) {
            s.u = (u * uv0x + v * uv1x) + (1 - u - v) * uv2x;
            s.v = (u * uv0y + v * uv1y) + (1 - u - v) * uv2y;
        } else {
            s.u = u;
            s.v = v;
        }
        s.mat = sv.tri_mat[id];
        s.tri = (int)id;
    } else {
        const float4 sp = sv.spheres[id];
        V3 c{sp.x, sp.y, sp.z};
        V3 outward = (s.p - c) / sp.w;
        s.front = dot(r.d, outward) < 0;
        s.n = s.front ? outward : -outward;
        s.mat = sv.sphere_mat[id];
        s.tri = -1;
    }
    return s;
}

// The hit record of a hit found through an instance (inst >= 0): formed in object space, then the point taken from
// the world ray and the normal brought back through mat3(transpose(inverse(M))) and normalised (the oracle's
// Instance::Hit).  Column j of that matrix is row j of the inverse, i.e. its rows are the inverse's columns.
__device__ __forceinline__ Surface surfaceAtInstanced(const DevScene& sc, const SceneView& sv, const Ray& r, const Hit& h, int inst) {
    if (inst < 0) return surfaceAt(sc, sv, r, h);
    const float4* I = sc.instances + 5 * inst;
    const Ray ro = toObjectSpace(I, r);
    Surface s = surfaceAt(sc, sv, ro, h);
    s.p = r.o + h.t * r.d;
    const float4 c0 = I[0], c1 = I[1], c2 = I[2];
    const V3 n = s.n;
    s.n = normalize(V3{(c0.x * n.x + c0.y * n.y) + c0.z * n.z, (c1.x * n.x + c1.y * n.y) + c1.z * n.z,
                       (c2.x * n.x + c2.y * n.y) + c2.z * n.z});
    return s;
}

// Texture::At, R/src/Material/Texture.h:37-43: nearest texel, row stride = height, and the
// 8-bit -> float map of stbi_loadf.  Out-of-range indices are clamped (the reference reads
// out of bounds there).
__device__ __forceinline__ uint32_t texelAt(const DevScene& sc, int tex, float u, float v) {
    const int4 d = sc.tex_desc[tex];
    int tx = (int)((u * d.y) - 0.5f);
    int ty = (int)(((1 - v) * d.z) - 0.5f);
    long long idx = (long long)ty * d.z + tx;
    const long long n = (long long)d.y * d.z;
    idx = idx < 0 ? 0 : (idx >= n ? n - 1 : idx);
    return sc.texels[(size_t)d.x + (size_t)idx];
}
// STAGED: the kernel has the map in LDS (fillCounterLut with the scene's table)
template <bool STAGED = false>
__device__ __forceinline__ V3 texelColour(const DevScene& sc, uint32_t px) {
    if (STAGED) return {g_texLut[px & 255u], g_texLut[(px >> 8) & 255u], g_texLut[(px >> 16) & 255u]};
    return {sc.lut[px & 255u], sc.lut[(px >> 8) & 255u], sc.lut[(px >> 16) & 255u]};
}
__device__ __forceinline__ V3 textureAt(const DevScene& sc, int tex, float u, float v) {
    return texelColour(sc, texelAt(sc, tex, u, v));
}

// The same texel as plain byte / 255: a normal map holds directions, not colours, and must not go
// through the gamma map of stbi_loadf (which is what the reference's Texture does to every file).
__device__ __forceinline__ V3 textureAtLinear(const DevScene& sc, int tex, float u, float v) {
    const int4 d = sc.tex_desc[tex];
    int tx = (int)((u * d.y) - 0.5f);
    int ty = (int)(((1 - v) * d.z) - 0.5f);
    long long idx = (long long)ty * d.z + tx;
    const long long n = (long long)d.y * d.z;
    idx = idx < 0 ? 0 : (idx >= n ? n - 1 : idx);
    const uint32_t px = sc.texels[(size_t)d.x + (size_t)idx];
    return {(float)(px & 255u) / 255.0f, (float)((px >> 8) & 255u) / 255.0f, (float)((px >> 16) & 255u) / 255.0f};
}

// vecNearZero, R/src/Core/RTWeekend.cpp:39-43: every |component| < 1e-8, compared as doubles there.
// The float nearest to 1e-8 lies below it (0x322BCC77 = 9.99999994e-9) and the next float lies above,
// so for a float x, (double)x < 1e-8 holds exactly when x <= that float.
__device__ __forceinline__ bool nearZero(V3 v) {
    const float lim = __uint_as_float(0x322BCC77u);
    return fabsf(v.x) <= lim && fabsf(v.y) <= lim && fabsf(v.z) <= lim;
}

// Dielectric::reflectance, R/src/Material/Material.h:90-95: r0 + (1 - r0) * pow(1 - cosine, 5), where
// pow(float, int) and the sum run in double and the result is returned as float.  The C library's
// pow is correctly rounded but for rare near-ties, so the fifth power is formed here to the same
// standard: x^2 is exact in double (x has 24 significant bits), and the rounding residuals of the
// two remaining products are recovered with fused multiply-adds and added back before the final
// rounding.
__device__ __forceinline__ float reflectance(float cosine, float refIdx) {
    float r0 = (1 - refIdx) / (1 + refIdx);
    r0 = r0 * r0;
    const double x = (double)(1 - cosine);
    const double x2 = x * x;                            // exact
    const double x4 = x2 * x2, e4 = __builtin_fma(x2, x2, -x4);  // x^4 = x4 + e4
    const double p = x4 * x, e5 = __builtin_fma(x4, x, -p);      // x4 * x = p + e5
    const double x5 = p + (e5 + e4 * x);
    return (float)((double)r0 + (double)(1 - r0) * x5);
}

// Material::scatter for the five material kinds (R/src/Material/Material.h:22-160).
// Returns true if the path continues with `out` and `atten`; false ends it with `emit`.
// HAS without S_PBR: no material of the scene is a PBRMaterial (texture code left out); without S_DIEL: none is a
// Dielectric.  NMAP = false: the scene
// has normal mapping switched off (the reference's behaviour), and the code for it is left out.
// Lambertian, Metal and PBR each draw exactly one point in the unit sphere and nothing before it, so the
// rejection loop runs once for all the lanes of a wave, whatever their material, and the caller runs it
// (`ball`; see drawsBall).  (Cut off after 1 / 2 / 3 / 4 candidates per SHADE execution, the lanes still without
// a point coming back with the next execution: 24.4 / 21.3 / 20.55 / 20.25 ms against 20.2 -- lanes that wait
// for another SHADE execution are missing from the box steps.)
template <int HAS = S_ALL>
__device__ __forceinline__ bool drawsBall(int kind) {
    return kind == 0 || kind == 1 || ((HAS & S_PBR) && kind == 4);
}
template <bool COUNT, int HAS = S_ALL, bool NMAP = true, bool STAGED = false>
__device__ __forceinline__ bool scatter(const DevScene& sc, const float4& m0, const float4& m1, const Ray& in,
                                        const Surface& s, const V3& ball, Rng& rng, Ray& out, V3& atten, V3& emit,
                                        Stats* st) {
    const int kind = (int)fbits(m1.x);
    emit = {0.0f, 0.0f, 0.0f};
    out.o = s.p;
    // likewise glm::normalize(rayIn.direction) of Metal, Dielectric and PBR
    V3 unitIn{0.0f, 0.0f, 0.0f};
    if (kind == 1 || ((HAS & S_DIEL) && kind == 2) || ((HAS & S_PBR) && kind == 4)) unitIn = normalize(in.d);
    if (kind == 0) {  // Lambertian, Material.h:22-36
        if (COUNT) st->v[C_LAMB]++;
        V3 dir = s.n + normalize(ball);
        if (nearZero(dir)) dir = s.n;
        out.d = dir;
        atten = {m0.x, m0.y, m0.z};
        return true;
    }
    if (kind == 1) {  // Metal, Material.h:47-53
        if (COUNT) st->v[C_METAL]++;
        V3 reflected = reflect(unitIn, s.n);
        out.d = reflected + m0.w * ball;
        atten = {m0.x, m0.y, m0.z};
        return dot(out.d, s.n) > 0;
    }
    if ((HAS & S_DIEL) && kind == 2) {  // Dielectric, Material.h:65-95
        if (COUNT) st->v[C_DIEL]++;
        atten = {1.0f, 1.0f, 1.0f};
        const float ir = m0.w;
        float ratio = s.front ? (1.0f / ir) : ir;
        V3 unit = unitIn;
        float cosTheta = fminf(dot(-unit, s.n), 1.0f);
        float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
        bool reflectIt = ratio * sinTheta > 1.0f;
        if (!reflectIt) reflectIt = reflectance(cosTheta, ratio) > rng.next();
        if (reflectIt) {
            out.d = reflect(unit, s.n);
        } else {  // refract, R/src/Core/RTWeekend.cpp:45-51
            float ct = fminf(dot(-unit, s.n), 1.0f);
            V3 perp = ratio * (unit + ct * s.n);
            V3 par = -sqrtf(fabsf(1.0f - dot(perp, perp))) * s.n;
            out.d = perp + par;
        }
        return true;
    }
    if (kind == 3 || !(HAS & S_PBR)) {  // DiffuseLight, Material.h:103-110
        if (COUNT) st->v[C_LIGHT]++;
        emit = {m0.x, m0.y, m0.z};
        return false;
    }
    // PBRMaterial, Material.h:121-160.  The reference forms the normal-mapped normal (:123-138) and then
    // overwrites it with rec.normal (:140), so by default only the diffuse and roughness maps influence the
    // path.  With normal mapping switched on for the scene the mapped normal is used instead:
    // normalize(TBN * (2 texel - 1)) with the triangle's unit tangent and bitangent and the shading normal.
    if (COUNT) st->v[C_PBR]++;
    const int texD = (int)fbits(m1.y), texR = (int)fbits(m1.z), texN = (int)fbits(m1.w);
    V3 nrm = s.n;
    if (NMAP && sc.normal_mapping && texN >= 0 && s.tri >= 0) {
        const float4 f0 = sc.tri_frame[2 * s.tri], f1 = sc.tri_frame[2 * s.tri + 1];
        if (f0.w != 0.0f) {
            const V3 T = normalize(V3{f0.x, f0.y, f0.z}), B = normalize(V3{f1.x, f1.y, f1.z}), N = normalize(s.n);
            const V3 c = textureAtLinear(sc, texN, s.u, s.v);
            const V3 m{c.x * 2.0f - 1.0f, c.y * 2.0f - 1.0f, c.z * 2.0f - 1.0f};
            nrm = normalize(V3{(T.x * m.x + B.x * m.y) + N.x * m.z, (T.y * m.x + B.y * m.y) + N.y * m.z,
                               (T.z * m.x + B.z * m.y) + N.z * m.z});
        }
    }
    // both texels are asked for before either is used: two trips to memory side by side instead of one after the other
    const uint32_t texelD = texelAt(sc, texD, s.u, s.v);
    const uint32_t texelR = texR >= 0 ? texelAt(sc, texR, s.u, s.v) : 0u;
    atten = texelColour<STAGED>(sc, texelD);
    if (texR >= 0) {
        V3 reflected = reflect(unitIn, nrm);
        V3 rough = texelColour<STAGED>(sc, texelR);
        out.d = reflected + rough * ball;
        return dot(out.d, nrm) > 0;
    }
    V3 dir = nrm + normalize(ball);
    if (nearZero(dir)) dir = nrm;
    out.d = dir;
    return true;
}

// Camera::GetRay, R/src/Core/Camera.h:27-33, with randomInUnitDisk (RTWeekend.cpp:29-37; y is
// drawn before x in the g++ build).  The disk is sampled even when lensRadius is 0.
__device__ __forceinline__ Ray cameraRay(const DevScene& sc, float s, float t, Rng& rng) {
    const float* c = sc.cam;
    float px, py;
    for (;;) {
        rng.nextSigned2(py, px);
        if ((px * px + py * py) + 0.0f * 0.0f >= 1) continue;
        break;
    }
    const float lr = c[21];
    const float rdx = lr * px, rdy = lr * py;
    const V3 org{c[0], c[1], c[2]}, llc{c[3], c[4], c[5]}, hor{c[6], c[7], c[8]}, ver{c[9], c[10], c[11]};
    const V3 cu{c[12], c[13], c[14]}, cv{c[15], c[16], c[17]};
    V3 offset = cu * rdx + cv * rdy;
    Ray r;
    r.o = org + offset;
    r.d = llc + s * hor + t * ver - org - offset;
    return r;
}

// rayColor, R/src/Core/Raytracer.cpp:3-20, unrolled into a loop.  The attenuation product is
// accumulated front to back (throughput) and multiplied into the terminal radiance; no
// reference material both emits and scatters, so this is the recursion's value up to the
// association order of the float products.
template <bool COUNT>
__device__ __forceinline__ V3 rayColor(const DevScene& sc, const SceneView& sv, Ray r, int depth, Rng& rng, Stats* st) {
    V3 throughput{1.0f, 1.0f, 1.0f};
    for (;;) {
        if (depth <= 0) {
            if (COUNT) st->v[C_CUTOFF]++;
            return throughput * V3{0.0f, 0.0f, 0.0f};
        }
        if (COUNT) st->v[C_RAYS]++;
        int inst;
        Hit h = closestHit<COUNT>(sc, sv, r, st, inst);
        if (h.ref == kNoHit) {
            if (COUNT) st->v[C_MISS]++;
            return throughput * V3{sc.bg[0], sc.bg[1], sc.bg[2]};
        }
        if (COUNT) st->v[C_HITS]++;
        Surface s = surfaceAtInstanced(sc, sv, r, h, inst);
        Ray out;
        V3 atten, emit;
        const float4 m0 = sv.materials[2 * s.mat], m1 = sv.materials[2 * s.mat + 1];
        V3 ball{0.0f, 0.0f, 0.0f};
        if (drawsBall((int)fbits(m1.x))) ball = randomInUnitSphere(rng);
        if (!scatter<COUNT>(sc, m0, m1, r, s, ball, rng, out, atten, emit, st)) return throughput * emit;
        throughput = throughput * atten;
        r = out;
        depth--;
    }
}

extern __shared__ float4 rt_lds[];

// Walk entry of the search structure in LDS (persistent kernel): ten dwords
//   {min.x, max.x, min.x,  min.y, max.y, min.y,  min.z, max.z, min.z,  first_ref << 16 | skip}
// A lane reads each axis as the two dwords that start at 0 (direction component positive) or 4 bytes
// (negative) into the triple, which hands it {entry plane, exit plane} with no comparison.  `skip` is
// the LDS address of the entry to go on with after a failed test; first_ref is 0xFFFF for an entry
// without a leaf run.  The table ends with an entry every ray passes whose first_ref is kWalkEnd:
// reaching it is how a lane learns that its walk is over.
constexpr int kWalkEntryBytes = 40;
constexpr uint32_t kWalkNoLeaf = 0xFFFFu, kWalkEnd = 0xFFFEu;

typedef __attribute__((address_space(3))) const char* LdsBytes;
__device__ __forceinline__ uint32_t ldsAddress(const void* p) { return (uint32_t)(uintptr_t)(LdsBytes)p; }

// Stages the scene tables into LDS and returns views into it (layout = ldsSceneBytes()).
__device__ __forceinline__ SceneView stageScene(const DevScene& sc, bool withReferenceArrays, bool walkEntries = false) {
    char* base = reinterpret_cast<char*>(rt_lds);
    int off = 0;
    const int tid = threadIdx.x, nt = blockDim.x;
    auto take16 = [&](const float4* src, int n) {
        float4* dst = reinterpret_cast<float4*>(base + off);
        off += (n * 16 + 15) & ~15;
        for (int i = tid; i < n; i += nt) dst[i] = src[i];
        return dst;
    };
    auto take4 = [&](const uint32_t* src, int n) {
        uint32_t* dst = reinterpret_cast<uint32_t*>(base + off);
        off += (n * 4 + 15) & ~15;
        for (int i = tid; i < n; i += nt) dst[i] = src[i];
        return dst;
    };
    auto take4i = [&](const int* src, int n) {
        return reinterpret_cast<const int*>(take4(reinterpret_cast<const uint32_t*>(src), n));
    };
    SceneView v;
    v.wnodes = nullptr;
    if (walkEntries) {
        const int n = sc.n_fast_nodes;
        uint32_t* dst = reinterpret_cast<uint32_t*>(base + off);
        const uint32_t at = ldsAddress(dst);  // skip links are LDS addresses, as a lane's `idx` is
        off += ((n + 1) * kWalkEntryBytes + 15) & ~15;
        for (int i = tid; i <= n; i += nt) {
            uint32_t* e = dst + 10 * i;
            if (i < n) {
                const float4 n0 = sc.fast_nodes[2 * i], n1 = sc.fast_nodes[2 * i + 1];
                const uint32_t skip = fbits(n1.z), first = fbits(n1.w);
                e[0] = e[2] = fbits(n0.x);
                e[1] = fbits(n0.y);
                e[3] = e[5] = fbits(n0.z);
                e[4] = fbits(n0.w);
                e[6] = e[8] = fbits(n1.x);
                e[7] = fbits(n1.y);
                e[9] = (((int)first < 0 ? kWalkNoLeaf : first) << 16) | (at + (skip < (uint32_t)n ? skip : (uint32_t)n) * kWalkEntryBytes);
            } else {
                const uint32_t lo = fbits(-__builtin_inff()), hi = fbits(__builtin_inff());
                e[0] = e[2] = e[3] = e[5] = e[6] = e[8] = lo;
                e[1] = e[4] = e[7] = hi;
                e[9] = (kWalkEnd << 16) | (at + (uint32_t)(n * kWalkEntryBytes));
            }
        }
        v.wnodes = reinterpret_cast<const char*>(dst);
        v.fnodes = sc.fast_nodes;
    } else {
        v.fnodes = take16(sc.fast_nodes, sc.n_fast_nodes * 2);
    }
    v.frefs = take4(sc.fast_refs, sc.n_fast_refs);
    v.spheres = take16(sc.spheres, sc.n_spheres);
    v.tris = take16(sc.tri_isect, sc.n_tris * 3);
    v.sphere_mat = take4i(sc.sphere_mat, sc.n_spheres);
    v.tri_mat = take4i(sc.tri_mat, sc.n_tris);
    v.materials = take16(sc.materials, sc.n_mats * 2);
    v.sphere_ord = take4i(sc.sphere_ord, sc.n_spheres);
    v.tri_ord = take4i(sc.tri_ord, sc.n_tris);
    if (withReferenceArrays) {
        v.nodes = take16(sc.nodes, sc.n_nodes * 2);
        v.refs = take4(sc.prim_refs, sc.n_prim_refs);
        v.prim_guard = take4i(sc.prim_guard, sc.n_prim_refs);
        v.node_outer = take4i(sc.node_outer, sc.n_nodes);
    } else {
        v.nodes = sc.nodes;
        v.refs = sc.prim_refs;
        v.prim_guard = sc.prim_guard;
        v.node_outer = sc.node_outer;
    }
    __syncthreads();
    return v;
}

// Pixel loops of RaytracerNormal::Run (R/src/Core/Raytracer.cpp:76-93) for the tiles this
// rank owns.  Wave w of the launch takes local tiles w, w + nWaves, ... (or pulls them from
// `queue` when it is non-null); lane l of a tile is pixel (l & 7, l >> 3) inside it.
template <bool LDS, bool COUNT>
__global__ void __launch_bounds__(256) renderKernel(RenderArgs a) {
    const DevScene& sc = a.sc;
    fillCounterLut();
    SceneView sv;
    if (LDS) {
        sv = stageScene(sc, a.knob[2] != 0);
    } else {
        sv = globalView(sc);
    }
    const int lane = threadIdx.x & 63;
    const int wavesPerBlock = blockDim.x >> 6;
    const int wave = blockIdx.x * wavesPerBlock + (threadIdx.x >> 6);
    const int nWaves = gridDim.x * wavesPerBlock;
    Stats st;
    if (COUNT)
        for (int i = 0; i < kNumStats; i++) st.v[i] = 0;

    int tile = wave;
    if (a.queue) {
        unsigned int t = 0;
        if (lane == 0) t = atomicAdd(a.queue, 1u);
        tile = (int)__builtin_amdgcn_readfirstlane(t);
    }
    while (tile < a.n_local_tiles) {
        const int ty = tile / a.tiles_per_row_local;
        const int k = tile - ty * a.tiles_per_row_local;
        int base = (a.rank - ty) % a.world;
        if (base < 0) base += a.world;
        const int tx = base + k * a.world;
        const int x = tx * kTile + (lane & 7), y = ty * kTile + (lane >> 3);
        if (tx < a.tiles_x && x < a.width && y < a.height) {
            float4 acc = a.accum[(size_t)tile * 64 + lane];
            V3 sum{acc.x, acc.y, acc.z};
            for (int s = a.first_sample; s < a.first_sample + a.n_samples; ++s) {
                Rng rng;
                rng.init((uint32_t)x, (uint32_t)y, s);
                if (COUNT) st.v[C_SAMPLES]++;
                // u is drawn before v (R/src/Core/Raytracer.cpp:86-87)
                float u = ((float)x + rng.next()) / (float)(a.width - 1);
                float v = ((float)y + rng.next()) / (float)(a.height - 1);
                Ray r = cameraRay(sc, u, v, rng);
                V3 c = rayColor<COUNT>(sc, sv, r, a.max_depth, rng, &st);
                sum = sum + c;
            }
            a.accum[(size_t)tile * 64 + lane] = make_float4(sum.x, sum.y, sum.z, (float)(a.first_sample + a.n_samples));
        }
        if (a.queue) {
            unsigned int t = 0;
            if (lane == 0) t = atomicAdd(a.queue, 1u);
            tile = (int)__builtin_amdgcn_readfirstlane(t);
        } else {
            tile += nWaves;
        }
    }
    if (COUNT) {
        for (int i = 0; i < kNumStats; i++) {
            unsigned long long v = st.v[i];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            // the scheduler statistics sit between the 13 path counters and the fallback count
            if (lane == 0 && v) atomicAdd(&a.counters[i < 13 ? i : 27], v);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Persistent wavefront loop.
//
// The unit of work is one pixel sample (one camera path).  Units are numbered sample-major,
// unit = sample * n_slots + pixel slot, and handed out in order from a launch-wide queue, 64 at a
// time per wave: one 8x8 tile of one sample.  Every lane is a small state machine
//   NEWPATH -> (BOX | PRIM)* -> SHADE -> (BOX ... | NEWPATH) ...
// that carries the name of the operation it needs next.  Each trip round the loop the wave
// ballots the lanes per operation, picks one operation, and runs it for exactly the lanes that
// want it; the others wait with their state in registers.  A lane whose path ends writes the
// path's radiance to colors[unit] and takes the next unit, so no lane idles until the queue is
// dry, and the queue is dry only when less than one path per lane is left.  (With whole pixels as
// units, lanes holding expensive pixels ran on long after the queue was dry: 67 ms instead of
// 39.5 ms per 1080p 64-spp frame, and 290 ms for one of eight ranks at 512 spp.)
// accumulateColors() then adds every pixel's radiances in sample order, which is the reference's
// `pixelColor +=` order (R/src/Core/Raytracer.cpp:84-90).  Which lane runs when never changes
// what a path computes: the random stream is keyed by (x, y, sample, counter).
// OP_BOX walks the search structure, OP_REFBOX the reference-order array (lanes with mode == 1).
enum : int { OP_BOX = 0, OP_PRIM = 1, OP_SHADE = 2, OP_NEWPATH = 3, OP_REFBOX = 4, OP_DONE = 5 };

// Lanes set in a ballot mask, as a 32-bit scalar.  (__builtin_popcountll keeps the count 64 bits
// wide, and every comparison between counts then goes through 64-bit vector compares.)
__device__ __forceinline__ int popc64(unsigned long long m) {
    int c;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(c) : "s"(m) : "scc");
    return c;
}

// Occupancy.  Tables in LDS: 512-thread workgroups, three per CU = 6 waves per SIMD (80 registers per lane).
// Measured on the cover scene: 4 waves per SIMD 37.6 ms, 6 waves 33.1 ms; later, at 25.7 ms with 6, 7 waves 28.1 ms
// and 8 waves 28.4 ms; at 20.0-20.6 ms with 6, 5 / 7 / 8 waves 24.2 / 24.5 / 25.9 ms: the LDS-resident loop is
// issue-bound and more waves only add contention.
// Tables in global memory: 8 waves per SIMD (64 registers per lane), 256-thread workgroups, eight per CU -- there
// the loads go to L1 / L2 and two more waves hide their latency: Cornell + vase 265.8 -> 239.8 ms, 1 M triangles
// 60.3 -> 58.5 ms, 10 M triangles unchanged (142 ms).
#ifndef WF_WAVES_LDS
#define WF_WAVES_LDS 6
#endif
#ifndef WF_WAVES_GLOBAL
#define WF_WAVES_GLOBAL 8
#endif
#ifndef WF_WAVES_INST
#define WF_WAVES_INST 8
#endif
// TIME (diagnostic): instead of path counters, each wave adds up its clock ticks per operation
// -- box bursts, primitive runs, SHADE, NEWPATH, reference-order box steps, and everything else
// (votes, loop overhead) -- into the scheduling-statistics slots.
// SPLIT: second pass of a two-pass launch.  primaryPass() has traced and shaded every path's camera ray a whole
// tile at a time and left one record per path that goes on; NEWPATH then hands a lane the next record instead of
// a pixel sample, and the queue counts chunks of 64 records.
template <bool LDS, bool COUNT, bool TIME = false, int HAS = S_ALL, bool NMAP = false, bool SPLIT = false, bool ONEPHASE = false,
          bool INST = false>
__global__ void __launch_bounds__(WF_MAX_THREADS, LDS ? WF_WAVES_LDS : (INST ? WF_WAVES_INST : WF_WAVES_GLOBAL)) renderWavefront(RenderArgs a) {
    const DevScene& sc = a.sc;
    // Second pass of a two-pass launch: SHADE and NEWPATH run as ONE phase.  Taking the next record is a handful of
    // loads there, both end with a new ray being set up, and a lane whose path ends in SHADE takes its next record in
    // the same turn of the loop (two-pass 20.1 -> 18.7 ms; with camera rays in NEWPATH the merged phase loses:
    // single pass 20.0 -> 20.4 ms).  ONEPHASE asks for the same in a single-pass launch: it pays for small scenes in
    // global memory, whose paths are long and whose time goes into shading (Cornell + vase 222 -> 211.5 ms), not for
    // the large ones (1 M triangles 60.6 -> 62.5 ms).
    constexpr bool MERGE = SPLIT || ONEPHASE;
    // a first pass that was cut short has not written all its records: nothing to go on with
    if (SPLIT && a.queue[1]) return;
    fillCounterLut<(HAS & S_PBR) != 0>(sc.lut);
    SceneView sv;
    if (LDS) {
        sv = stageScene(sc, a.knob[2] != 0, true);
    } else {
        sv = globalView(sc);
    }
    const int lane = threadIdx.x & 63;
    // `idx` of a lane that walks the search structure: the entry's number, or with the tables in LDS the
    // LDS address of its walk entry; walkEnd is the value that means "past the last entry"
    const int nNodesRef = INST ? sc.n_world_nodes : sc.n_nodes, nNodesFast = sc.n_fast_nodes;
    const int walkBegin = LDS ? (int)ldsAddress(sv.wnodes) : 0;
    const int walkEnd = LDS ? walkBegin + nNodesFast * kWalkEntryBytes : nNodesFast;
    const float originLimit = sc.origin_limit;
    const int nTiles = a.n_local_tiles;
    const unsigned int nChunks = (unsigned int)nTiles * (unsigned int)a.n_samples;  // 64 units each
    // the queue hands out groups of consecutive chunks (see fetchChunk)
    const unsigned int groupSize = SPLIT ? 4u : (unsigned int)a.chunk_group;
    const unsigned int nGroups = (nChunks + groupSize - 1u) / groupSize;
    const int boxKeep = a.knob[0];    // keep running BOX steps while at least this many lanes want one
    const int shadeMin = a.knob[1];   // run SHADE / NEWPATH early once this many lanes wait for it
    Stats st;
    if (COUNT)
        for (int i = 0; i < kNumStats; i++) st.v[i] = 0;

    int op = OP_NEWPATH;
    int idx = 0, p = 0;
    constexpr bool PARK = RT_PARK && !LDS && !COUNT && !TIME;
    int depthKept = 0;
    unsigned int unitKept = 0;
    V3 thrKept{1.f, 1.f, 1.f};
    auto getThr = [&]() { return PARK ? V3{g_parked[0][threadIdx.x], g_parked[1][threadIdx.x], g_parked[2][threadIdx.x]} : thrKept; };
    auto setThr = [&](V3 v) {
        if (PARK) {
            g_parked[0][threadIdx.x] = v.x;
            g_parked[1][threadIdx.x] = v.y;
            g_parked[2][threadIdx.x] = v.z;
        } else {
            thrKept = v;
        }
    };
    auto getUnit = [&]() { return PARK ? fbits(g_parked[3][threadIdx.x]) : unitKept; };
    auto setUnit = [&](unsigned int u) {
        if (PARK)
            g_parked[3][threadIdx.x] = __uint_as_float(u);
        else
            unitKept = u;
    };
    auto getDepth = [&]() { return PARK ? (int)fbits(g_parked[4][threadIdx.x]) : depthKept; };
    auto setDepth = [&](int d) {
        if (PARK)
            g_parked[4][threadIdx.x] = __uint_as_float((uint32_t)d);
        else
            depthKept = d;
    };
    setDepth(0);
    if (PARK && lane == 0) g_chunkAt[threadIdx.x >> 6][3] = a.max_depth;
    Rng rng{0u, 0u};
    Ray r{{0.f, 0.f, 0.f}, {0.f, 0.f, 1.f}};
    float invx = 0.f, invy = 0.f, invz = 0.f;
    float aa = 1.f;  // dot(d, d) for Sphere::Hit; a build for spheres alone forms it where it is needed instead
    constexpr bool carriesAa = (HAS & S_TRIS) && (HAS & S_SPHERES);
    float nox = 0.f, noy = 0.f, noz = 0.f;  // -origin / direction, for the search structure's box test
    float epad = 0.f;                       // this ray's widening of every box (farRayPad); zero nearly always
    Hit h{0.f, kNoHit, 0.f, 0.f};
    // 0: the lane walks the search structure; 1: it walks the reference-order array (diagnostic
    // setting, zero direction component, or a re-trace after a failed guard test / NaN distance)
    int mode = 0;
    // INST (scenes with instances; they walk the reference-order entries only): while the lane is inside a prototype's
    // entries -- with its ray in object space, the world ray parked in g_worldRay -- `back` is the world entry to go
    // on with afterwards (-1: not inside one) and rangeEnd the end of the prototype's entries (else of the world's);
    // inInst / hitInst: the placement being walked, and the one the closest hit so far was found through (-1: none)
    int back = -1, rangeEnd = nNodesRef, inInst = -1, hitInst = -1;

    // The wave's current chunk of the queue: 64 units = the 64 pixels of one tile for one sample.
    // chunkNext counts the lanes' worth already handed out (64 = used up), chunk < 0 = queue dry.
    int chunk = 0, chunkNext = 64, chunkLive = 64;
    int groupLeft = 0;  // chunks of the current group after this one
    int chunkSample = 0, chunkX0 = 0, chunkY0 = 0;
    bool chunkInside = false;
    unsigned int chunkAhead = 0;  // lane 0: the queue index fetched ahead of need
    // SPLIT: the record counts (a byte per chunk, four chunks per group) travel one step behind the queue index:
    // countsAhead (lane 0) belongs to group nextGroup, whose index the queue gave out one access earlier
    unsigned int countsAhead = 0, nextGroup = 0, groupCounts = 0;
    // Raytracer::Cancel() (R/src/Core/Raytracer.h:59-62; the reference polls its flag once per scanline,
    // Raytracer.cpp:54,78,148,176).  The caller's flag lives in pinned host memory, and a read of it
    // crosses PCIe, so only the first wave of a workgroup reads it, on every 16th chunk it fetches, and
    // passes what it saw to the workgroup's other waves through LDS (a read per wave per chunk -- two
    // million small PCIe reads per frame -- made a 27 ms launch take 180 ms).  A wave that learns of
    // the flag takes no more chunks (the paths it holds run to their end) and marks the launch as cut
    // short, which makes accumulateColors() leave the image alone.
    const bool pollsHost = (threadIdx.x >> 6) == 0;
    unsigned int fetches = 15;  // the first fetch polls
    // The queue is kNumQueues counters on cache lines of their own.  Atomics on ONE address retire at about 80 M/s
    // on this machine whoever issues them, and a launch of 2 M chunks in 25 ms asks for exactly that: with one
    // counter the 1080p 64-spp launch took 25.4 ms, with two chunks per access 20.3 ms, with this 19.x.
    // Counter q deals out the groups q, q + Q, q + 2 Q, ...; a wave starts at the counter of its number and moves
    // on to the next one when its own has run dry, so the last groups are still shared out one at a time.
    const unsigned int nQueues = (unsigned int)a.n_queues;
    unsigned int curQueue = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % nQueues, queueMoves = 0;
    constexpr unsigned int kCancelled = 0xFFFFFFFFu;
    auto fetchChunk = [&]() {
        unsigned int c = atomicAdd(a.queue + kQueueFirst + kQueueStride * curQueue, 1u) * nQueues + curQueue;
        if (c >= kCancelled) c = kCancelled - 1u;
        if (a.cancel) {
            if (pollsHost && (++fetches & 15u) == 0u &&
                __hip_atomic_load(a.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))
                __hip_atomic_store(&g_cancelSeen, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (__hip_atomic_load(&g_cancelSeen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                a.queue[1] = 1u;
                c = kCancelled;
            }
        }
        return c;
    };
    // What lane 0 fetched, for the wave: a group, or "no more" (>= nGroups) once every counter has run dry (or the
    // launch was cancelled).  A dry counter sends the wave to the next one, where it waits for the reply: that
    // happens at most nQueues - 1 times per wave, at the end of the launch.
    auto fetched = [&](unsigned int ahead) {
        unsigned int g = __builtin_amdgcn_readfirstlane(ahead);
        while (g >= nGroups && g != kCancelled && queueMoves + 1u < nQueues) {
            queueMoves += 1u;
            curQueue = curQueue + 1u == nQueues ? 0u : curQueue + 1u;
            unsigned int t = 0;
            if (lane == 0) t = fetchChunk();
            g = __builtin_amdgcn_readfirstlane(t);
        }
        return g;
    };
    if (SPLIT) {
        unsigned int g = 0;
        if (lane == 0) g = fetchChunk();
        nextGroup = fetched(g);
        if (lane == 0) countsAhead = nextGroup < nGroups ? reinterpret_cast<const uint32_t*>(a.record_counts)[nextGroup] : 0u;
    }
    if (lane == 0) chunkAhead = fetchChunk();
    unsigned long long schedExec[7] = {0, 0, 0, 0, 0, 0, 0}, schedLanes[7] = {0, 0, 0, 0, 0, 0, 0};
    // TIME: ticks per stretch of the loop (T_* below); tMark = last reading
    enum { T_BOX, T_PRIM, T_GUARD, T_SURFACE, T_SCATTER, T_SHADE_TAIL, T_FINISH, T_REFILL, T_CAMERA, T_REFBOX, T_REST, T_RAYSETUP, T_N };
    unsigned long long ticks[T_N] = {0}, tMark = 0;
    auto lap = [&](int slot) {  // charges the time since the last reading to `slot`
        if (TIME) {
            const unsigned long long now = (unsigned long long)clock64();
            ticks[slot] += now - tMark;
            tMark = now;
        }
    };
    if (TIME) tMark = (unsigned long long)clock64();

    auto startRay = [&]() {
        invx = 1.0f / r.d.x;
        invy = 1.0f / r.d.y;
        invz = 1.0f / r.d.z;
        if (carriesAa) aa = dot(r.d, r.d);
        h.t = __builtin_inff();
        h.ref = kNoHit;
        // with direction-ordered copies of the search structure: the copy for this ray's signs
        idx = LDS ? walkBegin : ((r.d.x < 0.0f ? 1 : 0) | (r.d.y < 0.0f ? 2 : 0) | (r.d.z < 0.0f ? 4 : 0)) * sc.fast_oct_stride;
        nox = -(r.o.x * invx);
        noy = -(r.o.y * invy);
        noz = -(r.o.z * invz);
        // a ray that starts far from the scene's small primitives (farRayPad): with the tables in LDS it widens
        // the boxes it tests; with the tables in global memory it walks the reference-order array instead
        // tables in LDS: the widening itself, branch-free; tables in global memory: only whether there is one
        if (LDS) epad = farRayPad(sc, r.o);
        mode = (INST || sc.force_ref || (!LDS && startsFar(sc, r.o)) || !fastTestUsable(r.o, originLimit, invx, invy, invz, nox, noy, noz)) ? 1 : 0;
        op = mode ? OP_REFBOX : OP_BOX;
        if (mode) idx = 0;  // the reference-order array has one order only
        if (INST) {
            back = -1;
            rangeEnd = nNodesRef;
            hitInst = -1;
        }
        if (COUNT) st.v[C_RAYS]++;
    };
    // INST: the end of a range of reference-order entries -- of a prototype's (back to the world's entries, with the
    // world ray) or of the world's (the walk is over)
    auto endOfRange = [&]() {
        if (INST && back >= 0) {
            r.o = V3{g_worldRay[0][threadIdx.x], g_worldRay[1][threadIdx.x], g_worldRay[2][threadIdx.x]};
            r.d = V3{g_worldRay[3][threadIdx.x], g_worldRay[4][threadIdx.x], g_worldRay[5][threadIdx.x]};
            invx = 1.0f / r.d.x;
            invy = 1.0f / r.d.y;
            invz = 1.0f / r.d.z;
            if (carriesAa) aa = dot(r.d, r.d);
            idx = back;
            back = -1;
            rangeEnd = nNodesRef;
            op = idx >= nNodesRef ? (h.ref == kNoHit ? OP_NEWPATH : OP_SHADE) : OP_REFBOX;
        } else {
            op = h.ref == kNoHit ? OP_NEWPATH : OP_SHADE;
        }
    };
    auto retraceInReferenceOrder = [&]() {
        h.t = __builtin_inff();
        h.ref = kNoHit;
        idx = 0;
        op = OP_REFBOX;
        mode = 1;
        if (COUNT) st.v[C_FALLBACK]++;
    };
    auto endPath = [&](V3 terminal) {
        const V3 c = getThr() * terminal;
        a.colors[getUnit()] = Radiance{c.x, c.y, c.z};
        setDepth(0);  // nothing left to finish (see afterWalk)
        op = OP_NEWPATH;
    };
    // Where a lane goes when its walk is over.  A ray that hit nothing needs no shading: the lane
    // goes straight on to NEWPATH, which first writes throughput x background for the path it
    // still holds (depth > 0 marks that), so SHADE executions carry hits only.
    auto afterWalk = [&]() { return h.ref == kNoHit ? OP_NEWPATH : OP_SHADE; };

    for (;;) {
        const int cBox = INST ? 0 : popc64(__ballot(op == OP_BOX)), cPrim = popc64(__ballot(op == OP_PRIM)),
                  cShade = popc64(__ballot(op == OP_SHADE)), cPath = popc64(__ballot(op == OP_NEWPATH)),
                  cRef = popc64(__ballot(op == OP_REFBOX));
        if ((cBox | cPrim | cShade | cPath | cRef) == 0) break;

        // Pick the operation to run: whichever has the most lanes waiting; shading and path
        // set-up are the costly ones, so they also go as soon as `shadeMin` lanes have piled up.
        int chosen;
        if (MERGE) {
            // SHADE and NEWPATH are one phase (both end with a new ray being set up): lanes waiting for either count together
            const int cRegen = cShade + cPath;
            if (cRegen >= shadeMin) {
                chosen = OP_SHADE;
            } else {
                chosen = OP_BOX;
                int best = cBox;
                if (cRef > best) { best = cRef; chosen = OP_REFBOX; }
                if (cPrim > best) { best = cPrim; chosen = OP_PRIM; }
                if (cRegen > best) { best = cRegen; chosen = OP_SHADE; }
            }
        } else if (cShade >= shadeMin || cPath >= shadeMin) {
            chosen = cShade >= cPath ? OP_SHADE : OP_NEWPATH;
        } else {
            chosen = OP_BOX;
            int best = cBox;
            if (cRef > best) { best = cRef; chosen = OP_REFBOX; }
            if (cPrim > best) { best = cPrim; chosen = OP_PRIM; }
            if (cShade > best) { best = cShade; chosen = OP_SHADE; }
            if (cPath > best) { best = cPath; chosen = OP_NEWPATH; }
        }

        if (COUNT) {  // scheduling statistics: executions and waiting lanes per operation
            const int cnt[5] = {cBox, cPrim, cShade, cPath, cRef};
            const int slot = chosen == OP_REFBOX ? 6 : chosen;
            schedExec[slot] += 1;
            schedLanes[slot] += MERGE && chosen == OP_SHADE ? cShade + cPath : cnt[chosen];
        }
        lap(T_REST);
        if (chosen == OP_BOX || chosen == OP_PRIM) {
            // Traversal phase: box bursts and primitive tests alternate, with a two-ballot vote
            // between them, for as long as one of the two has more lanes waiting than all the
            // other operations together; then back to the full vote.
            const int total = cBox + cPrim + cShade + cPath + cRef;  // lanes that are not done
            int nBox = cBox, cur = chosen;
            for (;;) {
                if (cur == OP_BOX && LDS && !INST) {
                    // Search structure in LDS: conservative fused box test on walk entries.  For the length
                    // of the burst a lane's membership is the flag `walking` (a lane mask in scalar registers)
                    // rather than `op`, the entry/exit planes come sorted from the entry's layout, and the end
                    // of the walk is an entry like any other, so a step is: three address adds, four LDS
                    // reads, three packed multiply-adds, two three-way and two plain min/max, a compare, and
                    // the choice of the next entry.
                    const int keep = boxKeep < nBox ? boxKeep : nBox;  // never spin below what is available
                    const uint32_t sx = (fbits(invx) >> 29) & 4u, sy = (fbits(invy) >> 29) & 4u, sz = (fbits(invz) >> 29) & 4u;
                    bool walking = op == OP_BOX;
                    uint32_t w = 0;
                    // the addends {entry, exit}: -o/d for both, or moved apart by the ray's own widening in the
                    // rare burst in which some lane has one
                    f32x2 ax{nox, nox}, ay{noy, noy}, az{noz, noz};
                    if (__ballot(walking && epad != 0.0f)) {
                        const float ex = epad * fabsf(invx), ey = epad * fabsf(invy), ez = epad * fabsf(invz);
                        ax = f32x2{nox - ex, nox + ex};
                        ay = f32x2{noy - ey, noy + ey};
                        az = f32x2{noz - ez, noz + ez};
                    }
                    auto boxStep = [&]() {
                        if (COUNT) {
                            schedExec[5] += 1;
                            schedLanes[5] += popc64(__ballot(walking));
                        }
                        if (walking) {
                            const LdsBytes e = (LdsBytes)(uintptr_t)(uint32_t)idx;
                            typedef __attribute__((address_space(3))) const float* LdsFloats;
                            typedef __attribute__((address_space(3))) const uint32_t* LdsWord;
                            const LdsFloats qx = (LdsFloats)(e + sx), qy = (LdsFloats)(e + 12 + sy), qz = (LdsFloats)(e + 24 + sz);
                            const f32x2u px{qx[0], qx[1]}, py{qy[0], qy[1]}, pz{qz[0], qz[1]};
                            w = *(LdsWord)(e + 36);
                            if (COUNT && idx != walkEnd) st.v[C_NODE]++;  // the closing entry is not a box of the scene
                            const f32x2 tx = __builtin_elementwise_fma(f32x2{px.x, px.y}, f32x2{invx, invx}, ax);
                            const f32x2 ty = __builtin_elementwise_fma(f32x2{py.x, py.y}, f32x2{invy, invy}, ay);
                            const f32x2 tz = __builtin_elementwise_fma(f32x2{pz.x, pz.y}, f32x2{invz, invz}, az);
                            const float tn = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, kTMin));
                            // min(tz.y, h.t) on the bit patterns: h.t is never negative and neither value is a NaN
                            // here, so the signed integer order is the float order, and fminf()'s canonicalisation
                            // of h.t (an instruction per step) is not needed
                            const int zi = __float_as_int(tz.y), hi = __float_as_int(h.t);
                            const float tf = fminf(fminf(tx.y, ty.y), __int_as_float(zi < hi ? zi : hi));
                            const bool pass = tn <= tf;
                            idx = pass ? idx + kWalkEntryBytes : (int)(w & 0xFFFFu);
                            walking = !(pass && w < (kWalkNoLeaf << 16));
                        }
                    };
                    int still;
                    do {  // three steps per look at the lane count; the look comes before the third step
                        boxStep();
                        boxStep();
                        still = popc64(__ballot(walking));
                        boxStep();
                    } while (still >= keep && keep > 0);
                    if (op == OP_BOX && !walking) {  // left the burst: at a leaf run, or at the end of the walk
                        const uint32_t first = w >> 16;
                        p = (int)first;
                        op = first == kWalkEnd ? afterWalk() : OP_PRIM;
                    }
                    lap(T_BOX);
                } else if (cur == OP_BOX && !INST) {
                    // Search structure: conservative fused box test.  An entry that passes and carries
                    // a leaf run sends the lane to PRIM; `idx` already names the entry to go on with.
                    const int keep = boxKeep < nBox ? boxKeep : nBox;  // never spin below what is available
                    // compiled twice: for the direction-ordered copies (planes stored sorted) and for a single copy
                    auto burst = [&](auto sortedPlanes) {
                        auto boxStep = [&]() {
                            if (COUNT) {
                                schedExec[5] += 1;
                                schedLanes[5] += popc64(__ballot(op == OP_BOX));
                            }
                            if (op == OP_BOX) {
                                // entry idx at a 32-bit byte offset from the (uniform) table address: the load then
                                // takes "scalar base + vector offset" and no 64-bit address is formed per lane
                                // (entries are limited to 2^27 by the upload)
                                const float4* e = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sv.fnodes) + ((uint32_t)idx << 5));
                                const float4 n0 = e[0], n1 = e[1];
                                const int first = (int)fbits(n1.w);
                                if (COUNT) st.v[C_NODE]++;
                                const bool pass = decltype(sortedPlanes)::value
                                                      ? slabPassSorted(n0, n1, invx, invy, invz, nox, noy, noz, h.t)
                                                      : slabPassFast<false>(n0, n1, invx, invy, invz, nox, noy, noz, h.t, 0.0f);
                                idx = pass ? idx + 1 : (int)fbits(n1.z);
                                p = first;  // read by PRIM only, which a lane enters with its leaf run's start
                                const int onward = idx >= nNodesFast ? afterWalk() : OP_BOX;
                                op = (pass & (first >= 0)) ? OP_PRIM : onward;
                            }
                        };
                        int still;
                        do {  // three steps per look at the lane count (1: 32.35 ms, 2: 31.75, 3: 31.6, 4: 31.7);
                              // the look comes before the third step, whose execution hides its latency
                            boxStep();
                            boxStep();
                            still = popc64(__ballot(op == OP_BOX));
                            boxStep();
                        } while (still >= keep && keep > 0);
                    };
                    if (sc.fast_oct_stride > 0)
                        burst(std::true_type{});
                    else
                        burst(std::false_type{});
                    lap(T_BOX);
                } else {
                    auto primStep = [&]() {
                        if (op == OP_PRIM) {
                            const uint32_t ref = (mode ? sv.refs : sv.frefs)[p];
                            bool bad = false;
                            // One copy of the primitive test for both walks (see primRef / primAnyOrder):
                            // the reference-order walk tests against the running closest distance and takes
                            // every hit; the search-structure walk tests without a limit and orders the hits.
                            float t, u = mode ? h.bu : 0.0f, v = mode ? h.bv : 0.0f;
                            if (primRoot<COUNT, HAS>(sv, ref, r, carriesAa ? aa : dot(r.d, r.d), mode ? h.t : __builtin_inff(), t, u, v, &st)) {
                                bool take = true;
                                if (!mode) {
                                    bad = t != t;
                                    take = !bad && (h.ref == kNoHit || t < h.t ||
                                                    (t == h.t && refOrder<HAS>(sv, ref) > refOrder<HAS>(sv, h.ref)));
                                }
                                if (take) {
                                    h.t = t;
                                    h.ref = ref & ~kRefLast;
                                    if (HAS & S_TRIS) {
                                        h.bu = u;
                                        h.bv = v;
                                    }
                                    if (INST) hitInst = back >= 0 ? inInst : -1;
                                }
                            }
                            if (bad) {
                                retraceInReferenceOrder();
                            } else if (ref & kRefLast) {
                                if (INST) {
                                    if (idx >= rangeEnd)
                                        endOfRange();
                                    else
                                        op = OP_REFBOX;
                                } else {
                                    op = idx >= (mode ? nNodesRef : walkEnd) ? afterWalk() : (mode ? OP_REFBOX : OP_BOX);
                                }
                            } else {
                                p += 1;
                            }
                        }
                    };
                    if (LDS) {
                        // go on through the leaf runs while this many lanes have more (with the tables in
                        // global memory the plain single step measured faster: 407 vs 448 ms on Cornell + vase)
                        const int primKeep = a.knob[3];
                        do {
                            primStep();
                        } while (popc64(__ballot(op == OP_PRIM)) >= primKeep);
                    } else {
                        primStep();
                    }
                    lap(T_PRIM);
                }
                nBox = INST ? 0 : popc64(__ballot(op == OP_BOX));
                const int nPrim = popc64(__ballot(op == OP_PRIM));
                const int others = total - nBox - nPrim;
                const int best = nBox > nPrim ? nBox : nPrim;
                if (best == 0 || others >= shadeMin || best < others) break;
                cur = nBox >= nPrim ? OP_BOX : OP_PRIM;
                if (COUNT) {
                    schedExec[cur] += 1;
                    schedLanes[cur] += best;
                }
                lap(T_REST);
            }
        }
        if (chosen == OP_REFBOX) {
            // Reference-order array: the reference's own test; list entries have no box.
            const int keep = boxKeep < cRef ? boxKeep : cRef;
            do {
                if (COUNT) {
                    schedExec[5] += 1;
                    schedLanes[5] += popc64(__ballot(op == OP_REFBOX));
                }
                if (op == OP_REFBOX) {
                    const float4 n0 = sv.nodes[2 * idx], n1 = sv.nodes[2 * idx + 1];
                    const uint32_t w3 = fbits(n0.w);
                    const int first = (int)fbits(n1.w);
                    const bool nobox = (w3 & kNoBox) != 0;
                    if (COUNT && !nobox) st.v[C_NODE]++;
                    const bool pass = slabPass(n0, n1, r, invx, invy, invz, h.t) || nobox;
                    if (INST && (w3 & kNodeInstance)) {
                        // a placement (the oracle's Instance::Hit): its box passed -> the prototype's entries with the ray
                        // in object space, the world ray parked; else on to the next world entry
                        if (pass) {
                            g_worldRay[0][threadIdx.x] = r.o.x;
                            g_worldRay[1][threadIdx.x] = r.o.y;
                            g_worldRay[2][threadIdx.x] = r.o.z;
                            g_worldRay[3][threadIdx.x] = r.d.x;
                            g_worldRay[4][threadIdx.x] = r.d.y;
                            g_worldRay[5][threadIdx.x] = r.d.z;
                            const float4* I = sc.instances + 5 * first;
                            const float4 range = I[4];
                            r = toObjectSpace(I, r);
                            invx = 1.0f / r.d.x;
                            invy = 1.0f / r.d.y;
                            invz = 1.0f / r.d.z;
                            if (carriesAa) aa = dot(r.d, r.d);
                            back = idx + 1;
                            inInst = first;
                            idx = (int)fbits(range.x);
                            rangeEnd = (int)fbits(range.y);
                        } else {
                            idx = idx + 1;
                            if (idx >= rangeEnd) endOfRange();
                        }
                    } else {
                        idx = pass ? idx + 1 : (int)(w3 & kSkipMask);
                        if (pass && first >= 0) {
                            p = first;
                            op = OP_PRIM;
                        } else if (idx >= (INST ? rangeEnd : nNodesRef)) {
                            if (INST)
                                endOfRange();
                            else
                                op = afterWalk();
                        }
                    }
                }
            } while (popc64(__ballot(op == OP_REFBOX)) >= keep && keep > 0);
            lap(T_REFBOX);
        }
        bool fresh = false, againRef = false;  // MERGE: the lane has a new ray, to be set up at the end of the phase
        (void)fresh;
        (void)againRef;
        if (chosen == OP_SHADE && (!MERGE || cShade > 0)) {
            // Flat on purpose (one level of conditions, outcomes chosen by selects): every level of a nest makes
            // the compiler copy the whole lane state.
            const bool shading = op == OP_SHADE;
            bool vouched = false;
            if (shading) vouched = mode != 0 || guardsPass<COUNT, HAS>(sv, r, invx, invy, invz, h, &st);
            lap(T_GUARD);
            if (shading & !vouched) retraceInReferenceOrder();
            if (shading & vouched) {
                Surface sf = INST ? surfaceAtInstanced(sc, sv, r, h, hitInst) : surfaceAt<HAS>(sc, sv, r, h);
                lap(T_SURFACE);
                const float4 m0 = sv.materials[2 * sf.mat], m1 = sv.materials[2 * sf.mat + 1];
                V3 ball{0.0f, 0.0f, 0.0f};
                if (drawsBall<HAS>((int)fbits(m1.x))) ball = randomInUnitSphere(rng);
                if (COUNT) st.v[C_HITS]++;
                Ray out{r.o, r.d};
                V3 atten{1.0f, 1.0f, 1.0f}, emit;
                const bool alive = scatter<COUNT, HAS, NMAP, true>(sc, m0, m1, r, sf, ball, rng, out, atten, emit, &st);
                lap(T_SCATTER);
                // The path goes on with throughput x attenuation and the scattered ray, unless it was absorbed
                // (terminal radiance `emit`) or has used up its depth (terminal radiance 0 = `emit` of a
                // scattering material, times the attenuated throughput as the reference forms it).
                V3 thr = getThr();
                int depth = getDepth();
                if (alive) {
                    thr = thr * atten;
                    depth -= 1;
                }
                const bool ends = !alive | (depth <= 0);
                if (COUNT && alive && depth <= 0) st.v[C_CUTOFF]++;
                if (ends) {
                    const V3 c = thr * emit;
                    a.colors[getUnit()] = Radiance{c.x, c.y, c.z};
                }
                r = out;
                lap(T_SHADE_TAIL);
                if (MERGE) {
                    fresh = !ends;
                    if (ends) {  // straight on to the next unit, below
                        depth = 0;
                        op = OP_NEWPATH;
                    }
                } else {
                    startRay();  // for a path that ends too: its lane state is dead, and NEWPATH overwrites it
                    lap(T_RAYSETUP);
                    if (ends) {
                        if (COUNT) st.v[C_RAYS]--;
                        depth = 0;
                        op = OP_NEWPATH;
                    }
                }
                setThr(thr);
                setDepth(depth);
            }
            lap(T_SHADE_TAIL);
        }
        // the next unit for every waiting lane, in lane order, and its camera ray
        if (MERGE ? (chosen == OP_SHADE && __ballot(op == OP_NEWPATH) != 0) : chosen == OP_NEWPATH) {
            const bool want = op == OP_NEWPATH;
            if (want && getDepth() > 0) {  // the path this lane holds ended on the background
                if (COUNT) st.v[C_MISS]++;
                endPath(V3{sc.bg[0], sc.bg[1], sc.bg[2]});
            }
            lap(T_FINISH);
            const unsigned long long m = __ballot(want);
            // chunk used up: the next one of its group, or the first of the group fetched ahead (SPLIT: chunks that
            // left no record are passed over)
            while (chunk >= 0 && chunkNext >= chunkLive) {
                chunkNext = 0;
                if (groupLeft > 0) {
                    groupLeft -= 1;
                    chunk += 1;
                    groupCounts >>= 8;
                } else {
                    unsigned int g;
                    if (SPLIT) {
                        g = nextGroup;
                        groupCounts = __builtin_amdgcn_readfirstlane(countsAhead);
                        nextGroup = fetched(chunkAhead);
                        if (lane == 0 && g < nGroups) {
                            countsAhead = nextGroup < nGroups ? reinterpret_cast<const uint32_t*>(a.record_counts)[nextGroup] : 0u;
                            chunkAhead = fetchChunk();
                        }
                    } else {
                        g = fetched(chunkAhead);
                        // and fetch the one after it now: the atomic's round trip to L2 (thousands of
                        // cycles) is over long before this wave needs the index
                        if (lane == 0 && g < nGroups) chunkAhead = fetchChunk();
                    }
                    if (g >= nGroups) {
                        chunk = -1;  // the queue is dry
                        break;
                    }
                    chunk = (int)(g * groupSize);
                    const unsigned int rest = nChunks - 1u - (unsigned int)chunk;
                    groupLeft = (int)(rest < groupSize - 1u ? rest : groupSize - 1u);
                }
                if (SPLIT) {
                    chunkLive = (int)(groupCounts & 0xFFu);
                } else {
                    // chunk -> (sample, tile) -> the tile's corner pixel, all wave-uniform
                    const unsigned int c = (unsigned int)chunk;
                    const int sIdx = (int)(c / (unsigned int)nTiles);
                    const int tile = (int)(c - (unsigned int)sIdx * (unsigned int)nTiles);
                    const int ty = tile / a.tiles_per_row_local;
                    const int k = tile - ty * a.tiles_per_row_local;
                    int base = (a.rank - ty) % a.world;
                    if (base < 0) base += a.world;
                    const int tx = base + k * a.world;
                    if (PARK) {
                        if (lane == 0) {
                            int* at = g_chunkAt[threadIdx.x >> 6];
                            at[0] = a.first_sample + sIdx;
                            at[1] = tx < a.tiles_x ? tx * kTile : -1;
                            at[2] = ty * kTile;
                        }
                    } else {
                        chunkSample = a.first_sample + sIdx;
                        chunkX0 = tx * kTile;
                        chunkY0 = ty * kTile;
                        chunkInside = tx < a.tiles_x;
                    }
                }
            }
            lap(T_REFILL);
            // One predicate for "this lane starts a path now", instead of nested conditions: every level of
            // a nest makes the compiler copy the whole lane state.  A lane whose turn comes after the chunk
            // ran out, or whose pixel lies outside the image, simply asks again.
            const int rankInMask =
                __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            const int l = chunkNext + rankInMask;  // pixel inside the tile
            if (want & (chunk < 0)) op = OP_DONE;
            if (SPLIT) {
                const unsigned int rec = (unsigned int)chunk * 64u + (unsigned int)l;
                if (want & (chunk >= 0) & (l < chunkLive)) {
                    const float4 q0 = a.records[rec], q1 = a.records[(size_t)a.records_cap + rec],
                                 q2 = a.records[2 * (size_t)a.records_cap + rec];
                    r.o = V3{q0.x, q0.y, q0.z};
                    r.d = V3{q0.w, q1.x, q1.y};
                    setThr(V3{q1.z, q1.w, q2.x});
                    rng.key = fbits(q2.y);
                    rng.counter = fbits(q2.z);
                    const uint32_t uw = fbits(q2.w);
                    const int again = (int)(uw >> 31);  // the camera ray itself, to be traced in reference order
                    setUnit(uw & 0x7FFFFFFFu);
                    setDepth(a.max_depth - 1 + again);
                    if (MERGE) {
                        fresh = true;
                        againRef = again != 0;
                    } else {
                        startRay();
                        if (again) {
                            idx = 0;
                            op = OP_REFBOX;
                            mode = 1;
                            if (COUNT) st.v[C_RAYS]--;  // counted by the first pass
                        }
                    }
                }
            } else {
            if (PARK) {  // (written by this wave's first lane; a wave's LDS accesses take effect in program order)
                const int* at = g_chunkAt[threadIdx.x >> 6];
                chunkSample = at[0];
                chunkX0 = at[1];
                chunkY0 = at[2];
                chunkInside = chunkX0 >= 0;
            }
            const int px = chunkX0 + (l & 7), py = chunkY0 + (l >> 3);
            if (want & (chunk >= 0) & (l < 64) & chunkInside & (px < a.width) & (py < a.height)) {
                setUnit((unsigned int)chunk * 64u + (unsigned int)l);
                rng.init((uint32_t)px, (uint32_t)py, chunkSample);
                if (COUNT) st.v[C_SAMPLES]++;
                // u is drawn before v (R/src/Core/Raytracer.cpp:86-87)
                float u = ((float)px + rng.next()) / (float)(a.width - 1);
                float v = ((float)py + rng.next()) / (float)(a.height - 1);
                r = cameraRay(sc, u, v, rng);
                setThr(V3{1.0f, 1.0f, 1.0f});
                const int maxDepth = PARK ? g_chunkAt[threadIdx.x >> 6][3] : a.max_depth;
                setDepth(maxDepth);
                if (maxDepth <= 0) {
                    if (COUNT) st.v[C_CUTOFF]++;
                    endPath(V3{0.0f, 0.0f, 0.0f});
                } else {
                    if (MERGE)
                        fresh = true;
                    else
                        startRay();
                }
            }
            }
            chunkNext += popc64(m);
            lap(T_CAMERA);
        }
        // one set-up of the new ray for the lanes that scattered and the lanes that took a new unit
        if (MERGE && chosen == OP_SHADE) {
            if (fresh) {
                startRay();
                if (SPLIT && againRef) {
                    idx = 0;
                    op = OP_REFBOX;
                    mode = 1;
                    if (COUNT) st.v[C_RAYS]--;  // counted by the first pass
                }
            }
            lap(T_RAYSETUP);
        }
    }
    if (TIME && lane == 0) {
        lap(T_REST);
        for (int i = 0; i < T_N; i++) atomicAdd(&a.counters[13 + i], ticks[i]);
        atomicAdd(&a.counters[26], 1ull);  // waves
    }
    if (COUNT && lane == 0) {
        for (int i = 0; i < 7; i++) {
            atomicAdd(&a.counters[13 + 2 * i], schedExec[i]);
            atomicAdd(&a.counters[14 + 2 * i], schedLanes[i]);
        }
    }
    if (COUNT) {
        for (int i = 0; i < kNumStats; i++) {
            unsigned long long v = st.v[i];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            // the scheduler statistics sit between the 13 path counters and the fallback count
            if (lane == 0 && v) atomicAdd(&a.counters[i < 13 ? i : 27], v);
        }
    }
}

// ---------------------------------------------------------------------------------------
// First pass of a two-pass launch: every path's camera ray, a whole tile at a time.
//
// The 64 camera rays of one 8x8 tile and one sample leave (nearly) the same point in (nearly) the same
// direction, so a wave walks the search structure ONCE for all of them: the entry's address is wave-uniform (a
// scalar), every lane tests the entry's box against its own ray, the wave descends when any lane's test
// passes and takes the skip link when none does, and a leaf run's primitives are tested by the lanes whose
// box test passed.  A lane therefore tests a superset of the entries its own walk would have visited, which
// can only add candidates to an exact, order-independent closest-hit rule (smallest distance, ties by the
// reference's visit order; see primAnyOrder): the winner is the one its own walk finds, and the guard test
// vouches for it as in the loop.  All 64 lanes are busy in every step, where the loop's box steps run for
// 24 of 64.
// Then the hit is shaded (Raytracer::rayColor's first level, R/src/Core/Raytracer.cpp:3-20) with the
// wave still whole, and the pass leaves
//   * the path's radiance in colors[unit] if it ended (background, light, absorbed, depth used up), or
//   * a record {scattered ray, throughput, random stream, unit} for the persistent loop
//     (renderWavefront<SPLIT = true>) to go on with.
// A camera ray whose answer cannot be vouched for (zero direction component, origin beyond the padding's
// reach, NaN distance, failed guard test) is handed over as it is, flagged, and the loop traces it along the
// reference-order array.  A chunk's records go to the front of the chunk's own 64 slots and their number into
// record_counts[chunk]: no atomic is involved, and the loop may take them in any order, since a path's radiance
// is a function of (pixel, sample) alone.
// Requires the tables in LDS (walk entries) and max_depth >= 1.
template <bool COUNT, int HAS, bool NMAP>
__global__ void __launch_bounds__(WF_MAX_THREADS, WF_WAVES_LDS) primaryPass(RenderArgs a) {
    const DevScene& sc = a.sc;
    fillCounterLut();
    const SceneView sv = stageScene(sc, a.knob[2] != 0, true);
    const int lane = threadIdx.x & 63;
    const uint32_t walkBegin = ldsAddress(sv.wnodes);
    const uint32_t walkEnd = walkBegin + (uint32_t)sc.n_fast_nodes * kWalkEntryBytes;
    const int nTiles = a.n_local_tiles;
    const unsigned int nChunks = (unsigned int)nTiles * (unsigned int)a.n_samples;
    const float originLimit = sc.origin_limit;
    Stats st;
    if (COUNT)
        for (int i = 0; i < kNumStats; i++) st.v[i] = 0;
    unsigned long long stepExec = 0, stepLanes = 0;

    // Chunks are dealt out in turn (the work per chunk is nearly the same everywhere, and a queue would cost an
    // atomic per chunk).  Cancel flag: see renderWavefront; the workgroup's first wave reads it on every 16th of
    // its chunks.
    const int wavesPerBlock = blockDim.x >> 6;
    const unsigned int nWaves = gridDim.x * (unsigned int)wavesPerBlock;
    const bool pollsHost = (threadIdx.x >> 6) == 0;
    unsigned int turns = 15;
    for (unsigned int c = blockIdx.x * (unsigned int)wavesPerBlock + (threadIdx.x >> 6); c < nChunks; c += nWaves) {
        if (a.cancel) {
            if (pollsHost && (++turns & 15u) == 0u &&
                __hip_atomic_load(a.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))
                __hip_atomic_store(&g_cancelSeen, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (__hip_atomic_load(&g_cancelSeen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                a.queue[1] = 1u;
                break;
            }
        }
        // chunk -> (sample, tile) -> pixel of this lane
        const int sIdx = (int)(c / (unsigned int)nTiles);
        const int tile = (int)(c - (unsigned int)sIdx * (unsigned int)nTiles);
        const int ty = tile / a.tiles_per_row_local;
        const int k = tile - ty * a.tiles_per_row_local;
        int base = (a.rank - ty) % a.world;
        if (base < 0) base += a.world;
        const int tx = base + k * a.world;
        const int px = tx * kTile + (lane & 7), py = ty * kTile + (lane >> 3);
        const bool in = (tx < a.tiles_x) & (px < a.width) & (py < a.height);
        if (__ballot(in) == 0) {
            if (lane == 0) a.record_counts[c] = 0;
            continue;
        }
        const unsigned int unit = c * 64u + (unsigned int)lane;

        Rng rng{0u, 0u};
        Ray r{{0.f, 0.f, 0.f}, {0.f, 0.f, 1.f}};
        if (in) {
            rng.init((uint32_t)px, (uint32_t)py, a.first_sample + sIdx);
            if (COUNT) {
                st.v[C_SAMPLES]++;
                st.v[C_RAYS]++;
            }
            // u is drawn before v (R/src/Core/Raytracer.cpp:86-87)
            const float u = ((float)px + rng.next()) / (float)(a.width - 1);
            const float v = ((float)py + rng.next()) / (float)(a.height - 1);
            r = cameraRay(sc, u, v, rng);
        }
        const float invx = 1.0f / r.d.x, invy = 1.0f / r.d.y, invz = 1.0f / r.d.z;
        const float aa = dot(r.d, r.d);
        const float nox = -(r.o.x * invx), noy = -(r.o.y * invy), noz = -(r.o.z * invz);
        const bool usable = in && fastTestUsable(r.o, originLimit, invx, invy, invz, nox, noy, noz);
        Hit h{__builtin_inff(), kNoHit, 0.f, 0.f};
        bool bad = false;
        if (__ballot(usable)) {
            const float epad = farRayPad(sc, r.o);
            const float ex = epad * fabsf(invx), ey = epad * fabsf(invy), ez = epad * fabsf(invz);
            const f32x2 ax{nox - ex, nox + ex}, ay{noy - ey, noy + ey}, az{noz - ez, noz + ez};
            const uint32_t sx = (fbits(invx) >> 29) & 4u, sy = (fbits(invy) >> 29) & 4u, sz = (fbits(invz) >> 29) & 4u;
            // The walk, compiled nine times: once for each sign pattern of the direction when all the wave's rays share
            // it (nearly always: the rays of a tile are nearly parallel) -- the {entry, exit} plane pairs then sit at
            // fixed offsets of the entry, and the four LDS reads of a step need no address arithmetic -- and once with
            // the offsets per lane.
            auto walk = [&](auto fixedSigns, auto signPattern) {
                constexpr bool FIXED = decltype(fixedSigns)::value;
                constexpr uint32_t KX = (decltype(signPattern)::value & 1) ? 4u : 0u, KY = (decltype(signPattern)::value & 2) ? 4u : 0u,
                                   KZ = (decltype(signPattern)::value & 4) ? 4u : 0u;
                uint32_t e = walkBegin;  // wave-uniform
                while (e < walkEnd) {
                    typedef __attribute__((address_space(3))) const float* LdsFloats;
                    typedef __attribute__((address_space(3))) const uint32_t* LdsWord;
                    const LdsBytes eb = (LdsBytes)(uintptr_t)e;
                    const LdsFloats qx = (LdsFloats)(eb + (FIXED ? KX : sx)), qy = (LdsFloats)(eb + 12 + (FIXED ? KY : sy)),
                                    qz = (LdsFloats)(eb + 24 + (FIXED ? KZ : sz));
                    const f32x2u bx{qx[0], qx[1]}, by{qy[0], qy[1]}, bz{qz[0], qz[1]};
                    const uint32_t w = __builtin_amdgcn_readfirstlane(*(LdsWord)(eb + 36));
                    const f32x2 tx2 = __builtin_elementwise_fma(f32x2{bx.x, bx.y}, f32x2{invx, invx}, ax);
                    const f32x2 ty2 = __builtin_elementwise_fma(f32x2{by.x, by.y}, f32x2{invy, invy}, ay);
                    const f32x2 tz2 = __builtin_elementwise_fma(f32x2{bz.x, bz.y}, f32x2{invz, invz}, az);
                    const float tn = fmaxf(fmaxf(tx2.x, ty2.x), fmaxf(tz2.x, kTMin));
                    // min(tz2.y, h.t) on the bit patterns, as in the loop's box step (neither is a NaN, h.t is not negative)
                    const int zi = __float_as_int(tz2.y), hi = __float_as_int(h.t);
                    const float tf = fminf(fminf(tx2.y, ty2.y), __int_as_float(zi < hi ? zi : hi));
                    const bool pass = usable & (tn <= tf);
                    const unsigned long long any = __builtin_amdgcn_ballot_w64(pass);
                    if (COUNT) {
                        if (usable) st.v[C_NODE]++;
                        stepExec += 1;
                        stepLanes += popc64(__ballot(usable));
                    }
                    if (any == 0) {
                        e = w & 0xFFFFu;
                        continue;
                    }
                    e += kWalkEntryBytes;
                    const uint32_t first = w >> 16;
                    if (first >= kWalkEnd) continue;  // no leaf run under this entry
                    uint32_t p = first, ref;
                    do {
                        ref = __builtin_amdgcn_readfirstlane(sv.frefs[p]);
                        p += 1;
                        if (pass) {
                            float t, u = 0.0f, v = 0.0f;
                            if (primRoot<COUNT, HAS>(sv, ref, r, aa, __builtin_inff(), t, u, v, &st)) {
                                if (t != t) {
                                    bad = true;
                                } else if (h.ref == kNoHit || t < h.t ||
                                           (t == h.t && refOrder<HAS>(sv, ref) > refOrder<HAS>(sv, h.ref))) {
                                    h.t = t;
                                    h.ref = ref & ~kRefLast;
                                    if (HAS & S_TRIS) {
                                        h.bu = u;
                                        h.bv = v;
                                    }
                                }
                            }
                        }
                    } while (!(ref & kRefLast));
                }
            };
            const unsigned long long mU = __builtin_amdgcn_ballot_w64(usable);
            const unsigned long long mX = __builtin_amdgcn_ballot_w64(usable && sx != 0), mY = __builtin_amdgcn_ballot_w64(usable && sy != 0),
                                     mZ = __builtin_amdgcn_ballot_w64(usable && sz != 0);
            if ((mX == 0 || mX == mU) && (mY == 0 || mY == mU) && (mZ == 0 || mZ == mU)) {
                switch ((mX ? 1 : 0) | (mY ? 2 : 0) | (mZ ? 4 : 0)) {
                    case 0: walk(std::true_type{}, std::integral_constant<int, 0>{}); break;
                    case 1: walk(std::true_type{}, std::integral_constant<int, 1>{}); break;
                    case 2: walk(std::true_type{}, std::integral_constant<int, 2>{}); break;
                    case 3: walk(std::true_type{}, std::integral_constant<int, 3>{}); break;
                    case 4: walk(std::true_type{}, std::integral_constant<int, 4>{}); break;
                    case 5: walk(std::true_type{}, std::integral_constant<int, 5>{}); break;
                    case 6: walk(std::true_type{}, std::integral_constant<int, 6>{}); break;
                    default: walk(std::true_type{}, std::integral_constant<int, 7>{}); break;
                }
            } else {
                walk(std::false_type{}, std::integral_constant<int, 0>{});
            }
        }
        // what becomes of the path: 0 = ended (colour written), 1 = goes on with the scattered ray, 2 = the camera
        // ray is handed over to be traced in reference order
        int fate = 0;
        V3 thr{1.0f, 1.0f, 1.0f};
        if (in) {
            const bool hit = h.ref != kNoHit;
            bool vouched = usable & !bad;
            if (vouched & hit) vouched = guardsPass<COUNT, HAS>(sv, r, invx, invy, invz, h, &st);
            if (!vouched) {
                if (COUNT && usable) st.v[C_FALLBACK]++;
                fate = 2;
            } else if (!hit) {
                if (COUNT) st.v[C_MISS]++;
                const V3 cc = thr * V3{sc.bg[0], sc.bg[1], sc.bg[2]};
                a.colors[unit] = Radiance{cc.x, cc.y, cc.z};
            } else {
                if (COUNT) st.v[C_HITS]++;
                const Surface sf = surfaceAt<HAS>(sc, sv, r, h);
                Ray out{r.o, r.d};
                V3 atten{1.0f, 1.0f, 1.0f}, emit;
                const float4 m0 = sv.materials[2 * sf.mat], m1 = sv.materials[2 * sf.mat + 1];
                V3 ball{0.0f, 0.0f, 0.0f};
                if (drawsBall<HAS>((int)fbits(m1.x))) ball = randomInUnitSphere(rng);
                const bool alive = scatter<COUNT, HAS, NMAP>(sc, m0, m1, r, sf, ball, rng, out, atten, emit, &st);
                int depth = a.max_depth;
                if (alive) {
                    thr = thr * atten;
                    depth -= 1;
                }
                if (!alive | (depth <= 0)) {
                    if (COUNT && alive) st.v[C_CUTOFF]++;
                    const V3 cc = thr * emit;
                    a.colors[unit] = Radiance{cc.x, cc.y, cc.z};
                } else {
                    fate = 1;
                    r = out;
                }
            }
        }
        // the chunk's records go to the front of its 64 slots, in lane order
        const unsigned long long keep = __ballot(fate != 0);
        if (lane == 0) a.record_counts[c] = (uint8_t)popc64(keep);
        if (fate != 0) {
            const unsigned int at =
                c * 64u + __builtin_amdgcn_mbcnt_hi((unsigned)(keep >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)keep, 0u));
            const uint32_t uw = unit | (fate == 2 ? 0x80000000u : 0u);
            a.records[at] = make_float4(r.o.x, r.o.y, r.o.z, r.d.x);
            a.records[(size_t)a.records_cap + at] = make_float4(r.d.y, r.d.z, thr.x, thr.y);
            a.records[2 * (size_t)a.records_cap + at] =
                make_float4(thr.z, __uint_as_float(rng.key), __uint_as_float(rng.counter), __uint_as_float(uw));
        }
    }
    if (COUNT && lane == 0) {  // the uniform steps have a pair of their own in the scheduling statistics
        atomicAdd(&a.counters[13 + 2 * 4], stepExec);
        atomicAdd(&a.counters[14 + 2 * 4], stepLanes);
    }
    if (COUNT) {
        for (int i = 0; i < kNumStats; i++) {
            unsigned long long v = st.v[i];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && v) atomicAdd(&a.counters[i < 13 ? i : 27], v);
        }
    }
}

// Second step of a render launch: accum[q] += colors[s * n_slots + q] for the launch's samples in
// order (the reference's `pixelColor +=` order, R/src/Core/Raytracer.cpp:84-90); .w = samples so far.
__global__ void __launch_bounds__(256) accumulateColors(RenderArgs a) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int nSlots = a.n_local_tiles * 64;
    if (q >= nSlots) return;
    if (a.queue && a.queue[1]) return;  // the render launch was cancelled part-way: its samples are dropped
    const int tile = q >> 6, l = q & 63;
    const int ty = tile / a.tiles_per_row_local;
    const int k = tile - ty * a.tiles_per_row_local;
    int base = (a.rank - ty) % a.world;
    if (base < 0) base += a.world;
    const int tx = base + k * a.world;
    const int px = tx * kTile + (l & 7), py = ty * kTile + (l >> 3);
    if (!(tx < a.tiles_x && px < a.width && py < a.height)) {
        // a slot outside the image holds no colour, but it does carry the sample count like every other slot
        // (the first slot of a rank's buffer is where the host reads it, and on a narrow frame shared by many
        // ranks that slot can lie outside the image)
        a.accum[q] = make_float4(0.0f, 0.0f, 0.0f, (float)(a.first_sample + a.n_samples));
        return;
    }
    const float4 acc = a.accum[q];
    V3 sum{acc.x, acc.y, acc.z};
    for (int s = 0; s < a.n_samples; s++) {
        const Radiance c = a.colors[(size_t)s * nSlots + q];
        sum = sum + V3{c.r, c.g, c.b};
    }
    a.accum[q] = make_float4(sum.x, sum.y, sum.z, (float)(a.first_sample + a.n_samples));
}

// Closest hit + hit record of arbitrary rays (test hook behind rt_trace).
__global__ void __launch_bounds__(256) traceKernel(DevScene sc, int n, const float* rays, int* kind, int* id, float* t,
                                                   float* normal, int* front, float* uv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SceneView sv = globalView(sc);
    Ray r;
    r.o = {rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]};
    r.d = {rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]};
    int inst;
    Hit h = closestHit<false>(sc, sv, r, nullptr, inst);
    if (h.ref == kNoHit) {
        kind[i] = -1;
        id[i] = -1;
        t[i] = __builtin_inff();
        normal[3 * i] = normal[3 * i + 1] = normal[3 * i + 2] = 0.0f;
        front[i] = 0;
        uv[2 * i] = uv[2 * i + 1] = 0.0f;
        return;
    }
    Surface s = surfaceAtInstanced(sc, sv, r, h, inst);
    kind[i] = (h.ref & kRefTri) ? 1 : 0;
    id[i] = (int)(h.ref & kRefId);
    t[i] = h.t;
    normal[3 * i] = s.n.x;
    normal[3 * i + 1] = s.n.y;
    normal[3 * i + 2] = s.n.z;
    front[i] = s.front ? 1 : 0;
    uv[2 * i] = s.u;
    uv[2 * i + 1] = s.v;
}

// One scatter event with injected geometry (test hook).
__global__ void scatterKernel(DevScene sc, int n, const int* mat, const float* rayIn, const float* p, const float* nrm,
                              const int* front, const float* uv, const unsigned int* key4, float* atten, float* rayOut,
                              int* alive, unsigned int* counterOut) {
    fillCounterLut();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray in;
    in.o = {rayIn[6 * i], rayIn[6 * i + 1], rayIn[6 * i + 2]};
    in.d = {rayIn[6 * i + 3], rayIn[6 * i + 4], rayIn[6 * i + 5]};
    Surface s;
    s.p = {p[3 * i], p[3 * i + 1], p[3 * i + 2]};
    s.n = {nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]};
    s.front = front[i] != 0;
    s.u = uv[2 * i];
    s.v = uv[2 * i + 1];
    s.mat = mat[i];
    s.tri = -1;
    Rng rng;
    rng.init(key4[4 * i], key4[4 * i + 1], (int)key4[4 * i + 2]);
    rng.counter = key4[4 * i + 3];
    Ray out;
    out.o = {0, 0, 0};
    out.d = {0, 0, 0};
    V3 att{0, 0, 0}, emit;
    const float4 m0 = sc.materials[2 * s.mat], m1 = sc.materials[2 * s.mat + 1];
    V3 ball{0.0f, 0.0f, 0.0f};
    if (drawsBall((int)fbits(m1.x))) ball = randomInUnitSphere(rng);
    bool ok = scatter<false>(sc, m0, m1, in, s, ball, rng, out, att, emit, nullptr);
    atten[3 * i] = att.x;
    atten[3 * i + 1] = att.y;
    atten[3 * i + 2] = att.z;
    rayOut[6 * i] = out.o.x;
    rayOut[6 * i + 1] = out.o.y;
    rayOut[6 * i + 2] = out.o.z;
    rayOut[6 * i + 3] = out.d.x;
    rayOut[6 * i + 4] = out.d.y;
    rayOut[6 * i + 5] = out.d.z;
    alive[i] = ok ? 1 : 0;
    counterOut[i] = rng.counter;
}

// Dielectric::reflectance for n (cosine, refraction ratio) pairs (test hook).
__global__ void reflectanceKernel(int n, const float* cosine, const float* refIdx, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = reflectance(cosine[i], refIdx[i]);
}

// Tiled rank payloads -> row-major image.  `payloads` holds `world` accumulation buffers of
// `stride4` float4 each (world == 1 with rank r: only that rank's pixels are filled).
// mode 0: RGBA32F sums with alpha 1; mode 1: writeColor bytes (R/src/Core/Raytracer.cpp:22-44).
__global__ void __launch_bounds__(256) resolveKernel(const float4* payloads, size_t stride4, int width, int height,
                                                     int tiles_per_row_local, int world, int only_rank, int mode,
                                                     int samples_per_pixel, float4* out32, uint32_t* out8) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= width || y >= height) return;
    const int tx = x >> 3, ty = y >> 3;
    const int owner = (tx + ty) % world;
    int base = (owner - ty) % world;
    if (base < 0) base += world;
    const int k = (tx - base) / world;
    const int lane = (x & 7) | ((y & 7) << 3);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (only_rank < 0)
        v = payloads[(size_t)owner * stride4 + ((size_t)(ty * tiles_per_row_local + k)) * 64 + lane];
    else if (owner == only_rank)
        v = payloads[((size_t)(ty * tiles_per_row_local + k)) * 64 + lane];
    const size_t o = (size_t)y * width + x;
    if (mode == 0) {
        out32[o] = make_float4(v.x, v.y, v.z, 1.0f);
    } else {
        const float scale = 1.0f / samples_per_pixel;
        float r = sqrtf(scale * v.x), g = sqrtf(scale * v.y), b = sqrtf(scale * v.z);
        auto cl = [](float q) { return q < 0.0f ? 0.0f : (q > 0.999f ? 0.999f : q); };
        uint32_t ri = (uint32_t)(uint8_t)(int)(256 * cl(r)), gi = (uint32_t)(uint8_t)(int)(256 * cl(g)),
                 bi = (uint32_t)(uint8_t)(int)(256 * cl(b));
        out8[o] = ri | (gi << 8) | (bi << 16) | (255u << 24);
    }
}

// ------------------------------------------------------------------ launch wrappers
template <bool LDS, bool COUNT, bool TIME = false, int HAS = S_ALL, bool NMAP = false, bool ONEPHASE = false, bool INST = false>
static hipError_t launchWavefrontT(const RenderArgs& a, int blocks, int threads, size_t shmem, hipStream_t stream) {
    if (threads > (LDS ? WF_MAX_THREADS : WF_GLOBAL_THREADS)) return hipErrorInvalidValue;  // per-lane LDS arrays are sized for this
    (void)hipGetLastError();  // an earlier, unrelated failure must not be reported as this launch's
    if (LDS)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(renderWavefront<LDS, COUNT, TIME, HAS, NMAP, false, ONEPHASE, INST>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    renderWavefront<LDS, COUNT, TIME, HAS, NMAP, false, ONEPHASE, INST><<<blocks, threads, shmem, stream>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int nSlots = a.n_local_tiles * 64;
    accumulateColors<<<(nSlots + 255) / 256, 256, 0, stream>>>(a);
    return hipGetLastError();
}

// Which build of the loop a scene runs: the one that holds code for what the scene holds and nothing else.
// Spheres without PBR materials (the cover scene) leave the triangle and texture code out; triangles without
// Dielectric materials (the Cornell box with the vase, the mesh scenes) leave out the sphere code and the Schlick
// term's double arithmetic.  Fewer registers are then live in every lane through every operation.
static int sceneKind(const DevScene& sc) {
    const char* force = std::getenv("RT_SCENE_KIND");  // "all": the general build whatever the scene holds (A/B runs)
    if (force && std::strcmp(force, "all") == 0) return S_ALL;
    if (sc.n_tris == 0 && !sc.has_pbr) return S_PLAIN;
    if (sc.n_spheres == 0 && !sc.has_dielectric) return S_MESH;
    return S_ALL;
}

hipError_t launchWavefront(const RenderArgs& a, bool lds, bool count, int blocks, int threads, hipStream_t stream) {
    (void)hipGetLastError();  // an earlier, unrelated failure must not be reported as this launch's
    size_t shmem = lds ? (size_t)a.lds_bytes : 0;
    // scenes with instances: the build of the loop that walks the two levels (reference-order entries, global memory)
    if (a.sc.n_instances > 0)
        return count ? launchWavefrontT<false, true, false, S_ALL, false, false, true>(a, blocks, threads, 0, stream)
                     : launchWavefrontT<false, false, false, S_ALL, false, false, true>(a, blocks, threads, 0, stream);
    if (count && a.timing)
        return lds ? launchWavefrontT<true, false, true>(a, blocks, threads, shmem, stream)
                   : launchWavefrontT<false, false, true>(a, blocks, threads, 0, stream);
    // scenes with normal mapping switched on run the build that has the code for it (counting launches of such
    // scenes too: the mapped normal changes the paths)
    if (a.sc.normal_mapping) {
        if (lds) return count ? launchWavefrontT<true, true, false, S_ALL, true>(a, blocks, threads, shmem, stream)
                              : launchWavefrontT<true, false, false, S_ALL, true>(a, blocks, threads, shmem, stream);
        return count ? launchWavefrontT<false, true, false, S_ALL, true>(a, blocks, threads, 0, stream)
                     : launchWavefrontT<false, false, false, S_ALL, true>(a, blocks, threads, 0, stream);
    }
    if (count)
        return lds ? launchWavefrontT<true, true>(a, blocks, threads, shmem, stream)
                   : launchWavefrontT<false, true>(a, blocks, threads, 0, stream);
    // small scenes read through the caches (one XCD's L2 holds the eight copies of their search structure) run
    // SHADE and NEWPATH as one phase
    const char* oneEnv = std::getenv("RT_ONE_PHASE");
    const bool onePhase = !lds && (oneEnv ? std::atoi(oneEnv) != 0 : (size_t)a.sc.n_fast_nodes * 32 <= ((size_t)4 << 20));
    if (onePhase) {
        // this shape's best vote thresholds, unless the caller set its own (Cornell + vase: box bursts kept up for 16
        // lanes and the joint phase started at 40 waiting lanes 207.7 ms, at the loop's general 8 / 32 211.6)
        RenderArgs b = a;
        if (!std::getenv("RT_BOX_KEEP")) b.knob[0] = 16;
        if (!std::getenv("RT_SHADE_MIN")) b.knob[1] = 40;
        switch (sceneKind(b.sc)) {
            case S_PLAIN: return launchWavefrontT<false, false, false, S_PLAIN, false, true>(b, blocks, threads, 0, stream);
            case S_MESH: return launchWavefrontT<false, false, false, S_MESH, false, true>(b, blocks, threads, 0, stream);
            default: return launchWavefrontT<false, false, false, S_ALL, false, true>(b, blocks, threads, 0, stream);
        }
    }
    switch (sceneKind(a.sc)) {
        case S_PLAIN:
            return lds ? launchWavefrontT<true, false, false, S_PLAIN>(a, blocks, threads, shmem, stream)
                       : launchWavefrontT<false, false, false, S_PLAIN>(a, blocks, threads, 0, stream);
        case S_MESH:
            return lds ? launchWavefrontT<true, false, false, S_MESH>(a, blocks, threads, shmem, stream)
                       : launchWavefrontT<false, false, false, S_MESH>(a, blocks, threads, 0, stream);
        default:
            return lds ? launchWavefrontT<true, false>(a, blocks, threads, shmem, stream)
                       : launchWavefrontT<false, false>(a, blocks, threads, 0, stream);
    }
}

// Two-pass launch (tables in LDS): primaryPass, the persistent loop over its records, the per-pixel sums.
template <bool COUNT, int HAS, bool NMAP, bool TIME = false>
static hipError_t launchTwoPassT(const RenderArgs& a, int blocks, int threads, size_t shmem, hipStream_t stream) {
    (void)hipGetLastError();  // an earlier, unrelated failure must not be reported as this launch's
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(primaryPass<COUNT, HAS, NMAP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(renderWavefront<true, COUNT, TIME, HAS, NMAP, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    primaryPass<COUNT, HAS, NMAP><<<blocks, threads, shmem, stream>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    renderWavefront<true, COUNT, TIME, HAS, NMAP, true><<<blocks, threads, shmem, stream>>>(a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int nSlots = a.n_local_tiles * 64;
    accumulateColors<<<(nSlots + 255) / 256, 256, 0, stream>>>(a);
    return hipGetLastError();
}

hipError_t launchTwoPass(const RenderArgs& a, bool count, int blocks, int threads, hipStream_t stream) {
    const size_t shmem = (size_t)a.lds_bytes;
    if (count && a.timing) return launchTwoPassT<false, S_ALL, false, true>(a, blocks, threads, shmem, stream);
    if (a.sc.normal_mapping)
        return count ? launchTwoPassT<true, S_ALL, true>(a, blocks, threads, shmem, stream)
                     : launchTwoPassT<false, S_ALL, true>(a, blocks, threads, shmem, stream);
    if (!count && sceneKind(a.sc) == S_PLAIN) return launchTwoPassT<false, S_PLAIN, false>(a, blocks, threads, shmem, stream);
    return count ? launchTwoPassT<true, S_ALL, false>(a, blocks, threads, shmem, stream)
                 : launchTwoPassT<false, S_ALL, false>(a, blocks, threads, shmem, stream);
}

hipError_t launchRender(const RenderArgs& a, bool lds, bool count, int blocks, int threads, hipStream_t stream) {
    (void)hipGetLastError();  // an earlier, unrelated failure must not be reported as this launch's
    size_t shmem = lds ? (size_t)a.lds_bytes : 0;
    if (lds && !count) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(renderKernel<true, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        renderKernel<true, false><<<blocks, threads, shmem, stream>>>(a);
    } else if (lds && count) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(renderKernel<true, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        renderKernel<true, true><<<blocks, threads, shmem, stream>>>(a);
    } else if (!count) {
        renderKernel<false, false><<<blocks, threads, 0, stream>>>(a);
    } else {
        renderKernel<false, true><<<blocks, threads, 0, stream>>>(a);
    }
    return hipGetLastError();
}

hipError_t launchTrace(const DevScene& sc, int n, const float* rays, int* kind, int* id, float* t, float* normal,
                       int* front, float* uv, hipStream_t stream) {
    (void)hipGetLastError();  // an earlier, unrelated failure must not be reported as this launch's
    traceKernel<<<(n + 255) / 256, 256, 0, stream>>>(sc, n, rays, kind, id, t, normal, front, uv);
    return hipGetLastError();
}

hipError_t launchScatter(const DevScene& sc, int n, const int* mat, const float* rayIn, const float* p, const float* nrm,
                         const int* front, const float* uv, const unsigned int* key4, float* atten, float* rayOut,
                         int* alive, unsigned int* counterOut, hipStream_t stream) {
    (void)hipGetLastError();  // an earlier, unrelated failure must not be reported as this launch's
    scatterKernel<<<(n + 63) / 64, 64, 0, stream>>>(sc, n, mat, rayIn, p, nrm, front, uv, key4, atten, rayOut, alive,
                                                   counterOut);
    return hipGetLastError();
}

hipError_t launchReflectance(int n, const float* cosine, const float* refIdx, float* out, hipStream_t stream) {
    (void)hipGetLastError();  // an earlier, unrelated failure must not be reported as this launch's
    reflectanceKernel<<<(n + 255) / 256, 256, 0, stream>>>(n, cosine, refIdx, out);
    return hipGetLastError();
}

hipError_t launchResolve(const float4* payloads, size_t stride4, int width, int height, int tiles_per_row_local,
                         int world, int only_rank, int mode, int spp, float4* out32, uint32_t* out8, hipStream_t stream) {
    (void)hipGetLastError();  // an earlier, unrelated failure must not be reported as this launch's
    dim3 grid((width + 31) / 32, (height + 7) / 8);
    resolveKernel<<<grid, 256, 0, stream>>>(payloads, stride4, width, height, tiles_per_row_local, world, only_rank,
                                            mode, spp, out32, out8);
    return hipGetLastError();
}

}  // namespace rt
