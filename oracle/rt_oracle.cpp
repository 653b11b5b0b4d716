// =============================================================================
//  rt_oracle.cpp  --  TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
//
//  A scalar CPU restatement of the reference renderer's hot path, written to
//  mirror the reference's *structure* (virtual Hittable tree, recursive BVH,
//  recursive rayColor, one process-global mt19937) so that the flattened,
//  iterative HIP path can be checked against something that is organised
//  differently.  Only tests/, __graft_entry__.smoke() and bench.py's
//  cpu_baseline leg may load this library; the product library never does.
//
//  PARITY STATUS: "parity unpinned".  The reference ships no tests, golden
//  vectors or fixtures for this path, and its sources need glm (an empty,
//  un-vendored submodule: /root/reference/.gitmodules:1-3), so they cannot be
//  compiled here without writing a stand-in for that library.  glm's rounding
//  behaviour is therefore an *assumption*, stated once in `namespace glmx`.
//
//  Every function cites the reference file:line it follows
//  (R/ = /root/reference/Raytracing-In-A-Weekend/).
//
//  Build: g++ -std=c++17 -O2 -ffp-contract=off -fPIC -shared (see Makefile).
// =============================================================================
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <random>
#include <thread>
#include <vector>

namespace {

// -----------------------------------------------------------------------------
// glm semantics as assumed (glm is absent; these are the textbook definitions
// of glm 0.9.9's scalar code paths).
// -----------------------------------------------------------------------------
namespace glmx {
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline float dot(vec3 a, vec3 b) {
    vec3 t = a * b;
    return t.x + t.y + t.z;  // (x + y) + z
}
inline vec3 cross(vec3 a, vec3 b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline vec3 normalize(vec3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline vec3 reflect(vec3 I, vec3 N) { return I - N * dot(N, I) * 2.0f; }
inline float radians(float d) { return d * 0.01745329251994329576923690768489f; }

struct mat4 {  // column-major: c[col][row]
    float c[4][4];
};
inline mat4 identity() {
    mat4 m{};
    for (int i = 0; i < 4; i++) m.c[i][i] = 1.0f;
    return m;
}
// mat4 * vec4(p, w):  (m0*x + m1*y) + (m2*z + m3*w)
inline void mulPoint(const mat4& m, vec3 p, float w, float out[4]) {
    for (int r = 0; r < 4; r++) {
        float mul0 = m.c[0][r] * p.x, mul1 = m.c[1][r] * p.y;
        float mul2 = m.c[2][r] * p.z, mul3 = m.c[3][r] * w;
        out[r] = (mul0 + mul1) + (mul2 + mul3);
    }
}
// glm::inverse(mat4): cofactor expansion, then scale by 1/det.
inline mat4 inverse(const mat4& mm) {
    const float(*m)[4] = mm.c;
    float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    float Fac0[4] = {Coef00, Coef00, Coef02, Coef03};
    float Fac1[4] = {Coef04, Coef04, Coef06, Coef07};
    float Fac2[4] = {Coef08, Coef08, Coef10, Coef11};
    float Fac3[4] = {Coef12, Coef12, Coef14, Coef15};
    float Fac4[4] = {Coef16, Coef16, Coef18, Coef19};
    float Fac5[4] = {Coef20, Coef20, Coef22, Coef23};
    float Vec0[4] = {m[1][0], m[0][0], m[0][0], m[0][0]};
    float Vec1[4] = {m[1][1], m[0][1], m[0][1], m[0][1]};
    float Vec2[4] = {m[1][2], m[0][2], m[0][2], m[0][2]};
    float Vec3[4] = {m[1][3], m[0][3], m[0][3], m[0][3]};
    const float SignA[4] = {+1, -1, +1, -1}, SignB[4] = {-1, +1, -1, +1};
    mat4 inv;
    for (int i = 0; i < 4; i++) {
        float Inv0 = (Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i]) + Vec3[i] * Fac2[i];
        float Inv1 = (Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i]) + Vec3[i] * Fac4[i];
        float Inv2 = (Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i]) + Vec3[i] * Fac5[i];
        float Inv3 = (Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i]) + Vec2[i] * Fac5[i];
        inv.c[0][i] = Inv0 * SignA[i];
        inv.c[1][i] = Inv1 * SignB[i];
        inv.c[2][i] = Inv2 * SignA[i];
        inv.c[3][i] = Inv3 * SignB[i];
    }
    float d0 = m[0][0] * inv.c[0][0], d1 = m[0][1] * inv.c[1][0];
    float d2 = m[0][2] * inv.c[2][0], d3 = m[0][3] * inv.c[3][0];
    float det = (d0 + d1) + (d2 + d3);
    float ood = 1.0f / det;
    for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++) inv.c[a][b] = inv.c[a][b] * ood;
    return inv;
}
}  // namespace glmx
using glmx::vec3;

const float kInfinity = std::numeric_limits<float>::infinity();  // R/src/Core/RTWeekend.h:15

// -----------------------------------------------------------------------------
// RNG.  R/src/Core/RTWeekend.h:26-58 (global mt19937 + uniform_real<float>) and
// R/assets/shaders/raytracer.comp:35-54 (stateless hash keyed x,y,sample,counter).
// -----------------------------------------------------------------------------
inline uint32_t glslHash(uint32_t x) {  // raytracer.comp:35-42
    x += (x << 10u);
    x ^= (x >> 6u);
    x += (x << 3u);
    x ^= (x >> 11u);
    x += (x << 15u);
    return x;
}
inline uint32_t fbits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
inline float hashRandom(uint32_t px, uint32_t py, int32_t sample, uint32_t counter) {
    // raytracer.comp:46-54: floatBitsToUint(vec4(x, y, sampleCount, counter))
    uint32_t vx = fbits((float)px), vy = fbits((float)py);
    uint32_t vz = fbits((float)sample), vw = fbits((float)counter);
    uint32_t m = glslHash(vx ^ glslHash(vy) ^ glslHash(vz) ^ glslHash(vw));
    m &= 0x007FFFFFu;
    m |= 0x3F800000u;
    float f;
    std::memcpy(&f, &m, 4);
    return f - 1.0f;
}

struct Counters {
    uint64_t samples = 0, rays = 0, nodeTests = 0, sphereTests = 0, triTests = 0;
    uint64_t hits = 0, scatterLambert = 0, scatterMetal = 0, scatterDielectric = 0;
    uint64_t scatterLight = 0, scatterPbr = 0, misses = 0, depthCutoffs = 0;
    void add(const Counters& o) {
        samples += o.samples; rays += o.rays; nodeTests += o.nodeTests;
        sphereTests += o.sphereTests; triTests += o.triTests; hits += o.hits;
        scatterLambert += o.scatterLambert; scatterMetal += o.scatterMetal;
        scatterDielectric += o.scatterDielectric; scatterLight += o.scatterLight;
        scatterPbr += o.scatterPbr; misses += o.misses; depthCutoffs += o.depthCutoffs;
    }
};

struct Rng {
    enum Mode { MT = 0, HASH = 1 } mode = MT;
    std::mt19937* gen = nullptr;  // the process-global stream (MT mode)
    std::uniform_real_distribution<float> dist{0.0f, 1.0f};
    uint32_t px = 0, py = 0, counter = 0;
    int32_t sample = 0;
    Counters* cnt = nullptr;
    bool dedupe = true;  // count span-1 BVH leaves once (the HIP traversal's policy)

    float randomFloat() {  // RTWeekend.h:26-31
        if (mode == MT) return dist(*gen);
        counter += 1;  // raytracer.comp:47
        return hashRandom(px, py, sample, counter);
    }
    float randomFloat(float mn, float mx) { return mn + (mx - mn) * randomFloat(); }  // :33-36
    // RTWeekend.h:45-53.  g++ evaluates the three constructor arguments right to
    // left, so z is drawn first; written out so the order is compiler-independent.
    vec3 randomVec() {
        float z = randomFloat(), y = randomFloat(), x = randomFloat();
        return {x, y, z};
    }
    vec3 randomVec(float mn, float mx) {
        float z = randomFloat(mn, mx), y = randomFloat(mn, mx), x = randomFloat(mn, mx);
        return {x, y, z};
    }
    int randomInt(int mn, int mx) { return static_cast<int>(randomFloat((float)mn, (float)(mx + 1))); }  // :55-58
    vec3 randomVecInUnitSphere() {  // RTWeekend.cpp:3-12
        while (true) {
            vec3 p = randomVec(-1.0f, 1.0f);
            if (glmx::dot(p, p) >= 1.0f) continue;
            return p;
        }
    }
    vec3 randomUnitVector() { return glmx::normalize(randomVecInUnitSphere()); }  // :14-18
    vec3 randomInUnitDisk() {  // RTWeekend.cpp:29-37 (y drawn before x under g++)
        while (true) {
            float y = randomFloat(-1.0f, 1.0f), x = randomFloat(-1.0f, 1.0f);
            vec3 p(x, y, 0.0f);
            if (glmx::dot(p, p) >= 1) continue;
            return p;
        }
    }
};

inline bool vecNearZero(vec3 v) {  // RTWeekend.cpp:39-43
    const double s = 1e-8;
    return (std::fabs(v.x) < s) && (std::fabs(v.y) < s) && (std::fabs(v.z) < s);
}
inline vec3 refract(vec3 uv, vec3 n, float etaiOverEtat) {  // RTWeekend.cpp:45-51
    float cosTheta = std::fmin(glmx::dot(-uv, n), 1.0f);
    vec3 rOutPerp = etaiOverEtat * (uv + cosTheta * n);
    vec3 rOutParallel = -std::sqrt(std::fabs(1.0f - glmx::dot(rOutPerp, rOutPerp))) * n;
    return rOutPerp + rOutParallel;
}
inline float clampf(float x, float mn, float mx) {  // RTWeekend.h:38-43
    if (x < mn) return mn;
    if (x > mx) return mx;
    return x;
}

struct Ray {  // R/src/Core/Ray.h
    vec3 origin, direction;
    vec3 At(float t) const { return origin + t * direction; }
};

struct AABB {  // R/src/AccelerationStructures/AABB.h:4-28
    vec3 minimum, maximum;
    bool Hit(const Ray& r, float tMin, float tMax) const {
        for (int a = 0; a < 3; a++) {
            float invD = 1.0f / r.direction[a];
            float t0 = (minimum[a] - r.origin[a]) * invD;
            float t1 = (maximum[a] - r.origin[a]) * invD;
            if (invD < 0.0f) std::swap(t0, t1);
            tMin = t0 > tMin ? t0 : tMin;
            tMax = t1 < tMax ? t1 : tMax;
            if (tMax <= tMin) return false;
        }
        return true;
    }
};
inline AABB surroundingBox(AABB b0, AABB b1) {  // AABB.cpp:3-12
    vec3 small(std::fmin(b0.minimum.x, b1.minimum.x), std::fmin(b0.minimum.y, b1.minimum.y),
               std::fmin(b0.minimum.z, b1.minimum.z));
    vec3 big(std::fmax(b0.maximum.x, b1.maximum.x), std::fmax(b0.maximum.y, b1.maximum.y),
             std::fmax(b0.maximum.z, b1.maximum.z));
    return {small, big};
}

struct Material;
struct HitRecord {  // R/src/Core/Hittable.h:10-28 (tangent/bitangent/modelMatrix feed only dead code)
    vec3 p, normal;
    const Material* mat = nullptr;
    float t = 0;
    bool frontFace = false;  // uninitialised in the reference; defined false here
    float u = 0, v = 0;
    vec3 tangent, bitangent;  // per-triangle frame (normal mapping, an option the reference does not have working)
    bool hasFrame = false;
    int primKind = -1, primId = -1;  // bookkeeping for hit-index parity (0 sphere, 1 triangle)
    int instance = -1;               // the placement the primitive was hit through (Instance), or -1
    void setFaceNormal(const Ray& r, vec3 outward) {
        frontFace = glmx::dot(r.direction, outward) < 0;
        normal = frontFace ? outward : -outward;
    }
};

struct Texture {  // R/src/Material/Texture.h:4-48 with stb_image.h:1840-1861's LDR->float map
    int w = 0, h = 0;
    std::vector<uint8_t> rgb;
    float lut[256];
    Texture(int W, int H, const uint8_t* data) : w(W), h(H), rgb(data, data + (size_t)W * H * 3) {
        // stbi__ldr_to_hdr: (float)(pow(byte / 255.0f, gamma) * scale) with float gamma = 2.2f and
        // scale = 1.0f; the file is compiled as C++, where pow(float, float) is the float overload
        for (int i = 0; i < 256; i++) lut[i] = (float)(std::pow(i / 255.0f, 2.2f) * 1.0f);
    }
    vec3 At(float uCoord, float vCoord) const {  // Texture.h:37-43 (row stride is the HEIGHT)
        int texelX = (int)((uCoord * w) - 0.5f);
        int texelY = (int)(((1 - vCoord) * h) - 0.5f);
        long idx = (long)texelY * h + texelX;
        long n = (long)w * h;
        if (idx < 0) idx = 0;  // the reference reads out of bounds here; clamped instead
        if (idx >= n) idx = n - 1;
        const uint8_t* p = &rgb[(size_t)idx * 3];
        return {lut[p[0]], lut[p[1]], lut[p[2]]};
    }
    vec3 AtLinear(float uCoord, float vCoord) const {  // the same texel as byte / 255 (normal maps hold directions)
        int texelX = (int)((uCoord * w) - 0.5f);
        int texelY = (int)(((1 - vCoord) * h) - 0.5f);
        long idx = (long)texelY * h + texelX;
        long n = (long)w * h;
        if (idx < 0) idx = 0;
        if (idx >= n) idx = n - 1;
        const uint8_t* p = &rgb[(size_t)idx * 3];
        return {(float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f};
    }
};

struct Material {  // R/src/Material/Material.h
    enum Kind { LAMBERTIAN = 0, METAL = 1, DIELECTRIC = 2, LIGHT = 3, PBR = 4 } kind;
    vec3 albedo;       // albedo or emit
    float param = 0;   // fuzz or ir
    const Texture *diffuse = nullptr, *rough = nullptr, *normalTex = nullptr;
    int id = -1, texD = -1, texR = -1, texN = -1;
    const bool* normalMapping = nullptr;  // the scene's switch

    vec3 emitted() const { return kind == LIGHT ? albedo : vec3(0, 0, 0); }  // :12-14, :108-110

    static float reflectance(float cosine, float refIdx) {  // :90-95, pow() and the sum run in double
        float r0 = (1 - refIdx) / (1 + refIdx);
        r0 = r0 * r0;
        return (float)((double)r0 + (double)(1 - r0) * std::pow((double)(1 - cosine), 5.0));
    }

    bool scatter(const Ray& rIn, const HitRecord& rec, vec3& attenuation, Ray& scattered, Rng& rng) const {
        using namespace glmx;
        switch (kind) {
            case LAMBERTIAN: {  // :22-36
                if (rng.cnt) rng.cnt->scatterLambert++;
                vec3 dir = rec.normal + rng.randomUnitVector();
                if (vecNearZero(dir)) dir = rec.normal;
                scattered = Ray{rec.p, dir};
                attenuation = albedo;
                return true;
            }
            case METAL: {  // :47-53
                if (rng.cnt) rng.cnt->scatterMetal++;
                vec3 reflected = reflect(normalize(rIn.direction), rec.normal);
                scattered = Ray{rec.p, reflected + param * rng.randomVecInUnitSphere()};
                attenuation = albedo;
                return dot(scattered.direction, rec.normal) > 0;
            }
            case DIELECTRIC: {  // :65-84
                if (rng.cnt) rng.cnt->scatterDielectric++;
                attenuation = vec3(1.0f, 1.0f, 1.0f);
                float ratio = rec.frontFace ? (1.0f / param) : param;
                vec3 unit = normalize(rIn.direction);
                float cosTheta = std::fmin(dot(-unit, rec.normal), 1.0f);
                float sinTheta = std::sqrt(1.0f - cosTheta * cosTheta);
                bool cannotRefract = ratio * sinTheta > 1.0f;
                vec3 dir;
                if (cannotRefract || reflectance(cosTheta, ratio) > rng.randomFloat())
                    dir = reflect(unit, rec.normal);
                else
                    dir = refract(unit, rec.normal, ratio);
                scattered = Ray{rec.p, dir};
                return true;
            }
            case LIGHT:  // :103-106
                if (rng.cnt) rng.cnt->scatterLight++;
                return false;
            case PBR: {  // :121-160; the normal-map branch is overwritten at :140, so it is skipped
                if (rng.cnt) rng.cnt->scatterPbr++;
                vec3 normal = rec.normal;
                // option (off by default): use the mapped normal the reference computes at :123-138 and drops
                // at :140 -- normalize(TBN * (2 texel - 1)), unit tangent / bitangent of the triangle, unit normal
                if (normalMapping && *normalMapping && normalTex && rec.hasFrame) {
                    vec3 T = normalize(rec.tangent), B = normalize(rec.bitangent), N = normalize(rec.normal);
                    vec3 c = normalTex->AtLinear(rec.u, rec.v);
                    vec3 m(c.x * 2.0f - 1.0f, c.y * 2.0f - 1.0f, c.z * 2.0f - 1.0f);
                    normal = normalize(vec3((T.x * m.x + B.x * m.y) + N.x * m.z, (T.y * m.x + B.y * m.y) + N.y * m.z,
                                            (T.z * m.x + B.z * m.y) + N.z * m.z));
                }
                attenuation = diffuse->At(rec.u, rec.v);
                if (rough) {
                    vec3 reflected = reflect(normalize(rIn.direction), normal);
                    scattered = Ray{rec.p, reflected + rough->At(rec.u, rec.v) * rng.randomVecInUnitSphere()};
                    return dot(scattered.direction, normal) > 0;
                }
                vec3 dir = normal + rng.randomUnitVector();
                if (vecNearZero(dir)) dir = normal;
                scattered = Ray{rec.p, dir};
                return true;
            }
        }
        return false;
    }
};

struct Hittable {  // Hittable.h:30-35
    virtual ~Hittable() = default;
    virtual bool Hit(const Ray& r, float tMin, float tMax, HitRecord& rec, Rng& ctx) const = 0;
    virtual bool BoundingBox(AABB& out) const = 0;
    virtual int kindCode() const = 0;  // 0 BVH, 1 list, 2 mesh, 3 sphere, 4 triangle
};
using HPtr = std::shared_ptr<Hittable>;

struct Vertex {  // Hittable.h:37-44 (tangent / bitangent only feed dead code)
    vec3 position, normal, textureCoord;
};

struct Triangle : Hittable {  // Hittable.h:46-220
    Vertex vertices[3];
    const Material* mat;
    int id;
    bool Hit(const Ray& r, float tMin, float tMax, HitRecord& rec, Rng& ctx) const override {
        using namespace glmx;
        if (ctx.cnt) ctx.cnt->triTests++;
        const float EPSILON = 1e-8f;
        vec3 edge1 = vertices[1].position - vertices[0].position;
        vec3 edge2 = vertices[2].position - vertices[0].position;
        vec3 h = cross(r.direction, edge2);
        float a = dot(edge1, h);
        if (a > -EPSILON && a < EPSILON) return false;
        float f = 1.0f / a;
        vec3 s = r.origin - vertices[0].position;
        float u = f * dot(s, h);
        if (u < 0.0f || u > 1.0f) return false;
        vec3 q = cross(s, edge1);
        float v = f * dot(r.direction, q);
        if (v < 0.0f || u + v > 1.0f) return false;
        float t = f * dot(edge2, q);
        if (t < tMin || tMax < t) return false;
        if (!(t > EPSILON)) return false;
        rec.t = t;
        rec.p = r.At(rec.t);
        const vec3& n0 = vertices[0].normal;
        if (n0.x == 0.0f && n0.y == 0.0f && n0.z == 0.0f)
            rec.setFaceNormal(r, normalize(cross(edge2, edge1)));
        else  // weights are permuted relative to textbook Moller-Trumbore; frontFace is left as it was
            rec.normal = normalize(u * vertices[0].normal + v * vertices[1].normal + (1 - u - v) * vertices[2].normal);
        if (vertices[0].textureCoord.x != -1.0f) {
            vec3 bc = u * vertices[0].textureCoord + v * vertices[1].textureCoord + (1 - u - v) * vertices[2].textureCoord;
            rec.u = bc.x;
            rec.v = bc.y;
        } else {
            rec.u = u;
            rec.v = v;
        }
        // tangent frame from the uv deltas (Hittable.h:110-125, each uv difference paired with the edge to the
        // same vertex); only the normal-mapping option reads it
        rec.hasFrame = false;
        if (vertices[0].textureCoord.x != -1.0f) {
            const float du1 = vertices[1].textureCoord.x - vertices[0].textureCoord.x,
                        dv1 = vertices[1].textureCoord.y - vertices[0].textureCoord.y,
                        du2 = vertices[2].textureCoord.x - vertices[0].textureCoord.x,
                        dv2 = vertices[2].textureCoord.y - vertices[0].textureCoord.y;
            const float det = du1 * dv2 - du2 * dv1;
            const float ff = 1.0f / det;
            if (det != 0.0f && std::isfinite(det) && std::isfinite(ff)) {
                float T[3], B[3];
                bool anyT = false, anyB = false, fin = true;
                for (int k = 0; k < 3; k++) {
                    const float e1 = vertices[1].position[k] - vertices[0].position[k],
                                e2 = vertices[2].position[k] - vertices[0].position[k];
                    T[k] = ff * (dv2 * e1 - dv1 * e2);
                    B[k] = ff * (-du2 * e1 + du1 * e2);
                    anyT = anyT || T[k] != 0.0f;
                    anyB = anyB || B[k] != 0.0f;
                    fin = fin && std::isfinite(T[k]) && std::isfinite(B[k]);
                }
                if (anyT && anyB && fin) {
                    rec.tangent = vec3(T[0], T[1], T[2]);
                    rec.bitangent = vec3(B[0], B[1], B[2]);
                    rec.hasFrame = true;
                }
            }
        }
        rec.mat = mat;
        rec.primKind = 1;
        rec.primId = id;
        rec.instance = -1;
        return true;
    }
    bool BoundingBox(AABB& out) const override {  // :143-213, max starts at 1e-8 and flat axes get +-1
        float mn[3] = {kInfinity, kInfinity, kInfinity};
        float mx[3] = {1e-8f, 1e-8f, 1e-8f};
        for (int a = 0; a < 3; a++) {
            for (int k = 0; k < 3; k++)
                if (vertices[k].position[a] < mn[a]) mn[a] = vertices[k].position[a];
            for (int k = 0; k < 3; k++)
                if (vertices[k].position[a] > mx[a]) mx[a] = vertices[k].position[a];
            if (mn[a] == mx[a]) {
                mn[a]--;
                mx[a]++;
            }
        }
        out = AABB{{mn[0], mn[1], mn[2]}, {mx[0], mx[1], mx[2]}};
        return true;
    }
    int kindCode() const override { return 4; }
};

struct Sphere : Hittable {  // Hittable.h:222-270
    vec3 center;
    float radius;
    const Material* mat;
    int id;
    bool Hit(const Ray& r, float tMin, float tMax, HitRecord& rec, Rng& ctx) const override {
        using namespace glmx;
        if (ctx.cnt) ctx.cnt->sphereTests++;
        vec3 oc = r.origin - center;
        float a = dot(r.direction, r.direction);
        float halfB = dot(oc, r.direction);
        float c = dot(oc, oc) - radius * radius;
        float discriminant = halfB * halfB - a * c;
        if (discriminant < 0) return false;
        float sqrtd = std::sqrt(discriminant);
        float root = (-halfB - sqrtd) / a;
        if (root < tMin || tMax < root) {
            root = (-halfB + sqrtd) / a;
            if (root < tMin || tMax < root) return false;
        }
        rec.t = root;
        rec.p = r.At(rec.t);
        vec3 outward = (rec.p - center) / radius;
        rec.setFaceNormal(r, outward);
        rec.mat = mat;
        rec.primKind = 0;
        rec.primId = id;
        rec.instance = -1;
        return true;
    }
    bool BoundingBox(AABB& out) const override {
        out = AABB{center - vec3(radius, radius, radius), center + vec3(radius, radius, radius)};
        return true;
    }
    int kindCode() const override { return 3; }
};

struct HittableList : Hittable {  // Hittable.h:272-316
    std::vector<HPtr> objects;
    bool Hit(const Ray& r, float tMin, float tMax, HitRecord& rec, Rng& ctx) const override {
        HitRecord tempRec;
        bool hitAnything = false;
        float closestSoFar = tMax;
        for (const auto& o : objects) {
            if (o->Hit(r, tMin, closestSoFar, tempRec, ctx)) {
                hitAnything = true;
                closestSoFar = tempRec.t;
                rec = tempRec;
            }
        }
        return hitAnything;
    }
    bool BoundingBox(AABB& out) const override {
        if (objects.empty()) return false;
        AABB temp;
        bool first = true;
        for (const auto& o : objects) {
            if (!o->BoundingBox(temp)) return false;
            out = first ? temp : surroundingBox(out, temp);
            first = false;
        }
        return true;
    }
    int kindCode() const override { return 1; }
};

inline bool boxCompare(const HPtr& a, const HPtr& b, int axis) {  // Bvh.cpp:3-12
    AABB ba, bb;
    a->BoundingBox(ba);
    b->BoundingBox(bb);
    return ba.minimum[axis] < bb.minimum[axis];
}

struct BVHNode : Hittable {  // Bvh.h:13-100
    HPtr left, right;
    AABB box;
    // Bvh.h:20 copies the whole source vector in every node and sorts its own [start, end) of the
    // copy.  The children receive the parent's sorted copy and touch only their own disjoint
    // sub-ranges, so one working copy sorted in place gives every node exactly the elements and
    // order it would see in its private copy; the per-node copy (quadratic in the object count)
    // is therefore made once, at the root.
    BVHNode(const std::vector<HPtr>& src, size_t start, size_t end, int maxDepth, Rng& rng) {
        auto objects = src;
        build(objects, start, end, maxDepth, rng);
    }
    struct InPlace {};
    BVHNode(InPlace, std::vector<HPtr>& objects, size_t start, size_t end, int maxDepth, Rng& rng) {
        build(objects, start, end, maxDepth, rng);
    }
    void build(std::vector<HPtr>& objects, size_t start, size_t end, int maxDepth, Rng& rng) {
        int axis = rng.randomInt(0, 2);
        auto comparator = [axis](const HPtr& a, const HPtr& b) { return boxCompare(a, b, axis); };
        size_t span = end - start;
        if (span == 1) {
            left = right = objects[start];
        } else if (span == 2) {
            if (comparator(objects[start], objects[start + 1])) {
                left = objects[start];
                right = objects[start + 1];
            } else {
                left = objects[start + 1];
                right = objects[start];
            }
        } else {
            // std::sort(objects.begin() + start, objects.begin() + end, comparator)  (Bvh.h:46) with the
            // comparator's key -- minimum[axis] of each operand's BoundingBox() -- computed once per
            // element instead of twice per comparison.  std::sort's permutation is decided by the
            // outcomes of its comparisons alone, and those are the same on the stored keys, so the
            // resulting order is the one the reference's call produces (ties included).
            {
                std::vector<std::pair<float, HPtr>> keyed;
                keyed.reserve(span);
                for (size_t i = start; i < end; i++) {
                    AABB b;
                    objects[i]->BoundingBox(b);
                    keyed.emplace_back(b.minimum[axis], std::move(objects[i]));
                }
                std::sort(keyed.begin(), keyed.end(),
                          [](const std::pair<float, HPtr>& a, const std::pair<float, HPtr>& b) { return a.first < b.first; });
                for (size_t i = start; i < end; i++) objects[i] = std::move(keyed[i - start].second);
            }
            size_t mid = start + span / 2;
            if (maxDepth == 0) {
                auto l = std::make_shared<HittableList>(), r = std::make_shared<HittableList>();
                for (size_t i = start; i < mid; i++) l->objects.push_back(objects[i]);
                for (size_t i = mid; i < end; i++) r->objects.push_back(objects[i]);
                left = l;
                right = r;
            } else {
                left = std::make_shared<BVHNode>(InPlace{}, objects, start, mid, maxDepth - 1, rng);
                right = std::make_shared<BVHNode>(InPlace{}, objects, mid, end, maxDepth - 1, rng);
            }
        }
        AABB bl, br;
        left->BoundingBox(bl);
        right->BoundingBox(br);
        box = surroundingBox(bl, br);
    }
    bool Hit(const Ray& r, float tMin, float tMax, HitRecord& rec, Rng& ctx) const override {  // :80-88
        if (ctx.cnt) ctx.cnt->nodeTests++;
        if (!box.Hit(r, tMin, tMax)) return false;
        bool hitLeft = left->Hit(r, tMin, tMax, rec, ctx);
        if (left.get() == right.get() && ctx.dedupe) {
            // span-1 node: the reference re-tests the same object with tMax = rec.t and gets the
            // same answer; the counting policy of the HIP traversal tests it once.
            return hitLeft;
        }
        bool hitRight = right->Hit(r, tMin, hitLeft ? rec.t : tMax, rec, ctx);
        return hitLeft || hitRight;
    }
    bool BoundingBox(AABB& out) const override {
        out = box;
        return true;
    }
    int kindCode() const override { return 0; }
};

struct Mesh : Hittable {  // R/src/Core/Mesh.h:8-50
    AABB boundingBox;
    std::vector<HPtr> triangles;  // holds just the BVH root after construction (Mesh.cpp:29-31)
    bool Hit(const Ray& r, float tMin, float tMax, HitRecord& rec, Rng& ctx) const override {
        if (ctx.cnt) ctx.cnt->nodeTests++;
        if (!boundingBox.Hit(r, tMin, tMax)) return false;
        HitRecord tempRec;
        bool hitAnything = false;
        float closestSoFar = tMax;
        for (const auto& tri : triangles) {
            if (tri->Hit(r, tMin, closestSoFar, tempRec, ctx)) {
                hitAnything = true;
                closestSoFar = tempRec.t;
                rec = tempRec;
            }
        }
        return hitAnything;
    }
    bool BoundingBox(AABB& out) const override {
        out = boundingBox;
        return true;
    }
    int kindCode() const override { return 2; }
};

// A placement of a shared object.  NO COUNTERPART IN THE REFERENCE, which bakes every placement into world-space
// triangles when a Mesh is built (R/src/Core/Mesh.cpp:106-208) and gives each its own tree; this class defines
// the opt-in instanced mode of the product, the way PBR normal mapping is defined here.  Hit: the placement's
// world-space box first (as Mesh::Hit tests its own), then the ray taken into object space -- origin as a
// point, direction as a vector that is NOT normalised, so that hit distances are the same numbers in both
// spaces -- the shared object's Hit with the caller's limits, and the record brought back: the point from the
// world ray, the normal through mat3(transpose(inverse(M))) (what Mesh.cpp:144,148-150 applies to baked
// normals) and normalised again.
struct Instance : Hittable {
    using mat4 = glmx::mat4;
    std::shared_ptr<HittableList> object;
    mat4 M, inv;
    AABB box;
    int id = 0;
    void place(const mat4& model, int instanceId) {
        using namespace glmx;
        M = model;
        inv = inverse(model);
        id = instanceId;
        AABB ob;
        object->BoundingBox(ob);
        float mn[3], mx[3];
        for (int c = 0; c < 8; c++) {
            vec3 p((c & 1) ? ob.maximum.x : ob.minimum.x, (c & 2) ? ob.maximum.y : ob.minimum.y,
                   (c & 4) ? ob.maximum.z : ob.minimum.z);
            float o[4];
            mulPoint(M, p, 1.0f, o);
            for (int k = 0; k < 3; k++) {
                if (c == 0 || o[k] < mn[k]) mn[k] = o[k];
                if (c == 0 || o[k] > mx[k]) mx[k] = o[k];
            }
        }
        box = AABB{{mn[0], mn[1], mn[2]}, {mx[0], mx[1], mx[2]}};
    }
    Ray toObject(const Ray& r) const {
        using namespace glmx;
        float o[4];
        mulPoint(inv, r.origin, 1.0f, o);
        const vec3& d = r.direction;
        vec3 dd((inv.c[0][0] * d.x + inv.c[1][0] * d.y) + inv.c[2][0] * d.z,
                (inv.c[0][1] * d.x + inv.c[1][1] * d.y) + inv.c[2][1] * d.z,
                (inv.c[0][2] * d.x + inv.c[1][2] * d.y) + inv.c[2][2] * d.z);
        return Ray{vec3(o[0], o[1], o[2]), dd};
    }
    bool Hit(const Ray& r, float tMin, float tMax, HitRecord& rec, Rng& ctx) const override {
        using namespace glmx;
        if (ctx.cnt) ctx.cnt->nodeTests++;
        if (!box.Hit(r, tMin, tMax)) return false;
        const Ray ro = toObject(r);
        if (!object->Hit(ro, tMin, tMax, rec, ctx)) return false;
        rec.p = r.At(rec.t);
        const vec3 n = rec.normal;
        rec.normal = normalize(vec3((inv.c[0][0] * n.x + inv.c[0][1] * n.y) + inv.c[0][2] * n.z,
                                    (inv.c[1][0] * n.x + inv.c[1][1] * n.y) + inv.c[1][2] * n.z,
                                    (inv.c[2][0] * n.x + inv.c[2][1] * n.y) + inv.c[2][2] * n.z));
        rec.instance = id;
        return true;
    }
    bool BoundingBox(AABB& out) const override {
        out = box;
        return true;
    }
    int kindCode() const override { return 5; }
};

struct Camera {  // R/src/Core/Camera.h:3-40
    vec3 origin, lowerLeftCorner, horizontal, vertical, u, v, w;
    float lensRadius = 0;
    void set(vec3 lookfrom, vec3 lookat, vec3 vup, float vfov, float aspect, float aperture, float focusDist) {
        using namespace glmx;
        float theta = radians(vfov);
        float h = std::tan(theta / 2.0f);
        float viewportHeight = 2.0f * h;
        float viewportWidth = aspect * viewportHeight;
        w = normalize(lookfrom - lookat);
        u = normalize(cross(vup, w));
        v = cross(w, u);
        origin = lookfrom;
        horizontal = focusDist * viewportWidth * u;
        vertical = focusDist * viewportHeight * v;
        lowerLeftCorner = origin - horizontal / 2.0f - vertical / 2.0f - focusDist * w;
        lensRadius = aperture / 2;
    }
    Ray GetRay(float s, float t, Rng& rng) const {
        vec3 rd = lensRadius * rng.randomInUnitDisk();
        vec3 offset = u * rd.x + v * rd.y;
        return Ray{origin + offset, lowerLeftCorner + s * horizontal + t * vertical - origin - offset};
    }
};

struct Scene {
    std::mt19937 gen;  // RTWeekend.h:29, default-seeded (5489), never reseeded
    Rng rng;
    HittableList world;
    Camera camera;
    vec3 background;
    bool normalMapping = false;
    std::vector<std::unique_ptr<Material>> materials;
    std::vector<std::unique_ptr<Texture>> textures;
    int nSpheres = 0, nTriangles = 0;
    std::vector<std::shared_ptr<Sphere>> spheres;
    std::vector<std::shared_ptr<Triangle>> tris;
    // objects that are placed by Instance rather than put into the world themselves; while one is being
    // defined, everything that would go into the world goes into it
    std::vector<std::shared_ptr<HittableList>> prototypes;
    int openPrototype = -1, nInstances = 0;
    std::vector<HPtr>& sink() { return openPrototype >= 0 ? prototypes[openPrototype]->objects : world.objects; }
    Scene() {
        rng.mode = Rng::MT;
        rng.gen = &gen;
    }
    Material* addMaterial(Material::Kind k, vec3 a, float p) {
        auto m = std::make_unique<Material>();
        m->kind = k;
        m->albedo = a;
        m->param = p;
        m->id = (int)materials.size();
        m->normalMapping = &normalMapping;
        materials.push_back(std::move(m));
        return materials.back().get();
    }
    std::shared_ptr<Sphere> makeSphere(vec3 c, float r, const Material* m) {
        auto s = std::make_shared<Sphere>();
        s->center = c;
        s->radius = r;
        s->mat = m;
        s->id = nSpheres++;
        spheres.push_back(s);
        return s;
    }
    std::shared_ptr<Triangle> makeTriangle(const Vertex& a, const Vertex& b, const Vertex& c, const Material* m) {
        auto t = std::make_shared<Triangle>();
        t->vertices[0] = a;
        t->vertices[1] = b;
        t->vertices[2] = c;
        t->mat = m;
        t->id = nTriangles++;
        tris.push_back(t);
        return t;
    }
};

// R/src/RaytracingApplication.cpp:359-405.  The rejection test at :370 calls glm's
// vec3::length(), which is the component count (3), so no sphere is ever rejected.
void randomScene(Scene& sc) {
    Rng& rng = sc.rng;
    auto ground = sc.addMaterial(Material::LAMBERTIAN, vec3(0.5f, 0.5f, 0.5f), 0);
    sc.sink().push_back(sc.makeSphere(vec3(0.0f, -1000.0f, 0.0f), 1000.0f, ground));
    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++) {
            float chooseMat = rng.randomFloat();
            // g++ evaluates the constructor arguments right to left: z's draw comes first
            float rz = rng.randomFloat();
            float rx = rng.randomFloat();
            vec3 center(a + 0.9f * rx, 0.2f, b + 0.9f * rz);
            if (3 > 0.9f) {
                if (chooseMat < 0.8f) {
                    vec3 second = rng.randomVec();  // right operand of operator* first
                    vec3 first = rng.randomVec();
                    vec3 albedo = first * second;
                    sc.sink().push_back(sc.makeSphere(center, 0.2f, sc.addMaterial(Material::LAMBERTIAN, albedo, 0)));
                } else if (chooseMat < 0.95f) {
                    vec3 albedo = rng.randomVec(0.5f, 1.0f);
                    float fuzz = rng.randomFloat(0.0f, 0.5f);
                    sc.sink().push_back(
                        sc.makeSphere(center, 0.2f, sc.addMaterial(Material::METAL, albedo, fuzz < 1 ? fuzz : 1)));
                } else {
                    sc.sink().push_back(sc.makeSphere(center, 0.2f, sc.addMaterial(Material::DIELECTRIC, vec3(), 1.5f)));
                }
            }
        }
    }
    sc.sink().push_back(sc.makeSphere(vec3(0.0f, 1.0f, 0.0f), 1.0f, sc.addMaterial(Material::DIELECTRIC, vec3(), 1.5f)));
    sc.sink().push_back(
        sc.makeSphere(vec3(-4.0f, 1.0f, 0.0f), 1.0f, sc.addMaterial(Material::LAMBERTIAN, vec3(0.4f, 0.2f, 0.1f), 0)));
    sc.sink().push_back(
        sc.makeSphere(vec3(4.0f, 1.0f, 0.0f), 1.0f, sc.addMaterial(Material::METAL, vec3(0.7f, 0.6f, 0.5f), 0.0f)));
}

// R/src/RaytracingApplication.cpp:407-478
void cornellBox(Scene& sc) {
    auto red = sc.addMaterial(Material::LAMBERTIAN, vec3(0.65f, 0.05f, 0.05f), 0);
    auto white = sc.addMaterial(Material::LAMBERTIAN, vec3(0.73f, 0.73f, 0.73f), 0);
    auto green = sc.addMaterial(Material::LAMBERTIAN, vec3(0.12f, 0.45f, 0.15f), 0);
    auto light = sc.addMaterial(Material::LIGHT, vec3(15.0f, 15.0f, 15.0f), 0);
    auto tri = [&](vec3 a, vec3 b, vec3 c, const Material* m) {
        Vertex v0, v1, v2;  // normals zero; texture coordinates are uninitialised in the reference
        v0.position = a;
        v1.position = b;
        v2.position = c;
        sc.sink().push_back(sc.makeTriangle(v0, v1, v2, m));
    };
    tri({555, 0, 0}, {555, 555, 0}, {555, 555, 555}, green);
    tri({555, 0, 0}, {555, 555, 555}, {555, 0, 555}, green);
    tri({0, 0, 0}, {0, 555, 0}, {0, 555, 555}, red);
    tri({0, 0, 0}, {0, 555, 555}, {0, 0, 555}, red);
    tri({213, 554, 227}, {343, 554, 227}, {343, 554, 332}, light);
    tri({213, 554, 227}, {343, 554, 332}, {213, 554, 332}, light);
    tri({0, 0, 0}, {555, 0, 0}, {555, 0, 555}, white);
    tri({0, 0, 0}, {555, 0, 555}, {0, 0, 555}, white);
    tri({0, 555, 0}, {555, 555, 0}, {555, 555, 555}, white);
    tri({0, 555, 0}, {555, 555, 555}, {0, 555, 555}, white);
    tri({0, 0, 555}, {555, 0, 555}, {555, 555, 555}, white);
    tri({0, 0, 555}, {555, 555, 555}, {0, 555, 555}, white);
}

// R/src/Core/Mesh.cpp:7-210 with the importer's output passed in as arrays
// (positions, normals, texture coordinates already in the importer's convention).
void addMesh(Scene& sc, const float* model16, int nVerts, const float* pos, const float* nrm, const float* uv,
             int nTris, const uint32_t* idx, const Material* mat, int maxDepth) {
    using namespace glmx;
    mat4 M;
    std::memcpy(M.c, model16, sizeof(float) * 16);
    (void)nVerts;
    float mn[3] = {kInfinity, kInfinity, kInfinity}, mx[3] = {1e-8f, 1e-8f, 1e-8f};
    std::vector<HPtr> triangles;
    mat4 inv = inverse(M);
    // normalMatrix = mat3(transpose(inverse(M))): column j of it is row j of inv
    for (int f = 0; f < nTris; f++) {
        Vertex vert[3];
        for (int k = 0; k < 3; k++) {
            uint32_t vi = idx[3 * f + k];
            vec3 p(pos[3 * vi], pos[3 * vi + 1], pos[3 * vi + 2]);
            vec3 tex = k == 0 ? vec3(0.0f, 0.0f, 0.0f) : (k == 1 ? vec3(0.1f, 0.0f, 0.0f) : vec3(0.5f, 1.0f, 0.0f));
            if (uv) {
                tex.x = uv[2 * vi];
                tex.y = uv[2 * vi + 1];
            }
            vec3 n(0.0f, 0.0f, 0.0f);
            if (nrm) n = normalize(vec3(nrm[3 * vi], nrm[3 * vi + 1], nrm[3 * vi + 2]));
            float o[4];
            mulPoint(M, p, 1.0f, o);
            vert[k].position = vec3(o[0], o[1], o[2]);
            // mat3 * vec3: (c0*x + c1*y) + c2*z with c_j[r] = inv.c[r][j]
            vec3 nn((inv.c[0][0] * n.x + inv.c[0][1] * n.y) + inv.c[0][2] * n.z,
                    (inv.c[1][0] * n.x + inv.c[1][1] * n.y) + inv.c[1][2] * n.z,
                    (inv.c[2][0] * n.x + inv.c[2][1] * n.y) + inv.c[2][2] * n.z);
            vert[k].normal = nn;
            vert[k].textureCoord = tex;
        }
        triangles.push_back(sc.makeTriangle(vert[0], vert[1], vert[2], mat));
        for (int a = 0; a < 3; a++) {
            for (int k = 0; k < 3; k++)
                if (vert[k].position[a] < mn[a]) mn[a] = vert[k].position[a];
            for (int k = 0; k < 3; k++)
                if (vert[k].position[a] > mx[a]) mx[a] = vert[k].position[a];
        }
    }
    auto mesh = std::make_shared<Mesh>();
    mesh->boundingBox = AABB{{mn[0], mn[1], mn[2]}, {mx[0], mx[1], mx[2]}};
    mesh->triangles.push_back(std::make_shared<BVHNode>(triangles, 0, triangles.size(), maxDepth, sc.rng));
    sc.sink().push_back(mesh);
}

// R/src/Core/Raytracer.cpp:3-20
vec3 rayColor(const Scene& sc, const Ray& r, int depth, Rng& rng) {
    HitRecord rec;
    if (depth <= 0) {
        if (rng.cnt) rng.cnt->depthCutoffs++;
        return vec3(0.0f, 0.0f, 0.0f);
    }
    if (rng.cnt) rng.cnt->rays++;
    if (!sc.world.Hit(r, 0.001f, kInfinity, rec, rng)) {
        if (rng.cnt) rng.cnt->misses++;
        return sc.background;
    }
    if (rng.cnt) rng.cnt->hits++;
    Ray scattered;
    vec3 attenuation;
    vec3 emitted = rec.mat->emitted();
    if (!rec.mat->scatter(r, rec, attenuation, scattered, rng)) return emitted;
    return emitted + attenuation * rayColor(sc, scattered, depth - 1, rng);
}

// The same estimator with the attenuation product accumulated front to back
// (throughput *= attenuation; colour = throughput * terminal radiance).  This is the
// evaluation order of the HIP kernel; it differs from the recursion above only by
// re-association of the float products (same path, same hits, same random draws).
// Valid because no reference material both emits and scatters (Material.h:98-114).
vec3 rayColorThroughput(const Scene& sc, Ray r, int depth, Rng& rng) {
    vec3 throughput(1.0f, 1.0f, 1.0f);
    for (;;) {
        HitRecord rec;
        if (depth <= 0) {
            if (rng.cnt) rng.cnt->depthCutoffs++;
            return throughput * vec3(0.0f, 0.0f, 0.0f);
        }
        if (rng.cnt) rng.cnt->rays++;
        if (!sc.world.Hit(r, 0.001f, kInfinity, rec, rng)) {
            if (rng.cnt) rng.cnt->misses++;
            return throughput * sc.background;
        }
        if (rng.cnt) rng.cnt->hits++;
        Ray scattered;
        vec3 attenuation;
        vec3 emitted = rec.mat->emitted();
        if (!rec.mat->scatter(r, rec, attenuation, scattered, rng)) return throughput * emitted;
        throughput = throughput * attenuation;
        r = scattered;
        depth--;
    }
}

inline void writeColorBytes(vec3 sum, int samplesPerPixel, uint8_t* out4) {  // Raytracer.cpp:22-44
    float scale = 1.0f / samplesPerPixel;
    float r = std::sqrt(scale * sum.x), g = std::sqrt(scale * sum.y), b = std::sqrt(scale * sum.z);
    out4[0] = (uint8_t) static_cast<int>(256 * clampf(r, 0.0f, 0.999f));
    out4[1] = (uint8_t) static_cast<int>(256 * clampf(g, 0.0f, 0.999f));
    out4[2] = (uint8_t) static_cast<int>(256 * clampf(b, 0.0f, 0.999f));
    out4[3] = 255;
}

}  // namespace

// =============================================================================
// C interface used by the tests (ctypes) and bench.py's cpu_baseline leg.
// =============================================================================
extern "C" {

struct OrcCounters {
    uint64_t samples, rays, nodeTests, sphereTests, triTests, hits;
    uint64_t scatterLambert, scatterMetal, scatterDielectric, scatterLight, scatterPbr, misses, depthCutoffs;
};

void* orc_scene_new() { return new Scene(); }
void orc_scene_free(void* s) { delete (Scene*)s; }
void orc_scene_random(void* s) { randomScene(*(Scene*)s); }
void orc_scene_cornell(void* s) { cornellBox(*(Scene*)s); }

int orc_add_material(void* s, int kind, float r, float g, float b, float param, int texD, int texR, int texN) {
    Scene& sc = *(Scene*)s;
    if (kind == Material::METAL) param = param < 1 ? param : 1;  // Material.h:45
    Material* m = sc.addMaterial((Material::Kind)kind, vec3(r, g, b), param);
    m->texD = texD;
    m->texR = texR;
    m->texN = texN;
    if (texD >= 0) m->diffuse = sc.textures[texD].get();
    if (texR >= 0) m->rough = sc.textures[texR].get();
    if (texN >= 0) m->normalTex = sc.textures[texN].get();
    return m->id;
}
int orc_add_texture(void* s, int w, int h, const uint8_t* rgb) {
    Scene& sc = *(Scene*)s;
    sc.textures.push_back(std::make_unique<Texture>(w, h, rgb));
    return (int)sc.textures.size() - 1;
}
int orc_add_sphere(void* s, float cx, float cy, float cz, float r, int mat) {
    Scene& sc = *(Scene*)s;
    sc.sink().push_back(sc.makeSphere(vec3(cx, cy, cz), r, sc.materials[mat].get()));
    return sc.nSpheres - 1;
}
// pos[9], nrm[9] (all-zero first normal = flat shading), uv[6] (uv[0] == -1 means "none")
int orc_add_triangle(void* s, const float* pos, const float* nrm, const float* uv, int mat) {
    Scene& sc = *(Scene*)s;
    Vertex v[3];
    for (int k = 0; k < 3; k++) {
        v[k].position = vec3(pos[3 * k], pos[3 * k + 1], pos[3 * k + 2]);
        v[k].normal = vec3(nrm[3 * k], nrm[3 * k + 1], nrm[3 * k + 2]);
        v[k].textureCoord = vec3(uv[2 * k], uv[2 * k + 1], 0.0f);
    }
    sc.sink().push_back(sc.makeTriangle(v[0], v[1], v[2], sc.materials[mat].get()));
    return sc.nTriangles - 1;
}
void orc_add_mesh(void* s, const float* model16, int nVerts, const float* pos, const float* nrm, const float* uv,
                  int nTris, const uint32_t* idx, int mat, int maxDepth) {
    Scene& sc = *(Scene*)s;
    addMesh(sc, model16, nVerts, pos, nrm, uv, nTris, idx, sc.materials[mat].get(), maxDepth);
}
int orc_begin_prototype(void* s) {
    Scene& sc = *(Scene*)s;
    if (sc.openPrototype >= 0) return -1;
    sc.prototypes.push_back(std::make_shared<HittableList>());
    sc.openPrototype = (int)sc.prototypes.size() - 1;
    return sc.openPrototype;
}
int orc_end_prototype(void* s) {
    Scene& sc = *(Scene*)s;
    const int id = sc.openPrototype;
    sc.openPrototype = -1;
    return id;
}
int orc_add_instance(void* s, int prototype, const float* model16) {
    Scene& sc = *(Scene*)s;
    if (prototype < 0 || prototype >= (int)sc.prototypes.size() || prototype == sc.openPrototype) return -1;
    auto inst = std::make_shared<Instance>();
    inst->object = sc.prototypes[prototype];
    glmx::mat4 M;
    std::memcpy(M.c, model16, sizeof(float) * 16);
    inst->place(M, sc.nInstances++);
    sc.sink().push_back(inst);
    return inst->id;
}
// world = { BVHNode(world) }  (Bvh.h:17: unlimited depth), consuming the scene's RNG stream
void orc_wrap_bvh(void* s) {
    Scene& sc = *(Scene*)s;
    auto root = std::make_shared<BVHNode>(sc.world.objects, 0, sc.world.objects.size(), -1, sc.rng);
    sc.world.objects.clear();
    sc.world.objects.push_back(root);
}
void orc_set_camera(void* s, const float* lookfrom, const float* lookat, const float* vup, float vfov, float aspect,
                    float aperture, float focusDist) {
    Scene& sc = *(Scene*)s;
    sc.camera.set(vec3(lookfrom[0], lookfrom[1], lookfrom[2]), vec3(lookat[0], lookat[1], lookat[2]),
                  vec3(vup[0], vup[1], vup[2]), vfov, aspect, aperture, focusDist);
}
void orc_get_camera(void* s, float* out22) {
    const Camera& c = ((Scene*)s)->camera;
    const vec3* v[7] = {&c.origin, &c.lowerLeftCorner, &c.horizontal, &c.vertical, &c.u, &c.v, &c.w};
    for (int i = 0; i < 7; i++) {
        out22[3 * i] = v[i]->x;
        out22[3 * i + 1] = v[i]->y;
        out22[3 * i + 2] = v[i]->z;
    }
    out22[21] = c.lensRadius;
}
void orc_set_background(void* s, float r, float g, float b) { ((Scene*)s)->background = vec3(r, g, b); }
void orc_set_normal_mapping(void* s, int on) { ((Scene*)s)->normalMapping = on != 0; }

void orc_rng_floats(void* s, int n, float* out) {
    Scene& sc = *(Scene*)s;
    for (int i = 0; i < n; i++) out[i] = sc.rng.randomFloat();
}
float orc_hash_random(uint32_t x, uint32_t y, int32_t sample, uint32_t counter) { return hashRandom(x, y, sample, counter); }

int orc_num_spheres(void* s) { return ((Scene*)s)->nSpheres; }
int orc_num_triangles(void* s) { return ((Scene*)s)->nTriangles; }
int orc_num_materials(void* s) { return (int)((Scene*)s)->materials.size(); }
void orc_dump_spheres(void* s, float* xyzr, int* mat) {
    Scene& sc = *(Scene*)s;
    for (int i = 0; i < sc.nSpheres; i++) {
        xyzr[4 * i] = sc.spheres[i]->center.x;
        xyzr[4 * i + 1] = sc.spheres[i]->center.y;
        xyzr[4 * i + 2] = sc.spheres[i]->center.z;
        xyzr[4 * i + 3] = sc.spheres[i]->radius;
        mat[i] = sc.spheres[i]->mat->id;
    }
}
// per triangle: 9 pos, 9 normal, 6 uv = 24 floats
void orc_dump_triangles(void* s, float* out24, int* mat) {
    Scene& sc = *(Scene*)s;
    for (int i = 0; i < sc.nTriangles; i++) {
        const Triangle& t = *sc.tris[i];
        float* o = out24 + 24 * i;
        for (int k = 0; k < 3; k++) {
            o[3 * k] = t.vertices[k].position.x;
            o[3 * k + 1] = t.vertices[k].position.y;
            o[3 * k + 2] = t.vertices[k].position.z;
            o[9 + 3 * k] = t.vertices[k].normal.x;
            o[9 + 3 * k + 1] = t.vertices[k].normal.y;
            o[9 + 3 * k + 2] = t.vertices[k].normal.z;
            o[18 + 2 * k] = t.vertices[k].textureCoord.x;
            o[18 + 2 * k + 1] = t.vertices[k].textureCoord.y;
        }
        mat[i] = t.mat->id;
    }
}
// per material: kind, then r g b param as floats
void orc_dump_materials(void* s, int* kind, float* rgbp) {
    Scene& sc = *(Scene*)s;
    for (size_t i = 0; i < sc.materials.size(); i++) {
        kind[i] = sc.materials[i]->kind;
        rgbp[4 * i] = sc.materials[i]->albedo.x;
        rgbp[4 * i + 1] = sc.materials[i]->albedo.y;
        rgbp[4 * i + 2] = sc.materials[i]->albedo.z;
        rgbp[4 * i + 3] = sc.materials[i]->param;
    }
}

// Visit-order dump of the Hittable tree.  For every box-tested node (BVHNode or Mesh), in the
// order a ray that hits every box would test them: 6 floats of box and the number of box-tested
// nodes in its subtree (itself excluded).  For every primitive in the order it would be tested
// (span-1 BVH leaves listed once): kind (0 sphere, 1 triangle) and id.
static void walk(const Hittable* h, std::vector<float>* boxes, std::vector<int>* desc, std::vector<int>* prims) {
    switch (h->kindCode()) {
        case 0: {
            const BVHNode* n = (const BVHNode*)h;
            size_t me = desc->size();
            desc->push_back(0);
            const float b[6] = {n->box.minimum.x, n->box.minimum.y, n->box.minimum.z,
                                n->box.maximum.x, n->box.maximum.y, n->box.maximum.z};
            boxes->insert(boxes->end(), b, b + 6);
            walk(n->left.get(), boxes, desc, prims);
            if (n->left.get() != n->right.get()) walk(n->right.get(), boxes, desc, prims);
            (*desc)[me] = (int)(desc->size() - me - 1);
            break;
        }
        case 1:
            for (auto& o : ((const HittableList*)h)->objects) walk(o.get(), boxes, desc, prims);
            break;
        case 2: {
            const Mesh* m = (const Mesh*)h;
            size_t me = desc->size();
            desc->push_back(0);
            const float b[6] = {m->boundingBox.minimum.x, m->boundingBox.minimum.y, m->boundingBox.minimum.z,
                                m->boundingBox.maximum.x, m->boundingBox.maximum.y, m->boundingBox.maximum.z};
            boxes->insert(boxes->end(), b, b + 6);
            for (auto& o : m->triangles) walk(o.get(), boxes, desc, prims);
            (*desc)[me] = (int)(desc->size() - me - 1);
            break;
        }
        case 3:
            prims->push_back(0);
            prims->push_back(((const Sphere*)h)->id);
            break;
        case 4:
            prims->push_back(1);
            prims->push_back(((const Triangle*)h)->id);
            break;
    }
}
// Call with null outputs to get the counts, then again with buffers.
void orc_dump_tree(void* s, int* nBoxes, int* nPrims, float* boxes6, int* descendants, int* primKindId) {
    Scene& sc = *(Scene*)s;
    std::vector<float> b;
    std::vector<int> d, p;
    walk(&sc.world, &b, &d, &p);
    *nBoxes = (int)d.size();
    *nPrims = (int)p.size() / 2;
    if (boxes6) std::memcpy(boxes6, b.data(), b.size() * sizeof(float));
    if (descendants) std::memcpy(descendants, d.data(), d.size() * sizeof(int));
    if (primKindId) std::memcpy(primKindId, p.data(), p.size() * sizeof(int));
}

// Closest hit of n rays (6 floats each: origin, direction) against the world with the
// renderer's tMin = 0.001, tMax = inf.  primKind -1 = miss.
void orc_trace(void* s, int n, const float* rays, int* primKind, int* primId, float* t, float* normal3, int* frontFace,
               float* uv2) {
    Scene& sc = *(Scene*)s;
    Rng ctx;
    for (int i = 0; i < n; i++) {
        Ray r{vec3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]), vec3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5])};
        HitRecord rec;
        bool hit = sc.world.Hit(r, 0.001f, kInfinity, rec, ctx);
        primKind[i] = hit ? rec.primKind : -1;
        primId[i] = hit ? rec.primId : -1;
        t[i] = hit ? rec.t : kInfinity;
        if (normal3) {
            normal3[3 * i] = hit ? rec.normal.x : 0;
            normal3[3 * i + 1] = hit ? rec.normal.y : 0;
            normal3[3 * i + 2] = hit ? rec.normal.z : 0;
        }
        if (frontFace) frontFace[i] = hit ? (rec.frontFace ? 1 : 0) : 0;
        if (uv2) {
            uv2[2 * i] = hit ? rec.u : 0;
            uv2[2 * i + 1] = hit ? rec.v : 0;
        }
    }
}

// One scatter event with injected geometry; the random draws come from the hash stream
// keyed (px, py, sample) starting at counter0.  Returns alive flag; outputs attenuation and ray.
int orc_scatter(void* s, int mat, const float* rayIn6, const float* p3, const float* n3, int frontFace, float u, float v,
                uint32_t px, uint32_t py, int32_t sample, uint32_t counter0, float* atten3, float* rayOut6,
                uint32_t* counterOut) {
    Scene& sc = *(Scene*)s;
    Rng rng;
    rng.mode = Rng::HASH;
    rng.px = px;
    rng.py = py;
    rng.sample = sample;
    rng.counter = counter0;
    HitRecord rec;
    rec.p = vec3(p3[0], p3[1], p3[2]);
    rec.normal = vec3(n3[0], n3[1], n3[2]);
    rec.frontFace = frontFace != 0;
    rec.u = u;
    rec.v = v;
    Ray in{vec3(rayIn6[0], rayIn6[1], rayIn6[2]), vec3(rayIn6[3], rayIn6[4], rayIn6[5])};
    Ray out{vec3(), vec3()};
    vec3 att;
    bool alive = sc.materials[mat]->scatter(in, rec, att, out, rng);
    atten3[0] = att.x;
    atten3[1] = att.y;
    atten3[2] = att.z;
    rayOut6[0] = out.origin.x;
    rayOut6[1] = out.origin.y;
    rayOut6[2] = out.origin.z;
    rayOut6[3] = out.direction.x;
    rayOut6[4] = out.direction.y;
    rayOut6[5] = out.direction.z;
    *counterOut = rng.counter;
    return alive ? 1 : 0;
}

void orc_texture_at(void* s, int tex, int n, const float* uv, float* rgb) {
    Scene& sc = *(Scene*)s;
    for (int i = 0; i < n; i++) {
        vec3 c = sc.textures[tex]->At(uv[2 * i], uv[2 * i + 1]);
        rgb[3 * i] = c.x;
        rgb[3 * i + 1] = c.y;
        rgb[3 * i + 2] = c.z;
    }
}
// Dielectric::reflectance for n (cosine, refraction ratio) pairs
void orc_reflectance(int n, const float* cosine, const float* refIdx, float* out) {
    for (int i = 0; i < n; i++) out[i] = Material::reflectance(cosine[i], refIdx[i]);
}

// the 256-entry byte -> float table a Texture applies
void orc_texture_lut(float* out256) {
    const uint8_t px[3] = {0, 0, 0};
    Texture t(1, 1, px);
    for (int i = 0; i < 256; i++) out256[i] = t.lut[i];
}

void orc_write_color(int n, const float* sums3, int samplesPerPixel, uint8_t* rgba) {
    for (int i = 0; i < n; i++) writeColorBytes(vec3(sums3[3 * i], sums3[3 * i + 1], sums3[3 * i + 2]), samplesPerPixel, rgba + 4 * i);
}

// Render.  rngMode 0: the scene's sequential mt19937 stream, loops exactly as
//   RaytracerNormal::Run with buildUp = false (Raytracer.cpp:74-94); single thread only.
// rngMode 1: hash stream keyed (i, j, sampleIndex, counter); pixels are independent, so rows
//   may be split across `threads` host threads without changing any result.
// order 0: recursive estimator (reference evaluation order); 1: front-to-back throughput order.
// accumRGBA (W*H*4 floats, row j at offset j*W, row 0 = bottom) receives the float sums
// (alpha = sample count); rgba8 the writeColor bytes.  Rows [row0, row1) only.
// The rays of one hash-stream path (pixel, sample), one after the other, and what each of them hit:
// rays6[k] = origin, direction; hits[k] = {primitive kind (-1 none, 0 sphere, 1 triangle), id}; t[k].
// Returns the number of rays.  For narrowing down a difference between two renderers to one ray.
int orc_path_rays(void* s, int W, int H, int px, int py, int sample, int maxDepth, int maxRays, float* rays6, int* hits2,
                  float* t) {
    const Scene& sc = *(Scene*)s;
    Rng rng;
    rng.mode = Rng::HASH;
    rng.px = (uint32_t)px;
    rng.py = (uint32_t)py;
    rng.sample = sample;
    rng.counter = 0;
    float u = (px + rng.randomFloat()) / (W - 1);
    float v = (py + rng.randomFloat()) / (H - 1);
    Ray r = sc.camera.GetRay(u, v, rng);
    int n = 0;
    for (int depth = maxDepth; depth > 0 && n < maxRays; depth--) {
        HitRecord rec;
        const bool hit = sc.world.Hit(r, 0.001f, kInfinity, rec, rng);
        rays6[6 * n] = r.origin.x; rays6[6 * n + 1] = r.origin.y; rays6[6 * n + 2] = r.origin.z;
        rays6[6 * n + 3] = r.direction.x; rays6[6 * n + 4] = r.direction.y; rays6[6 * n + 5] = r.direction.z;
        hits2[2 * n] = hit ? rec.primKind : -1;
        hits2[2 * n + 1] = hit ? rec.primId : -1;
        t[n] = hit ? rec.t : kInfinity;
        n++;
        if (!hit) break;
        Ray scattered;
        vec3 attenuation;
        if (!rec.mat->scatter(r, rec, attenuation, scattered, rng)) break;
        r = scattered;
    }
    return n;
}

void orc_render(void* s, int W, int H, int firstSample, int nSamples, int maxDepth, int rngMode, int order, int threads,
                int row0, int row1, float* accumRGBA, uint8_t* rgba8, OrcCounters* countersOut) {
    Scene& sc = *(Scene*)s;
    if (row1 > H) row1 = H;
    if (rngMode == 0) threads = 1;
    if (threads < 1) threads = 1;
    std::vector<Counters> cnts(threads);
    std::atomic<int> nextRow(row1 - 1);
    auto work = [&](int tid) {
        Rng rng;
        if (rngMode == 0) {
            rng = sc.rng;  // shares the scene's generator pointer
        } else {
            rng.mode = Rng::HASH;
        }
        rng.cnt = countersOut ? &cnts[tid] : nullptr;
        for (;;) {
            int j = nextRow.fetch_sub(1);
            if (j < row0) break;
            for (int i = 0; i < W; ++i) {
                vec3 pixelColor(0.0f, 0.0f, 0.0f);
                size_t px = (size_t)j * W + i;
                if (accumRGBA && firstSample > 0) pixelColor = vec3(accumRGBA[4 * px], accumRGBA[4 * px + 1], accumRGBA[4 * px + 2]);
                for (int sIdx = firstSample; sIdx < firstSample + nSamples; ++sIdx) {
                    if (rngMode == 1) {
                        rng.px = (uint32_t)i;
                        rng.py = (uint32_t)j;
                        rng.sample = sIdx;
                        rng.counter = 0;
                    }
                    if (rng.cnt) rng.cnt->samples++;
                    float u = (i + rng.randomFloat()) / (W - 1);
                    float v = (j + rng.randomFloat()) / (H - 1);
                    Ray r = sc.camera.GetRay(u, v, rng);
                    vec3 c = order == 0 ? rayColor(sc, r, maxDepth, rng) : rayColorThroughput(sc, r, maxDepth, rng);
                    pixelColor = pixelColor + c;
                }
                if (accumRGBA) {
                    accumRGBA[4 * px] = pixelColor.x;
                    accumRGBA[4 * px + 1] = pixelColor.y;
                    accumRGBA[4 * px + 2] = pixelColor.z;
                    accumRGBA[4 * px + 3] = (float)(firstSample + nSamples);
                }
                if (rgba8) writeColorBytes(pixelColor, firstSample + nSamples, rgba8 + 4 * px);
            }
        }
    };
    if (threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(work, t);
        for (auto& t : pool) t.join();
    }
    if (countersOut) {
        Counters total;
        for (auto& c : cnts) total.add(c);
        countersOut->samples = total.samples;
        countersOut->rays = total.rays;
        countersOut->nodeTests = total.nodeTests;
        countersOut->sphereTests = total.sphereTests;
        countersOut->triTests = total.triTests;
        countersOut->hits = total.hits;
        countersOut->scatterLambert = total.scatterLambert;
        countersOut->scatterMetal = total.scatterMetal;
        countersOut->scatterDielectric = total.scatterDielectric;
        countersOut->scatterLight = total.scatterLight;
        countersOut->scatterPbr = total.scatterPbr;
        countersOut->misses = total.misses;
        countersOut->depthCutoffs = total.depthCutoffs;
    }
}

}  // extern "C"
