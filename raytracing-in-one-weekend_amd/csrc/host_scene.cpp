// Host-side scene construction, reference-order BVH build and the flatten step.
// All arithmetic is IEEE binary32 evaluated in the order the reference's expressions
// evaluate (compile with -ffp-contract=off).  glm is not available to check against;
// the operation orders used for dot / cross / normalize / mat4*vec4 / inverse are glm
// 0.9.9's scalar code paths as commonly documented.
// R/ = /root/reference/Raytracing-In-A-Weekend/.
#include <chrono>
#include <cstdio>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include "host_scene.h"
#include "image_decode.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <sstream>

namespace rt {

namespace {

const float kInf = std::numeric_limits<float>::infinity();

inline float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
inline void cross3(const float* a, const float* b, float* o) {
    float x = a[1] * b[2] - b[1] * a[2];
    float y = a[2] * b[0] - b[2] * a[0];
    float z = a[0] * b[1] - b[0] * a[1];
    o[0] = x;
    o[1] = y;
    o[2] = z;
}
inline void normalize3(const float* v, float* o) {
    float inv = 1.0f / std::sqrt(dot3(v, v));
    o[0] = v[0] * inv;
    o[1] = v[1] * inv;
    o[2] = v[2] * inv;
}

// Column-major 4x4 helpers: m[col*4 + row].
inline void mulPoint(const float* m, const float* p, float w, float* out3) {
    for (int r = 0; r < 3; r++) {
        float a = m[0 * 4 + r] * p[0], b = m[1 * 4 + r] * p[1];
        float c = m[2 * 4 + r] * p[2], d = m[3 * 4 + r] * w;
        out3[r] = (a + b) + (c + d);
    }
}

// Inverse by cofactors scaled with the reciprocal determinant (glm::inverse for mat4).
void inverse4(const float* src, float* dst) {
    auto m = [&](int c, int r) { return src[c * 4 + r]; };
    float c00 = m(2, 2) * m(3, 3) - m(3, 2) * m(2, 3);
    float c02 = m(1, 2) * m(3, 3) - m(3, 2) * m(1, 3);
    float c03 = m(1, 2) * m(2, 3) - m(2, 2) * m(1, 3);
    float c04 = m(2, 1) * m(3, 3) - m(3, 1) * m(2, 3);
    float c06 = m(1, 1) * m(3, 3) - m(3, 1) * m(1, 3);
    float c07 = m(1, 1) * m(2, 3) - m(2, 1) * m(1, 3);
    float c08 = m(2, 1) * m(3, 2) - m(3, 1) * m(2, 2);
    float c10 = m(1, 1) * m(3, 2) - m(3, 1) * m(1, 2);
    float c11 = m(1, 1) * m(2, 2) - m(2, 1) * m(1, 2);
    float c12 = m(2, 0) * m(3, 3) - m(3, 0) * m(2, 3);
    float c14 = m(1, 0) * m(3, 3) - m(3, 0) * m(1, 3);
    float c15 = m(1, 0) * m(2, 3) - m(2, 0) * m(1, 3);
    float c16 = m(2, 0) * m(3, 2) - m(3, 0) * m(2, 2);
    float c18 = m(1, 0) * m(3, 2) - m(3, 0) * m(1, 2);
    float c19 = m(1, 0) * m(2, 2) - m(2, 0) * m(1, 2);
    float c20 = m(2, 0) * m(3, 1) - m(3, 0) * m(2, 1);
    float c22 = m(1, 0) * m(3, 1) - m(3, 0) * m(1, 1);
    float c23 = m(1, 0) * m(2, 1) - m(2, 0) * m(1, 1);
    const float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
    const float f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    const float v0[4] = {m(1, 0), m(0, 0), m(0, 0), m(0, 0)}, v1[4] = {m(1, 1), m(0, 1), m(0, 1), m(0, 1)};
    const float v2[4] = {m(1, 2), m(0, 2), m(0, 2), m(0, 2)}, v3[4] = {m(1, 3), m(0, 3), m(0, 3), m(0, 3)};
    const float sa[4] = {+1, -1, +1, -1}, sb[4] = {-1, +1, -1, +1};
    for (int i = 0; i < 4; i++) {
        dst[0 * 4 + i] = ((v1[i] * f0[i] - v2[i] * f1[i]) + v3[i] * f2[i]) * sa[i];
        dst[1 * 4 + i] = ((v0[i] * f0[i] - v2[i] * f3[i]) + v3[i] * f4[i]) * sb[i];
        dst[2 * 4 + i] = ((v0[i] * f1[i] - v1[i] * f3[i]) + v3[i] * f5[i]) * sa[i];
        dst[3 * 4 + i] = ((v0[i] * f2[i] - v1[i] * f4[i]) + v2[i] * f5[i]) * sb[i];
    }
    float det = (src[0] * dst[0 * 4 + 0] + src[1] * dst[1 * 4 + 0]) + (src[2] * dst[2 * 4 + 0] + src[3] * dst[3 * 4 + 0]);
    float ood = 1.0f / det;
    for (int i = 0; i < 16; i++) dst[i] = dst[i] * ood;
}

inline Box3 unionBox(const Box3& a, const Box3& b) {  // surroundingBox, R/src/AccelerationStructures/AABB.cpp:3-12
    Box3 o;
    for (int k = 0; k < 3; k++) {
        o.mn[k] = std::fmin(a.mn[k], b.mn[k]);
        o.mx[k] = std::fmax(a.mx[k], b.mx[k]);
    }
    return o;
}

// Triangle::BoundingBox, R/src/Core/Hittable.h:143-213: the maximum starts at 1e-8 (not -inf)
// and an axis with min == max is widened by one unit each way.
Box3 triangleBox(const HostTriangle& t) {
    Box3 b;
    for (int a = 0; a < 3; a++) {
        float mn = kInf, mx = 1e-8f;
        for (int k = 0; k < 3; k++)
            if (t.pos[3 * k + a] < mn) mn = t.pos[3 * k + a];
        for (int k = 0; k < 3; k++)
            if (t.pos[3 * k + a] > mx) mx = t.pos[3 * k + a];
        if (mn == mx) {
            mn--;
            mx++;
        }
        b.mn[a] = mn;
        b.mx[a] = mx;
    }
    return b;
}

inline uint32_t bitsOf(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

// ---------------------------------------------------------------- minimal JSON (glTF) reader
struct JValue {
    enum T { NUL, BOOL, NUM, STR, ARR, OBJ } t = NUL;
    double num = 0;
    bool b = false;
    std::string str;
    std::vector<JValue> arr;
    std::map<std::string, JValue> obj;
    const JValue* get(const char* k) const {
        auto it = obj.find(k);
        return it == obj.end() ? nullptr : &it->second;
    }
};
struct JParser {
    const std::string& s;
    size_t i = 0;
    bool ok = true;
    explicit JParser(const std::string& src) : s(src) {}
    void ws() {
        while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) i++;
    }
    JValue parse() {
        ws();
        JValue v;
        if (i >= s.size()) {
            ok = false;
            return v;
        }
        char c = s[i];
        if (c == '{') {
            v.t = JValue::OBJ;
            i++;
            ws();
            if (i < s.size() && s[i] == '}') {
                i++;
                return v;
            }
            while (ok) {
                ws();
                JValue k = parse();
                if (k.t != JValue::STR) {
                    ok = false;
                    break;
                }
                ws();
                if (i >= s.size() || s[i] != ':') {
                    ok = false;
                    break;
                }
                i++;
                v.obj[k.str] = parse();
                ws();
                if (i < s.size() && s[i] == ',') {
                    i++;
                    continue;
                }
                if (i < s.size() && s[i] == '}') {
                    i++;
                    break;
                }
                ok = false;
            }
        } else if (c == '[') {
            v.t = JValue::ARR;
            i++;
            ws();
            if (i < s.size() && s[i] == ']') {
                i++;
                return v;
            }
            while (ok) {
                v.arr.push_back(parse());
                ws();
                if (i < s.size() && s[i] == ',') {
                    i++;
                    continue;
                }
                if (i < s.size() && s[i] == ']') {
                    i++;
                    break;
                }
                ok = false;
            }
        } else if (c == '"') {
            v.t = JValue::STR;
            i++;
            while (i < s.size() && s[i] != '"') {
                if (s[i] == '\\' && i + 1 < s.size()) {
                    i++;
                    char e = s[i];
                    v.str += (e == 'n' ? '\n' : e == 't' ? '\t' : e);
                } else {
                    v.str += s[i];
                }
                i++;
            }
            if (i >= s.size()) ok = false;
            i++;
        } else if (c == 't' && s.compare(i, 4, "true") == 0) {
            v.t = JValue::BOOL;
            v.b = true;
            i += 4;
        } else if (c == 'f' && s.compare(i, 5, "false") == 0) {
            v.t = JValue::BOOL;
            i += 5;
        } else if (c == 'n' && s.compare(i, 4, "null") == 0) {
            i += 4;
        } else {
            size_t j = i;
            while (j < s.size() && (std::isdigit((unsigned char)s[j]) || s[j] == '-' || s[j] == '+' || s[j] == '.' ||
                                    s[j] == 'e' || s[j] == 'E'))
                j++;
            if (j == i) {
                ok = false;
                return v;
            }
            v.t = JValue::NUM;
            v.num = std::strtod(s.substr(i, j - i).c_str(), nullptr);
            i = j;
        }
        return v;
    }
};

bool readFile(const std::string& path, std::string* out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::ostringstream ss;
    ss << f.rdbuf();
    *out = ss.str();
    return true;
}

}  // namespace

// ---------------------------------------------------------------------------- RNG
Scene::Scene() : gen_() {}  // std::mt19937 default seed 5489, as `static std::mt19937 generator;`

// uniform_real_distribution<float>(0,1) over mt19937 in libstdc++ is one 32-bit draw scaled
// by 2^-32 in float, with results that round up to 1 replaced by the float just below 1
// (R/src/Core/RTWeekend.h:26-31).
float Scene::randomFloat() {
    uint32_t r = (uint32_t)gen_();
    float sum = (float)r;             // rounds to 24 bits
    float ret = sum / 4294967296.0f;  // exact scaling
    if (ret >= 1.0f) ret = std::nextafter(1.0f, 0.0f);
    return ret;
}
// randomVec(): the three constructor arguments are evaluated right to left by g++, the
// compiler that defines the canonical scene: z first (R/src/Core/RTWeekend.h:45-53).
void Scene::randomVec(float out[3]) {
    out[2] = randomFloat();
    out[1] = randomFloat();
    out[0] = randomFloat();
}
void Scene::randomVec(float mn, float mx, float out[3]) {
    out[2] = randomFloat(mn, mx);
    out[1] = randomFloat(mn, mx);
    out[0] = randomFloat(mn, mx);
}

// ---------------------------------------------------------------------------- building blocks
int Scene::addMaterial(int kind, float r, float g, float b, float param) {
    if (kind == 1) param = param < 1 ? param : 1;  // Metal clamps fuzz, R/src/Material/Material.h:45
    materials.push_back({kind, r, g, b, param, -1, -1, -1});
    flat.valid = false;
    return (int)materials.size() - 1;
}
int Scene::addTexture(int w, int h, const uint8_t* rgb) {
    HostTexture t;
    t.w = w;
    t.h = h;
    t.rgb.assign(rgb, rgb + (size_t)w * h * 3);
    textures.push_back(std::move(t));
    flat.valid = false;
    return (int)textures.size() - 1;
}
int Scene::addPbr(int td, int tr, int tn) {
    materials.push_back({4, 0, 0, 0, 0, td, tr, tn});
    flat.valid = false;
    return (int)materials.size() - 1;
}
int Scene::addSphere(float cx, float cy, float cz, float r, int mat) {
    spheres.push_back({{cx, cy, cz}, r, mat});
    place({Obj::SPHERE, (int)spheres.size() - 1});
    flat.valid = false;
    return (int)spheres.size() - 1;
}
int Scene::newTriangle(const float* pos9, const float* nrm9, const float* uv6, int mat) {
    HostTriangle t;
    std::memcpy(t.pos, pos9, sizeof t.pos);
    std::memcpy(t.nrm, nrm9, sizeof t.nrm);
    std::memcpy(t.uv, uv6, sizeof t.uv);
    t.mat = mat;
    triangles.push_back(t);
    flat.valid = false;
    return (int)triangles.size() - 1;
}
int Scene::addTriangle(const float* pos9, const float* nrm9, const float* uv6, int mat) {
    int id = newTriangle(pos9, nrm9, uv6, mat);
    place({Obj::TRIANGLE, id});
    return id;
}

// Box::Box, R/src/Core/Hittable.h:322-385: twelve triangles with zero normals from the eight
// transformed corners, in the reference's order.  Box::BoundingBox is (p0, p1) untransformed.
int Scene::addBox(const float* p0, const float* p1, int mat, const float* model16) {
    // corner selector per vertex: 0 takes the coordinate from p0, 1 from p1
    static const int sel[12][3][3] = {
        {{0, 0, 0}, {0, 0, 1}, {0, 1, 0}}, {{0, 0, 1}, {0, 1, 1}, {0, 1, 0}},  // left
        {{1, 0, 0}, {1, 0, 1}, {1, 1, 0}}, {{1, 0, 1}, {1, 1, 1}, {1, 1, 0}},  // right
        {{0, 0, 0}, {1, 0, 0}, {0, 0, 1}}, {{1, 0, 0}, {1, 0, 1}, {0, 0, 1}},  // floor
        {{0, 1, 0}, {1, 1, 0}, {0, 1, 1}}, {{1, 1, 0}, {1, 1, 1}, {0, 1, 1}},  // top
        {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}}, {{1, 0, 0}, {1, 1, 0}, {0, 1, 0}},  // front
        {{1, 1, 1}, {0, 1, 1}, {0, 0, 1}}, {{1, 1, 1}, {0, 0, 1}, {1, 0, 1}},  // back
    };
    Composite c;
    c.kind = Composite::LIST;
    c.has_box_test = false;
    for (int k = 0; k < 3; k++) {
        c.box.mn[k] = p0[k];
        c.box.mx[k] = p1[k];
    }
    int first = -1;
    for (int f = 0; f < 12; f++) {
        float pos[9], nrm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, uv[6] = {0, 0, 0, 0, 0, 0};
        for (int v = 0; v < 3; v++) {
            float p[3];
            for (int k = 0; k < 3; k++) p[k] = sel[f][v][k] ? p1[k] : p0[k];
            mulPoint(model16, p, 1.0f, &pos[3 * v]);
        }
        int id = newTriangle(pos, nrm, uv, mat);
        if (first < 0) first = id;
        c.objs.push_back({Obj::TRIANGLE, id});
    }
    composites.push_back(std::move(c));
    place({Obj::COMPOSITE, (int)composites.size() - 1});
    return first;
}

void Scene::place(const Obj& o) {
    if (!open_groups_.empty())
        composites[open_groups_.back()].objs.push_back(o);
    else if (open_prototype_ >= 0)
        prototypes[open_prototype_].objs.push_back(o);
    else
        world.push_back(o);
    flat.valid = false;
}

int Scene::beginPrototype() {
    if (open_prototype_ >= 0 || !open_groups_.empty()) return -1;
    prototypes.emplace_back();
    open_prototype_ = (int)prototypes.size() - 1;
    return open_prototype_;
}

int Scene::endPrototype() {
    if (open_prototype_ < 0 || !open_groups_.empty()) return -1;
    const int id = open_prototype_;
    open_prototype_ = -1;
    return id;
}

int Scene::addInstance(int prototype, const float* model16) {
    if (prototype < 0 || prototype >= (int)prototypes.size() || prototype == open_prototype_) return -1;
    const Prototype& p = prototypes[prototype];
    if (p.objs.empty()) return -1;
    for (const Obj& o : p.objs)
        if (o.kind == Obj::INSTANCE) return -1;  // one level of instancing
    Instance in;
    in.prototype = prototype;
    std::memcpy(in.model, model16, sizeof in.model);
    inverse4(model16, in.inv);
    // HittableList::BoundingBox over the prototype's objects (R/src/Core/Hittable.h:298-313), then its eight
    // corners under the model matrix
    Box3 ob = objectBox(p.objs[0]);
    for (size_t i = 1; i < p.objs.size(); i++) ob = unionBox(ob, objectBox(p.objs[i]));
    for (int c = 0; c < 8; c++) {
        const float corner[3] = {(c & 1) ? ob.mx[0] : ob.mn[0], (c & 2) ? ob.mx[1] : ob.mn[1], (c & 4) ? ob.mx[2] : ob.mn[2]};
        float w[3];
        mulPoint(model16, corner, 1.0f, w);
        for (int k = 0; k < 3; k++) {
            if (c == 0 || w[k] < in.box.mn[k]) in.box.mn[k] = w[k];
            if (c == 0 || w[k] > in.box.mx[k]) in.box.mx[k] = w[k];
        }
    }
    instances.push_back(in);
    place({Obj::INSTANCE, (int)instances.size() - 1});
    return (int)instances.size() - 1;
}

int Scene::beginGroup(bool boxed, const float* box6) {
    Composite c;
    c.kind = boxed ? Composite::MESH : Composite::LIST;  // MESH: box test, then the children in order
    c.has_box_test = boxed;
    for (int k = 0; k < 3; k++) {
        c.box.mn[k] = boxed ? box6[k] : 0.0f;
        c.box.mx[k] = boxed ? box6[3 + k] : 0.0f;
    }
    composites.push_back(std::move(c));
    const int id = (int)composites.size() - 1;
    place({Obj::COMPOSITE, id});
    open_groups_.push_back(id);
    return id;
}

int Scene::endGroup() {
    if (open_groups_.empty()) return -1;
    open_groups_.pop_back();
    return 0;
}

Box3 Scene::objectBox(const Obj& o) const {
    Box3 b;
    if (o.kind == Obj::SPHERE) {  // Sphere::BoundingBox, R/src/Core/Hittable.h:257-264
        const HostSphere& s = spheres[o.id];
        for (int k = 0; k < 3; k++) {
            b.mn[k] = s.c[k] - s.r;
            b.mx[k] = s.c[k] + s.r;
        }
        return b;
    }
    if (o.kind == Obj::TRIANGLE) return triangleBox(triangles[o.id]);
    if (o.kind == Obj::INSTANCE) return instances[o.id].box;
    return composites[o.id].box;
}

// BVHNode::BVHNode, R/src/AccelerationStructures/Bvh.h:18-78.  One randomInt(0,2) per node in
// pre-order (leaves included); std::sort on the node's own sub-range (a child only ever reads
// its own sub-range of the parent's sorted copy, so sorting in place is equivalent);
// mid = start + span/2; max_depth == 0 turns the halves into plain lists.
// The objects travel with their bounding boxes, computed once: boxCompare (Bvh.cpp:3-12) calls
// BoundingBox() on both operands of every comparison, and std::sort's permutation depends only
// on the outcomes of the comparisons, which are the same on the stored boxes.
// Number of BVHNode objects the constructor creates for `span` objects with `max_depth` levels left
// (Bvh.h:17-67): one for the node itself, and two subtrees unless the span is at most 2 or the
// depth limit makes the halves plain lists.  Spans on one level differ by at most one, so the
// recursion with a memo visits O(levels) distinct arguments.
static size_t bvhNodeCount(size_t span, int max_depth) {
    thread_local std::map<std::pair<size_t, int>, size_t> memo;
    if (span <= 2 || max_depth == 0) return 1;
    const int d = max_depth < 0 ? -1 : max_depth;
    auto key = std::make_pair(span, d);
    auto it = memo.find(key);
    if (it != memo.end()) return it->second;
    const size_t n = 1 + bvhNodeCount(span / 2, max_depth - 1) + bvhNodeCount(span - span / 2, max_depth - 1);
    memo[key] = n;
    return n;
}

// BVHNode::BVHNode (Bvh.h:17-67) over objs[start, end).  The constructor draws its axis before it
// recurses (left subtree first), so node k in pre-order uses the k-th draw: the draws are made up
// front, in order, from the scene's generator (which ends in the same state), and the node with
// pre-order index `me` reads axes[me].  Everything else a node does touches only its own range of
// `objs` and its own slots of c.nodes, so the two halves can be built by different threads; the
// result does not depend on how the work is spread.
void Scene::buildBvhAt(Composite& c, std::vector<BoxedObj>& objs, size_t start, size_t end, int max_depth, size_t me,
                       const std::vector<uint8_t>& axes, const std::atomic<size_t>& drawn, std::atomic<int>& spare_threads) {
    while (drawn.load(std::memory_order_acquire) <= me) std::this_thread::yield();  // (see buildBvh)
    const int axis = axes[me];
    auto less = [axis](const BoxedObj& a, const BoxedObj& b) { return a.box.mn[axis] < b.box.mn[axis]; };
    size_t span = end - start;
    Box3 bl, br;
    if (span == 1) {
        // left = right = object: Hit() tests it twice with the same outcome; kept once
        c.nodes[me].leaf = {objs[start].obj};
        bl = br = objs[start].box;
    } else if (span == 2) {
        BoxedObj a = objs[start], b = objs[start + 1];
        if (!less(a, b)) std::swap(a, b);
        c.nodes[me].leaf = {a.obj, b.obj};
        bl = a.box;
        br = b.box;
    } else {
        if (span >= 32768)
            sortLikeStdSort(objs.data() + start, objs.data() + end, less, spare_threads);
        else
            std::sort(objs.begin() + start, objs.begin() + end, less);
        size_t mid = start + span / 2;
        if (max_depth == 0) {
            // two HittableLists; HittableList::BoundingBox folds surroundingBox left to right
            std::vector<Obj> leaf;
            leaf.reserve(span);
            for (size_t i = start; i < end; i++) leaf.push_back(objs[i].obj);
            bl = objs[start].box;
            for (size_t i = start + 1; i < mid; i++) bl = unionBox(bl, objs[i].box);
            br = objs[mid].box;
            for (size_t i = mid + 1; i < end; i++) br = unionBox(br, objs[i].box);
            c.nodes[me].leaf = std::move(leaf);
        } else {
            const size_t l = me + 1;
            const size_t r = l + bvhNodeCount(mid - start, max_depth - 1);
            std::thread other;
            if (span >= 32768 && spare_threads.fetch_sub(1) > 0 &&
                tryThread(other, [&] { buildBvhAt(c, objs, start, mid, max_depth - 1, l, axes, drawn, spare_threads); })) {
                buildBvhAt(c, objs, mid, end, max_depth - 1, r, axes, drawn, spare_threads);
                other.join();
                spare_threads.fetch_add(1);
            } else {
                if (span >= 32768) spare_threads.fetch_add(1);  // nothing was taken
                buildBvhAt(c, objs, start, mid, max_depth - 1, l, axes, drawn, spare_threads);
                buildBvhAt(c, objs, mid, end, max_depth - 1, r, axes, drawn, spare_threads);
            }
            c.nodes[me].left = (int)l;
            c.nodes[me].right = (int)r;
            bl = c.nodes[l].box;
            br = c.nodes[r].box;
        }
    }
    c.nodes[me].box = unionBox(bl, br);
}

int Scene::buildBvh(Composite& c, std::vector<BoxedObj>& objs, size_t start, size_t end, int max_depth) {
    const size_t total = bvhNodeCount(end - start, max_depth);
    // The draws of a large tree (13 ns each, one thread: the generator is sequential) are made by a thread of their
    // own while the builders sort; a node waits for its draw if it gets there first, which the top-level sort all
    // but rules out.  Nobody else touches the generator meanwhile.
    const size_t base = c.nodes.size();
    std::vector<uint8_t> axes(base + total);  // indexed by node number, like c.nodes
    std::atomic<size_t> drawn(base);
    auto draw = [&]() {
        for (size_t i = 0; i < total; i++) {
            axes[base + i] = (uint8_t)randomInt(0, 2);
            if ((i & 8191) == 8191) drawn.store(base + i + 1, std::memory_order_release);
        }
        drawn.store(base + total, std::memory_order_release);
    };
    std::thread drawer;
    if (!(total >= 65536 && buildThreads() > 1 && tryThread(drawer, draw))) draw();
    c.nodes.resize(base + total);
    std::atomic<int> spare((int)buildThreads() - 1);
    buildBvhAt(c, objs, start, end, max_depth, base, axes, drawn, spare);
    if (drawer.joinable()) drawer.join();
    c.built_first = base;
    c.built_count = total;
    return (int)base;
}

// Threads the host-side builders may use: the machine's, at most 16 (RT_BUILD_THREADS overrides).
bool tangentFrame(const float* pos, const float* uv, float* tangent, float* bitangent) {
    for (int k = 0; k < 3; k++) tangent[k] = bitangent[k] = 0.0f;
    if (uv[0] == -1.0f) return false;
    const float du1 = uv[2] - uv[0], dv1 = uv[3] - uv[1], du2 = uv[4] - uv[0], dv2 = uv[5] - uv[1];
    const float det = du1 * dv2 - du2 * dv1;
    if (det == 0.0f || !std::isfinite(det)) return false;
    const float f = 1.0f / det;
    if (!std::isfinite(f)) return false;
    bool any = false;
    for (int k = 0; k < 3; k++) {
        const float e1 = pos[3 + k] - pos[k], e2 = pos[6 + k] - pos[k];
        tangent[k] = f * (dv2 * e1 - dv1 * e2);
        bitangent[k] = f * (-du2 * e1 + du1 * e2);
        any = any || tangent[k] != 0.0f;
    }
    bool anyB = false, finite = true;
    for (int k = 0; k < 3; k++) {
        anyB = anyB || bitangent[k] != 0.0f;
        finite = finite && std::isfinite(tangent[k]) && std::isfinite(bitangent[k]);
    }
    if (!(any && anyB && finite)) {
        for (int k = 0; k < 3; k++) tangent[k] = bitangent[k] = 0.0f;
        return false;
    }
    return true;
}

unsigned buildThreads() {
    if (const char* e = std::getenv("RT_BUILD_THREADS")) return (unsigned)std::max(1, std::atoi(e));
    if (const char* e = std::getenv("OMP_NUM_THREADS")) return (unsigned)std::min(16, std::max(1, std::atoi(e)));
    return std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
}

std::vector<Scene::BoxedObj> Scene::withBoxes(const std::vector<Obj>& objs) const {
    std::vector<BoxedObj> out(objs.size());
    parallelPieces(objs.size(), [&](size_t from, size_t to) {
        for (size_t i = from; i < to; i++) out[i] = {objs[i], objectBox(objs[i])};
    });
    return out;
}

// world = HittableList(make_shared<BVHNode>(world)): Bvh.h:17, unlimited depth.
void Scene::wrapBvh() {
    Composite c;
    c.kind = Composite::BVH;
    c.has_box_test = true;
    std::vector<BoxedObj> objs = withBoxes(world);
    c.root = buildBvh(c, objs, 0, objs.size(), -1);
    c.box = c.nodes[c.root].box;
    composites.push_back(std::move(c));
    world.clear();
    world.push_back({Obj::COMPOSITE, (int)composites.size() - 1});
    flat.valid = false;
}

// Mesh::Mesh + processMesh, R/src/Core/Mesh.cpp:7-37,106-208.
int Scene::addMesh(const float* model16, int n_vertices, const float* positions, const float* normals, const float* uvs,
                   int n_triangles, const uint32_t* indices, int mat, int max_depth, const int* tri_materials) {
    float inv[16];
    inverse4(model16, inv);
    Composite mesh;
    mesh.kind = Composite::MESH;
    mesh.has_box_test = true;
    float mn[3] = {kInf, kInf, kInf}, mx[3] = {1e-8f, 1e-8f, 1e-8f};  // Mesh.cpp:16-21
    std::vector<Obj> tris;
    const bool timing = std::getenv("RT_HOST_TIMING") && std::atoi(std::getenv("RT_HOST_TIMING")) != 0;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[add_mesh] %-41s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t0).count());
        t0 = now;
    };
    // an index beyond the vertices refuses the mesh before anything of it has been taken in
    {
        std::atomic<bool> bad{false};
        parallelPieces((size_t)n_triangles * 3, [&](size_t from, size_t to) {
            for (size_t i = from; i < to; i++)
                if (indices[i] >= (uint32_t)n_vertices) bad = true;
        });
        if (bad) return -1;
    }
    // The faces are independent of each other (pieces of them go to the builders' threads); the mesh's box is a
    // minimum and a maximum, which come out the same in any order.
    const size_t base = triangles.size();
    const int first = n_triangles > 0 ? (int)base : -1;
    triangles.resize(base + (size_t)n_triangles);
    tris.resize((size_t)n_triangles);
    flat.valid = false;
    std::mutex fold;
    parallelPieces((size_t)n_triangles, [&](size_t from, size_t to) {
      float lmn[3] = {kInf, kInf, kInf}, lmx[3] = {1e-8f, 1e-8f, 1e-8f};
      for (size_t f = from; f < to; f++) {
        float pos[9], nrm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        float uv[6] = {0.0f, 0.0f, 0.1f, 0.0f, 0.5f, 1.0f};  // Mesh.cpp:122-124 defaults
        for (int k = 0; k < 3; k++) {
            uint32_t vi = indices[3 * f + k];
            mulPoint(model16, &positions[3 * vi], 1.0f, &pos[3 * k]);  // Mesh.cpp:145-147
            if (uvs) {
                uv[2 * k] = uvs[2 * vi];
                uv[2 * k + 1] = uvs[2 * vi + 1];
            }
            if (normals) {
                float n[3];
                normalize3(&normals[3 * vi], n);  // Mesh.cpp:139-141
                // mat3(transpose(inverse(M))) * n: column j of that matrix is row j of inv (Mesh.cpp:144,148-150)
                for (int r = 0; r < 3; r++)
                    nrm[3 * k + r] = (inv[r * 4 + 0] * n[0] + inv[r * 4 + 1] * n[1]) + inv[r * 4 + 2] * n[2];
            }
        }
        HostTriangle& t = triangles[base + f];
        std::memcpy(t.pos, pos, sizeof t.pos);
        std::memcpy(t.nrm, nrm, sizeof t.nrm);
        std::memcpy(t.uv, uv, sizeof t.uv);
        t.mat = tri_materials ? tri_materials[f] : mat;
        tris[f] = {Obj::TRIANGLE, (int)(base + f)};
        for (int a = 0; a < 3; a++) {  // Mesh.cpp:165-207
            for (int k = 0; k < 3; k++)
                if (pos[3 * k + a] < lmn[a]) lmn[a] = pos[3 * k + a];
            for (int k = 0; k < 3; k++)
                if (pos[3 * k + a] > lmx[a]) lmx[a] = pos[3 * k + a];
        }
      }
      std::lock_guard<std::mutex> g(fold);
      for (int a = 0; a < 3; a++) {
          if (lmn[a] < mn[a]) mn[a] = lmn[a];
          if (lmx[a] > mx[a]) mx[a] = lmx[a];
      }
    });
    for (int a = 0; a < 3; a++) {
        mesh.box.mn[a] = mn[a];
        mesh.box.mx[a] = mx[a];
    }
    // BVHNode(triangles, 0, n, maxDepth); triangles = { bvh }  (Mesh.cpp:29-31)
    Composite bvh;
    bvh.kind = Composite::BVH;
    bvh.has_box_test = true;
    lap("triangles (Mesh::processMesh)");
    std::vector<BoxedObj> boxed = withBoxes(tris);
    lap("their boxes");
    bvh.root = buildBvh(bvh, boxed, 0, boxed.size(), max_depth);
    lap("BVHNode over them");
    bvh.box = bvh.nodes[bvh.root].box;
    composites.push_back(std::move(bvh));
    mesh.objs.push_back({Obj::COMPOSITE, (int)composites.size() - 1});
    composites.push_back(std::move(mesh));
    place({Obj::COMPOSITE, (int)composites.size() - 1});
    return first;
}

// glTF 2.0 reader (the job Assimp::Importer::ReadFile does for R/src/Core/Mesh.cpp:10-12): .gltf and .glb,
// buffers and images in files, "data:" URIs or the container's binary chunk; triangle lists, strips and fans with
// float POSITION / NORMAL, float or normalised u8 / u16 TEXCOORD_0, and u8 / u16 / u32 (or no) indices.  Meshes are
// visited the way Mesh::processNode walks the importer's node tree -- a node's meshes, then its children, from the scene's root nodes -- and, like the
// reference, node transforms are NOT applied (processNode never reads aiNode::mTransformation; only
// the caller's model matrix is).  The V texture coordinate is flipped (1 - v), which is what the
// importer used by the reference hands to Mesh.cpp:125-133 for glTF files.
namespace {

struct GltfDoc {
    JValue root;
    std::string dir;
    std::vector<std::string> blobs;
    const JValue *meshes = nullptr, *accessors = nullptr, *views = nullptr, *nodes = nullptr, *materials = nullptr,
                 *textures = nullptr, *images = nullptr;

    static int asIndex(const JValue* v) {  // a JSON number used as an array index: -1 unless in int range
        return v && v->t == JValue::NUM && v->num >= 0 && v->num < 2147483647.0 ? (int)v->num : -1;
    }
    static const JValue* at(const JValue* arr, int i) {
        return arr && arr->t == JValue::ARR && i >= 0 && i < (int)arr->arr.size() ? &arr->arr[(size_t)i] : nullptr;
    }

    // RFC 4648 base64 (the payload of a "data:" URI after its comma); false on a character outside the alphabet
    static bool base64(const std::string& in, size_t from, std::string* out) {
        uint32_t acc = 0;
        int bits = 0;
        out->clear();
        out->reserve((in.size() - from) * 3 / 4);
        for (size_t i = from; i < in.size(); i++) {
            const char c = in[i];
            int v;
            if (c >= 'A' && c <= 'Z') v = c - 'A';
            else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
            else if (c >= '0' && c <= '9') v = c - '0' + 52;
            else if (c == '+' || c == '-') v = 62;
            else if (c == '/' || c == '_') v = 63;
            else if (c == '=' || c == '\n' || c == '\r' || c == ' ') continue;
            else return false;
            acc = (acc << 6) | (uint32_t)v;
            bits += 6;
            if (bits >= 8) {
                bits -= 8;
                out->push_back((char)((acc >> bits) & 0xFF));
            }
        }
        return true;
    }
    static bool dataUri(const std::string& uri, std::string* out) {
        const size_t comma = uri.find(',');
        return comma != std::string::npos && uri.find(";base64") < comma && base64(uri, comma + 1, out);
    }

    std::string glbBin;  // the binary chunk of a .glb container: buffer 0 when that buffer names no uri

    bool open(const char* path, std::string* err) {
        std::string text;
        if (!readFile(path, &text)) {
            *err = std::string("cannot read ") + path;
            return false;
        }
        // Binary container (glTF 2.0 specification, section 4.4): 12-byte header {"glTF", version, length}, then
        // chunks {length, type, data}: the JSON first, optionally one binary chunk after it.
        if (text.size() >= 12 && text.compare(0, 4, "glTF") == 0) {
            auto u32 = [&](size_t at) {
                return (uint32_t)(uint8_t)text[at] | ((uint32_t)(uint8_t)text[at + 1] << 8) | ((uint32_t)(uint8_t)text[at + 2] << 16) |
                       ((uint32_t)(uint8_t)text[at + 3] << 24);
            };
            if (u32(4) != 2) {
                *err = "unsupported .glb version";
                return false;
            }
            std::string json;
            size_t at = 12;
            bool haveJson = false;
            while (at + 8 <= text.size()) {
                const size_t len = u32(at), type = u32(at + 4);
                if (len > text.size() - at - 8) {
                    *err = "truncated .glb chunk";
                    return false;
                }
                if (type == 0x4E4F534Au && !haveJson) {
                    json.assign(text, at + 8, len);
                    haveJson = true;
                } else if (type == 0x004E4942u && glbBin.empty()) {
                    glbBin.assign(text, at + 8, len);
                }
                at += 8 + ((len + 3) & ~(size_t)3);
            }
            if (!haveJson) {
                *err = ".glb file without a JSON chunk";
                return false;
            }
            text.swap(json);
        }
        JParser jp(text);
        root = jp.parse();
        if (!jp.ok || root.t != JValue::OBJ) {
            *err = "malformed glTF JSON";
            return false;
        }
        meshes = root.get("meshes");
        accessors = root.get("accessors");
        views = root.get("bufferViews");
        nodes = root.get("nodes");
        materials = root.get("materials");
        textures = root.get("textures");
        images = root.get("images");
        const JValue* buffers = root.get("buffers");
        if (!meshes || !accessors || !views || !buffers || meshes->t != JValue::ARR || accessors->t != JValue::ARR ||
            views->t != JValue::ARR || buffers->t != JValue::ARR || meshes->arr.empty()) {
            *err = "glTF lacks meshes/accessors/bufferViews/buffers";
            return false;
        }
        dir = path;
        const size_t slash = dir.find_last_of('/');
        dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
        blobs.resize(buffers->arr.size());
        for (size_t i = 0; i < buffers->arr.size(); i++) {
            const JValue* uri = buffers->arr[i].get("uri");
            bool got;
            if (!uri)
                got = i == 0 && !glbBin.empty() && (blobs[0] = glbBin, true);  // the .glb's own binary chunk
            else if (uri->t != JValue::STR)
                got = false;
            else if (uri->str.compare(0, 5, "data:") == 0)
                got = dataUri(uri->str, &blobs[i]);
            else
                got = readFile(dir + "/" + uri->str, &blobs[i]);
            if (!got) {
                *err = "cannot read glTF buffer " + std::to_string(i);
                return false;
            }
        }
        return true;
    }

    struct View {
        const char* data = nullptr;
        size_t count = 0;
        int comp = 0;
        size_t stride = 0;
    };
    bool view(int acc, int ncomp, size_t compSize, View* out) const {
        const JValue* a = at(accessors, acc);
        if (!a) return false;
        const JValue *bv = a->get("bufferView"), *cnt = a->get("count"), *ct = a->get("componentType");
        if (!bv || !cnt || !ct || bv->t != JValue::NUM || cnt->t != JValue::NUM || ct->t != JValue::NUM) return false;
        if (!(cnt->num >= 0) || cnt->num > 4e9) return false;
        const JValue* v = at(views, asIndex(bv));
        if (!v) return false;
        const JValue *buf = v->get("buffer"), *off = v->get("byteOffset"), *st = v->get("byteStride");
        const double bufIdx = buf ? buf->num : 0.0;
        if (!(bufIdx >= 0) || bufIdx >= (double)blobs.size()) return false;
        const double o1 = off ? off->num : 0.0, o2 = a->get("byteOffset") ? a->get("byteOffset")->num : 0.0;
        const double strideNum = st ? st->num : (double)(ncomp * compSize);
        if (!(o1 >= 0) || !(o2 >= 0) || o1 > 4e9 || o2 > 4e9 || !(strideNum >= (double)(ncomp * compSize)) || strideNum > 4e9)
            return false;
        const size_t offset = (size_t)o1 + (size_t)o2;
        const std::string& blob = blobs[(size_t)bufIdx];
        out->count = (size_t)cnt->num;
        out->comp = (int)ct->num;
        out->stride = (size_t)strideNum;
        if (offset + (out->count ? (out->count - 1) * out->stride + ncomp * compSize : 0) > blob.size()) return false;
        out->data = blob.data() + offset;
        return true;
    }

    // Meshes in the order Mesh::processNode reaches them (R/src/Core/Mesh.cpp:39-52).
    void meshOrder(std::vector<int>* out) const {
        std::vector<int> roots;
        const JValue* scenes = root.get("scenes");
        const JValue* sc = at(scenes, std::max(0, asIndex(root.get("scene"))));
        if (sc && sc->get("nodes") && sc->get("nodes")->t == JValue::ARR)
            for (const JValue& r : sc->get("nodes")->arr) roots.push_back(asIndex(&r));
        std::vector<char> seen(nodes && nodes->t == JValue::ARR ? nodes->arr.size() : 0, 0);
        std::vector<int> stack(roots.rbegin(), roots.rend());
        while (!stack.empty()) {
            const int n = stack.back();
            stack.pop_back();
            const JValue* node = at(nodes, n);
            if (!node || seen[(size_t)n]) continue;  // a node reachable twice (or a cycle) is visited once
            seen[(size_t)n] = 1;
            const int m = asIndex(node->get("mesh"));
            if (at(meshes, m)) out->push_back(m);
            const JValue* kids = node->get("children");
            if (kids && kids->t == JValue::ARR)
                for (auto it = kids->arr.rbegin(); it != kids->arr.rend(); ++it) stack.push_back(asIndex(&*it));
        }
        if (out->empty())  // no scene graph: every mesh, in file order
            for (size_t m = 0; m < meshes->arr.size(); m++) out->push_back((int)m);
    }

    // File name of the image behind `material.<slot>` (slot inside pbrMetallicRoughness when pbr is
    // set), or "" when the material has no such texture.
    std::string textureFile(const JValue* material, const char* slot, bool pbr) const {
        if (!material) return "";
        const JValue* holder = pbr ? material->get("pbrMetallicRoughness") : material;
        const JValue* ref = holder ? holder->get(slot) : nullptr;
        const JValue* tex = ref ? at(textures, asIndex(ref->get("index"))) : nullptr;
        const int source = tex ? asIndex(tex->get("source")) : -1;
        const JValue* image = at(images, source);
        if (!image) return "";
        const JValue* uri = image->get("uri");
        if (uri && uri->t == JValue::STR && uri->str.compare(0, 5, "data:") != 0) return uri->str;
        // an image held in the file itself (a "data:" URI, or a bufferView as .glb files have them) goes by its number
        if ((uri && uri->t == JValue::STR) || image->get("bufferView")) return kEmbedded + std::to_string(source);
        return "";
    }
    static constexpr const char* kEmbedded = "\x01image:";
    // The encoded bytes (JPEG / PNG) of an image held in the file itself.
    bool embeddedImage(const std::string& key, std::string* out) const {
        const JValue* image = at(images, std::atoi(key.c_str() + std::strlen(kEmbedded)));
        if (!image) return false;
        const JValue* uri = image->get("uri");
        if (uri && uri->t == JValue::STR) return dataUri(uri->str, out);
        const JValue* v = at(views, asIndex(image->get("bufferView")));
        if (!v) return false;
        const JValue *buf = v->get("buffer"), *off = v->get("byteOffset"), *len = v->get("byteLength");
        const double b = buf ? buf->num : 0.0, o = off ? off->num : 0.0, l = len ? len->num : -1.0;
        if (!(b >= 0) || b >= (double)blobs.size() || !(o >= 0) || !(l >= 0) || o > 4e9 || l > 4e9) return false;
        const std::string& blob = blobs[(size_t)b];
        if ((size_t)o + (size_t)l > blob.size()) return false;
        out->assign(blob, (size_t)o, (size_t)l);
        return true;
    }
};

}  // namespace

int Scene::textureFromFile(const std::string& path, std::string* err) {
    std::string bytes;
    if (!readFile(path, &bytes)) {
        *err = "cannot read " + path;
        return -1;
    }
    return textureFromBytes(bytes, path, err);
}

int Scene::textureFromBytes(const std::string& bytes, const std::string& label, std::string* err) {
    Image img;
    std::string why;
    if (!decodeImage(reinterpret_cast<const uint8_t*>(bytes.data()), bytes.size(), &img, &why)) {
        *err = label + ": " + why;
        return -1;
    }
    std::vector<uint8_t> rgb;
    toRgb(img, &rgb);
    return addTexture(img.w, img.h, rgb.data());
}

// Mesh::Mesh(model, location) (R/src/Core/Mesh.cpp:7-37) for a glTF file.
//   material >= 0: every triangle gets that material and no image is read (textures bound by the caller).
//   material < 0, per_primitive = false: the reference's rule (Mesh.cpp:56-104) -- ONE PBRMaterial for
//     the whole Mesh, created from the first primitive whose material has a base-colour texture; the
//     metallic-roughness and normal textures of the primitives met from then on are attached to it.
//   per_primitive = true (beyond the reference): every primitive gets a material of its own -- PBR with
//     its own three textures when it has a base-colour texture, else Lambertian with its baseColorFactor.
// first_primitive_only restricts the mesh to the first primitive of the first mesh.
int Scene::addMeshGltf(const float* model16, const char* path, int material, int max_depth, bool per_primitive,
                       bool first_primitive_only, std::string* err) {
    GltfDoc g;
    if (!g.open(path, err)) return -1;
    std::vector<int> order;
    if (first_primitive_only)
        order.push_back(0);
    else
        g.meshOrder(&order);

    std::vector<float> P, N, T;
    std::vector<uint32_t> I;
    std::vector<int> triMat;
    bool anyN = false, anyT = false, allN = true, allT = true;
    std::map<std::string, int> loaded;  // image file -> texture index (each file is decoded once)
    auto texture = [&](const std::string& file) -> int {
        if (file.empty()) return -1;
        auto it = loaded.find(file);
        if (it != loaded.end()) return it->second;
        int t;
        if (file.compare(0, std::strlen(GltfDoc::kEmbedded), GltfDoc::kEmbedded) == 0) {
            std::string bytes;
            if (!g.embeddedImage(file, &bytes)) {
                *err = "cannot read image " + file.substr(std::strlen(GltfDoc::kEmbedded)) + " of the glTF file";
                return -1;
            }
            t = textureFromBytes(bytes, "glTF image " + file.substr(std::strlen(GltfDoc::kEmbedded)), err);
        } else {
            t = textureFromFile(g.dir + "/" + file, err);
        }
        if (t >= 0) loaded[file] = t;
        return t;
    };
    int shared = material;  // the one material of the reference's rule, once it exists
    std::string sharedR, sharedN;
    for (int m : order) {
        const JValue* prims = g.meshes->arr[(size_t)m].get("primitives");
        if (!prims || prims->t != JValue::ARR || prims->arr.empty()) {
            *err = "mesh without primitives";
            return -1;
        }
        for (size_t pi = 0; pi < prims->arr.size(); pi++) {
            if (first_primitive_only && pi > 0) break;
            const JValue& prim = prims->arr[pi];
            const JValue* attrs = prim.get("attributes");
            if (!attrs || !attrs->get("POSITION")) {
                *err = "primitive needs POSITION";
                return -1;
            }
            // triangle lists, strips and fans; points and lines have no surface to hit
            const int mode = prim.get("mode") ? GltfDoc::asIndex(prim.get("mode")) : 4;
            if (mode < 4 || mode > 6) {
                *err = "only triangle primitives (modes 4, 5, 6) are supported";
                return -1;
            }
            GltfDoc::View vp, vn, vt, vi;
            if (!g.view(GltfDoc::asIndex(attrs->get("POSITION")), 3, 4, &vp) || vp.comp != 5126) {
                *err = "bad POSITION accessor";
                return -1;
            }
            // optional attributes: absent is fine, present but unreadable is an error
            const bool hasN = attrs->get("NORMAL") != nullptr, hasT = attrs->get("TEXCOORD_0") != nullptr;
            // texture coordinates: float, or unsigned bytes / shorts that stand for value / 255 and value / 65535
            // (glTF 2.0 specification, section 3.7.2.1)
            int uvComp = 5126;
            if (hasT) {
                const JValue* acc = GltfDoc::at(g.accessors, GltfDoc::asIndex(attrs->get("TEXCOORD_0")));
                const JValue* ct = acc ? acc->get("componentType") : nullptr;
                uvComp = ct && ct->t == JValue::NUM ? (int)ct->num : 0;
            }
            const size_t uvSize = uvComp == 5126 ? 4 : (uvComp == 5123 ? 2 : (uvComp == 5121 ? 1 : 0));
            if ((hasN && !(g.view(GltfDoc::asIndex(attrs->get("NORMAL")), 3, 4, &vn) && vn.comp == 5126 && vn.count == vp.count)) ||
                (hasT && !(uvSize && g.view(GltfDoc::asIndex(attrs->get("TEXCOORD_0")), 2, uvSize, &vt) && vt.count == vp.count))) {
                *err = "bad NORMAL / TEXCOORD_0 accessor";
                return -1;
            }
            const size_t base = P.size() / 3, nv = vp.count;
            if (base + nv > 0x7FFFFFFFu) {
                *err = "too many vertices";
                return -1;
            }
            P.resize((base + nv) * 3);
            N.resize((base + nv) * 3, 0.0f);
            T.resize((base + nv) * 2, 0.0f);
            for (size_t i = 0; i < nv; i++) std::memcpy(&P[3 * (base + i)], vp.data + i * vp.stride, 12);
            if (hasN)
                for (size_t i = 0; i < nv; i++) std::memcpy(&N[3 * (base + i)], vn.data + i * vn.stride, 12);
            if (hasT)
                for (size_t i = 0; i < nv; i++) {
                    float uvv[2];
                    const char* src = vt.data + i * vt.stride;
                    if (uvComp == 5126) {
                        std::memcpy(uvv, src, 8);
                    } else if (uvComp == 5123) {
                        uint16_t q[2];
                        std::memcpy(q, src, 4);
                        uvv[0] = (float)q[0] / 65535.0f;
                        uvv[1] = (float)q[1] / 65535.0f;
                    } else {
                        uvv[0] = (float)(uint8_t)src[0] / 255.0f;
                        uvv[1] = (float)(uint8_t)src[1] / 255.0f;
                    }
                    T[2 * (base + i)] = uvv[0];
                    T[2 * (base + i) + 1] = 1.0f - uvv[1];
                }
            anyN = anyN || hasN;
            anyT = anyT || hasT;
            allN = allN && hasN;
            allT = allT && hasT;
            const size_t firstIndex = I.size();
            std::vector<uint32_t> run;  // strips and fans: the primitive's vertex sequence, made into triangles below
            std::vector<uint32_t>& I0 = mode == 4 ? I : run;
            if (prim.get("indices")) {
                const int idxAcc = GltfDoc::asIndex(prim.get("indices"));
                const JValue* acc = GltfDoc::at(g.accessors, idxAcc);
                const JValue* ct = acc ? acc->get("componentType") : nullptr;
                const int idxComp = ct && ct->t == JValue::NUM ? (int)ct->num : 0;
                const size_t idxSize = idxComp == 5121 ? 1 : (idxComp == 5123 ? 2 : (idxComp == 5125 ? 4 : 0));
                if (!idxSize || !g.view(idxAcc, 1, idxSize, &vi)) {
                    *err = "bad index accessor";
                    return -1;
                }
                for (size_t i = 0; i < vi.count; i++) {
                    uint32_t v = 0;
                    std::memcpy(&v, vi.data + i * vi.stride, idxSize);  // little-endian host
                    if (v >= nv) {
                        *err = "index out of range in glTF mesh";
                        return -1;
                    }
                    I0.push_back((uint32_t)base + v);
                }
            } else {
                for (size_t i = 0; i < nv; i++) I0.push_back((uint32_t)(base + i));
            }
            // Strip: triangle i is (i, i+1, i+2), with the first two swapped for odd i so that all wind the same way;
            // fan: (0, i+1, i+2) -- the orders the importer the reference uses gives its faces (an assumption, like
            // everything about that importer: it is not part of the reference tree).
            for (size_t i = 0; mode != 4 && i + 2 < run.size(); i++) {
                if (mode == 5) {
                    I.push_back(run[i + (i & 1)]);
                    I.push_back(run[i + 1 - (i & 1)]);
                    I.push_back(run[i + 2]);
                } else {
                    I.push_back(run[0]);
                    I.push_back(run[i + 1]);
                    I.push_back(run[i + 2]);
                }
            }
            const size_t nTri = (I.size() - firstIndex) / 3;
            I.resize(firstIndex + nTri * 3);

            // ---- material of this primitive
            const JValue* mat = GltfDoc::at(g.materials, GltfDoc::asIndex(prim.get("material")));
            int use = material;
            if (material < 0) {
                const std::string fd = g.textureFile(mat, "baseColorTexture", true);
                const std::string fr = g.textureFile(mat, "metallicRoughnessTexture", true);
                const std::string fn = g.textureFile(mat, "normalTexture", false);
                if (per_primitive) {
                    if (!fd.empty()) {
                        const int td = texture(fd), tr = texture(fr), tn = texture(fn);
                        if (td < 0 || (!fr.empty() && tr < 0) || (!fn.empty() && tn < 0)) return -1;
                        use = addPbr(td, tr, tn);
                    } else {
                        float rgb[3] = {1.0f, 1.0f, 1.0f};  // glTF's default baseColorFactor
                        const JValue* pbr = mat ? mat->get("pbrMetallicRoughness") : nullptr;
                        const JValue* f = pbr ? pbr->get("baseColorFactor") : nullptr;
                        if (f && f->t == JValue::ARR && f->arr.size() >= 3)
                            for (int k = 0; k < 3; k++) rgb[k] = (float)f->arr[(size_t)k].num;
                        use = addMaterial(0, rgb[0], rgb[1], rgb[2], 0.0f);
                    }
                } else {
                    // Mesh.cpp:58-104: a texture file that was already loaded is not loaded (or attached) again
                    if (!fd.empty() && shared < 0) {
                        const int td = texture(fd);
                        if (td < 0) return -1;
                        shared = addPbr(td, -1, -1);
                    }
                    if (shared >= 0 && !fr.empty() && !loaded.count(fr)) {
                        const int tr = texture(fr);
                        if (tr < 0) return -1;
                        materials[(size_t)shared].tex_r = tr;
                    }
                    if (shared >= 0 && !fn.empty() && !loaded.count(fn)) {
                        const int tn = texture(fn);
                        if (tn < 0) return -1;
                        materials[(size_t)shared].tex_n = tn;
                    }
                    use = -2;  // resolved after the loop: every triangle shares the one material
                }
            }
            triMat.insert(triMat.end(), nTri, use);
        }
    }
    if (material < 0 && !per_primitive) {
        if (shared < 0) {
            // the reference would leave matPtr null here and crash on the first hit
            *err = "no material of this glTF file has a base-colour texture; pass a material";
            return -1;
        }
        for (int& t : triMat) t = shared;
    }
    if (I.empty()) {
        *err = "glTF mesh without triangles";
        return -1;
    }
    // Mesh.cpp:122-142: default texture coordinates / zero normals apply per aiMesh; with several
    // primitives a missing attribute is therefore only representable when all of them lack it
    if ((anyN && !allN) || (anyT && !allT)) {
        *err = "primitives of one file must agree on having NORMAL / TEXCOORD_0";
        return -1;
    }
    int r = addMesh(model16, (int)(P.size() / 3), P.data(), anyN ? N.data() : nullptr, anyT ? T.data() : nullptr,
                    (int)(I.size() / 3), I.data(), triMat[0], max_depth, triMat.data());
    if (r < 0) *err = "index out of range in glTF mesh";
    return r;
}

// randomScene(), R/src/RaytracingApplication.cpp:359-405.  The distance test at :370 calls
// vec3::length(), glm's component count (3), so it never rejects: 22*22 + 4 = 488 spheres.
// Draw order inside one constructor argument list is the g++ order (right to left).
void Scene::randomScene() {
    addSphere(0.0f, -1000.0f, 0.0f, 1000.0f, addMaterial(0, 0.5f, 0.5f, 0.5f, 0));
    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++) {
            float chooseMat = randomFloat();
            float drawZ = randomFloat();  // third constructor argument is evaluated first
            float drawX = randomFloat();
            float cx = a + 0.9f * drawX, cy = 0.2f, cz = b + 0.9f * drawZ;
            int m;
            if (chooseMat < 0.8f) {
                float p[3], q[3];
                randomVec(p);
                randomVec(q);
                m = addMaterial(0, q[0] * p[0], q[1] * p[1], q[2] * p[2], 0);
            } else if (chooseMat < 0.95f) {
                float alb[3];
                randomVec(0.5f, 1.0f, alb);
                float fuzz = randomFloat(0.0f, 0.5f);
                m = addMaterial(1, alb[0], alb[1], alb[2], fuzz);
            } else {
                m = addMaterial(2, 0, 0, 0, 1.5f);
            }
            addSphere(cx, cy, cz, 0.2f, m);
        }
    }
    addSphere(0.0f, 1.0f, 0.0f, 1.0f, addMaterial(2, 0, 0, 0, 1.5f));
    addSphere(-4.0f, 1.0f, 0.0f, 1.0f, addMaterial(0, 0.4f, 0.2f, 0.1f, 0));
    addSphere(4.0f, 1.0f, 0.0f, 1.0f, addMaterial(1, 0.7f, 0.6f, 0.5f, 0.0f));
}

// cornellBox(), R/src/RaytracingApplication.cpp:407-478: twelve flat-shaded triangles.
void Scene::cornellBox() {
    int red = addMaterial(0, 0.65f, 0.05f, 0.05f, 0);
    int white = addMaterial(0, 0.73f, 0.73f, 0.73f, 0);
    int green = addMaterial(0, 0.12f, 0.45f, 0.15f, 0);
    int light = addMaterial(3, 15.0f, 15.0f, 15.0f, 0);
    struct T {
        float p[9];
        int m;
    };
    const T tris[12] = {
        {{555, 0, 0, 555, 555, 0, 555, 555, 555}, green},   {{555, 0, 0, 555, 555, 555, 555, 0, 555}, green},
        {{0, 0, 0, 0, 555, 0, 0, 555, 555}, red},           {{0, 0, 0, 0, 555, 555, 0, 0, 555}, red},
        {{213, 554, 227, 343, 554, 227, 343, 554, 332}, light}, {{213, 554, 227, 343, 554, 332, 213, 554, 332}, light},
        {{0, 0, 0, 555, 0, 0, 555, 0, 555}, white},         {{0, 0, 0, 555, 0, 555, 0, 0, 555}, white},
        {{0, 555, 0, 555, 555, 0, 555, 555, 555}, white},   {{0, 555, 0, 555, 555, 555, 0, 555, 555}, white},
        {{0, 0, 555, 555, 0, 555, 555, 555, 555}, white},   {{0, 0, 555, 555, 555, 555, 0, 555, 555}, white},
    };
    const float zero9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, zero6[6] = {0, 0, 0, 0, 0, 0};
    for (const T& t : tris) addTriangle(t.p, zero9, zero6, t.m);
}

// Camera::Camera, R/src/Core/Camera.h:7-25.
void Scene::setCamera(const float* lookfrom, const float* lookat, const float* vup, float vfov, float aspect,
                      float aperture, float focus) {
    float theta = vfov * 0.01745329251994329576923690768489f;  // glm::radians
    float h = std::tan(theta / 2.0f);
    float viewportHeight = 2.0f * h;
    float viewportWidth = aspect * viewportHeight;
    float d[3] = {lookfrom[0] - lookat[0], lookfrom[1] - lookat[1], lookfrom[2] - lookat[2]};
    float w[3], u[3], v[3], c[3];
    normalize3(d, w);
    cross3(vup, w, c);
    normalize3(c, u);
    cross3(w, u, v);
    float fw = focus * viewportWidth, fh = focus * viewportHeight;
    // the search structure orders its entries by distance from the eye: a moved eye makes it stale
    // (still correct, only slower), and the next flatten() rebuilds it
    if (lookfrom[0] != camera.origin[0] || lookfrom[1] != camera.origin[1] || lookfrom[2] != camera.origin[2])
        flat.fast_stale = true;
    for (int k = 0; k < 3; k++) {
        camera.origin[k] = lookfrom[k];
        camera.horizontal[k] = fw * u[k];
        camera.vertical[k] = fh * v[k];
        camera.u[k] = u[k];
        camera.v[k] = v[k];
        camera.w[k] = w[k];
        camera.llc[k] = ((camera.origin[k] - camera.horizontal[k] / 2.0f) - camera.vertical[k] / 2.0f) - focus * w[k];
    }
    camera.lens_radius = aperture / 2;
}

// ---------------------------------------------------------------------------- flatten
int Scene::newEntry(const Box3* box) {
    int idx = (int)(flat.nodes.size() / 8);
    uint32_t w[8];
    for (int k = 0; k < 3; k++) {
        w[k] = box ? bitsOf(box->mn[k]) : 0u;
        w[4 + k] = box ? bitsOf(box->mx[k]) : 0u;
    }
    w[3] = (uint32_t)(idx + 1) | (box ? 0u : kNodeNoBox);
    w[7] = 0xFFFFFFFFu;  // first_ref = -1
    flat.nodes.insert(flat.nodes.end(), w, w + 8);
    return idx;
}

// Emits a HittableList's objects in order.  A run of consecutive primitives becomes one
// leaf run; the first run directly under a box-tested entry rides on that entry.
void Scene::emitObjects(const std::vector<Obj>& objs, int parent_entry) {
    bool can_attach = parent_entry >= 0;
    size_t i = 0;
    auto isPrimitive = [](const Obj& o) { return o.kind == Obj::SPHERE || o.kind == Obj::TRIANGLE; };
    while (i < objs.size()) {
        if (objs[i].kind == Obj::INSTANCE) {
            // an entry of its own: the placement's box, the instance's number where a leaf run's start would be
            const int e = newEntry(&instances[objs[i].id].box);
            flat.nodes[(size_t)e * 8 + 3] |= kNodeInstance;
            flat.nodes[(size_t)e * 8 + 7] = (uint32_t)objs[i].id;
            i++;
        } else if (isPrimitive(objs[i])) {
            size_t j = i;
            uint32_t first = (uint32_t)flat.prim_refs.size();
            while (j < objs.size() && isPrimitive(objs[j])) {
                flat.prim_refs.push_back((uint32_t)objs[j].id | (objs[j].kind == Obj::TRIANGLE ? kRefTriangle : 0u));
                j++;
            }
            flat.prim_refs.back() |= kRefLast;
            flat.max_leaf = std::max(flat.max_leaf, (int)(j - i));
            int e = can_attach ? parent_entry : newEntry(nullptr);
            flat.nodes[(size_t)e * 8 + 7] = first;
            i = j;
        } else {
            emitComposite(composites[objs[i].id]);
            i++;
        }
        can_attach = false;
    }
}

void Scene::emitBvhNode(const Composite& c, int node) {
    const BvhNode& n = c.nodes[node];
    int e = newEntry(&n.box);
    if (n.left >= 0) {
        emitBvhNode(c, n.left);
        emitBvhNode(c, n.right);
    } else {
        emitObjects(n.leaf, e);
    }
    uint32_t end = (uint32_t)(flat.nodes.size() / 8);
    flat.nodes[(size_t)e * 8 + 3] = end;  // a miss jumps past the whole subtree
}

// emitBvhNode for a large tree that buildBvh made and whose leaves hold primitives only: its nodes lie in pre-order,
// so node k of the tree is entry k of its block, a subtree's end follows from the parents before it, and a leaf
// run starts where the leaves before it end.  One sequential pass over the node headers for those two numbers, then
// the entries and runs are written in pieces on the builders' threads.  False (and nothing emitted) when a leaf
// holds anything but primitives or the numbering is not the expected one.
bool Scene::emitBvhFlat(const Composite& c) {
    const size_t N = c.built_count, root = (size_t)c.root;
    std::vector<uint32_t> endOf(N), runAt(N);
    endOf[0] = (uint32_t)N;
    size_t refs = 0;
    for (size_t i = 0; i < N; i++) {
        const BvhNode& n = c.nodes[root + i];
        runAt[i] = (uint32_t)refs;
        if (n.left >= 0) {
            const size_t l = (size_t)n.left - root, r = (size_t)n.right - root;
            if (l != i + 1 || r <= l || r >= endOf[i]) return false;
            endOf[l] = (uint32_t)r;
            endOf[r] = endOf[i];
        } else {
            if (n.leaf.empty() || endOf[i] != i + 1) return false;
            refs += n.leaf.size();
        }
    }
    const size_t entryBase = flat.nodes.size() / 8, refBase = flat.prim_refs.size();
    if (entryBase + N >= kNodeSkipMask || refBase + refs >= kRefIdMask) return false;
    flat.nodes.resize((entryBase + N) * 8);
    flat.prim_refs.resize(refBase + refs);
    std::atomic<bool> plain(true);
    std::atomic<int> longest(0);
    parallelPieces(N, [&](size_t first, size_t last) {
        int mine = 0;
        for (size_t i = first; i < last; i++) {
            const BvhNode& n = c.nodes[root + i];
            uint32_t* w = &flat.nodes[(entryBase + i) * 8];
            for (int k = 0; k < 3; k++) {
                w[k] = bitsOf(n.box.mn[k]);
                w[4 + k] = bitsOf(n.box.mx[k]);
            }
            w[3] = (uint32_t)(entryBase + endOf[i]);  // a miss jumps past the whole subtree
            w[7] = 0xFFFFFFFFu;
            if (n.left >= 0) continue;
            uint32_t* run = &flat.prim_refs[refBase + runAt[i]];
            for (size_t j = 0; j < n.leaf.size(); j++) {
                const Obj& o = n.leaf[j];
                if (o.kind != Obj::SPHERE && o.kind != Obj::TRIANGLE) {
                    plain.store(false);
                    return;
                }
                run[j] = (uint32_t)o.id | (o.kind == Obj::TRIANGLE ? kRefTriangle : 0u);
            }
            run[n.leaf.size() - 1] |= kRefLast;
            w[7] = (uint32_t)(refBase + runAt[i]);
            mine = std::max(mine, (int)n.leaf.size());
        }
        int seen = longest.load();
        while (mine > seen && !longest.compare_exchange_weak(seen, mine)) {}
    });
    if (!plain.load()) {
        flat.nodes.resize(entryBase * 8);
        flat.prim_refs.resize(refBase);
        return false;
    }
    flat.max_leaf = std::max(flat.max_leaf, longest.load());
    if (std::getenv("RT_HOST_TIMING") && std::atoi(std::getenv("RT_HOST_TIMING")) != 0)
        std::fprintf(stderr, "[flatten] a built tree of %zu nodes emitted in pieces\n", N);
    return true;
}

void Scene::emitComposite(const Composite& c) {
    if (c.kind == Composite::BVH) {
        const bool large = c.built_count >= 65536 && (size_t)c.root == c.built_first;
        if (!(large && emitBvhFlat(c))) emitBvhNode(c, c.root);
    } else if (c.kind == Composite::MESH) {
        int e = newEntry(&c.box);
        emitObjects(c.objs, e);
        flat.nodes[(size_t)e * 8 + 3] = (uint32_t)(flat.nodes.size() / 8);
    } else {
        emitObjects(c.objs, -1);
    }
}

void Scene::flatten() {
    if (flat.valid) {
        if (flat.fast_stale) buildFastStructure();
        return;
    }
    // RT_HOST_TIMING=1: the two halves of this function on stderr
    const bool timing = std::getenv("RT_HOST_TIMING") && std::atoi(std::getenv("RT_HOST_TIMING")) != 0;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[flatten] %-42s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t0).count());
        t0 = now;
    };
    flat.nodes.clear();
    flat.prim_refs.clear();
    flat.max_leaf = 0;
    emitObjects(world, -1);
    // An empty world (HittableList::Hit over no objects: every ray misses) is one entry without a
    // box test and without a leaf run, so the walks have something to read.
    if (flat.nodes.empty()) newEntry(nullptr);
    flat.n_world_nodes = (int)(flat.nodes.size() / 8);
    flat.proto_range.clear();
    if (!instances.empty()) {
        // the prototypes' entries follow the world's; each is walked from its first entry to its last by the
        // rays that enter one of its placements
        for (const Prototype& p : prototypes) {
            flat.proto_range.push_back((int32_t)(flat.nodes.size() / 8));
            emitObjects(p.objs, -1);
            flat.proto_range.push_back((int32_t)(flat.nodes.size() / 8));
        }
    }
    lap("reference-order entries");
    buildFastStructure();
    lap(flat.defer_search_build ? "guard tables, inputs of the device build" : "guard tables, search structure");
    flat.valid = true;
}

}  // namespace rt
