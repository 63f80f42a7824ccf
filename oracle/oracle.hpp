// CPU oracle: TEST INFRASTRUCTURE ONLY.
//
// A scalar restatement of the reference renderer's CPU path-tracing path
// (kurono267/Raytracer), used as the checker for the HIP library and as the
// timed CPU baseline.  Nothing in the shipped product may include, link or
// call this file; only tests/, the smoke check and the cpu_baseline leg of
// bench.py do.
//
// PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures
// for this path, and it cannot be compiled here (its Mango / simd / glm
// dependencies are absent).  Fidelity rests on line-by-line correspondence
// with the cited reference lines and on known-answer tests of the
// reference's observable quirks.
//
// Floating-point model: IEEE-754 binary32, round-to-nearest, no FMA
// contraction (compile with -ffp-contract=off), correctly rounded / and
// sqrt, every expression evaluated in the operand order written here, which
// is the order glm's generic (non-SIMD) definitions give the cited reference
// expressions.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

namespace orc {

// ---------------------------------------------------------------- vec math
// glm generic definitions: dot3 = a.x*b.x + a.y*b.y + a.z*b.z (left to
// right), cross, normalize = v * (1/sqrt(dot(v,v))), min(x,y)=(y<x)?y:x,
// max(x,y)=(x<y)?y:x.
struct vec2 { float x, y; };
struct vec3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline vec3 V3(float a) { return {a, a, a}; }
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3 operator/(vec3 a, vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline vec3 operator/(float s, vec3 a) { return {s / a.x, s / a.y, s / a.z}; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 x, vec3 y) {
    return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
inline float gmin(float x, float y) { return (y < x) ? y : x; }
inline float gmax(float x, float y) { return (x < y) ? y : x; }
inline vec3 gmin(vec3 a, vec3 b) { return {gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)}; }
inline vec3 gmax(vec3 a, vec3 b) { return {gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)}; }
inline vec3 normalize(vec3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }

// ---------------------------------------------------------------- transcendentals
// The reference calls libm (std::sin/cos/acos/atan2, and log2 stands in for Mango's lod); libm's last bits differ
// from one C library to the next and from the GPU's device library, so a film can only be compared with a
// tolerance.  Mode 1 swaps the five functions for fixed single-precision polynomials (Cephes-style: Cody-Waite
// reduction, minimax polynomials, only + - * / sqrt and integer steps, each a single IEEE operation), which the
// HIP kernels restate operation for operation (exp, log and pow of the microfacet lobe included): in that mode a
// film is comparable bit for bit.  Mode 0 (libm) is
// the default and is what the CPU baseline times; tests bound the distance between the two modes.
inline int& mathMode() { static int mode = 0; return mode; }

namespace dm {
inline float fromBits(uint32_t b) { float f; std::memcpy(&f, &b, 4); return f; }
inline uint32_t toBits(float f) { uint32_t b; std::memcpy(&b, &f, 4); return b; }
// sine and cosine of |x| < 1e4 (this path's arguments lie in [-pi/4, 2 pi]); NaN otherwise
inline void sincos(float x, float& s, float& c) {
    const float ax = std::fabs(x);
    if (!(ax < 1e4f)) { s = c = fromBits(0x7FC00000u); return; }
    int j = (int)(ax * 1.27323954473516f);   // multiples of pi/4, rounded up to even
    j += (j & 1);
    const float y = (float)j;
    const float r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    const float z = r * r;
    const float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    const float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    const int q = (j >> 1) & 3;              // |x| = r + q * pi/2
    const float sa = (q == 0) ? ps : (q == 1) ? pc : (q == 2) ? -ps : -pc;
    c = (q == 0) ? pc : (q == 1) ? -ps : (q == 2) ? -pc : ps;
    s = (x < 0.0f) ? -sa : sa;
}
inline float sin(float x) { float s, c; sincos(x, s, c); return s; }
inline float cos(float x) { float s, c; sincos(x, s, c); return c; }
inline float asinPoly(float x) {   // |x| <= 0.5
    const float z = x * x;
    return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * x + x;
}
inline float acos(float x) {       // x in [-1, 1]
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * asinPoly(std::sqrt(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * asinPoly(std::sqrt(0.5f * (1.0f - x)));
    return 1.57079632679489661923f - asinPoly(x);
}
inline float atanPos(float x) {    // x >= 0 (inf included)
    float y0 = 0.0f;
    if (x > 2.414213562373095f) { y0 = 1.57079632679489661923f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y0 = 0.78539816339744830962f; x = (x - 1.0f) / (x + 1.0f); }
    const float z = x * x;
    return y0 + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
}
inline float atan2(float y, float x) {   // (-pi, pi]; atan2(0, 0) = 0 whatever the signs of the zeros
    const float ax = std::fabs(x), ay = std::fabs(y);
    if (ax == 0.0f && ay == 0.0f) return 0.0f;
    float a = atanPos(ay / ax);
    if (x < 0.0f) a = 3.14159265358979323846f - a;
    return (y < 0.0f) ? -a : a;
}
inline float log2(float x) {             // positive normal x; +inf gives 128
    const uint32_t b = toBits(x);
    int e = (int)(b >> 23) - 127;
    float m = fromBits((b & 0x007FFFFFu) | 0x3F800000u);   // [1, 2)
    if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f, z = f * f;
    float y = ((((((((7.0376836292e-2f * f - 1.1514610310e-1f) * f + 1.1676998740e-1f) * f - 1.2420140846e-1f) * f + 1.4249322787e-1f) * f -
                  1.6668057665e-1f) * f + 2.0000714765e-1f) * f - 2.4999993993e-1f) * f + 3.3333331174e-1f) * f * z;
    y = y - 0.5f * z;
    return (float)e + (f + y) * 1.44269504088896341f;
}
// natural logarithm of x >= 0 (zero gives -inf, denormals are scaled up first); NaN for negative x and NaN
inline float log(float x) {
    if (!(x >= 0.0f)) return fromBits(0x7FC00000u);
    if (x == 0.0f) return -std::numeric_limits<float>::infinity();
    if (!(x < std::numeric_limits<float>::infinity())) return x;
    int e = 0;
    if (x < 1.17549435e-38f) { x = x * 8388608.0f; e = -23; }
    const uint32_t b = toBits(x);
    e += (int)(b >> 23) - 127;
    float m = fromBits((b & 0x007FFFFFu) | 0x3F800000u);   // [1, 2)
    if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f, z = f * f, fe = (float)e;
    float y = ((((((((7.0376836292e-2f * f - 1.1514610310e-1f) * f + 1.1676998740e-1f) * f - 1.2420140846e-1f) * f + 1.4249322787e-1f) * f -
                  1.6668057665e-1f) * f + 2.0000714765e-1f) * f - 2.4999993993e-1f) * f + 3.3333331174e-1f) * f * z;
    y = y + -2.12194440e-4f * fe;
    y = y - 0.5f * z;
    return (f + y) + 0.693359375f * fe;
}
inline float pow2i(int k) { return fromBits((uint32_t)(k + 127) << 23); }   // 2^k, k in [-126, 127]
// e^x: +inf above 88.72, 0 below -103.9 (-inf included); NaN for NaN
inline float exp(float x) {
    if (x != x) return x;
    if (x > 88.72283905206835f) return std::numeric_limits<float>::infinity();
    if (x < -103.9f) return 0.0f;
    const float n = std::floor(1.44269504088896341f * x + 0.5f);
    x = x - n * 0.693359375f;
    x = x - n * -2.12194440e-4f;
    const float z = x * x;
    const float p = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x +
                     5.0000001201e-1f) * z + x + 1.0f;
    const int k = (int)n, k1 = k / 2;
    return p * pow2i(k1) * pow2i(k - k1);
}
// x^y for x >= 0 as exp(y log x) (the lobe raises a number of (0, 1] to a power near 1)
inline float pow(float x, float y) { return exp(y * log(x)); }
}  // namespace dm

inline float mlog(float x) { return mathMode() ? dm::log(x) : std::log(x); }
inline float mexp(float x) { return mathMode() ? dm::exp(x) : std::exp(x); }
inline float mpow(float x, float y) { return mathMode() ? dm::pow(x, y) : std::pow(x, y); }
inline float msin(float x) { return mathMode() ? dm::sin(x) : std::sin(x); }
inline float mcos(float x) { return mathMode() ? dm::cos(x) : std::cos(x); }
inline float macos(float x) { return mathMode() ? dm::acos(x) : std::acos(x); }
inline float matan2(float y, float x) { return mathMode() ? dm::atan2(y, x) : std::atan2(y, x); }
inline float mlog2(float x) { return mathMode() ? dm::log2(x) : std::log2(x); }
inline float length2(vec3 v) { return dot(v, v); }
inline bool is_zero(vec3 v) { return v.x == 0.0f && v.y == 0.0f && v.z == 0.0f; }

// src/scene/Common.hpp:9-11 (std::max(std::min(v,hi),lo))
inline float clampf(float v, float lo, float hi) { return std::max(std::min(v, hi), lo); }

// src/scene/Common.hpp:27-30
inline float powerHeuristic(int nf, float fPdf, int ng, float gPdf) {
    float f = nf * fPdf, g = ng * gPdf;
    return (f * f) / (f * f + g * g);
}

// src/scene/Common.hpp:13-25; the final clamp runs in float and converts back
template <typename Pred>
inline int findInterval(int size, const Pred& pred) {
    int first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (pred(middle)) {
            first = middle + 1;
            len -= half + 1;
        } else
            len = half;
    }
    return (int)clampf((float)(first - 1), (float)0, (float)(size - 2));
}

// ------------------------------------------------- deterministic sampler
// Replaces std::mt19937(std::random_device()) of src/cpu/PathTracer.cpp:20-22,
// which cannot be reproduced run to run.  Counter based: one draw is a pure
// function of (seed, pixel, pass, dimension).
inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline float rng_u01(uint64_t seed, uint32_t pixel, uint32_t pass, uint32_t dim) {
    // 16 bits each for the dimension and the low half of the pass number, 32 for the pixel; the high half of the
    // pass number re-keys the stream, so accumulation can run past 65535 passes without repeating a sequence
    const uint32_t pass_hi = pass >> 16;
    uint64_t seed_mixed = splitmix64(seed);
    if (pass_hi) seed_mixed = splitmix64(seed_mixed ^ ((uint64_t)pass_hi << 32));
    uint64_t ctr = ((uint64_t)(pass & 0xFFFFu) << 48) ^ ((uint64_t)pixel << 16) ^ (uint64_t)dim;
    uint64_t z = splitmix64(seed_mixed ^ ctr);
    return (float)(z >> 40) * 0x1p-24f;
}

// ------------------------------------------------------------ Ray / hit
// src/scene/Ray.hpp:10-24
struct Ray {
    vec3 org, dir, invdir;
    Ray() = default;
    Ray(vec3 o, vec3 d) : org(o), dir(d) {
        invdir = 1.0f / dir;
        if (dir.x == 0.0f) invdir.x = 0.0f;
        if (dir.y == 0.0f) invdir.y = 0.0f;
        if (dir.z == 0.0f) invdir.z = 0.0f;
    }
};
// src/scene/Ray.hpp:26-35, Ray.cpp:9-15
struct RayHit {
    bool status = false;
    float dist = std::numeric_limits<float>::infinity();
    int id0 = -1, id1 = -1;
    vec3 bc = {0.f, 0.f, 0.f};
};
struct BBox {
    vec3 min = V3(std::numeric_limits<float>::infinity());
    vec3 max = V3(-std::numeric_limits<float>::infinity());
    // src/scene/BBox.cpp:10-17
    void expand(vec3 mn, vec3 mx) { min = gmin(mn, min); max = gmax(mx, max); }
};

// src/scene/Ray.cpp:17-34
inline RayHit intersectBBox(const Ray& ray, const BBox& box) {
    RayHit hit;
    vec3 lov = ray.invdir * (box.min - ray.org);
    vec3 hiv = ray.invdir * (box.max - ray.org);
    vec3 max_v = gmax(lov, hiv);
    vec3 min_v = gmin(lov, hiv);
    float tmin = gmax(min_v.x, gmax(min_v.y, min_v.z));
    float tmax = gmin(max_v.x, gmin(max_v.y, max_v.z));
    if (tmin <= tmax && (tmax > 0.0f)) {
        hit.dist = tmin;
        hit.status = true;
    }
    return hit;
}

// src/scene/Ray.cpp:36-67
inline RayHit intersectTriangle(const Ray& ray, vec3 v0, vec3 v1, vec3 v2) {
    RayHit hit;
    vec3 e1 = v1 - v0;
    vec3 e2 = v2 - v0;
    vec3 pvec = cross(ray.dir, e2);
    float det = dot(e1, pvec);
    if ((double)std::fabs(det) < 1e-8) return hit;  // double-typed literal
    float inv_det = 1 / det;
    vec3 tvec = ray.org - v0;
    float u = dot(tvec, pvec) * inv_det;
    if (u < 0.0f || u > 1.0f) return hit;
    vec3 qvec = cross(tvec, e1);
    float v = dot(ray.dir, qvec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return hit;
    hit.dist = dot(e2, qvec) * inv_det;
    hit.status = true;
    hit.bc = {1.0f - u - v, u, v};
    return hit;
}

// ------------------------------------------------------------------ BVH
// src/scene/BBox.hpp:27-56 (the fields the scalar path reads)
struct BVHNode {
    BBox box;
    int left = -1, right = -1;
    int tri = -1;      // triIds[0]: index-buffer offset of the triangle
    int model = -1;    // modelIds[0]
    int depth = 0;
    bool isLeaf = false;
};

struct Vertex {
    vec3 pos{0, 0, 0}, normal{0, 0, 0};
    vec2 uv{0, 0};
    vec3 tangent{0, 0, 0}, binormal{0, 0, 0};
};

// src/scene/BVH.cpp:191-198
inline uint32_t expandBits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
// src/scene/BVH.cpp:202-212; glm::clamp = min(max(x,lo),hi)
inline uint32_t morton3D(vec3 v) {
    auto c = [](float a) { return gmin(gmax(a * 1024.f, 0.f), 1023.f); };
    float x = c(v.x), y = c(v.y), z = c(v.z);
    // a NaN coordinate (flat scene axis: 0/0) is an undefined conversion in
    // the reference; both sides here define it as cell 0
    uint32_t xx = expandBits(x == x ? (uint32_t)x : 0u);
    uint32_t yy = expandBits(y == y ? (uint32_t)y : 0u);
    uint32_t zz = expandBits(z == z ? (uint32_t)z : 0u);
    return xx * 4 + yy * 2 + zz;
}

struct Counters {
    uint64_t n_inner = 0;  // inner nodes visited (2 child box tests each)
    uint64_t n_tri = 0;    // triangle tests
    uint64_t n_rays = 0;
    // inner nodes entered as the far child of a pair although their box distance no longer beats the hit
    // the near child's subtree produced; both of their child tests fail (a child box lies inside its
    // parent's, and the slab arithmetic is monotonic), so they are the visits a traversal may skip
    // without changing any result
    uint64_t n_inner_stale = 0;
    // triangle tests that produced exactly the distance of the hit already held: the strict `<` of
    // src/scene/BVH.cpp:139 keeps the first-visited triangle, so these are the ties whose resolution depends
    // on the traversal order
    uint64_t n_ties = 0;
};

struct BVH {
    struct Prim {
        vec3 minBox = V3(std::numeric_limits<float>::infinity());
        vec3 maxBox = V3(-std::numeric_limits<float>::infinity());
        vec3 center{0, 0, 0};
        int id = 0, modelId = 0;
        uint32_t mortonCode = 0;
    };
    std::vector<uint32_t> indices;
    std::vector<Vertex> vertices;
    std::vector<BVHNode> nodes;
    size_t rootId = 0;
    int maxDepth = 0;

    // src/scene/BVH.cpp:214-240 (std::__clz -> __builtin_clz)
    static uint32_t findSplit(const std::vector<Prim>& p, uint32_t start, uint32_t end) {
        uint32_t startMorton = p[start].mortonCode, endMorton = p[end].mortonCode;
        if (startMorton == endMorton) return (start + end) >> 1;
        int commonPrefix = __builtin_clz(startMorton ^ endMorton);
        uint32_t split = start, step = end - start;
        do {
            step = (step + 1) >> 1;
            uint32_t newSplit = split + step;
            if (newSplit < end) {
                uint32_t splitCode = p[newSplit].mortonCode;
                // identical codes: clz(0) is undefined; treat as the longest prefix
                uint32_t x = startMorton ^ splitCode;
                int splitPrefix = x ? __builtin_clz(x) : 32;
                if (splitPrefix > commonPrefix) split = newSplit;
            }
        } while (step > 1);
        return split;
    }

    // src/scene/BVH.cpp:242-285
    void recursive(BVHNode& root, const std::vector<Prim>& p, uint32_t start, uint32_t end, int depth) {
        if (maxDepth < depth) maxDepth = depth;
        BBox aabb;
        for (uint32_t i = start; i <= end; ++i) aabb.expand(p[i].minBox, p[i].maxBox);
        root.box = aabb;
        uint32_t size = end - start + 1;
        if (size <= 1) {
            root.isLeaf = true;
            root.tri = p[start].id;
            root.model = p[start].modelId;
            return;
        }
        uint32_t split = findSplit(p, start, end);
        BVHNode left, right;
        nodes.push_back(left);
        root.left = (int)nodes.size() - 1;
        nodes.push_back(right);
        root.right = (int)nodes.size() - 1;
        recursive(left, p, start, split, depth + 1);
        recursive(right, p, split + 1, end, depth + 1);
        nodes[root.left] = left;
        nodes[root.right] = right;
        root.depth = depth;
    }

    // src/scene/BVH.cpp:287-336.  `verts`/`idx` are the already flattened
    // arrays (indices offset by vbOffset), `triModel` the model of each
    // triangle; Prim.id is the offset of the triangle's first index.
    void runLBVH(const std::vector<Vertex>& verts, const std::vector<uint32_t>& idx,
                 const std::vector<int>& triModel) {
        vertices = verts;
        indices = idx;
        std::vector<Prim> prims;
        BBox mainBox;
        for (size_t i = 0; i + 2 < idx.size(); i += 3) {
            Prim prim;
            for (int v = 0; v < 3; ++v) {
                vec3 vPos = verts[idx[i + v]].pos;
                prim.minBox = gmin(prim.minBox, vPos);
                prim.maxBox = gmax(prim.maxBox, vPos);
                prim.center = prim.center + vPos;
            }
            prim.id = (int)i;
            prim.modelId = triModel[i / 3];
            mainBox.min = gmin(prim.minBox, mainBox.min);
            mainBox.max = gmax(prim.maxBox, mainBox.max);
            prim.center = prim.center / 3.0f;
            prims.push_back(prim);
        }
        for (auto& p : prims) {
            vec3 n = (p.center - mainBox.min) / (mainBox.max - mainBox.min);
            p.mortonCode = morton3D(n);
        }
        // reference: std::sort (unstable); ties broken by Prim.id here, which
        // is one of the orders std::sort may legally produce
        std::stable_sort(prims.begin(), prims.end(),
                         [](const Prim& a, const Prim& b) { return a.mortonCode < b.mortonCode; });
        nodes.clear();
        nodes.emplace_back();
        rootId = 0;
        maxDepth = 0;
        if (prims.empty()) return;
        BVHNode root;
        recursive(root, prims, 0, (uint32_t)prims.size() - 1, 0);
        nodes[rootId] = root;
    }

    // src/scene/BVH.cpp:129-170
    void intersectRec(const Ray& ray, RayHit& hit, const BVHNode& curr, Counters* c) const {
        if (curr.isLeaf) {
            int triId = curr.tri;
            if (triId >= 0) {
                if (c) c->n_tri++;
                uint32_t i0 = indices[triId], i1 = indices[triId + 1], i2 = indices[triId + 2];
                RayHit test = intersectTriangle(ray, vertices[i0].pos, vertices[i1].pos, vertices[i2].pos);
                if (c && test.status && test.dist > 0.0001f && test.dist == hit.dist) c->n_ties++;
                if (test.status && test.dist > 0.0001f && test.dist < hit.dist) {
                    hit = test;
                    hit.id1 = triId;
                    hit.id0 = curr.model;
                }
            }
        } else {
            if (c) c->n_inner++;
            const BVHNode& leftNode = nodes[curr.left];
            const BVHNode& rightNode = nodes[curr.right];
            RayHit left = intersectBBox(ray, leftNode.box);
            RayHit right = intersectBBox(ray, rightNode.box);
            bool leftStatus = left.status && left.dist < hit.dist;
            bool rightStatus = right.status && right.dist < hit.dist;
            if (leftStatus && rightStatus) {
                if (right.dist < left.dist) {
                    intersectRec(ray, hit, rightNode, c);
                    if (c && !leftNode.isLeaf && !(left.dist < hit.dist)) c->n_inner_stale++;
                    intersectRec(ray, hit, leftNode, c);
                } else {
                    intersectRec(ray, hit, leftNode, c);
                    if (c && !rightNode.isLeaf && !(right.dist < hit.dist)) c->n_inner_stale++;
                    intersectRec(ray, hit, rightNode, c);
                }
            } else if (leftStatus)
                intersectRec(ray, hit, leftNode, c);
            else if (rightStatus)
                intersectRec(ray, hit, rightNode, c);
        }
    }
    // src/scene/BVH.cpp:49-61
    RayHit intersect(const Ray& ray, Counters* c = nullptr) const {
        if (c) c->n_rays++;
        RayHit hit = intersectBBox(ray, nodes[rootId].box);
        if (!hit.status) return hit;
        hit.status = false;
        hit.dist = std::numeric_limits<float>::infinity();
        intersectRec(ray, hit, nodes[rootId], c);
        return hit;
    }
    // src/scene/BVH.cpp:35-45,63-127
    // tmax: far bound for shadow rays towards a point on an area light (not
    // in the reference, whose shadow rays are unbounded: infinity reproduces it).
    // It takes the place of the never-shrinking hit.dist of the reference loop.
    RayHit occluded(const Ray& ray, Counters* c = nullptr, float tmax = std::numeric_limits<float>::infinity()) const {
        if (c) c->n_rays++;
        RayHit hit = intersectBBox(ray, nodes[rootId].box);
        if (!hit.status) return hit;
        hit.status = false;
        hit.dist = tmax;
        const bool bounded = tmax < std::numeric_limits<float>::infinity();
        const int MAX_STACK = 128;
        int stack[MAX_STACK];
        int top = 0;
        stack[0] = (int)rootId;
        auto add = [&](int id) { if (top < MAX_STACK - 1) stack[++top] = id; };
        while (top >= 0) {
            int currId = stack[top--];
            const BVHNode& curr = nodes[currId];
            if (curr.isLeaf) {
                int triId = curr.tri;
                if (triId >= 0) {
                    if (c) c->n_tri++;
                    uint32_t i0 = indices[triId], i1 = indices[triId + 1], i2 = indices[triId + 2];
                    RayHit test = intersectTriangle(ray, vertices[i0].pos, vertices[i1].pos, vertices[i2].pos);
                    if (test.status && test.dist > 0.0001f && (!bounded || test.dist < tmax)) {
                        hit = test;
                        hit.id1 = triId;
                        hit.id0 = curr.model;
                        return hit;
                    }
                }
            } else {
                if (c) c->n_inner++;
                int leftID = curr.left, rightID = curr.right;
                RayHit left = intersectBBox(ray, nodes[leftID].box);
                RayHit right = intersectBBox(ray, nodes[rightID].box);
                bool leftStatus = left.status && left.dist < hit.dist;
                bool rightStatus = right.status && right.dist < hit.dist;
                if (leftStatus && rightStatus) {
                    if (right.dist < left.dist) std::swap(leftID, rightID);
                    add(rightID);
                    add(leftID);
                } else if (leftStatus)
                    add(leftID);
                else if (rightStatus)
                    add(rightID);
            }
        }
        return hit;
    }
    // src/scene/BVH.cpp:172-187.  The accumulation starts from a zeroed vertex.
    Vertex postIntersect(const Ray& ray, const RayHit& hit, bool haveFrame) const {
        Vertex h;
        h.pos = ray.org + ray.dir * hit.dist;
        for (int v = 0; v < 3; ++v) {
            const Vertex& vert = vertices[indices[hit.id1 + v]];
            float b = hit.bc[v];
            h.normal = h.normal + vert.normal * b;
            h.uv = {h.uv.x + vert.uv.x * b, h.uv.y + vert.uv.y * b};
            h.tangent = h.tangent + vert.tangent * b;
            h.binormal = h.binormal + vert.binormal * b;
        }
        h.normal = normalize(h.normal);
        if (haveFrame) {
            h.binormal = normalize(h.binormal);
            h.tangent = normalize(h.tangent);
        } else {
            // Divergence from the reference (it would normalise a zero vector
            // into NaN for meshes without tangents): a deterministic
            // orthonormal frame built from the normal.
            vec3 n = h.normal, t;
            if (std::fabs(n.x) > std::fabs(n.y)) {
                float il = 1.0f / std::sqrt(n.x * n.x + n.z * n.z);
                t = {-n.z * il, 0.0f, n.x * il};
            } else {
                float il = 1.0f / std::sqrt(n.y * n.y + n.z * n.z);
                t = {0.0f, n.z * il, -n.y * il};
            }
            h.tangent = t;
            h.binormal = cross(n, t);
        }
        return h;
    }
};

// ----------------------------------------------------------------- BSDF
constexpr float INV_PI = 1.0f / (float)M_PI;             // src/bsdf/Samples.hpp:10-13
constexpr float INV_2PI = 1.0f / (2.f * (float)M_PI);
constexpr float PI_OVER_2 = (float)M_PI / 2.f;
constexpr float PI_OVER_4 = PI_OVER_2 / 2.f;

// src/bsdf/Samples.hpp:15-28
inline vec2 concentricSampleDisk(vec2 u) {
    vec2 uo = {2.f * u.x - 1.f, 2.f * u.y - 1.f};
    if (uo.x == 0 && uo.y == 0) return {0.f, 0.f};
    float theta, r;
    if (std::fabs(uo.x) > std::fabs(uo.y)) {
        r = uo.x;
        theta = PI_OVER_4 * (uo.y / uo.x);
    } else {
        r = uo.y;
        theta = PI_OVER_2 - PI_OVER_4 * (uo.x / uo.y);
    }
    return {r * mcos(theta), r * msin(theta)};
}
// src/bsdf/Samples.hpp:30-34 — the cosine-weighted "up" goes into y
inline vec3 cosineSampleHemisphere(vec2 u) {
    vec2 d = concentricSampleDisk(u);
    float z = std::sqrt(std::max(0.f, 1.f - d.x * d.x - d.y * d.y));
    return {d.x, z, d.y};
}
// src/bsdf/Samples.hpp:36-43
inline float sphericalTheta(vec3 v) { return macos(clampf(v.y, -1.f, 1.f)); }
inline float sphericalPhi(vec3 v) {
    float p = matan2(-v.z, -v.x);
    return (p < 0.f) ? (p + 2.f * (float)M_PI) : p;
}

// src/bsdf/BxDF.hpp:38-64
inline float cosTheta(vec3 w) { return w.z; }
inline float absCosTheta(vec3 w) { return std::fabs(w.z); }
inline float sin2Theta(vec3 w) { return std::max(0.f, 1.f - w.z * w.z); }
inline float sinTheta(vec3 w) { return std::sqrt(sin2Theta(w)); }
inline float cosPhi(vec3 w) { float s = sinTheta(w); return (s == 0) ? 1 : clampf(w.x / s, -1, 1); }
inline float sinPhi(vec3 w) { float s = sinTheta(w); return (s == 0) ? 0 : clampf(w.y / s, -1, 1); }
inline bool sameHemisphere(vec3 w, vec3 wp) { return w.z * wp.z > 0; }

// src/bsdf/Diffuse.cpp:7-32 (Oren-Nayar; sigma used as given)
struct Diffuse {
    vec3 color;
    float A, B;
    Diffuse() = default;
    Diffuse(vec3 c, float sigma) : color(c) {
        float sigma2 = sigma * sigma;
        A = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
        B = 0.45f * sigma2 / (sigma2 + 0.09f);
    }
    vec3 f(vec3 out, vec3 in) const {
        float sinThetaI = sinTheta(in), sinThetaO = sinTheta(out);
        float maxCos = 0;
        if ((double)sinThetaI > 1e-4 && (double)sinThetaO > 1e-4) {
            float sinPhiI = sinPhi(in), cosPhiI = cosPhi(in);
            float sinPhiO = sinPhi(out), cosPhiO = cosPhi(out);
            float dCos = cosPhiI * cosPhiO + sinPhiI * sinPhiO;
            maxCos = std::max(0.f, dCos);
        }
        float sinAlpha, tanBeta;
        if (absCosTheta(in) > absCosTheta(out)) {
            sinAlpha = sinThetaO;
            tanBeta = sinThetaI / absCosTheta(in);
        } else {
            sinAlpha = sinThetaI;
            tanBeta = sinThetaO / absCosTheta(out);
        }
        return color * INV_PI * (A + B * maxCos * sinAlpha * tanBeta);
    }
    // src/bsdf/BxDF.cpp:15-17
    float pdf(vec3 out, vec3 in) const { return sameHemisphere(out, in) ? absCosTheta(in) * INV_PI : 0; }
    // src/bsdf/BxDF.cpp:8-13
    vec3 sampled(vec3 out, vec2 u, vec3& in, float& p) const {
        in = cosineSampleHemisphere(u);
        if (out.z < 0) in.z *= -1;
        p = pdf(out, in);
        return f(out, in);
    }
};

// ------------------------------------------------------- glossy lobe
// The reference carries a Torrance-Sparrow lobe (src/bsdf/Microfacet.cpp)
// that Material::computeBSDF only adds inside a comment
// (src/scene/Material.cpp:205-209).  It is restated here as an OPTION of a
// material (glossy = 0 reproduces the reference as shipped), with two
// corrections without which the lobe cannot be sampled at all: the reflected
// direction is -out + 2 (out.wh) wh (the reference calls glm::reflect(out, wh),
// which mirrors an incident vector and lands in the wrong hemisphere,
// Microfacet.cpp:216), and the slope sampler uses the inverse error function
// where the reference calls erfc (Microfacet.cpp:57,76).  GGX has no
// counterpart in the reference: it is the Trowbridge-Reitz distribution as
// published in pbrt-v3 (src/core/microfacet.cpp), from which the reference's
// Beckmann code descends.
inline float cos2Theta(vec3 w) { return w.z * w.z; }                       // src/bsdf/BxDF.hpp:39
inline float tanTheta(vec3 w) { return sinTheta(w) / cosTheta(w); }         // :45
inline float tan2Theta(vec3 w) { return sin2Theta(w) / cos2Theta(w); }      // :46
inline float cos2Phi(vec3 w) { return cosPhi(w) * cosPhi(w); }              // :57-59
inline float sin2Phi(vec3 w) { return sinPhi(w) * sinPhi(w); }              // :60-62

// pbrt-v3 src/core/pbrt.h ErfInv (Giles' single-precision polynomial)
inline float erfInv(float x) {
    float w, p;
    x = clampf(x, -.99999f, .99999f);
    w = -mlog((1 - x) * (1 + x));
    if (w < 5) {
        w = w - 2.5f;
        p = 2.81022636e-08f;
        p = 3.43273939e-07f + p * w;
        p = -3.5233877e-06f + p * w;
        p = -4.39150654e-06f + p * w;
        p = 0.00021858087f + p * w;
        p = -0.00125372503f + p * w;
        p = -0.00417768164f + p * w;
        p = 0.246640727f + p * w;
        p = 1.50140941f + p * w;
    } else {
        w = std::sqrt(w) - 3;
        p = -0.000200214257f;
        p = 0.000100950558f + p * w;
        p = 0.00134934322f + p * w;
        p = -0.00367342844f + p * w;
        p = 0.00573950773f + p * w;
        p = -0.0076224613f + p * w;
        p = 0.00943887047f + p * w;
        p = 1.00167406f + p * w;
        p = 2.83297682f + p * w;
    }
    return p * x;
}
// pbrt-v3 src/core/pbrt.h Erf (Abramowitz-Stegun 7.1.26); stands in for the
// libm erf of Microfacet.cpp:30 so that host and device evaluate the same polynomial
inline float erfPoly(float x) {
    const float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f, a4 = -1.453152027f, a5 = 1.061405429f;
    const float p = 0.3275911f;
    int sign = 1;
    if (x < 0) sign = -1;
    x = std::fabs(x);
    float t = 1 / (1 + p * x);
    float y = 1 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * mexp(-x * x);
    return sign * y;
}

enum GlossyKind { GLOSSY_NONE = 0, GLOSSY_BECKMANN = 1, GLOSSY_GGX = 2 };

// src/bsdf/Microfacet.cpp:7-80 (Beckmann) with erfInv restored
inline void beckmannSample11(float cosThetaI, float U1, float U2, float* slope_x, float* slope_y) {
    if ((double)cosThetaI > .9999) {
        float r = std::sqrt(-mlog(1.0f - U1));
        float sinPhi = msin(2 * (float)M_PI * U2);
        float cosPhi = mcos(2 * (float)M_PI * U2);
        *slope_x = r * cosPhi;
        *slope_y = r * sinPhi;
        return;
    }
    float sinThetaI = std::sqrt(std::max((float)0, (float)1 - cosThetaI * cosThetaI));
    float tanThetaI = sinThetaI / cosThetaI;
    float cotThetaI = 1 / tanThetaI;
    float a = -1, c = erfPoly(cotThetaI);
    float sample_x = std::max(U1, (float)1e-6f);
    float thetaI = macos(cosThetaI);
    float fit = 1 + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
    float b = c - (1 + c) * mpow(1 - sample_x, fit);
    const float SQRT_PI_INV = 1.f / std::sqrt((float)M_PI);
    float normalization = 1 / (1 + c + SQRT_PI_INV * tanThetaI * mexp(-cotThetaI * cotThetaI));
    int it = 0;
    while (++it < 10) {
        if (!(b >= a && b <= c)) b = 0.5f * (a + c);
        float invErf = erfInv(b);
        float value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * mexp(-invErf * invErf)) - sample_x;
        float derivative = normalization * (1 - invErf * tanThetaI);
        if (std::fabs(value) < 1e-5f) break;
        if (value > 0)
            c = b;
        else
            a = b;
        b -= value / derivative;
    }
    *slope_x = erfInv(b);
    *slope_y = erfInv(2.0f * std::max(U2, (float)1e-6f) - 1.0f);
}
// pbrt-v3 src/core/microfacet.cpp TrowbridgeReitzSample11
inline void ggxSample11(float cosTheta, float U1, float U2, float* slope_x, float* slope_y) {
    if ((double)cosTheta > .9999) {
        float r = std::sqrt(U1 / (1 - U1));
        float phi = 6.28318530718f * U2;
        *slope_x = r * mcos(phi);
        *slope_y = r * msin(phi);
        return;
    }
    float sinTheta = std::sqrt(std::max((float)0, (float)1 - cosTheta * cosTheta));
    float tanTheta = sinTheta / cosTheta;
    float a = 1 / tanTheta;
    float G1 = 2 / (1 + std::sqrt(1.f + 1.f / (a * a)));
    float A = 2 * U1 / G1 - 1;
    float tmp = 1.f / (A * A - 1.f);
    if (tmp > 1e10f) tmp = 1e10f;
    float B = tanTheta;
    float D = std::sqrt(std::max(float(B * B * tmp * tmp - (A * A - B * B) * tmp), float(0)));
    float slope_x_1 = B * tmp - D;
    float slope_x_2 = B * tmp + D;
    *slope_x = (A < 0 || slope_x_2 > 1.f / tanTheta) ? slope_x_1 : slope_x_2;
    float S;
    if (U2 > 0.5f) {
        S = 1.f;
        U2 = 2.f * (U2 - .5f);
    } else {
        S = -1.f;
        U2 = 2.f * (.5f - U2);
    }
    float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) /
              (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
    *slope_y = S * z * std::sqrt(1.f + *slope_x * *slope_x);
}

// MicrofacetDistribution with visible-area sampling (src/bsdf/Microfacet.hpp:9-50)
struct Microfacet {
    int kind = GLOSSY_BECKMANN;
    float alphax = 0.5f, alphay = 0.6f;
    // src/bsdf/Microfacet.cpp:108-115 / pbrt-v3 TrowbridgeReitzDistribution::D
    float D(vec3 wh) const {
        float t2 = tan2Theta(wh);
        if (std::isinf(t2)) return 0.f;
        float cos4Theta = cos2Theta(wh) * cos2Theta(wh);
        if (kind == GLOSSY_BECKMANN)
            return mexp(-t2 * (cos2Phi(wh) / (alphax * alphax) + sin2Phi(wh) / (alphay * alphay))) /
                   ((float)M_PI * alphax * alphay * cos4Theta);
        float e = (cos2Phi(wh) / (alphax * alphax) + sin2Phi(wh) / (alphay * alphay)) * t2;
        return 1 / ((float)M_PI * alphax * alphay * cos4Theta * (1 + e) * (1 + e));
    }
    // src/bsdf/Microfacet.cpp:117-126 / pbrt-v3 TrowbridgeReitzDistribution::Lambda
    float lambda(vec3 w) const {
        float absTanTheta = std::fabs(tanTheta(w));
        if (std::isinf(absTanTheta)) return 0.f;
        float alpha = std::sqrt(cos2Phi(w) * alphax * alphax + sin2Phi(w) * alphay * alphay);
        if (kind == GLOSSY_BECKMANN) {
            float a = 1 / (alpha * absTanTheta);
            if (a >= 1.6f) return 0;
            return (1 - 1.259f * a + 0.396f * a * a) / (3.535f * a + 2.181f * a * a);
        }
        float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
        return (-1 + std::sqrt(1.f + alpha2Tan2Theta)) / 2;
    }
    float G1(vec3 w) const { return 1 / (1 + lambda(w)); }                       // Microfacet.hpp:14-17
    float G(vec3 wo, vec3 wi) const { return 1 / (1 + lambda(wo) + lambda(wi)); }  // :18-20
    // src/bsdf/Microfacet.cpp:82-103,158-165 (stretch, sample slopes, rotate, unstretch)
    vec3 sampleWh(vec3 wo, vec2 u) const {
        bool flip = wo.z < 0;
        vec3 wi = flip ? -wo : wo;
        vec3 wiStretched = normalize(vec3{alphax * wi.x, alphay * wi.y, wi.z});
        float slope_x, slope_y;
        if (kind == GLOSSY_BECKMANN)
            beckmannSample11(cosTheta(wiStretched), u.x, u.y, &slope_x, &slope_y);
        else
            ggxSample11(cosTheta(wiStretched), u.x, u.y, &slope_x, &slope_y);
        float tmp = cosPhi(wiStretched) * slope_x - sinPhi(wiStretched) * slope_y;
        slope_y = sinPhi(wiStretched) * slope_x + cosPhi(wiStretched) * slope_y;
        slope_x = tmp;
        slope_x = alphax * slope_x;
        slope_y = alphay * slope_y;
        vec3 wh = normalize(vec3{-slope_x, -slope_y, 1.f});
        if (flip) wh = -wh;
        return wh;
    }
    // src/bsdf/Microfacet.cpp:168-174 with sampleVisibleArea = true (Microfacet.hpp:40)
    float pdf(vec3 wo, vec3 wh) const { return D(wh) * G1(wo) * std::fabs(dot(wo, wh)) / absCosTheta(wo); }
};

// src/bsdf/Microfacet.cpp:176-198
struct FresnelConductor {
    vec3 etaI{0.5f, 0.5f, 0.5f}, etaT{0.5f, 0.5f, 0.5f}, k{0, 0, 0};
    vec3 eval(float cosI) const {
        float cosThetaI = clampf(cosI, -1, 1);
        vec3 eta = etaT / etaI;
        vec3 etak = k / etaI;
        float cosThetaI2 = cosThetaI * cosThetaI;
        float sinThetaI2 = 1.f - cosThetaI2;
        vec3 eta2 = eta * eta;
        vec3 etak2 = etak * etak;
        auto sq = [](vec3 v) { return vec3{std::sqrt(v.x), std::sqrt(v.y), std::sqrt(v.z)}; };
        vec3 t0 = eta2 - etak2 - V3(sinThetaI2);
        vec3 a2plusb2 = sq(t0 * t0 + 4.f * eta2 * etak2);
        vec3 t1 = a2plusb2 + V3(cosThetaI2);
        vec3 a = sq(0.5f * (a2plusb2 + t0));
        vec3 t2 = 2.f * cosThetaI * a;
        vec3 Rs = (t1 - t2) / (t1 + t2);
        vec3 t3 = cosThetaI2 * a2plusb2 + V3(sinThetaI2 * sinThetaI2);
        vec3 t4 = t2 * sinThetaI2;
        vec3 Rp = Rs * (t3 - t4) / (t3 + t4);
        return 0.5f * (Rp + Rs);
    }
};

// src/bsdf/Microfacet.cpp:200-228
struct Reflection {
    vec3 R{1, 1, 1};
    Microfacet distribution;
    FresnelConductor fresnel;
    vec3 f(vec3 wo, vec3 wi) const {
        float cosThetaO = absCosTheta(wo), cosThetaI = absCosTheta(wi);
        vec3 wh = wi + wo;
        if (cosThetaI == 0 || cosThetaO == 0) return {0.f, 0.f, 0.f};
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) return {0.f, 0.f, 0.f};
        wh = normalize(wh);
        vec3 F = fresnel.eval(dot(wi, wh));
        return R * distribution.D(wh) * distribution.G(wo, wi) * F / (4 * cosThetaI * cosThetaO);
    }
    // pdf is left untouched on the two early returns, as in the reference
    vec3 sampled(vec3 out, vec2 u, vec3& in, float& pdf) const {
        if (out.z == 0) return {0.f, 0.f, 0.f};
        vec3 wh = distribution.sampleWh(out, u);
        in = -out + 2.f * dot(out, wh) * wh;   // corrected, see the note above
        if (!sameHemisphere(out, in)) return {0.f, 0.f, 0.f};
        pdf = distribution.pdf(out, wh) / (4.f * dot(out, wh));
        return f(out, in);
    }
    float pdf(vec3 wo, vec3 wi) const {
        if (!sameHemisphere(wo, wi)) return 0;
        vec3 wh = normalize(wo + wi);
        return distribution.pdf(wo, wh) / (4.f * dot(wo, wh));
    }
};

// src/bsdf/BSDF.cpp:7-104 for the lobe sets src/scene/Material.cpp:195-225
// can build: the Diffuse lobe, plus the Reflection lobe when the material asks
// for it.  Both are REFLECTION lobes matched by the ALL flags.
struct BSDF {
    vec3 t, b, n;
    Diffuse lobe;
    Reflection glossy;
    int numParts = 1;
    BSDF() = default;
    BSDF(const Vertex& v, const Diffuse& d) : t(v.tangent), b(v.binormal), n(v.normal), lobe(d) {}
    void add(const Reflection& r) { glossy = r; numParts = 2; }
    vec3 world2local(vec3 v) const { return {dot(t, v), dot(b, v), dot(n, v)}; }
    vec3 local2world(vec3 v) const {
        return {t.x * v.x + b.x * v.y + n.x * v.z, t.y * v.x + b.y * v.y + n.y * v.z,
                t.z * v.x + b.z * v.y + n.z * v.z};
    }
    // :36-48
    vec3 f(vec3 outW, vec3 inW) const {
        vec3 in = world2local(inW), out = world2local(outW);
        bool reflect = dot(inW, n) * dot(outW, n) > 0;
        vec3 r{0.f, 0.f, 0.f};
        if (reflect) r = r + lobe.f(out, in);
        if (reflect && numParts > 1) r = r + glossy.f(out, in);
        return r;
    }
    // :91-104
    float pdf(vec3 outW, vec3 inW) const {
        vec3 out = world2local(outW), in = world2local(inW);
        float p = 0.f;
        p += lobe.pdf(out, in);
        if (numParts > 1) p += glossy.pdf(out, in);
        p /= (float)numParts;
        return p;
    }
    // :50-89
    vec3 sample(vec3 outW, vec2 u, vec3& inW, float& p) const {
        const int matchingComps = numParts;
        p = 0.f;
        int comp = std::min((int)std::floor(u.x * matchingComps), matchingComps - 1);
        vec2 uR = {u.x * matchingComps - comp, u.y};
        vec3 wi, wo = world2local(outW);
        vec3 fr = comp == 0 ? lobe.sampled(wo, uR, wi, p) : glossy.sampled(wo, uR, wi, p);
        if (p == 0) return {0.f, 0.f, 0.f};
        inW = local2world(wi);
        if (matchingComps > 1) {
            p += comp == 0 ? glossy.pdf(wo, wi) : lobe.pdf(wo, wi);
            p /= matchingComps;
            bool reflect = dot(inW, n) * dot(outW, n) > 0;
            fr = {0.f, 0.f, 0.f};
            if (reflect) {
                fr = fr + lobe.f(wo, wi);
                fr = fr + glossy.f(wo, wi);
            }
        }
        return fr;
    }
};

// ---------------------------------------------------------- distributions
// src/samplers/Distributions.cpp:7-44
struct Distribution1D {
    std::vector<float> func, cdf;
    float funcInt = 0;
    Distribution1D() = default;
    explicit Distribution1D(const std::vector<float>& data) : func(data) {
        cdf.resize(data.size() + 1);
        cdf[0] = 0.f;
        size_t n = data.size();
        for (size_t i = 1; i < cdf.size(); ++i) cdf[i] = cdf[i - 1] + func[i - 1] / (float)n;
        funcInt = cdf[n];
        if (funcInt == 0) {
            for (size_t i = 1; i < cdf.size(); ++i) cdf[i] = (float)i / (float)n;
        } else {
            for (size_t i = 1; i < cdf.size(); ++i) cdf[i] /= funcInt;
        }
    }
    int count() const { return (int)func.size(); }
    float sampleContinuous(float u, float& pdf, int* off = nullptr) const {
        int offset = findInterval((int)cdf.size(), [&](int i) { return cdf[i] <= u; });
        if (off) *off = offset;
        float du = u - cdf[offset];
        if ((cdf[offset + 1] - cdf[offset]) > 0) du /= (cdf[offset + 1] - cdf[offset]);
        pdf = func[offset] / funcInt;
        return (offset + du) / count();
    }
};
// src/samplers/Distributions.cpp:46-76
struct Distribution2D {
    std::vector<Distribution1D> cond;
    Distribution1D marginal;
    Distribution2D() = default;
    Distribution2D(const std::vector<float>& data, int width, int height) {
        std::vector<float> mf;
        for (int v = 0; v < height; ++v) {
            std::vector<float> line(data.begin() + (size_t)v * width, data.begin() + (size_t)(v + 1) * width);
            cond.emplace_back(line);
            mf.push_back(cond.back().funcInt);
        }
        marginal = Distribution1D(mf);
    }
    vec2 sampleContinuous(vec2 u, float& pdf) const {
        float pdfs[2];
        int v;
        float d1 = marginal.sampleContinuous(u.y, pdfs[1], &v);
        float d0 = cond[v].sampleContinuous(u.x, pdfs[0]);
        pdf = pdfs[0] * pdfs[1];
        return {d0, d1};
    }
    float pdf(vec2 p) const {
        int iu = (int)clampf((float)int(p.x * cond[0].count()), 0.f, (float)(cond[0].count() - 1));
        int iv = (int)clampf((float)int(p.y * marginal.count()), 0.f, (float)(marginal.count() - 1));
        return cond[iv].func[iu] / marginal.funcInt;
    }
};

// ----------------------------------------------------------------- lights
// Bilinear RGBA32F fetch standing in for Mango's filterLinear(image, uv, 0)
// (absent from the reference tree: build-defined; repeat wrap, texel centres
// at +0.5).
inline vec3 filterLinear(const float* img, int W, int H, vec2 uv) {
    float x = uv.x * (float)W - 0.5f, y = uv.y * (float)H - 0.5f;
    float fx0 = std::floor(x), fy0 = std::floor(y);
    float fx = x - fx0, fy = y - fy0;
    int x0 = (int)fx0, y0 = (int)fy0;
    auto wrap = [](int i, int n) { i %= n; return i < 0 ? i + n : i; };
    int xa = wrap(x0, W), xb = wrap(x0 + 1, W), ya = wrap(y0, H), yb = wrap(y0 + 1, H);
    auto px = [&](int xx, int yy) {
        const float* p = img + ((size_t)yy * W + xx) * 4;
        return vec3{p[0], p[1], p[2]};
    };
    vec3 top = px(xa, ya) * (1.0f - fx) + px(xb, ya) * fx;
    vec3 bot = px(xa, yb) * (1.0f - fx) + px(xb, yb) * fx;
    return top * (1.0f - fy) + bot * fy;
}

enum LightType { LIGHT_POINT = 0, LIGHT_AREA = 2, LIGHT_ENV = 3 };  // src/lights/LightSource.hpp:12-17

struct Light {
    int type = LIGHT_POINT;
    vec3 pos{0, 0, 0};        // point: light2world * (0,0,0,1)
    vec3 I{0, 0, 0};          // point intensity / env scale
    std::vector<float> img;   // env: RGBA32F, row-major, W*H*4
    int W = 0, H = 0;
    Distribution2D dist;
    // Area: one emissive triangle.  The reference only names the enum value
    // (src/lights/LightSource.hpp:15); the definition is pbrt-v3's
    // DiffuseAreaLight over a Triangle behind the LightSource interface:
    // uniform area sampling, solid-angle pdf dist^2 / (|cos| area), one-sided
    // emission I on the side of cross(p1-p0, p2-p0), le() = 0 (the reference's
    // integrator has no emission term at hits).
    vec3 p0{0, 0, 0}, p1{0, 0, 0}, p2{0, 0, 0}, nrm{0, 0, 1};
    float area = 0;

    bool isDelta() const { return type == LIGHT_POINT; }
    void buildArea() {
        vec3 c = cross(p1 - p0, p2 - p0);
        area = 0.5f * std::sqrt(length2(c));
        nrm = normalize(c);
    }

    // src/lights/EnvLight.cpp:8-26.  glm::mat3 is filled column-major from
    // the literal, so (M*rgb).y = 0.3575761 r + 0.7151522 g + 0.1191920 b.
    // The table image is (H wide, W high) and is written at (y, x).
    void buildEnv() {
        std::vector<float> table((size_t)W * H);
        for (int y = 0; y < H; ++y) {
            float sinTheta = std::sin((float)M_PI * (y + .5f) / (float)H);
            for (int x = 0; x < W; ++x) {
                const float* c = &img[((size_t)y * W + x) * 4];
                float lum = 0.3575761f * c[0] + 0.7151522f * c[1] + 0.1191920f * c[2];
                table[(size_t)x * H + y] = sinTheta * lum;
            }
        }
        dist = Distribution2D(table, H, W);
    }
    // src/lights/PointLight.cpp:14-18, src/lights/EnvLight.cpp:28-43
    // tmax: where a shadow ray towards the sample must stop (infinity for
    // the reference's lights, whose shadow rays are unbounded)
    vec3 sampleLi(const Vertex& vtx, vec2 u, vec3& in, float& pdf, float* tmax = nullptr) const {
        if (tmax) *tmax = std::numeric_limits<float>::infinity();
        if (type == LIGHT_POINT) {
            in = normalize(pos - vtx.pos);
            pdf = 1.f;
            return I / length2(pos - vtx.pos);
        }
        if (type == LIGHT_AREA) {
            float su0 = std::sqrt(u.x);
            float b0 = 1.f - su0, b1 = u.y * su0;
            vec3 p = p0 * b0 + p1 * b1 + p2 * (1.f - b0 - b1);
            vec3 w = p - vtx.pos;
            float d2 = length2(w);
            pdf = 0.0f;
            in = {0.f, 0.f, 0.f};
            if (d2 == 0) return {0.f, 0.f, 0.f};
            float d = std::sqrt(d2);
            in = w / d;
            float cosL = dot(nrm, -in);
            pdf = d2 / (std::fabs(cosL) * area);
            if (!std::isfinite(pdf)) { pdf = 0.0f; return {0.f, 0.f, 0.f}; }
            if (tmax) *tmax = d * (1.0f - 1e-4f);
            return cosL > 0 ? I : vec3{0.f, 0.f, 0.f};
        }
        float mapPdf;
        vec2 uv = dist.sampleContinuous(u, mapPdf);
        pdf = 0.0f;
        if (mapPdf == 0) return {0.f, 0.f, 0.f};
        float theta = uv.x * (float)M_PI, phi = uv.y * 2.f * (float)M_PI;
        float cosT = mcos(theta), sinT = msin(theta);
        float sinP = msin(phi), cosP = mcos(phi);
        in = {sinT * cosP, sinT * sinP, cosT};
        pdf = mapPdf / (2 * (float)M_PI * (float)M_PI * sinT);
        if (!std::isfinite(pdf)) pdf = 0.0f;
        return filterLinear(img.data(), W, H, uv) * I;
    }
    // src/lights/PointLight.cpp:20-22, src/lights/EnvLight.cpp:45-52
    float pdfLi(const Vertex& vtx, vec3 inW) const {
        if (type == LIGHT_POINT) return 1.0f;
        if (type == LIGHT_AREA) {
            RayHit h = intersectTriangle(Ray(vtx.pos, inW), p0, p1, p2);
            if (!h.status || !(h.dist > 0.0001f)) return 0;
            float p = (h.dist * h.dist) / (std::fabs(dot(nrm, -inW)) * area);
            return std::isfinite(p) ? p : 0;
        }
        float theta = sphericalTheta(inW), phi = sphericalPhi(inW);
        float sinT = msin(theta);
        if (sinT == 0) return 0;
        return dist.pdf({phi * INV_2PI, theta * INV_PI}) / (2 * (float)M_PI * (float)M_PI * sinT);
    }
    // src/lights/LightSource.hpp:28-30, src/lights/EnvLight.cpp:60-65
    vec3 le(const Ray& ray) const {
        if (type != LIGHT_ENV) return {0.f, 0.f, 0.f};
        vec3 w = normalize(ray.dir);
        vec2 st = {sphericalPhi(w) * INV_2PI, sphericalTheta(w) * INV_PI};
        return filterLinear(img.data(), W, H, st);
    }
};

// RGBA8 texture with its mip chain (the reference's spImage4b; its loader
// and filters live in the absent Mango library, so the definitions below are
// build-defined): level k+1 is the 2x2 box average of level k in integers,
// (a+b+c+d+2)>>2 per channel, sizes max(1, w>>1).
struct Texture {
    std::vector<std::vector<uint8_t>> levels;
    std::vector<int> w, h;
    void build(const uint8_t* rgba, int W, int H) {
        levels.clear(); w.clear(); h.clear();
        levels.emplace_back(rgba, rgba + (size_t)W * H * 4);
        w.push_back(W); h.push_back(H);
        while (w.back() > 1 || h.back() > 1) {
            int pw = w.back(), ph = h.back(), nw = std::max(1, pw >> 1), nh = std::max(1, ph >> 1);
            const std::vector<uint8_t>& src = levels.back();
            std::vector<uint8_t> dst((size_t)nw * nh * 4);
            for (int y = 0; y < nh; ++y)
                for (int x = 0; x < nw; ++x) {
                    int x0 = std::min(2 * x, pw - 1), x1 = std::min(2 * x + 1, pw - 1);
                    int y0 = std::min(2 * y, ph - 1), y1 = std::min(2 * y + 1, ph - 1);
                    for (int c = 0; c < 4; ++c) {
                        int sum = src[((size_t)y0 * pw + x0) * 4 + c] + src[((size_t)y0 * pw + x1) * 4 + c] +
                                  src[((size_t)y1 * pw + x0) * 4 + c] + src[((size_t)y1 * pw + x1) * 4 + c];
                        dst[((size_t)y * nw + x) * 4 + c] = (uint8_t)((sum + 2) >> 2);
                    }
                }
            levels.push_back(std::move(dst));
            w.push_back(nw); h.push_back(nh);
        }
    }
    // bilinear fetch of one level, channels in 0..255 (repeat wrap, texel centres at +0.5)
    vec3 bilinear(int level, vec2 uv) const {
        const int W = w[level], H = h[level];
        const uint8_t* img = levels[level].data();
        float x = uv.x * (float)W - 0.5f, y = uv.y * (float)H - 0.5f;
        float fx0 = std::floor(x), fy0 = std::floor(y);
        float fx = x - fx0, fy = y - fy0;
        int x0 = (int)fx0, y0 = (int)fy0;
        auto wrap = [](int i, int n) { i %= n; return i < 0 ? i + n : i; };
        int xa = wrap(x0, W), xb = wrap(x0 + 1, W), ya = wrap(y0, H), yb = wrap(y0 + 1, H);
        auto px = [&](int xx, int yy) {
            const uint8_t* p = img + ((size_t)yy * W + xx) * 4;
            return vec3{(float)p[0], (float)p[1], (float)p[2]};
        };
        vec3 top = px(xa, ya) * (1.0f - fx) + px(xb, ya) * fx;
        vec3 bot = px(xa, yb) * (1.0f - fx) + px(xb, yb) * fx;
        return top * (1.0f - fy) + bot * fy;
    }
    // stands in for filterLinear(*tex, uv, 0) (src/scene/Material.cpp:217)
    vec3 filterLinear(vec2 uv) const { return bilinear(0, uv); }
    // stands in for filterTrilinear(*tex, uv, dUVx, dUVy) (src/scene/Material.cpp:198):
    // lod = log2 of the longer footprint axis in level-0 texels, two bilinear fetches
    vec3 filterTrilinear(vec2 uv, vec2 dUVx, vec2 dUVy) const {
        const float W = (float)w[0], H = (float)h[0];
        float ax = dUVx.x * W, ay = dUVx.y * H, bx = dUVy.x * W, by = dUVy.y * H;
        float lx = std::sqrt(ax * ax + ay * ay), ly = std::sqrt(bx * bx + by * by);
        float rho = (lx < ly) ? ly : lx;
        const int last = (int)levels.size() - 1;
        float lod = 0.0f;
        if (rho > 1.0f) lod = mlog2(rho);  // false for NaN
        if (!(lod < (float)last)) lod = (float)last;
        int l0 = (int)std::floor(lod);
        int l1 = std::min(l0 + 1, last);
        float f = lod - (float)l0;
        return bilinear(l0, uv) * (1.0f - f) + bilinear(l1, uv) * f;
    }
};

struct MaterialData {
    vec3 diffuseColor{1, 1, 1};
    float roughness = 0.5f;  // src/scene/Material.cpp:9
    int texture = -1;        // index into Scene::textures, -1 = none (_diffTexture null)
    int glossy = GLOSSY_NONE;  // GlossyKind: adds the Reflection lobe of Material.cpp:205-209
};

// The lobe of the commented block at src/scene/Material.cpp:205-209:
// FresnelConductor(0.5, 0.5, diffColor), alpha = (roughness, roughness + 0.1), R = 1
inline void addGlossy(BSDF& bsdf, const MaterialData& m, vec3 diffColor) {
    if (m.glossy == GLOSSY_NONE) return;
    Reflection r;
    r.R = {1.f, 1.f, 1.f};
    r.fresnel.etaI = {0.5f, 0.5f, 0.5f};
    r.fresnel.etaT = {0.5f, 0.5f, 0.5f};
    r.fresnel.k = diffColor;
    r.distribution.kind = m.glossy;
    r.distribution.alphax = m.roughness;
    r.distribution.alphay = m.roughness + 0.1f;
    bsdf.add(r);
}

struct Camera {
    vec3 pos, right, up, forward;  // src/cpu/Integrator.cpp:85-92
};

// ------------------------------------------------------------- integrator
struct Scene {
    BVH bvh;
    std::vector<MaterialData> materials;  // one per model
    std::vector<Texture> textures;
    std::vector<Light> lights;
    bool haveFrame = true;
};

struct RenderCounters {
    std::atomic<uint64_t> closest{0}, shadow{0};
    // node visits and triangle tests of those rays (the Counters of BVH::intersect / BVH::occluded summed), and, of
    // them, the share of the rays that are the same in every pass for a still camera: the camera rays
    // (src/cpu/PathTracer.cpp:43-53 has no jitter) and the shadow rays from the primary vertex towards point lights
    // (PointLight::sampleLi ignores its random numbers, src/lights/PointLight.cpp:14-18)
    std::atomic<uint64_t> closestInner{0}, closestTri{0}, shadowInner{0}, shadowTri{0};
    std::atomic<uint64_t> primRays{0}, primInner{0}, primTri{0}, primPointRays{0}, primPointInner{0}, primPointTri{0};
};

// Test/bench hook: records the bounce ray each pixel traces after its
// `depth`-th shaded vertex (depth 0 = the ray leaving the primary hit).
// rays: 6 planes (ox oy oz dx dy dz) of n floats indexed by pixel; flags[pixel]
// = 1 where a ray was recorded.
struct RayCapture {
    float* rays = nullptr;
    uint8_t* flags = nullptr;
    int depth = 0;
    size_t n = 0;
};

// Random-number slots of one path vertex at depth d with L lights: per
// light 2 (light sample) + 2 (the BSDF sample of the non-delta branch), then
// 2 for the bounce.  Fixed slots so that a skipped draw does not shift later
// ones.
inline uint32_t dim_light(int d, int L, int l, int k) { return (uint32_t)(d * (4 * L + 2) + 4 * l + k); }
inline uint32_t dim_bounce(int d, int L, int k) { return (uint32_t)(d * (4 * L + 2) + 4 * L + k); }

// src/cpu/PathTracer.cpp:43-52
inline Ray cameraRay(const Camera& cam, int x, int y, int W, int H) {
    float ni = ((float)x / (float)W) - 0.5f;
    float nj = ((float)y / (float)H) - 0.5f;
    nj = -nj;
    nj *= ((float)H / (float)W);
    vec3 ip = ni * cam.right + nj * cam.up + cam.forward;
    return Ray(cam.pos, normalize(ip));
}

inline vec2 fract2(vec2 v) { return {v.x - std::floor(v.x), v.y - std::floor(v.y)}; }  // glm::fract

// src/scene/Material.cpp:195-212: the primary hit filters the texture with
// the quad's UV differences
inline BSDF computeBSDF(const Scene& s, const Vertex& v, int model, vec2 dUVx, vec2 dUVy) {
    const MaterialData& m = s.materials[model];
    vec2 repeatUV = fract2(v.uv);
    vec3 c = m.texture >= 0 ? s.textures[m.texture].filterTrilinear(repeatUV, dUVx, dUVy) : V3(255.0f);
    c = c / 255.0f;
    c = c * m.diffuseColor;
    BSDF bsdf(v, Diffuse(c, m.roughness));
    addGlossy(bsdf, m, c);
    return bsdf;
}
// src/scene/Material.cpp:214-225: later bounces use the plain bilinear fetch
inline BSDF computeBSDF(const Scene& s, const Vertex& v, int model) {
    const MaterialData& m = s.materials[model];
    vec2 repeatUV = fract2(v.uv);
    vec3 c = m.texture >= 0 ? s.textures[m.texture].filterLinear(repeatUV) : V3(255.0f);
    c = c / 255.0f;
    c = c * m.diffuseColor;
    BSDF bsdf(v, Diffuse(c, m.roughness));
    addGlossy(bsdf, m, c);   // the reference's comment sits in the first overload only; a material keeps its lobes on every bounce here
    return bsdf;
}

// PASS 2 of src/cpu/PathTracer.cpp:70-160 for one pixel whose primary ray
// hit: `vertex`/`bsdf` of the first hit come from the caller.  `frames` =
// passes already accumulated; film is W*H RGBA32F, updated in place (the
// reference's ping-pong pair reads the previous value of the same pixel only).
inline void shadePixel(const Scene& s, int x, int y, int W, int maxDepth, uint64_t seed, uint32_t pass, float frames, float* film,
                       Ray ray, Vertex vertex, BSDF bsdf, RenderCounters* rc, const RayCapture* cap = nullptr) {
    const int L = (int)s.lights.size();
    const uint32_t pixel = (uint32_t)(y * W + x);
    float* out4 = film + (size_t)pixel * 4;
    uint64_t nClosest = 0, nShadow = 0, nPrimPoint = 0;
    Counters cClosest, cShadow, cPrimPoint;
    vec3 out = -ray.dir;
    vec3 pixelColor{0.f, 0.f, 0.f};
    vec3 threshold{1.f, 1.f, 1.f};
    for (int d = 0; d < maxDepth; ++d) {
        for (int li_ = 0; li_ < L; ++li_) {
            const Light& light = s.lights[li_];
            vec3 in{0.f, 0.f, 0.f};
            float lightPdf, tmax;
            vec2 u = {rng_u01(seed, pixel, pass, dim_light(d, L, li_, 0)),
                      rng_u01(seed, pixel, pass, dim_light(d, L, li_, 1))};
            vec3 li = light.sampleLi(vertex, u, in, lightPdf, &tmax);
            if (lightPdf == 0.0f) continue;
            nShadow++;
            const bool primPoint = d == 0 && light.isDelta();
            if (primPoint) nPrimPoint++;
            RayHit shadowHit = s.bvh.occluded(Ray(vertex.pos, in), rc ? (primPoint ? &cPrimPoint : &cShadow) : nullptr, tmax);
            if (shadowHit.status) continue;
            vec3 bsdfColor = bsdf.f(out, in);
            float bsdfPdf = bsdf.pdf(out, in);
            if (light.isDelta()) {
                pixelColor = pixelColor + threshold * bsdfColor * li / lightPdf;
            } else {
                pixelColor = pixelColor + threshold * bsdfColor * li * powerHeuristic(1, lightPdf, 1, bsdfPdf) / lightPdf;
                vec3 inBSDF;
                vec2 u2 = {rng_u01(seed, pixel, pass, dim_light(d, L, li_, 2)),
                           rng_u01(seed, pixel, pass, dim_light(d, L, li_, 3))};
                vec3 f = bsdf.sample(out, u2, inBSDF, bsdfPdf);
                if (bsdfPdf == 0.0f) continue;
                f = f * std::fabs(dot(in, vertex.normal));
                float weight = 1.f;
                lightPdf = light.pdfLi(vertex, in);
                if (lightPdf == 0.0f) continue;
                weight = powerHeuristic(1, bsdfPdf, 1, lightPdf);
                vec3 li2 = light.le(ray);
                if (!is_zero(li2)) pixelColor = pixelColor + f * li2 * weight / bsdfPdf;
            }
        }
        vec3 nextDir{0.f, 0.f, 0.f};
        float nextPDF = 0.0f;
        vec2 u3 = {rng_u01(seed, pixel, pass, dim_bounce(d, L, 0)), rng_u01(seed, pixel, pass, dim_bounce(d, L, 1))};
        vec3 bsdfColor = bsdf.sample(out, u3, nextDir, nextPDF);
        if (is_zero(bsdfColor) || nextPDF == 0.0f) break;
        threshold = threshold * (bsdfColor * std::fabs(dot(nextDir, vertex.normal)) / nextPDF);
        ray = Ray(vertex.pos, nextDir);
        if (cap && cap->depth == d) {
            const size_t n = cap->n;
            cap->rays[pixel] = ray.org.x; cap->rays[n + pixel] = ray.org.y; cap->rays[2 * n + pixel] = ray.org.z;
            cap->rays[3 * n + pixel] = ray.dir.x; cap->rays[4 * n + pixel] = ray.dir.y; cap->rays[5 * n + pixel] = ray.dir.z;
            cap->flags[pixel] = 1;
        }
        nClosest++;
        RayHit hit = s.bvh.intersect(ray, rc ? &cClosest : nullptr);
        if (hit.status) {
            vertex = s.bvh.postIntersect(ray, hit, s.haveFrame);
            bsdf = computeBSDF(s, vertex, hit.id0);
            out = -ray.dir;
        } else {
            break;
        }
    }
    if (frames != 0.0f) {
        vec3 prev{out4[0], out4[1], out4[2]};
        prev = prev * frames;
        prev = prev + pixelColor;
        prev = prev / (frames + 1.f);
        out4[0] = prev.x; out4[1] = prev.y; out4[2] = prev.z; out4[3] = 1.0f;
    } else {
        out4[0] = pixelColor.x; out4[1] = pixelColor.y; out4[2] = pixelColor.z; out4[3] = 1.0f;
    }
    if (rc) {
        rc->closest += nClosest; rc->shadow += nShadow;
        rc->closestInner += cClosest.n_inner; rc->closestTri += cClosest.n_tri;
        rc->shadowInner += cShadow.n_inner + cPrimPoint.n_inner; rc->shadowTri += cShadow.n_tri + cPrimPoint.n_tri;
        rc->primPointRays += nPrimPoint; rc->primPointInner += cPrimPoint.n_inner; rc->primPointTri += cPrimPoint.n_tri;
    }
}

// PathTracer::computeTile, src/cpu/PathTracer.cpp:17-164: 2x2 pixel quads,
// PASS 1 traces the primary rays of a quad, PASS 2 shades the pixels that hit
// with UV differences taken inside the quad.  vertexBlock lives for the whole
// tile and is only overwritten by pixels that hit, so a quad pixel that
// missed (or lies outside the film) contributes the UV of an earlier quad --
// zero at the start of a tile (sVertex's members are taken as zero-initialised).
inline void computeTile(const Scene& s, const Camera& cam, int startX, int startY, int W, int H, int tileSize, int maxDepth,
                        uint64_t seed, uint32_t pass, float frames, float* film, RenderCounters* rc, float* duvOut = nullptr,
                        const RayCapture* cap = nullptr) {
    bool statusBlock[4] = {false, false, false, false};
    Vertex vertexBlock[4];
    int modelBlock[4] = {0, 0, 0, 0};
    Ray raysBlock[4];
    const vec2 ptSize = {(float)W, (float)H};
    for (int yBlock = startY; yBlock < startY + tileSize; yBlock += 2) {
        if (yBlock >= H) break;
        for (int xBlock = startX; xBlock < startX + tileSize; xBlock += 2) {
            if (xBlock >= W) break;
            for (int yB = 0; yB < 2; ++yB) {
                int y = yBlock + yB;
                if (y >= H) break;
                for (int xB = 0; xB < 2; ++xB) {
                    int x = xBlock + xB;
                    if (x >= W) break;
                    Ray r = cameraRay(cam, x, y, W, H);
                    Counters cPrim;
                    RayHit hit = s.bvh.intersect(r, rc ? &cPrim : nullptr);
                    if (rc) {
                        rc->closest += 1; rc->closestInner += cPrim.n_inner; rc->closestTri += cPrim.n_tri;
                        rc->primRays += 1; rc->primInner += cPrim.n_inner; rc->primTri += cPrim.n_tri;
                    }
                    int id = yB * 2 + xB;
                    statusBlock[id] = hit.status;
                    if (hit.status) {
                        vertexBlock[id] = s.bvh.postIntersect(r, hit, s.haveFrame);
                        modelBlock[id] = hit.id0;
                        raysBlock[id] = r;
                    } else {
                        vec3 Lm{0.f, 0.f, 0.f};
                        for (const Light& l : s.lights) Lm = Lm + l.le(r);
                        float* o = film + ((size_t)y * W + x) * 4;
                        o[0] = Lm.x; o[1] = Lm.y; o[2] = Lm.z; o[3] = 1.0f;
                    }
                }
            }
            for (int yB = 0; yB < 2; ++yB) {
                int y = yBlock + yB;
                if (y >= H) break;
                for (int xB = 0; xB < 2; ++xB) {
                    int x = xBlock + xB;
                    if (x >= W) break;
                    if (!statusBlock[yB * 2 + xB]) continue;
                    const Vertex& vertex = vertexBlock[yB * 2 + xB];
                    vec2 dUVx = {vertexBlock[yB * 2 + 1].uv.x - vertexBlock[yB * 2 + 0].uv.x,
                                 vertexBlock[yB * 2 + 1].uv.y - vertexBlock[yB * 2 + 0].uv.y};
                    vec2 dUVy = {vertexBlock[1 * 2 + xB].uv.x - vertexBlock[xB].uv.x,
                                 vertexBlock[1 * 2 + xB].uv.y - vertexBlock[xB].uv.y};
                    dUVx = {dUVx.x * ptSize.x, dUVx.y * ptSize.y};
                    dUVy = {dUVy.x * ptSize.x, dUVy.y * ptSize.y};
                    if (duvOut) {  // test hook: the differences this pixel's primary hit was filtered with
                        float* o = duvOut + ((size_t)y * W + x) * 4;
                        o[0] = dUVx.x; o[1] = dUVx.y; o[2] = dUVy.x; o[3] = dUVy.y;
                    }
                    BSDF bsdf = computeBSDF(s, vertex, modelBlock[yB * 2 + xB], dUVx, dUVy);
                    shadePixel(s, x, y, W, maxDepth, seed, pass, frames, film, raysBlock[yB * 2 + xB], vertex, bsdf, rc, cap);
                }
            }
        }
    }
}

// One sample pass over the film: src/cpu/Integrator.cpp:26-67 (64x64 tiles
// pulled from an atomic counter by `threads` workers; the tile grid uses the
// height for its y extent, where the reference's width only adds empty tiles).
inline void renderPass(const Scene& s, const Camera& cam, int W, int H, int maxDepth, uint64_t seed,
                       uint32_t pass, float frames, float* film, int threads, RenderCounters* rc, int tileSize = 64,
                       float* duvOut = nullptr, const RayCapture* cap = nullptr) {
    int tx = (W + tileSize - 1) / tileSize, ty = (H + tileSize - 1) / tileSize;
    std::atomic<int> tileCurr{0};
    int tileCount = tx * ty;
    auto worker = [&]() {
        for (;;) {
            int t = tileCurr++;
            if (t >= tileCount) break;
            computeTile(s, cam, (t % tx) * tileSize, (t / tx) * tileSize, W, H, tileSize, maxDepth, seed, pass, frames, film, rc, duvOut, cap);
        }
    };
    if (threads <= 1) { worker(); return; }
    std::vector<std::thread> pool;
    for (int i = 0; i < threads; ++i) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
}

}  // namespace orc
