// Device-side scalar maths for the gfx950 kernels.  Every expression keeps
// the operand order of the reference's glm-based code so that, compiled with
// -ffp-contract=off, each float operation is the same single IEEE-754
// round-to-nearest operation the scalar CPU path performs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

#define RT_DEV __device__ __forceinline__

struct f3 { float x, y, z; };
struct f2 { float x, y; };

RT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
RT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_DEV f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
RT_DEV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
RT_DEV f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
// glm: dot = a.x*b.x + a.y*b.y + a.z*b.z, left to right
RT_DEV float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// glm: cross(x,y) = (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y)
RT_DEV f3 cross3(f3 x, f3 y) { return mk3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
// glm: min(x,y) = (y<x)?y:x ; max(x,y) = (x<y)?y:x
RT_DEV float gmin(float x, float y) { return (y < x) ? y : x; }
RT_DEV float gmax(float x, float y) { return (x < y) ? y : x; }
// std::min(a,b) = (b<a)?b:a ; std::max(a,b) = (a<b)?b:a  (same selections)
// glm: normalize(v) = v * (1/sqrt(dot(v,v)))
RT_DEV f3 normalize3(f3 v) { return v * (1.0f / sqrtf(dot3(v, v))); }
RT_DEV bool is_zero3(f3 v) { return v.x == 0.0f && v.y == 0.0f && v.z == 0.0f; }
// src/scene/Common.hpp:9-11
RT_DEV float clampf(float v, float lo, float hi) { return gmax(gmin(v, hi), lo); }
// src/scene/Common.hpp:27-30
RT_DEV float power_heuristic(float fPdf, float gPdf) {
    float f = fPdf, g = gPdf;
    return (f * f) / (f * f + g * g);
}

// Counter-based sampler: one draw is a pure function of (seed, pixel, pass,
// dimension).  Stands in for the per-tile std::mt19937(std::random_device())
// of src/cpu/PathTracer.cpp:20-22, which no run can reproduce.
RT_DEV uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
RT_DEV float rng_u01(uint64_t seed_mixed, uint32_t pixel, uint32_t pass, uint32_t dim) {
    uint64_t ctr = ((uint64_t)pass << 48) ^ ((uint64_t)pixel << 16) ^ (uint64_t)dim;
    uint64_t z = splitmix64(seed_mixed ^ ctr);
    return (float)(z >> 40) * 0x1p-24f;
}

}  // namespace rt
