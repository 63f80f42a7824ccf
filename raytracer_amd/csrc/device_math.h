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

// Transcendentals.  The reference calls libm, whose last bits no device library reproduces; the kernels use fixed
// single-precision polynomials instead (Cephes-style: Cody-Waite reduction, minimax polynomials), every step one
// IEEE operation, so that a CPU restatement of the same steps yields the same bits and films can be compared
// exactly.  Accuracy against the exact functions: sin/cos 1.5 ulp (absolute 8e-8 near their zeros), acos 1.3 ulp,
// atan2 3.1 ulp, log2 1.7 ulp; log and exp (the microfacet lobe) 1 ulp class, pow = exp(y log x).
// sine and cosine of |x| < 1e4 (this path's arguments lie in [-pi/4, 2 pi]); NaN otherwise
RT_DEV void dm_sincos(float x, float& s, float& c) {
    const float ax = fabsf(x);
    if (!(ax < 1e4f)) { s = c = __uint_as_float(0x7FC00000u); return; }
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
RT_DEV float dm_sin(float x) { float s, c; dm_sincos(x, s, c); return s; }
RT_DEV float dm_cos(float x) { float s, c; dm_sincos(x, s, c); return c; }
RT_DEV float dm_asin_poly(float x) {   // |x| <= 0.5
    const float z = x * x;
    return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * x + x;
}
RT_DEV float dm_acos(float x) {        // x in [-1, 1]
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * dm_asin_poly(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * dm_asin_poly(sqrtf(0.5f * (1.0f - x)));
    return 1.57079632679489661923f - dm_asin_poly(x);
}
RT_DEV float dm_atan_pos(float x) {    // x >= 0 (inf included)
    float y0 = 0.0f;
    if (x > 2.414213562373095f) { y0 = 1.57079632679489661923f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y0 = 0.78539816339744830962f; x = (x - 1.0f) / (x + 1.0f); }
    const float z = x * x;
    return y0 + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
}
RT_DEV float dm_atan2(float y, float x) {   // (-pi, pi]; atan2(0, 0) = 0 whatever the signs of the zeros
    const float ax = fabsf(x), ay = fabsf(y);
    if (ax == 0.0f && ay == 0.0f) return 0.0f;
    float a = dm_atan_pos(ay / ax);
    if (x < 0.0f) a = 3.14159265358979323846f - a;
    return (y < 0.0f) ? -a : a;
}
RT_DEV float dm_log2(float x) {             // positive normal x; +inf gives 128
    const uint32_t b = __float_as_uint(x);
    int e = (int)(b >> 23) - 127;
    float m = __uint_as_float((b & 0x007FFFFFu) | 0x3F800000u);   // [1, 2)
    if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f, z = f * f;
    float y = ((((((((7.0376836292e-2f * f - 1.1514610310e-1f) * f + 1.1676998740e-1f) * f - 1.2420140846e-1f) * f + 1.4249322787e-1f) * f -
                  1.6668057665e-1f) * f + 2.0000714765e-1f) * f - 2.4999993993e-1f) * f + 3.3333331174e-1f) * f * z;
    y = y - 0.5f * z;
    return (float)e + (f + y) * 1.44269504088896341f;
}

// natural logarithm of x >= 0 (zero gives -inf, denormals are scaled up first); NaN for negative x and NaN
RT_DEV float dm_log(float x) {
    if (!(x >= 0.0f)) return __uint_as_float(0x7FC00000u);
    if (x == 0.0f) return -__builtin_inff();
    if (!(x < __builtin_inff())) return x;
    int e = 0;
    if (x < 1.17549435e-38f) { x = x * 8388608.0f; e = -23; }
    const uint32_t b = __float_as_uint(x);
    e += (int)(b >> 23) - 127;
    float m = __uint_as_float((b & 0x007FFFFFu) | 0x3F800000u);   // [1, 2)
    if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f, z = f * f, fe = (float)e;
    float y = ((((((((7.0376836292e-2f * f - 1.1514610310e-1f) * f + 1.1676998740e-1f) * f - 1.2420140846e-1f) * f + 1.4249322787e-1f) * f -
                  1.6668057665e-1f) * f + 2.0000714765e-1f) * f - 2.4999993993e-1f) * f + 3.3333331174e-1f) * f * z;
    y = y + -2.12194440e-4f * fe;
    y = y - 0.5f * z;
    return (f + y) + 0.693359375f * fe;
}
RT_DEV float dm_pow2i(int k) { return __uint_as_float((uint32_t)(k + 127) << 23); }   // 2^k, k in [-126, 127]
// e^x: +inf above 88.72, 0 below -103.9 (-inf included); NaN for NaN
RT_DEV float dm_exp(float x) {
    if (x != x) return x;
    if (x > 88.72283905206835f) return __builtin_inff();
    if (x < -103.9f) return 0.0f;
    const float n = floorf(1.44269504088896341f * x + 0.5f);
    x = x - n * 0.693359375f;
    x = x - n * -2.12194440e-4f;
    const float z = x * x;
    const float p = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x +
                     5.0000001201e-1f) * z + x + 1.0f;
    const int k = (int)n, k1 = k / 2;
    return p * dm_pow2i(k1) * dm_pow2i(k - k1);
}
// x^y for x >= 0 as exp(y log x) (the lobe raises a number of (0, 1] to a power near 1)
RT_DEV float dm_pow(float x, float y) { return dm_exp(y * dm_log(x)); }

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
    // 16 bits each for the dimension and the low half of the pass number, 32 for the pixel; the high half of the
    // pass number re-keys the stream, so accumulation can run past 65535 passes without repeating a sequence
    const uint32_t pass_hi = pass >> 16;
    if (pass_hi) seed_mixed = splitmix64(seed_mixed ^ ((uint64_t)pass_hi << 32));
    uint64_t ctr = ((uint64_t)(pass & 0xFFFFu) << 48) ^ ((uint64_t)pixel << 16) ^ (uint64_t)dim;
    uint64_t z = splitmix64(seed_mixed ^ ctr);
    return (float)(z >> 40) * 0x1p-24f;
}

}  // namespace rt
