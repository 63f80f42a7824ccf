// Wavefront path-tracing stages for gfx950 (everything of
// PathTracer::computeTile, src/cpu/PathTracer.cpp:17-164, that is not BVH
// traversal): camera-ray generation, per-bounce shading with light sampling
// and BSDF sampling, shadow-ray resolution, film accumulation.  Path state
// is SoA; a slot is a film pixel of this rank in one pass of the batch, and
// the paths alive at a bounce occupy consecutive indices of the state arrays
// (compacted by k_shade with one atomic per workgroup).
#pragma once
#include "trace.h"
#include "trace_wide.h"

namespace rt {

#define RT_LIGHT_POINT_K 0
#define RT_LIGHT_AREA_K 2
#define RT_LIGHT_ENV_K 3

#define RT_GLOSSY_NONE_K 0
#define RT_GLOSSY_BECKMANN_K 1
#define RT_GLOSSY_GGX_K 2

struct DevLight {
    int32_t type;
    float px, py, pz;      // point position
    float ix, iy, iz;      // point intensity / env scale
    int32_t W, H;          // env image size
    const float4* img;     // env texels, row-major
    // Distribution2D built as EnvLight's ctor does (src/lights/EnvLight.cpp:16-25):
    // W conditional rows of H entries each, marginal over the W rows.  Each Distribution1D of n entries is
    // stored as n + 1 records (DistEntry below), so that the guided search reads its bracket of the guide
    // table with one 4-byte load and the interval's CDF pair and function value with one 16-byte load.
    const float4* cond_pack; // [W][H+1]
    const float* cond_int;   // [W]: funcInt of each conditional row
    const float4* marg_pack; // [W+1]
    float marg_int;
    // area: one emissive triangle, radiance (ix, iy, iz) on the side of n = cross(p1-p0, p2-p0)
    float p0[3], p1[3], p2[3], n[3];
    float area;
};

struct PathBuffers {
    // State of the paths of the current bounce, indexed by PATH INDEX: at bounce
    // 0 the path index is the slot; from bounce 1 on, k_shade writes the
    // surviving paths to consecutive indices of the *_next arrays, so every
    // bounce reads and writes dense, coalesced planes however few paths are left.
    float* rays;        // 6 planes: ox oy oz dx dy dz
    float* beta;        // 3 planes (the reference's `threshold`)
    uint32_t* slot;     // slot of each path
    float* rays_next;
    float* beta_next;
    uint32_t* slot_next;
    float* hit_t;       // hit of each path's ray, by path index
    int32_t* hit_id0;
    int32_t* hit_id1;
    float* hit_u;
    float* hit_v;
    float* vpos;        // 3 planes: position of the vertex being shaded, by path index
    // Per slot (a film pixel of one pass of the batch)
    float* L;           // 3 planes (the reference's `pixelColor`)
    uint32_t* status;   // 1 = the primary ray hit (film takes the running mean)
    const uint32_t* slot_pixel;  // y*W+x, or 0xFFFFFFFF for a slot outside the film
    float4* film;       // n1 entries: one per film pixel slot
    uint32_t n;         // slots = plane stride: n1 * (passes traced together)
    uint32_t n1;        // slots of one pass; slot s belongs to pass s / n1 of the batch and to film slot s % n1
    // What cannot change from pass to pass while the camera stands still, per FILM slot (n1 entries a plane): the
    // camera ray has no jitter (src/cpu/PathTracer.cpp:43-50), so the primary ray and its hit are the same in every
    // pass, and a point light's sample ignores the random numbers (src/lights/PointLight.cpp:14-18), so the shadow
    // ray from the primary hit towards it is too.  Filled by the first pass after a camera change.
    float* prim_rays;   // 6 planes of n1
    float* prim_t;      // primary hit, by film slot
    int32_t* prim_id0;
    int32_t* prim_id1;
    float* prim_u;
    float* prim_v;
    uint8_t* prim_vis;  // [light][n1]: shadow-entry state of the primary vertex towards each point light
                        // (0 no ray, 1 occluded, 2 unoccluded)
};

// One mip level of a diffuse map inside the shared texel buffer.
struct TexLevel {
    uint32_t offset;   // first texel
    int32_t w, h;
};
// RGBA8 diffuse map (the reference's spImage4b, src/scene/Material.hpp:61):
// levels first_level .. first_level + n_levels - 1 of the level table.
struct DevTexture {
    int32_t first_level, n_levels;
};

struct ShadeScene {
    const float* tri_attr;      // 36 floats per triangle: 3 x (normal3 uv2 tangent3 binormal3), 3 pad
    const float4* materials;    // per model: rgb, roughness
    const int32_t* material_texture;  // per model: texture index or -1
    const int32_t* material_glossy;   // per model: 0 none, 1 Beckmann, 2 GGX lobe; null when no model has one
    const DevTexture* textures;       // null when the scene has no diffuse maps
    const TexLevel* tex_levels;
    const uchar4* texels;
    const float4* duv;          // per slot: dUVx.xy, dUVy.xy of the primary hit, in film pixels (textured scenes only)
    const DevLight* lights;
    int32_t n_lights;
    int32_t have_frame;
    int32_t env_staged;         // the environment light whose marginal distribution k_shade keeps in LDS (its W + 1
                                // records fit RT_MARG_LDS), or -1
};

#define RT_MARG_LDS 2049        // records of a marginal distribution k_shade can stage in LDS (32 KB)

struct CameraK {
    float pos[3], right[3], up[3], forward[3];
    int32_t W, H;
};

constexpr float kPi = 3.14159265358979323846f;  // (float)M_PI
constexpr float kInvPi = 1.0f / kPi;            // src/bsdf/Samples.hpp:10-13
constexpr float kInv2Pi = 1.0f / (2.f * kPi);
constexpr float kPiOver2 = kPi / 2.f;
constexpr float kPiOver4 = kPiOver2 / 2.f;

// Appends the lanes with pred set to a queue: one atomic per wave.  Must be
// reached by every lane of the wave.
RT_DEV uint32_t wave_append(bool pred, uint32_t* counter) {
    unsigned long long m = __ballot(pred);
    uint32_t lane = __lane_id();
    uint32_t prefix = __popcll(m & ((1ull << lane) - 1ull));
    uint32_t base = 0;
    if (m) {
        int leader = __ffsll((unsigned long long)m) - 1;
        if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
        base = __shfl(base, leader);
    }
    return base + prefix;
}

// The same for a whole workgroup: one atomic per workgroup.  Same-address
// atomics retire at roughly 90 per microsecond on this part, so a queue that
// every wave appends to by itself caps a kernel at ~6 G lanes/s.  Must be
// reached by every thread of the workgroup; s_cnt holds RT_BLOCK/64 + 1 words.
RT_DEV uint32_t block_append(bool pred, uint32_t* counter, uint32_t* s_cnt) {
    constexpr uint32_t WAVES = RT_BLOCK / 64;
    unsigned long long m = __ballot(pred);
    const uint32_t lane = __lane_id(), w = threadIdx.x >> 6;
    if (lane == 0) s_cnt[w] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (uint32_t k = 0; k < WAVES; ++k) {
            uint32_t c = s_cnt[k];
            s_cnt[k] = tot;
            tot += c;
        }
        s_cnt[WAVES] = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    uint32_t r = s_cnt[WAVES] + s_cnt[w] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    return r;
}

// --------------------------------------------------------------- sampling
// src/bsdf/Samples.hpp:15-28
RT_DEV f2 concentric_sample_disk(f2 u) {
    f2 uo; uo.x = 2.f * u.x - 1.f; uo.y = 2.f * u.y - 1.f;
    f2 r0; r0.x = 0.f; r0.y = 0.f;
    if (uo.x == 0 && uo.y == 0) return r0;
    float theta, r;
    if (fabsf(uo.x) > fabsf(uo.y)) {
        r = uo.x;
        theta = kPiOver4 * (uo.y / uo.x);
    } else {
        r = uo.y;
        theta = kPiOver2 - kPiOver4 * (uo.x / uo.y);
    }
    f2 o; o.x = r * dm_cos(theta); o.y = r * dm_sin(theta);
    return o;
}
// src/bsdf/Samples.hpp:30-34: the cosine-weighted axis is written to y
RT_DEV f3 cosine_sample_hemisphere(f2 u) {
    f2 d = concentric_sample_disk(u);
    float z = sqrtf(gmax(0.f, 1.f - d.x * d.x - d.y * d.y));
    return mk3(d.x, z, d.y);
}
// src/bsdf/Samples.hpp:36-43
RT_DEV float spherical_theta(f3 v) { return dm_acos(clampf(v.y, -1.f, 1.f)); }
RT_DEV float spherical_phi(f3 v) {
    float p = dm_atan2(-v.z, -v.x);
    return (p < 0.f) ? (p + 2.f * kPi) : p;
}

// ------------------------------------------------------------------- BSDF
// src/bsdf/BxDF.hpp:38-64
RT_DEV float abs_cos_theta(f3 w) { return fabsf(w.z); }
RT_DEV float sin2_theta(f3 w) { return gmax(0.f, 1.f - w.z * w.z); }
RT_DEV float sin_theta(f3 w) { return sqrtf(sin2_theta(w)); }
RT_DEV float cos_phi(f3 w) { float s = sin_theta(w); return (s == 0) ? 1.f : clampf(w.x / s, -1.f, 1.f); }
RT_DEV float sin_phi(f3 w) { float s = sin_theta(w); return (s == 0) ? 0.f : clampf(w.y / s, -1.f, 1.f); }

// Oren–Nayar lobe, src/bsdf/Diffuse.cpp:7-32
struct DiffuseK {
    f3 color;
    float A, B;
};
RT_DEV DiffuseK make_diffuse(f3 color, float sigma) {
    DiffuseK d;
    d.color = color;
    float sigma2 = sigma * sigma;
    d.A = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
    d.B = 0.45f * sigma2 / (sigma2 + 0.09f);
    return d;
}
RT_DEV f3 diffuse_f(const DiffuseK& k, f3 out, f3 in) {
    float sinThetaI = sin_theta(in), sinThetaO = sin_theta(out);
    float maxCos = 0;
    if ((double)sinThetaI > 1e-4 && (double)sinThetaO > 1e-4) {
        float sinPhiI = sin_phi(in), cosPhiI = cos_phi(in);
        float sinPhiO = sin_phi(out), cosPhiO = cos_phi(out);
        float dCos = cosPhiI * cosPhiO + sinPhiI * sinPhiO;
        maxCos = gmax(0.f, dCos);
    }
    float sinAlpha, tanBeta;
    if (abs_cos_theta(in) > abs_cos_theta(out)) {
        sinAlpha = sinThetaO;
        tanBeta = sinThetaI / abs_cos_theta(in);
    } else {
        sinAlpha = sinThetaI;
        tanBeta = sinThetaO / abs_cos_theta(out);
    }
    return k.color * kInvPi * (k.A + k.B * maxCos * sinAlpha * tanBeta);
}
// src/bsdf/BxDF.cpp:15-17
RT_DEV float diffuse_pdf(f3 out, f3 in) { return (out.z * in.z > 0) ? abs_cos_theta(in) * kInvPi : 0.f; }
// src/bsdf/BxDF.cpp:8-13
RT_DEV f3 diffuse_sampled(const DiffuseK& k, f3 out, f2 u, f3& in, float& pdf) {
    in = cosine_sample_hemisphere(u);
    if (out.z < 0) in.z *= -1;
    pdf = diffuse_pdf(out, in);
    return diffuse_f(k, out, in);
}

// --------------------------------------------------------- glossy lobe
// Torrance-Sparrow reflection lobe of src/bsdf/Microfacet.cpp (which
// Material::computeBSDF only mentions in a comment, src/scene/Material.cpp:205-209)
// as an option of a material.  Two corrections, without which the lobe cannot
// be sampled: the reflected direction is -out + 2 (out.wh) wh (Microfacet.cpp:216
// mirrors an incident vector instead), and the Beckmann slope sampler uses the
// inverse error function where Microfacet.cpp:57,76 calls erfc.  GGX is the
// published Trowbridge-Reitz distribution (pbrt-v3, src/core/microfacet.cpp).
RT_DEV float cos2_theta(f3 w) { return w.z * w.z; }
RT_DEV float tan_theta(f3 w) { return sin_theta(w) / w.z; }
RT_DEV float tan2_theta(f3 w) { return sin2_theta(w) / cos2_theta(w); }
RT_DEV float cos2_phi(f3 w) { return cos_phi(w) * cos_phi(w); }
RT_DEV float sin2_phi(f3 w) { return sin_phi(w) * sin_phi(w); }

// pbrt-v3 src/core/pbrt.h ErfInv
RT_DEV float erf_inv(float x) {
    float w, p;
    x = clampf(x, -.99999f, .99999f);
    w = -dm_log((1 - x) * (1 + x));
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
        w = sqrtf(w) - 3;
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
// pbrt-v3 src/core/pbrt.h Erf, in place of the libm erf of Microfacet.cpp:30
RT_DEV float erf_poly(float x) {
    const float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f, a4 = -1.453152027f, a5 = 1.061405429f;
    const float p = 0.3275911f;
    int sign = 1;
    if (x < 0) sign = -1;
    x = fabsf(x);
    float t = 1 / (1 + p * x);
    float y = 1 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * dm_exp(-x * x);
    return sign * y;
}

// src/bsdf/Microfacet.cpp:7-80
RT_DEV void beckmann_sample11(float cosThetaI, float U1, float U2, float& slope_x, float& slope_y) {
    if ((double)cosThetaI > .9999) {
        float r = sqrtf(-dm_log(1.0f - U1));
        float sinPhi = dm_sin(2 * kPi * U2);
        float cosPhi = dm_cos(2 * kPi * U2);
        slope_x = r * cosPhi;
        slope_y = r * sinPhi;
        return;
    }
    float sinThetaI = sqrtf(gmax(0.f, 1.f - cosThetaI * cosThetaI));
    float tanThetaI = sinThetaI / cosThetaI;
    float cotThetaI = 1 / tanThetaI;
    float a = -1, c = erf_poly(cotThetaI);
    float sample_x = gmax(U1, 1e-6f);
    float thetaI = dm_acos(cosThetaI);
    float fit = 1 + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
    float b = c - (1 + c) * dm_pow(1 - sample_x, fit);
    const float SQRT_PI_INV = 1.f / sqrtf(kPi);
    float normalization = 1 / (1 + c + SQRT_PI_INV * tanThetaI * dm_exp(-cotThetaI * cotThetaI));
    int it = 0;
    while (++it < 10) {
        if (!(b >= a && b <= c)) b = 0.5f * (a + c);
        float invErf = erf_inv(b);
        float value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * dm_exp(-invErf * invErf)) - sample_x;
        float derivative = normalization * (1 - invErf * tanThetaI);
        if (fabsf(value) < 1e-5f) break;
        if (value > 0)
            c = b;
        else
            a = b;
        b -= value / derivative;
    }
    slope_x = erf_inv(b);
    slope_y = erf_inv(2.0f * gmax(U2, 1e-6f) - 1.0f);
}
// pbrt-v3 TrowbridgeReitzSample11
RT_DEV void ggx_sample11(float cosTheta, float U1, float U2, float& slope_x, float& slope_y) {
    if ((double)cosTheta > .9999) {
        float r = sqrtf(U1 / (1 - U1));
        float phi = 6.28318530718f * U2;
        slope_x = r * dm_cos(phi);
        slope_y = r * dm_sin(phi);
        return;
    }
    float sinTheta = sqrtf(gmax(0.f, 1.f - cosTheta * cosTheta));
    float tanTheta = sinTheta / cosTheta;
    float a = 1 / tanTheta;
    float G1 = 2 / (1 + sqrtf(1.f + 1.f / (a * a)));
    float A = 2 * U1 / G1 - 1;
    float tmp = 1.f / (A * A - 1.f);
    if (tmp > 1e10f) tmp = 1e10f;
    float B = tanTheta;
    float D = sqrtf(gmax(B * B * tmp * tmp - (A * A - B * B) * tmp, 0.f));
    float slope_x_1 = B * tmp - D;
    float slope_x_2 = B * tmp + D;
    slope_x = (A < 0 || slope_x_2 > 1.f / tanTheta) ? slope_x_1 : slope_x_2;
    float S;
    if (U2 > 0.5f) {
        S = 1.f;
        U2 = 2.f * (U2 - .5f);
    } else {
        S = -1.f;
        U2 = 2.f * (.5f - U2);
    }
    float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) / (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
    slope_y = S * z * sqrtf(1.f + slope_x * slope_x);
}

// MicrofacetDistribution with visible-area sampling, src/bsdf/Microfacet.hpp:9-50
struct MicrofacetK {
    int kind;
    float ax, ay;
};
RT_DEV float mf_D(const MicrofacetK& m, f3 wh) {
    float t2 = tan2_theta(wh);
    if (isinf(t2)) return 0.f;
    float cos4Theta = cos2_theta(wh) * cos2_theta(wh);
    if (m.kind == RT_GLOSSY_BECKMANN_K)
        return dm_exp(-t2 * (cos2_phi(wh) / (m.ax * m.ax) + sin2_phi(wh) / (m.ay * m.ay))) / (kPi * m.ax * m.ay * cos4Theta);
    float e = (cos2_phi(wh) / (m.ax * m.ax) + sin2_phi(wh) / (m.ay * m.ay)) * t2;
    return 1 / (kPi * m.ax * m.ay * cos4Theta * (1 + e) * (1 + e));
}
RT_DEV float mf_lambda(const MicrofacetK& m, f3 w) {
    float absTanTheta = fabsf(tan_theta(w));
    if (isinf(absTanTheta)) return 0.f;
    float alpha = sqrtf(cos2_phi(w) * m.ax * m.ax + sin2_phi(w) * m.ay * m.ay);
    if (m.kind == RT_GLOSSY_BECKMANN_K) {
        float a = 1 / (alpha * absTanTheta);
        if (a >= 1.6f) return 0;
        return (1 - 1.259f * a + 0.396f * a * a) / (3.535f * a + 2.181f * a * a);
    }
    float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
    return (-1 + sqrtf(1.f + alpha2Tan2Theta)) / 2;
}
RT_DEV float mf_G1(const MicrofacetK& m, f3 w) { return 1 / (1 + mf_lambda(m, w)); }
RT_DEV float mf_G(const MicrofacetK& m, f3 wo, f3 wi) { return 1 / (1 + mf_lambda(m, wo) + mf_lambda(m, wi)); }
RT_DEV f3 mf_sample_wh(const MicrofacetK& m, f3 wo, f2 u) {
    bool flip = wo.z < 0;
    f3 wi = flip ? -wo : wo;
    f3 ws = normalize3(mk3(m.ax * wi.x, m.ay * wi.y, wi.z));
    float slope_x, slope_y;
    if (m.kind == RT_GLOSSY_BECKMANN_K)
        beckmann_sample11(ws.z, u.x, u.y, slope_x, slope_y);
    else
        ggx_sample11(ws.z, u.x, u.y, slope_x, slope_y);
    float tmp = cos_phi(ws) * slope_x - sin_phi(ws) * slope_y;
    slope_y = sin_phi(ws) * slope_x + cos_phi(ws) * slope_y;
    slope_x = tmp;
    slope_x = m.ax * slope_x;
    slope_y = m.ay * slope_y;
    f3 wh = normalize3(mk3(-slope_x, -slope_y, 1.f));
    if (flip) wh = -wh;
    return wh;
}
RT_DEV float mf_pdf(const MicrofacetK& m, f3 wo, f3 wh) { return mf_D(m, wh) * mf_G1(m, wo) * fabsf(dot3(wo, wh)) / abs_cos_theta(wo); }

RT_DEV f3 sqrt3(f3 v) { return mk3(sqrtf(v.x), sqrtf(v.y), sqrtf(v.z)); }
RT_DEV f3 splat3(float a) { return mk3(a, a, a); }
// src/bsdf/Microfacet.cpp:176-198
RT_DEV f3 fresnel_conductor(f3 etaI, f3 etaT, f3 k, float cosI) {
    float cosThetaI = clampf(cosI, -1, 1);
    f3 eta = etaT / etaI;
    f3 etak = k / etaI;
    float cosThetaI2 = cosThetaI * cosThetaI;
    float sinThetaI2 = 1.f - cosThetaI2;
    f3 eta2 = eta * eta;
    f3 etak2 = etak * etak;
    f3 t0 = eta2 - etak2 - splat3(sinThetaI2);
    f3 a2plusb2 = sqrt3(t0 * t0 + 4.f * eta2 * etak2);
    f3 t1 = a2plusb2 + splat3(cosThetaI2);
    f3 a = sqrt3(0.5f * (a2plusb2 + t0));
    f3 t2 = 2.f * cosThetaI * a;
    f3 Rs = (t1 - t2) / (t1 + t2);
    f3 t3 = cosThetaI2 * a2plusb2 + splat3(sinThetaI2 * sinThetaI2);
    f3 t4 = t2 * sinThetaI2;
    f3 Rp = Rs * (t3 - t4) / (t3 + t4);
    return 0.5f * (Rp + Rs);
}

// src/bsdf/Microfacet.cpp:200-228 with the parameters of Material.cpp:205-208:
// R = 1, FresnelConductor(0.5, 0.5, diffColor)
struct ReflectionK {
    MicrofacetK d;
    f3 k;
};
RT_DEV f3 reflection_f(const ReflectionK& r, f3 wo, f3 wi) {
    float cosThetaO = abs_cos_theta(wo), cosThetaI = abs_cos_theta(wi);
    f3 wh = wi + wo;
    if (cosThetaI == 0 || cosThetaO == 0) return mk3(0.f, 0.f, 0.f);
    if (wh.x == 0 && wh.y == 0 && wh.z == 0) return mk3(0.f, 0.f, 0.f);
    wh = normalize3(wh);
    f3 F = fresnel_conductor(splat3(0.5f), splat3(0.5f), r.k, dot3(wi, wh));
    return mk3(1.f, 1.f, 1.f) * mf_D(r.d, wh) * mf_G(r.d, wo, wi) * F / (4 * cosThetaI * cosThetaO);
}
RT_DEV f3 reflection_sampled(const ReflectionK& r, f3 out, f2 u, f3& in, float& pdf) {
    if (out.z == 0) return mk3(0.f, 0.f, 0.f);
    f3 wh = mf_sample_wh(r.d, out, u);
    in = -out + 2.f * dot3(out, wh) * wh;
    if (!(out.z * in.z > 0)) return mk3(0.f, 0.f, 0.f);
    pdf = mf_pdf(r.d, out, wh) / (4.f * dot3(out, wh));
    return reflection_f(r, out, in);
}
RT_DEV float reflection_pdf(const ReflectionK& r, f3 wo, f3 wi) {
    if (!(wo.z * wi.z > 0)) return 0;
    f3 wh = normalize3(wo + wi);
    return mf_pdf(r.d, wo, wh) / (4.f * dot3(wo, wh));
}

// BSDF, src/bsdf/BSDF.cpp:7-104: the Diffuse lobe, plus the Reflection lobe
// when the material carries one (GLOSSY kernels only)
struct BsdfK {
    f3 t, b, n;
    DiffuseK lobe;
    ReflectionK glossy;
    int parts;
};
RT_DEV f3 world2local(const BsdfK& s, f3 v) { return mk3(dot3(s.t, v), dot3(s.b, v), dot3(s.n, v)); }
RT_DEV f3 local2world(const BsdfK& s, f3 v) {
    return mk3(s.t.x * v.x + s.b.x * v.y + s.n.x * v.z, s.t.y * v.x + s.b.y * v.y + s.n.y * v.z,
               s.t.z * v.x + s.b.z * v.y + s.n.z * v.z);
}
template <bool GLOSSY>
RT_DEV f3 bsdf_f(const BsdfK& s, f3 outW, f3 inW) {
    f3 in = world2local(s, inW), out = world2local(s, outW);
    bool reflect = dot3(inW, s.n) * dot3(outW, s.n) > 0;
    f3 r = mk3(0.f, 0.f, 0.f);
    if (reflect) r = r + diffuse_f(s.lobe, out, in);
    if (GLOSSY && reflect && s.parts > 1) r = r + reflection_f(s.glossy, out, in);
    return r;
}
template <bool GLOSSY>
RT_DEV float bsdf_pdf(const BsdfK& s, f3 outW, f3 inW) {
    f3 out = world2local(s, outW), in = world2local(s, inW);
    float p = 0.f;
    p += diffuse_pdf(out, in);
    if (GLOSSY && s.parts > 1) {
        p += reflection_pdf(s.glossy, out, in);
        p /= 2.0f;
    } else {
        p /= 1.0f;
    }
    return p;
}
template <bool GLOSSY>
RT_DEV f3 bsdf_sample(const BsdfK& s, f3 outW, f2 u, f3& inW, float& pdf) {
    pdf = 0.f;
    const int mc = (GLOSSY && s.parts > 1) ? 2 : 1;
    int comp = min((int)floorf(u.x * mc), mc - 1);
    f2 uR; uR.x = u.x * mc - comp; uR.y = u.y;
    f3 wi = mk3(0.f, 0.f, 0.f), wo = world2local(s, outW);
    f3 fr;
    if (GLOSSY && comp == 1) fr = reflection_sampled(s.glossy, wo, uR, wi, pdf);
    else fr = diffuse_sampled(s.lobe, wo, uR, wi, pdf);
    if (pdf == 0) return mk3(0.f, 0.f, 0.f);
    inW = local2world(s, wi);
    if (GLOSSY && mc > 1) {
        pdf += comp == 0 ? reflection_pdf(s.glossy, wo, wi) : diffuse_pdf(wo, wi);
        pdf /= mc;
        bool reflect = dot3(inW, s.n) * dot3(outW, s.n) > 0;
        fr = mk3(0.f, 0.f, 0.f);
        if (reflect) {
            fr = fr + diffuse_f(s.lobe, wo, wi);
            fr = fr + reflection_f(s.glossy, wo, wi);
        }
    }
    return fr;
}

// ---------------------------------------------------------- distributions
// src/scene/Common.hpp:13-25 with the predicate cdf[i] <= u
RT_DEV int find_interval_cdf(const float* cdf, int size, float u) {
    int first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (cdf[middle] <= u) {
            first = middle + 1;
            len -= half + 1;
        } else
            len = half;
    }
    return (int)clampf((float)(first - 1), 0.f, (float)(size - 2));
}
// Record i of a packed Distribution1D with n entries (i = 0..n):
//   x = cdf[i], y = cdf[i+1] (i < n), z = func[i] (i < n),
//   w = bits: guide bracket of bucket i = (lo | hi << 16) with lo = guide[max(i-1, 0)], hi = guide[min(i+2, n)],
// guide[b] = number of CDF entries <= b/n.  The CDF is non-decreasing, so the partition point of `cdf[k] <= u`
// that findInterval looks for (src/scene/Common.hpp:13-25) is unique and lies inside the bracket of bucket
// floor(u*n) (one bucket of margin each side absorbs the rounding of u*n and b/n): the search below probes the same
// predicate over a shorter range and lands on the same interval.
// src/samplers/Distributions.cpp:26-35
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4* GlobalPack;   // pointers read out of a DevLight are generic to the compiler
typedef const __attribute__((address_space(1))) float* GlobalFloats;
typedef const __attribute__((address_space(3))) f32x4* LdsPack;   // said outright: a generic pointer would read through flat loads
template <typename Ptr>
RT_DEV float dist1d_sample(Ptr pack, float funcInt, int n, float u, float& pdf, int& off) {
    int b = (int)(u * (float)n);
    b = b < 0 ? 0 : (b > n - 1 ? n - 1 : b);
    const uint32_t g = __float_as_uint(pack[b].w);
    const int lo = (int)(g & 0xFFFFu), hi = (int)(g >> 16);
    // entries below lo are <= u and entries from hi on are > u by construction
    int first = lo, len = hi - lo;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (pack[middle].x <= u) {
            first = middle + 1;
            len -= half + 1;
        } else
            len = half;
    }
    const int offset = (int)clampf((float)(first - 1), 0.f, (float)(n - 1));
    off = offset;
    const f32x4 e = pack[offset];
    float du = u - e.x;
    if ((e.y - e.x) > 0) du /= (e.y - e.x);
    pdf = e.z / funcInt;
    return ((float)offset + du) / (float)n;
}

// Bilinear RGBA32F fetch (repeat wrap, texel centres at +0.5); the
// reference's filterLinear lives in its absent Mango dependency.
RT_DEV f3 filter_linear(const float4* img_, int W, int H, f2 uv) {
    const GlobalPack img = (GlobalPack)img_;
    float x = uv.x * (float)W - 0.5f, y = uv.y * (float)H - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int x0 = (int)fx0, y0 = (int)fy0;
    int xa = x0 % W; if (xa < 0) xa += W;
    int xb = (x0 + 1) % W; if (xb < 0) xb += W;
    int ya = y0 % H; if (ya < 0) ya += H;
    int yb = (y0 + 1) % H; if (yb < 0) yb += H;
    f32x4 p00 = img[(size_t)ya * W + xa], p10 = img[(size_t)ya * W + xb];
    f32x4 p01 = img[(size_t)yb * W + xa], p11 = img[(size_t)yb * W + xb];
    f3 top = mk3(p00.x, p00.y, p00.z) * (1.0f - fx) + mk3(p10.x, p10.y, p10.z) * fx;
    f3 bot = mk3(p01.x, p01.y, p01.z) * (1.0f - fx) + mk3(p11.x, p11.y, p11.z) * fx;
    return top * (1.0f - fy) + bot * fy;
}

// ------------------------------------------------------------ diffuse maps
// Bilinear fetch of one RGBA8 mip level, channels in 0..255 (repeat wrap,
// texel centres at +0.5).
RT_DEV f3 tex_bilinear(const ShadeScene& sc, const TexLevel lv, f2 uv) {
    const int W = lv.w, H = lv.h;
    const uchar4* img = sc.texels + lv.offset;
    float x = uv.x * (float)W - 0.5f, y = uv.y * (float)H - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int x0 = (int)fx0, y0 = (int)fy0;
    int xa = x0 % W; if (xa < 0) xa += W;
    int xb = (x0 + 1) % W; if (xb < 0) xb += W;
    int ya = y0 % H; if (ya < 0) ya += H;
    int yb = (y0 + 1) % H; if (yb < 0) yb += H;
    uchar4 p00 = img[(size_t)ya * W + xa], p10 = img[(size_t)ya * W + xb];
    uchar4 p01 = img[(size_t)yb * W + xa], p11 = img[(size_t)yb * W + xb];
    f3 top = mk3((float)p00.x, (float)p00.y, (float)p00.z) * (1.0f - fx) + mk3((float)p10.x, (float)p10.y, (float)p10.z) * fx;
    f3 bot = mk3((float)p01.x, (float)p01.y, (float)p01.z) * (1.0f - fx) + mk3((float)p11.x, (float)p11.y, (float)p11.z) * fx;
    return top * (1.0f - fy) + bot * fy;
}
// Stands in for filterTrilinear(*tex, uv, dUVx, dUVy) of src/scene/Material.cpp:198
// (the filter itself lives in the absent Mango library): lod = log2 of the
// longer footprint axis in level-0 texels, two bilinear fetches blended.
RT_DEV f3 tex_trilinear(const ShadeScene& sc, const DevTexture t, f2 uv, float4 duv) {
    const TexLevel l0v = sc.tex_levels[t.first_level];
    const float W = (float)l0v.w, H = (float)l0v.h;
    float ax = duv.x * W, ay = duv.y * H, bx = duv.z * W, by = duv.w * H;
    float lx = sqrtf(ax * ax + ay * ay), ly = sqrtf(bx * bx + by * by);
    float rho = (lx < ly) ? ly : lx;
    const int last = t.n_levels - 1;
    float lod = 0.0f;
    if (rho > 1.0f) lod = dm_log2(rho);
    if (!(lod < (float)last)) lod = (float)last;
    int l0 = (int)floorf(lod);
    int l1 = min(l0 + 1, last);
    float f = lod - (float)l0;
    return tex_bilinear(sc, sc.tex_levels[t.first_level + l0], uv) * (1.0f - f) +
           tex_bilinear(sc, sc.tex_levels[t.first_level + l1], uv) * f;
}
RT_DEV f2 fract2(f2 v) { f2 r; r.x = v.x - floorf(v.x); r.y = v.y - floorf(v.y); return r; }

// The 36-float attribute block of a triangle (144 bytes, 16-byte aligned) as nine 16-byte loads.
RT_DEV void load_tri_attr(const float* tri_attr, int32_t tri, float (&a)[36]) {
    const float4* p = reinterpret_cast<const float4*>(tri_attr) + (size_t)tri * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float4 q = p[k];
        a[4 * k] = q.x; a[4 * k + 1] = q.y; a[4 * k + 2] = q.z; a[4 * k + 3] = q.w;
    }
}

// UV of a hit: the uv part of BVH::postIntersect (src/scene/BVH.cpp:172-187)
RT_DEV f2 hit_uv(const float* tri_attr, int32_t id1, float hu, float hv) {
    const float* at = tri_attr + (size_t)(id1 / 3) * 36;
    float bc[3] = {1.0f - hu - hv, hu, hv};
    f2 uv; uv.x = 0.f; uv.y = 0.f;
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        uv.x = uv.x + at[v * 11 + 3] * bc[v];
        uv.y = uv.y + at[v * 11 + 4] * bc[v];
    }
    return uv;
}

// ----------------------------------------------------------------- lights
// src/lights/PointLight.cpp:14-18, src/lights/EnvLight.cpp:28-43
// tmax: where the shadow ray towards the sample stops (+inf for the
// reference's lights, whose shadow rays are unbounded).  Area lights (one
// emissive triangle; the reference only names the type, LightSource.hpp:15)
// follow the published DiffuseAreaLight-over-Triangle model: uniform area
// sampling, pdf dist^2 / (|cos| area), one-sided emission.
RT_DEV f3 light_sample(const DevLight& l, f3 p, f2 u, f3& in, float& pdf, float& tmax, const float4* lds_marg) {
    tmax = __builtin_inff();
    if (l.type == RT_LIGHT_POINT_K) {
        f3 lp = mk3(l.px, l.py, l.pz);
        in = normalize3(lp - p);
        pdf = 1.f;
        f3 d = lp - p;
        return mk3(l.ix, l.iy, l.iz) / dot3(d, d);
    }
    if (l.type == RT_LIGHT_AREA_K) {
        float su0 = sqrtf(u.x);
        float b0 = 1.f - su0, b1 = u.y * su0;
        f3 q = mk3(l.p0[0], l.p0[1], l.p0[2]) * b0 + mk3(l.p1[0], l.p1[1], l.p1[2]) * b1 + mk3(l.p2[0], l.p2[1], l.p2[2]) * (1.f - b0 - b1);
        f3 w = q - p;
        float d2 = dot3(w, w);
        pdf = 0.0f;
        in = mk3(0.f, 0.f, 0.f);
        if (d2 == 0) return mk3(0.f, 0.f, 0.f);
        float d = sqrtf(d2);
        in = w / d;
        float cosL = dot3(mk3(l.n[0], l.n[1], l.n[2]), -in);
        pdf = d2 / (fabsf(cosL) * l.area);
        if (!isfinite(pdf)) { pdf = 0.0f; return mk3(0.f, 0.f, 0.f); }
        tmax = d * (1.0f - 1e-4f);
        return cosL > 0 ? mk3(l.ix, l.iy, l.iz) : mk3(0.f, 0.f, 0.f);
    }
    // Distribution2D::sampleContinuous, src/samplers/Distributions.cpp:60-68
    float pdf0, pdf1;
    int v, dummy;
    // the marginal's records sit in LDS when the workgroup staged them (lds_marg), else in global memory
    float d1 = lds_marg ? dist1d_sample((LdsPack)lds_marg, l.marg_int, l.W, u.y, pdf1, v)
                        : dist1d_sample((GlobalPack)(l.marg_pack), l.marg_int, l.W, u.y, pdf1, v);
    float d0 = dist1d_sample((GlobalPack)(l.cond_pack) + (size_t)v * (l.H + 1), ((GlobalFloats)l.cond_int)[v], l.H, u.x, pdf0, dummy);
    float mapPdf = pdf0 * pdf1;
    pdf = 0.0f;
    in = mk3(0.f, 0.f, 0.f);
    if (mapPdf == 0) return mk3(0.f, 0.f, 0.f);
    float theta = d0 * kPi, phi = d1 * 2.f * kPi;
    float cosT = dm_cos(theta), sinT = dm_sin(theta);
    float sinP = dm_sin(phi), cosP = dm_cos(phi);
    in = mk3(sinT * cosP, sinT * sinP, cosT);
    pdf = mapPdf / (2 * kPi * kPi * sinT);
    if (!isfinite(pdf)) pdf = 0.0f;
    f2 uv; uv.x = d0; uv.y = d1;
    return filter_linear(l.img, l.W, l.H, uv) * mk3(l.ix, l.iy, l.iz);
}
// src/lights/PointLight.cpp:20-22, src/lights/EnvLight.cpp:45-52 with
// Distribution2D::pdf, src/samplers/Distributions.cpp:70-76
RT_DEV float light_pdf(const DevLight& l, f3 p, f3 inW) {
    if (l.type == RT_LIGHT_POINT_K) return 1.0f;
    if (l.type == RT_LIGHT_AREA_K) {
        float t, bu, bv;
        if (!tri_test(p, inW, mk3(l.p0[0], l.p0[1], l.p0[2]), mk3(l.p1[0], l.p1[1], l.p1[2]), mk3(l.p2[0], l.p2[1], l.p2[2]), t, bu, bv)) return 0;
        if (!(t > 0.0001f)) return 0;
        float pd = (t * t) / (fabsf(dot3(mk3(l.n[0], l.n[1], l.n[2]), -inW)) * l.area);
        return isfinite(pd) ? pd : 0;
    }
    float theta = spherical_theta(inW), phi = spherical_phi(inW);
    float sinT = dm_sin(theta);
    if (sinT == 0) return 0.f;
    float p0 = phi * kInv2Pi, p1 = theta * kInvPi;
    int iu = (int)clampf((float)(int)(p0 * l.H), 0.f, (float)(l.H - 1));
    int iv = (int)clampf((float)(int)(p1 * l.W), 0.f, (float)(l.W - 1));
    float dp = ((GlobalPack)l.cond_pack)[(size_t)iv * (l.H + 1) + iu].z / l.marg_int;
    return dp / (2 * kPi * kPi * sinT);
}
// src/lights/LightSource.hpp:28-30, src/lights/EnvLight.cpp:60-65
RT_DEV f3 light_le(const DevLight& l, f3 dir) {
    if (l.type != RT_LIGHT_ENV_K) return mk3(0.f, 0.f, 0.f);
    f3 w = normalize3(dir);
    f2 st; st.x = spherical_phi(w) * kInv2Pi; st.y = spherical_theta(w) * kInvPi;
    return filter_linear(l.img, l.W, l.H, st);
}

// Random-number slot layout of one path vertex at depth d with L lights.
RT_DEV uint32_t dim_light(int d, int L, int l, int k) { return (uint32_t)(d * (4 * L + 2) + 4 * l + k); }
RT_DEV uint32_t dim_bounce(int d, int L, int k) { return (uint32_t)(d * (4 * L + 2) + 4 * L + k); }

// ---------------------------------------------------------------- kernels
// Camera rays, src/cpu/PathTracer.cpp:43-52, one per film slot of this rank (they do not depend on the pass).
__global__ __launch_bounds__(RT_BLOCK) void k_generate(PathBuffers pb, CameraK cam, uint32_t count) {
    uint32_t s = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (s >= count) return;
    uint32_t pix = pb.slot_pixel[s];
    if (pix == 0xFFFFFFFFu) return;
    const size_t n = pb.n1;
    int x = (int)(pix % (uint32_t)cam.W), y = (int)(pix / (uint32_t)cam.W);
    float ni = ((float)x / (float)cam.W) - 0.5f;
    float nj = ((float)y / (float)cam.H) - 0.5f;
    nj = -nj;
    nj *= ((float)cam.H / (float)cam.W);
    f3 right = mk3(cam.right[0], cam.right[1], cam.right[2]);
    f3 up = mk3(cam.up[0], cam.up[1], cam.up[2]);
    f3 fwd = mk3(cam.forward[0], cam.forward[1], cam.forward[2]);
    f3 ip = ni * right + nj * up + fwd;
    f3 d = normalize3(ip);
    pb.prim_rays[s] = cam.pos[0]; pb.prim_rays[n + s] = cam.pos[1]; pb.prim_rays[2 * n + s] = cam.pos[2];
    pb.prim_rays[3 * n + s] = d.x; pb.prim_rays[4 * n + s] = d.y; pb.prim_rays[5 * n + s] = d.z;
}

// UV differences of the primary hits, PASS 1/PASS 2 of
// src/cpu/PathTracer.cpp:31-90: the film is walked in 2x2 pixel quads, tile by
// tile; a pixel's dUVx is the difference of the two hits of its quad row,
// dUVy of its quad column, scaled by the film size.  The reference keeps the
// four hit vertices in a block that lives for the whole tile and is only
// overwritten by pixels that hit, so a neighbour that missed (or lies outside
// the film) contributes the UV of the last earlier quad of the tile (in scan
// order) whose pixel at that quad position did hit -- zero if there is none.
// One workgroup per tile: every thread owns a contiguous run of quads,
// finds the last hit per quad position in its run, the runs are chained
// through LDS, then each thread replays its run with the carried UVs.
__global__ __launch_bounds__(RT_BLOCK) void k_quad_duv(PathBuffers pb, const float* __restrict__ tri_attr, float4* __restrict__ duv, int tile,
                                                       float film_w, float film_h) {
    __shared__ int32_t s_last[4][RT_BLOCK];
    const uint32_t tid = threadIdx.x;
    const int qpr = tile >> 1;                       // quads per tile row
    const int nq = qpr * qpr;
    const int per = (nq + RT_BLOCK - 1) / RT_BLOCK;
    const int q_lo = min((int)tid * per, nq), q_hi = min(q_lo + per, nq);
    const int bpr = tile >> 3;
    const uint32_t tile_base = blockIdx.x * (uint32_t)(tile * tile);
    auto slot_of = [&](int q, int id) -> uint32_t {
        int x = 2 * (q % qpr) + (id & 1), y = 2 * (q / qpr) + (id >> 1);
        return tile_base + (uint32_t)((((y >> 3) * bpr + (x >> 3)) << 6) + ((y & 7) << 3) + (x & 7));
    };
    auto valid_of = [&](uint32_t s) -> bool { return pb.slot_pixel[s] != 0xFFFFFFFFu && pb.prim_id1[s] >= 0; };
    int32_t last[4] = {-1, -1, -1, -1};
    for (int q = q_lo; q < q_hi; ++q)
#pragma unroll
        for (int id = 0; id < 4; ++id)
            if (valid_of(slot_of(q, id))) last[id] = q;
#pragma unroll
    for (int id = 0; id < 4; ++id) s_last[id][tid] = last[id];
    __syncthreads();
    // inclusive max-scan over the threads' runs (quad indices grow with tid)
    for (uint32_t off = 1; off < RT_BLOCK; off <<= 1) {
        int32_t v[4];
#pragma unroll
        for (int id = 0; id < 4; ++id) v[id] = (tid >= off) ? max(s_last[id][tid], s_last[id][tid - off]) : s_last[id][tid];
        __syncthreads();
#pragma unroll
        for (int id = 0; id < 4; ++id) s_last[id][tid] = v[id];
        __syncthreads();
    }
    f2 cur[4];
#pragma unroll
    for (int id = 0; id < 4; ++id) {
        cur[id].x = 0.f; cur[id].y = 0.f;
        int32_t ql = tid ? s_last[id][tid - 1] : -1;
        if (ql >= 0) {
            uint32_t s = slot_of(ql, id);
            cur[id] = hit_uv(tri_attr, pb.prim_id1[s], pb.prim_u[s], pb.prim_v[s]);
        }
    }
    for (int q = q_lo; q < q_hi; ++q) {
        bool ok[4];
#pragma unroll
        for (int id = 0; id < 4; ++id) {
            uint32_t s = slot_of(q, id);
            ok[id] = valid_of(s);
            if (ok[id]) cur[id] = hit_uv(tri_attr, pb.prim_id1[s], pb.prim_u[s], pb.prim_v[s]);
        }
#pragma unroll
        for (int id = 0; id < 4; ++id) {
            if (!ok[id]) continue;
            const int yB = id >> 1, xB = id & 1;
            float dxx = cur[yB * 2 + 1].x - cur[yB * 2].x, dxy = cur[yB * 2 + 1].y - cur[yB * 2].y;
            float dyx = cur[2 + xB].x - cur[xB].x, dyy = cur[2 + xB].y - cur[xB].y;
            duv[slot_of(q, id)] = make_float4(dxx * film_w, dxy * film_h, dyx * film_w, dyy * film_h);
        }
    }
}

// Shadow rays of one bounce.  Nothing is compacted: everything is indexed by
// e = light * n_slots + slot, and the work items of k_shadow are the pairs
// (light, entry of the bounce's path queue).  Per entry: the ray direction
// (the origin is the path's shaded vertex), its far bound, the two
// contributions the reference adds one after the other once the ray is found
// unoccluded (src/cpu/PathTracer.cpp:107 or :109, then :123), and a state
// byte: 0 no ray, 1 ray pending, 2 ray found unoccluded.
struct ShadowQueue {
    float* dir;        // 3 planes of cap
    float* tmax;       // [cap] far bound of each ray, or null when the scene has no area light
    float* contrib;    // 6 planes of cap
    uint8_t* state;    // [cap]
    uint32_t cap;      // n_lights * n_slots
    // The lights whose entries the traversal launch of this bounce walks: all of them, or, at bounce 0 of a pass
    // that reads the per-camera cache, only those whose shadow ray is not cached (k_shade has resolved the others).
    const int32_t* trace_lights;
    int32_t n_trace;
};

// One bounce of src/cpu/PathTracer.cpp:56-65 (d = 0 miss) and :95-144 for
// every path in the queue: BVH::postIntersect (src/scene/BVH.cpp:172-187),
// Material::computeBSDF (src/scene/Material.cpp:214-225), per-light sampling
// with the contributions the reference would add once the shadow ray is
// found unoccluded, then the BSDF bounce.
//
// At depth 0 the ray and its hit are the cached primary ones of the path's film slot (pb.prim_*), beta is one and
// the path index is the slot.  `prime` != 0 is the first pass after a camera change: a primary ray that missed
// writes the film (it does so once: nothing about it changes until the camera moves) and the shadow rays towards
// point lights are traced, their outcome then being copied to pb.prim_vis by k_cache_visibility; with prime == 0
// that outcome is taken from the cache and no such ray is queued.
//
// The three parameter blocks (about 85 dwords) come through one pointer to device memory instead of by value: as
// kernel arguments they were all fetched into scalar registers up front and spilled from there; through the pointer
// each field is a scalar load where it is used.
struct ShadeArgs {
    PathBuffers pb;
    ShadeScene sc;
    ShadowQueue sq;
};
template <bool GLOSSY>
__global__ __launch_bounds__(RT_BLOCK) void k_shade(const ShadeArgs* __restrict__ args, int depth, int max_depth, int prime, uint64_t seed_mixed,
                                                    uint32_t first_pass, const uint32_t* __restrict__ queue,
                                                    const uint32_t* __restrict__ d_count, uint32_t* __restrict__ next_count,
                                                    unsigned long long* __restrict__ shadow_ray_counter) {
    const PathBuffers& pb = args->pb;
    const ShadeScene& sc = args->sc;
    const ShadowQueue& sq = args->sq;
    __shared__ uint32_t s_cnt[RT_BLOCK / 64 + 1];
    __shared__ float4 s_marg[RT_MARG_LDS];
    const uint32_t count = *d_count;
    const uint32_t gsize = gridDim.x * RT_BLOCK;
    const size_t n = pb.n;
    const int L = sc.n_lights;
    // The marginal distribution of the environment light (every path searches it once per bounce) is staged in LDS
    // by the workgroups that have paths to shade.
    if (sc.env_staged >= 0 && blockIdx.x * RT_BLOCK < count) {
        const DevLight& el = sc.lights[sc.env_staged];
        for (int k = threadIdx.x; k <= el.W; k += RT_BLOCK) s_marg[k] = el.marg_pack[k];
        __syncthreads();
    }
    uint32_t n_emit = 0;   // shadow rays this wave queued (wave-uniform)
    // every thread of a workgroup runs the same number of iterations (barriers inside)
    for (uint32_t bbase = blockIdx.x * RT_BLOCK; bbase < count; bbase += gsize) {
        uint32_t i = bbase + threadIdx.x;
        bool valid = i < count;
        // path index: the entries of `queue` at bounce 0 (slots inside the film), the entry number afterwards
        const uint32_t pi = valid ? (queue ? queue[i] : i) : 0u;
        const uint32_t s = valid ? (depth == 0 ? pi : pb.slot[pi]) : 0u;
        const uint32_t fs = s % pb.n1;   // film slot
        bool hit = false;
        f3 org = mk3(0.f, 0.f, 0.f), dir = mk3(0.f, 0.f, 1.f);
        int32_t id0 = -1, id1 = -1;
        float ht = 0.f, hu = 0.f, hv = 0.f;
        if (valid) {
            if (depth == 0) {
                const size_t n1 = pb.n1;
                org = mk3(pb.prim_rays[fs], pb.prim_rays[n1 + fs], pb.prim_rays[2 * n1 + fs]);
                dir = mk3(pb.prim_rays[3 * n1 + fs], pb.prim_rays[4 * n1 + fs], pb.prim_rays[5 * n1 + fs]);
                id1 = pb.prim_id1[fs];
                id0 = pb.prim_id0[fs];
                ht = pb.prim_t[fs]; hu = pb.prim_u[fs]; hv = pb.prim_v[fs];
            } else {
                org = mk3(pb.rays[pi], pb.rays[n + pi], pb.rays[2 * n + pi]);
                dir = mk3(pb.rays[3 * n + pi], pb.rays[4 * n + pi], pb.rays[5 * n + pi]);
                id1 = pb.hit_id1[pi];
                id0 = pb.hit_id0[pi];
                ht = pb.hit_t[pi]; hu = pb.hit_u[pi]; hv = pb.hit_v[pi];
            }
            hit = id1 >= 0;
            if (depth == 0) {
                // per-pass state of the slot: the radiance sum starts at zero, status says whether the film
                // takes the running mean (src/cpu/PathTracer.cpp:148-157)
                pb.status[s] = hit ? 1u : 0u;
                if (hit) { pb.L[s] = 0.f; pb.L[n + s] = 0.f; pb.L[2 * n + s] = 0.f; }
                if (!hit && prime && s < pb.n1) {
                    // src/cpu/PathTracer.cpp:61-65: the film takes the summed le, no accumulation
                    f3 Lm = mk3(0.f, 0.f, 0.f);
                    for (int l = 0; l < L; ++l) Lm = Lm + light_le(sc.lights[l], dir);
                    pb.film[fs] = make_float4(Lm.x, Lm.y, Lm.z, 1.0f);
                }
            }
        }
        BsdfK bsdf;
        bsdf.t = mk3(1.f, 0.f, 0.f); bsdf.b = mk3(0.f, 1.f, 0.f); bsdf.n = mk3(0.f, 0.f, 1.f);
        bsdf.lobe = make_diffuse(mk3(0.f, 0.f, 0.f), 0.f);
        bsdf.parts = 1;
        bsdf.glossy.d.kind = RT_GLOSSY_NONE_K; bsdf.glossy.d.ax = 1.f; bsdf.glossy.d.ay = 1.f;
        bsdf.glossy.k = mk3(0.f, 0.f, 0.f);
        f3 vpos = mk3(0.f, 0.f, 0.f), vn = mk3(0.f, 0.f, 1.f), out = mk3(0.f, 0.f, 1.f);
        f3 beta = mk3(1.f, 1.f, 1.f);
        uint32_t pixel = 0;
        const uint32_t pass = first_pass + s / pb.n1;
        if (hit) {
            pixel = pb.slot_pixel[s];
            const float t = ht;
            float bc[3] = {1.0f - hu - hv, hu, hv};
            vpos = org + dir * t;
            float at[36];
            load_tri_attr(sc.tri_attr, id1 / 3, at);
            f3 nrm = mk3(0.f, 0.f, 0.f), tan = mk3(0.f, 0.f, 0.f), bin = mk3(0.f, 0.f, 0.f);
            f2 uv; uv.x = 0.f; uv.y = 0.f;
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const float* a = at + v * 11;
                nrm = nrm + mk3(a[0], a[1], a[2]) * bc[v];
                uv.x = uv.x + a[3] * bc[v];
                uv.y = uv.y + a[4] * bc[v];
                tan = tan + mk3(a[5], a[6], a[7]) * bc[v];
                bin = bin + mk3(a[8], a[9], a[10]) * bc[v];
            }
            nrm = normalize3(nrm);
            if (sc.have_frame) {
                bin = normalize3(bin);
                tan = normalize3(tan);
            } else {
                // no tangents supplied: orthonormal frame from the normal
                if (fabsf(nrm.x) > fabsf(nrm.y)) {
                    float il = 1.0f / sqrtf(nrm.x * nrm.x + nrm.z * nrm.z);
                    tan = mk3(-nrm.z * il, 0.0f, nrm.x * il);
                } else {
                    float il = 1.0f / sqrtf(nrm.y * nrm.y + nrm.z * nrm.z);
                    tan = mk3(0.0f, nrm.z * il, -nrm.y * il);
                }
                bin = cross3(nrm, tan);
            }
            vn = nrm;
            float4 m = sc.materials[id0];
            // Material::computeBSDF: trilinear with the quad's UV differences at
            // the primary hit (src/scene/Material.cpp:195-212), bilinear level 0
            // afterwards (:214-225)
            f3 col = mk3(255.0f, 255.0f, 255.0f);
            if (sc.textures) {
                int32_t ti = sc.material_texture[id0];
                if (ti >= 0) {
                    const DevTexture tx = sc.textures[ti];
                    f2 ruv = fract2(uv);
                    col = (depth == 0) ? tex_trilinear(sc, tx, ruv, sc.duv[fs]) : tex_bilinear(sc, sc.tex_levels[tx.first_level], ruv);
                }
            }
            col = col / 255.0f;
            col = col * mk3(m.x, m.y, m.z);
            bsdf.t = tan; bsdf.b = bin; bsdf.n = nrm;
            bsdf.lobe = make_diffuse(col, m.w);
            if (GLOSSY) {
                // the lobe of src/scene/Material.cpp:205-208: alpha = (roughness, roughness + 0.1), k = diffColor
                int g = sc.material_glossy[id0];
                if (g != RT_GLOSSY_NONE_K) {
                    bsdf.parts = 2;
                    bsdf.glossy.d.kind = g;
                    bsdf.glossy.d.ax = m.w;
                    bsdf.glossy.d.ay = m.w + 0.1f;
                    bsdf.glossy.k = col;
                }
            }
            out = -dir;
            pb.vpos[pi] = vpos.x; pb.vpos[n + pi] = vpos.y; pb.vpos[2 * n + pi] = vpos.z;
            if (depth != 0) beta = mk3(pb.beta[pi], pb.beta[n + pi], pb.beta[2 * n + pi]);
        }

        // ---- light loop, src/cpu/PathTracer.cpp:96-125
        for (int l = 0; l < L; ++l) {
            bool emit = false;
            f3 in = mk3(0.f, 0.f, 0.f), C = mk3(0.f, 0.f, 0.f), C2 = mk3(0.f, 0.f, 0.f);
            float tmax = __builtin_inff();
            if (hit) {
                const DevLight& light = sc.lights[l];
                float lightPdf;
                f2 u;
                u.x = rng_u01(seed_mixed, pixel, pass, dim_light(depth, L, l, 0));
                u.y = rng_u01(seed_mixed, pixel, pass, dim_light(depth, L, l, 1));
                f3 li = light_sample(light, vpos, u, in, lightPdf, tmax, l == sc.env_staged ? s_marg : nullptr);
                if (lightPdf != 0.0f) {
                    emit = true;
                    f3 bsdfColor = bsdf_f<GLOSSY>(bsdf, out, in);
                    float bsdfPdf = bsdf_pdf<GLOSSY>(bsdf, out, in);
                    if (light.type == RT_LIGHT_POINT_K) {
                        C = beta * bsdfColor * li / lightPdf;
                    } else {
                        C = beta * bsdfColor * li * power_heuristic(lightPdf, bsdfPdf) / lightPdf;
                        f3 inBSDF;
                        f2 u2;
                        u2.x = rng_u01(seed_mixed, pixel, pass, dim_light(depth, L, l, 2));
                        u2.y = rng_u01(seed_mixed, pixel, pass, dim_light(depth, L, l, 3));
                        f3 f = bsdf_sample<GLOSSY>(bsdf, out, u2, inBSDF, bsdfPdf);
                        if (bsdfPdf != 0.0f) {
                            f = f * fabsf(dot3(in, vn));
                            float lp = light_pdf(light, vpos, in);
                            if (lp != 0.0f) {
                                float weight = power_heuristic(bsdfPdf, lp);
                                f3 li2 = light_le(light, dir);
                                if (!is_zero3(li2)) C2 = f * li2 * weight / bsdfPdf;
                            }
                        }
                    }
                }
            }
            // a shadow ray that could only add exact zeros changes nothing whether
            // or not it is occluded (L + 0 == L), so it is not traced
            emit = emit && !(is_zero3(C) && is_zero3(C2));
            uint8_t st = emit ? 1 : 0;
            bool traced = emit;
            if (depth == 0 && !prime && hit && sc.lights[l].type == RT_LIGHT_POINT_K) {
                // the ray from the primary vertex to a point light was resolved by the first pass
                st = (pb.prim_vis[(size_t)l * pb.n1 + fs] == 2) ? 2 : 0;
                emit = st == 2;
                traced = false;
            }
            n_emit += (uint32_t)__popcll(__ballot(traced));
            if (valid) sq.state[(size_t)l * n + pi] = st;
            if (emit) {
                const size_t cap = sq.cap, e = (size_t)l * n + pi;
                sq.dir[e] = in.x; sq.dir[cap + e] = in.y; sq.dir[2 * cap + e] = in.z;
                if (sq.tmax) sq.tmax[e] = tmax;
                sq.contrib[e] = C.x; sq.contrib[cap + e] = C.y; sq.contrib[2 * cap + e] = C.z;
                sq.contrib[3 * cap + e] = C2.x; sq.contrib[4 * cap + e] = C2.y; sq.contrib[5 * cap + e] = C2.z;
            }
        }

        // ---- BSDF bounce, src/cpu/PathTracer.cpp:127-144.  The closest-hit
        // ray of the last iteration feeds nothing and is not generated.
        bool alive = false;
        f3 nextDir = mk3(0.f, 0.f, 0.f);
        if (hit && depth + 1 < max_depth) {
            float nextPDF = 0.0f;
            f2 u3;
            u3.x = rng_u01(seed_mixed, pixel, pass, dim_bounce(depth, L, 0));
            u3.y = rng_u01(seed_mixed, pixel, pass, dim_bounce(depth, L, 1));
            f3 bsdfColor = bsdf_sample<GLOSSY>(bsdf, out, u3, nextDir, nextPDF);
            if (!(is_zero3(bsdfColor) || nextPDF == 0.0f)) {
                alive = true;
                beta = beta * (bsdfColor * fabsf(dot3(nextDir, vn)) / nextPDF);
            }
        }
        // the surviving path moves to the next free index of the next bounce's arrays
        uint32_t k = block_append(alive, next_count, s_cnt);
        if (alive) {
            pb.slot_next[k] = s;
            pb.beta_next[k] = beta.x; pb.beta_next[n + k] = beta.y; pb.beta_next[2 * n + k] = beta.z;
            pb.rays_next[k] = vpos.x; pb.rays_next[n + k] = vpos.y; pb.rays_next[2 * n + k] = vpos.z;
            pb.rays_next[3 * n + k] = nextDir.x; pb.rays_next[4 * n + k] = nextDir.y; pb.rays_next[5 * n + k] = nextDir.z;
        }
    }
    // ray statistics: one atomic per workgroup
    if ((threadIdx.x & 63u) == 0) s_cnt[threadIdx.x >> 6] = n_emit;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (uint32_t k = 0; k < RT_BLOCK / 64; ++k) tot += s_cnt[k];
        if (tot) atomicAdd(shadow_ray_counter, (unsigned long long)tot);
    }
}

// `probe` of the launches that walk shadow entries: wraps a callable that says whether work item j carries a ray.
template <typename F>
struct EntryProbe {
    static constexpr bool kFilters = true;
    F f;
    RT_DEV bool operator()(uint32_t j) const { return f(j); }
};
template <typename F>
RT_DEV EntryProbe<F> make_probe(F f) { return EntryProbe<F>{f}; }

// Traces the bounce's shadow rays, every light in one launch:
// BVH::occluded from the shaded vertex (src/cpu/PathTracer.cpp:101-102).  An
// unoccluded ray only marks its (light, slot) state; k_accumulate then adds
// the contributions in the reference's order.
// WIDE: the launch walks the 4-wide collapse of the tree (trace_wide.h); any-hit results do not depend on it.
template <bool EXACT, bool BOUNDED, bool WIDE = false>
__global__ __launch_bounds__(RT_TBLOCK RT_LB_WAVES) void k_shadow(TraceScene ts, PathBuffers pb, ShadowQueue q, int n_lights,
                                                     const uint32_t* __restrict__ queue, const uint32_t* __restrict__ d_count,
                                                     TraceWork* __restrict__ work, uint32_t* __restrict__ deferred,
                                                     unsigned long long* __restrict__ ray_counter) {
    static_assert(!(WIDE && EXACT), "the exact follow-up launch walks the binary tree");
    __shared__ int32_t s_stack[(WIDE ? RT_WSTACK : RT_STACK_LDS) * RT_TBLOCK];
    __shared__ __attribute__((aligned(16))) char s_top[WIDE ? RT_WTOP_BYTES : RT_TOP_BYTES];
    const auto top = stage_top<WIDE>(ts, s_top);
    const uint32_t paths = *d_count;
    // work item j = (light j / paths, path-queue entry j % paths)
    const uint32_t count = EXACT ? work->n_deferred : paths * (uint32_t)q.n_trace;
    const size_t n = pb.n, cap = q.cap;
    if (EXACT && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(ray_counter + 2, (unsigned long long)count);
    // work item j -> (light, path index); `queue` lists the path indices at bounce 0 and is null afterwards
    auto entry = [&](uint32_t j, uint32_t& s) -> size_t {
        const uint32_t li = j / paths;
        const uint32_t e = j - li * paths;
        s = queue ? queue[e] : e;
        return (size_t)q.trace_lights[li] * n + s;
    };
    auto mark = [&](uint32_t j, const HitRec& h) {
        uint32_t s;
        if (h.id1 < 0) q.state[entry(j, s)] = 2;
    };
    if constexpr (BOUNDED) {
        trace_rays<true, EXACT, true, false>(
            ts, count, work, deferred, s_stack, top,
            [&](uint32_t j, f3& org, f3& dir, float& tfar) -> bool {
                uint32_t s;
                const size_t e = entry(j, s);
                if (q.state[e] != 1) return false;
                org = mk3(pb.vpos[s], pb.vpos[n + s], pb.vpos[2 * n + s]);
                dir = mk3(q.dir[e], q.dir[cap + e], q.dir[2 * cap + e]);
                tfar = q.tmax[e];
                return true;
            },
            mark, make_probe([&](uint32_t j) -> bool { uint32_t s; return q.state[entry(j, s)] == 1; }));
    } else {
        trace_rays<true, EXACT, false, false>(
            ts, count, work, deferred, s_stack, top,
            [&](uint32_t j, f3& org, f3& dir) -> bool {
                uint32_t s;
                const size_t e = entry(j, s);
                if (q.state[e] != 1) return false;
                org = mk3(pb.vpos[s], pb.vpos[n + s], pb.vpos[2 * n + s]);
                dir = mk3(q.dir[e], q.dir[cap + e], q.dir[2 * cap + e]);
                return true;
            },
            mark, make_probe([&](uint32_t j) -> bool { uint32_t s; return q.state[entry(j, s)] == 1; }));
    }
}

// One traversal launch per bounce: the closest-hit rays of the next bounce
// (the first *d_next_count entries of the next-bounce arrays, written by
// k_shade) and the shadow rays of this bounce (work items (light, path of
// this bounce)) are both ready once k_shade
// has run, so they share one persistent launch: item j < n_next is a
// closest-hit ray, the others are shadow rays.
// WIDE: on the 4-wide tree; the closest-hit rays are then tie-tolerant (trace_wide.h).
template <bool EXACT, bool WIDE = false>
__global__ __launch_bounds__(RT_TBLOCK RT_LB_WAVES) void k_bounce(TraceScene ts, PathBuffers pb, ShadowQueue q, int n_lights,
                                                                 const uint32_t* __restrict__ queue, const uint32_t* __restrict__ d_count,
                                                                 const uint32_t* __restrict__ d_next_count, TraceWork* __restrict__ work,
                                                                 uint32_t* __restrict__ deferred,
                                                                 unsigned long long* __restrict__ ray_counter) {
    static_assert(!(WIDE && EXACT), "the exact follow-up launch walks the binary tree");
    __shared__ int32_t s_stack[(WIDE ? RT_WSTACK : RT_STACK_LDS) * RT_TBLOCK];
    __shared__ __attribute__((aligned(16))) char s_top[WIDE ? RT_WTOP_BYTES : RT_TOP_BYTES];
    const auto top = stage_top<WIDE>(ts, s_top);
    const uint32_t paths = *d_count, n_next = *d_next_count;
    const uint32_t count = EXACT ? work->n_deferred : n_next + paths * (uint32_t)q.n_trace;
    const size_t n = pb.n, cap = q.cap;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (!EXACT) atomicAdd(ray_counter, (unsigned long long)n_next);   // closest-hit rays; k_shade counted the shadow rays
        else atomicAdd(ray_counter + 2, (unsigned long long)count);
    }
    auto entry = [&](uint32_t j, uint32_t& s) -> size_t {
        const uint32_t k = j - n_next;
        const uint32_t li = k / paths;
        const uint32_t e = k - li * paths;
        s = queue ? queue[e] : e;
        return (size_t)q.trace_lights[li] * n + s;
    };
    trace_rays<false, EXACT, false, true>(
        ts, count, work, deferred, s_stack, top,
        [&](uint32_t j, f3& org, f3& dir, float& tfar, bool& any) -> bool {
            if (j < n_next) {   // path j of the next bounce
                any = false;
                org = mk3(pb.rays_next[j], pb.rays_next[n + j], pb.rays_next[2 * n + j]);
                dir = mk3(pb.rays_next[3 * n + j], pb.rays_next[4 * n + j], pb.rays_next[5 * n + j]);
                return true;
            }
            uint32_t s;
            const size_t e = entry(j, s);
            any = true;
            if (q.state[e] != 1) return false;
            org = mk3(pb.vpos[s], pb.vpos[n + s], pb.vpos[2 * n + s]);
            dir = mk3(q.dir[e], q.dir[cap + e], q.dir[2 * cap + e]);
            tfar = q.tmax ? q.tmax[e] : __builtin_inff();
            return true;
        },
        [&](uint32_t j, const HitRec& h) {
            if (j < n_next) {
                pb.hit_t[j] = h.t; pb.hit_id0[j] = h.id0; pb.hit_id1[j] = h.id1; pb.hit_u[j] = h.u; pb.hit_v[j] = h.v;
            } else {
                uint32_t s;
                if (h.id1 < 0) q.state[entry(j, s)] = 2;
            }
        },
        make_probe([&](uint32_t j) -> bool {
            if (j < n_next) return true;
            uint32_t s;
            return q.state[entry(j, s)] == 1;
        }));
}

// After the traversal launch of bounce 0 in the first pass after a camera change: keeps, per film slot and point
// light, what became of the shadow ray from the primary vertex (entries of the batch's first pass).
__global__ __launch_bounds__(RT_BLOCK) void k_cache_visibility(PathBuffers pb, ShadowQueue q, const DevLight* __restrict__ lights, int n_lights) {
    const uint32_t s = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (s >= pb.n1) return;
    for (int l = 0; l < n_lights; ++l)
        if (lights[l].type == RT_LIGHT_POINT_K) pb.prim_vis[(size_t)l * pb.n1 + s] = q.state[(size_t)l * pb.n + s];
}

// The adds of src/cpu/PathTracer.cpp:107/:109 and :123 for every path of the
// bounce, light by light in the reference's loop order.
__global__ __launch_bounds__(RT_BLOCK) void k_accumulate(PathBuffers pb, ShadowQueue q, int n_lights, const uint32_t* __restrict__ queue,
                                                         const uint32_t* __restrict__ d_count) {
    const uint32_t count = *d_count;
    const size_t n = pb.n, cap = q.cap;
    for (uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x; i < count; i += gridDim.x * RT_BLOCK) {
        const uint32_t pi = queue ? queue[i] : i;
        const uint32_t s = queue ? pi : pb.slot[pi];   // bounce 0 (the only one with a queue): path index = slot
        f3 Lp = mk3(pb.L[s], pb.L[n + s], pb.L[2 * n + s]);
        bool any = false;
        for (int l = 0; l < n_lights; ++l) {
            const size_t e = (size_t)l * n + pi;
            if (q.state[e] != 2) continue;
            any = true;
            Lp = Lp + mk3(q.contrib[e], q.contrib[cap + e], q.contrib[2 * cap + e]);
            f3 C2 = mk3(q.contrib[3 * cap + e], q.contrib[4 * cap + e], q.contrib[5 * cap + e]);
            if (!is_zero3(C2)) Lp = Lp + C2;
        }
        if (any) { pb.L[s] = Lp.x; pb.L[n + s] = Lp.y; pb.L[2 * n + s] = Lp.z; }
    }
}

// Film update, src/cpu/PathTracer.cpp:148-157: running mean over passes for
// pixels whose primary ray hit.  The passes of a batch are folded in one
// after the other, in pass order, so the film is the one n_passes separate
// updates would leave.  In place: the reference's ping-pong pair only ever
// reads the previous value of the same pixel.
__global__ __launch_bounds__(RT_BLOCK) void k_film(PathBuffers pb, float first_frames, uint32_t n_passes) {
    uint32_t s1 = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (s1 >= pb.n1) return;
    const size_t n = pb.n;
    float4 p = pb.film[s1];
    bool any = false;
    float frames = first_frames;
    for (uint32_t b = 0; b < n_passes; ++b, frames += 1.0f) {
        const size_t s = (size_t)b * pb.n1 + s1;
        if (pb.status[s] != 1) continue;
        any = true;
        f3 c = mk3(pb.L[s], pb.L[n + s], pb.L[2 * n + s]);
        if (frames != 0.0f) {
            f3 prev = mk3(p.x, p.y, p.z);
            prev = prev * frames;
            prev = prev + c;
            prev = prev / (frames + 1.f);
            c = prev;
        }
        p = make_float4(c.x, c.y, c.z, 1.0f);
    }
    if (any) pb.film[s1] = p;
}

// Film assembly after the all-gather of the ranks' tile slabs: pixel (x, y) lies in tile t = ty * tiles_x + tx, which
// rank t % world holds as its (t / world)-th tile, in 8x8 pixel blocks (the slot order of the tracer).
__global__ __launch_bounds__(RT_BLOCK) void k_unpack_slabs(const float4* __restrict__ slabs, uint64_t per_rank4, float4* __restrict__ film,
                                                          int W, int H, int tile, int world) {
    const uint32_t p = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (p >= (uint32_t)W * (uint32_t)H) return;
    const int x = (int)(p % (uint32_t)W), y = (int)(p / (uint32_t)W);
    const int tiles_x = (W + tile - 1) / tile, bpr = tile / 8;
    const int tx = x / tile, ty = y / tile, lx = x - tx * tile, ly = y - ty * tile;
    const uint32_t t = (uint32_t)ty * tiles_x + tx;
    const uint32_t rank = t % (uint32_t)world, k = t / (uint32_t)world;
    const uint32_t i = (uint32_t)((ly >> 3) * bpr + (lx >> 3)) * 64u + (uint32_t)((ly & 7) * 8 + (lx & 7));
    film[p] = slabs[(size_t)rank * per_rank4 + (size_t)k * tile * tile + i];
}

}  // namespace rt
