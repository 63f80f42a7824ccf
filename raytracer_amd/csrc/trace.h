// BVH traversal kernels for gfx950: closest hit (BVH::intersect,
// src/scene/BVH.cpp:49-61,129-170) and any hit (BVH::occluded,
// src/scene/BVH.cpp:63-127) over SoA ray buffers.
//
// One ray per lane.  The reference's recursion is replayed with an explicit
// per-lane stack: when both children are hit the far one is pushed and is
// NOT re-tested against the shrunken hit distance when popped, exactly like
// the second recursive call at src/scene/BVH.cpp:165-166, so every cull
// decision and therefore every hit id equals the scalar path's.
//
// Stack: the first RT_STACK_LDS entries of each lane live in LDS, laid out
// [level][thread] so that a wave's access to any mix of levels is
// bank-conflict-free; deeper entries spill to a global buffer indexed by the
// launch-wide thread id (the grid is persistent, so that buffer is small).
#pragma once
#include "device_math.h"

namespace rt {

#define RT_BLOCK 256
#define RT_STACK_LDS 24

struct TraceScene {
    const float4* nodes;   // 4 float4 per inner node (DevNode)
    const float4* tris;    // 3 float4 per triangle (DevTri)
    int32_t root_link;
    float rmin[3], rmax[3];
    int32_t* ovf;          // overflow stack, [level - RT_STACK_LDS][grid thread]
    uint32_t ovf_stride;   // threads in the persistent grid
    uint32_t ovf_levels;
};

// src/scene/Ray.cpp:17-34
RT_DEV bool slab_test(f3 org, f3 inv, float bx0, float by0, float bz0, float bx1, float by1, float bz1, float& dist) {
    float lox = inv.x * (bx0 - org.x), loy = inv.y * (by0 - org.y), loz = inv.z * (bz0 - org.z);
    float hix = inv.x * (bx1 - org.x), hiy = inv.y * (by1 - org.y), hiz = inv.z * (bz1 - org.z);
    float mxx = gmax(lox, hix), mxy = gmax(loy, hiy), mxz = gmax(loz, hiz);
    float mnx = gmin(lox, hix), mny = gmin(loy, hiy), mnz = gmin(loz, hiz);
    float tmin = gmax(mnx, gmax(mny, mnz));
    float tmax = gmin(mxx, gmin(mxy, mxz));
    dist = tmin;
    return (tmin <= tmax) && (tmax > 0.0f);
}

// src/scene/Ray.cpp:36-67
RT_DEV bool tri_test(f3 org, f3 dir, f3 v0, f3 v1, f3 v2, float& dist, float& u, float& v) {
    f3 e1 = v1 - v0;
    f3 e2 = v2 - v0;
    f3 pvec = cross3(dir, e2);
    float det = dot3(e1, pvec);
    if ((double)fabsf(det) < 1e-8) return false;  // the literal is a double in the reference
    float inv_det = 1 / det;
    f3 tvec = org - v0;
    u = dot3(tvec, pvec) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    f3 qvec = cross3(tvec, e1);
    v = dot3(dir, qvec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    dist = dot3(e2, qvec) * inv_det;
    return true;
}

struct HitRec {
    float t, u, v;
    int32_t id0, id1;
};

// src/scene/Ray.hpp:10-24
RT_DEV f3 make_invdir(f3 d) {
    f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    if (d.x == 0.0f) inv.x = 0.0f;
    if (d.y == 0.0f) inv.y = 0.0f;
    if (d.z == 0.0f) inv.z = 0.0f;
    return inv;
}

template <bool ANY_HIT>
RT_DEV HitRec trace_ray(const TraceScene& sc, f3 org, f3 dir, int32_t* lds_stack, uint32_t gtid) {
    HitRec h;
    h.t = __builtin_inff();
    h.u = 0.f; h.v = 0.f; h.id0 = -1; h.id1 = -1;
    f3 inv = make_invdir(dir);
    float rd;
    if (!slab_test(org, inv, sc.rmin[0], sc.rmin[1], sc.rmin[2], sc.rmax[0], sc.rmax[1], sc.rmax[2], rd)) return h;

    int32_t cur = sc.root_link;
    int sp = 0;
    const uint32_t tid = threadIdx.x;
    for (;;) {
        if (cur >= 0) {
            const float4* np = sc.nodes + (size_t)cur * 4;
            float4 a = np[0], b = np[1], c = np[2], d = np[3];
            float ld, rdist;
            bool ls = slab_test(org, inv, a.x, a.y, a.z, a.w, b.x, b.y, ld);
            bool rs = slab_test(org, inv, b.z, b.w, c.x, c.y, c.z, c.w, rdist);
            ls = ls && (ld < h.t);
            rs = rs && (rdist < h.t);
            int32_t left = __float_as_int(d.x), right = __float_as_int(d.y);
            if (ls && rs) {
                int32_t nearL = left, farL = right;
                if (rdist < ld) { nearL = right; farL = left; }
                if (sp < RT_STACK_LDS) lds_stack[sp * RT_BLOCK + tid] = farL;
                else sc.ovf[(size_t)(sp - RT_STACK_LDS) * sc.ovf_stride + gtid] = farL;
                ++sp;
                cur = nearL;
                continue;
            } else if (ls) {
                cur = left;
                continue;
            } else if (rs) {
                cur = right;
                continue;
            }
        } else {
            const float4* tp = sc.tris + (size_t)(~cur) * 3;
            float4 a = tp[0], b = tp[1], c = tp[2];
            float t, u, v;
            if (tri_test(org, dir, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), t, u, v)) {
                if (ANY_HIT) {
                    if (t > 0.0001f) { h.t = t; h.u = u; h.v = v; h.id1 = __float_as_int(c.y); h.id0 = __float_as_int(c.z); return h; }
                } else if (t > 0.0001f && t < h.t) {
                    h.t = t; h.u = u; h.v = v;
                    h.id1 = __float_as_int(c.y);
                    h.id0 = __float_as_int(c.z);
                }
            }
        }
        if (sp == 0) break;
        --sp;
        cur = (sp < RT_STACK_LDS) ? lds_stack[sp * RT_BLOCK + tid]
                                  : sc.ovf[(size_t)(sp - RT_STACK_LDS) * sc.ovf_stride + gtid];
    }
    if (ANY_HIT) h.t = __builtin_inff();
    return h;
}

// Closest hit over a ray set.  rays: 6 planes of `stride` floats.  When
// `queue` is non-null entry i names the ray slot, otherwise slot = i.  The
// number of entries is *d_count when d_count is non-null, else n.
__global__ __launch_bounds__(RT_BLOCK) void k_trace_closest(TraceScene sc, const float* __restrict__ rays, uint32_t stride,
                                                            const uint32_t* __restrict__ queue, const uint32_t* __restrict__ d_count,
                                                            uint32_t n, float* __restrict__ o_t, int32_t* __restrict__ o_id0,
                                                            int32_t* __restrict__ o_id1, float* __restrict__ o_u, float* __restrict__ o_v,
                                                            unsigned long long* __restrict__ ray_counter) {
    __shared__ int32_t s_stack[RT_STACK_LDS * RT_BLOCK];
    const uint32_t count = d_count ? *d_count : n;
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t gsize = gridDim.x * RT_BLOCK;
    if (ray_counter && gtid == 0) atomicAdd(ray_counter, (unsigned long long)count);
    for (uint32_t i = gtid; i < count; i += gsize) {
        uint32_t s = queue ? queue[i] : i;
        f3 org = mk3(rays[s], rays[stride + s], rays[2 * (size_t)stride + s]);
        f3 dir = mk3(rays[3 * (size_t)stride + s], rays[4 * (size_t)stride + s], rays[5 * (size_t)stride + s]);
        HitRec h = trace_ray<false>(sc, org, dir, s_stack, gtid);
        o_t[s] = h.t; o_id0[s] = h.id0; o_id1[s] = h.id1; o_u[s] = h.u; o_v[s] = h.v;
    }
}

// Any hit over a ray set; status[i] = 1 when something is hit.
__global__ __launch_bounds__(RT_BLOCK) void k_trace_occluded(TraceScene sc, const float* __restrict__ rays, uint32_t stride, uint32_t n,
                                                             int32_t* __restrict__ status) {
    __shared__ int32_t s_stack[RT_STACK_LDS * RT_BLOCK];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t gsize = gridDim.x * RT_BLOCK;
    for (uint32_t i = gtid; i < n; i += gsize) {
        f3 org = mk3(rays[i], rays[stride + i], rays[2 * (size_t)stride + i]);
        f3 dir = mk3(rays[3 * (size_t)stride + i], rays[4 * (size_t)stride + i], rays[5 * (size_t)stride + i]);
        HitRec h = trace_ray<true>(sc, org, dir, s_stack, gtid);
        status[i] = (h.id1 >= 0) ? 1 : 0;
    }
}

}  // namespace rt
