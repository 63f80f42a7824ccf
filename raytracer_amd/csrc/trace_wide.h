// Traversal of the 4-wide collapse of the LBVH (wide_bvh.hip): closest hit and any hit over the same work-item
// interface as trace_persistent (trace.h).
//
// What is the same as in the binary kernels: persistent waves that pull chunks of items from per-XCD work counters,
// one ray per lane, a loop of wave-uniform votes (refill / inner-node steps / one triangle test), the box test of
// src/scene/Ray.cpp:17-34 with its cull `dist < bound` (src/scene/BVH.cpp:155-158), the triangle test of
// src/scene/Ray.cpp:36-67 with the acceptance of src/scene/BVH.cpp:139 (closest) / :93 (any).  Rays the hardware
// min/max box test cannot decide like the reference's selects (non-finite operands) go to the deferred list and
// are re-traced by the binary tree's exact launch.
//
// What differs: a step tests the four child boxes of a wide node and visits the children that passed nearest
// first (sorted by their own box distance).  The set of triangles a ray can reach is the reference's
// (wide_bvh.hip explains why), so any-hit results are identical and closest-hit distances are the minimum over
// the same candidates; the visiting ORDER is not the reference's, so of several triangles at exactly the same
// distance another one may be reported (strict `<` of src/scene/BVH.cpp:139: the first visited wins).
//
// Inner step without per-lane branches:
//  * hit <=> max(tmin, tiny) <= min(tmax, prev(bound)), with tiny the smallest positive float and prev(x) the
//    float below x: for operands that are not NaN this is (tmin <= tmax) && (tmax > 0) && (tmin < bound), the
//    reference's three comparisons, as one.  Unused slots hold boxes at +inf, which fail it for every ray that
//    passed the root test.
//  * the near and far plane of each axis are fetched by per-ray offsets (the sign of the direction says which of
//    a box's two planes is which), so the test needs no min/max per axis;
//  * the four (key, link) pairs go through four comparators (nearest to position 0, farthest to 3), key =
//    max(tmin, tiny) or +inf;
//  * the traversal stack lives in LDS only, [level][thread], with a marker at the bottom: the three farther
//    children are stored above the stack pointer unconditionally (entries that did not pass land above the
//    new top and are overwritten later), the pointer advances by (children that passed) - 1, and a lane with no
//    child left takes the entry below the pointer, which was read at the start of the step.  A lane whose stack
//    would overflow sits out until the loop's (rare) fix-up section has moved its oldest entries to a global
//    buffer; popping the bottom marker of such a lane brings them back.
#pragma once
#include "trace.h"

namespace rt {

#ifndef RT_WSTACK
#define RT_WSTACK 10        // stack entries per lane in LDS, the bottom marker included
#endif
#ifndef RT_WSPILL
#define RT_WSPILL 4         // entries moved to / from the global buffer at a time
#endif
// The LDS-resident top of the 4-wide tree: the first RT_WTOP_NODES nodes of the breadth-first array (what is left
// of a workgroup's LDS beside the stack), each as the first 112 bytes of its record (six box planes and the
// links).  A record's planes sit at the same offsets as in global memory, so one set of per-ray plane offsets
// serves both; the stride of 28 dwords spreads random nodes over all sixteen 16-byte bank groups.
#ifndef RT_WLDS
#define RT_WLDS 81920       // LDS bytes of a workgroup: half of a CU's 160 KB, two workgroups of 16 waves per CU
#endif
#define RT_WTOP_NODES ((RT_WLDS - RT_WSTACK * RT_TBLOCK * 4) / 112)
#define RT_WTOP_BYTES (RT_WTOP_NODES * 112)
struct WideTop {
    uint32_t base;   // LDS byte address of node 0
    int32_t n;
};
RT_DEV WideTop stage_wide_top(const float4* wnodes, int32_t n_wnodes, char* lds) {
    float4* q = reinterpret_cast<float4*>(lds);
#ifdef RT_WEXP_NOLDS
    const int32_t n = n_wnodes < 1 ? n_wnodes : 1;
#else
    const int32_t n = n_wnodes < RT_WTOP_NODES ? n_wnodes : RT_WTOP_NODES;
#endif
    for (int32_t k = threadIdx.x; k < n * 7; k += RT_TBLOCK) {
        const int32_t node = k / 7, p = k - node * 7;
        q[k] = wnodes[(size_t)node * 8 + p];
    }
    clear_work_dry_mask();
    __syncthreads();
    WideTop t;
    t.base = (uint32_t)(uintptr_t)(RT_LDS char*)lds; t.n = n;
    return t;
}
RT_DEV tr_f32x4 lds128(uint32_t addr) { return *(const RT_LDS tr_f32x4*)(uintptr_t)addr; }
#define RT_UNSPILL0 ((int32_t)0x80000002)   // bottom marker of a lane with spilled entries: RT_UNSPILL0 + their number
#define RT_LEAF_LO ((int32_t)0xC0000000)    // leaf links (~slot) lie in [RT_LEAF_LO, -1]; the state codes lie below

#ifndef RT_WLB_WAVES
#define RT_WLB_WAVES RT_LB_WAVES
#endif
#ifndef RT_WLEAF_MIN
#define RT_WLEAF_MIN 8
#endif
#ifndef RT_WLEAF_MIN_ANY
#define RT_WLEAF_MIN_ANY 8
#endif
#ifndef RT_WPRIO
#define RT_WPRIO 1          // raise the wave's priority while it issues a node fetch
#endif
#ifndef RT_WINNER_REPS
#define RT_WINNER_REPS 2
#endif
#ifndef RT_WREFILL_MIN
#define RT_WREFILL_MIN 24   // refill when at least this many lanes are idle
#endif

#ifdef RT_PROBE_TIME
// Timing probe (development, tools/probe_tail.py): per wave of a wide launch, wall-clock (100 MHz) stamps written with
// plain stores: start, last successful claim of work, the moment it found every counter dry, end.
#define RT_TPROBE_WAVES 16384
__device__ unsigned long long g_trec[RT_TPROBE_WAVES * 4];
#define RT_TPROBE(slot) do { if (lane == 0 && (gtid >> 6) < RT_TPROBE_WAVES) g_trec[(gtid >> 6) * 4 + (slot)] = wall_clock64(); } while (0)
#else
#define RT_TPROBE(slot) do { } while (0)
#endif

RT_DEV float prev_bound(float b) {   // the largest x with x < b for b > 0 (b = +inf gives FLT_MAX); -1 if nothing is below b and above 0
    return (b > 0.0f) ? __int_as_float(__float_as_int(b) - 1) : -1.0f;
}

template <bool ANY_HIT, bool BOUNDED = false, bool MIXED = false, typename Load, typename Done, typename Probe = NoProbe>
RT_DEV void trace_wide(const TraceScene& sc, uint32_t count, TraceWork* work, uint32_t* deferred, int32_t* lds_stack,
                       const WideTop& top, Load load, Done done, Probe probe = Probe()) {
    const uint32_t tid = threadIdx.x;
    const uint32_t gtid = blockIdx.x * RT_TBLOCK + tid;
    const uint32_t lane = tid & 63u;
    const float inf = __builtin_inff();
    const float tiny = __int_as_float(1);
    RT_LDS int32_t* const stk = (RT_LDS int32_t*)lds_stack + tid;   // level j at stk[j * RT_TBLOCK]
    const __amdgpu_buffer_rsrc_t rs_ovf = __builtin_amdgcn_make_buffer_rsrc((void*)sc.ovf, (short)0, (int)(sc.ovf_levels * sc.ovf_stride * 4u), 0x00020000);
    const uint32_t ovf_lane = gtid * 4u, ovf_pitch = sc.ovf_stride * 4u;
    const uint32_t per_wave = count / (gridDim.x * (RT_TBLOCK / 64));
    const uint32_t chunk = per_wave >= RT_CHUNK ? RT_CHUNK : (per_wave >= 64 ? 64 : (per_wave >= 32 ? 32 : 16));
    const __amdgpu_buffer_rsrc_t rs_nodes = __builtin_amdgcn_make_buffer_rsrc((void*)sc.wnodes, (short)0, (int)((uint32_t)sc.n_wnodes * 128u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_tris = __builtin_amdgcn_make_buffer_rsrc((void*)sc.tris, (short)0, (int)((uint32_t)sc.n_tris * 48u), 0x00020000);
    int exhausted = 0;
    uint32_t steal = 0;
    uint32_t c_next = 0, c_end = 0;
    uint32_t item = 0;
    int32_t cur = RT_IDLE;
    int sp = 1;
    f3 org = mk3(0.f, 0.f, 0.f), dir = mk3(0.f, 0.f, 1.f), inv = mk3(0.f, 0.f, 1.f);
    HitRec h;
    h.t = inf; h.u = 0.f; h.v = 0.f; h.id0 = -1; h.id1 = -1;
    // per axis 48 if the ray moves towards smaller coordinates (its near plane is then the box's max plane), else 0
    uint32_t nx = 0, ny = 0, nz = 0;
    float boundm = 0.f;    // prev(bound): bound = the hit distance held (closest) or the far bound (any hit)
    static_assert(!BOUNDED || ANY_HIT, "a far bound only makes sense for any-hit rays");
    static_assert(!MIXED || (!ANY_HIT && !BOUNDED), "MIXED takes the ray kind per item");
    // MIXED launches keep the ray kind in the top bit of `item` (set = any hit); item indices stay below 2^31
    constexpr int leaf_min = (ANY_HIT || MIXED) ? RT_WLEAF_MIN_ANY : RT_WLEAF_MIN;

    RT_TPROBE(0);
    for (;;) {
        c_next = __builtin_amdgcn_readfirstlane(c_next);
        c_end = __builtin_amdgcn_readfirstlane(c_end);
        exhausted = __builtin_amdgcn_readfirstlane(exhausted);
        steal = __builtin_amdgcn_readfirstlane(steal);
        const unsigned long long m_idle = __ballot(cur <= RT_DONE);
        const int n_idle = __popcll(m_idle);
        if (n_idle >= RT_WREFILL_MIN && (c_next < c_end || !exhausted)) {
            RT_PROBE_ADD(0, 1); RT_PROBE_ADD(1, n_idle);
            // ---- refill idle lanes from the wave's chunk (as in trace_persistent)
            uint32_t mine;
            const bool have = take_items<false>(work, deferred, count, chunk, lane, cur <= RT_DONE, steal, c_next, c_end, exhausted, probe, mine);
            if (__ballot(have) != 0ull) RT_TPROBE(1);
            if (exhausted) RT_TPROBE(2);
            if (have) {
                if (cur == RT_DONE) done(MIXED ? (item & 0x7FFFFFFFu) : item, h);
                cur = RT_IDLE;
                const uint32_t i = mine;
                item = i;
                bool present;
                float tfar = inf;
                bool any_lane = ANY_HIT;
                if constexpr (MIXED) { present = load(i, org, dir, tfar, any_lane); if (any_lane) item |= 0x80000000u; }
                else if constexpr (BOUNDED) present = load(i, org, dir, tfar);
                else present = load(i, org, dir);
                inv = make_invdir(dir);
                nx = inv.x < 0.0f ? 48u : 0u; ny = inv.y < 0.0f ? 48u : 0u; nz = inv.z < 0.0f ? 48u : 0u;
                h.t = inf; h.u = 0.f; h.v = 0.f; h.id0 = -1; h.id1 = -1;
                const bool far_bounded = MIXED ? any_lane : BOUNDED;
                // every later test against the bound (boxes: dist < bound; triangles: dist < bound) is `<= boundm`
                boundm = far_bounded ? prev_bound(tfar) : 3.402823466e+38f;
                float rd;
                // root box test of src/scene/BVH.cpp:51-53
                if (!present) {
                } else if (slab_test(org, inv, sc.rmin[0], sc.rmin[1], sc.rmin[2], sc.rmax[0], sc.rmax[1], sc.rmax[2], rd)) {
                    // (a NaN far bound goes to the exact launch too: one register cannot say both what the reference
                    // does with it at a box and at a triangle)
                    const bool tame = sc.tame && moderate(org.x) && moderate(org.y) && moderate(org.z) &&
                                      moderate(inv.x) && moderate(inv.y) && moderate(inv.z) && (!far_bounded || tfar == tfar);
                    if (tame) {
                        cur = sc.wroot;
                        sp = 1;
                        stk[0] = RT_DONE;   // bottom marker
                    } else {
                        deferred[atomicAdd(&work->n_deferred, 1u)] = i;
                    }
                } else {
                    cur = RT_DONE;
                }
            }
            continue;
        }
        if (n_idle == 64) break;
        // ---- fix-up (rare): lanes whose LDS stack is nearly full move their oldest entries out; lanes that popped
        // the marker of a spilled stack bring entries back
        {
            const bool need_spill = cur > RT_DONE && sp > RT_WSTACK - 3;
            const bool need_fill = cur > RT_DONE && cur < RT_LEAF_LO;
            if (__ballot(need_spill || need_fill) != 0ull) {
                RT_PROBE_ADD(6, 1);
                if (need_spill) {
                    const int32_t m = stk[0];
                    const uint32_t have_out = (m == RT_DONE) ? 0u : (uint32_t)(m - RT_UNSPILL0);
                    for (int j = 0; j < RT_WSPILL; ++j)
                        __builtin_amdgcn_raw_buffer_store_b32((uint32_t)stk[(1 + j) * RT_TBLOCK], rs_ovf, (int)((have_out + j) * ovf_pitch + ovf_lane), 0, 0);
                    for (int j = 1 + RT_WSPILL; j < RT_WSTACK; ++j)
                        if (j < sp) stk[(j - RT_WSPILL) * RT_TBLOCK] = stk[j * RT_TBLOCK];
                    stk[0] = RT_UNSPILL0 + (int32_t)(have_out + RT_WSPILL);
                    sp -= RT_WSPILL;
                }
                if (need_fill) {
                    const uint32_t have_out = (uint32_t)(cur - RT_UNSPILL0);   // >= RT_WSPILL, a multiple of it
                    for (int j = 0; j < RT_WSPILL; ++j)
                        stk[(1 + j) * RT_TBLOCK] = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_ovf, (int)((have_out - RT_WSPILL + j) * ovf_pitch + ovf_lane), 0, 0);
                    stk[0] = (have_out > RT_WSPILL) ? RT_UNSPILL0 + (int32_t)(have_out - RT_WSPILL) : RT_DONE;
                    // the youngest of them is the entry the lane was about to take
                    cur = stk[RT_WSPILL * RT_TBLOCK];
                    sp = RT_WSPILL;
                }
            }
        }
        const bool at_leaf0 = cur < 0 && cur >= RT_LEAF_LO;
        if (__popcll(__ballot(at_leaf0)) < leaf_min && __ballot(cur >= 0 && sp <= RT_WSTACK - 3) != 0ull) {
#pragma unroll 1
            for (int rep = 0; rep < RT_WINNER_REPS; ++rep) {
                const bool act = cur >= 0 && sp <= RT_WSTACK - 3;
                RT_PROBE_ADD(2, 1); RT_PROBE_ADD(3, __popcll(__ballot(act))); RT_PROBE_ADD(7, __popcll(__ballot(cur >= 0 && !act)));
                if (act) {
                    RT_LDS int32_t* const sa = stk + (sp - 1) * RT_TBLOCK;
                    const int32_t tos = sa[0];
                    // The near plane of an axis is the box's min plane when the ray moves towards larger coordinates and
                    // its max plane otherwise: with lo = inv * (min - org) and hi = inv * (max - org), min <= max and
                    // rounding monotone, lo <= hi for inv > 0 and lo >= hi for inv < 0, so min(lo, hi) and max(lo, hi)
                    // of src/scene/Ray.cpp:23-24 are these two products themselves (for inv = 0 both are zero).  The
                    // planes are fetched by per-ray offsets instead of being selected after the fetch.
                    tr_f32x4 xn, yn, zn, xf, yf, zf, lkf;
                    // A wave that is about to issue its node fetch goes ahead of the waves that are doing arithmetic:
                    // the sooner the loads are out, the more of their latency the other waves' arithmetic covers
                    // (three-set launch -2 %, render pass -2.4 %, Bistro-class pass -4.8 %; the incoherent set launched
                    // alone +6 %).
#if RT_WPRIO
                    __builtin_amdgcn_s_setprio(3);
#endif
                    if (cur < top.n) {
                        const uint32_t a = top.base + (uint32_t)cur * 112u;
                        xn = lds128(a + nx); xf = lds128(a - nx + 48u);
                        yn = lds128(a + ny + 16u); yf = lds128(a - ny + 64u);
                        zn = lds128(a + nz + 32u); zf = lds128(a - nz + 80u);
                        lkf = lds128(a + 96u);
                    } else {
                        // (computing the six addresses once for both homes of a node, LDS stride 112 and global stride
                        // 128, needs seven more live registers: measured no faster even where they are free)
                        const uint32_t a = (uint32_t)cur << 7;   // < 4 GiB: checked by the host; cur >= 1 here (node 0 is in LDS)
                        const float4 a0 = ldb128(rs_nodes, a + nx), a3 = ldb128(rs_nodes, a - nx + 48u);
                        const float4 a1 = ldb128(rs_nodes, a + ny + 16u), a4 = ldb128(rs_nodes, a - ny + 64u);
                        const float4 a2 = ldb128(rs_nodes, a + nz + 32u), a5 = ldb128(rs_nodes, a - nz + 80u);
                        const float4 a6 = ldb128(rs_nodes, a + 96u);
                        xn = tr_f32x4{a0.x, a0.y, a0.z, a0.w}; yn = tr_f32x4{a1.x, a1.y, a1.z, a1.w}; zn = tr_f32x4{a2.x, a2.y, a2.z, a2.w};
                        xf = tr_f32x4{a3.x, a3.y, a3.z, a3.w}; yf = tr_f32x4{a4.x, a4.y, a4.z, a4.w}; zf = tr_f32x4{a5.x, a5.y, a5.z, a5.w};
                        lkf = tr_f32x4{a6.x, a6.y, a6.z, a6.w};
                    }
#if RT_WPRIO
                    __builtin_amdgcn_s_setprio(0);
#endif
                    float k0, k1, k2, k3;
                    bool h0, h1, h2, h3;
#define RT_WSLAB(c, key, hit)                                                                                                    \
    do {                                                                                                                         \
        const float tn = fmaxf(fmaxf(inv.x * (xn.c - org.x), fmaxf(inv.y * (yn.c - org.y), inv.z * (zn.c - org.z))), tiny);       \
        const float tf = fminf(fminf(inv.x * (xf.c - org.x), fminf(inv.y * (yf.c - org.y), inv.z * (zf.c - org.z))), boundm);     \
        hit = tn <= tf;                                                                                                          \
        key = hit ? tn : inf;                                                                                                    \
    } while (0)
                    RT_WSLAB(x, k0, h0); RT_WSLAB(y, k1, h1); RT_WSLAB(z, k2, h2); RT_WSLAB(w, k3, h3);
#undef RT_WSLAB
                    int32_t l0 = __float_as_int(lkf.x), l1 = __float_as_int(lkf.y), l2 = __float_as_int(lkf.z), l3 = __float_as_int(lkf.w);
                    // Four comparators: position 0 ends up with the nearest child and position 3 with the farthest; the
                    // two in the middle stay unordered (it only matters when three or four children passed).  On
                    // equal keys the lower slot stays first.  Which positions hold a child that passed follows from
                    // the four flags alone.
#define RT_CSWAP(ka, la, kb, lb)                                         \
    do {                                                                 \
        const bool sw_ = (kb) < (ka);                                    \
        const float kn_ = sw_ ? (kb) : (ka), kx_ = sw_ ? (ka) : (kb);    \
        const int32_t ln_ = sw_ ? (lb) : (la), lx_ = sw_ ? (la) : (lb);  \
        (ka) = kn_; (kb) = kx_; (la) = ln_; (lb) = lx_;                  \
    } while (0)
                    RT_CSWAP(k0, l0, k1, l1);
                    RT_CSWAP(k2, l2, k3, l3);
                    RT_CSWAP(k0, l0, k2, l2);
                    RT_CSWAP(k1, l1, k3, l3);
#undef RT_CSWAP
                    const bool v0 = h0 || h1 || h2 || h3;               // position 0: min of all
                    const bool v2 = (h0 || h1) && (h2 || h3);           // position 2: max(min(0,1), min(2,3))
                    const bool v3 = h0 && h1 && h2 && h3;               // position 3: max of all
                    // the children behind the nearest, compacted to the front (the farthest first when all four passed)
                    sa[RT_TBLOCK] = v3 ? l3 : (v2 ? l2 : l1);
                    sa[2 * RT_TBLOCK] = v3 ? l2 : l1;
                    sa[3 * RT_TBLOCK] = l1;
                    sp += (int)h0 + (int)h1 + (int)h2 + (int)h3 - 1;
                    cur = v0 ? l0 : tos;
                }
            }
        }
        // ---- one triangle test for the lanes standing on a leaf
        const bool at_leaf = cur < 0 && cur >= RT_LEAF_LO;
        const int n_leaf = __popcll(__ballot(at_leaf));
        if (n_leaf >= leaf_min || (n_leaf > 0 && __ballot(cur >= 0) == 0ull)) {
            RT_PROBE_ADD(4, 1); RT_PROBE_ADD(5, n_leaf);
            if (at_leaf) {
                const uint32_t toff = (uint32_t)(~cur) * 48u;
                float4 a = ldb128(rs_tris, toff), b = ldb128(rs_tris, toff + 16), c = ldb128(rs_tris, toff + 32);
                --sp;
                const int32_t below = stk[sp * RT_TBLOCK];
                float t, u, v;
                bool finished = false;
                if (tri_test(org, dir, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), t, u, v)) {
                    // src/scene/BVH.cpp:139 (closest) / :93 (any)
                    // boundm = prev(bound): t <= boundm is t < hit.dist (closest) or t < far bound (any hit; FLT_MAX = none)
                    const bool any_ray = MIXED ? ((int32_t)item < 0) : ANY_HIT;
                    if (t > 0.0001f && t <= boundm) {
                        h.t = t; h.u = u; h.v = v;
                        h.id1 = __float_as_int(c.y);
                        h.id0 = __float_as_int(c.z);
                        if (any_ray) finished = true;
                        else boundm = __int_as_float(__float_as_int(t) - 1);
                    }
                }
                cur = finished ? RT_DONE : below;
            }
        }
    }
    if (cur == RT_DONE) done(MIXED ? (item & 0x7FFFFFFFu) : item, h);
    RT_TPROBE(3);
}

// Closest hit over a ray set on the 4-wide tree (the batch form of BVH::intersect, tie-tolerant as described
// above); the follow-up launch over the deferred list is k_trace_closest<true>.
__global__ __launch_bounds__(RT_TBLOCK RT_WLB_WAVES) void k_trace_closest_wide(TraceScene sc, const float* __restrict__ rays, uint32_t stride, uint32_t n,
                                                                 TraceWork* __restrict__ work, uint32_t* __restrict__ deferred,
                                                                 float* __restrict__ o_t, int32_t* __restrict__ o_id0,
                                                                 int32_t* __restrict__ o_id1, float* __restrict__ o_u, float* __restrict__ o_v,
                                                                 unsigned long long* __restrict__ ray_counter) {
    __shared__ int32_t s_stack[RT_WSTACK * RT_TBLOCK];
    __shared__ __attribute__((aligned(16))) char s_top[RT_WTOP_BYTES];
    const WideTop top = stage_wide_top(sc.wnodes, sc.n_wnodes, s_top);
    if (ray_counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(ray_counter, (unsigned long long)n);
    const size_t st = stride;
    trace_wide<false>(
        sc, n, work, deferred, s_stack, top,
        [&](uint32_t i, f3& org, f3& dir) -> bool {
            org = mk3(rays[i], rays[st + i], rays[2 * st + i]);
            dir = mk3(rays[3 * st + i], rays[4 * st + i], rays[5 * st + i]);
            return true;
        },
        [&](uint32_t i, const HitRec& h) {
            o_t[i] = h.t; o_id0[i] = h.id0; o_id1[i] = h.id1; o_u[i] = h.u; o_v[i] = h.v;
        });
}

// Any hit over a ray set on the 4-wide tree (BVH::occluded; BOUNDED: with a far bound per ray).
template <bool BOUNDED>
__global__ __launch_bounds__(RT_TBLOCK RT_WLB_WAVES) void k_trace_occluded_wide(TraceScene sc, const float* __restrict__ rays, const float* __restrict__ tmax,
                                                                  uint32_t stride, uint32_t n, TraceWork* __restrict__ work,
                                                                  uint32_t* __restrict__ deferred, int32_t* __restrict__ status) {
    __shared__ int32_t s_stack[RT_WSTACK * RT_TBLOCK];
    __shared__ __attribute__((aligned(16))) char s_top[RT_WTOP_BYTES];
    const WideTop top = stage_wide_top(sc.wnodes, sc.n_wnodes, s_top);
    const size_t st = stride;
    if constexpr (BOUNDED) {
        trace_wide<true, true>(
            sc, n, work, deferred, s_stack, top,
            [&](uint32_t i, f3& org, f3& dir, float& tfar) -> bool {
                org = mk3(rays[i], rays[st + i], rays[2 * st + i]);
                dir = mk3(rays[3 * st + i], rays[4 * st + i], rays[5 * st + i]);
                tfar = tmax[i];
                return true;
            },
            [&](uint32_t i, const HitRec& h) { status[i] = (h.id1 >= 0) ? 1 : 0; });
    } else {
        trace_wide<true>(
            sc, n, work, deferred, s_stack, top,
            [&](uint32_t i, f3& org, f3& dir) -> bool {
                org = mk3(rays[i], rays[st + i], rays[2 * st + i]);
                dir = mk3(rays[3 * st + i], rays[4 * st + i], rays[5 * st + i]);
                return true;
            },
            [&](uint32_t i, const HitRec& h) { status[i] = (h.id1 >= 0) ? 1 : 0; });
    }
}

// Dispatch between the two trees by the type of the staged top: the kernels of the wavefront tracer are written once.
template <bool WIDE>
RT_DEV auto stage_top(const TraceScene& sc, char* lds) {
    if constexpr (WIDE) return stage_wide_top(sc.wnodes, sc.n_wnodes, lds);
    else return stage_top_tree(sc.nodes, sc.n_nodes, lds);
}
template <bool ANY_HIT, bool EXACT, bool BOUNDED, bool MIXED, typename Load, typename Done, typename Probe = NoProbe>
RT_DEV void trace_rays(const TraceScene& sc, uint32_t count, TraceWork* work, uint32_t* deferred, int32_t* lds_stack,
                       const TopTree& top, Load load, Done done, Probe probe = Probe()) {
    trace_persistent<ANY_HIT, EXACT, BOUNDED, MIXED>(sc, count, work, deferred, lds_stack, top, load, done, probe);
}
template <bool ANY_HIT, bool EXACT, bool BOUNDED, bool MIXED, typename Load, typename Done, typename Probe = NoProbe>
RT_DEV void trace_rays(const TraceScene& sc, uint32_t count, TraceWork* work, uint32_t* deferred, int32_t* lds_stack,
                       const WideTop& top, Load load, Done done, Probe probe = Probe()) {
    static_assert(!EXACT, "the exact follow-up launch walks the binary tree");
    trace_wide<ANY_HIT, BOUNDED, MIXED>(sc, count, work, deferred, lds_stack, top, load, done, probe);
}

// k_extend (trace.h) on the 4-wide tree: closest hit for the slots a queue names.
__global__ __launch_bounds__(RT_TBLOCK RT_WLB_WAVES) void k_extend_wide(TraceScene sc, const float* __restrict__ rays, uint32_t stride,
                                                          const uint32_t* __restrict__ queue, const uint32_t* __restrict__ d_count,
                                                          TraceWork* __restrict__ work, uint32_t* __restrict__ deferred,
                                                          float* __restrict__ o_t, int32_t* __restrict__ o_id0, int32_t* __restrict__ o_id1,
                                                          float* __restrict__ o_u, float* __restrict__ o_v,
                                                          unsigned long long* __restrict__ ray_counter) {
    __shared__ int32_t s_stack[RT_WSTACK * RT_TBLOCK];
    __shared__ __attribute__((aligned(16))) char s_top[RT_WTOP_BYTES];
    const WideTop top = stage_wide_top(sc.wnodes, sc.n_wnodes, s_top);
    const uint32_t count = *d_count;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(ray_counter, (unsigned long long)count);
    const size_t st = stride;
    trace_wide<false>(
        sc, count, work, deferred, s_stack, top,
        [&](uint32_t i, f3& org, f3& dir) -> bool {
            const uint32_t s = queue[i];
            org = mk3(rays[s], rays[st + s], rays[2 * st + s]);
            dir = mk3(rays[3 * st + s], rays[4 * st + s], rays[5 * st + s]);
            return true;
        },
        [&](uint32_t i, const HitRec& h) {
            const uint32_t s = queue[i];
            o_t[s] = h.t; o_id0[s] = h.id0; o_id1[s] = h.id1; o_u[s] = h.u; o_v[s] = h.v;
        });
}

}  // namespace rt
