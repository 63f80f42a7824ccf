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
// Scheduling (free, because each lane's own sequence of steps is what fixes
// the result): waves are persistent and pull rays in chunks from the launch's
// work counters (one per contiguous range of the ray set, the ranges dealt to
// the XCDs); every loop iteration the wave runs up to three sections, each
// behind a wave-uniform vote -- refill idle lanes with new rays, a few
// inner-node steps, one triangle test -- so lanes whose ray has finished do
// not idle until the slowest ray of the wave is done, and the triangle body
// only issues when enough lanes wait for it.
//
// LDS (per 1024-thread workgroup: 1023 nodes x 56 B = 56 KB of nodes + 24 KB
// of stack; two workgroups = 32 waves per CU take 2 x 81,920 B, all 160 KB):
//  * the top of the tree: the inner nodes of levels 0..RT_TOP_LEVELS-1, which
//    the builder stores at their heap index, as four SoA planes.  About half
//    of all node visits land there and cost no memory traffic;
//  * the traversal stack: the first RT_STACK_LDS entries of each lane, laid
//    out [level][thread] so a wave's access to any mix of levels is
//    bank-conflict-free; deeper entries spill to a global buffer indexed by
//    the launch-wide thread id (the grid is persistent, so it is small).
#pragma once
#include "device_math.h"

namespace rt {

#define RT_BLOCK 256        // shading kernels
#ifndef RT_TBLOCK
#define RT_TBLOCK 1024      // traversal kernels: 2 workgroups (32 waves) per CU, each with its own top-tree copy
#endif
#ifndef RT_STACK_LDS
#define RT_STACK_LDS 6
#endif
#ifndef RT_TOP_LEVELS
#define RT_TOP_LEVELS 10    // tree levels whose inner nodes are kept in LDS (56 B per slot)
#endif
#define RT_TOP_NODES ((1 << RT_TOP_LEVELS) - 1)
// vote thresholds (lanes of 64)
#ifndef RT_LB_WAVES
#define RT_LB_WAVES , 8      // second launch bound: 8 waves per SIMD, i.e. at most 64 VGPRs
#endif
#ifndef RT_INNER_REPS
#define RT_INNER_REPS 4      // inner-node steps per vote of the traversal loop
#endif
#ifndef RT_LEAF_MIN_ANY
#define RT_LEAF_MIN_ANY 12   // leaf-step threshold of launches that carry any-hit rays
#endif
#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 20   // refill when at least this many lanes are idle
#endif
#ifndef RT_LEAF_MIN
#define RT_LEAF_MIN 8      // run a triangle step when at least this many lanes wait on a leaf
#endif
#ifndef RT_CHUNK
#define RT_CHUNK 96        // items a wave takes from a work counter at a time
#endif
#ifndef RT_BPRIO
#define RT_BPRIO 1         // raise the wave's priority between the end of the box tests and the next node fetch
#endif
#ifndef RT_XCD_RANGES
#define RT_XCD_RANGES 8    // the items of a launch are cut into 8 x this many contiguous ranges
#endif

struct TraceScene {
    const float4* nodes;   // 4 float4 per inner node (DevNode)
    const float4* tris;    // 3 float4 per triangle (DevTri)
    int32_t root_link;
    int32_t n_nodes;       // node slots
    int32_t n_tris;        // triangle slots
    float rmin[3], rmax[3];
    int32_t* ovf;          // overflow stack, [level - RT_STACK_LDS][grid thread]
    uint32_t ovf_stride;   // threads in the persistent grid
    uint32_t ovf_levels;
    int32_t tame;          // every box coordinate is finite and below 1e30 in magnitude
    const float4* wnodes;  // 4-wide collapse of the same tree: 8 float4 per node (wide_bvh.h), breadth-first
    int32_t n_wnodes;
    int32_t wroot;         // 0, or the root's leaf link when the tree is a single triangle
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

// The same test with hardware min/max.  v_min_f32/v_max_f32 pick the same
// operand as glm's (y<x)?y:x / (x<y)?y:x whenever neither input is a NaN
// (they may differ in the sign of a zero result, which no comparison below
// can see), so this form is used only for rays whose origin and inverse
// direction are finite and moderate: then inv*(b-org) is never inf*0.
RT_DEV bool slab_test_fast(f3 org, f3 inv, float bx0, float by0, float bz0, float bx1, float by1, float bz1, float& dist) {
    float lox = inv.x * (bx0 - org.x), loy = inv.y * (by0 - org.y), loz = inv.z * (bz0 - org.z);
    float hix = inv.x * (bx1 - org.x), hiy = inv.y * (by1 - org.y), hiz = inv.z * (bz1 - org.z);
    float tmin = fmaxf(fminf(lox, hix), fmaxf(fminf(loy, hiy), fminf(loz, hiz)));
    float tmax = fminf(fmaxf(lox, hix), fminf(fmaxf(loy, hiy), fmaxf(loz, hiz)));
    dist = tmin;
    return (tmin <= tmax) && (tmax > 0.0f);
}

typedef float tr_f32x4 __attribute__((ext_vector_type(4)));
typedef int32_t tr_i32x2 __attribute__((ext_vector_type(2)));
#define RT_LDS __attribute__((address_space(3)))

// The workgroup's mask of work ranges that one of its waves has found dry (bit x = range x), in LDS; the kernels'
// top-tree staging functions clear it before their barrier.
RT_DEV RT_LDS uint32_t* work_dry_mask() {
    __shared__ uint32_t mask[4];
    return (RT_LDS uint32_t*)mask;
}
RT_DEV void clear_work_dry_mask() {
    if (threadIdx.x < 4) work_dry_mask()[threadIdx.x] = 0u;
}

// Pointers into LDS are typed as such: through generic pointers the compiler merges the LDS and the global form of
// a fetch into flat loads of a selected address.

// Per-lane traversal stack: levels below RT_STACK_LDS in LDS, the rest in
// the global overflow buffer.  (Putting the overflow accesses behind a
// wave-uniform vote instead of a per-lane branch measured 1-4 % slower: some
// lane of a wave is that deep most of the time.)
struct LaneStack {
    RT_LDS int32_t* lds;           // &stack[tid]
    __amdgpu_buffer_rsrc_t ovf;    // the overflow buffer [level - RT_STACK_LDS][grid thread] as a buffer resource
    uint32_t ovf_lane;             // byte offset of this thread's column
    uint32_t ovf_pitch;            // bytes per level
    RT_DEV void push(int sp, int32_t x) const {
        if (sp < RT_STACK_LDS) lds[sp * RT_TBLOCK] = x;
        else __builtin_amdgcn_raw_buffer_store_b32((uint32_t)x, ovf, (int)((uint32_t)(sp - RT_STACK_LDS) * ovf_pitch + ovf_lane), 0, 0);
    }
    RT_DEV int32_t pop(int sp) const {
        int32_t x;
        if (sp < RT_STACK_LDS) {
            x = lds[sp * RT_TBLOCK];
        } else {
            x = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(ovf, (int)((uint32_t)(sp - RT_STACK_LDS) * ovf_pitch + ovf_lane), 0, 0);
        }
        return x;
    }
};

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

RT_DEV bool moderate(float x) { return fabsf(x) < 1e30f; }  // false for NaN and inf too

// The LDS-resident top of the tree, SoA: node k = (b0[k], b1[k], b2[k]) child
// boxes (48 B, DevNode's first three float4) and lk[k] = (left, right).
struct TopTree {
    const RT_LDS tr_f32x4* b0;
    const RT_LDS tr_f32x4* b1;
    const RT_LDS tr_f32x4* b2;
    const RT_LDS tr_i32x2* lk;
    int32_t n;   // nodes resident (a prefix of the node array)
};

#define RT_TOP_BYTES (RT_TOP_NODES * 56)

// Copies the first min(n_nodes, RT_TOP_NODES) nodes into LDS; every thread
// of the workgroup calls it, followed by a barrier.
RT_DEV TopTree stage_top_tree(const float4* nodes, int32_t n_nodes, char* lds) {
    float4* b0 = reinterpret_cast<float4*>(lds);
    float4* b1 = b0 + RT_TOP_NODES;
    float4* b2 = b1 + RT_TOP_NODES;
    int2* lk = reinterpret_cast<int2*>(b2 + RT_TOP_NODES);
    const int32_t n = n_nodes < RT_TOP_NODES ? n_nodes : RT_TOP_NODES;
    for (int32_t k = threadIdx.x; k < n; k += RT_TBLOCK) {
        const float4* np = nodes + (size_t)k * 4;
        b0[k] = np[0]; b1[k] = np[1]; b2[k] = np[2];
        float4 d = np[3];
        lk[k] = make_int2(__float_as_int(d.x), __float_as_int(d.y));
    }
    clear_work_dry_mask();
    __syncthreads();
    TopTree t;
    t.b0 = (const RT_LDS tr_f32x4*)b0; t.b1 = (const RT_LDS tr_f32x4*)b1; t.b2 = (const RT_LDS tr_f32x4*)b2;
    t.lk = (const RT_LDS tr_i32x2*)lk; t.n = n;
    return t;
}

// Work counters of one traversal launch (zeroed before the launch).
struct TraceWork {
    uint32_t n_deferred;   // items handed to the exact launch
    uint32_t next_exact;   // next unclaimed entry of the deferred list
    uint32_t pad[14];
    uint32_t next_x[8 * RT_XCD_RANGES * 16];  // next unclaimed item of each range of the main launch, relative to the
                                              // range's start (one counter per 64-byte line)
};

// Persistent traversal of `count` work items.  `load(i, org, dir)` fetches
// work item i and returns false when the item holds no ray (it is then
// dropped without a call to `done`); `done(i, hit)` consumes a ray's result
// (hit.id1 < 0 = nothing hit).
//
// EXACT = false is the main launch: box tests use hardware min/max, and a
// ray whose origin or inverse direction is not finite and moderate (where
// inf*0 could make a NaN and min/max would no longer equal the reference's
// selects) is not traced but appended to `deferred`.  EXACT = true is the
// follow-up launch over that list (count = work->n_deferred) with the
// select-based box test; it is almost always empty.
//
// BOUNDED (any-hit only): `load` also yields a far bound; the ray is only
// occluded by hits in front of it, and the bound takes the place of the
// never-shrinking hit.dist of the reference's loop (src/scene/BVH.cpp:70,106-107).
// A bound of +inf is the reference's unbounded shadow ray.
//
// MIXED: closest-hit and any-hit rays in one launch; `load` says per item which
// it is (and gives any-hit rays their far bound, +inf if none).
//
// Shape of the loop (it decides the instruction count: an earlier form with
// `continue` on several paths and a per-lane `have` flag made the compiler
// treat the loop as divergent and copy all ~20 state registers twice per
// iteration, about one inner step's worth of VALU work):
//  * a lane's state is `cur` alone: RT_IDLE = no ray, RT_DONE = no ray but a result to write (the two
//    smallest integers), >= 0 = inner node, otherwise ~triangle slot;
//  * the chunk bounds and the exhausted flag go through readfirstlane at the
//    top of every iteration, so every vote is a scalar branch and the loop has
//    one uniform back edge;
//  * a finished ray's result stays in the lane's registers (state RT_DONE) and is
//    written when the lane is refilled (or after the loop), which keeps the
//    five result stores out of the inner-node and triangle sections.
#define RT_IDLE ((int32_t)0x80000000)   // no ray
#define RT_DONE ((int32_t)0x80000001)   // no ray in flight, but the lane holds the unwritten result of its last ray
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// 16 bytes at byte offset `off` of a buffer resource
RT_DEV float4 ldb128(__amdgpu_buffer_rsrc_t rs, uint32_t off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
#ifdef RT_PROBE
// development probe (never in the shipped build): wave-step and lane-step counts of the loop's sections,
// before and after the work counter ran dry
__device__ unsigned long long g_probe[16];
#define RT_PROBE_ADD(k, v) do { const unsigned long long pv_ = (unsigned long long)(v); if (lane == 0) atomicAdd(&g_probe[(k) + (exhausted ? 8 : 0)], pv_); } while (0)
#else
#define RT_PROBE_ADD(k, v) do { } while (0)
#endif
// Default `probe` of trace_persistent: every item carries a ray.
struct NoProbe {
    static constexpr bool kFilters = false;
    RT_DEV bool operator()(uint32_t) const { return true; }
};
// The wave's next chunk of work items.  The items of a launch are cut into 8 x RT_XCD_RANGES contiguous ranges with
// a counter each (one per 64-byte line).  Workgroups go to the eight XCDs round-robin, so workgroup b sits on XCD
// b % 8: its waves take chunks of that XCD's own ranges x, x + 8, x + 16, ... first -- a mix of the whole launch, so
// that the XCDs get like shares of a ray set whose cost varies along it, yet each XCD's L2 sees a few compact parts of
// the set instead of all of it -- and then help with the other XCDs' ranges.  The workgroups of an XCD start on
// different ranges of its own, so their claims go to different counters.  (Were the mapping of workgroups to XCDs
// different the result would be the same, only the locality would be lost.)  `steal` counts the ranges this wave has
// found dry; a range one wave of the workgroup has found dry is not asked again by the fifteen others: at the end of a
// launch every wave walks all the ranges, and atomics on one address retire one after the other.
RT_DEV bool claim_chunk(TraceWork* work, uint32_t count, uint32_t chunk, uint32_t lane, uint32_t& steal, uint32_t& c_next, uint32_t& c_end) {
    constexpr uint32_t n_ranges = 8u * RT_XCD_RANGES;
    static_assert(n_ranges <= 128, "the dry mask is four words");
    RT_LDS uint32_t* const dry = work_dry_mask();
    const uint32_t w0 = (blockIdx.x >> 3) % RT_XCD_RANGES;
    while (steal < n_ranges) {
        const uint32_t x = ((blockIdx.x + steal / RT_XCD_RANGES) & 7u) + 8u * ((steal + w0) % RT_XCD_RANGES);
        const uint32_t dry_word = __builtin_amdgcn_readfirstlane(((volatile RT_LDS uint32_t*)dry)[x >> 5]);
        if ((dry_word >> (x & 31u)) & 1u) { ++steal; continue; }
        const uint32_t lo = (uint32_t)((unsigned long long)count * x / n_ranges);
        const uint32_t hi = (uint32_t)((unsigned long long)count * (x + 1) / n_ranges);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&work->next_x[x * 16], chunk);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base < hi - lo) {
            c_next = lo + base;
            c_end = (hi - lo - base < chunk) ? hi : lo + base + chunk;
            return true;
        }
        if (lane == 0) __hip_atomic_fetch_or(dry + (x >> 5), 1u << (x & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ++steal;
    }
    return false;
}

#ifndef RT_PROBE_ROUNDS
#define RT_PROBE_ROUNDS 3
#endif

// One refill: the idle lanes of the wave (`idle`) take the next work items, claiming further chunks as the wave's own
// runs out, so that every lane that wants an item gets one while the launch has any.  Returns whether this lane took
// one (`mine`).  An item may carry no ray (a shadow entry that was never queued): `probe` tells from one byte, and a
// lane that drew such an item draws again, up to RT_PROBE_ROUNDS times, before the rays of the lanes that found one
// are loaded together by the caller.  EXACT launches run over the deferred list with its single counter.
template <bool EXACT, typename Probe>
RT_DEV bool take_items(TraceWork* work, const uint32_t* deferred, uint32_t count, uint32_t chunk, uint32_t lane, bool idle,
                       uint32_t& steal, uint32_t& c_next, uint32_t& c_end, int& exhausted, Probe& probe, uint32_t& mine) {
    bool have = false;
    mine = 0;
    int round = 0;
    for (;;) {
        const unsigned long long m_want = __ballot(idle && !have);
        const uint32_t n_want = (uint32_t)__popcll(m_want);
        if (n_want == 0) break;
        if (c_next >= c_end) {
            if (exhausted) break;
            bool got;
            if constexpr (EXACT) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&work->next_exact, chunk);
                base = __builtin_amdgcn_readfirstlane(base);
                got = base < count;
                if (got) {
                    c_next = base;
                    c_end = (count - base < chunk) ? count : base + chunk;
                }
            } else {
                got = claim_chunk(work, count, chunk, lane, steal, c_next, c_end);
            }
            if (!got) {
                exhausted = 1;
                break;
            }
        }
        const uint32_t avail = c_end - c_next;
        const uint32_t k = avail < n_want ? avail : n_want;
        const uint32_t r = __popcll(m_want & ((1ull << lane) - 1ull));
        if (idle && !have && r < k) {
            uint32_t i = c_next + r;
            if (EXACT) i = deferred[i];
            if (probe(i)) { have = true; mine = i; }
        }
        c_next += k;
        if (Probe::kFilters) {
            if (++round >= RT_PROBE_ROUNDS) break;
        } else if (k == n_want) {
            break;
        }
    }
    return have;
}
template <bool ANY_HIT, bool EXACT, bool BOUNDED = false, bool MIXED = false, typename Load, typename Done, typename Probe = NoProbe>
RT_DEV void trace_persistent(const TraceScene& sc, uint32_t count, TraceWork* work, uint32_t* deferred, int32_t* lds_stack,
                             const TopTree& top, Load load, Done done, Probe probe = Probe()) {
    const uint32_t tid = threadIdx.x;
    const uint32_t gtid = blockIdx.x * RT_TBLOCK + tid;
    const uint32_t lane = tid & 63u;
    const float inf = __builtin_inff();
    LaneStack stk;
    stk.lds = (RT_LDS int32_t*)lds_stack + tid;
    // 32-bit byte offsets: the host keeps levels x grid threads x 4 bytes below 4 GiB (at most 64 levels x 2^19 threads)
    stk.ovf = __builtin_amdgcn_make_buffer_rsrc((void*)sc.ovf, (short)0, (int)(sc.ovf_levels * sc.ovf_stride * 4u), 0x00020000);
    stk.ovf_lane = gtid * 4u;
    stk.ovf_pitch = sc.ovf_stride * 4u;
    // Items a wave takes from the shared counter at a time: RT_CHUNK when
    // there is plenty of work (few atomics, contiguous rays per wave), fewer
    // when the launch is small, so that a short queue spreads over many waves
    // instead of running as long serial chains on a few of them.
    const uint32_t per_wave = count / (gridDim.x * (RT_TBLOCK / 64));
    const uint32_t chunk = per_wave >= RT_CHUNK ? RT_CHUNK : (per_wave >= 64 ? 64 : (per_wave >= 32 ? 32 : 16));

    // nodes and triangles are fetched through buffer resources (32-bit byte offsets: the host refuses scenes whose
    // arrays reach 4 GiB; a fetch past the end would read zeros instead of faulting)
    const __amdgpu_buffer_rsrc_t rs_nodes = __builtin_amdgcn_make_buffer_rsrc((void*)sc.nodes, (short)0, (int)((uint32_t)sc.n_nodes * 64u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_tris = __builtin_amdgcn_make_buffer_rsrc((void*)sc.tris, (short)0, (int)((uint32_t)sc.n_tris * 48u), 0x00020000);
    int exhausted = 0;         // wave-uniform: the work counter ran past count
    uint32_t steal = 0;        // wave-uniform: ranges found empty so far (the XCD's own ranges come first)
    uint32_t c_next = 0, c_end = 0;  // wave-uniform: the chunk of items this wave is handing out
    uint32_t item = 0;
    int32_t cur = RT_IDLE;
    int sp = 0;
    f3 org = mk3(0.f, 0.f, 0.f), dir = mk3(0.f, 0.f, 1.f), inv = mk3(0.f, 0.f, 1.f);
    HitRec h;
    h.t = inf; h.u = 0.f; h.v = 0.f; h.id0 = -1; h.id1 = -1;
    float tfar = inf;
    static_assert(!BOUNDED || ANY_HIT, "a far bound only makes sense for any-hit rays");
    static_assert(!MIXED || (!ANY_HIT && !BOUNDED), "MIXED takes the ray kind per item");
    bool any_lane = ANY_HIT;
    constexpr int leaf_min = (ANY_HIT || MIXED) ? RT_LEAF_MIN_ANY : RT_LEAF_MIN;

    for (;;) {
        // wave-uniform by construction; said again so that the votes compile to scalar branches
        c_next = __builtin_amdgcn_readfirstlane(c_next);
        c_end = __builtin_amdgcn_readfirstlane(c_end);
        exhausted = __builtin_amdgcn_readfirstlane(exhausted);
        steal = __builtin_amdgcn_readfirstlane(steal);
        const unsigned long long m_idle = __ballot(cur <= RT_DONE);
        const int n_idle = __popcll(m_idle);
        if (n_idle >= RT_REFILL_MIN && (c_next < c_end || !exhausted)) {
            RT_PROBE_ADD(0, 1); RT_PROBE_ADD(1, n_idle);
            // ---- refill idle lanes from the wave's chunk; one atomic per chunk keeps the shared counter
            // far from saturation and keeps a wave inside one contiguous run of rays
            uint32_t mine;
            const bool have = take_items<EXACT>(work, deferred, count, chunk, lane, cur <= RT_DONE, steal, c_next, c_end, exhausted, probe, mine);
            if (have) {
                if (cur == RT_DONE) done(item, h);
                cur = RT_IDLE;
                const uint32_t i = mine;
                item = i;
                bool present;
                if constexpr (MIXED) present = load(i, org, dir, tfar, any_lane);
                else if constexpr (BOUNDED) present = load(i, org, dir, tfar);
                else present = load(i, org, dir);
                inv = make_invdir(dir);
                h.t = inf; h.u = 0.f; h.v = 0.f; h.id0 = -1; h.id1 = -1;
                float rd;
                // root box test of src/scene/BVH.cpp:51-53
                if (!present) {
                } else if (slab_test(org, inv, sc.rmin[0], sc.rmin[1], sc.rmin[2], sc.rmax[0], sc.rmax[1], sc.rmax[2], rd)) {
                    const bool tame = EXACT || (sc.tame && moderate(org.x) && moderate(org.y) && moderate(org.z) &&
                                                moderate(inv.x) && moderate(inv.y) && moderate(inv.z));
                    if (tame) {
                        cur = sc.root_link;
                        sp = 0;
                    } else {
                        deferred[atomicAdd(&work->n_deferred, 1u)] = i;
                    }
                } else {
                    cur = RT_DONE;   // missed the root box: the (empty) result is still to be written
                }
            }
            continue;
        }
        if (n_idle == 64) break;   // nothing in flight and nothing left to fetch
        if (__popcll(__ballot(cur < 0)) - n_idle < leaf_min && __ballot(cur >= 0) != 0ull) {
            // ---- a few inner-node steps for the lanes standing on an inner node, unless enough lanes already
            // wait at a leaf
#pragma unroll 1
            for (int rep = 0; rep < RT_INNER_REPS; ++rep) {
                RT_PROBE_ADD(2, 1); RT_PROBE_ADD(3, __popcll(__ballot(cur >= 0))); RT_PROBE_ADD(6, __popcll(__ballot(cur >= 0 && cur >= top.n)));
                RT_PROBE_ADD(7, (__ballot(cur >= 0 && cur >= top.n) != 0ull) ? 1 : 0);
                if (cur >= 0) {
                    float4 a, b, c;
                    int32_t left, right;
                    if (cur < top.n) {
                        const tr_f32x4 la = top.b0[cur], lb = top.b1[cur], lc = top.b2[cur];
                        const tr_i32x2 l2 = top.lk[cur];
                        a = make_float4(la.x, la.y, la.z, la.w); b = make_float4(lb.x, lb.y, lb.z, lb.w); c = make_float4(lc.x, lc.y, lc.z, lc.w);
                        left = l2.x; right = l2.y;
                    } else {
                        // through a buffer resource: one 32-bit offset instead of a 64-bit address per lane
                        const uint32_t off = (uint32_t)cur << 6;   // < 4 GiB: checked by the host
                        a = ldb128(rs_nodes, off); b = ldb128(rs_nodes, off + 16); c = ldb128(rs_nodes, off + 32);
                        const u32x2 d = __builtin_amdgcn_raw_buffer_load_b64(rs_nodes, (int)(off + 48u), 0, 0);   // the other 8 bytes are padding
                        left = (int32_t)d.x; right = (int32_t)d.y;
                    }
#if RT_BPRIO
                    __builtin_amdgcn_s_setprio(0);
#endif
                    float ld, rdist;
                    bool ls, rs;
                    if (EXACT) {
                        ls = slab_test(org, inv, a.x, a.y, a.z, a.w, b.x, b.y, ld);
                        rs = slab_test(org, inv, b.z, b.w, c.x, c.y, c.z, c.w, rdist);
                    } else {
                        ls = slab_test_fast(org, inv, a.x, a.y, a.z, a.w, b.x, b.y, ld);
                        rs = slab_test_fast(org, inv, b.z, b.w, c.x, c.y, c.z, c.w, rdist);
                    }
                    const float bound = MIXED ? (any_lane ? tfar : h.t) : (BOUNDED ? tfar : h.t);
                    ls = ls && (ld < bound);
                    rs = rs && (rdist < bound);
                    // src/scene/BVH.cpp:160-168: both hit -> nearer child first (left on a tie), the other one is
                    // pushed; one hit -> that child; none -> back to the youngest pending child
                    // From here to the next node fetch the wave goes ahead of the waves that are doing arithmetic: the
                    // sooner its loads are out, the more of their latency the others cover (ray sets -1 ... -2 %).
#if RT_BPRIO
                    __builtin_amdgcn_s_setprio(3);
#endif
                    const bool swap = rdist < ld;
                    int32_t nxt = (rs && (swap || !ls)) ? right : left;
                    if (ls && rs) {
                        stk.push(sp, swap ? left : right);
                        ++sp;
                    }
                    if (!(ls || rs)) {
                        nxt = RT_DONE;
                        if (sp > 0) {
                            --sp;
                            nxt = stk.pop(sp);
                        }
                    }
                    cur = nxt;
                }
            }
        }
        // ---- one triangle test for the lanes standing on a leaf, once enough of them wait there or no lane
        // has an inner node left
        const bool at_leaf = cur < 0 && cur > RT_DONE;
        const int n_leaf = __popcll(__ballot(at_leaf));
        if (n_leaf >= leaf_min || (n_leaf > 0 && __ballot(cur >= 0) == 0ull)) {
            RT_PROBE_ADD(4, 1); RT_PROBE_ADD(5, n_leaf);
            if (at_leaf) {
                const uint32_t toff = (uint32_t)(~cur) * 48u;   // < 4 GiB: checked by the host
                float4 a = ldb128(rs_tris, toff), b = ldb128(rs_tris, toff + 16), c = ldb128(rs_tris, toff + 32);
                float t, u, v;
                bool finished = false;
                if (tri_test(org, dir, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), t, u, v)) {
                    // src/scene/BVH.cpp:139 (closest) / :93 (any)
                    const bool any_ray = MIXED ? any_lane : ANY_HIT;
                    const bool in_range = any_ray ? (!(BOUNDED || MIXED) || !(tfar < inf) || t < tfar) : (t < h.t);
                    if (t > 0.0001f && in_range) {
                        h.t = t; h.u = u; h.v = v;
                        h.id1 = __float_as_int(c.y);
                        h.id0 = __float_as_int(c.z);
                        if (any_ray) finished = true;
                    }
                }
                cur = RT_DONE;
                if (!finished && sp > 0) {
                    --sp;
                    cur = stk.pop(sp);
                }
            }
        }
    }
    if (cur == RT_DONE) done(item, h);
}

// Closest hit over a ray set (the batch form of BVH::intersect).  rays: 6
// planes of `stride` floats; outputs indexed like the rays.
template <bool EXACT>
__global__ __launch_bounds__(RT_TBLOCK RT_LB_WAVES) void k_trace_closest(TraceScene sc, const float* __restrict__ rays, uint32_t stride, uint32_t n,
                                                            TraceWork* __restrict__ work, uint32_t* __restrict__ deferred,
                                                            float* __restrict__ o_t, int32_t* __restrict__ o_id0,
                                                            int32_t* __restrict__ o_id1, float* __restrict__ o_u, float* __restrict__ o_v,
                                                            unsigned long long* __restrict__ ray_counter) {
    __shared__ int32_t s_stack[RT_STACK_LDS * RT_TBLOCK];
    __shared__ __attribute__((aligned(16))) char s_top[RT_TOP_BYTES];
    const TopTree top = stage_top_tree(sc.nodes, sc.n_nodes, s_top);
    if (!EXACT && ray_counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(ray_counter, (unsigned long long)n);
    const uint32_t count = EXACT ? work->n_deferred : n;
    if (EXACT && ray_counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(ray_counter + 2, (unsigned long long)count);
    const size_t st = stride;
    trace_persistent<false, EXACT>(
        sc, count, work, deferred, s_stack, top,
        [&](uint32_t i, f3& org, f3& dir) -> bool {
            org = mk3(rays[i], rays[st + i], rays[2 * st + i]);
            dir = mk3(rays[3 * st + i], rays[4 * st + i], rays[5 * st + i]);
            return true;
        },
        [&](uint32_t i, const HitRec& h) {
            o_t[i] = h.t; o_id0[i] = h.id0; o_id1[i] = h.id1; o_u[i] = h.u; o_v[i] = h.v;
        });
}

// The same traversal driven by a queue of path slots (one bounce of the
// wavefront tracer): entry i of `queue` names the slot whose ray is traced
// and whose hit record is written.
template <bool EXACT>
__global__ __launch_bounds__(RT_TBLOCK RT_LB_WAVES) void k_extend(TraceScene sc, const float* __restrict__ rays, uint32_t stride,
                                                     const uint32_t* __restrict__ queue, const uint32_t* __restrict__ d_count,
                                                     TraceWork* __restrict__ work, uint32_t* __restrict__ deferred,
                                                     float* __restrict__ o_t, int32_t* __restrict__ o_id0, int32_t* __restrict__ o_id1,
                                                     float* __restrict__ o_u, float* __restrict__ o_v,
                                                     unsigned long long* __restrict__ ray_counter) {
    __shared__ int32_t s_stack[RT_STACK_LDS * RT_TBLOCK];
    __shared__ __attribute__((aligned(16))) char s_top[RT_TOP_BYTES];
    const TopTree top = stage_top_tree(sc.nodes, sc.n_nodes, s_top);
    const uint32_t count = EXACT ? work->n_deferred : *d_count;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(ray_counter + (EXACT ? 2 : 0), (unsigned long long)count);
    const size_t st = stride;
    trace_persistent<false, EXACT>(
        sc, count, work, deferred, s_stack, top,
        [&](uint32_t i, f3& org, f3& dir) -> bool {
            uint32_t s = queue[i];
            org = mk3(rays[s], rays[st + s], rays[2 * st + s]);
            dir = mk3(rays[3 * st + s], rays[4 * st + s], rays[5 * st + s]);
            return true;
        },
        [&](uint32_t i, const HitRec& h) {
            uint32_t s = queue[i];
            o_t[s] = h.t; o_id0[s] = h.id0; o_id1[s] = h.id1; o_u[s] = h.u; o_v[s] = h.v;
        });
}

// Any hit over a ray set (the batch form of BVH::occluded); status[i] = 1
// when something is hit.
// Any hit in front of a per-ray far bound (rays plane 6 = tmax): the shadow
// rays the area lights send.
template <bool EXACT>
__global__ __launch_bounds__(RT_TBLOCK RT_LB_WAVES) void k_trace_occluded_bounded(TraceScene sc, const float* __restrict__ rays, const float* __restrict__ tmax,
                                                                     uint32_t stride, uint32_t n, TraceWork* __restrict__ work,
                                                                     uint32_t* __restrict__ deferred, int32_t* __restrict__ status) {
    __shared__ int32_t s_stack[RT_STACK_LDS * RT_TBLOCK];
    __shared__ __attribute__((aligned(16))) char s_top[RT_TOP_BYTES];
    const TopTree top = stage_top_tree(sc.nodes, sc.n_nodes, s_top);
    const uint32_t count = EXACT ? work->n_deferred : n;
    const size_t st = stride;
    trace_persistent<true, EXACT, true>(
        sc, count, work, deferred, s_stack, top,
        [&](uint32_t i, f3& org, f3& dir, float& tfar) -> bool {
            org = mk3(rays[i], rays[st + i], rays[2 * st + i]);
            dir = mk3(rays[3 * st + i], rays[4 * st + i], rays[5 * st + i]);
            tfar = tmax[i];
            return true;
        },
        [&](uint32_t i, const HitRec& h) { status[i] = (h.id1 >= 0) ? 1 : 0; });
}

template <bool EXACT>
__global__ __launch_bounds__(RT_TBLOCK RT_LB_WAVES) void k_trace_occluded(TraceScene sc, const float* __restrict__ rays, uint32_t stride, uint32_t n,
                                                             TraceWork* __restrict__ work, uint32_t* __restrict__ deferred,
                                                             int32_t* __restrict__ status) {
    __shared__ int32_t s_stack[RT_STACK_LDS * RT_TBLOCK];
    __shared__ __attribute__((aligned(16))) char s_top[RT_TOP_BYTES];
    const TopTree top = stage_top_tree(sc.nodes, sc.n_nodes, s_top);
    const uint32_t count = EXACT ? work->n_deferred : n;
    const size_t st = stride;
    trace_persistent<true, EXACT>(
        sc, count, work, deferred, s_stack, top,
        [&](uint32_t i, f3& org, f3& dir) -> bool {
            org = mk3(rays[i], rays[st + i], rays[2 * st + i]);
            dir = mk3(rays[3 * st + i], rays[4 * st + i], rays[5 * st + i]);
            return true;
        },
        [&](uint32_t i, const HitRec& h) { status[i] = (h.id1 >= 0) ? 1 : 0; });
}

}  // namespace rt
