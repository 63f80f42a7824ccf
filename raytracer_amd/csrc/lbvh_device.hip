// GPU build of the reference's LBVH (BVH::runLBVH, src/scene/BVH.cpp:189-336):
// the same tree, node for node, as the host builder in lbvh_host.cpp, produced
// on the device so that multi-million-triangle scenes do not pay an
// O(N * depth) host pass.
//
//   k_prims    per triangle: box, centroid ((0+v0)+v1+v2)/3, scene bounds
//   k_morton   30-bit Morton code of the normalised centroid
//   sort       stable LSD radix sort of (code, triangle) pairs (rocPRIM via hipCUB):
//              ties keep ascending triangle order, like the host's stable sort
//   k_split    one launch per tree level: every node of the level finds its
//              split with the reference's binary search (findSplitLBVH) and
//              emits its two children.  Node ids are the reference's: a
//              node over sorted leaves [start, end] reached with `lt` left
//              turns has preorder rank start + lt among the inner nodes, and
//              its children are nodes 1 + 2*rank and 2 + 2*rank.
//   k_refit    one launch per level, deepest first: boxes as exact min/max
//              unions, and the 64-byte device nodes (both child boxes + links)
//   k_tris     device triangles in leaf order
//
// All arithmetic that decides the tree (centroid, normalisation, Morton code)
// is done in the reference's operation order without FMA contraction, so the
// sorted order and every split are bit-identical to the host build.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <string>
#include <vector>

#include "lbvh_device.h"

namespace rt {
namespace {

#define LB_BLOCK 256

__device__ __forceinline__ float sel_min(float x, float y) { return (y < x) ? y : x; }  // glm::min(x, y)
__device__ __forceinline__ float sel_max(float x, float y) { return (x < y) ? y : x; }  // glm::max(x, y)

// order-preserving float <-> uint map for atomic min/max
__device__ __forceinline__ uint32_t f2o(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float o2f(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(u);
#else
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}

// src/scene/BVH.cpp:299-314
__global__ __launch_bounds__(LB_BLOCK) void k_prims(const float* __restrict__ pos, const uint32_t* __restrict__ idx, uint32_t T,
                                                    float* __restrict__ pmin, float* __restrict__ pmax, float* __restrict__ cen,
                                                    uint32_t* __restrict__ bounds /* min xyz, max xyz, ordered-uint */) {
    __shared__ uint32_t s_b[6];
    if (threadIdx.x < 3) s_b[threadIdx.x] = 0xFFFFFFFFu;
    else if (threadIdx.x < 6) s_b[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t t = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (t < T) {
        const float inf = __builtin_inff();
        float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf}, c[3] = {0.f, 0.f, 0.f};
        for (int v = 0; v < 3; ++v) {
            const float* q = pos + (size_t)idx[3 * (size_t)t + v] * 3;
            for (int k = 0; k < 3; ++k) {
                float x = q[k];
                mn[k] = sel_min(mn[k], x);
                mx[k] = sel_max(mx[k], x);
                c[k] = c[k] + x;
            }
        }
        for (int k = 0; k < 3; ++k) {
            pmin[(size_t)3 * t + k] = mn[k];
            pmax[(size_t)3 * t + k] = mx[k];
            cen[(size_t)3 * t + k] = c[k] / 3.0f;
            atomicMin(&s_b[k], f2o(mn[k]));
            atomicMax(&s_b[3 + k], f2o(mx[k]));
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&bounds[threadIdx.x], s_b[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&bounds[threadIdx.x], s_b[threadIdx.x]);
}

// src/scene/BVH.cpp:191-212
__device__ __forceinline__ uint32_t expand_bits_d(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__global__ __launch_bounds__(LB_BLOCK) void k_morton(const float* __restrict__ cen, const uint32_t* __restrict__ bounds, uint32_t T,
                                                     uint32_t* __restrict__ codes, uint32_t* __restrict__ ids) {
    uint32_t t = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (t >= T) return;
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        float smin = o2f(bounds[k]), smax = o2f(bounds[3 + k]);
        float n = (cen[(size_t)3 * t + k] - smin) / (smax - smin);
        float s = sel_min(sel_max(n * 1024.f, 0.f), 1023.f);
        q[k] = (s == s) ? (uint32_t)s : 0u;  // NaN (flat axis) -> cell 0, as on the host
    }
    codes[t] = expand_bits_d(q[0]) * 4 + expand_bits_d(q[1]) * 2 + expand_bits_d(q[2]);
    ids[t] = t;
}

struct WorkItem {
    uint32_t node;     // reference node id
    uint32_t start, end;
    uint32_t lturns;
    uint32_t heap;     // heap index while depth < top_levels
};

__device__ __forceinline__ int clz32(uint32_t x) { return x ? __clz((int)x) : 32; }

// One tree level.  src/scene/BVH.cpp:214-285 (split search and child creation).
__global__ __launch_bounds__(LB_BLOCK) void k_split(const WorkItem* __restrict__ in, uint32_t n_in, WorkItem* __restrict__ out,
                                                    uint32_t* __restrict__ n_out, const uint32_t* __restrict__ codes,
                                                    const uint32_t* __restrict__ ids, const int32_t* __restrict__ tri_model,
                                                    const float* __restrict__ pmin, const float* __restrict__ pmax, uint32_t depth,
                                                    uint32_t top_levels, uint32_t top_slots, LbvhRefArrays ref, int32_t* __restrict__ link) {
    uint32_t i = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (i >= n_in) return;
    WorkItem w = in[i];
    if (w.start == w.end) {
        uint32_t t = ids[w.start];
        ref.leaf[w.node] = 1;
        ref.left[w.node] = -1;
        ref.right[w.node] = -1;
        ref.tri[w.node] = (int32_t)(3 * t);
        ref.model[w.node] = tri_model[t];
        for (int k = 0; k < 3; ++k) {
            ref.boxes[(size_t)6 * w.node + k] = pmin[(size_t)3 * t + k];
            ref.boxes[(size_t)6 * w.node + 3 + k] = pmax[(size_t)3 * t + k];
        }
        link[w.node] = ~(int32_t)w.start;
        return;
    }
    // findSplitLBVH
    uint32_t a = codes[w.start], b = codes[w.end], split;
    if (a == b) {
        split = (w.start + w.end) >> 1;
    } else {
        int common = clz32(a ^ b);
        split = w.start;
        uint32_t step = w.end - w.start;
        do {
            step = (step + 1) >> 1;
            uint32_t cand = split + step;
            if (cand < w.end && clz32(a ^ codes[cand]) > common) split = cand;
        } while (step > 1);
    }
    const uint32_t rank = w.start + w.lturns;
    const uint32_t l = 1 + 2 * rank;
    ref.leaf[w.node] = 0;
    ref.left[w.node] = (int32_t)l;
    ref.right[w.node] = (int32_t)(l + 1);
    ref.tri[w.node] = -1;
    ref.model[w.node] = -1;
    link[w.node] = (int32_t)(depth < top_levels ? w.heap : top_slots + rank);
    uint32_t o = atomicAdd(n_out, 2u);
    WorkItem cl = {l, w.start, split, w.lturns + 1, 2 * w.heap + 1};
    WorkItem cr = {l + 1, split + 1, w.end, w.lturns, 2 * w.heap + 2};
    out[o] = cl;
    out[o + 1] = cr;
}

// Boxes of one level's inner nodes from their children (already final), and
// the device nodes.  Union order as on the host: min(right, left).
__global__ __launch_bounds__(LB_BLOCK) void k_refit(const WorkItem* __restrict__ items, uint32_t n_items, LbvhRefArrays ref,
                                                    const int32_t* __restrict__ link, float4* __restrict__ nodes) {
    uint32_t i = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (i >= n_items) return;
    const uint32_t node = items[i].node;
    if (ref.leaf[node]) return;
    const int32_t l = ref.left[node], r = ref.right[node];
    float a[6], b[6];
    for (int k = 0; k < 6; ++k) { a[k] = ref.boxes[(size_t)6 * l + k]; b[k] = ref.boxes[(size_t)6 * r + k]; }
    for (int k = 0; k < 3; ++k) {
        ref.boxes[(size_t)6 * node + k] = sel_min(b[k], a[k]);
        ref.boxes[(size_t)6 * node + 3 + k] = sel_max(b[3 + k], a[3 + k]);
    }
    float4* d = nodes + (size_t)link[node] * 4;
    d[0] = make_float4(a[0], a[1], a[2], a[3]);
    d[1] = make_float4(a[4], a[5], b[0], b[1]);
    d[2] = make_float4(b[2], b[3], b[4], b[5]);
    d[3] = make_float4(__int_as_float(link[l]), __int_as_float(link[r]), 0.f, 0.f);
}

__global__ __launch_bounds__(LB_BLOCK) void k_tris(const float* __restrict__ pos, const uint32_t* __restrict__ idx,
                                                   const uint32_t* __restrict__ ids, const int32_t* __restrict__ tri_model, uint32_t T,
                                                   float4* __restrict__ tris) {
    uint32_t s = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (s >= T) return;
    uint32_t t = ids[s];
    const float* a = pos + (size_t)idx[3 * (size_t)t] * 3;
    const float* b = pos + (size_t)idx[3 * (size_t)t + 1] * 3;
    const float* c = pos + (size_t)idx[3 * (size_t)t + 2] * 3;
    float4* d = tris + (size_t)s * 3;
    d[0] = make_float4(a[0], a[1], a[2], b[0]);
    d[1] = make_float4(b[1], b[2], c[0], c[1]);
    d[2] = make_float4(c[2], __int_as_float((int32_t)(3 * t)), __int_as_float(tri_model[t]), 0.f);
}

#define LB_TRY(call)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            if (err) *err = std::string(#call) + " failed: " + hipGetErrorString(e_);                        \
            return -1;                                                                                       \
        }                                                                                                    \
    } while (0)

template <typename T>
hipError_t dmalloc(std::vector<void*>& owned, T** p, size_t count) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
    if (e == hipSuccess) { owned.push_back(q); *p = static_cast<T*>(q); }
    return e;
}

}  // namespace

void DeviceLbvh::release() {
    for (void* p : owned) (void)hipFree(p);
    owned.clear();
}

int build_lbvh_device(const float* d_pos, const uint32_t* d_idx, uint32_t n_indices, const int32_t* d_tri_model, uint32_t top_levels,
                      hipStream_t stream, DeviceLbvh* out, std::string* err) {
    const uint32_t T = n_indices / 3;
    const size_t NN = 2 * (size_t)T - 1;
    const uint32_t top_slots = (1u << top_levels) - 1;
    out->n_tris = T;
    out->n_ref_nodes = (uint32_t)NN;
    out->n_node_slots = T > 1 ? top_slots + (T - 1) : 0;
    // scratch freed on every way out (the error returns of LB_TRY included)
    struct Scratch : std::vector<void*> {
        hipStream_t stream;
        explicit Scratch(hipStream_t s) : stream(s) {}
        void release() {
            if (!empty()) (void)hipStreamSynchronize(stream);
            for (void* p : *this) (void)hipFree(p);
            clear();
        }
        ~Scratch() { release(); }
    } tmp(stream);
    auto free_tmp = [&]() { tmp.release(); };

    float *pmin, *pmax, *cen;
    uint32_t *bounds, *codes, *ids, *codes_s, *ids_s, *counter;
    WorkItem* items;
    int32_t* link;
    LB_TRY(dmalloc(tmp, &pmin, (size_t)3 * T));
    LB_TRY(dmalloc(tmp, &pmax, (size_t)3 * T));
    LB_TRY(dmalloc(tmp, &cen, (size_t)3 * T));
    LB_TRY(dmalloc(tmp, &bounds, 6));
    LB_TRY(dmalloc(tmp, &codes, T));
    LB_TRY(dmalloc(tmp, &ids, T));
    LB_TRY(dmalloc(tmp, &codes_s, T));
    LB_TRY(dmalloc(tmp, &ids_s, T));
    LB_TRY(dmalloc(tmp, &counter, 1));
    LB_TRY(dmalloc(tmp, &items, NN));  // every node appears once, level after level
    LB_TRY(dmalloc(tmp, &link, NN));
    LB_TRY(dmalloc(out->owned, &out->ref.boxes, 6 * NN));
    LB_TRY(dmalloc(out->owned, &out->ref.left, NN));
    LB_TRY(dmalloc(out->owned, &out->ref.right, NN));
    LB_TRY(dmalloc(out->owned, &out->ref.tri, NN));
    LB_TRY(dmalloc(out->owned, &out->ref.model, NN));
    LB_TRY(dmalloc(out->owned, &out->ref.leaf, NN));
    LB_TRY(dmalloc(out->owned, &out->nodes, out->n_node_slots * 4));
    LB_TRY(dmalloc(out->owned, &out->tris, (size_t)T * 3));
    if (out->n_node_slots) LB_TRY(hipMemsetAsync(out->nodes, 0, out->n_node_slots * 64, stream));

    const uint32_t gT = (T + LB_BLOCK - 1) / LB_BLOCK;
    const uint32_t init_bounds[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    LB_TRY(hipMemcpyAsync(bounds, init_bounds, sizeof init_bounds, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_prims, dim3(gT), dim3(LB_BLOCK), 0, stream, d_pos, d_idx, T, pmin, pmax, cen, bounds);
    hipLaunchKernelGGL(k_morton, dim3(gT), dim3(LB_BLOCK), 0, stream, cen, bounds, T, codes, ids);
    {
        size_t bytes = 0;
        LB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, codes, codes_s, ids, ids_s, (int)T, 0, 30, stream));
        void* scratch = nullptr;
        LB_TRY(dmalloc(tmp, reinterpret_cast<char**>(&scratch), bytes));
        LB_TRY(hipcub::DeviceRadixSort::SortPairs(scratch, bytes, codes, codes_s, ids, ids_s, (int)T, 0, 30, stream));
    }
    hipLaunchKernelGGL(k_tris, dim3(gT), dim3(LB_BLOCK), 0, stream, d_pos, d_idx, ids_s, d_tri_model, T, out->tris);

    // top-down, one launch per level; the items of level d occupy items[off[d], off[d+1])
    std::vector<size_t> off{0, 1};
    const WorkItem root = {0, 0, T - 1, 0, 0};
    LB_TRY(hipMemcpyAsync(items, &root, sizeof root, hipMemcpyHostToDevice, stream));
    uint32_t depth = 0;
    for (;;) {
        const uint32_t n_in = (uint32_t)(off[depth + 1] - off[depth]);
        LB_TRY(hipMemsetAsync(counter, 0, sizeof(uint32_t), stream));
        hipLaunchKernelGGL(k_split, dim3((n_in + LB_BLOCK - 1) / LB_BLOCK), dim3(LB_BLOCK), 0, stream, items + off[depth], n_in,
                           items + off[depth + 1], counter, codes_s, ids_s, d_tri_model, pmin, pmax, depth, top_levels, top_slots,
                           out->ref, link);
        uint32_t n_out = 0;
        LB_TRY(hipMemcpyAsync(&n_out, counter, sizeof n_out, hipMemcpyDeviceToHost, stream));
        LB_TRY(hipStreamSynchronize(stream));
        if (n_out == 0) break;
        if (off[depth + 1] + n_out > NN) {
            if (err) *err = "LBVH build produced more nodes than 2T-1";
            free_tmp();
            return -1;
        }
        off.push_back(off[depth + 1] + n_out);
        ++depth;
    }
    out->max_depth = (int32_t)depth;
    for (int d = (int)depth; d >= 0; --d) {
        const uint32_t n_items = (uint32_t)(off[d + 1] - off[d]);
        hipLaunchKernelGGL(k_refit, dim3((n_items + LB_BLOCK - 1) / LB_BLOCK), dim3(LB_BLOCK), 0, stream, items + off[d], n_items,
                           out->ref, link, out->nodes);
    }
    float rootbox[6];
    int32_t rl = 0;
    LB_TRY(hipMemcpyAsync(rootbox, out->ref.boxes, sizeof rootbox, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(&rl, link, sizeof rl, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipStreamSynchronize(stream));
    LB_TRY(hipGetLastError());
    for (int k = 0; k < 3; ++k) { out->root_min[k] = rootbox[k]; out->root_max[k] = rootbox[3 + k]; }
    out->root_link = rl;
    free_tmp();
    return 0;
}

}  // namespace rt
