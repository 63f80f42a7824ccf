// 4-wide collapse of the binary LBVH, on the device.
//
// The binary tree is the reference's (BVH::runLBVH, src/scene/BVH.cpp:287-336).  A wide node is made from a binary
// inner node P: it starts as P's two children and, while it has fewer than four slots and one of them is an inner
// node, the inner slot with the largest box surface is replaced by its two children (slots keep the tree's
// left-to-right order).  Every slot is therefore a node of the binary tree with that node's own box, and every
// triangle hangs below exactly one slot: a ray reaches a triangle through wide nodes iff it passes the slab test of a
// subset of the boxes the scalar recursion tests on the way to it (src/scene/BVH.cpp:148-158).  A box lies inside
// the boxes of all its ancestors and the slab arithmetic is monotonic in the box coordinates, so passing the test of
// a node implies passing the tests of the ancestors that were skipped: both traversals reach the same triangles.
//
// Built level by level, breadth first (the nodes of a level are contiguous, the root is node 0, so a prefix of
// the array is the top of the tree): per level one launch expands the level's binary nodes into wide nodes and
// counts their inner slots, a scan turns the counts into offsets, a second launch writes the child links and the
// next level's work list.  The numbering is deterministic (no atomics).
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "wide_bvh.h"

namespace rt {
namespace {

#define WB_BLOCK 256
#define WB_SCAN 1024

struct Slot {
    float b[6];     // min xyz, max xyz
    int32_t link;   // binary link: >= 0 inner node, < 0 ~triangle slot
};

__device__ __forceinline__ float box_area(const float* b) {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
}

__device__ __forceinline__ void load_children(const float4* __restrict__ bnodes, int32_t node, Slot& l, Slot& r) {
    const float4* p = bnodes + (size_t)node * 4;
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    l.b[0] = a.x; l.b[1] = a.y; l.b[2] = a.z; l.b[3] = a.w; l.b[4] = b.x; l.b[5] = b.y;
    r.b[0] = b.z; r.b[1] = b.w; r.b[2] = c.x; r.b[3] = c.y; r.b[4] = c.z; r.b[5] = c.w;
    l.link = __float_as_int(d.x);
    r.link = __float_as_int(d.y);
}

// One level: work item i = binary inner node in[i] becomes wide node base + i.  The links written here are still
// binary links; counts[i] = number of inner slots.
__global__ __launch_bounds__(WB_BLOCK) void k_wide_expand(const float4* __restrict__ bnodes, const int32_t* __restrict__ in, uint32_t n_in,
                                                         uint32_t base, float4* __restrict__ wnodes, uint32_t* __restrict__ counts) {
    const uint32_t i = blockIdx.x * WB_BLOCK + threadIdx.x;
    if (i >= n_in) return;
    const int32_t node = in[i];
    Slot s[4];
    int n = 2;
    load_children(bnodes, node, s[0], s[1]);
    for (int round = 0; round < 2; ++round) {
        int best = -1;
        float best_area = -1.0f;
        for (int k = 0; k < n; ++k)
            if (s[k].link >= 0) {
                const float a = box_area(s[k].b);
                if (best < 0 || a > best_area) { best = k; best_area = a; }
            }
        if (best < 0) break;
        for (int k = n - 1; k > best; --k) s[k + 1] = s[k];
        Slot l, r;
        load_children(bnodes, s[best].link, l, r);
        s[best] = l;
        s[best + 1] = r;
        ++n;
    }
    const float pinf = __builtin_inff();
    for (int k = n; k < 4; ++k) {
        for (int c = 0; c < 6; ++c) s[k].b[c] = pinf;
        s[k].link = RT_WIDE_NONE;
    }
    float4* w = wnodes + (size_t)(base + i) * 8;
    for (int c = 0; c < 6; ++c) w[c] = make_float4(s[0].b[c], s[1].b[c], s[2].b[c], s[3].b[c]);
    w[6] = make_float4(__int_as_float(s[0].link), __int_as_float(s[1].link), __int_as_float(s[2].link), __int_as_float(s[3].link));
    w[7] = make_float4(__int_as_float(n), __int_as_float(node), 0.f, 0.f);
    uint32_t inner = 0;
    for (int k = 0; k < n; ++k) inner += s[k].link >= 0 ? 1u : 0u;
    counts[i] = inner;
}

// Exclusive prefix sum of counts[0..n) by one workgroup; *total = the sum.
__global__ __launch_bounds__(WB_SCAN) void k_wide_scan(const uint32_t* __restrict__ counts, uint32_t n, uint32_t* __restrict__ offsets,
                                                      uint32_t* __restrict__ total) {
    __shared__ uint32_t s_wave[WB_SCAN / 64];
    __shared__ uint32_t s_carry;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t start = 0; start < n; start += WB_SCAN) {
        const uint32_t i = start + threadIdx.x;
        const uint32_t v = i < n ? counts[i] : 0u;
        uint32_t x = v;   // inclusive scan inside the wave
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d);
            if ((int)lane >= d) x += y;
        }
        if (lane == 63) s_wave[wave] = x;
        __syncthreads();
        uint32_t before = s_carry;
        for (uint32_t w = 0; w < wave; ++w) before += s_wave[w];
        if (i < n) offsets[i] = before + x - v;
        __syncthreads();
        if (threadIdx.x == WB_SCAN - 1) s_carry = before + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}

// Second launch of a level: inner slot number r of work item i is wide node next_base + offsets[i] + r, and its
// binary node goes to the next level's work list at that position.
__global__ __launch_bounds__(WB_BLOCK) void k_wide_link(uint32_t n_in, uint32_t base, uint32_t next_base, const uint32_t* __restrict__ offsets,
                                                       float4* __restrict__ wnodes, int32_t* __restrict__ next) {
    const uint32_t i = blockIdx.x * WB_BLOCK + threadIdx.x;
    if (i >= n_in) return;
    float4* w = wnodes + (size_t)(base + i) * 8;
    const float4 q = w[6];
    int32_t l[4] = {__float_as_int(q.x), __float_as_int(q.y), __float_as_int(q.z), __float_as_int(q.w)};
    uint32_t pos = offsets[i];
    for (int k = 0; k < 4; ++k)
        if (l[k] >= 0) {
            next[pos] = l[k];
            l[k] = (int32_t)(next_base + pos);
            ++pos;
        }
    w[6] = make_float4(__int_as_float(l[0]), __int_as_float(l[1]), __int_as_float(l[2]), __int_as_float(l[3]));
}

#define WB_TRY(call)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            if (err) *err = std::string(#call) + " failed: " + hipGetErrorString(e_);                        \
            return -1;                                                                                       \
        }                                                                                                    \
    } while (0)

}  // namespace

int build_wide_bvh(const float4* bnodes, int32_t root_link, uint32_t n_binary_inner, hipStream_t stream, WideBvh* out, std::string* err) {
    out->n_nodes = 0;
    out->depth = 0;
    if (root_link < 0 || n_binary_inner == 0) return 0;   // a single triangle: no inner node at all
    struct Scratch : std::vector<void*> {
        hipStream_t stream;
        explicit Scratch(hipStream_t s) : stream(s) {}
        ~Scratch() {
            if (!empty()) (void)hipStreamSynchronize(stream);
            for (void* p : *this) (void)hipFree(p);
        }
    } tmp(stream);
    auto dmalloc = [&](void** p, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(p, bytes ? bytes : 1);
        if (e == hipSuccess) tmp.push_back(*p);
        return e;
    };
    const uint32_t cap = n_binary_inner;
    int32_t *q0 = nullptr, *q1 = nullptr;
    uint32_t *counts = nullptr, *offsets = nullptr, *total = nullptr;
    WB_TRY(dmalloc((void**)&q0, (size_t)cap * 4));
    WB_TRY(dmalloc((void**)&q1, (size_t)cap * 4));
    WB_TRY(dmalloc((void**)&counts, (size_t)cap * 4));
    WB_TRY(dmalloc((void**)&offsets, (size_t)cap * 4));
    WB_TRY(dmalloc((void**)&total, 4));
    void* nodes = nullptr;
    WB_TRY(hipMalloc(&nodes, (size_t)cap * 128));
    out->nodes = static_cast<float4*>(nodes);
    out->capacity = cap;
    WB_TRY(hipMemcpyAsync(q0, &root_link, 4, hipMemcpyHostToDevice, stream));
    uint32_t base = 0, n_in = 1;
    int32_t depth = 0;
    while (n_in) {
        if (base + n_in > cap) {
            if (err) *err = "wide collapse produced more nodes than the binary tree has inner nodes";
            return -1;
        }
        const uint32_t g = (n_in + WB_BLOCK - 1) / WB_BLOCK;
        hipLaunchKernelGGL(k_wide_expand, dim3(g), dim3(WB_BLOCK), 0, stream, bnodes, q0, n_in, base, out->nodes, counts);
        hipLaunchKernelGGL(k_wide_scan, dim3(1), dim3(WB_SCAN), 0, stream, counts, n_in, offsets, total);
        hipLaunchKernelGGL(k_wide_link, dim3(g), dim3(WB_BLOCK), 0, stream, n_in, base, base + n_in, offsets, out->nodes, q1);
        uint32_t n_next = 0;
        WB_TRY(hipMemcpyAsync(&n_next, total, 4, hipMemcpyDeviceToHost, stream));
        WB_TRY(hipStreamSynchronize(stream));
        WB_TRY(hipGetLastError());
        base += n_in;
        n_in = n_next;
        std::swap(q0, q1);
        ++depth;
    }
    out->n_nodes = base;
    out->depth = depth;
    return 0;
}

}  // namespace rt
