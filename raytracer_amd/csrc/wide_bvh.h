// 4-wide collapse of the binary LBVH (see wide_bvh.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

namespace rt {

// Wide node, 128 bytes = 8 float4:
//   q0..q2 = min x / y / z of the four child boxes, q3..q5 = max x / y / z,
//   q6     = the four child links (int32: >= 0 wide node, < 0 ~triangle slot of the binary tree's triangle array),
//   q7     = (number of used slots, binary node the wide node was made from, 0, 0).
// An unused slot holds +inf in all six box planes and the link RT_WIDE_NONE: on an axis where the ray moves the
// slab interval of such a box is [+inf, +inf] or [-inf, -inf], so the box fails the test for every ray that has
// passed the root box (a ray with no moving axis has not).
#define RT_WIDE_NONE ((int32_t)0x80000000)

struct WideBvh {
    float4* nodes = nullptr;   // capacity x 8 float4, nodes in breadth-first order (root = 0)
    uint32_t n_nodes = 0;
    uint32_t capacity = 0;
    int32_t depth = 0;         // wide levels
    void release() { if (nodes) (void)hipFree(nodes); nodes = nullptr; n_nodes = 0; }
};

// bnodes: the binary device nodes (4 float4 each: left box, right box, left link, right link); root_link >= 0.
// n_binary_inner: number of inner nodes of the binary tree (T - 1), an upper bound of the wide node count.
// Returns 0 on success; on failure *err holds the reason.
int build_wide_bvh(const float4* bnodes, int32_t root_link, uint32_t n_binary_inner, hipStream_t stream, WideBvh* out, std::string* err);

}  // namespace rt
