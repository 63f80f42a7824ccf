// Host-side LBVH builder: the tree of the reference's BVH::runLBVH
// (src/scene/BVH.cpp:189-336) in the reference's node numbering, plus the
// compact device layout the traversal kernels read.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace rt {

// One node in the reference's layout (src/scene/BBox.hpp:27-56), keeping the
// fields the scalar path reads.
struct RefNode {
    float bmin[3], bmax[3];
    int32_t left, right;   // data.left / data.right (inner)
    int32_t tri, model;    // triIds[0] / modelIds[0] (leaf)
    int32_t leaf;
};

// Device inner node, 64 bytes: both child boxes live in the parent so one
// visit is one 64-byte fetch.  A link >= 0 is an inner-node index, a link
// < 0 is ~(triangle slot).
struct alignas(16) DevNode {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    int32_t left, right;
    int32_t pad[2];
};
static_assert(sizeof(DevNode) == 64, "DevNode must be 64 bytes");

// Device triangle, 48 bytes: the three positions Möller–Trumbore reads and
// the ids a hit reports (src/scene/BVH.cpp:139-143).
struct alignas(16) DevTri {
    float v0[3], v1[3], v2[3];
    int32_t id1;     // index-buffer offset of the triangle's first index
    int32_t id0;     // model index
    int32_t pad;
};
static_assert(sizeof(DevTri) == 48, "DevTri must be 48 bytes");

struct LbvhResult {
    std::vector<RefNode> ref;     // 2T-1 nodes, root = 0
    std::vector<DevNode> nodes;   // (2^top_levels - 1) + (T - 1) slots (empty when T == 1)
    std::vector<DevTri> tris;     // T triangles in leaf (Morton) order
    int32_t root_link = 0;        // link of the root (inner index or ~slot)
    float root_min[3], root_max[3];
    int32_t max_depth = 0;
};

// pos: strided vertex positions; indices already offset; tri_model per triangle.
// Inner nodes above level top_levels are stored at their heap index (the part
// the kernels stage in LDS), deeper ones after them in preorder.
void build_lbvh(const float* pos, uint32_t pos_stride_bytes, const uint32_t* indices, uint32_t n_indices,
                const int32_t* tri_model, LbvhResult& out, uint32_t top_levels);

uint32_t expand_bits(uint32_t v);
uint32_t morton3d(float x, float y, float z);

}  // namespace rt
