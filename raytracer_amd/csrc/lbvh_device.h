// GPU builder of the reference's LBVH (see lbvh_device.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

namespace rt {

// The tree in the reference's node layout (src/scene/BBox.hpp:27-56), device arrays of 2T-1 entries.
struct LbvhRefArrays {
    float* boxes;     // 6 floats per node: min xyz, max xyz
    int32_t* left;
    int32_t* right;
    int32_t* tri;     // triIds[0]
    int32_t* model;   // modelIds[0]
    int32_t* leaf;
};

struct DeviceLbvh {
    LbvhRefArrays ref{};
    float4* nodes = nullptr;      // n_node_slots x 4 float4 (DevNode)
    float4* tris = nullptr;       // n_tris x 3 float4 (DevTri)
    size_t n_node_slots = 0;
    uint32_t n_tris = 0, n_ref_nodes = 0;
    int32_t root_link = 0, max_depth = 0;
    float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};
    std::vector<void*> owned;     // device allocations behind the pointers above
    void release();
};

// d_pos: packed xyz per vertex; d_idx: indices already offset; d_tri_model: model per triangle.
// Returns 0 on success; on failure *err holds the reason.
int build_lbvh_device(const float* d_pos, const uint32_t* d_idx, uint32_t n_indices, const int32_t* d_tri_model, uint32_t top_levels,
                      hipStream_t stream, DeviceLbvh* out, std::string* err);

}  // namespace rt
