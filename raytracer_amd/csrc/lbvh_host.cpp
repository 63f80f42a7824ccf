// Host LBVH build reproducing the tree of BVH::runLBVH
// (src/scene/BVH.cpp:287-336): 30-bit Morton codes of the normalised triangle
// centroids, sort by code, top-down split at the highest differing bit, one
// triangle per leaf, children appended as a (left, right) pair before either
// subtree is built.  Compile with -ffp-contract=off: the centroid and the
// normalisation must round exactly as the scalar reference expressions do.
#include "lbvh_host.h"

#include <algorithm>
#include <limits>
#include <numeric>

namespace rt {

// src/scene/BVH.cpp:191-198
uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

static inline float sel_min(float x, float y) { return (y < x) ? y : x; }  // glm::min(x, y)
static inline float sel_max(float x, float y) { return (x < y) ? y : x; }  // glm::max(x, y)

// src/scene/BVH.cpp:202-212.  A NaN coordinate (an axis on which the scene
// is flat gives 0/0) maps to cell 0.
uint32_t morton3d(float x, float y, float z) {
    float c[3] = {x, y, z};
    uint32_t q[3];
    for (int i = 0; i < 3; ++i) {
        float s = sel_min(sel_max(c[i] * 1024.f, 0.f), 1023.f);
        q[i] = (s == s) ? (uint32_t)s : 0u;
    }
    return expand_bits(q[0]) * 4 + expand_bits(q[1]) * 2 + expand_bits(q[2]);
}

namespace {

struct Prim {
    float mn[3], mx[3];
    int32_t id, model;
    uint32_t code;
};

inline int clz32(uint32_t x) { return x ? __builtin_clz(x) : 32; }

// src/scene/BVH.cpp:214-240
uint32_t find_split(const std::vector<Prim>& p, uint32_t start, uint32_t end) {
    uint32_t a = p[start].code, b = p[end].code;
    if (a == b) return (start + end) >> 1;
    int common = clz32(a ^ b);
    uint32_t split = start, step = end - start;
    do {
        step = (step + 1) >> 1;
        uint32_t cand = split + step;
        if (cand < end && clz32(a ^ p[cand].code) > common) split = cand;
    } while (step > 1);
    return split;
}

}  // namespace

void build_lbvh(const float* pos, uint32_t stride, const uint32_t* indices, uint32_t n_indices,
                const int32_t* tri_model, LbvhResult& out, uint32_t top_levels) {
    const uint32_t T = n_indices / 3;
    auto P = [&](uint32_t vi) { return reinterpret_cast<const float*>(reinterpret_cast<const char*>(pos) + (size_t)vi * stride); };
    const float inf = std::numeric_limits<float>::infinity();

    std::vector<Prim> prims(T);
    std::vector<float> cen((size_t)T * 3);
    float smin[3] = {inf, inf, inf}, smax[3] = {-inf, -inf, -inf};
    for (uint32_t t = 0; t < T; ++t) {
        Prim& pr = prims[t];
        float c[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < 3; ++k) { pr.mn[k] = inf; pr.mx[k] = -inf; }
        for (int v = 0; v < 3; ++v) {
            const float* q = P(indices[3 * t + v]);
            for (int k = 0; k < 3; ++k) {
                pr.mn[k] = sel_min(pr.mn[k], q[k]);
                pr.mx[k] = sel_max(pr.mx[k], q[k]);
                c[k] = c[k] + q[k];
            }
        }
        pr.id = (int32_t)(3 * t);
        pr.model = tri_model[t];
        for (int k = 0; k < 3; ++k) {
            smin[k] = sel_min(pr.mn[k], smin[k]);
            smax[k] = sel_max(pr.mx[k], smax[k]);
            cen[(size_t)3 * t + k] = c[k] / 3.0f;
        }
    }
    for (uint32_t t = 0; t < T; ++t) {
        float n[3];
        for (int k = 0; k < 3; ++k) n[k] = (cen[(size_t)3 * t + k] - smin[k]) / (smax[k] - smin[k]);
        prims[t].code = morton3d(n[0], n[1], n[2]);
    }
    // The reference's std::sort leaves the order of equal codes open; ties
    // keep input order here (ascending Prim.id).
    std::stable_sort(prims.begin(), prims.end(), [](const Prim& a, const Prim& b) { return a.code < b.code; });

    // Topology in the reference's numbering: DFS with the child pair
    // allocated when the parent is expanded, left subtree first.
    std::vector<RefNode>& ref = out.ref;
    ref.clear();
    ref.reserve(T ? 2 * (size_t)T - 1 : 1);
    struct Item { uint32_t node, start, end; int32_t depth; };
    std::vector<Item> stack;
    std::vector<uint32_t> leaf_slot(T ? 2 * (size_t)T - 1 : 1, 0);
    std::vector<uint32_t> node_start(T ? 2 * (size_t)T - 1 : 1, 0);  // first sorted leaf under each node
    RefNode blank{};
    blank.left = blank.right = blank.tri = blank.model = -1;
    ref.push_back(blank);
    out.max_depth = 0;
    if (T) stack.push_back({0, 0, T - 1, 0});
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        if (it.depth > out.max_depth) out.max_depth = it.depth;
        node_start[it.node] = it.start;
        if (it.start == it.end) {
            RefNode& n = ref[it.node];
            n.leaf = 1;
            n.tri = prims[it.start].id;
            n.model = prims[it.start].model;
            leaf_slot[it.node] = it.start;
            continue;
        }
        uint32_t split = find_split(prims, it.start, it.end);
        uint32_t l = (uint32_t)ref.size();
        ref.push_back(blank);
        ref.push_back(blank);
        ref[it.node].left = (int32_t)l;
        ref[it.node].right = (int32_t)(l + 1);
        stack.push_back({l + 1, split + 1, it.end, it.depth + 1});
        stack.push_back({l, it.start, split, it.depth + 1});
    }
    // Boxes bottom-up: children always have larger indices than their parent,
    // and min/max unions are exact, so this equals the per-range recomputation
    // of src/scene/BVH.cpp:247-252.
    for (size_t i = ref.size(); i-- > 0;) {
        RefNode& n = ref[i];
        if (n.leaf) {
            const Prim& p = prims[leaf_slot[i]];
            for (int k = 0; k < 3; ++k) { n.bmin[k] = p.mn[k]; n.bmax[k] = p.mx[k]; }
        } else if (n.left >= 0) {
            const RefNode &a = ref[n.left], &b = ref[n.right];
            for (int k = 0; k < 3; ++k) { n.bmin[k] = sel_min(b.bmin[k], a.bmin[k]); n.bmax[k] = sel_max(b.bmax[k], a.bmax[k]); }
        } else {
            for (int k = 0; k < 3; ++k) { n.bmin[k] = inf; n.bmax[k] = -inf; }
        }
    }
    for (int k = 0; k < 3; ++k) { out.root_min[k] = ref[0].bmin[k]; out.root_max[k] = ref[0].bmax[k]; }

    // Device layout.  Triangles in leaf (sorted) order.  Inner nodes: a node
    // above level top_levels sits at its heap index (root 0, children 2h+1 and
    // 2h+2: the block the traversal kernels keep in LDS; slots of absent nodes
    // stay unused), a deeper node at (2^top_levels - 1) + its preorder rank
    // among the inner nodes, so that a node's left child follows it directly
    // (parent and near child often share a 128-byte line).  The preorder rank
    // of the node over sorted leaves [start, end] is start + the number of
    // left turns on the path from the root.
    out.tris.resize(T);
    for (uint32_t s = 0; s < T; ++s) {
        DevTri& d = out.tris[s];
        const Prim& p = prims[s];
        const float* a = P(indices[p.id]);
        const float* b = P(indices[p.id + 1]);
        const float* c = P(indices[p.id + 2]);
        for (int k = 0; k < 3; ++k) { d.v0[k] = a[k]; d.v1[k] = b[k]; d.v2[k] = c[k]; }
        d.id1 = p.id;
        d.id0 = p.model;
        d.pad = 0;
    }
    const size_t top_slots = ((size_t)1 << top_levels) - 1;
    std::vector<int32_t> link(ref.size(), 0);
    {
        struct W { uint32_t node, start, lturns, depth; size_t heap; };
        std::vector<W> st;
        if (T) st.push_back({0, 0, 0, 0, 0});
        while (!st.empty()) {
            W w = st.back();
            st.pop_back();
            const RefNode& n = ref[w.node];
            if (n.leaf) {
                link[w.node] = ~(int32_t)leaf_slot[w.node];
                continue;
            }
            const uint32_t rank = w.start + w.lturns;
            link[w.node] = (int32_t)(w.depth < top_levels ? w.heap : top_slots + rank);
            st.push_back({(uint32_t)n.right, node_start[n.right], w.lturns, w.depth + 1, 2 * w.heap + 2});
            st.push_back({(uint32_t)n.left, w.start, w.lturns + 1, w.depth + 1, 2 * w.heap + 1});
        }
    }
    out.nodes.assign(T > 1 ? top_slots + (T - 1) : 0, DevNode{});
    for (size_t i = 0; i < ref.size(); ++i) {
        if (ref[i].leaf || ref[i].left < 0) continue;
        DevNode& d = out.nodes[link[i]];
        const RefNode &a = ref[ref[i].left], &b = ref[ref[i].right];
        for (int k = 0; k < 3; ++k) { d.lmin[k] = a.bmin[k]; d.lmax[k] = a.bmax[k]; d.rmin[k] = b.bmin[k]; d.rmax[k] = b.bmax[k]; }
        d.left = link[ref[i].left];
        d.right = link[ref[i].right];
        d.pad[0] = d.pad[1] = 0;
    }
    out.root_link = T ? link[0] : 0;
}

}  // namespace rt
