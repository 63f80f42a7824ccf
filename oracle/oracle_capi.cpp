// C entry points over oracle.hpp for ctypes.  TEST INFRASTRUCTURE ONLY: the
// checker and the timed CPU baseline; never linked into the product.
#include "oracle.hpp"

using namespace orc;

extern "C" {

struct orc_light_desc {
    int32_t type;           // 0 point, 3 env
    float pos[3];
    float intensity[3];     // point intensity / env scale
    const float* env_rgba;  // env: W*H*4 floats
    int32_t env_w, env_h;
    float p0[3], p1[3], p2[3];  // area (type 2): the emissive triangle, radiance = intensity
};

struct orc_texture_desc {
    int32_t width, height;
    const uint8_t* rgba;   // level 0, width*height*4 bytes
};

struct orc_scene {
    Scene s;
};

// Arrays are packed: pos/normal/tangent/binormal 3 floats per vertex, uv 2.
// tangent/binormal may be null (then a frame is generated from the normal).
orc_scene* orc_create(uint32_t n_vertices, const float* pos, const float* normal, const float* uv,
                      const float* tangent, const float* binormal, uint32_t n_indices,
                      const uint32_t* indices, const int32_t* tri_model, uint32_t n_models,
                      const float* materials /* rgb + roughness per model */, uint32_t n_lights,
                      const orc_light_desc* lights, const int32_t* material_texture /* per model or null */,
                      uint32_t n_textures, const orc_texture_desc* textures,
                      const int32_t* material_glossy /* per model or null: 0 none, 1 Beckmann, 2 GGX */) {
    auto* o = new orc_scene();
    std::vector<Vertex> verts(n_vertices);
    for (uint32_t i = 0; i < n_vertices; ++i) {
        Vertex& v = verts[i];
        v.pos = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
        if (normal) v.normal = {normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]};
        if (uv) v.uv = {uv[2 * i], uv[2 * i + 1]};
        if (tangent) v.tangent = {tangent[3 * i], tangent[3 * i + 1], tangent[3 * i + 2]};
        if (binormal) v.binormal = {binormal[3 * i], binormal[3 * i + 1], binormal[3 * i + 2]};
    }
    o->s.haveFrame = tangent && binormal;
    std::vector<uint32_t> idx(indices, indices + n_indices);
    std::vector<int> tm(tri_model, tri_model + n_indices / 3);
    o->s.bvh.runLBVH(verts, idx, tm);
    o->s.materials.resize(n_models);
    for (uint32_t m = 0; m < n_models; ++m) {
        o->s.materials[m].diffuseColor = {materials[4 * m], materials[4 * m + 1], materials[4 * m + 2]};
        o->s.materials[m].roughness = materials[4 * m + 3];
        o->s.materials[m].texture = material_texture ? material_texture[m] : -1;
        o->s.materials[m].glossy = material_glossy ? material_glossy[m] : 0;
    }
    o->s.textures.resize(n_textures);
    for (uint32_t t = 0; t < n_textures; ++t) o->s.textures[t].build(textures[t].rgba, textures[t].width, textures[t].height);
    for (uint32_t l = 0; l < n_lights; ++l) {
        Light lt;
        lt.type = lights[l].type;
        lt.pos = {lights[l].pos[0], lights[l].pos[1], lights[l].pos[2]};
        lt.I = {lights[l].intensity[0], lights[l].intensity[1], lights[l].intensity[2]};
        if (lt.type == LIGHT_ENV) {
            lt.W = lights[l].env_w;
            lt.H = lights[l].env_h;
            lt.img.assign(lights[l].env_rgba, lights[l].env_rgba + (size_t)lt.W * lt.H * 4);
            lt.buildEnv();
        }
        if (lt.type == LIGHT_AREA) {
            lt.p0 = {lights[l].p0[0], lights[l].p0[1], lights[l].p0[2]};
            lt.p1 = {lights[l].p1[0], lights[l].p1[1], lights[l].p1[2]};
            lt.p2 = {lights[l].p2[0], lights[l].p2[1], lights[l].p2[2]};
            lt.buildArea();
        }
        o->s.lights.push_back(std::move(lt));
    }
    return o;
}

void orc_destroy(orc_scene* o) { delete o; }

void orc_bvh_info(const orc_scene* o, uint32_t* n_nodes, int32_t* max_depth) {
    *n_nodes = (uint32_t)o->s.bvh.nodes.size();
    *max_depth = o->s.bvh.maxDepth;
}

// boxes: n*6 floats (min xyz, max xyz); left/right/tri/model/leaf: n ints
void orc_bvh_nodes(const orc_scene* o, float* boxes, int32_t* left, int32_t* right, int32_t* tri,
                   int32_t* model, int32_t* leaf) {
    const auto& nodes = o->s.bvh.nodes;
    for (size_t i = 0; i < nodes.size(); ++i) {
        const BVHNode& n = nodes[i];
        boxes[6 * i + 0] = n.box.min.x; boxes[6 * i + 1] = n.box.min.y; boxes[6 * i + 2] = n.box.min.z;
        boxes[6 * i + 3] = n.box.max.x; boxes[6 * i + 4] = n.box.max.y; boxes[6 * i + 5] = n.box.max.z;
        left[i] = n.left; right[i] = n.right; tri[i] = n.tri; model[i] = n.model; leaf[i] = n.isLeaf ? 1 : 0;
    }
}

// Rays: 6 SoA planes (ox,oy,oz,dx,dy,dz) of n floats.  counters[5] =
// {inner nodes visited, triangle tests, rays, stale far-child visits, exact-distance ties} summed over the set (may be null).
void orc_intersect(const orc_scene* o, uint64_t n, const float* rays, float* t, int32_t* id0,
                   int32_t* id1, float* u, float* v, uint64_t* counters, int threads) {
    const float *ox = rays, *oy = rays + n, *oz = rays + 2 * n, *dx = rays + 3 * n, *dy = rays + 4 * n,
                *dz = rays + 5 * n;
    if (threads < 1) threads = 1;
    std::vector<Counters> cs(threads);
    auto work = [&](int th) {
        uint64_t lo = n * th / threads, hi = n * (th + 1) / threads;
        for (uint64_t i = lo; i < hi; ++i) {
            Ray r({ox[i], oy[i], oz[i]}, {dx[i], dy[i], dz[i]});
            RayHit h = o->s.bvh.intersect(r, counters ? &cs[th] : nullptr);
            t[i] = h.dist; id0[i] = h.id0; id1[i] = h.id1; u[i] = h.bc.y; v[i] = h.bc.z;
        }
    };
    if (threads == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int i = 0; i < threads; ++i) pool.emplace_back(work, i);
        for (auto& th : pool) th.join();
    }
    if (counters) {
        counters[0] = counters[1] = counters[2] = counters[3] = counters[4] = 0;
        for (auto& c : cs) { counters[0] += c.n_inner; counters[1] += c.n_tri; counters[2] += c.n_rays; counters[3] += c.n_inner_stale; counters[4] += c.n_ties; }
    }
}

// Per-ray tie flag, independent of any visiting order: flags[i] = 1 when at least two triangles that the recursion
// can reach (every box from the root to the leaf passes intersectBBox, src/scene/Ray.cpp:17-34) pass the triangle
// test (src/scene/Ray.cpp:36-67) with dist > 1e-4 at exactly the distance of the closest hit.  The strict `<` of
// src/scene/BVH.cpp:139 then keeps whichever of them is visited first, so the reported id depends on the order
// (the reference's own traversal may never test the second one: its box is culled once the first is held).
void orc_intersect_ties(const orc_scene* o, uint64_t n, const float* rays, uint8_t* flags, int threads) {
    const float *ox = rays, *oy = rays + n, *oz = rays + 2 * n, *dx = rays + 3 * n, *dy = rays + 4 * n,
                *dz = rays + 5 * n;
    if (threads < 1) threads = 1;
    const BVH& b = o->s.bvh;
    auto work = [&](int th) {
        uint64_t lo = n * th / threads, hi = n * (th + 1) / threads;
        std::vector<int> stack;
        for (uint64_t i = lo; i < hi; ++i) {
            flags[i] = 0;
            Ray r({ox[i], oy[i], oz[i]}, {dx[i], dy[i], dz[i]});
            const RayHit h = b.intersect(r);
            if (!h.status) continue;
            int at_min = 0;
            stack.clear();
            if (intersectBBox(r, b.nodes[b.rootId].box).status) stack.push_back((int)b.rootId);
            while (!stack.empty() && at_min < 2) {
                const BVHNode& nd = b.nodes[stack.back()];
                stack.pop_back();
                if (nd.isLeaf) {
                    if (nd.tri < 0) continue;
                    const RayHit tst = intersectTriangle(r, b.vertices[b.indices[nd.tri]].pos, b.vertices[b.indices[nd.tri + 1]].pos,
                                                         b.vertices[b.indices[nd.tri + 2]].pos);
                    if (tst.status && tst.dist > 0.0001f && tst.dist == h.dist) ++at_min;
                } else {
                    if (intersectBBox(r, b.nodes[nd.left].box).status) stack.push_back(nd.left);
                    if (intersectBBox(r, b.nodes[nd.right].box).status) stack.push_back(nd.right);
                }
            }
            flags[i] = at_min >= 2 ? 1 : 0;
        }
    };
    if (threads == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int i = 0; i < threads; ++i) pool.emplace_back(work, i);
        for (auto& th : pool) th.join();
    }
}

// Checks candidate hits another traversal order reported: for ray i and the triangle whose first index sits at
// id1[i] (skipped when negative), the reference's own triangle test (src/scene/Ray.cpp:36-67) gives t, u, v, and
// ok[i] gets bit 0 when that test passes with t > 1e-4 (the acceptance of src/scene/BVH.cpp:139 without the
// distance compare) and bit 1 when the boxes of all nodes from the root to the triangle's leaf pass
// intersectBBox (src/scene/Ray.cpp:17-34), i.e. when the recursion could reach the triangle at all.
void orc_check_candidates(const orc_scene* o, uint64_t n, const float* rays, const int32_t* id1, float* t, float* u, float* v,
                          uint8_t* ok) {
    const BVH& b = o->s.bvh;
    const size_t NN = b.nodes.size();
    std::vector<int> parent(NN, -1);
    std::vector<int> leaf_of(b.indices.size() / 3, -1);
    for (size_t k = 0; k < NN; ++k) {
        const BVHNode& nd = b.nodes[k];
        if (nd.isLeaf) { if (nd.tri >= 0) leaf_of[nd.tri / 3] = (int)k; }
        else { parent[nd.left] = (int)k; parent[nd.right] = (int)k; }
    }
    const float *ox = rays, *oy = rays + n, *oz = rays + 2 * n, *dx = rays + 3 * n, *dy = rays + 4 * n,
                *dz = rays + 5 * n;
    for (uint64_t i = 0; i < n; ++i) {
        t[i] = std::numeric_limits<float>::infinity(); u[i] = v[i] = 0.f; ok[i] = 0;
        if (id1[i] < 0 || (size_t)id1[i] + 2 >= b.indices.size() + 0 || id1[i] % 3) continue;
        Ray r({ox[i], oy[i], oz[i]}, {dx[i], dy[i], dz[i]});
        const uint32_t i0 = b.indices[id1[i]], i1 = b.indices[id1[i] + 1], i2 = b.indices[id1[i] + 2];
        RayHit h = intersectTriangle(r, b.vertices[i0].pos, b.vertices[i1].pos, b.vertices[i2].pos);
        t[i] = h.dist; u[i] = h.bc.y; v[i] = h.bc.z;
        if (h.status && h.dist > 0.0001f) ok[i] |= 1;
        bool reach = true;
        for (int k = leaf_of[id1[i] / 3]; k >= 0 && reach; k = parent[k]) reach = intersectBBox(r, b.nodes[k].box).status;
        if (reach && leaf_of[id1[i] / 3] >= 0) ok[i] |= 2;
    }
}

// 0 = libm (default), 1 = the fixed single-precision polynomials the HIP kernels restate (bit-comparable films).
// Process-wide; not to be switched while a render is running.
void orc_set_math(int mode) { mathMode() = mode ? 1 : 0; }
int orc_get_math() { return mathMode(); }
// the functions of mode 1, for accuracy tests: which = 0 sin, 1 cos, 2 acos, 3 atan2(x, y), 4 log2, 5 log, 6 exp, 7 pow(x, y)
void orc_dm_eval(int which, uint64_t n, const float* x, const float* y, float* out) {
    for (uint64_t i = 0; i < n; ++i) {
        switch (which) {
            case 0: out[i] = dm::sin(x[i]); break;
            case 1: out[i] = dm::cos(x[i]); break;
            case 2: out[i] = dm::acos(x[i]); break;
            case 3: out[i] = dm::atan2(x[i], y[i]); break;
            case 4: out[i] = dm::log2(x[i]); break;
            case 5: out[i] = dm::log(x[i]); break;
            case 6: out[i] = dm::exp(x[i]); break;
            default: out[i] = dm::pow(x[i], y[i]); break;
        }
    }
}

// Inner-node visits of every ray of a set (the length of each ray's traversal), for studying launch tails.
void orc_intersect_visits(const orc_scene* o, uint64_t n, const float* rays, uint32_t* visits) {
    const float *ox = rays, *oy = rays + n, *oz = rays + 2 * n, *dx = rays + 3 * n, *dy = rays + 4 * n,
                *dz = rays + 5 * n;
    for (uint64_t i = 0; i < n; ++i) {
        Counters c;
        Ray r({ox[i], oy[i], oz[i]}, {dx[i], dy[i], dz[i]});
        o->s.bvh.intersect(r, &c);
        visits[i] = (uint32_t)(c.n_inner + c.n_tri);
    }
}

void orc_occluded(const orc_scene* o, uint64_t n, const float* rays, int32_t* status, uint64_t* counters) {
    const float *ox = rays, *oy = rays + n, *oz = rays + 2 * n, *dx = rays + 3 * n, *dy = rays + 4 * n,
                *dz = rays + 5 * n;
    Counters c;
    for (uint64_t i = 0; i < n; ++i) {
        Ray r({ox[i], oy[i], oz[i]}, {dx[i], dy[i], dz[i]});
        status[i] = o->s.bvh.occluded(r, counters ? &c : nullptr).status ? 1 : 0;
    }
    if (counters) { counters[0] = c.n_inner; counters[1] = c.n_tri; counters[2] = c.n_rays; }
}

// Any-hit with a far bound per ray (shadow rays towards area lights)
void orc_occluded_bounded(const orc_scene* o, uint64_t n, const float* rays, const float* tmax, int32_t* status) {
    const float *ox = rays, *oy = rays + n, *oz = rays + 2 * n, *dx = rays + 3 * n, *dy = rays + 4 * n,
                *dz = rays + 5 * n;
    for (uint64_t i = 0; i < n; ++i) {
        Ray r({ox[i], oy[i], oz[i]}, {dx[i], dy[i], dz[i]});
        status[i] = o->s.bvh.occluded(r, nullptr, tmax[i]).status ? 1 : 0;
    }
}

// Brute force over every triangle with the same triangle test and the same
// acceptance rule, in index-buffer order (checks the BVH, not the maths).
void orc_brute_force(const orc_scene* o, uint64_t n, const float* rays, float* t, int32_t* id1) {
    const float *ox = rays, *oy = rays + n, *oz = rays + 2 * n, *dx = rays + 3 * n, *dy = rays + 4 * n,
                *dz = rays + 5 * n;
    const BVH& b = o->s.bvh;
    for (uint64_t i = 0; i < n; ++i) {
        Ray r({ox[i], oy[i], oz[i]}, {dx[i], dy[i], dz[i]});
        float best = std::numeric_limits<float>::infinity();
        int bid = -1;
        for (size_t k = 0; k + 2 < b.indices.size(); k += 3) {
            RayHit h = intersectTriangle(r, b.vertices[b.indices[k]].pos, b.vertices[b.indices[k + 1]].pos,
                                         b.vertices[b.indices[k + 2]].pos);
            if (h.status && h.dist > 0.0001f && h.dist < best) { best = h.dist; bid = (int)k; }
        }
        t[i] = best; id1[i] = bid;
    }
}

// camera: pos, right, up, forward (12 floats).  Renders passes
// [first_pass, first_pass+n_passes) into film (W*H*4 floats, read-modify-write).
// ray_counts[12] (may be null): see the end of the function.
void orc_render(const orc_scene* o, const float* camera, int W, int H, int max_depth, uint64_t seed,
                uint32_t first_pass, uint32_t n_passes, int threads, float* film, uint64_t* ray_counts, int tile_size,
                float* duv_out /* W*H*4 or null: dUVx, dUVy of each primary hit (last pass), untouched where it missed */) {
    Camera cam;
    cam.pos = {camera[0], camera[1], camera[2]};
    cam.right = {camera[3], camera[4], camera[5]};
    cam.up = {camera[6], camera[7], camera[8]};
    cam.forward = {camera[9], camera[10], camera[11]};
    RenderCounters rc;
    for (uint32_t p = 0; p < n_passes; ++p)
        renderPass(o->s, cam, W, H, max_depth, seed, first_pass + p, (float)(first_pass + p), film, threads,
                   ray_counts ? &rc : nullptr, tile_size > 0 ? tile_size : 64, duv_out);
    if (ray_counts) {
        // [0..1] rays, [2..5] node visits / triangle tests of the closest-hit and of the shadow rays, [6..11] of those, the
        // camera rays and the shadow rays from the primary vertex towards point lights (rays, visits, tests each)
        ray_counts[0] = rc.closest; ray_counts[1] = rc.shadow;
        ray_counts[2] = rc.closestInner; ray_counts[3] = rc.closestTri; ray_counts[4] = rc.shadowInner; ray_counts[5] = rc.shadowTri;
        ray_counts[6] = rc.primRays; ray_counts[7] = rc.primInner; ray_counts[8] = rc.primTri;
        ray_counts[9] = rc.primPointRays; ray_counts[10] = rc.primPointInner; ray_counts[11] = rc.primPointTri;
    }
}

// Camera rays of src/cpu/PathTracer.cpp:43-52 for the whole film, row-major.
// One pass rendered only to record the bounce rays leaving the `depth`-th
// shaded vertex of every pixel (rays: 6 planes of W*H floats, flags: W*H bytes).
void orc_capture_bounce_rays(const orc_scene* o, const float* camera, int W, int H, int max_depth, uint64_t seed, uint32_t pass,
                             int threads, int depth, float* rays, uint8_t* flags) {
    Camera cam;
    cam.pos = {camera[0], camera[1], camera[2]};
    cam.right = {camera[3], camera[4], camera[5]};
    cam.up = {camera[6], camera[7], camera[8]};
    cam.forward = {camera[9], camera[10], camera[11]};
    std::vector<float> film((size_t)W * H * 4, 0.f);
    std::memset(flags, 0, (size_t)W * H);
    RayCapture cap;
    cap.rays = rays; cap.flags = flags; cap.depth = depth; cap.n = (size_t)W * H;
    renderPass(o->s, cam, W, H, max_depth, seed, pass, 0.0f, film.data(), threads, nullptr, 64, nullptr, &cap);
}

void orc_camera_rays(const float* camera, int W, int H, float* rays) {
    Camera cam;
    cam.pos = {camera[0], camera[1], camera[2]};
    cam.right = {camera[3], camera[4], camera[5]};
    cam.up = {camera[6], camera[7], camera[8]};
    cam.forward = {camera[9], camera[10], camera[11]};
    size_t n = (size_t)W * H;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            Ray r = cameraRay(cam, x, y, W, H);
            size_t i = (size_t)y * W + x;
            rays[i] = r.org.x; rays[n + i] = r.org.y; rays[2 * n + i] = r.org.z;
            rays[3 * n + i] = r.dir.x; rays[4 * n + i] = r.dir.y; rays[5 * n + i] = r.dir.z;
        }
}

// ---- unit-level hooks for known-answer tests
uint32_t orc_morton3d(float x, float y, float z) { return morton3D({x, y, z}); }
uint32_t orc_expand_bits(uint32_t v) { return expandBits(v); }

int orc_intersect_bbox(const float* org, const float* dir, const float* bmin, const float* bmax, float* dist) {
    Ray r({org[0], org[1], org[2]}, {dir[0], dir[1], dir[2]});
    BBox b;
    b.min = {bmin[0], bmin[1], bmin[2]};
    b.max = {bmax[0], bmax[1], bmax[2]};
    RayHit h = intersectBBox(r, b);
    *dist = h.dist;
    return h.status ? 1 : 0;
}

int orc_intersect_triangle(const float* org, const float* dir, const float* v0, const float* v1,
                           const float* v2, float* out /* dist, bc0, bc1, bc2 */) {
    Ray r({org[0], org[1], org[2]}, {dir[0], dir[1], dir[2]});
    RayHit h = intersectTriangle(r, {v0[0], v0[1], v0[2]}, {v1[0], v1[1], v1[2]}, {v2[0], v2[1], v2[2]});
    out[0] = h.dist; out[1] = h.bc.x; out[2] = h.bc.y; out[3] = h.bc.z;
    return h.status ? 1 : 0;
}

void orc_cosine_sample_hemisphere(float u0, float u1, float* out3) {
    vec3 d = cosineSampleHemisphere({u0, u1});
    out3[0] = d.x; out3[1] = d.y; out3[2] = d.z;
}

// Oren-Nayar lobe in the local frame (z = normal)
void orc_diffuse_f(const float* color, float sigma, const float* out, const float* in, float* f3) {
    Diffuse d({color[0], color[1], color[2]}, sigma);
    vec3 f = d.f({out[0], out[1], out[2]}, {in[0], in[1], in[2]});
    f3[0] = f.x; f3[1] = f.y; f3[2] = f.z;
}

float orc_diffuse_sample(const float* color, float sigma, const float* out, float u0, float u1, float* in3,
                         float* f3) {
    Diffuse d({color[0], color[1], color[2]}, sigma);
    vec3 in;
    float p;
    vec3 f = d.sampled({out[0], out[1], out[2]}, {u0, u1}, in, p);
    in3[0] = in.x; in3[1] = in.y; in3[2] = in.z;
    f3[0] = f.x; f3[1] = f.y; f3[2] = f.z;
    return p;
}

float orc_power_heuristic(float f, float g) { return powerHeuristic(1, f, 1, g); }

// Distribution1D: returns the sampled value; writes pdf and offset
float orc_dist1d_sample(const float* func, int n, float u, float* pdf, int* off) {
    Distribution1D d(std::vector<float>(func, func + n));
    return d.sampleContinuous(u, *pdf, off);
}
void orc_dist1d_cdf(const float* func, int n, float* cdf, float* func_int) {
    Distribution1D d(std::vector<float>(func, func + n));
    std::memcpy(cdf, d.cdf.data(), sizeof(float) * (n + 1));
    *func_int = d.funcInt;
}
void orc_dist2d_sample(const float* data, int w, int h, float u0, float u1, float* out2, float* pdf) {
    Distribution2D d(std::vector<float>(data, data + (size_t)w * h), w, h);
    vec2 r = d.sampleContinuous({u0, u1}, *pdf);
    out2[0] = r.x; out2[1] = r.y;
}
float orc_dist2d_pdf(const float* data, int w, int h, float p0, float p1) {
    Distribution2D d(std::vector<float>(data, data + (size_t)w * h), w, h);
    return d.pdf({p0, p1});
}

// Glossy-lobe hooks (directions in the shading frame)
static Microfacet mf_of(int kind, float ax, float ay) {
    Microfacet m;
    m.kind = kind; m.alphax = ax; m.alphay = ay;
    return m;
}
float orc_erf_inv(float x) { return erfInv(x); }
float orc_erf_poly(float x) { return erfPoly(x); }
float orc_mf_D(int kind, float ax, float ay, const float* wh) { return mf_of(kind, ax, ay).D({wh[0], wh[1], wh[2]}); }
float orc_mf_lambda(int kind, float ax, float ay, const float* w) { return mf_of(kind, ax, ay).lambda({w[0], w[1], w[2]}); }
float orc_mf_pdf(int kind, float ax, float ay, const float* wo, const float* wh) {
    return mf_of(kind, ax, ay).pdf({wo[0], wo[1], wo[2]}, {wh[0], wh[1], wh[2]});
}
void orc_mf_sample_wh(int kind, float ax, float ay, const float* wo, float u0, float u1, float* wh) {
    vec3 h = mf_of(kind, ax, ay).sampleWh({wo[0], wo[1], wo[2]}, {u0, u1});
    wh[0] = h.x; wh[1] = h.y; wh[2] = h.z;
}
void orc_fresnel_conductor(const float* etaI, const float* etaT, const float* k, float cosI, float* out) {
    FresnelConductor f;
    f.etaI = {etaI[0], etaI[1], etaI[2]}; f.etaT = {etaT[0], etaT[1], etaT[2]}; f.k = {k[0], k[1], k[2]};
    vec3 r = f.eval(cosI);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
// The BSDF a material with the given lobe builds, in an identity frame
static BSDF bsdf_of(int glossy, float roughness, const float* color) {
    Vertex v;
    v.tangent = {1, 0, 0}; v.binormal = {0, 1, 0}; v.normal = {0, 0, 1};
    MaterialData m;
    m.roughness = roughness; m.glossy = glossy;
    vec3 c = {color[0], color[1], color[2]};
    BSDF b(v, Diffuse(c, roughness));
    addGlossy(b, m, c);
    return b;
}
float orc_bsdf_f_pdf(int glossy, float roughness, const float* color, const float* out, const float* in, float* f3) {
    BSDF b = bsdf_of(glossy, roughness, color);
    vec3 o = {out[0], out[1], out[2]}, i = {in[0], in[1], in[2]};
    vec3 f = b.f(o, i);
    f3[0] = f.x; f3[1] = f.y; f3[2] = f.z;
    return b.pdf(o, i);
}
float orc_bsdf_sample(int glossy, float roughness, const float* color, const float* out, float u0, float u1, float* in3, float* f3) {
    BSDF b = bsdf_of(glossy, roughness, color);
    vec3 in{0, 0, 0};
    float pdf = 0;
    vec3 f = b.sample({out[0], out[1], out[2]}, {u0, u1}, in, pdf);
    in3[0] = in.x; in3[1] = in.y; in3[2] = in.z;
    f3[0] = f.x; f3[1] = f.y; f3[2] = f.z;
    return pdf;
}

// Texture hooks: mip level sizes / texels and the two filters (channels 0..255)
int orc_texture_levels(const orc_scene* o, int t) { return (int)o->s.textures[t].levels.size(); }
void orc_texture_level(const orc_scene* o, int t, int level, int32_t* w, int32_t* h, uint8_t* rgba_out) {
    const Texture& tx = o->s.textures[t];
    *w = tx.w[level]; *h = tx.h[level];
    if (rgba_out) std::memcpy(rgba_out, tx.levels[level].data(), tx.levels[level].size());
}
void orc_texture_bilinear(const orc_scene* o, int t, int level, float u, float v, float* rgb) {
    vec3 c = o->s.textures[t].bilinear(level, {u, v});
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}
void orc_texture_trilinear(const orc_scene* o, int t, float u, float v, const float* duvx, const float* duvy, float* rgb) {
    vec3 c = o->s.textures[t].filterTrilinear({u, v}, {duvx[0], duvx[1]}, {duvy[0], duvy[1]});
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

float orc_rng_u01(uint64_t seed, uint32_t pixel, uint32_t pass, uint32_t dim) {
    return rng_u01(seed, pixel, pass, dim);
}

// Light hooks: sample / pdf / le of light `l` at a vertex position
void orc_light_sample(const orc_scene* o, int l, const float* p, float u0, float u1, float* in3, float* pdf,
                      float* li3) {
    Vertex v;
    v.pos = {p[0], p[1], p[2]};
    vec3 in{0, 0, 0};
    vec3 li = o->s.lights[l].sampleLi(v, {u0, u1}, in, *pdf);
    in3[0] = in.x; in3[1] = in.y; in3[2] = in.z;
    li3[0] = li.x; li3[1] = li.y; li3[2] = li.z;
}
// Same with the shadow ray's far bound, and the pdf seen from a given position
float orc_light_sample_t(const orc_scene* o, int l, const float* p, float u0, float u1, float* in3, float* pdf, float* li3) {
    Vertex v;
    v.pos = {p[0], p[1], p[2]};
    vec3 in{0, 0, 0};
    float tmax = 0;
    vec3 li = o->s.lights[l].sampleLi(v, {u0, u1}, in, *pdf, &tmax);
    in3[0] = in.x; in3[1] = in.y; in3[2] = in.z;
    li3[0] = li.x; li3[1] = li.y; li3[2] = li.z;
    return tmax;
}
float orc_light_pdf_at(const orc_scene* o, int l, const float* p, const float* in) {
    Vertex v;
    v.pos = {p[0], p[1], p[2]};
    return o->s.lights[l].pdfLi(v, {in[0], in[1], in[2]});
}
float orc_light_pdf(const orc_scene* o, int l, const float* in) {
    Vertex v;
    return o->s.lights[l].pdfLi(v, {in[0], in[1], in[2]});
}
void orc_light_le(const orc_scene* o, int l, const float* dir, float* le3) {
    Ray r({0, 0, 0}, {dir[0], dir[1], dir[2]});
    vec3 e = o->s.lights[l].le(r);
    le3[0] = e.x; le3[1] = e.y; le3[2] = e.z;
}

}  // extern "C"
