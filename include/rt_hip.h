/* rt_hip.h — C-ABI of the MI355X (gfx950) wavefront path-tracing core.
 *
 * Drop-in boundary for the reference renderer's CPU integrator: an
 * `Integrator` subclass (src/cpu/Integrator.hpp:14-58) calls this library
 * once per sample pass instead of running `Integrator::update()`
 * (src/cpu/Integrator.cpp:26-67) on its 16-thread tile pool.  Plain pointers
 * and sizes only; every call returns 0 on success or a negative rt_status,
 * and rt_last_error() gives the text.  A context is used by one thread at a
 * time (the reference drives its integrator from one render thread,
 * src/cpu/Integrator.cpp:69-79).
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

enum rt_status {
    RT_OK = 0,
    RT_ERR_ARG = -1,      /* null / inconsistent argument */
    RT_ERR_HIP = -2,      /* a HIP runtime call failed */
    RT_ERR_NO_DEVICE = -3,/* no usable gfx950 device */
    RT_ERR_LIMIT = -4,    /* size beyond what the context was created for */
    RT_ERR_STATE = -5     /* call made in the wrong state (e.g. no camera) */
};

/* Light kinds of src/lights/LightSource.hpp:12-17 that the reference implements. */
enum rt_light_type { RT_LIGHT_POINT = 0, RT_LIGHT_AREA = 2, RT_LIGHT_ENV = 3 };

/* Glossy lobe of a material.  NONE is the reference as shipped: its
 * Torrance-Sparrow lobe only exists inside a comment
 * (src/scene/Material.cpp:205-209).  BECKMANN adds that lobe
 * (src/bsdf/Microfacet.cpp, with the reflected direction and the inverse error
 * function corrected), GGX the same lobe over a Trowbridge-Reitz distribution;
 * parameters as in the comment: alpha = (roughness, roughness + 0.1),
 * FresnelConductor(0.5, 0.5, diffuse colour), R = 1. */
enum rt_glossy { RT_GLOSSY_NONE = 0, RT_GLOSSY_BECKMANN = 1, RT_GLOSSY_GGX = 2 };

/* Per-model material: the part of MaterialUBO (src/scene/Material.hpp:17-21)
 * that Material::computeBSDF reads (src/scene/Material.cpp:195-225). */
typedef struct rt_material {
    float diffuse_color[3];
    float roughness;
    int32_t texture;         /* index into rt_scene_desc.textures (Material::_diffTexture,
                                src/scene/Material.hpp:61), or -1 for none */
    int32_t glossy;          /* rt_glossy */
} rt_material;

/* Diffuse map: the RGBA8 image Material::setDiffuseTexture loads
 * (src/scene/Material.cpp:92-96).  Level 0 only, row-major, 4 bytes a texel;
 * the library builds the mip chain (2x2 box average, rounded). */
typedef struct rt_texture {
    int32_t width, height;
    const uint8_t* rgba8;
} rt_texture;

/* PointLight{pos = light2world*(0,0,0,1), I} (src/lights/PointLight.cpp:9-18)
 * or EnvLight{RGBA32F lat-long image, scale} (src/lights/EnvLight.cpp:14-26),
 * or an area light: one emissive triangle behind the LightSource interface
 * (src/lights/LightSource.hpp:15 names the type, the reference has no class
 * for it): uniform area sampling, one-sided radiance `intensity` on the side
 * of cross(p1-p0, p2-p0), shadow rays bounded at the sampled point. */
typedef struct rt_light {
    int32_t type;            /* rt_light_type */
    float pos[3];            /* point only */
    float intensity[3];      /* point: I; env: the `light` scale */
    const float* env_rgba;   /* env only: env_w*env_h*4 floats, row-major */
    int32_t env_w, env_h;
    float p0[3], p1[3], p2[3]; /* area only */
} rt_light;

/* The scene as BVH::runLBVH sees it (src/scene/BVH.cpp:293-317): all models'
 * vertices concatenated, indices already offset by each model's vertex base,
 * one model id per triangle.  Attribute pointers are strided because the
 * byte layout of mango::sVertex is not part of the reference tree.  tangent
 * and binormal may both be null: the library then derives an orthonormal
 * frame from the shading normal. */
typedef struct rt_scene_desc {
    uint32_t n_vertices;
    const float* pos;      uint32_t pos_stride;      /* strides in bytes */
    const float* normal;   uint32_t normal_stride;
    const float* uv;       uint32_t uv_stride;
    const float* tangent;  uint32_t tangent_stride;
    const float* binormal; uint32_t binormal_stride;
    uint32_t n_indices;                               /* 3 * triangles */
    const uint32_t* indices;
    const int32_t* tri_model;                         /* n_indices/3 entries */
    uint32_t n_models;
    const rt_material* materials;                     /* one per model */
    uint32_t n_lights;
    const rt_light* lights;
    uint32_t n_textures;
    const rt_texture* textures;                       /* may be null when n_textures == 0 */
} rt_scene_desc;

/* Compile-time constants of the reference made parameters: frame size
 * (src/app/App.cpp:9-10), maxDepth=4 (src/cpu/PathTracer.cpp:15), tileSize=64
 * (src/cpu/Integrator.hpp:56).  rank/world shard the film by interleaved
 * tiles: tile t belongs to rank t % world. */
typedef struct rt_params {
    int32_t width, height;
    int32_t max_depth;
    int32_t tile_size;     /* 0 -> 64 */
    uint64_t seed;
    int32_t device;        /* HIP device ordinal */
    int32_t rank, world;   /* world 0 or 1 -> whole film */
    uint32_t flags;        /* rt_flags; 0 = defaults */
    uint32_t batch;        /* passes rt_render traces together as one wavefront (more rays per launch;
                              path state grows with it: 144 B per pixel slot and 41 B per slot and light, 45 B
                              with area lights); 0 -> 32, limited by a budget of 64 GiB or 60 % of the device's free memory
                              (rt_stats.batch reports what was chosen).  The film is
                              the one that many separate passes leave, bit for bit. */
} rt_params;

enum rt_flags {
    RT_FLAG_HOST_BVH = 1,  /* build the LBVH with the host builder instead of on the GPU (same tree) */
    RT_FLAG_WIDE_BVH = 2   /* also collapse the LBVH into 4-wide nodes on the GPU and let rt_trace_closest* /
                              rt_trace_occluded* walk those: the same triangles are reachable and the same
                              distance is returned, but children are visited in order of their own box distance,
                              so of several triangles at exactly the same distance another one than the
                              reference's (src/scene/BVH.cpp:139,160-166) may be reported.  Any-hit results are
                              identical. */
};

typedef struct rt_stats {
    uint64_t closest_rays;     /* closest-hit rays traced since create/reset */
    uint64_t shadow_rays;      /* any-hit rays traced */
    uint64_t passes;           /* sample passes accumulated (the reference's _frames) */
    uint32_t n_nodes;          /* nodes of the reference-layout BVH (2T-1) */
    uint32_t n_triangles;
    int32_t bvh_max_depth;
    uint32_t n_local_pixels;   /* pixels of this rank's tiles */
    float last_trace_ms;       /* device time of the last rt_trace_* kernel (HIP events) */
    float last_render_ms;      /* device time of the last rt_render call */
    float bvh_build_ms;        /* LBVH build (GPU or host) including its uploads, wall */
    uint64_t exact_closest_rays; /* of closest_rays / shadow_rays: rays whose operands the fast box test cannot */
    uint64_t exact_shadow_rays;  /* decide like the reference's selects, re-traced by the exact follow-up launch */
    uint64_t device_bytes;       /* device memory the context holds (scene, tables, path state, staging) */
    uint64_t path_state_bytes;   /* of that: path state, shadow entries and film, the part that grows with batch */
    uint32_t batch;              /* passes rt_render traces together (rt_params.batch after the budget) */
} rt_stats;

/* Replaces Integrator::Integrator (src/cpu/Integrator.cpp:7-15): builds the
 * LBVH of src/scene/BVH.cpp:287-336 (on the GPU by default), uploads scene
 * and film.  RT_ERR_LIMIT when the node array (64 B per node slot, about one
 * slot per triangle) or the triangle array (48 B each) would reach 4 GiB,
 * i.e. from about 67 M triangles: the traversal kernels address both with
 * unsigned 32-bit byte offsets.  Also when max_depth * (4 * n_lights + 2)
 * exceeds the sampler's 65536 dimensions. */
int rt_create(const rt_scene_desc* scene, const rt_params* params, rt_ctx** out);
void rt_destroy(rt_ctx* ctx);
/* Error text of the last failed call on ctx (ctx may be null for rt_create). */
const char* rt_last_error(const rt_ctx* ctx);

/* Replaces Integrator::nextFrame's camera snapshot (src/cpu/Integrator.cpp:85-92).
 * A changed camera restarts accumulation like src/cpu/Integrator.cpp:27-30. */
int rt_set_camera(rt_ctx* ctx, const float pos[3], const float right[3], const float up[3],
                  const float forward[3]);
int rt_reset_accumulation(rt_ctx* ctx);

/* Replaces n_passes calls of Integrator::update (src/cpu/Integrator.cpp:26-67),
 * each one PathTracer::computeTile over every tile of this rank
 * (src/cpu/PathTracer.cpp:17-164): one sample per pixel per pass, running mean. */
int rt_render(rt_ctx* ctx, uint32_t n_passes);

/* Replaces Integrator::sync (src/cpu/Integrator.cpp:21-24): the linear
 * RGBA32F film, row-major, y=0 at the top, width*height*4 floats.  Pixels of
 * other ranks' tiles are written as zero. */
int rt_read_film(rt_ctx* ctx, float* rgba_out);
/* Diagnostic: the UV differences (dUVx.xy, dUVy.xy, in film pixels) that the
 * last pass handed to Material::computeBSDF for each primary hit
 * (src/cpu/PathTracer.cpp:83-90), width*height*4 floats of this rank's pixels;
 * zero where the primary ray missed or the scene has no textures. */
int rt_read_primary_duv(rt_ctx* ctx, float* duv_out);
/* Diagnostic / workload capture: the rays the last rt_render traced at its
 * deepest bounce (the `ray` of src/cpu/PathTracer.cpp:134-135 at d = max_depth-2),
 * in slot order, as 6 SoA planes of `cap` floats; *n_out = how many there were
 * (at most cap are copied; cap = 0 only asks for the count). */
int rt_read_last_bounce_rays(rt_ctx* ctx, float* rays_soa, uint32_t cap, uint32_t* n_out);

/* This rank's packed tile slab on the device: n_tiles*tile*tile*4 floats,
 * tiles in increasing global tile index.  For gathering the film across ranks. */
int rt_film_slab(rt_ctx* ctx, const float** device_ptr, uint32_t* n_tiles, uint64_t* n_floats);
/* Copies the slab into caller-owned device memory (n_floats from rt_film_slab). */
int rt_copy_film_slab(rt_ctx* ctx, void* device_dst, uint64_t n_floats);
/* Scatters a gathered set of slabs (rank-major, each padded to
 * slab_floats_per_rank) into a width*height*4 host film. */
int rt_unpack_slabs(int32_t width, int32_t height, int32_t tile_size, int32_t world,
                    const float* slabs_host, uint64_t slab_floats_per_rank, float* rgba_out);

/* The same scatter on the device: d_slabs is the all-gathered buffer (rank-major, device memory), d_rgba_out a
 * width*height*4 device film in the layout `Integrator::sync()` copies out (src/cpu/Integrator.cpp:21-24).  Runs on
 * the context's stream and returns when it is done; width, height and tile size are the context's. */
int rt_unpack_slabs_dev(rt_ctx* ctx, int32_t world, const float* d_slabs, uint64_t slab_floats_per_rank, float* d_rgba_out);

/* Host-side inverse of rt_unpack_slabs for one rank: packs that rank's tiles
 * of a width*height*4 film into slab order.  rt_slab_floats gives the slab
 * size (floats) a rank holds.  Neither touches a device. */
uint64_t rt_slab_floats(int32_t width, int32_t height, int32_t tile_size, int32_t rank, int32_t world);
int rt_pack_slab(int32_t width, int32_t height, int32_t tile_size, int32_t rank, int32_t world,
                 const float* rgba_in, float* slab_out);

/* BVH::intersect (src/scene/BVH.cpp:49-61,129-170) over n rays.  rays: six
 * planes ox,oy,oz,dx,dy,dz of n floats (host memory).  Outputs (host, n
 * entries each): dist (+inf on a miss), id0 = model index, id1 = index-buffer
 * offset of the triangle (both -1 on a miss), barycentrics u,v. */
int rt_trace_closest(rt_ctx* ctx, uint32_t n, const float* rays_soa, float* dist, int32_t* id0,
                     int32_t* id1, float* u, float* v);
/* BVH::occluded (src/scene/BVH.cpp:63-127): status[i] = 1 if anything is hit. */
int rt_trace_occluded(rt_ctx* ctx, uint32_t n, const float* rays_soa, int32_t* status);
/* The same with a far bound per ray: only hits with 1e-4 < dist < tmax[i]
 * occlude (tmax = +inf is rt_trace_occluded).  The reference's shadow rays are
 * unbounded (src/cpu/PathTracer.cpp:101); the bound belongs to the area
 * lights, whose samples lie at a finite distance. */
int rt_trace_occluded_bounded(rt_ctx* ctx, uint32_t n, const float* rays_soa, const float* tmax, int32_t* status);

/* The same two with every buffer already in device memory (HBM-resident
 * inputs; the benchmark path).  Launches on the context's stream; `repeat`
 * back-to-back launches are timed together with HIP events on that stream
 * and the mean per-launch time is stored in rt_stats.last_trace_ms. */
int rt_trace_closest_dev(rt_ctx* ctx, uint32_t n, const float* d_rays_soa, float* d_dist,
                         int32_t* d_id0, int32_t* d_id1, float* d_u, float* d_v, uint32_t repeat);
int rt_trace_occluded_dev(rt_ctx* ctx, uint32_t n, const float* d_rays_soa, int32_t* d_status,
                          uint32_t repeat);

/* rt_trace_closest_dev for callers that cannot allocate device memory (code
 * under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc): it has no counterpart in the reference.  Same inputs, outputs
 * and results as rt_trace_closest_dev with repeat = 1 (on the tree the context
 * was created with, RT_FLAG_WIDE_BVH included), but the launch's work
 * counters and its list of rays for the exact follow-up launch live in a fixed
 * array inside the library (16 MiB + 4.25 KiB per device, held until the
 * process ends) instead of memory the context grows on demand, so the call
 * never allocates, frees or synchronises: it enqueues on the context's stream
 * and returns (rt_synchronize waits; rt_stats.last_trace_ms is not updated).
 * At most RT_NOALLOC_MAX_RAYS rays per call, RT_ERR_LIMIT beyond.  Every call
 * on a GPU shares that GPU's array, whichever context makes it: calls that
 * overlap on one GPU (two contexts, two threads) race; calls through one
 * context are ordered by its stream. */
#define RT_NOALLOC_MAX_RAYS (1u << 22)
int rt_trace_closest_noalloc(rt_ctx* ctx, uint32_t n, const float* d_rays_soa, float* d_dist,
                             int32_t* d_id0, int32_t* d_id1, float* d_u, float* d_v);

/* The LBVH in the reference's node layout (src/scene/BBox.hpp:27-56), for
 * checking tree identity: boxes n*6 floats (min xyz, max xyz); left, right,
 * tri (triIds[0]), model (modelIds[0]), leaf: n int32 each.  Pass nulls to
 * query n_nodes only. */
int rt_get_bvh(rt_ctx* ctx, uint32_t* n_nodes, float* boxes, int32_t* left, int32_t* right,
               int32_t* tri, int32_t* model, int32_t* leaf);

/* Diagnostic: the 4-wide nodes (RT_FLAG_WIDE_BVH), 32 floats each: six planes of four (min x, min y, min z,
 * max x, max y, max z of the four children), four int32 links (>= 0 wide node, < 0 ~slot of a leaf in Morton
 * order), then the slot count, the binary node id and two zeros.  Pass nodes32 = null to query the counts. */
int rt_get_wide_bvh(rt_ctx* ctx, uint32_t* n_nodes, int32_t* depth, float* nodes32);

/* Host-only run of the same LBVH builder (no device needed): fills the
 * reference-layout node arrays for n_indices/3 triangles; every array holds
 * 2*(n_indices/3)-1 entries.  Replaces BVH::runLBVH (src/scene/BVH.cpp:287-336)
 * for callers that only want the tree. */
int rt_lbvh_host(const float* pos, uint32_t pos_stride, const uint32_t* indices, uint32_t n_indices,
                 const int32_t* tri_model, int32_t* max_depth, float* boxes, int32_t* left, int32_t* right,
                 int32_t* tri, int32_t* model, int32_t* leaf);
/* Morton code of src/scene/BVH.cpp:189-212 for a point of the unit cube. */
uint32_t rt_morton3d(float x, float y, float z);

/* ---- several GPUs of one node driven by ONE process (the C++ host of the reference is a single process:
 * src/app/App.cpp:40 makes one integrator).  Replaces the tile pool of Integrator::update
 * (src/cpu/Integrator.cpp:33-49: 16 threads pulling 64x64 tiles) by n_devices GPUs, device i taking the tiles
 * t % n_devices == i, and Integrator::sync (src/cpu/Integrator.cpp:21-24) by a gather of the devices' tile slabs:
 * one RCCL communicator per device (ncclCommInitAll), one group of ncclSend / ncclRecv pairs over xGMI towards the
 * first device (RT_NODE_GATHER=allgather in the environment: one grouped ncclAllGather instead), one scatter kernel
 * there.  The scene is replicated; params->device, rank and world are ignored (set per device).
 * Every context renders on a host thread of its own inside rt_node_render.  RCCL is loaded at run time
 * (librccl.so.1, or only the path in RT_RCCL_LIB when that is set), before anything is allocated; RT_ERR_NO_DEVICE
 * if it cannot be. */
typedef struct rt_node rt_node;
int rt_node_create(const rt_scene_desc* scene, const rt_params* params, const int32_t* devices, int32_t n_devices, rt_node** out);
void rt_node_destroy(rt_node* node);
/* Error text of the last failed call on node (node may be null for rt_node_create). */
const char* rt_node_last_error(const rt_node* node);
int rt_node_device_count(const rt_node* node);
/* The context of device i (for rt_get_stats and the trace entry points); owned by the node. */
rt_ctx* rt_node_context(rt_node* node, int32_t i);
int rt_node_set_camera(rt_node* node, const float pos[3], const float right[3], const float up[3], const float forward[3]);
int rt_node_reset_accumulation(rt_node* node);
/* n_passes calls of Integrator::update over the whole film, every device tracing its own tiles. */
int rt_node_render(rt_node* node, uint32_t n_passes);
/* Gathers the film on the first device; *d_film_out = width*height*4 floats of device memory owned by the node
 * (row-major, y = 0 at the top), valid until the next gather. */
int rt_node_gather_film(rt_node* node, const float** d_film_out);
/* rt_node_gather_film + copy to host, replacing Integrator::sync's source (src/cpu/Integrator.cpp:21-24):
 * *host_film_out = the whole film, width*height*4 floats, in pinned host memory owned by the node (allocated at the
 * first call), valid until the next rt_node_map_film / rt_node_read_film.  A host that only uploads the film to its
 * own staging buffer reads it from here without another copy. */
int rt_node_map_film(rt_node* node, const float** host_film_out);
/* rt_node_map_film + memcpy into caller memory: the whole film, width*height*4 floats. */
int rt_node_read_film(rt_node* node, float* rgba_out);
/* Device time of the last all-gather (HIP events on the first device's stream). */
float rt_node_last_gather_ms(const rt_node* node);

int rt_get_stats(rt_ctx* ctx, rt_stats* out);
/* rt_stats.last_trace_ms without the device round trip of rt_get_stats. */
float rt_last_trace_ms(const rt_ctx* ctx);
int rt_synchronize(rt_ctx* ctx);

/* Library identity: "rt_hip <version> gfx950". */
const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
