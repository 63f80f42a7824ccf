// C-ABI of the gfx950 path-tracing core (declared in include/rt_hip.h):
// context, scene upload, the per-pass wavefront schedule, film read-back.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "lbvh_device.h"
#include "lbvh_host.h"
#include "shade.h"
#include "trace_wide.h"
#include "wide_bvh.h"

using namespace rt;

#ifndef RT_NOALLOC_MAX_RAYS
#define RT_NOALLOC_MAX_RAYS (1u << 22)   // 4,194,304 rays: 16 MiB of deferred list + the work counters
#endif

// Scratch of rt_trace_closest_noalloc: the work counters of one traversal launch followed by the list of rays
// handed to the exact follow-up launch (one index per ray at most).  File scope, so the entry point never
// allocates; every device the library runs on holds its own copy until the process ends.
constexpr uint32_t kNoAllocMaxRays = RT_NOALLOC_MAX_RAYS;
constexpr size_t kNoAllocWorkBytes = (sizeof(rt::TraceWork) + 255) / 256 * 256;
constexpr size_t kNoAllocBytes = kNoAllocWorkBytes + (size_t)kNoAllocMaxRays * sizeof(uint32_t);
static __device__ __align__(256) unsigned char scratch[kNoAllocBytes];


namespace {

thread_local std::string g_create_error;

// Device address of `scratch` on each device (the array's name is only a host-side handle: never pass it to a
// kernel or a memset).  Filled once per device by scratch_address(), with that device current.
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[kMaxDevices] = {};

// Call with `device` current.  Returns nullptr and sets *e on failure.
unsigned char* scratch_address(int device, hipError_t* e) {
    *e = hipSuccess;
    if (device < 0 || device >= kMaxDevices) { *e = hipErrorInvalidDevice; return nullptr; }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        *e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (*e != hipSuccess) return nullptr;
        g_scratch_ptr[device] = static_cast<unsigned char*>(p);
    }
    return g_scratch_ptr[device];
}

struct EnvTables {
    float4* img = nullptr;
    float4 *cond_pack = nullptr, *marg_pack = nullptr;
    float* cond_int = nullptr;
};

}  // namespace

struct rt_ctx {
    int device = 0;
    int num_cu = 256;
    hipStream_t stream = nullptr;          // main stream: extend / shade chain, rt_trace_*
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;

    rt_params params{};
    int tile = 64, tiles_x = 0, tiles_y = 0;
    uint32_t n_local_tiles = 0, n_slots = 0, n_valid = 0;   // of one pass
    uint32_t batch = 1;        // passes traced together
    uint32_t n_total = 0;      // n_slots * batch: slots of the path state

    // geometry
    LbvhResult bvh;            // host build (RT_FLAG_HOST_BVH)
    DeviceLbvh dbvh;           // GPU build (default)
    bool device_built = false;
    uint32_t n_triangles = 0, n_ref_nodes = 0;
    int32_t bvh_max_depth = 0;
    float4* d_nodes = nullptr;
    float4* d_tris = nullptr;
    int32_t* d_ovf = nullptr;
    WideBvh wide;              // 4-wide collapse of the same tree (RT_FLAG_WIDE_BVH)
    bool use_wide = false;
    TraceScene ts{};
    uint32_t grid_trace = 0;

    // shading
    float* d_tri_attr = nullptr;
    float4* d_materials = nullptr;
    DevLight* d_lights = nullptr;
    int32_t* d_material_texture = nullptr;
    int32_t* d_material_glossy = nullptr;
    bool any_glossy = false, any_area = false, any_point = false;
    DevTexture* d_textures = nullptr;
    TexLevel* d_tex_levels = nullptr;
    uchar4* d_texels = nullptr;
    float4* d_duv = nullptr;
    uint32_t n_textures = 0;
    std::vector<EnvTables> env;
    ShadeScene ss{};
    int n_lights = 0;

    // path state
    std::vector<void*> allocs;
    std::vector<size_t> alloc_sizes;
    uint64_t device_bytes = 0;       // everything dev_alloc holds
    uint64_t path_state_bytes = 0;   // of that, what grows with slots x batch (path state and shadow entries)
    PathBuffers pb{};
    uint32_t* d_q0 = nullptr;
    // path state of the current / next bounce (rays, beta, slot ids), swapped after every bounce
    float* d_rays[2] = {nullptr, nullptr};
    float* d_beta[2] = {nullptr, nullptr};
    uint32_t* d_slot[2] = {nullptr, nullptr};
    int last_buf = 0;   // which of the two holds the rays of the deepest bounce of the last batch
    // [D+1] queue counts, then [D] shadow-queue counts
    uint32_t* d_counts = nullptr;
    uint32_t* d_counts_init = nullptr;  // template copied in at the start of a pass
    uint32_t n_counts = 0;
    ShadowQueue sq{};                      // shadow entries of the bounce being resolved
    ShadeArgs* d_shade_args = nullptr;     // [2]: k_shade's parameter blocks with path-state buffer 0 / 1 as the current one
    int32_t* d_light_list = nullptr;       // [L] all lights, then [n_uncached] the lights without cached primary visibility
    int n_uncached_lights = 0;
    uint32_t* d_pass_deferred_b = nullptr; // deferred list of the shadow launches
    unsigned long long* d_raystats = nullptr;  // [0] closest, [1] shadow, [2] / [3] of those, rays re-traced by the exact follow-up launches
    TraceWork* d_work = nullptr;               // work counters of the rt_trace_* launches
    uint32_t* d_deferred = nullptr;            // rays handed to the exact follow-up launch (rt_trace_*)
    uint32_t deferred_cap = 0;
    TraceWork* d_pass_work = nullptr;          // [D] closest + [D] shadow launches of a pass
    uint32_t n_pass_work = 0;
    uint32_t* d_pass_deferred = nullptr;       // n_slots entries, reused by the launches of a pass
    uint32_t grid_exact = 8;                   // blocks of the (almost always empty) follow-up launches
    uint32_t* d_slot_pixel = nullptr;
    std::vector<uint32_t> h_slot_pixel;

    // camera / accumulation
    CameraK cam{};
    bool have_camera = false;
    bool prim_valid = false;    // the per-camera cache (primary rays and hits, UV differences, point-light visibility) is filled
    uint32_t* d_prime_count = nullptr;   // n_valid, as a device word for the primary-ray launch
    uint64_t frames = 0;

    // rt_trace_* staging (host-pointer variants)
    float* d_tr_rays = nullptr;
    float* d_tr_out = nullptr;  // t,u,v planes + id0,id1 planes
    uint32_t tr_cap = 0;

    unsigned char* scratch_ptr = nullptr;      // this device's copy of the file-scope scratch array

    float last_trace_ms = 0.f, last_render_ms = 0.f, bvh_build_ms = 0.f;
};

namespace {

#define HIP_TRY(ctx, call)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            char b_[512];                                                                      \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            (ctx)->err = b_;                                                                   \
            return RT_ERR_HIP;                                                                 \
        }                                                                                      \
    } while (0)

template <typename T>
int dev_alloc(rt_ctx* c, T** p, size_t count) {
    void* q = nullptr;
    size_t bytes = (count ? count : 1) * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) {
        char b[256];
        snprintf(b, sizeof b, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        c->err = b;
        return RT_ERR_HIP;
    }
    c->allocs.push_back(q);
    c->alloc_sizes.push_back(bytes);
    c->device_bytes += bytes;
    *p = static_cast<T*>(q);
    return RT_OK;
}

// Frees one dev_alloc'ed buffer early (grow-only staging that has been outgrown).
void dev_release(rt_ctx* c, void* p) {
    if (!p) return;
    for (size_t i = 0; i < c->allocs.size(); ++i)
        if (c->allocs[i] == p) {
            (void)hipFree(p);
            c->device_bytes -= c->alloc_sizes[i];
            c->allocs.erase(c->allocs.begin() + (long)i);
            c->alloc_sizes.erase(c->alloc_sizes.begin() + (long)i);
            return;
        }
}

template <typename T>
int dev_upload(rt_ctx* c, T** p, const T* host, size_t count) {
    int r = dev_alloc(c, p, count);
    if (r) return r;
    if (count) HIP_TRY(c, hipMemcpy(*p, host, count * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

inline const float* strided(const float* base, uint32_t stride, uint32_t i) {
    return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)i * stride);
}

// Distribution1D's constructor, src/samplers/Distributions.cpp:7-22
float build_cdf(const float* func, int n, float* cdf) {
    cdf[0] = 0.f;
    for (int i = 1; i <= n; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / (float)n;
    float func_int = cdf[n];
    if (func_int == 0) {
        for (int i = 1; i <= n; ++i) cdf[i] = (float)i / (float)n;
    } else {
        for (int i = 1; i <= n; ++i) cdf[i] /= func_int;
    }
    return func_int;
}

// Guide table of a CDF with n+1 entries: guide[b] = number of entries <= b/n,
// b = 0..n.  The partition point findInterval looks for (src/scene/Common.hpp:13-25)
// is non-decreasing in u, so for u in bucket b it lies between two guide
// entries and the search only has to cover that stretch.
void build_guide(const float* cdf, int n, int32_t* guide) {
    int p = 0;
    for (int b = 0; b <= n; ++b) {
        const float u = (float)b / (float)n;
        while (p <= n && cdf[p] <= u) ++p;
        guide[b] = p;
    }
}

// EnvLight's constructor, src/lights/EnvLight.cpp:8-26.  The glm::mat3 there
// is filled column-major from a row-major literal, so the luminance row it
// applies is (0.3575761, 0.7151522, 0.1191920); the table image is H wide
// and W high, written at (y, x), and handed to Distribution2D as W rows of H.
int build_env(rt_ctx* c, const rt_light& src, DevLight& dst, EnvTables& tab) {
    const int W = src.env_w, H = src.env_h;
    if (!src.env_rgba || W < 2 || H < 2) {
        c->err = "env light needs an image of at least 2x2";
        return RT_ERR_ARG;
    }
    std::vector<float> table((size_t)W * H), cdf((size_t)W * (H + 1)), cint(W), mcdf(W + 1);
    for (int y = 0; y < H; ++y) {
        float sinTheta = std::sin((float)M_PI * (y + .5f) / (float)H);
        for (int x = 0; x < W; ++x) {
            const float* px = src.env_rgba + ((size_t)y * W + x) * 4;
            float lum = 0.3575761f * px[0] + 0.7151522f * px[1] + 0.1191920f * px[2];
            table[(size_t)x * H + y] = sinTheta * lum;
        }
    }
    for (int v = 0; v < W; ++v) cint[v] = build_cdf(&table[(size_t)v * H], H, &cdf[(size_t)v * (H + 1)]);
    float mint = build_cdf(cint.data(), W, mcdf.data());
    // packed records (DistEntry layout of shade.h): cdf pair, function value and the guide bracket of bucket i
    if (W > 65534 || H > 65534) {
        c->err = "env image larger than 65534 in one direction";
        return RT_ERR_LIMIT;
    }
    std::vector<int32_t> guide;
    auto pack_rows = [&](const float* func, const float* cdfp, int n, float4* out) {
        guide.resize(n + 1);
        build_guide(cdfp, n, guide.data());
        for (int i = 0; i <= n; ++i) {
            const uint32_t lo = (uint32_t)guide[i > 0 ? i - 1 : 0], hi = (uint32_t)guide[i + 2 < n ? i + 2 : n];
            const uint32_t bits = lo | (hi << 16);
            float w;
            memcpy(&w, &bits, 4);
            out[i] = make_float4(cdfp[i], i < n ? cdfp[i + 1] : cdfp[i], i < n ? func[i] : 0.f, w);
        }
    };
    std::vector<float4> cpack((size_t)W * (H + 1)), mpack(W + 1);
    for (int v = 0; v < W; ++v) pack_rows(&table[(size_t)v * H], &cdf[(size_t)v * (H + 1)], H, &cpack[(size_t)v * (H + 1)]);
    pack_rows(cint.data(), mcdf.data(), W, mpack.data());
    int r;
    if ((r = dev_upload(c, &tab.img, reinterpret_cast<const float4*>(src.env_rgba), (size_t)W * H))) return r;
    if ((r = dev_upload(c, &tab.cond_pack, cpack.data(), cpack.size()))) return r;
    if ((r = dev_upload(c, &tab.marg_pack, mpack.data(), mpack.size()))) return r;
    if ((r = dev_upload(c, &tab.cond_int, cint.data(), cint.size()))) return r;
    dst.W = W; dst.H = H;
    dst.img = tab.img;
    dst.cond_pack = tab.cond_pack; dst.cond_int = tab.cond_int;
    dst.marg_pack = tab.marg_pack; dst.marg_int = mint;
    return RT_OK;
}

uint32_t grid_for(const rt_ctx* c, uint32_t n) {
    uint32_t g = (n + RT_TBLOCK - 1) / RT_TBLOCK;
    if (g > c->grid_trace) g = c->grid_trace;
    return g ? g : 1;
}

int ensure_trace_staging(rt_ctx* c, uint32_t n) {
    if (n <= c->tr_cap) return RT_OK;
    // grow-only staging for the host-pointer entry points; the outgrown pair is idle here (every entry point that
    // uses it waits for its copies before it returns)
    int r;
    dev_release(c, c->d_tr_rays); c->d_tr_rays = nullptr;
    dev_release(c, c->d_tr_out); c->d_tr_out = nullptr;
    c->tr_cap = 0;
    if ((r = dev_alloc(c, &c->d_tr_rays, (size_t)n * 6))) return r;
    if ((r = dev_alloc(c, &c->d_tr_out, (size_t)n * 5))) return r;
    c->tr_cap = n;
    return RT_OK;
}

// The deferred list of the rt_trace_* launches must be able to hold every ray.
int ensure_deferred(rt_ctx* c, uint32_t n) {
    if (n <= c->deferred_cap) return RT_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // an earlier launch may still append to the old list
    dev_release(c, c->d_deferred); c->d_deferred = nullptr;
    c->deferred_cap = 0;
    int r = dev_alloc(c, &c->d_deferred, n);
    if (r) return r;
    c->deferred_cap = n;
    return RT_OK;
}

int launch_closest(rt_ctx* c, uint32_t n, const float* d_rays, float* d_t, int32_t* d_id0, int32_t* d_id1, float* d_u,
                   float* d_v, uint32_t repeat) {
    if (repeat == 0) repeat = 1;
    if (int r = ensure_deferred(c, n)) return r;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    for (uint32_t r = 0; r < repeat; ++r) {
        HIP_TRY(c, hipMemsetAsync(c->d_work, 0, sizeof(TraceWork), c->stream));
        if (c->use_wide)
            hipLaunchKernelGGL(k_trace_closest_wide, dim3(grid_for(c, n)), dim3(RT_TBLOCK), 0, c->stream, c->ts, d_rays, n, n, c->d_work,
                               c->d_deferred, d_t, d_id0, d_id1, d_u, d_v, c->d_raystats);
        else
            hipLaunchKernelGGL(k_trace_closest<false>, dim3(grid_for(c, n)), dim3(RT_TBLOCK), 0, c->stream, c->ts, d_rays, n, n, c->d_work,
                               c->d_deferred, d_t, d_id0, d_id1, d_u, d_v, c->d_raystats);
        hipLaunchKernelGGL(k_trace_closest<true>, dim3(c->grid_exact), dim3(RT_TBLOCK), 0, c->stream, c->ts, d_rays, n, n, c->d_work,
                           c->d_deferred, d_t, d_id0, d_id1, d_u, d_v, c->d_raystats);
    }
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last_trace_ms = ms / (float)repeat;
    return RT_OK;
}

int launch_occluded(rt_ctx* c, uint32_t n, const float* d_rays, int32_t* d_status, uint32_t repeat, const float* d_tmax = nullptr) {
    if (repeat == 0) repeat = 1;
    if (int r = ensure_deferred(c, n)) return r;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    for (uint32_t r = 0; r < repeat; ++r) {
        HIP_TRY(c, hipMemsetAsync(c->d_work, 0, sizeof(TraceWork), c->stream));
        if (d_tmax) {
            if (c->use_wide)
                hipLaunchKernelGGL(k_trace_occluded_wide<true>, dim3(grid_for(c, n)), dim3(RT_TBLOCK), 0, c->stream, c->ts, d_rays, d_tmax,
                                   n, n, c->d_work, c->d_deferred, d_status);
            else
                hipLaunchKernelGGL(k_trace_occluded_bounded<false>, dim3(grid_for(c, n)), dim3(RT_TBLOCK), 0, c->stream, c->ts, d_rays, d_tmax,
                                   n, n, c->d_work, c->d_deferred, d_status);
            hipLaunchKernelGGL(k_trace_occluded_bounded<true>, dim3(c->grid_exact), dim3(RT_TBLOCK), 0, c->stream, c->ts, d_rays, d_tmax, n,
                               n, c->d_work, c->d_deferred, d_status);
            continue;
        }
        if (c->use_wide)
            hipLaunchKernelGGL(k_trace_occluded_wide<false>, dim3(grid_for(c, n)), dim3(RT_TBLOCK), 0, c->stream, c->ts, d_rays, nullptr, n, n,
                               c->d_work, c->d_deferred, d_status);
        else
            hipLaunchKernelGGL(k_trace_occluded<false>, dim3(grid_for(c, n)), dim3(RT_TBLOCK), 0, c->stream, c->ts, d_rays, n, n, c->d_work,
                               c->d_deferred, d_status);
        hipLaunchKernelGGL(k_trace_occluded<true>, dim3(c->grid_exact), dim3(RT_TBLOCK), 0, c->stream, c->ts, d_rays, n, n, c->d_work,
                           c->d_deferred, d_status);
    }
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last_trace_ms = ms / (float)repeat;
    return RT_OK;
}

// nb sample passes traced as one wavefront: the GPU form of nb calls of
// Integrator::update (src/cpu/Integrator.cpp:26-67).  Slot b * n_slots + s is
// film slot s of pass (frames + b).
int run_pass(rt_ctx* c, uint32_t nb) {
    const int D = c->params.max_depth, L = c->n_lights;
    const uint32_t pass = (uint32_t)c->frames;
    const uint64_t seed_mixed = [&] {
        uint64_t z = c->params.seed + 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }();
    HIP_TRY(c, hipMemcpyAsync(c->d_counts, c->d_counts_init + (size_t)(nb - 1) * c->n_counts, c->n_counts * sizeof(uint32_t),
                              hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_pass_work, 0, c->n_pass_work * sizeof(TraceWork), c->stream));
    const uint32_t live = nb * c->n_slots;
    const uint32_t gslots = (live + RT_BLOCK - 1) / RT_BLOCK;
    const uint32_t g1 = (c->n_slots + RT_BLOCK - 1) / RT_BLOCK;
    c->pb.rays = c->d_rays[0]; c->pb.beta = c->d_beta[0]; c->pb.slot = c->d_slot[0];
    c->pb.rays_next = c->d_rays[1]; c->pb.beta_next = c->d_beta[1]; c->pb.slot_next = c->d_slot[1];
    const uint32_t gtrace = grid_for(c, nb * c->n_valid);
    const uint32_t gshade = std::min<uint32_t>(gslots, (uint32_t)c->num_cu * 8u);
    TraceWork* work = c->d_pass_work;
    // The first pass after a camera change fills the per-camera cache: the camera ray has no jitter
    // (src/cpu/PathTracer.cpp:43-50), so primary rays, their hits and the quad UV differences are those of every
    // later pass; they are traced once per film slot, not once per pass.
    const int prime = c->prim_valid ? 0 : 1;
    if (prime) {
        hipLaunchKernelGGL(k_generate, dim3(g1), dim3(RT_BLOCK), 0, c->stream, c->pb, c->cam, c->n_slots);
        // the first n_valid entries of d_q0 are the film slots of the batch's first pass
        if (c->use_wide)
            hipLaunchKernelGGL(k_extend_wide, dim3(grid_for(c, c->n_valid)), dim3(RT_TBLOCK), 0, c->stream, c->ts, c->pb.prim_rays, c->n_slots,
                               c->d_q0, c->d_prime_count, work + D, c->d_pass_deferred, c->pb.prim_t, c->pb.prim_id0, c->pb.prim_id1,
                               c->pb.prim_u, c->pb.prim_v, c->d_raystats);
        else
            hipLaunchKernelGGL(k_extend<false>, dim3(grid_for(c, c->n_valid)), dim3(RT_TBLOCK), 0, c->stream, c->ts, c->pb.prim_rays, c->n_slots,
                               c->d_q0, c->d_prime_count, work + D, c->d_pass_deferred, c->pb.prim_t, c->pb.prim_id0, c->pb.prim_id1,
                               c->pb.prim_u, c->pb.prim_v, c->d_raystats);
        hipLaunchKernelGGL(k_extend<true>, dim3(c->grid_exact), dim3(RT_TBLOCK), 0, c->stream, c->ts, c->pb.prim_rays, c->n_slots, c->d_q0,
                           c->d_prime_count, work + D, c->d_pass_deferred, c->pb.prim_t, c->pb.prim_id0, c->pb.prim_id1, c->pb.prim_u,
                           c->pb.prim_v, c->d_raystats);
        if (c->n_textures && c->n_local_tiles)
            hipLaunchKernelGGL(k_quad_duv, dim3(c->n_local_tiles), dim3(RT_BLOCK), 0, c->stream, c->pb, c->d_tri_attr, c->d_duv, c->tile,
                               (float)c->params.width, (float)c->params.height);
    }
    // Per bounce: shade(d) -> ONE traversal launch carrying the shadow rays of bounce d and the closest-hit rays of
    // bounce d+1 (both are ready once shade(d) has run) -> accumulate(d).  One persistent launch
    // per bounce means one drain per bounce; everything is ordered on one stream
    // (k_accumulate beside k_shade on a second stream measured 2 % slower).
    // Bounce 0 walks the slots inside the film (d_q0) with path index = slot;
    // from bounce 1 on the paths are the first counts[d] entries of the
    // current arrays (queue == nullptr).
    const uint32_t* qin = c->d_q0;
    int cur = 0;
    for (int d = 0; d < D; ++d) {
        uint32_t* cnt_in = c->d_counts + d;
        uint32_t* cnt_out = c->d_counts + d + 1;
        PathBuffers pbd = c->pb;
        pbd.rays = c->d_rays[cur]; pbd.beta = c->d_beta[cur]; pbd.slot = c->d_slot[cur];
        pbd.rays_next = c->d_rays[cur ^ 1]; pbd.beta_next = c->d_beta[cur ^ 1]; pbd.slot_next = c->d_slot[cur ^ 1];
        if (c->any_glossy)
            hipLaunchKernelGGL(k_shade<true>, dim3(gshade), dim3(RT_BLOCK), 0, c->stream, c->d_shade_args + cur, d, D, prime, seed_mixed, pass, qin,
                               cnt_in, cnt_out, c->d_raystats + 1);
        else
            hipLaunchKernelGGL(k_shade<false>, dim3(gshade), dim3(RT_BLOCK), 0, c->stream, c->d_shade_args + cur, d, D, prime, seed_mixed, pass, qin,
                               cnt_in, cnt_out, c->d_raystats + 1);
        TraceWork* w = work + d;
        // the shadow entries this bounce's traversal launch walks (at bounce 0 the cached ones are already resolved)
        ShadowQueue sqd = c->sq;
        const bool cached0 = d == 0 && !prime;
        sqd.trace_lights = c->d_light_list + (cached0 ? L : 0);
        sqd.n_trace = cached0 ? c->n_uncached_lights : L;
        if (d + 1 < D) {
            if (c->use_wide)
                hipLaunchKernelGGL((k_bounce<false, true>), dim3(gtrace), dim3(RT_TBLOCK), 0, c->stream, c->ts, pbd, sqd, L, qin, cnt_in, cnt_out, w,
                                   c->d_pass_deferred_b, c->d_raystats);
            else
                hipLaunchKernelGGL((k_bounce<false, false>), dim3(gtrace), dim3(RT_TBLOCK), 0, c->stream, c->ts, pbd, sqd, L, qin, cnt_in, cnt_out, w,
                                   c->d_pass_deferred_b, c->d_raystats);
            hipLaunchKernelGGL((k_bounce<true, false>), dim3(c->grid_exact), dim3(RT_TBLOCK), 0, c->stream, c->ts, pbd, sqd, L, qin, cnt_in, cnt_out,
                               w, c->d_pass_deferred_b, c->d_raystats);
        } else if (L > 0) {
            if (c->any_area) {   // shadow rays towards area lights stop at the light
                if (c->use_wide)
                    hipLaunchKernelGGL((k_shadow<false, true, true>), dim3(gtrace), dim3(RT_TBLOCK), 0, c->stream, c->ts, pbd, sqd, L, qin, cnt_in, w,
                                       c->d_pass_deferred_b, c->d_raystats + 1);
                else
                    hipLaunchKernelGGL((k_shadow<false, true, false>), dim3(gtrace), dim3(RT_TBLOCK), 0, c->stream, c->ts, pbd, sqd, L, qin, cnt_in, w,
                                       c->d_pass_deferred_b, c->d_raystats + 1);
                hipLaunchKernelGGL((k_shadow<true, true>), dim3(c->grid_exact), dim3(RT_TBLOCK), 0, c->stream, c->ts, pbd, sqd, L, qin, cnt_in,
                                   w, c->d_pass_deferred_b, c->d_raystats + 1);
            } else {
                if (c->use_wide)
                    hipLaunchKernelGGL((k_shadow<false, false, true>), dim3(gtrace), dim3(RT_TBLOCK), 0, c->stream, c->ts, pbd, sqd, L, qin, cnt_in, w,
                                       c->d_pass_deferred_b, c->d_raystats + 1);
                else
                    hipLaunchKernelGGL((k_shadow<false, false, false>), dim3(gtrace), dim3(RT_TBLOCK), 0, c->stream, c->ts, pbd, sqd, L, qin, cnt_in, w,
                                       c->d_pass_deferred_b, c->d_raystats + 1);
                hipLaunchKernelGGL((k_shadow<true, false>), dim3(c->grid_exact), dim3(RT_TBLOCK), 0, c->stream, c->ts, pbd, sqd, L, qin, cnt_in,
                                   w, c->d_pass_deferred_b, c->d_raystats + 1);
            }
        }
        if (d == 0 && prime && c->any_point)
            hipLaunchKernelGGL(k_cache_visibility, dim3(g1), dim3(RT_BLOCK), 0, c->stream, pbd, c->sq, c->d_lights, L);
        if (L > 0) hipLaunchKernelGGL(k_accumulate, dim3(gshade), dim3(RT_BLOCK), 0, c->stream, pbd, c->sq, L, qin, cnt_in);
        if (d + 1 < D) {
            qin = nullptr;
            cur ^= 1;
        }
    }
    c->last_buf = cur;
    // the reference counts passes in a float (src/cpu/Integrator.hpp:50, `_frames++`), which stops growing at 2^24
    const float frames_f = (float)std::min<uint64_t>(c->frames, 1ull << 24);
    hipLaunchKernelGGL(k_film, dim3((c->n_slots + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, c->stream, c->pb, frames_f, nb);
    HIP_TRY(c, hipGetLastError());
    c->frames += nb;
    c->prim_valid = true;
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_version(void) { return "rt_hip 0.1 gfx950"; }

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void rt_destroy(rt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void* p : c->allocs) (void)hipFree(p);
    c->dbvh.release();
    c->wide.release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int create_impl(rt_ctx* c, const rt_scene_desc* sd, const rt_params* pr) {
    if (!sd->pos || !sd->indices || !sd->tri_model || sd->n_indices < 3 || sd->n_indices % 3 != 0 || sd->n_vertices == 0) {
        c->err = "scene needs positions, indices (a multiple of 3, at least one triangle) and tri_model";
        return RT_ERR_ARG;
    }
    if (!sd->normal || !sd->materials || sd->n_models == 0) {
        c->err = "scene needs normals and at least one material";
        return RT_ERR_ARG;
    }
    if ((sd->tangent == nullptr) != (sd->binormal == nullptr)) {
        c->err = "tangent and binormal must both be given or both be null";
        return RT_ERR_ARG;
    }
    if (pr->width <= 0 || pr->height <= 0 || pr->max_depth <= 0 || pr->max_depth > 64) {
        c->err = "width, height must be positive and max_depth in 1..64";
        return RT_ERR_ARG;
    }
    // the sampler gives the dimension d * (4 L + 2) + 4 l + k sixteen bits
    if ((uint64_t)pr->max_depth * (4ull * sd->n_lights + 2ull) > 0x10000ull) {
        c->err = "max_depth * (4 * n_lights + 2) exceeds 65536 sampler dimensions";
        return RT_ERR_LIMIT;
    }
    if ((uint64_t)pr->width * (uint64_t)pr->height >= 0xFFFFFFFFull) {
        c->err = "film too large";
        return RT_ERR_LIMIT;
    }
    c->params = *pr;
    c->tile = pr->tile_size > 0 ? pr->tile_size : 64;
    if (c->tile % 8 != 0) {
        c->err = "tile_size must be a multiple of 8";
        return RT_ERR_ARG;
    }
    const int world = pr->world > 0 ? pr->world : 1;
    const int rank = pr->rank;
    if (rank < 0 || rank >= world) {
        c->err = "rank must be in [0, world)";
        return RT_ERR_ARG;
    }
    if (sd->n_textures && !sd->textures) {
        c->err = "n_textures > 0 but textures is null";
        return RT_ERR_ARG;
    }
    for (uint32_t t = 0; t < sd->n_textures; ++t)
        if (!sd->textures[t].rgba8 || sd->textures[t].width < 1 || sd->textures[t].height < 1 || sd->textures[t].width > 32768 ||
            sd->textures[t].height > 32768) {
            c->err = "texture needs texels and a size of 1..32768 in each direction";
            return RT_ERR_ARG;
        }
    for (uint32_t m = 0; m < sd->n_models; ++m)
        if (sd->materials[m].glossy < RT_GLOSSY_NONE || sd->materials[m].glossy > RT_GLOSSY_GGX) {
            c->err = "material glossy must be 0 (none), 1 (Beckmann) or 2 (GGX)";
            return RT_ERR_ARG;
        }
    for (uint32_t m = 0; m < sd->n_models; ++m)
        if (sd->materials[m].texture < -1 || sd->materials[m].texture >= (int32_t)sd->n_textures) {
            c->err = "material texture index out of range";
            return RT_ERR_ARG;
        }
    const uint32_t T = sd->n_indices / 3;
    for (uint32_t i = 0; i < sd->n_indices; ++i)
        if (sd->indices[i] >= sd->n_vertices) {
            c->err = "index out of range";
            return RT_ERR_ARG;
        }
    for (uint32_t t = 0; t < T; ++t)
        if (sd->tri_model[t] < 0 || (uint32_t)sd->tri_model[t] >= sd->n_models) {
            c->err = "tri_model out of range";
            return RT_ERR_ARG;
        }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        c->err = "no HIP device available";
        return RT_ERR_NO_DEVICE;
    }
    if (pr->device < 0 || pr->device >= ndev) {
        c->err = "device ordinal out of range";
        return RT_ERR_ARG;
    }
    c->device = pr->device;
    HIP_TRY(c, hipSetDevice(c->device));
    hipDeviceProp_t prop;
    HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreate(&c->ev0));
    HIP_TRY(c, hipEventCreate(&c->ev1));
    {
        hipError_t se;
        c->scratch_ptr = scratch_address(c->device, &se);
        HIP_TRY(c, se);
    }

    // ---- BVH: built on the GPU unless the caller asks for the host builder
    auto t0 = std::chrono::steady_clock::now();
    int r;
    const uint32_t pstride = sd->pos_stride ? sd->pos_stride : 12;
    float rmin[3], rmax[3];
    int32_t root_link;
    size_t n_node_slots;
    c->n_triangles = T;
    c->n_ref_nodes = 2 * T - 1;
    if (pr->flags & RT_FLAG_HOST_BVH) {
        build_lbvh(sd->pos, pstride, sd->indices, sd->n_indices, sd->tri_model, c->bvh, RT_TOP_LEVELS);
        if ((r = dev_upload(c, &c->d_nodes, reinterpret_cast<const float4*>(c->bvh.nodes.data()), c->bvh.nodes.size() * 4))) return r;
        if ((r = dev_upload(c, &c->d_tris, reinterpret_cast<const float4*>(c->bvh.tris.data()), c->bvh.tris.size() * 3))) return r;
        root_link = c->bvh.root_link;
        n_node_slots = c->bvh.nodes.size();
        c->bvh_max_depth = c->bvh.max_depth;
        for (int k = 0; k < 3; ++k) { rmin[k] = c->bvh.root_min[k]; rmax[k] = c->bvh.root_max[k]; }
    } else {
        std::vector<float> packed((size_t)sd->n_vertices * 3);
        for (uint32_t v = 0; v < sd->n_vertices; ++v) {
            const float* q = strided(sd->pos, pstride, v);
            packed[3 * (size_t)v] = q[0]; packed[3 * (size_t)v + 1] = q[1]; packed[3 * (size_t)v + 2] = q[2];
        }
        // build inputs live only for the build: freed on every way out
        struct Scratch {
            void* p[3] = {nullptr, nullptr, nullptr};
            ~Scratch() { for (void* q : p) if (q) (void)hipFree(q); }
        } tmp;
        HIP_TRY(c, hipMalloc(&tmp.p[0], packed.size() * sizeof(float)));
        HIP_TRY(c, hipMalloc(&tmp.p[1], (size_t)sd->n_indices * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&tmp.p[2], (size_t)T * sizeof(int32_t)));
        float* d_pos = static_cast<float*>(tmp.p[0]);
        uint32_t* d_idx = static_cast<uint32_t*>(tmp.p[1]);
        int32_t* d_tm = static_cast<int32_t*>(tmp.p[2]);
        HIP_TRY(c, hipMemcpy(d_pos, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(d_idx, sd->indices, (size_t)sd->n_indices * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(d_tm, sd->tri_model, (size_t)T * sizeof(int32_t), hipMemcpyHostToDevice));
        std::string berr;
        int br = build_lbvh_device(d_pos, d_idx, sd->n_indices, d_tm, RT_TOP_LEVELS, c->stream, &c->dbvh, &berr);
        if (br != 0) {
            c->err = "GPU LBVH build: " + berr;
            return RT_ERR_HIP;
        }
        c->device_built = true;
        c->d_nodes = c->dbvh.nodes;
        c->d_tris = c->dbvh.tris;
        root_link = c->dbvh.root_link;
        n_node_slots = c->dbvh.n_node_slots;
        c->bvh_max_depth = c->dbvh.max_depth;
        for (int k = 0; k < 3; ++k) { rmin[k] = c->dbvh.root_min[k]; rmax[k] = c->dbvh.root_max[k]; }
    }
    // ---- the 4-wide collapse of that tree, made on the device from the binary device nodes
    c->ts.wnodes = nullptr;
    c->ts.n_wnodes = 0;
    c->ts.wroot = root_link;
    if (pr->flags & RT_FLAG_WIDE_BVH) {
        std::string werr;
        if (build_wide_bvh(c->d_nodes, root_link, T - 1, c->stream, &c->wide, &werr) != 0) {
            c->err = "wide BVH collapse: " + werr;
            return RT_ERR_HIP;
        }
        if ((uint64_t)c->wide.n_nodes * 128ull > 0xFFFFFFFFull) {
            c->err = "scene too large: the wide node array reaches 4 GiB";
            return RT_ERR_LIMIT;
        }
        c->use_wide = true;
        c->ts.wnodes = c->wide.nodes;
        c->ts.n_wnodes = (int32_t)c->wide.n_nodes;
        c->ts.wroot = c->wide.n_nodes ? 0 : root_link;
        c->device_bytes += (uint64_t)c->wide.capacity * 128ull;
    }
    c->bvh_build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();

    // persistent traversal grid: as many workgroups per CU as LDS and registers admit
    int bpc = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, k_extend<false>, RT_TBLOCK, 0));
    if (bpc < 1) bpc = 1;
    c->grid_trace = (uint32_t)c->num_cu * (uint32_t)bpc;
    c->grid_exact = 8;
    // pending entries of a lane: at most one per binary level, at most three per wide level
    const uint32_t ovf_levels = (uint32_t)std::max(1, std::max(c->bvh_max_depth + 2 - RT_STACK_LDS, 3 * c->wide.depth + 2 + RT_WSPILL));
    const uint32_t ovf_stride = c->grid_trace * RT_TBLOCK;
    if ((uint64_t)ovf_levels * ovf_stride * 4ull > 0xFFFFFFFFull) {   // the kernels address it with 32-bit byte offsets
        c->err = "BVH too deep: the traversal overflow stack would reach 4 GiB";
        return RT_ERR_LIMIT;
    }
    if ((r = dev_alloc(c, &c->d_ovf, (size_t)ovf_levels * ovf_stride))) return r;
    // the traversal kernels address nodes (64 B) and triangles (48 B) with 32-bit byte offsets
    if ((uint64_t)n_node_slots * 64ull > 0xFFFFFFFFull || (uint64_t)T * 48ull > 0xFFFFFFFFull) {
        c->err = "scene too large: the node or triangle array reaches 4 GiB";
        return RT_ERR_LIMIT;
    }
    c->ts.nodes = c->d_nodes;
    c->ts.tris = c->d_tris;
    c->ts.root_link = root_link;
    c->ts.n_nodes = (int32_t)n_node_slots;
    c->ts.n_tris = (int32_t)T;
    for (int k = 0; k < 3; ++k) { c->ts.rmin[k] = rmin[k]; c->ts.rmax[k] = rmax[k]; }
    c->ts.ovf = c->d_ovf;
    c->ts.ovf_stride = ovf_stride;
    c->ts.ovf_levels = ovf_levels;
    c->ts.tame = 1;
    for (int k = 0; k < 3; ++k)
        if (!(std::fabs(rmin[k]) < 1e30f) || !(std::fabs(rmax[k]) < 1e30f)) c->ts.tame = 0;

    // ---- shading tables: per-triangle attribute blocks, 3 x (normal uv tangent binormal)
    {
        std::vector<float> attr((size_t)T * 36, 0.f);
        const bool frame = sd->tangent && sd->binormal;
        for (uint32_t t = 0; t < T; ++t)
            for (int v = 0; v < 3; ++v) {
                uint32_t vi = sd->indices[3 * t + v];
                float* a = &attr[(size_t)t * 36 + v * 11];
                const float* nn = strided(sd->normal, sd->normal_stride ? sd->normal_stride : 12, vi);
                a[0] = nn[0]; a[1] = nn[1]; a[2] = nn[2];
                if (sd->uv) {
                    const float* uu = strided(sd->uv, sd->uv_stride ? sd->uv_stride : 8, vi);
                    a[3] = uu[0]; a[4] = uu[1];
                }
                if (frame) {
                    const float* tt = strided(sd->tangent, sd->tangent_stride ? sd->tangent_stride : 12, vi);
                    const float* bb = strided(sd->binormal, sd->binormal_stride ? sd->binormal_stride : 12, vi);
                    a[5] = tt[0]; a[6] = tt[1]; a[7] = tt[2];
                    a[8] = bb[0]; a[9] = bb[1]; a[10] = bb[2];
                }
            }
        if ((r = dev_upload(c, &c->d_tri_attr, attr.data(), attr.size()))) return r;
        c->ss.have_frame = frame ? 1 : 0;
    }
    {
        std::vector<float4> mats(sd->n_models);
        for (uint32_t m = 0; m < sd->n_models; ++m)
            mats[m] = make_float4(sd->materials[m].diffuse_color[0], sd->materials[m].diffuse_color[1],
                                  sd->materials[m].diffuse_color[2], sd->materials[m].roughness);
        if ((r = dev_upload(c, &c->d_materials, mats.data(), mats.size()))) return r;
        std::vector<int32_t> mg(sd->n_models);
        for (uint32_t m = 0; m < sd->n_models; ++m) {
            mg[m] = sd->materials[m].glossy;
            if (mg[m] != RT_GLOSSY_NONE) c->any_glossy = true;
        }
        if (c->any_glossy && (r = dev_upload(c, &c->d_material_glossy, mg.data(), mg.size()))) return r;
    }
    // ---- diffuse maps: mip chains by 2x2 box average in integers, (a+b+c+d+2)>>2,
    // level sizes max(1, w>>1); all levels of all maps in one texel buffer
    c->n_textures = sd->n_textures;
    if (sd->n_textures) {
        std::vector<int32_t> mt(sd->n_models);
        for (uint32_t m = 0; m < sd->n_models; ++m) mt[m] = sd->materials[m].texture;
        std::vector<DevTexture> tex(sd->n_textures);
        std::vector<TexLevel> levels;
        std::vector<uchar4> texels;
        for (uint32_t t = 0; t < sd->n_textures; ++t) {
            int w = sd->textures[t].width, h = sd->textures[t].height;
            tex[t].first_level = (int32_t)levels.size();
            size_t prev = texels.size();
            levels.push_back({(uint32_t)prev, w, h});
            const uchar4* src0 = reinterpret_cast<const uchar4*>(sd->textures[t].rgba8);
            texels.insert(texels.end(), src0, src0 + (size_t)w * h);
            while (w > 1 || h > 1) {
                const int nw = std::max(1, w >> 1), nh = std::max(1, h >> 1);
                const size_t dst = texels.size();
                texels.resize(dst + (size_t)nw * nh);
                for (int y = 0; y < nh; ++y)
                    for (int x = 0; x < nw; ++x) {
                        const int x0 = std::min(2 * x, w - 1), x1 = std::min(2 * x + 1, w - 1);
                        const int y0 = std::min(2 * y, h - 1), y1 = std::min(2 * y + 1, h - 1);
                        const uchar4 a = texels[prev + (size_t)y0 * w + x0], b = texels[prev + (size_t)y0 * w + x1];
                        const uchar4 cc = texels[prev + (size_t)y1 * w + x0], d = texels[prev + (size_t)y1 * w + x1];
                        uchar4 o;
                        o.x = (unsigned char)((a.x + b.x + cc.x + d.x + 2) >> 2);
                        o.y = (unsigned char)((a.y + b.y + cc.y + d.y + 2) >> 2);
                        o.z = (unsigned char)((a.z + b.z + cc.z + d.z + 2) >> 2);
                        o.w = (unsigned char)((a.w + b.w + cc.w + d.w + 2) >> 2);
                        texels[dst + (size_t)y * nw + x] = o;
                    }
                if (texels.size() >= 0xFFFFFFFFull) {
                    c->err = "textures exceed 2^32 texels";
                    return RT_ERR_LIMIT;
                }
                levels.push_back({(uint32_t)dst, nw, nh});
                prev = dst; w = nw; h = nh;
            }
            tex[t].n_levels = (int32_t)levels.size() - tex[t].first_level;
        }
        if ((r = dev_upload(c, &c->d_material_texture, mt.data(), mt.size()))) return r;
        if ((r = dev_upload(c, &c->d_textures, tex.data(), tex.size()))) return r;
        if ((r = dev_upload(c, &c->d_tex_levels, levels.data(), levels.size()))) return r;
        if ((r = dev_upload(c, &c->d_texels, texels.data(), texels.size()))) return r;
    }
    c->n_lights = (int)sd->n_lights;
    c->ss.env_staged = -1;
    {
        std::vector<DevLight> lights(sd->n_lights);
        c->env.resize(sd->n_lights);
        for (uint32_t l = 0; l < sd->n_lights; ++l) {
            const rt_light& s = sd->lights[l];
            DevLight& d = lights[l];
            memset(&d, 0, sizeof d);
            d.type = s.type;
            d.px = s.pos[0]; d.py = s.pos[1]; d.pz = s.pos[2];
            d.ix = s.intensity[0]; d.iy = s.intensity[1]; d.iz = s.intensity[2];
            if (s.type == RT_LIGHT_ENV) {
                if ((r = build_env(c, s, d, c->env[l]))) return r;
                if (c->ss.env_staged < 0 && s.env_w + 1 <= RT_MARG_LDS) c->ss.env_staged = (int32_t)l;
            } else if (s.type == RT_LIGHT_AREA) {
                float e1[3], e2[3];
                for (int k = 0; k < 3; ++k) {
                    d.p0[k] = s.p0[k]; d.p1[k] = s.p1[k]; d.p2[k] = s.p2[k];
                    e1[k] = s.p1[k] - s.p0[k]; e2[k] = s.p2[k] - s.p0[k];
                }
                // glm::cross / glm::normalize operand order, rounded as the scalar reference does
                float cx = e1[1] * e2[2] - e2[1] * e1[2], cy = e1[2] * e2[0] - e2[2] * e1[0], cz = e1[0] * e2[1] - e2[0] * e1[1];
                float l2 = cx * cx + cy * cy + cz * cz;
                if (!(l2 > 0.f) || !std::isfinite(l2)) {
                    c->err = "area light triangle is degenerate";
                    return RT_ERR_ARG;
                }
                d.area = 0.5f * std::sqrt(l2);
                float il = 1.0f / std::sqrt(l2);
                d.n[0] = cx * il; d.n[1] = cy * il; d.n[2] = cz * il;
                c->any_area = true;
            } else if (s.type == RT_LIGHT_POINT) {
                c->any_point = true;
            } else {
                c->err = "unsupported light type (point = 0, area = 2 and env = 3 are implemented)";
                return RT_ERR_ARG;
            }
        }
        if ((r = dev_upload(c, &c->d_lights, lights.data(), lights.size()))) return r;
        // work lists of the traversal launches: every light; the lights whose shadow ray from the primary vertex is
        // not kept in the per-camera cache (everything but point lights)
        std::vector<int32_t> list;
        for (uint32_t l = 0; l < sd->n_lights; ++l) list.push_back((int32_t)l);
        for (uint32_t l = 0; l < sd->n_lights; ++l)
            if (sd->lights[l].type != RT_LIGHT_POINT) { list.push_back((int32_t)l); ++c->n_uncached_lights; }
        if ((r = dev_upload(c, &c->d_light_list, list.data(), list.size()))) return r;
    }
    c->ss.tri_attr = c->d_tri_attr;
    c->ss.materials = c->d_materials;
    c->ss.lights = c->d_lights;
    c->ss.n_lights = c->n_lights;
    c->ss.material_texture = c->d_material_texture;
    c->ss.material_glossy = c->d_material_glossy;
    c->ss.textures = c->d_textures;   // null without diffuse maps
    c->ss.tex_levels = c->d_tex_levels;
    c->ss.texels = c->d_texels;

    // ---- film tiles of this rank, slots in 8x8 pixel blocks inside each tile
    const int W = pr->width, H = pr->height, ts = c->tile;
    c->tiles_x = (W + ts - 1) / ts;
    c->tiles_y = (H + ts - 1) / ts;
    const uint32_t n_tiles = (uint32_t)c->tiles_x * (uint32_t)c->tiles_y;
    c->n_local_tiles = n_tiles > (uint32_t)rank ? (n_tiles - rank + world - 1) / world : 0;
    c->n_slots = c->n_local_tiles * ts * ts;
    c->h_slot_pixel.assign(c->n_slots, 0xFFFFFFFFu);
    std::vector<uint32_t> q0;
    q0.reserve(c->n_slots);
    const int bpr = ts / 8;
    for (uint32_t k = 0; k < c->n_local_tiles; ++k) {
        uint32_t t = rank + k * world;
        int tx = t % c->tiles_x, ty = t / c->tiles_x;
        for (int i = 0; i < ts * ts; ++i) {
            int blk = i >> 6, li = i & 63;
            int x = tx * ts + (blk % bpr) * 8 + (li & 7);
            int y = ty * ts + (blk / bpr) * 8 + (li >> 3);
            uint32_t s = k * ts * ts + i;
            if (x < W && y < H) {
                c->h_slot_pixel[s] = (uint32_t)y * W + x;
                q0.push_back(s);
            }
        }
    }
    c->n_valid = (uint32_t)q0.size();
    // ---- passes traced together: as many as asked for (default 32) within the
    // path-state budget and 32-bit entry indices
    {
        const uint64_t L1 = (uint64_t)std::max<uint32_t>(1, sd->n_lights);
        // bytes of path state per slot (rays, beta and slot ids twice; hit; L; vertex; status; pixel id; queue entry;
        // deferred list) and of shadow entries per slot and light (direction,
        // contributions, state, deferred list, far bound with area lights)
        const uint64_t per_slot = 144 + (41 + (c->any_area ? 4 : 0)) * L1;
        uint64_t S = pr->batch ? pr->batch : 32;
        const uint64_t one = std::max<uint64_t>(1, (uint64_t)c->n_slots);
        // budget: 64 GiB, or 60 % of what is free on the device right now when that is less (other contexts may
        // share the GPU); the batch only changes speed, never the film
        uint64_t budget = 64ull << 30;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min<uint64_t>(budget, (uint64_t)free_b / 10 * 6);
        else (void)hipGetLastError();
        S = std::min<uint64_t>(S, std::max<uint64_t>(1, budget / (per_slot * one)));
        S = std::min<uint64_t>(S, std::max<uint64_t>(1, 0x7FFFFFFFull / (L1 * one)));
        S = std::min<uint64_t>(S, 64);
        c->batch = (uint32_t)S;
        c->n_total = c->batch * c->n_slots;
    }
    const size_t n = c->n_total;
    const uint64_t bytes_before_state = c->device_bytes;
    {
        std::vector<uint32_t> sp(n ? n : 1, 0xFFFFFFFFu), qa;
        qa.reserve((size_t)c->batch * q0.size());
        for (uint32_t b = 0; b < c->batch; ++b) {
            std::copy(c->h_slot_pixel.begin(), c->h_slot_pixel.end(), sp.begin() + (size_t)b * c->n_slots);
            for (uint32_t s : q0) qa.push_back(b * c->n_slots + s);
        }
        if ((r = dev_upload(c, &c->d_slot_pixel, sp.data(), n))) return r;
        if ((r = dev_upload(c, &c->d_q0, qa.data(), qa.size()))) return r;
    }
    for (int k = 0; k < 2; ++k) {
        if ((r = dev_alloc(c, &c->d_rays[k], 6 * n))) return r;
        if ((r = dev_alloc(c, &c->d_beta[k], 3 * n))) return r;
        if ((r = dev_alloc(c, &c->d_slot[k], n))) return r;
    }
    c->pb.rays = c->d_rays[0];
    if ((r = dev_alloc(c, &c->pb.hit_t, n))) return r;
    if ((r = dev_alloc(c, &c->pb.hit_id0, n))) return r;
    if ((r = dev_alloc(c, &c->pb.hit_id1, n))) return r;
    if ((r = dev_alloc(c, &c->pb.hit_u, n))) return r;
    if ((r = dev_alloc(c, &c->pb.hit_v, n))) return r;
    if ((r = dev_alloc(c, &c->pb.L, 3 * n))) return r;
    if ((r = dev_alloc(c, &c->pb.vpos, 3 * n))) return r;
    if ((r = dev_alloc(c, &c->pb.status, n))) return r;
    if ((r = dev_alloc(c, &c->pb.film, (size_t)c->n_slots))) return r;
    HIP_TRY(c, hipMemset(c->pb.film, 0, (c->n_slots ? c->n_slots : 1) * sizeof(float4)));
    {
        const size_t n1a = c->n_slots ? c->n_slots : 1;
        if ((r = dev_alloc(c, &c->pb.prim_rays, 6 * n1a))) return r;
        if ((r = dev_alloc(c, &c->pb.prim_t, n1a))) return r;
        if ((r = dev_alloc(c, &c->pb.prim_id0, n1a))) return r;
        if ((r = dev_alloc(c, &c->pb.prim_id1, n1a))) return r;
        if ((r = dev_alloc(c, &c->pb.prim_u, n1a))) return r;
        if ((r = dev_alloc(c, &c->pb.prim_v, n1a))) return r;
        if (c->any_point && (r = dev_alloc(c, &c->pb.prim_vis, (size_t)sd->n_lights * n1a))) return r;
        HIP_TRY(c, hipMemset(c->pb.prim_rays, 0, 6 * n1a * sizeof(float)));
        if (n) HIP_TRY(c, hipMemset(c->pb.status, 0, n * sizeof(uint32_t)));   // slots outside the film stay 0 for good
        const uint32_t nv = c->n_valid;
        if ((r = dev_upload(c, &c->d_prime_count, &nv, 1))) return r;
    }
    c->pb.slot_pixel = c->d_slot_pixel;
    c->pb.n = (uint32_t)n;
    c->pb.n1 = c->n_slots ? c->n_slots : 1;
    if (c->n_textures) {
        if ((r = dev_alloc(c, &c->d_duv, (size_t)c->n_slots))) return r;
        HIP_TRY(c, hipMemset(c->d_duv, 0, (c->n_slots ? c->n_slots : 1) * sizeof(float4)));   // dev_alloc rounds an empty array up to one element
        c->ss.duv = c->d_duv;
    }

    const int D = pr->max_depth, L = c->n_lights;
    c->n_counts = (uint32_t)(D + 1 + D);
    c->n_pass_work = (uint32_t)(2 * D);
    if ((r = dev_alloc(c, &c->d_pass_work, c->n_pass_work))) return r;
    if ((r = dev_alloc(c, &c->d_pass_deferred, n))) return r;
    // one template of the queue counters per batch size: entry 0 = paths the pass starts with
    std::vector<uint32_t> init((size_t)c->n_counts * c->batch, 0u);
    for (uint32_t b = 0; b < c->batch; ++b) init[(size_t)b * c->n_counts] = (b + 1) * c->n_valid;
    if ((r = dev_upload(c, &c->d_counts_init, init.data(), init.size()))) return r;
    if ((r = dev_alloc(c, &c->d_counts, c->n_counts))) return r;
    {
        const size_t cap = (size_t)(L > 0 ? L : 1) * n;
        if (cap >= 0xFFFFFFFFull) {
            c->err = "lights x pixels exceeds the shadow queue limit";
            return RT_ERR_LIMIT;
        }
        ShadowQueue& q = c->sq;
        q.cap = (uint32_t)cap;
        if ((r = dev_alloc(c, &q.dir, 3 * cap))) return r;
        q.tmax = nullptr;
        if (c->any_area && (r = dev_alloc(c, &q.tmax, cap))) return r;
        if ((r = dev_alloc(c, &q.contrib, 6 * cap))) return r;
        if ((r = dev_alloc(c, &q.state, cap))) return r;
        // the exact follow-up launch of a bounce can defer up to every shadow entry and every path
        if ((r = dev_alloc(c, &c->d_pass_deferred_b, cap + n))) return r;
    }
    c->path_state_bytes = c->device_bytes - bytes_before_state;
    {
        // k_shade's parameter blocks, one per choice of the current path-state buffer (run_pass swaps them every bounce)
        ShadeArgs sa[2];
        for (int k = 0; k < 2; ++k) {
            sa[k].pb = c->pb;
            sa[k].pb.rays = c->d_rays[k]; sa[k].pb.beta = c->d_beta[k]; sa[k].pb.slot = c->d_slot[k];
            sa[k].pb.rays_next = c->d_rays[k ^ 1]; sa[k].pb.beta_next = c->d_beta[k ^ 1]; sa[k].pb.slot_next = c->d_slot[k ^ 1];
            sa[k].sc = c->ss;
            sa[k].sq = c->sq;
        }
        if ((r = dev_upload(c, &c->d_shade_args, sa, 2))) return r;
    }
    if ((r = dev_alloc(c, &c->d_work, 1))) return r;
    if ((r = dev_alloc(c, &c->d_raystats, 4))) return r;
    HIP_TRY(c, hipMemset(c->d_raystats, 0, 4 * sizeof(unsigned long long)));
    HIP_TRY(c, hipDeviceSynchronize());
    return RT_OK;
}

int rt_create(const rt_scene_desc* scene, const rt_params* params, rt_ctx** out) {
    if (!scene || !params || !out) {
        g_create_error = "rt_create: null argument";
        return RT_ERR_ARG;
    }
    *out = nullptr;
    rt_ctx* c = new rt_ctx();
    int r = create_impl(c, scene, params);
    if (r != RT_OK) {
        g_create_error = c->err;
        rt_destroy(c);
        return r;
    }
    *out = c;
    return RT_OK;
}

int rt_set_camera(rt_ctx* c, const float pos[3], const float right[3], const float up[3], const float forward[3]) {
    if (!c || !pos || !right || !up || !forward) return RT_ERR_ARG;
    CameraK k;
    for (int i = 0; i < 3; ++i) { k.pos[i] = pos[i]; k.right[i] = right[i]; k.up[i] = up[i]; k.forward[i] = forward[i]; }
    k.W = c->params.width;
    k.H = c->params.height;
    // a moved camera restarts accumulation (src/cpu/Integrator.cpp:27-30)
    if (!c->have_camera || memcmp(&k, &c->cam, sizeof k) != 0) {
        c->frames = 0;
        c->prim_valid = false;   // primary hits, UV differences and point-light visibility belong to the old camera
    }
    c->cam = k;
    c->have_camera = true;
    return RT_OK;
}

int rt_reset_accumulation(rt_ctx* c) {
    if (!c) return RT_ERR_ARG;
    c->frames = 0;
    return RT_OK;
}

int rt_render(rt_ctx* c, uint32_t n_passes) {
    if (!c) return RT_ERR_ARG;
    if (!c->have_camera) {
        c->err = "rt_render: no camera set";
        return RT_ERR_STATE;
    }
    if (c->frames + n_passes > 0xFFFFFFFFull) {   // the sampler keys a draw by a 32-bit pass number
        c->err = "rt_render: more than 2^32 - 1 accumulated passes";
        return RT_ERR_LIMIT;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->n_slots == 0) { c->frames += n_passes; return RT_OK; }
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    for (uint32_t p = 0; p < n_passes;) {
        const uint32_t nb = std::min(c->batch, n_passes - p);
        int r = run_pass(c, nb);
        if (r) return r;
        p += nb;
    }
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    HIP_TRY(c, hipEventElapsedTime(&c->last_render_ms, c->ev0, c->ev1));
    return RT_OK;
}

int rt_read_film(rt_ctx* c, float* rgba_out) {
    if (!c || !rgba_out) return RT_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = c->n_slots;
    std::vector<float> slab(n * 4);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (n) HIP_TRY(c, hipMemcpy(slab.data(), c->pb.film, n * sizeof(float4), hipMemcpyDeviceToHost));
    memset(rgba_out, 0, (size_t)c->params.width * c->params.height * 4 * sizeof(float));
    for (size_t s = 0; s < n; ++s) {
        uint32_t pix = c->h_slot_pixel[s];
        if (pix != 0xFFFFFFFFu) memcpy(rgba_out + (size_t)pix * 4, &slab[s * 4], 4 * sizeof(float));
    }
    return RT_OK;
}

int rt_read_primary_duv(rt_ctx* c, float* duv_out) {
    if (!c || !duv_out) return RT_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = c->n_slots;
    memset(duv_out, 0, (size_t)c->params.width * c->params.height * 4 * sizeof(float));
    if (!c->n_textures || !n) return RT_OK;
    std::vector<float> duv(n * 4);
    std::vector<uint32_t> status(n);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(duv.data(), c->d_duv, n * sizeof(float4), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(status.data(), c->pb.status, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (size_t s = 0; s < n; ++s) {
        uint32_t pix = c->h_slot_pixel[s];
        if (pix != 0xFFFFFFFFu && status[s] == 1) memcpy(duv_out + (size_t)pix * 4, &duv[s * 4], 4 * sizeof(float));
    }
    return RT_OK;
}

int rt_read_last_bounce_rays(rt_ctx* c, float* rays_soa, uint32_t cap, uint32_t* n_out) {
    if (!c || !n_out || (cap && !rays_soa)) return RT_ERR_ARG;
    *n_out = 0;
    const int D = c->params.max_depth;
    if (D < 2 || c->frames == 0 || c->n_total == 0) return RT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    uint32_t count = 0;
    HIP_TRY(c, hipMemcpy(&count, c->d_counts + (D - 1), sizeof count, hipMemcpyDeviceToHost));
    *n_out = count;
    if (!cap || !count) return RT_OK;
    const uint32_t take = std::min(count, cap);
    // the paths of the deepest bounce are the first `count` entries of the arrays that were current then
    const size_t n = c->n_total;
    std::vector<uint32_t> slot(count);
    std::vector<float> rays((size_t)6 * count);
    HIP_TRY(c, hipMemcpy(slot.data(), c->d_slot[c->last_buf], (size_t)count * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int k = 0; k < 6; ++k)
        HIP_TRY(c, hipMemcpy(&rays[(size_t)k * count], c->d_rays[c->last_buf] + (size_t)k * n, (size_t)count * sizeof(float),
                             hipMemcpyDeviceToHost));
    // they were appended in whatever order workgroups finished: hand them out in slot order
    std::vector<uint32_t> order(count);
    for (uint32_t i = 0; i < count; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return slot[a] < slot[b]; });
    for (uint32_t i = 0; i < take; ++i)
        for (int k = 0; k < 6; ++k) rays_soa[(size_t)k * cap + i] = rays[(size_t)k * count + order[i]];
    return RT_OK;
}

int rt_film_slab(rt_ctx* c, const float** device_ptr, uint32_t* n_tiles, uint64_t* n_floats) {
    if (!c) return RT_ERR_ARG;
    if (device_ptr) *device_ptr = reinterpret_cast<const float*>(c->pb.film);
    if (n_tiles) *n_tiles = c->n_local_tiles;
    if (n_floats) *n_floats = (uint64_t)c->n_slots * 4;
    return RT_OK;
}

int rt_copy_film_slab(rt_ctx* c, void* device_dst, uint64_t n_floats) {
    if (!c || !device_dst) return RT_ERR_ARG;
    if (n_floats != (uint64_t)c->n_slots * 4) {
        c->err = "rt_copy_film_slab: size does not match rt_film_slab";
        return RT_ERR_ARG;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (n_floats) HIP_TRY(c, hipMemcpyAsync(device_dst, c->pb.film, n_floats * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_unpack_slabs(int32_t W, int32_t H, int32_t tile, int32_t world, const float* slabs, uint64_t per_rank, float* rgba_out) {
    if (!slabs || !rgba_out || W <= 0 || H <= 0 || world <= 0) return RT_ERR_ARG;
    if (tile <= 0) tile = 64;
    if (tile % 8) return RT_ERR_ARG;
    const int tiles_x = (W + tile - 1) / tile, tiles_y = (H + tile - 1) / tile;
    const uint32_t n_tiles = (uint32_t)tiles_x * tiles_y;
    const int bpr = tile / 8;
    memset(rgba_out, 0, (size_t)W * H * 4 * sizeof(float));
    for (int rank = 0; rank < world; ++rank) {
        const float* slab = slabs + (size_t)rank * per_rank;
        uint32_t k = 0;
        for (uint32_t t = rank; t < n_tiles; t += world, ++k) {
            if (((uint64_t)k + 1) * tile * tile * 4 > per_rank) return RT_ERR_ARG;
            int tx = t % tiles_x, ty = t / tiles_x;
            for (int i = 0; i < tile * tile; ++i) {
                int blk = i >> 6, li = i & 63;
                int x = tx * tile + (blk % bpr) * 8 + (li & 7);
                int y = ty * tile + (blk / bpr) * 8 + (li >> 3);
                if (x < W && y < H)
                    memcpy(rgba_out + ((size_t)y * W + x) * 4, slab + ((size_t)k * tile * tile + i) * 4, 4 * sizeof(float));
            }
        }
    }
    return RT_OK;
}

int rt_unpack_slabs_dev(rt_ctx* c, int32_t world, const float* d_slabs, uint64_t per_rank, float* d_rgba_out) {
    if (!c || !d_slabs || !d_rgba_out || world <= 0) return RT_ERR_ARG;
    const int W = c->params.width, H = c->params.height, tile = c->tile;
    const uint32_t n_tiles = (uint32_t)((W + tile - 1) / tile) * (uint32_t)((H + tile - 1) / tile);
    // the largest slab must fit the stride the caller gathered with
    if ((uint64_t)((n_tiles + world - 1) / world) * tile * tile * 4 > per_rank) {
        c->err = "rt_unpack_slabs_dev: slab_floats_per_rank is smaller than a rank's slab";
        return RT_ERR_ARG;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    const uint32_t n_pix = (uint32_t)W * (uint32_t)H;
    hipLaunchKernelGGL(k_unpack_slabs, dim3((n_pix + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, c->stream,
                       reinterpret_cast<const float4*>(d_slabs), per_rank / 4, reinterpret_cast<float4*>(d_rgba_out), W, H, tile, world);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

uint64_t rt_slab_floats(int32_t W, int32_t H, int32_t tile, int32_t rank, int32_t world) {
    if (W <= 0 || H <= 0 || world <= 0 || rank < 0 || rank >= world) return 0;
    if (tile <= 0) tile = 64;
    const uint32_t n_tiles = (uint32_t)((W + tile - 1) / tile) * (uint32_t)((H + tile - 1) / tile);
    const uint32_t local = n_tiles > (uint32_t)rank ? (n_tiles - rank + world - 1) / world : 0;
    return (uint64_t)local * tile * tile * 4;
}

int rt_pack_slab(int32_t W, int32_t H, int32_t tile, int32_t rank, int32_t world, const float* rgba_in, float* slab_out) {
    if (!rgba_in || !slab_out || W <= 0 || H <= 0 || world <= 0 || rank < 0 || rank >= world) return RT_ERR_ARG;
    if (tile <= 0) tile = 64;
    if (tile % 8) return RT_ERR_ARG;
    const int tiles_x = (W + tile - 1) / tile, tiles_y = (H + tile - 1) / tile;
    const uint32_t n_tiles = (uint32_t)tiles_x * tiles_y;
    const int bpr = tile / 8;
    uint32_t k = 0;
    for (uint32_t t = rank; t < n_tiles; t += world, ++k) {
        int tx = t % tiles_x, ty = t / tiles_x;
        for (int i = 0; i < tile * tile; ++i) {
            int blk = i >> 6, li = i & 63;
            int x = tx * tile + (blk % bpr) * 8 + (li & 7);
            int y = ty * tile + (blk / bpr) * 8 + (li >> 3);
            float* dst = slab_out + ((size_t)k * tile * tile + i) * 4;
            if (x < W && y < H) memcpy(dst, rgba_in + ((size_t)y * W + x) * 4, 4 * sizeof(float));
            else memset(dst, 0, 4 * sizeof(float));
        }
    }
    return RT_OK;
}

int rt_trace_closest_dev(rt_ctx* c, uint32_t n, const float* d_rays, float* d_t, int32_t* d_id0, int32_t* d_id1, float* d_u,
                         float* d_v, uint32_t repeat) {
    if (!c || !d_rays || !d_t || !d_id0 || !d_id1 || !d_u || !d_v) return RT_ERR_ARG;
    if (n == 0) return RT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    return launch_closest(c, n, d_rays, d_t, d_id0, d_id1, d_u, d_v, repeat);
}

int rt_trace_closest_noalloc(rt_ctx* c, uint32_t n, const float* d_rays, float* d_t, int32_t* d_id0, int32_t* d_id1, float* d_u,
                             float* d_v) {
    if (!c || !d_rays || !d_t || !d_id0 || !d_id1 || !d_u || !d_v) return RT_ERR_ARG;
    if (n == 0) return RT_OK;
    if (n > kNoAllocMaxRays) {
        char b[160];
        snprintf(b, sizeof b, "rt_trace_closest_noalloc: %u rays exceed the fixed scratch of %u rays (use rt_trace_closest_dev)", n,
                 kNoAllocMaxRays);
        c->err = b;
        return RT_ERR_LIMIT;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    // the device address taken at rt_create, never the array's name
    TraceWork* work = reinterpret_cast<TraceWork*>(c->scratch_ptr);
    uint32_t* deferred = reinterpret_cast<uint32_t*>(c->scratch_ptr + kNoAllocWorkBytes);
    // the array keeps the last call's counters: they must start at zero (the list is rewritten before it is read)
    HIP_TRY(c, hipMemsetAsync(work, 0, sizeof(TraceWork), c->stream));
    if (c->use_wide)   // the same tree rt_trace_closest_dev walks in this context
        hipLaunchKernelGGL(k_trace_closest_wide, dim3(grid_for(c, n)), dim3(RT_TBLOCK), 0, c->stream, c->ts, d_rays, n, n, work, deferred, d_t,
                           d_id0, d_id1, d_u, d_v, c->d_raystats);
    else
        hipLaunchKernelGGL(k_trace_closest<false>, dim3(grid_for(c, n)), dim3(RT_TBLOCK), 0, c->stream, c->ts, d_rays, n, n, work, deferred, d_t,
                           d_id0, d_id1, d_u, d_v, c->d_raystats);
    hipLaunchKernelGGL(k_trace_closest<true>, dim3(c->grid_exact), dim3(RT_TBLOCK), 0, c->stream, c->ts, d_rays, n, n, work, deferred, d_t,
                       d_id0, d_id1, d_u, d_v, c->d_raystats);
    HIP_TRY(c, hipGetLastError());
    return RT_OK;
}

int rt_trace_occluded_dev(rt_ctx* c, uint32_t n, const float* d_rays, int32_t* d_status, uint32_t repeat) {
    if (!c || !d_rays || !d_status) return RT_ERR_ARG;
    if (n == 0) return RT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    return launch_occluded(c, n, d_rays, d_status, repeat);
}

int rt_trace_closest(rt_ctx* c, uint32_t n, const float* rays, float* dist, int32_t* id0, int32_t* id1, float* u, float* v) {
    if (!c || !rays || !dist || !id0 || !id1 || !u || !v) return RT_ERR_ARG;
    if (n == 0) return RT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    int r = ensure_trace_staging(c, n);
    if (r) return r;
    HIP_TRY(c, hipMemcpy(c->d_tr_rays, rays, (size_t)n * 6 * sizeof(float), hipMemcpyHostToDevice));
    float* o = c->d_tr_out;
    r = launch_closest(c, n, c->d_tr_rays, o, reinterpret_cast<int32_t*>(o + n), reinterpret_cast<int32_t*>(o + 2 * (size_t)n),
                       o + 3 * (size_t)n, o + 4 * (size_t)n, 1);
    if (r) return r;
    HIP_TRY(c, hipMemcpy(dist, o, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(id0, o + n, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(id1, o + 2 * (size_t)n, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(u, o + 3 * (size_t)n, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(v, o + 4 * (size_t)n, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_trace_occluded(rt_ctx* c, uint32_t n, const float* rays, int32_t* status) {
    if (!c || !rays || !status) return RT_ERR_ARG;
    if (n == 0) return RT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    int r = ensure_trace_staging(c, n);
    if (r) return r;
    HIP_TRY(c, hipMemcpy(c->d_tr_rays, rays, (size_t)n * 6 * sizeof(float), hipMemcpyHostToDevice));
    r = launch_occluded(c, n, c->d_tr_rays, reinterpret_cast<int32_t*>(c->d_tr_out), 1);
    if (r) return r;
    HIP_TRY(c, hipMemcpy(status, c->d_tr_out, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_trace_occluded_bounded(rt_ctx* c, uint32_t n, const float* rays, const float* tmax, int32_t* status) {
    if (!c || !rays || !tmax || !status) return RT_ERR_ARG;
    if (n == 0) return RT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    int r = ensure_trace_staging(c, n + n / 6 + 1);   // room for the tmax plane behind the six ray planes
    if (r) return r;
    float* d_tmax = c->d_tr_rays + (size_t)n * 6;
    HIP_TRY(c, hipMemcpy(c->d_tr_rays, rays, (size_t)n * 6 * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(d_tmax, tmax, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    r = launch_occluded(c, n, c->d_tr_rays, reinterpret_cast<int32_t*>(c->d_tr_out), 1, d_tmax);
    if (r) return r;
    HIP_TRY(c, hipMemcpy(status, c->d_tr_out, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_get_bvh(rt_ctx* c, uint32_t* n_nodes, float* boxes, int32_t* left, int32_t* right, int32_t* tri, int32_t* model,
               int32_t* leaf) {
    if (!c || !n_nodes) return RT_ERR_ARG;
    const size_t NN = c->n_ref_nodes;
    *n_nodes = (uint32_t)NN;
    if (!boxes) return RT_OK;
    if (!left || !right || !tri || !model || !leaf) return RT_ERR_ARG;
    if (c->device_built) {
        HIP_TRY(c, hipSetDevice(c->device));
        const LbvhRefArrays& ra = c->dbvh.ref;
        HIP_TRY(c, hipMemcpy(boxes, ra.boxes, NN * 6 * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(left, ra.left, NN * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(right, ra.right, NN * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(tri, ra.tri, NN * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(model, ra.model, NN * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(leaf, ra.leaf, NN * sizeof(int32_t), hipMemcpyDeviceToHost));
        return RT_OK;
    }
    const auto& ref = c->bvh.ref;
    for (size_t i = 0; i < ref.size(); ++i) {
        for (int k = 0; k < 3; ++k) { boxes[6 * i + k] = ref[i].bmin[k]; boxes[6 * i + 3 + k] = ref[i].bmax[k]; }
        left[i] = ref[i].left; right[i] = ref[i].right; tri[i] = ref[i].tri; model[i] = ref[i].model; leaf[i] = ref[i].leaf;
    }
    return RT_OK;
}

int rt_lbvh_host(const float* pos, uint32_t pos_stride, const uint32_t* indices, uint32_t n_indices, const int32_t* tri_model,
                 int32_t* max_depth, float* boxes, int32_t* left, int32_t* right, int32_t* tri, int32_t* model, int32_t* leaf) {
    if (!pos || !indices || !tri_model || n_indices < 3 || n_indices % 3 || !boxes || !left || !right || !tri || !model || !leaf)
        return RT_ERR_ARG;
    LbvhResult res;
    build_lbvh(pos, pos_stride ? pos_stride : 12, indices, n_indices, tri_model, res, RT_TOP_LEVELS);
    if (max_depth) *max_depth = res.max_depth;
    for (size_t i = 0; i < res.ref.size(); ++i) {
        for (int k = 0; k < 3; ++k) { boxes[6 * i + k] = res.ref[i].bmin[k]; boxes[6 * i + 3 + k] = res.ref[i].bmax[k]; }
        left[i] = res.ref[i].left; right[i] = res.ref[i].right; tri[i] = res.ref[i].tri; model[i] = res.ref[i].model;
        leaf[i] = res.ref[i].leaf;
    }
    return RT_OK;
}

int rt_get_wide_bvh(rt_ctx* c, uint32_t* n_nodes, int32_t* depth, float* nodes32) {
    if (!c || !n_nodes) return RT_ERR_ARG;
    *n_nodes = c->wide.n_nodes;
    if (depth) *depth = c->wide.depth;
    if (!nodes32 || !c->wide.n_nodes) return RT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpy(nodes32, c->wide.nodes, (size_t)c->wide.n_nodes * 128, hipMemcpyDeviceToHost));
    return RT_OK;
}

uint32_t rt_morton3d(float x, float y, float z) { return morton3d(x, y, z); }

int rt_get_stats(rt_ctx* c, rt_stats* out) {
    if (!c || !out) return RT_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned long long rs[4] = {0, 0, 0, 0};
    HIP_TRY(c, hipMemcpy(rs, c->d_raystats, sizeof rs, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    out->closest_rays = rs[0];
    out->shadow_rays = rs[1];
    out->exact_closest_rays = rs[2];
    out->exact_shadow_rays = rs[3];
    out->passes = c->frames;
    out->n_nodes = c->n_ref_nodes;
    out->n_triangles = c->n_triangles;
    out->bvh_max_depth = c->bvh_max_depth;
    out->n_local_pixels = c->n_valid;
    out->last_trace_ms = c->last_trace_ms;
    out->last_render_ms = c->last_render_ms;
    out->bvh_build_ms = c->bvh_build_ms;
    out->device_bytes = c->device_bytes;
    out->path_state_bytes = c->path_state_bytes;
    out->batch = c->batch;
    return RT_OK;
}

float rt_last_trace_ms(const rt_ctx* c) { return c ? c->last_trace_ms : 0.f; }

int rt_synchronize(rt_ctx* c) {
    if (!c) return RT_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

}  // extern "C"

#ifdef RT_PROBE_TIME
extern "C" int rt_debug_tprobe(unsigned long long* out, int n_waves) {   // out: n_waves x 4 stamps (see trace_wide.h)
    if (n_waves < 0 || n_waves > RT_TPROBE_WAVES) return 1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(rt::g_trec), (size_t)n_waves * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif
#ifdef RT_PROBE
extern "C" int rt_debug_probe(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(rt::g_probe), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rt::g_probe), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif
