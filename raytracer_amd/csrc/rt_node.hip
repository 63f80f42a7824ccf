// One process driving several GPUs of a node (declared in include/rt_hip.h, rt_node_*): the scene is replicated
// on every device, the film is sharded by interleaved tiles (rank i of n on device i: the GPU form of the
// reference's tile pool, src/cpu/Integrator.cpp:33-49), every device renders its tiles on a host thread of its
// own (persistent workers), and at read-out the tile slabs are gathered with RCCL on the first device -- one
// communicator per device (ncclCommInitAll), one group of ncclSend / ncclRecv pairs, so every slab crosses xGMI once
// (RT_NODE_GATHER=allgather selects one grouped ncclAllGather instead, which leaves the slabs on every device) -- and
// scattered into the film there.  The film reaches the host through a pinned buffer the node owns.
//
// RCCL is bound at run time (dlopen of librccl.so.1): the library then carries no link-time dependency on a
// particular RCCL build, and a process that already holds one (PyTorch ships its own) shares it.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_hip.h"

namespace {

typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
constexpr int kNcclFloat = 7;   // ncclFloat32 of rccl.h

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// RT_RCCL_LIB, when set, is the only library tried; otherwise the usual names.
bool load_rccl(Rccl& r, std::string& err) {
    const char* forced = getenv("RT_RCCL_LIB");
    const char* usual[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string why;
    auto try_open = [&](const char* n) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) {
            const char* e = dlerror();   // one call: it returns the text and clears it
            why = e ? e : "not found";
        }
        return r.handle != nullptr;
    };
    if (forced && *forced) {
        try_open(forced);
    } else {
        for (const char* n : usual)
            if (try_open(n)) break;
    }
    if (!r.handle) {
        err = std::string("cannot load RCCL (") + ((forced && *forced) ? forced : "librccl.so.1") + "): " + why;
        return false;
    }
    auto sym = [&](const char* s) { return dlsym(r.handle, s); };
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.AllGather || !r.Send || !r.Recv || !r.GetErrorString) {
        err = "the RCCL library lacks an expected entry point";
        dlclose(r.handle);
        r.handle = nullptr;
        return false;
    }
    return true;
}

// One host thread per device, alive as long as the node: rt_node_render is called once per sample pass by the host
// adapter, and starting and joining threads at that rate costs more than the hand-over.
class Workers {
public:
    explicit Workers(int n) : jobs_(n), rc_(n, 0), state_(n, 0) {
        for (int i = 0; i < n; ++i) threads_.emplace_back([this, i]() { loop(i); });
    }
    ~Workers() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    // runs fn(i) on worker i for every i and waits for all of them
    void run(const std::function<int(int)>& fn, std::vector<int>& rc) {
        {
            std::lock_guard<std::mutex> lk(m_);
            for (size_t i = 0; i < jobs_.size(); ++i) { jobs_[i] = fn; state_[i] = 1; }
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&]() { for (int s : state_) if (s != 0) return false; return true; });
        rc = rc_;
    }

private:
    void loop(int i) {
        for (;;) {
            std::function<int(int)> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&]() { return stop_ || state_[i] == 1; });
                if (stop_) return;
                job = jobs_[i];
                state_[i] = 2;
            }
            const int r = job(i);
            {
                std::lock_guard<std::mutex> lk(m_);
                rc_[i] = r;
                state_[i] = 0;
            }
            done_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::vector<std::function<int(int)>> jobs_;
    std::vector<int> rc_, state_;   // state: 0 idle, 1 posted, 2 running
    std::vector<std::thread> threads_;
    bool stop_ = false;
};

thread_local std::string g_node_create_error;

}  // namespace

struct rt_node {
    int n = 0;
    int width = 0, height = 0, tile = 64;
    std::vector<int> devices;
    std::vector<rt_ctx*> ctx;
    std::vector<hipStream_t> streams;     // one per device, for the slab copy and the gather
    std::vector<float*> d_local;          // this device's slab, padded to per_rank floats
    std::vector<float*> d_gathered;       // n * per_rank floats: on devices[0] only, or on every device in all-gather mode
    bool allgather = false;               // RT_NODE_GATHER=allgather
    float* d_film = nullptr;              // width * height * 4 floats on devices[0]
    float* h_film = nullptr;              // the same size, pinned host memory (allocated at the first read-out)
    Workers* workers = nullptr;           // one thread per device when n > 1
    uint64_t per_rank = 0;                // floats
    Rccl rccl;
    std::vector<ncclComm_t> comms;
    std::string err;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // on devices[0]: bracket the all-gather
    float last_gather_ms = 0.f;
};

namespace {

#define NODE_HIP(nd, call)                                                                              \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            (nd)->err = std::string(#call) + " failed: " + hipGetErrorString(e_);                       \
            return RT_ERR_HIP;                                                                          \
        }                                                                                               \
    } while (0)

#define NODE_NCCL(nd, call)                                                                             \
    do {                                                                                                \
        ncclResult_t r_ = (call);                                                                       \
        if (r_ != 0) {                                                                                  \
            (nd)->err = std::string(#call) + " failed: " + (nd)->rccl.GetErrorString(r_);               \
            return RT_ERR_HIP;                                                                          \
        }                                                                                               \
    } while (0)

// Runs fn(i) for every device on that device's worker thread and returns the first non-zero status.
template <typename F>
int for_each_device(rt_node* nd, F fn) {
    std::vector<int> rc(nd->n, RT_OK);
    if (nd->n == 1 || !nd->workers) {
        for (int i = 0; i < nd->n; ++i) rc[i] = fn(i);
    } else {
        nd->workers->run(fn, rc);
    }
    for (int i = 0; i < nd->n; ++i)
        if (rc[i] != RT_OK) {
            nd->err = std::string("device ") + std::to_string(nd->devices[i]) + ": " + rt_last_error(nd->ctx[i]);
            return rc[i];
        }
    return RT_OK;
}

int node_create_impl(rt_node* nd, const rt_scene_desc* scene, const rt_params* params, const int32_t* devices, int32_t n) {
    nd->n = n;
    nd->width = params->width;
    nd->height = params->height;
    nd->tile = params->tile_size > 0 ? params->tile_size : 64;
    nd->devices.assign(devices, devices + n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) {
                nd->err = "a device is listed twice";
                return RT_ERR_ARG;
            }
    // RCCL first: without it nothing below is of use, and nothing has been allocated yet
    if (!load_rccl(nd->rccl, nd->err)) return RT_ERR_NO_DEVICE;
    const char* gm = getenv("RT_NODE_GATHER");
    nd->allgather = gm && std::string(gm) == "allgather";
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        nd->err = std::string("device ") + std::to_string(devices[0]) + ": no HIP device available";
        return RT_ERR_NO_DEVICE;
    }
    for (int i = 0; i < n; ++i)
        if (devices[i] < 0 || devices[i] >= ndev) {
            nd->err = std::string("device ") + std::to_string(devices[i]) + ": ordinal out of range";
            return RT_ERR_ARG;
        }
    nd->ctx.assign(n, nullptr);
    // contexts one after the other: rt_create reports its error text per calling thread
    for (int i = 0; i < n; ++i) {
        rt_params p = *params;
        p.device = devices[i];
        p.rank = i;
        p.world = n;
        int r = rt_create(scene, &p, &nd->ctx[i]);
        if (r != RT_OK) {
            nd->err = std::string("device ") + std::to_string(devices[i]) + ": " + rt_last_error(nullptr);
            return r;
        }
    }
    nd->per_rank = 0;
    for (int i = 0; i < n; ++i) nd->per_rank = std::max<uint64_t>(nd->per_rank, rt_slab_floats(nd->width, nd->height, nd->tile, i, n));
    if (nd->per_rank == 0) nd->per_rank = 4;
    nd->streams.assign(n, nullptr);
    nd->d_local.assign(n, nullptr);
    nd->d_gathered.assign(n, nullptr);
    for (int i = 0; i < n; ++i) {
        NODE_HIP(nd, hipSetDevice(devices[i]));
        NODE_HIP(nd, hipStreamCreateWithFlags(&nd->streams[i], hipStreamNonBlocking));
        NODE_HIP(nd, hipMalloc(reinterpret_cast<void**>(&nd->d_local[i]), nd->per_rank * sizeof(float)));
        NODE_HIP(nd, hipMemset(nd->d_local[i], 0, nd->per_rank * sizeof(float)));   // the padding behind a short slab stays zero
        if (i == 0 || nd->allgather)
            NODE_HIP(nd, hipMalloc(reinterpret_cast<void**>(&nd->d_gathered[i]), (size_t)n * nd->per_rank * sizeof(float)));
    }
    NODE_HIP(nd, hipSetDevice(devices[0]));
    NODE_HIP(nd, hipMalloc(reinterpret_cast<void**>(&nd->d_film), (size_t)nd->width * nd->height * 4 * sizeof(float)));
    NODE_HIP(nd, hipEventCreate(&nd->ev0));
    NODE_HIP(nd, hipEventCreate(&nd->ev1));
    nd->comms.assign(n, nullptr);
    NODE_NCCL(nd, nd->rccl.CommInitAll(nd->comms.data(), n, nd->devices.data()));
    if (n > 1) nd->workers = new Workers(n);
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_node_last_error(const rt_node* nd) { return nd ? nd->err.c_str() : g_node_create_error.c_str(); }

void rt_node_destroy(rt_node* nd) {
    if (!nd) return;
    delete nd->workers;
    nd->workers = nullptr;
    for (int i = 0; i < nd->n; ++i) {
        if (i >= (int)nd->ctx.size() || !nd->ctx[i]) continue;   // create stopped before this device
        (void)hipSetDevice(nd->devices[i]);
        if (i < (int)nd->comms.size() && nd->comms[i]) (void)nd->rccl.CommDestroy(nd->comms[i]);
        if (i < (int)nd->streams.size() && nd->streams[i]) { (void)hipStreamSynchronize(nd->streams[i]); (void)hipStreamDestroy(nd->streams[i]); }
        if (i < (int)nd->d_local.size() && nd->d_local[i]) (void)hipFree(nd->d_local[i]);
        if (i < (int)nd->d_gathered.size() && nd->d_gathered[i]) (void)hipFree(nd->d_gathered[i]);
        if (i < (int)nd->ctx.size() && nd->ctx[i]) rt_destroy(nd->ctx[i]);
    }
    if (nd->d_film || nd->h_film || nd->ev0 || nd->ev1) {
        (void)hipSetDevice(nd->devices[0]);
        if (nd->d_film) (void)hipFree(nd->d_film);
        if (nd->h_film) (void)hipHostFree(nd->h_film);
        if (nd->ev0) (void)hipEventDestroy(nd->ev0);
        if (nd->ev1) (void)hipEventDestroy(nd->ev1);
    }
    (void)hipGetLastError();   // a failed create must not leave a stale error behind for the next HIP call to report
    delete nd;
}

int rt_node_create(const rt_scene_desc* scene, const rt_params* params, const int32_t* devices, int32_t n_devices, rt_node** out) {
    if (!scene || !params || !devices || !out || n_devices < 1 || n_devices > 64) {
        g_node_create_error = "rt_node_create: null argument or device count outside 1..64";
        return RT_ERR_ARG;
    }
    *out = nullptr;
    rt_node* nd = new rt_node();
    int r = node_create_impl(nd, scene, params, devices, n_devices);
    if (r != RT_OK) {
        g_node_create_error = nd->err;
        rt_node_destroy(nd);
        return r;
    }
    *out = nd;
    return RT_OK;
}

int rt_node_device_count(const rt_node* nd) { return nd ? nd->n : 0; }

rt_ctx* rt_node_context(rt_node* nd, int32_t i) { return (nd && i >= 0 && i < nd->n) ? nd->ctx[i] : nullptr; }

int rt_node_set_camera(rt_node* nd, const float pos[3], const float right[3], const float up[3], const float forward[3]) {
    if (!nd) return RT_ERR_ARG;
    for (int i = 0; i < nd->n; ++i) {
        int r = rt_set_camera(nd->ctx[i], pos, right, up, forward);
        if (r != RT_OK) { nd->err = rt_last_error(nd->ctx[i]); return r; }
    }
    return RT_OK;
}

int rt_node_reset_accumulation(rt_node* nd) {
    if (!nd) return RT_ERR_ARG;
    for (int i = 0; i < nd->n; ++i) rt_reset_accumulation(nd->ctx[i]);
    return RT_OK;
}

int rt_node_render(rt_node* nd, uint32_t n_passes) {
    if (!nd) return RT_ERR_ARG;
    return for_each_device(nd, [&](int i) { return rt_render(nd->ctx[i], n_passes); });
}

int rt_node_gather_film(rt_node* nd, const float** d_film_out) {
    if (!nd) return RT_ERR_ARG;
    const int n = nd->n;
    // every device: its slab into the padded send buffer (rt_copy_film_slab waits for the device's renders first)
    int r = for_each_device(nd, [&](int i) {
        const uint64_t nf = rt_slab_floats(nd->width, nd->height, nd->tile, i, n);
        return nf ? rt_copy_film_slab(nd->ctx[i], nd->d_local[i], nf) : (int)RT_OK;
    });
    if (r != RT_OK) return r;
    NODE_HIP(nd, hipSetDevice(nd->devices[0]));
    NODE_HIP(nd, hipEventRecord(nd->ev0, nd->streams[0]));
    if (nd->allgather) {
        NODE_NCCL(nd, nd->rccl.GroupStart());
        for (int i = 0; i < n; ++i) {
            ncclResult_t g = nd->rccl.AllGather(nd->d_local[i], nd->d_gathered[i], nd->per_rank, kNcclFloat, nd->comms[i], nd->streams[i]);
            if (g != 0) {
                (void)nd->rccl.GroupEnd();
                nd->err = std::string("ncclAllGather failed: ") + nd->rccl.GetErrorString(g);
                return RT_ERR_HIP;
            }
        }
        NODE_NCCL(nd, nd->rccl.GroupEnd());
    } else {
        // gather on the presenting device only: its own slab by a device copy, every other slab by one
        // ncclSend / ncclRecv pair (rank i of the communicator = device i)
        NODE_HIP(nd, hipMemcpyAsync(nd->d_gathered[0], nd->d_local[0], nd->per_rank * sizeof(float), hipMemcpyDeviceToDevice, nd->streams[0]));
        if (n > 1) {
            NODE_NCCL(nd, nd->rccl.GroupStart());
            ncclResult_t g = 0;
            for (int i = 1; i < n && g == 0; ++i) {
                g = nd->rccl.Recv(nd->d_gathered[0] + (size_t)i * nd->per_rank, nd->per_rank, kNcclFloat, i, nd->comms[0], nd->streams[0]);
                if (g == 0) g = nd->rccl.Send(nd->d_local[i], nd->per_rank, kNcclFloat, 0, nd->comms[i], nd->streams[i]);
            }
            if (g != 0) {
                (void)nd->rccl.GroupEnd();
                nd->err = std::string("ncclSend / ncclRecv failed: ") + nd->rccl.GetErrorString(g);
                return RT_ERR_HIP;
            }
            NODE_NCCL(nd, nd->rccl.GroupEnd());
        }
    }
    for (int i = 0; i < n; ++i) {
        NODE_HIP(nd, hipSetDevice(nd->devices[i]));
        NODE_HIP(nd, hipStreamSynchronize(nd->streams[i]));
    }
    NODE_HIP(nd, hipSetDevice(nd->devices[0]));
    NODE_HIP(nd, hipEventRecord(nd->ev1, nd->streams[0]));
    NODE_HIP(nd, hipEventSynchronize(nd->ev1));
    NODE_HIP(nd, hipEventElapsedTime(&nd->last_gather_ms, nd->ev0, nd->ev1));
    // the scatter runs on the first context's stream and returns when it is done
    r = rt_unpack_slabs_dev(nd->ctx[0], n, nd->d_gathered[0], nd->per_rank, nd->d_film);
    if (r != RT_OK) { nd->err = rt_last_error(nd->ctx[0]); return r; }
    if (d_film_out) *d_film_out = nd->d_film;
    return RT_OK;
}

int rt_node_map_film(rt_node* nd, const float** host_film_out) {
    if (!nd || !host_film_out) return RT_ERR_ARG;
    int r = rt_node_gather_film(nd, nullptr);
    if (r != RT_OK) return r;
    NODE_HIP(nd, hipSetDevice(nd->devices[0]));
    const size_t bytes = (size_t)nd->width * nd->height * 4 * sizeof(float);
    if (!nd->h_film) NODE_HIP(nd, hipHostMalloc(reinterpret_cast<void**>(&nd->h_film), bytes, hipHostMallocDefault));
    NODE_HIP(nd, hipMemcpyAsync(nd->h_film, nd->d_film, bytes, hipMemcpyDeviceToHost, nd->streams[0]));
    NODE_HIP(nd, hipStreamSynchronize(nd->streams[0]));
    *host_film_out = nd->h_film;
    return RT_OK;
}

int rt_node_read_film(rt_node* nd, float* rgba_out) {
    if (!nd || !rgba_out) return RT_ERR_ARG;
    const float* h = nullptr;
    int r = rt_node_map_film(nd, &h);
    if (r != RT_OK) return r;
    memcpy(rgba_out, h, (size_t)nd->width * nd->height * 4 * sizeof(float));
    return RT_OK;
}

float rt_node_last_gather_ms(const rt_node* nd) { return nd ? nd->last_gather_ms : 0.f; }

}  // extern "C"
