// HipPathTracer: the host-side stand-in for the reference's Integrator
// (src/cpu/Integrator.hpp:14-58) that drives the gfx950 library through its
// C-ABI instead of a CPU tile pool.  Same call protocol as the reference app
// uses (src/app/App.cpp:40-43,92-93,113-121):
//
//     HipPathTracer pt(scene_desc, width, height);     // Integrator(device, scene, w, h)
//     pt.nextFrame(pos, right, forward); pt.run();     // App.cpp:92-93
//     ... per UI tick:
//     if (pt.isUpdateFinish()) { pt.sync(dst); pt.nextFrame(pos, right, forward); }   // App.cpp:116-119
//     pt.finish();
//
// It needs no Mango/glm: the scene arrives as the flat arrays BVH::runLBVH
// flattens to (rt_scene_desc) and the camera as the three vectors nextFrame
// snapshots (src/cpu/Integrator.cpp:85-92).  In the reference tree the same
// class body becomes `class HipPathTracer : public Integrator`, with the
// arrays filled from Scene::models() and sync() writing the staging buffer.
#pragma once
#include <atomic>
#include <cmath>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_hip.h"

class HipPathTracer {
   public:
    // passes_per_update = 1 is the reference's protocol (one 1-spp pass per update(), src/cpu/Integrator.cpp:26-67);
    // a larger value makes every update() add that many passes, traced together as one wavefront.
    HipPathTracer(const rt_scene_desc& scene, int width, int height, int max_depth = 4, uint64_t seed = 0, int device = 0,
                  uint32_t passes_per_update = 1)
        : HipPathTracer(scene, width, height, std::vector<int32_t>{device}, max_depth, seed, passes_per_update) {}
    // Several GPUs of the node: the 64x64 tiles that the reference deals to its 16 threads
    // (src/cpu/Integrator.cpp:33-49) are dealt to the devices instead (tile t to devices[t % n]); sync() receives the
    // film gathered over RCCL.  One device keeps to the single-context entry points.
    // (use_node = true takes the multi-device entry points for a single device too.)
    HipPathTracer(const rt_scene_desc& scene, int width, int height, const std::vector<int32_t>& devices, int max_depth = 4,
                  uint64_t seed = 0, uint32_t passes_per_update = 1, bool use_node = false, uint32_t flags = 0)
        : _width(width), _height(height), _passes(passes_per_update ? passes_per_update : 1) {
        if (devices.empty()) throw std::runtime_error("HipPathTracer: no device given");
        rt_params p{};
        p.width = width; p.height = height; p.max_depth = max_depth; p.tile_size = 64;
        p.seed = seed; p.device = devices[0]; p.rank = 0; p.world = 1;
        p.batch = _passes;   // path state sized for what one update() traces
        p.flags = flags;     // RT_FLAG_* of rt_hip.h (RT_FLAG_WIDE_BVH: faster traversal, tie-tolerant hit ids)
        if (devices.size() == 1 && !use_node) {
            if (rt_create(&scene, &p, &_ctx) != RT_OK) throw std::runtime_error(std::string("rt_create: ") + rt_last_error(nullptr));
        } else if (rt_node_create(&scene, &p, devices.data(), (int32_t)devices.size(), &_node) != RT_OK) {
            throw std::runtime_error(std::string("rt_node_create: ") + rt_node_last_error(nullptr));
        }
        _frame.assign((size_t)width * height * 4, 0.f);
    }
    ~HipPathTracer() {
        finish();
        if (_node) rt_node_destroy(_node);
        else rt_destroy(_ctx);
    }
    HipPathTracer(const HipPathTracer&) = delete;
    HipPathTracer& operator=(const HipPathTracer&) = delete;

    // src/cpu/Integrator.cpp:69-79: the render loop; one library pass per update().  The thread leaves
    // _isFinish set whichever way it ends, so a host that polls isUpdateFinish() never waits on a dead thread.
    void run() {
        if (_renderThread) return;
        _renderThread = std::make_shared<std::thread>([this]() {
            while (_isRun) {
                if (!_isFinish) {
                    update();
                    _isFinish = true;
                } else {
                    std::this_thread::yield();
                }
            }
            _isFinish = true;
        });
    }
    bool isUpdateFinish() const { return _isFinish; }  // src/cpu/Integrator.cpp:81-83

    // src/cpu/Integrator.cpp:85-92: snapshot the camera; up = normalize(cross(right, forward)).  Does nothing once the
    // library has reported an error (see error()) or after finish(): there is no thread left to pick the frame up.
    void nextFrame(const float pos[3], const float right[3], const float forward[3]) {
        if (_failed || !_isRun) return;
        float c[3] = {right[1] * forward[2] - forward[1] * right[2], right[2] * forward[0] - forward[2] * right[0],
                      right[0] * forward[1] - forward[0] * right[1]};
        float inv = 1.0f / std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        for (int i = 0; i < 3; ++i) { _pos[i] = pos[i]; _right[i] = right[i]; _forward[i] = forward[i]; _up[i] = c[i] * inv; }
        _isFinish = false;
    }
    // src/cpu/Integrator.cpp:94-97: stops the loop and joins the thread (the reference spins on _isFinish, which a
    // thread that has already left its loop would never set)
    void finish() {
        _isRun = false;
        if (_renderThread && _renderThread->joinable()) _renderThread->join();
        _isFinish = true;
    }
    // Text of the library error that stopped the render loop; empty while all is well.  Read it after
    // isUpdateFinish() returned true.
    bool failed() const { return _failed; }
    const std::string& error() const { return _error; }
    // src/cpu/Integrator.cpp:21-24: hand the finished frame over (width*height*4 floats)
    void sync(float* rgba_dst) { std::memcpy(rgba_dst, _frame.data(), _frame.size() * sizeof(float)); }

    uint64_t frames() const { return _frames; }
    rt_ctx* context() { return _node ? rt_node_context(_node, 0) : _ctx; }
    int deviceCount() const { return _node ? rt_node_device_count(_node) : 1; }

   protected:
    // src/cpu/Integrator.cpp:26-67 with the tile pool replaced by one rt_render pass
    void update() {
        const bool ok = _node ? (rt_node_set_camera(_node, _pos, _right, _up, _forward) == RT_OK && rt_node_render(_node, _passes) == RT_OK &&
                                 rt_node_read_film(_node, _frame.data()) == RT_OK)
                              : (rt_set_camera(_ctx, _pos, _right, _up, _forward) == RT_OK && rt_render(_ctx, _passes) == RT_OK &&
                                 rt_read_film(_ctx, _frame.data()) == RT_OK);
        if (!ok) {
            _error = _node ? rt_node_last_error(_node) : rt_last_error(_ctx);
            _failed = true;
            _isRun = false;
            return;
        }
        _frames += _passes;
    }

    rt_ctx* _ctx = nullptr;     // one device
    rt_node* _node = nullptr;   // several devices
    int _width, _height;
    uint32_t _passes = 1;
    float _pos[3]{}, _right[3]{}, _forward[3]{}, _up[3]{};
    std::vector<float> _frame;
    uint64_t _frames = 0;
    std::string _error;
    std::shared_ptr<std::thread> _renderThread;
    std::atomic_bool _isFinish{true};
    std::atomic_bool _isRun{true};
    std::atomic_bool _failed{false};
};
