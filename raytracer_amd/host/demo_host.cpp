// Drives HipPathTracer the way the reference app drives its integrator
// (src/app/App.cpp:92-93,113-121) on a scene read from a flat binary dump.
// usage: demo_host <scene.bin> <width> <height> <frames> <out_film.f32> [devices, e.g. 0,1,2,3] [node]
// (`node` drives a single device through the multi-device entry points as well)
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "HipPathTracer.hpp"

template <typename T>
static std::vector<T> rd(std::ifstream& f, size_t n) {
    std::vector<T> v(n);
    f.read(reinterpret_cast<char*>(v.data()), n * sizeof(T));
    return v;
}

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s scene.bin W H frames out.f32\n", argv[0]); return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    // header: n_vertices n_indices n_models n_point_lights, then cam(pos,right,forward), arrays
    auto hdr = rd<uint32_t>(f, 4);
    auto cam = rd<float>(f, 9);
    auto pos = rd<float>(f, hdr[0] * 3), nrm = rd<float>(f, hdr[0] * 3), uv = rd<float>(f, hdr[0] * 2);
    auto tan = rd<float>(f, hdr[0] * 3), bin = rd<float>(f, hdr[0] * 3);
    auto idx = rd<uint32_t>(f, hdr[1]);
    auto tm = rd<int32_t>(f, hdr[1] / 3);
    auto mats = rd<float>(f, hdr[2] * 4);
    auto pl = rd<float>(f, hdr[3] * 6);
    std::vector<rt_material> materials(hdr[2]);
    for (uint32_t m = 0; m < hdr[2]; ++m) {
        for (int k = 0; k < 3; ++k) materials[m].diffuse_color[k] = mats[4 * m + k];
        materials[m].roughness = mats[4 * m + 3];
        materials[m].texture = -1;
        materials[m].glossy = RT_GLOSSY_NONE;
    }
    std::vector<rt_light> lights(hdr[3]);
    for (uint32_t l = 0; l < hdr[3]; ++l) {
        lights[l] = rt_light{};
        lights[l].type = RT_LIGHT_POINT;
        for (int k = 0; k < 3; ++k) { lights[l].pos[k] = pl[6 * l + k]; lights[l].intensity[k] = pl[6 * l + 3 + k]; }
    }
    rt_scene_desc sd{};
    sd.n_vertices = hdr[0];
    sd.pos = pos.data(); sd.pos_stride = 12;
    sd.normal = nrm.data(); sd.normal_stride = 12;
    sd.uv = uv.data(); sd.uv_stride = 8;
    sd.tangent = tan.data(); sd.tangent_stride = 12;
    sd.binormal = bin.data(); sd.binormal_stride = 12;
    sd.n_indices = hdr[1]; sd.indices = idx.data(); sd.tri_model = tm.data();
    sd.n_models = hdr[2]; sd.materials = materials.data();
    sd.n_lights = hdr[3]; sd.lights = lights.data();
    const int W = atoi(argv[2]), H = atoi(argv[3]), frames = atoi(argv[4]);
    std::vector<int32_t> devices;
    // comma-separated device ordinals; anything else ends the program instead of looping on text strtol cannot read
    for (const char* p = argc > 6 ? argv[6] : "0";;) {
        char* end = nullptr;
        const long v = strtol(p, &end, 10);
        if (end == p || v < 0 || (*end != ',' && *end != '\0')) {
            fprintf(stderr, "demo_host: device list must be ordinals separated by commas (got \"%s\")\n", argc > 6 ? argv[6] : "0");
            return 2;
        }
        devices.push_back((int32_t)v);
        if (*end == '\0') break;
        p = end + 1;
    }
    const bool use_node = argc > 7 && std::string(argv[7]) == "node";
    try {
        HipPathTracer pt(sd, W, H, devices, 4, 0, 1, use_node);
        std::vector<float> film((size_t)W * H * 4);
        pt.nextFrame(&cam[0], &cam[3], &cam[6]);
        pt.run();
        int shown = 0;
        while (shown < frames) {             // the app's update tick, src/app/App.cpp:113-121
            if (pt.isUpdateFinish()) {
                pt.sync(film.data());
                ++shown;
                if (shown < frames) pt.nextFrame(&cam[0], &cam[3], &cam[6]);
            } else {
                std::this_thread::yield();
            }
        }
        pt.finish();
        std::ofstream o(argv[5], std::ios::binary);
        o.write(reinterpret_cast<const char*>(film.data()), film.size() * sizeof(float));
        printf("frames %llu devices %d\n", (unsigned long long)pt.frames(), pt.deviceCount());
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
