// Microbenchmark (development aid): does a divergent 64-byte gather cost the
// L1 address/data path per instruction or per active lane?  Every wave issues
// the same number of 4 x dwordx4 node gathers; MASK says which lanes take part.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// MASK 0: all 64 lanes.  1: even lanes (32).  2: lanes 0..31.  3: lanes 0..15.  4: every fourth lane (16).  5: lane 0 only.
template <int MASK>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ nodes, uint32_t n_nodes, int iters, float* out) {
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    uint32_t s = hash(blockIdx.x * 256 + tid + 1);
    float acc = 0.f;
    bool on = true;
    if (MASK == 1) on = (lane & 1) == 0;
    if (MASK == 2) on = lane < 32;
    if (MASK == 3) on = lane < 16;
    if (MASK == 4) on = (lane & 3) == 0;
    if (MASK == 5) on = lane == 0;
    for (int it = 0; it < iters; ++it) {
        s = hash(s + it);
        uint32_t idx = s % n_nodes;
        if (on) {
            const float4* p = nodes + (size_t)idx * 4;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        }
        s += (uint32_t)(acc != 12345.678f);
    }
    out[blockIdx.x * 256 + tid] = acc;
}

template <int MASK>
void run(const char* name, const float4* d_nodes, uint32_t n_nodes, float* d_out, int blocks, int iters, int num_cu) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MASK>, dim3(blocks), dim3(256), 0, 0, d_nodes, n_nodes, iters, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MASK>, dim3(blocks), dim3(256), 0, 0, d_nodes, n_nodes, iters, d_out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double wave_visits = (double)blocks * 4 * iters;
    printf("  %-28s %8.3f ms  %7.2f G wave-gathers/s  %6.1f clk per wave-gather per CU @2.4GHz\n", name, ms, wave_visits / ms / 1e6,
           (ms * 1e-3) * 2.4e9 * num_cu / wave_visits);
}

// SHAPE 0: lane-private node, 4 x dwordx4.  1: 1 x dwordx4.  2: 2 x dwordx4.  3: 16 x dword.  4: 4 x dwordx4 where the
// four lanes of a quad read the four 16-byte pieces of ONE node per instruction (four nodes over the four instructions).
// 5: lane-private node, 1 x dwordx4 + 1 x dwordx2 + 1 x dword (28 B)
template <int SHAPE>
__global__ __launch_bounds__(256) void ks(const float4* __restrict__ nodes, uint32_t n_nodes, int iters, float* out) {
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    uint32_t s = hash(blockIdx.x * 256 + tid + 1);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        s = hash(s + it);
        uint32_t idx = s % n_nodes;
        const float4* p = nodes + (size_t)idx * 4;
        if (SHAPE == 0) { float4 a = p[0], b = p[1], c = p[2], d = p[3]; acc += a.x + b.y + c.z + d.w; }
        if (SHAPE == 1) { float4 a = p[0]; acc += a.x + a.w; }
        if (SHAPE == 2) { float4 a = p[0], b = p[1]; acc += a.x + b.w; }
        if (SHAPE == 3) { const float* q = reinterpret_cast<const float*>(p);
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += q[j]; }
        if (SHAPE == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t ij = __shfl(idx, (lane & ~3u) | j);
                float4 a = nodes[(size_t)ij * 4 + (lane & 3)];
                acc += a.x + a.w;
            }
        }
        if (SHAPE == 5) { float4 a = p[0]; float2 b = *reinterpret_cast<const float2*>(p + 1); float c = *reinterpret_cast<const float*>(p + 2); acc += a.x + b.y + c; }
        s += (uint32_t)(acc != 12345.678f);
    }
    out[blockIdx.x * 256 + tid] = acc;
}

template <int SHAPE>
void runs(const char* name, const float4* d_nodes, uint32_t n_nodes, float* d_out, int blocks, int iters, int num_cu) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(ks<SHAPE>, dim3(blocks), dim3(256), 0, 0, d_nodes, n_nodes, iters, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(ks<SHAPE>, dim3(blocks), dim3(256), 0, 0, d_nodes, n_nodes, iters, d_out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double wave_visits = (double)blocks * 4 * iters;
    printf("  %-40s %8.3f ms  %6.1f clk per wave-visit per CU @2.4GHz\n", name, ms, (ms * 1e-3) * 2.4e9 * num_cu / wave_visits);
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int num_cu = prop.multiProcessorCount;
    int blocks = num_cu * 8, iters = 400;
    float* d_out; CK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    const uint32_t kbs[] = {16, 1024, 16384, 65536};
    for (uint32_t kb : kbs) {
        uint32_t n_nodes = kb * 1024u / 64;
        std::vector<float> h((size_t)n_nodes * 16, 1.0f);
        float4* d_nodes; CK(hipMalloc(&d_nodes, (size_t)n_nodes * 64));
        CK(hipMemcpy(d_nodes, h.data(), (size_t)n_nodes * 64, hipMemcpyHostToDevice));
        printf("table %u KB, %d CUs, %d blocks x 256 (8 waves/SIMD), %d gathers per wave\n", kb, num_cu, blocks, iters);
        run<0>("64 lanes", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
        run<1>("32 lanes (even)", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
        run<2>("32 lanes (0..31)", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
        run<3>("16 lanes (0..15)", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
        run<4>("16 lanes (every 4th)", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
        run<5>("1 lane", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
        runs<0>("shape: 4 x dwordx4 (64 B)", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
        runs<2>("shape: 2 x dwordx4 (32 B)", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
        runs<1>("shape: 1 x dwordx4 (16 B)", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
        runs<5>("shape: x4 + x2 + x1 (28 B)", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
        runs<3>("shape: 16 x dword (64 B)", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
        runs<4>("shape: quad-cooperative 4 x dwordx4", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
        CK(hipFree(d_nodes));
    }
    return 0;
}
