// Microbenchmark (development aid): throughput of per-lane 64-byte node
// gathers on gfx950 under different access shapes.  Not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// MODE 0: lane-private node, 4 x dwordx4.  1: same node for the 4 lanes of a quad.
// 2: quad-cooperative regular loads (lane i loads piece i of the node of quad lane j), 4 instr.
// 3: as 2 through global_load_lds + ds_read.  4: lane-private, 16 x dword.  5: whole wave same node.
// 6: lane-private node, only 1 x dwordx4 (16 B).  7: lane-private 2 x dwordx4 (32 B)
template <int MODE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ nodes, uint32_t n_nodes, int iters, float* out) {
    __shared__ __attribute__((aligned(16))) char lds[4 * 4160];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    uint32_t s = hash(blockIdx.x * 256 + tid + 1);
    float acc = 0.f;
    char* wl = lds + (tid >> 6) * 4160;
    for (int it = 0; it < iters; ++it) {
        s = hash(s + it);
        uint32_t idx = s % n_nodes;
        if (MODE == 1) idx = __shfl(idx, lane & ~3u);
        if (MODE == 5) idx = __shfl(idx, 0);
        if (MODE == 0 || MODE == 1 || MODE == 5) {
            const float4* p = nodes + (size_t)idx * 4;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        } else if (MODE == 6) {
            const float4* p = nodes + (size_t)idx * 4;
            float4 a = p[0];
            acc += a.x + a.w;
        } else if (MODE == 7) {
            const float4* p = nodes + (size_t)idx * 4;
            float4 a = p[0], b = p[1];
            acc += a.x + b.w;
        } else if (MODE == 4) {
            const float* p = reinterpret_cast<const float*>(nodes + (size_t)idx * 4);
#pragma unroll
            for (int q = 0; q < 16; ++q) acc += p[q];
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t ij = __shfl(idx, (lane & ~3u) | j);
                float4 a = nodes[(size_t)ij * 4 + (lane & 3)];
                acc += a.x + a.w;
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t ij = __shfl(idx, (lane & ~3u) | j);
                const char* src = reinterpret_cast<const char*>(nodes) + (size_t)ij * 64 + (lane & 3) * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(wl + j * 1040), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float4* p = reinterpret_cast<const float4*>(wl + (lane & 3) * 1040 + (lane >> 2) * 64);
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        }
        s += (uint32_t)(acc != 12345.678f);
    }
    out[blockIdx.x * 256 + tid] = acc;
}

template <int MODE>
void run(const char* name, const float4* d_nodes, uint32_t n_nodes, float* d_out, int blocks, int iters, int num_cu) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_nodes, n_nodes, iters, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_nodes, n_nodes, iters, d_out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double visits = (double)blocks * 256 * iters;
    printf("%-44s %8.3f ms  %7.1f Gvisits/s  %6.2f lane-visits/clk/CU @2.4GHz\n", name, ms, visits / ms / 1e6,
           visits / (ms * 1e-3) / 2.4e9 / num_cu);
}

// streaming calibration: every thread reads 16 B, grid-strided, whole table once
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ t, size_t n4, float* out) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { float4 a = t[i]; acc += a.x + a.w; }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    uint32_t mb = argc > 1 ? atoi(argv[1]) : 16;
    uint32_t n_nodes = mb * 1024u * 1024u / 64;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int num_cu = prop.multiProcessorCount;
    std::vector<float> h((size_t)n_nodes * 16, 1.0f);
    float4* d_nodes; float* d_out;
    int blocks = num_cu * 8, iters = 400;
    CK(hipMalloc(&d_nodes, (size_t)n_nodes * 64)); CK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    CK(hipMemcpy(d_nodes, h.data(), (size_t)n_nodes * 64, hipMemcpyHostToDevice));
    printf("table %u MB (%u nodes), %d CUs, %d blocks x 256, %d iters\n", mb, n_nodes, num_cu, blocks, iters);
    if (argc > 2) {  // calibration mode for FETCH_SIZE: one streaming pass, one random-gather pass
        hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, d_nodes, (size_t)n_nodes * 4, d_out);
        CK(hipDeviceSynchronize());
        printf("k_stream read %.1f MB\n", n_nodes * 64.0 / 1e6);
        hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d_nodes, n_nodes, iters, d_out);
        CK(hipDeviceSynchronize());
        printf("k<0> gathered %.1f MB of 64-B nodes (%d x 256 x %d visits)\n", (double)blocks * 256 * iters * 64 / 1e6, blocks, iters);
        return 0;
    }
    run<0>("0 lane-private node, 4 x dwordx4", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
    run<1>("1 quad shares node, 4 x dwordx4", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
    run<5>("5 wave shares node, 4 x dwordx4", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
    run<2>("2 quad-cooperative, 4 x dwordx4 (regs)", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
    run<3>("3 quad-cooperative, 4 x load_lds + 4 ds_read", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
    run<4>("4 lane-private node, 16 x dword", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
    run<6>("6 lane-private node, 1 x dwordx4 (16 B)", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
    run<7>("7 lane-private node, 2 x dwordx4 (32 B)", d_nodes, n_nodes, d_out, blocks, iters, num_cu);
    return 0;
}
