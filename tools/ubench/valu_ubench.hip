// Microbenchmark: the fp32 vector rate the traversal arithmetic can reach on this GPU.  The path may not
// contract a*b+c (bit-exact results), so its peak is one v_add_f32 / v_mul_f32 per lane and issue slot, not an FMA.
// Every thread runs independent add and multiply chains (no memory traffic); the launch puts W waves on every SIMD
// (W = 1, 2, 4, 8) and reports lane-operations per second and cycles per wave instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int CHAINS = 16;      // independent accumulators per lane (each a dependent chain of its own)
constexpr int UNROLL = 8;       // add + mul rounds per loop iteration

// FMA = 0: v_add_f32 / v_mul_f32 alternating.  FMA = 1: v_fma_f32 (the device's peak, for comparison).
template <int FMA>
__global__ __launch_bounds__(256) void k_valu(float* out, int iters, float c0, float c1) {
    float a[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) a[i] = (float)(threadIdx.x + i) * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
                if (FMA) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
                } else {
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
                    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FMA>
double run(float* d_out, int num_cu, int waves_per_simd, int iters, double* cyc) {
    // 256-thread workgroups = one wave per SIMD; waves_per_simd workgroups per CU
    const int blocks = num_cu * waves_per_simd;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_valu<FMA>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 0.999f, 1e-3f);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_valu<FMA>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 0.999f, 1e-3f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double insts_per_wave = (double)iters * UNROLL * CHAINS * (FMA ? 1 : 2);
    const double lane_ops = insts_per_wave * 64.0 * blocks * 4;
    // cycles per wave instruction per SIMD at the nominal 2.4 GHz
    *cyc = (best * 1e-3) * 2.4e9 / (insts_per_wave * waves_per_simd);
    return lane_ops / (best * 1e-3);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int num_cu = prop.multiProcessorCount;
    float* d_out;
    CK(hipMalloc(&d_out, (size_t)num_cu * 8 * 256 * sizeof(float)));
    printf("{\"device\": \"%s\", \"num_cu\": %d, \"clock_mhz_nominal\": 2400, \"rows\": [\n", prop.gcnArchName, num_cu);
    const int ws[4] = {1, 2, 4, 8};
    for (int k = 0; k < 4; ++k) {
        double c0, c1;
        const double r0 = run<0>(d_out, num_cu, ws[k], 20000, &c0);
        const double r1 = run<1>(d_out, num_cu, ws[k], 20000, &c1);
        printf(" {\"waves_per_simd\": %d, \"add_mul_T_lane_ops_per_s\": %.2f, \"add_mul_cycles_per_wave_inst\": %.2f, "
               "\"fma_T_lane_insts_per_s\": %.2f, \"fma_cycles_per_wave_inst\": %.2f}%s\n",
               ws[k], r0 / 1e12, c0, r1 / 1e12, c1, k == 3 ? "" : ",");
    }
    printf("]}\n");
    CK(hipFree(d_out));
    return 0;
}
