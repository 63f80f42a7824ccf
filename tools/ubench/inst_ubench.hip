// Microbenchmark: issue cost of the instruction kinds the traversal steps are made of, in SIMD cycles per wave
// instruction at 8 waves per SIMD (256-thread workgroups, 8 per CU), no memory traffic.  Each variant repeats a
// block of independent instructions; cycles = launch time x clock / (instructions per wave x waves per SIMD), at
// the nominal 2.4 GHz (the chip may hold a lower clock, so ratios between rows are what to read).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int N = 8;   // independent registers per kind

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int KIND>
__global__ __launch_bounds__(256) void k_inst(float* out, int iters, float c0, float c1) {
    float a[N], b[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { a[i] = (float)(threadIdx.x + i) * 1e-3f; b[i] = (float)(threadIdx.x * 3 + i) * 1e-3f; }
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pa[N], pc = {c1, c1}, pd = {c0, c0};
#pragma unroll
    for (int i = 0; i < N; ++i) { pa[i].x = a[i]; pa[i].y = b[i]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#define ONE(i)                                                                                                                   \
    if (KIND == 0) { asm volatile("v_sub_f32 %0, %0, %1\n v_mul_f32 %0, %0, %2" : "+v"(a[i]) : "v"(c1), "v"(c0)); }              \
    if (KIND == 1) { asm volatile("v_min_f32 %0, %0, %1\n v_max_f32 %0, %0, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c0)); }            \
    if (KIND == 2) { asm volatile("v_max3_f32 %0, %0, %1, %2\n v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c0)); }  \
    if (KIND == 3) { asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_le_f32 vcc, %1, %0" : : "v"(a[i]), "v"(b[i]) : "vcc"); }     \
    if (KIND == 4) { asm volatile("v_cmp_lt_f32 s[40:41], %0, %1\n v_cmp_le_f32 s[42:43], %1, %0" : : "v"(a[i]), "v"(b[i]) : "s40", "s41", "s42", "s43"); } \
    if (KIND == 5) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(b[i])); }   \
    if (KIND == 6) { asm volatile("v_cndmask_b32 %0, %0, %1, s[40:41]\n v_cndmask_b32 %0, %1, %0, s[42:43]" : "+v"(a[i]) : "v"(b[i])); } \
    if (KIND == 7) { asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc"); } \
    if (KIND == 8) { asm volatile("v_cmp_lt_f32 s[40:41], %0, %1\n v_cndmask_b32 %0, %0, %1, s[40:41]" : "+v"(a[i]) : "v"(b[i]) : "s40", "s41"); } \
    if (KIND == 9) { asm volatile("v_addc_co_u32 %0, vcc, %0, 0, vcc\n v_addc_co_u32 %0, vcc, %0, 0, vcc" : "+v"(a[i]) : : "vcc"); } \
    if (KIND == 10) { asm volatile("v_lshl_add_u32 %0, %0, 7, %1\n v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c0)); } \
    if (KIND == 13) { asm volatile("v_min_u32 %0, %0, %1\n v_max_u32 %0, %0, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c0)); }            \
    if (KIND == 14) { asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc"); } \
    if (KIND == 16) { asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1" : "+v"(pa[i]) : "v"(pc)); }             \
    if (KIND == 17) { asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_mul_f32 %0, %0, %2" : "+v"(pa[i]) : "v"(pc), "v"(pd)); } \
    if (KIND == 18) { asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_mul_f32 %0, %0, %2 op_sel_hi:[1,0]" : "+v"(pa[i]) : "v"(pc), "v"(pd)); } \
    if (KIND == 15) { asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0)); }
            REP8(ONE)
#undef ONE
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) s += a[i] + pa[i].x + pa[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(float* d_out, int num_cu, const char* name, int per_block) {
    const int waves_per_simd = 8, iters = 4000;
    const int blocks = num_cu * waves_per_simd;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_inst<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 0.999f, 1e-3f);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_inst<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 0.999f, 1e-3f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double insts_per_wave = (double)iters * 4 * N * per_block;
    const double cyc = (best * 1e-3) * 2.4e9 / (insts_per_wave * waves_per_simd);
    printf(" {\"kind\": \"%s\", \"cycles_per_wave_inst\": %.2f},\n", name, cyc);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int num_cu = prop.multiProcessorCount;
    float* d_out;
    CK(hipMalloc(&d_out, (size_t)num_cu * 8 * 256 * sizeof(float)));
    printf("{\"device\": \"%s\", \"waves_per_simd\": 8, \"rows\": [\n", prop.gcnArchName);
    run<0>(d_out, num_cu, "v_sub_f32 + v_mul_f32 (dependent pair)", 2);
    run<15>(d_out, num_cu, "v_mul_f32 x2", 2);
    run<1>(d_out, num_cu, "v_min_f32 + v_max_f32", 2);
    run<2>(d_out, num_cu, "v_max3_f32 + v_min3_f32", 2);
    run<13>(d_out, num_cu, "v_min_u32 + v_max_u32", 2);
    run<3>(d_out, num_cu, "v_cmp_f32 -> vcc x2", 2);
    run<4>(d_out, num_cu, "v_cmp_f32 -> sgpr pair x2", 2);
    run<5>(d_out, num_cu, "v_cndmask_b32 vcc x2", 2);
    run<6>(d_out, num_cu, "v_cndmask_b32 sgpr pair x2", 2);
    run<7>(d_out, num_cu, "v_cmp -> vcc + dependent v_cndmask", 2);
    run<14>(d_out, num_cu, "v_cmp -> vcc + s_nop 1 + v_cndmask (3 counted)", 3);
    run<8>(d_out, num_cu, "v_cmp -> sgpr + dependent v_cndmask", 2);
    run<9>(d_out, num_cu, "v_addc_co_u32 x2", 2);
    run<10>(d_out, num_cu, "v_lshl_add_u32 + v_and_or_b32", 2);
    run<16>(d_out, num_cu, "v_pk_mul_f32 x2 (two floats each)", 2);
    run<17>(d_out, num_cu, "v_pk_add_f32(neg) + v_pk_mul_f32 (dependent pair, two floats each)", 2);
    run<18>(d_out, num_cu, "v_pk_add_f32(neg, broadcast lo) + v_pk_mul_f32 (broadcast lo)", 2);
    printf(" {\"kind\": \"end\", \"cycles_per_wave_inst\": 0}]}\n");
    CK(hipFree(d_out));
    return 0;
}
