// Issue-rate micro-benchmark for the VALU ops the depthwise kernels choose between (gfx950).
// One wave per SIMD (256 threads x 256 workgroups), N dependent-free instructions per loop, cycles via s_memtime.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 64
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* clk, uint32_t seed)
{
    float acc[8];
    f2 pacc[8];
    uint32_t a = seed + threadIdx.x, b = seed * 3 + threadIdx.x;
    float wf = (float)threadIdx.x * 1e-3f;
    f2 pa = {wf, wf + 1.f}, pb = {0.5f, 0.25f};
    for (int i = 0; i < 8; ++i) { acc[i] = i; pacc[i] = (f2){(float)i, (float)i}; }
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) acc[i] = __builtin_amdgcn_fdot2(*(h2*)&a, *(h2*)&b, acc[i], false);
                if (MODE == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[i]) : "v"(a), "v"(wf));
                if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pacc[i]) : "v"(pa), "v"(pb));
                if (MODE == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(wf), "v"(wf));
                if (MODE == 4) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (MODE == 5) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(b));
            }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i] + pacc[i][0] + pacc[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)a;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
int main()
{
    float* out; long long* clk;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 256 * 8);
    const char* names[] = {"fdot2 builtin", "v_fma_mix_f32", "v_pk_fma_f32", "v_fma_f32", "v_dot2c_f32_f16", "v_pk_fma_f16"};
    long long h[256];
    for (int m = 0; m < 6; ++m) {
        for (int it = 0; it < 2; ++it) {
            switch (m) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, clk, 1u); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, clk, 1u); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, clk, 1u); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, clk, 1u); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, out, clk, 1u); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(256), 0, 0, out, clk, 1u); break;
            }
            hipDeviceSynchronize();
        }
        hipMemcpy(h, clk, 256 * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
        printf("%-18s %.2f cycles / instruction (one wave per SIMD)\n", names[m], s / 256 / (REP * 32.0));
    }
    return 0;
}
