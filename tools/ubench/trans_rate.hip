// Issue-rate micro-benchmark for the SiLU ingredients (gfx950): transcendental vs plain VALU, f32 vs f16, and whether a
// second wave's plain VALU / MFMA work hides behind a wave's transcendentals.  Timed with HIP events over a grid that
// puts W waves on every SIMD; reported as ns per wave-instruction per SIMD and relative to v_fma_f32.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define REP 32768
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, uint32_t seed)
{
    float acc[8];
    uint32_t hv[8];
    float wf = (float)threadIdx.x * 1e-3f + 1.0f;
    for (int i = 0; i < 8; ++i) { acc[i] = wf + i; hv[i] = 0x3c003c00u + i + threadIdx.x; }
    f4 macc = {0, 0, 0, 0};
    h8 ma, mb;
    for (int i = 0; i < 8; ++i) { ma[i] = (_Float16)(i + 1); mb[i] = (_Float16)wf; }
    const int wave = threadIdx.x >> 6;
    if (MODE == 7 || MODE == 8 || MODE == 10 || MODE == 13) {
        // waves 0..3 sit one per SIMD, waves 4..7 are their SIMD-mates: role A on the first four, role B on the rest
        if (wave < 4) {
            for (int r = 0; r < REP; ++r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (MODE == 10) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(wf));
                    else asm volatile("v_exp_f32 %0, %0" : "+v"(acc[i]));
                }
            }
        } else {
            for (int r = 0; r < REP; ++r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (MODE == 7) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(wf));
                    else if (MODE == 13) asm volatile("v_rcp_f32 %0, %0" : "+v"(acc[i]));
                    else macc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ma, mb, macc, 0, 0, 0);
                }
            }
        }
    } else
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(wf));
            if (MODE == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(acc[i]));
            if (MODE == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(acc[i]));
            if (MODE == 3) asm volatile("v_exp_f16 %0, %0" : "+v"(hv[i]));
            if (MODE == 4) asm volatile("v_rcp_f16 %0, %0" : "+v"(hv[i]));
            if (MODE == 5) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(hv[i]) : "v"(hv[(i + 1) & 7]));
            if (MODE == 6) {   // the scaled SiLU: exp(-t), +1, rcp, *t
                float e, d, q;
                asm volatile("v_exp_f32 %0, -%1" : "=v"(e) : "v"(acc[i]));
                asm volatile("v_add_f32 %0, 1.0, %1" : "=v"(d) : "v"(e));
                asm volatile("v_rcp_f32 %0, %1" : "=v"(q) : "v"(d));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(q));
            }
            if (MODE == 9) macc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ma, mb, macc, 0, 0, 0);
            if (MODE == 11) asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(hv[i]));
            if (MODE == 12) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %1" : "=v"(hv[i]) : "v"(acc[i]));
        }
    }
    float s = macc[0] + macc[1] + macc[2] + macc[3];
    for (int i = 0; i < 8; ++i) s += acc[i] + (float)hv[i];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}
template <int MODE>
static void run(const char* name, float* out, int threads, double* base)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, 1u);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const int waves_per_simd = threads / 256;
    const double ns = best * 1e6 / ((double)REP * 8 * waves_per_simd);   // per wave-instruction (or per 4-op SiLU) per SIMD
    if (*base == 0) *base = ns;
    printf("%-44s waves/SIMD %d  %.3f ms  %.3f ns per wave-item per SIMD  (%.2f x v_fma_f32)\n", name, waves_per_simd, best, ns, ns / *base);
}
int main()
{
    float* out;
    hipMalloc(&out, 256 * 1024 * 4);
    double base = 0;
    for (int threads : {256, 512, 1024}) {
        run<0>("v_fma_f32", out, threads, &base);
        run<1>("v_exp_f32", out, threads, &base);
        run<2>("v_rcp_f32", out, threads, &base);
        run<3>("v_exp_f16", out, threads, &base);
        run<4>("v_rcp_f16", out, threads, &base);
        run<11>("v_exp_f16_sdwa (hi half)", out, threads, &base);
        run<5>("v_pk_mul_f16", out, threads, &base);
        run<12>("v_cvt_pkrtz_f16_f32", out, threads, &base);
        run<6>("SiLU = exp,add,rcp,mul (per element group)", out, threads, &base);
        run<9>("v_mfma_f32_16x16x32_f16", out, threads, &base);
    }
    run<7>("even waves exp / odd waves fma (per instr)", out, 512, &base);
    run<8>("even waves exp / odd waves mfma", out, 512, &base);
    run<10>("even waves fma / odd waves mfma", out, 512, &base);
    run<13>("even waves exp / odd waves rcp", out, 512, &base);
    return 0;
}
