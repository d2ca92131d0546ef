// v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands on gfx950: operand layout check and issue rate.
//   hypothesis H1: lane l holds row/column l % 16, K = 32 (l / 16) .. + 31 as 32 consecutive bytes (8 VGPRs)
//   hypothesis H2: lane l holds K = 16 (l / 16) .. + 15 (first 4 VGPRs) and 64 + 16 (l / 16) .. + 15 (last 4 VGPRs)
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_f8 mfma_f8.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ void lay_k(const uint8_t* A, const uint8_t* B, float* D, int hyp)   // A[16][128], B[128][16] as e4m3 bytes
{
    const int l = threadIdx.x, r = l & 15, kb = l >> 4;
    union { uint8_t b[32]; v8i v; } a, bb;
    for (int e = 0; e < 32; ++e) {
        const int k = hyp == 1 ? 32 * kb + e : (e < 16 ? 16 * kb + e : 64 + 16 * kb + (e - 16));
        a.b[e] = A[r * 128 + k];
        bb.b[e] = B[k * 16 + r];
    }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a.v, bb.v, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int i = 0; i < 4; ++i) D[(4 * kb + i) * 16 + r] = c[i];   // D: lane column r, rows 4 (l / 16) + i
}

#define REP 4096
template <int MODE>   // 0: f8f6f4 16x16x128 (4 accumulators), 1: f16 16x16x32, 2: fp8_fp8 16x16x32
__global__ __launch_bounds__(256) void rate_k(float* out, long long* cyc)
{
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + i; b[i] = 0x30303030 + (threadIdx.x & 3); }
    h8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.01f * i); hb[i] = (_Float16)(0.02f * i); }
    const long la = 0x3838383838383838L, lb = 0x3030303030303030L;
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0);
            if (MODE == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(la, lb, acc[i], 0, 0, 0);
        }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE>
static void run(const char* nm, float* out, long long* cyc)
{
    hipLaunchKernelGGL(rate_k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc);
    hipDeviceSynchronize();
    long long h[1024];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 1024; ++i) s += h[i];
    printf("%-40s %.2f cycles per MFMA (one wave per SIMD, 4 accumulators)\n", nm, s / 1024 / (REP * 4.0));
}
static uint8_t e4m3(int v2)   // v2 = value * 2 in {-4..4}: 0, +-0.5, +-1, +-1.5, +-2
{
    static const uint8_t tab[5] = {0x00, 0x30, 0x38, 0x3C, 0x40};
    const int a = v2 < 0 ? -v2 : v2;
    return (uint8_t)(tab[a] | (v2 < 0 ? 0x80 : 0));
}
int main()
{
    uint8_t hA[16 * 128], hB[128 * 16];
    int vA[16 * 128], vB[128 * 16];
    srand(3);
    for (int i = 0; i < 16 * 128; ++i) { vA[i] = rand() % 9 - 4; vB[i] = rand() % 9 - 4; hA[i] = e4m3(vA[i]); hB[i] = e4m3(vB[i]); }
    float ref[256], hD[256];
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float s = 0;
            for (int k = 0; k < 128; ++k) s += 0.25f * vA[i * 128 + k] * vB[k * 16 + j];
            ref[i * 16 + j] = s;
        }
    uint8_t *dA, *dB; float* dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int hyp = 1; hyp <= 2; ++hyp) {
        hipLaunchKernelGGL(lay_k, dim3(1), dim3(64), 0, 0, dA, dB, dD, hyp);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
        printf("16x16x128 f8f6f4 (e4m3, scales 1.0) layout H%d: %s (%d of 256 differ; D[0][0] %.2f ref %.2f)\n", hyp, bad ? "mismatch" : "CONFIRMED", bad, hD[0], ref[0]);
    }
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    run<0>("v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3)", out, cyc);
    run<1>("v_mfma_f32_16x16x32_f16", out, cyc);
    run<2>("v_mfma_f32_16x16x32_fp8_fp8", out, cyc);
    return 0;
}
