// v_mfma_f32_4x4x4_16B_f16 on gfx950: operand layout and issue rate -- the ingredients of a depthwise convolution on the matrix
// pipe WITHOUT the 1/16 occupancy of a block-diagonal 16x16x32 MFMA: the instruction multiplies 16 independent 4x4x4 blocks, so
// a block can be a CHANNEL (its own weights), A a 4x4 Toeplitz slice of the channel's taps, B four input columns of four rows.
//   part 1  layout: random small integers, every (block, i, j) of D compared with the host under the documented layout
//           A: lane = 4 * block + i holds A[i][0..3];  B: lane = 4 * block + j holds B[0..3][j];  D: lane = 4 * block + j, VGPR i.
//   part 2  cycles (s_memtime) per instruction: back to back on 1 / 4 accumulators, with n plain VALU instructions of the same
//           wave between two MFMAs, and beside a SIMD-mate wave that only runs VALU (v_fma / v_exp).
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma4x4 mfma4x4.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void layout_k(const _Float16* A, const _Float16* B, float* D)   // A[16][4][4] (b, i, k), B[16][4][4] (b, k, j)
{
    const int lane = threadIdx.x, b = lane >> 2, r = lane & 3;
    h4 a, bb;
    for (int k = 0; k < 4; ++k) { a[k] = A[(b * 4 + r) * 4 + k]; bb[k] = B[(b * 4 + k) * 4 + r]; }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x4f16(a, bb, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(b * 4 + i) * 4 + r] = c[i];   // hypothesis: VGPR i = row i, lane low bits = column j
}

#define REP 2048
// MODE 0: 4x4x4 on 4 accumulators; 1: 4x4x4 on one accumulator (dependent); 2: 16x16x32 on 4 accumulators (reference);
// 10+n: 4x4x4 (4 accumulators) with n v_fma_f32 of the same wave after each MFMA;
// 30+n: 16x16x32 with n v_fma_f32 after each;  50: waves 0-3 4x4x4, waves 4-7 v_fma;  51: waves 0-3 4x4x4, waves 4-7 v_exp
template <int MODE>
__global__ __launch_bounds__(512) void rate_k(float* out, long long* cyc)
{
    const int wave = threadIdx.x >> 6;
    h4 a4 = {(_Float16)1.0f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)(threadIdx.x & 3)}, b4 = a4;
    h8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(0.01f * i); b8[i] = (_Float16)(0.02f * i + (threadIdx.x & 7)); }
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 1.0f + 0.001f * threadIdx.x + i;
    const float wf = 1.0001f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (MODE >= 50 && wave >= 4) {
        for (int r = 0; r < REP; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 50) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(v[i]) : "v"(wf));
                else asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            }
    } else {
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                constexpr int NV = MODE >= 30 && MODE < 50 ? MODE - 30 : (MODE >= 10 && MODE < 30 ? MODE - 10 : 0);
                if (MODE == 2 || (MODE >= 30 && MODE < 50)) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
                else acc[MODE == 1 ? 0 : i] = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, acc[MODE == 1 ? 0 : i], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NV; ++n) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(v[n & 7]) : "v"(wf));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
static void run(const char* name, float* out, long long* cyc, int threads)
{
    hipLaunchKernelGGL(rate_k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc);
    hipLaunchKernelGGL(rate_k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc);
    hipDeviceSynchronize();
    long long h[256 * 8];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    double lo = 0, hi = 0; int nlo = 0, nhi = 0;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < nw; ++w) { if (w < 4) { lo += h[b * 8 + w]; ++nlo; } else { hi += h[b * 8 + w]; ++nhi; } }
    printf("%-66s waves/SIMD %d  waves 0-3: %.2f cycles per MFMA", name, nw / 4, lo / nlo / (REP * 4.0));
    if (nhi) printf("   waves 4-7: %.2f cycles per %s", hi / nhi / (REP * (MODE >= 50 ? 8.0 : 4.0)), MODE >= 50 ? "VALU instr" : "MFMA");
    printf("\n");
}

int main()
{
    // ---- layout
    _Float16 hA[256], hB[256];
    float hD[256], ref[256];
    srand(1);
    for (int i = 0; i < 256; ++i) { hA[i] = (_Float16)(float)(rand() % 7 - 3); hB[i] = (_Float16)(float)(rand() % 5 - 2); }
    for (int b = 0; b < 16; ++b)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                float s = 0;
                for (int k = 0; k < 4; ++k) s += (float)hA[(b * 4 + i) * 4 + k] * (float)hB[(b * 4 + k) * 4 + j];
                ref[(b * 4 + i) * 4 + j] = s;
            }
    _Float16 *dA, *dB; float* dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout_k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
    printf("layout A: lane=4b+i holds A[i][0..3]; B: lane=4b+j holds B[0..3][j]; D: lane=4b+j, VGPR i  ->  %s (%d of 256 differ)\n",
           bad ? "MISMATCH" : "confirmed", bad);
    if (bad) {   // print one block for diagnosis
        for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) printf(" %6.1f/%6.1f", hD[i * 4 + j], ref[i * 4 + j]); printf("\n"); }
    }
    // ---- rates
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    hipMemset(cyc, 0, 256 * 8 * 8);
    run<0>("4x4x4_16B_f16, 4 accumulators", out, cyc, 256);
    run<1>("4x4x4_16B_f16, 1 accumulator (dependent)", out, cyc, 256);
    run<2>("16x16x32_f16, 4 accumulators", out, cyc, 256);
    run<0>("4x4x4_16B_f16, 4 accumulators", out, cyc, 512);
    run<2>("16x16x32_f16, 4 accumulators", out, cyc, 512);
    run<11>("4x4x4 + 1 v_fma after each (same wave)", out, cyc, 256);
    run<12>("4x4x4 + 2 v_fma after each (same wave)", out, cyc, 256);
    run<13>("4x4x4 + 3 v_fma after each (same wave)", out, cyc, 256);
    run<14>("4x4x4 + 4 v_fma after each (same wave)", out, cyc, 256);
    run<32>("16x16x32 + 2 v_fma after each (same wave)", out, cyc, 256);
    run<34>("16x16x32 + 4 v_fma after each (same wave)", out, cyc, 256);
    run<12>("4x4x4 + 2 v_fma after each (same wave)", out, cyc, 512);
    run<50>("waves 0-3 4x4x4, waves 4-7 v_fma_f32", out, cyc, 512);
    run<51>("waves 0-3 4x4x4, waves 4-7 v_exp_f32", out, cyc, 512);
    return 0;
}
