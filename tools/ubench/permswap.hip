// v_permlane16_swap_b32 semantics check (gfx950): r = __builtin_amdgcn_permlane16_swap(a, b, false, false)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o)
{
    const unsigned a = 100 + threadIdx.x, b = 200 + threadIdx.x;
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
}
int main()
{
    unsigned *d, h[128];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int r = 0; r < 2; ++r) { printf("r[%d]:", r); for (int i = 0; i < 64; i += 8) printf(" lane%02d=%u", i, h[64 * r + i]); printf("\n"); }
    return 0;
}
