// What does a vector-memory INSTRUCTION cost a CU, by access width?  256 workgroups x 512 threads (one per CU), every wave issuing N
// back-to-back loads or stores of 2 / 4 / 8 / 16 bytes per lane to wave-contiguous addresses (L2-resident 8 MB buffer).
// Prints cycles per wave-instruction per CU (8 waves issue concurrently: CU rate = cycles / (8 N)).
//   hipcc -O3 --offload-arch=gfx950 vmem_rate.hip -o vmem_rate && ./vmem_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
template <typename T, int N, bool STORE>
__global__ __launch_bounds__(512) void k(T* __restrict__ buf, size_t elems_per_wg, uint32_t* out, long long* clk)
{
    const int tid = threadIdx.x;
    T* p = buf + (size_t)blockIdx.x * elems_per_wg + tid;
    T acc[N];
    long long t0 = __builtin_readcyclecounter();
    if (STORE) {
        T v; __builtin_memset(&v, 0, sizeof v); reinterpret_cast<unsigned char*>(&v)[0] = (unsigned char)tid;
#pragma unroll
        for (int i = 0; i < N; ++i) p[(size_t)i * 512] = v;
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = p[(size_t)i * 512];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    if (!STORE) {
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) s += reinterpret_cast<unsigned char*>(&acc[i])[0];
        out[blockIdx.x * 512 + tid] = s;
    }
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}
template <typename T, int N, bool STORE>
static void run(void* buf, uint32_t* out, long long* clk, const char* name)
{
    const int G = 256;
    const size_t per = (size_t)N * 512;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k<T, N, STORE>), dim3(G), dim3(512), 0, 0, (T*)buf, per, out, clk);
    hipDeviceSynchronize();
    std::vector<long long> h(G);
    hipMemcpy(h.data(), clk, G * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-6s %2zu B/lane x %3d per wave: median %7lld cycles = %5.1f cycles per wave-instruction per CU, %5.1f B/clk\n", name, sizeof(T), N, h[G / 2],
           (double)h[G / 2] / (8.0 * N), (double)(sizeof(T) * 512 * N) / h[G / 2]);
}
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
int main()
{
    void* buf; uint32_t* out; long long* clk;
    hipMalloc(&buf, (size_t)256 * 64 * 512 * 16); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8);
    hipMemset(buf, 1, (size_t)256 * 64 * 512 * 16);
    run<uint16_t, 48, false>(buf, out, clk, "load");
    run<uint32_t, 48, false>(buf, out, clk, "load");
    run<u2, 48, false>(buf, out, clk, "load");
    run<u4, 48, false>(buf, out, clk, "load");
    run<uint16_t, 48, true>(buf, out, clk, "store");
    run<uint32_t, 48, true>(buf, out, clk, "store");
    run<u2, 48, true>(buf, out, clk, "store");
    run<u4, 48, true>(buf, out, clk, "store");
    return 0;
}
