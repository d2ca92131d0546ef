// How fast can every workgroup stream the SAME L2-resident weight set (what tail7 / proj_patch do)?
// G workgroups x 512 threads read a W-byte buffer front to back with 16-byte loads (one wave = 1 KB per instruction,
// DEPTH instructions in flight per wave), optionally starting at a per-workgroup offset (rotation).  Cycles via s_memtime.
//   hipcc -O3 --offload-arch=gfx950 l2_stream.hip -o l2_stream && ./l2_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
template <int DEPTH>
__global__ __launch_bounds__(512) void k(const u4* __restrict__ w, int nchunks /* 8 KB chunks */, int rot, uint32_t* out, long long* clk)
{
    const int tid = threadIdx.x;
    u4 acc = {0, 0, 0, 0};
    const int start = rot ? (int)((blockIdx.x * 2654435761u) % (unsigned)nchunks) : 0;
    long long t0 = __builtin_readcyclecounter();
    for (int c0 = 0; c0 < nchunks; c0 += DEPTH) {
        u4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            int c = start + c0 + d;
            c = c >= nchunks ? c - nchunks : c;
            v[d] = w[(size_t)c * 512 + tid];
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { acc.x ^= v[d].x; acc.y += v[d].y; acc.z ^= v[d].z; acc.w += v[d].w; }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + tid] = acc.x + acc.y + acc.z + acc.w;
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}
template <int DEPTH>
static void run(const u4* w, size_t bytes, int G, int rot, uint32_t* out, long long* clk)
{
    const int nchunks = (int)(bytes / 8192);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k<DEPTH>, dim3(G), dim3(512), 0, 0, w, nchunks, rot, out, clk);
    hipDeviceSynchronize();
    std::vector<long long> h(G);
    hipMemcpy(h.data(), clk, G * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("W %5zu KB  G %3d  depth %2d  rot %d : median %8lld cycles = %5.1f B/clk per workgroup (slowest %5.1f)\n", bytes >> 10, G, DEPTH, rot,
           h[G / 2], (double)bytes / h[G / 2], (double)bytes / h[G - 1]);
}
// the same stream with 8-byte loads (one wave = 512 B per instruction): what tail7's squeeze-excite weight requests look like
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
template <int DEPTH>
__global__ __launch_bounds__(512) void k8(const u2* __restrict__ w, int nchunks /* 4 KB chunks */, uint32_t* out, long long* clk)
{
    const int tid = threadIdx.x;
    u2 acc = {0, 0};
    long long t0 = __builtin_readcyclecounter();
    for (int c0 = 0; c0 < nchunks; c0 += DEPTH) {
        u2 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = w[(size_t)(c0 + d) * 512 + tid];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { acc.x ^= v[d].x; acc.y += v[d].y; }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + tid] = acc.x + acc.y;
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}
template <int DEPTH>
static void run8(const u2* w, size_t bytes, int G, uint32_t* out, long long* clk)
{
    const int nchunks = (int)(bytes / 4096);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k8<DEPTH>, dim3(G), dim3(512), 0, 0, w, nchunks, out, clk);
    hipDeviceSynchronize();
    std::vector<long long> h(G);
    hipMemcpy(h.data(), clk, G * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("8-byte loads: W %5zu KB  G %3d  depth %2d : median %8lld cycles = %5.1f B/clk per workgroup (slowest %5.1f)\n", bytes >> 10, G, DEPTH,
           h[G / 2], (double)bytes / h[G / 2], (double)bytes / h[G - 1]);
}
int main()
{
    const size_t maxb = 8u << 20;
    u4* w; uint32_t* out; long long* clk;
    hipMalloc(&w, maxb); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8);
    hipMemset(w, 1, maxb);
    for (size_t bytes : {(size_t)448 << 10, (size_t)7 << 20})
        for (int G : {1, 32, 128, 256})
            for (int rot : {0, 1}) {
                run<4>(w, bytes, G, rot, out, clk);
                run<12>(w, bytes, G, rot, out, clk);
            }
    for (size_t bytes : {(size_t)216 << 10, (size_t)7 << 20})
        for (int G : {1, 128, 256}) {
            run8<12>((const u2*)w, bytes, G, out, clk);
            run8<24>((const u2*)w, bytes, G, out, clk);
        }
    return 0;
}
