// device_common.h -- typedefs, inline-asm wrappers and small device helpers shared by the kernel translation units
// (k_generic.hip, k_mbconv.hip, k_early.hip, k_mid.hip, k_tail.hip).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "kernels.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// v_exp_f32 + v_rcp_f32 (1 ulp) instead of an IEEE division: results are rounded to fp16 anyway.
static __device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
static __device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
// Scaled-domain SiLU.  Every SiLU-activated tensor is stored as T' = log2(e) * T: the host folds
// log2(e) into the producing convolution's weights and bias and 1/log2(e) into every consumer, so the
// epilogue gets t = log2(e) * x straight out of the accumulator (bias = accumulator init) and
//   log2(e) * silu(x) = t / (1 + 2^-t)   is v_exp_f32 (neg modifier) + v_add + v_rcp + v_mul.
static __device__ __forceinline__ float silu_scaled(float t)
{
    return t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-t));
}

// Two f32 -> one dword of two f16 (round to nearest even, gfx950's v_cvt_pk_f16_f32).  Written as asm because the compiler splits
// an h2 whose halves are stored separately back into two single conversions.
static __device__ __forceinline__ uint32_t cvt_pk_f16(float a, float b)
{
    uint32_t d;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// Pins the loads written before it where they are written: nothing is scheduled across.  hipcc's scheduler otherwise sinks
// a prefetch down to its first use, i.e. turns it into a plain load.  (A mask that lets ALU / MFMA / LDS instructions cross,
// 0x78F, was measured far worse: tail7's project 12 k -> 23 k cycles, head 28 k -> 55 k -- the loads moved again.)
#define PIN_VMEM() __builtin_amdgcn_sched_barrier(0)

// First tap of a depthwise accumulator: d = dot2(a, b) + c with c in its own register (VOP3P v_dot2_f32_f16).  The builtin is
// always selected as the two-address v_dot2c, which needs a v_mov per accumulator to start from the bias.
static __device__ __forceinline__ float dot2_from(uint32_t a, uint32_t b, float c)
{
    float d;
    asm("v_dot2_f32_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// Sum over the four lanes of a quad (lanes 4k .. 4k+3) on the vector ALU: two DPP quad_perm moves + adds -- __shfl_xor goes through
// the LDS crossbar (ds_bpermute: two dependent ~130-cycle round trips).  Every lane ends up with the total, in a fixed order.
static __device__ __forceinline__ float quad_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    return v;
}
// Sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), every lane getting the total: the quad sums, then row_half_mirror and
// row_mirror.  After the quad stage a quad is uniform, so lane 7-i holds what lane i^4 holds (and, one stage later, lane 15-i what
// lane i^8 holds): the additions are those of the xor butterfly v += shfl_xor(v, 1 | 2 | 4 | 8), bit for bit, without its four
// dependent ds_bpermute round trips.
static __device__ __forceinline__ float row16_sum(float v)
{
    v = quad_sum(v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

// The same on N accumulators at once, stage by stage (all exponentials, all adds, all reciprocals, all products): element by
// element the four-instruction chain exp -> add -> rcp -> mul stalls on each transcendental's latency (the compiler pads it
// with s_nop); staged, every instruction has N - 1 independent ones between it and its consumer.  Same values, bit for bit.
template <int N>
static __device__ __forceinline__ void silu_scaled_staged(float (&t)[N])
{
    // the add and the product run two values per instruction (v_pk_add_f32 / v_pk_mul_f32: same IEEE results, half the issue
    // slots); the transcendentals have no packed form
    float e[N];
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = __builtin_amdgcn_exp2f(-t[i]);
#pragma unroll
    for (int i = 0; i + 1 < N; i += 2) {
        f2 v = {e[i], e[i + 1]};
        v = v + (f2){1.0f, 1.0f};
        e[i] = v.x; e[i + 1] = v.y;
    }
    if (N & 1) e[N - 1] = 1.0f + e[N - 1];
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = __builtin_amdgcn_rcpf(e[i]);
#pragma unroll
    for (int i = 0; i + 1 < N; i += 2) {
        f2 v = {t[i], t[i + 1]}, r = {e[i], e[i + 1]};
        v = v * r;
        t[i] = v.x; t[i + 1] = v.y;
    }
    if (N & 1) t[N - 1] = t[N - 1] * e[N - 1];
}

typedef _Float16 h2 __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ uint4 gate_h8(uint4 x, f4 g0, f4 g1)
{
    // eight fp16 activations times their fp32 gates -> eight fp16 (each product in fp32, rounded once):
    // v_fma_mixlo/hi_f16 read the fp16 half directly and write one half of the destination.  The result is an MFMA
    // operand: the VALU-write -> MFMA-read wait states are not padded by the compiler inside asm, hence the s_nop.
    uint4 d;
    asm("v_fma_mixlo_f16 %0, %4, %8, 0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %4, %9, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %5, %10, 0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %5, %11, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %2, %6, %12, 0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %2, %6, %13, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, %7, %14, 0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %3, %7, %15, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "s_nop 1"
        : "=&v"(d.x), "=&v"(d.y), "=&v"(d.z), "=&v"(d.w)
        : "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w), "v"(g0[0]), "v"(g0[1]), "v"(g0[2]), "v"(g0[3]), "v"(g1[0]), "v"(g1[1]),
          "v"(g1[2]), "v"(g1[3]));
    return d;
}

// fp16 x fp32 + fp32 -> fp32 in ONE VALU instruction (v_fma_mix_f32 reads the low/high half of a packed
// fp16 pair directly): the depthwise taps need no v_cvt_f32_f16 at all.
static __device__ __forceinline__ float fma_mix_lo(uint32_t h2, float w, float acc)
{
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(w), "v"(acc));
    return d;
}
static __device__ __forceinline__ float fma_mix_hi(uint32_t h2, float w, float acc)
{
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(w), "v"(acc));
    return d;
}


#define GLOBAL_AS __attribute__((address_space(1)))
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
// a pointer that came out of memory, moved to SGPRs (wave-uniform by construction) and to the global address space
template <typename T>
static __device__ __forceinline__ const GLOBAL_AS T* sgpr_ptr(const void* p)
{
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const GLOBAL_AS T*>(((uint64_t)hi << 32) | lo);
}
// global load from a wave-uniform base plus a 32-bit byte offset: global_load ... v_off, s[base:base+1] (no 64-bit
// address arithmetic in VGPRs)
template <typename T>
static __device__ __forceinline__ T gload(const GLOBAL_AS void* base, unsigned byte_off)
{
    return *reinterpret_cast<const GLOBAL_AS T*>(reinterpret_cast<const GLOBAL_AS char*>(base) + byte_off);
}
// Workgroup barrier that only waits for this wave's LDS traffic (lgkmcnt), NOT for its outstanding global loads:
// __syncthreads() also drains vmcnt, which would serialise every weight prefetch issued across a phase boundary.
// No global data is exchanged between the threads of this kernel, so the LDS-only form is sufficient.
#define T7_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

#define LAUNCH_CHECK()                          \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

