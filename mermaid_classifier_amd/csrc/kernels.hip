// kernels.hip -- hand-written CDNA4 (gfx950) kernels for the EfficientNet-B0 patch
// feature-extraction path.  gfx950 only: 64-wide wavefronts, v_mfma_f32_16x16x32_f16,
// v_mfma_f32_16x16x4_f32, LDS.  No CUDA compatibility paths.
//
// Data layout in HBM: activations are NHWC fp16 ([patch][y][x][channel]); channel counts are
// multiples of 8 so every lane moves 16-byte vectors.  Accumulation is always fp32.
//
// What each kernel replaces in the reference's call graph (pyspacer EfficientNet.extract_features,
// invoked at scripts/build_feature_bucket.py:434):
//   stem_conv_kernel   transformation() + _conv_stem + _bn0 + swish
//   pw_gemm_kernel     _expand_conv+_bn0+swish | SE-scale + _project_conv+_bn2(+skip) | _conv_head+_bn1+swish+avgpool
//   dwconv_kernel      _depthwise_conv + _bn1 + swish, plus the squeeze-excite partial sums
//   se_gate_kernel     adaptive_avg_pool2d + _se_reduce + swish + _se_expand + sigmoid
//   mlp_gemm_f32_kernel / calibrate_kernel   CalibratedHead.forward (inference/head.py:66-89)
//   crop_kernel        pyspacer crop_patches (reflect pad + slice)
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

// ---------------------------------------------------------------------------------------------
// Stem: u8 HWC patch -> conv3x3 stride 2 (TF-same: pad right/bottom by 1) -> +bias -> SiLU -> fp16
// One workgroup = 16x16 output pixels of one patch; wave w owns output rows 4w..4w+3, one MFMA
// fragment (16 pixels x 32 channels) per row.  The 33x33x3 input tile is staged in LDS as exact
// fp16 integers (u8 - 128); normalisation (x/255-mean)/std is folded into weights and bias on the
// host, and padded pixels hold 255*mean-128 so they contribute exactly the folded zero.
// K packing (32 slots = 4 lane-quarters x 8): quarter q<3 = kernel row q, bytes 0..7 of the 9-byte
// (kx,c) run; quarter 3 = byte 8 of rows 0,1,2 then zeros.  Weights are packed to match on the host.
// ---------------------------------------------------------------------------------------------
#define STEM_TILE 16
#define STEM_IN (2 * STEM_TILE + 1)   // 33
#define STEM_ROWH 104                 // halves per LDS row (99 used), keeps rows 16-B aligned

template <int NT>   // NT fragments of 16 output channels: 2 for B0 (32), 3 for B4 (48)
__global__ __launch_bounds__(256) void stem_conv_kernel(const uint8_t* __restrict__ patches,  // [B][224][224][3]
                                                        const _Float16* __restrict__ w,        // [16 NT][32] (n, kslot)
                                                        const float* __restrict__ bias,        // [16 NT]
                                                        const float* __restrict__ padval,      // [3]  255*mean-128
                                                        _Float16* __restrict__ out)            // [B][112][112][16 NT]
{
    __shared__ __attribute__((aligned(16))) _Float16 tile[STEM_IN * STEM_ROWH];
    const int tx = blockIdx.x, ty = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x;
    const uint8_t* img = patches + (size_t)b * (224 * 224 * 3);
    const int iy0 = ty * 32, ix0 = tx * 32;
    const float pv0 = padval[0], pv1 = padval[1], pv2 = padval[2];
    // stage: 33 rows x 25 dwords
    for (int i = tid; i < STEM_IN * 25; i += 256) {
        const int r = i / 25, d = i - r * 25;
        const int iy = iy0 + r;
        const int boff = ix0 * 3 + d * 4;  // byte offset inside the image row
        uint32_t word = 0;
        const bool row_ok = iy < 224;
        if (row_ok && boff < 672) word = *reinterpret_cast<const uint32_t*>(img + (size_t)iy * 672 + boff);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int bb = d * 4 + e;  // byte inside the tile row
            if (bb < 99) {
                const int col = ix0 + bb / 3;
                const int c = bb % 3;
                float v;
                if (row_ok && col < 224) v = (float)((word >> (8 * e)) & 0xffu) - 128.0f;
                else v = (c == 0) ? pv0 : (c == 1 ? pv1 : pv2);
                tile[r * STEM_ROWH + bb] = (_Float16)v;
            }
        }
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    // weight fragments (A operand): rows = output channels
    h8 wf[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wf[t] = *reinterpret_cast<const h8*>(w + (t * 16 + m) * 32 + q * 8);
    float bs[4 * NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) bs[t * 4 + j] = bias[q * 4 * NT + t * 4 + j];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int oyl = wave * 4 + f;  // local output row
        h8 a;
        if (q < 3) {
            const _Float16* src = tile + (2 * oyl + q) * STEM_ROWH + 6 * m;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = src[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = (_Float16)0.0f;
#pragma unroll
            for (int j = 0; j < 3; ++j) a[j] = tile[(2 * oyl + j) * STEM_ROWH + 6 * m + 8];
        }
        f4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = (f4){bs[t * 4], bs[t * 4 + 1], bs[t * 4 + 2], bs[t * 4 + 3]};  // bias = accumulator init
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t], a, acc[t], 0, 0, 0);
        }
        const int oy = ty * 16 + oyl, ox = tx * 16 + m;
        _Float16* op = out + (((size_t)b * 112 + oy) * 112 + ox) * (16 * NT) + q * 4 * NT;
        if (NT == 2) {
            h8 o;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) o[t * 4 + j] = (_Float16)silu_scaled(acc[t][j]);
            *reinterpret_cast<h8*>(op) = o;
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                h4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (_Float16)silu_scaled(acc[t][j]);
                *reinterpret_cast<h4*>(op + 4 * t) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Pointwise (1x1) convolution as an MFMA GEMM:  Y[m][n] = epi( sum_k X[m][k] * W[n][k] + bias[n] ).
// Operands are swapped (weights = MFMA A operand, activations = B operand) so that the fp32 result
// fragment holds, per lane, 4*NT CONSECUTIVE output channels of one pixel -> 8-byte stores that
// coalesce to full lines.  Within a chunk of 16*NT channels the host permutes weight rows:
//   fragment row (t*16 + 4q + j)  <->  channel (chunk*16NT + q*4NT + 4t + j).
// Weights are packed on the host in FRAGMENT ORDER: the 1 KB a wave feeds to one MFMA (16 rows x
// 32 k, lane-linear) is contiguous, at ((chunk*KS32 + kstep)*NT + t) KB.  A workgroup (4 waves,
// 64*MT rows, one chunk) stages UK k-steps of weight fragments per batch through LDS with perfectly
// coalesced 16-byte copies, so each fragment leaves L2 once per workgroup instead of once per wave,
// and reads them back with conflict-free lane-linear ds_read_b128.  Activation fragments go straight
// from HBM/L2 to registers (each wave owns its rows), one batch ahead of the MFMAs.
// EPI_SILU   : y = silu(acc+bias)                              (expand conv)
// EPI_LINEAR : y = acc+bias (+ residual)                       (project conv), optional SE gate on X
// EPI_GAP    : out[patch][n] = mean over the patch's HW rows of silu(acc+bias)   (head conv + avgpool)
// ---------------------------------------------------------------------------------------------
template <int MT, int NT, int EPI, bool GATE, bool RES, int UK, bool DG>
__global__ __launch_bounds__(256) void pw_gemm_kernel(const _Float16* __restrict__ X, int M, int K,
                                                      const _Float16* __restrict__ Wp, int KS32,
                                                      const float* __restrict__ bias,  // natural channel order, zero padded
                                                      _Float16* __restrict__ Y, int N,
                                                      const float* __restrict__ gate,  // [patch][K] fp32
                                                      int HW,
                                                      const _Float16* __restrict__ res,
                                                      float* __restrict__ gap_out, float inv_hw)
{
    constexpr int NFRAG = UK * NT;             // weight fragments per batch
    constexpr int NPASS = (NFRAG + 3) / 4;     // 4 waves copy one fragment each per pass
    __shared__ __attribute__((aligned(16))) _Float16 wlds[NFRAG * 512];
    __shared__ float red[EPI == EPI_GAP ? 4 * 16 * NT : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int chunk = blockIdx.y;
    const _Float16* wsrc = Wp + (size_t)chunk * KS32 * NT * 512 + lane * 8;
    int row[MT];
    bool rok[MT];
    int gpatch[MT];
    if (EPI == EPI_GAP) {
        // one workgroup = one patch; rows beyond HW are masked
        const int ml = wave * 16 + m;
        row[0] = blockIdx.x * HW + ml;
        rok[0] = ml < HW;
        gpatch[0] = blockIdx.x;
    } else {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            row[i] = (blockIdx.x * 4 + wave) * (16 * MT) + i * 16 + m;
            rok[i] = row[i] < M;
            gpatch[i] = GATE ? (rok[i] ? row[i] / HW : 0) : 0;
        }
    }
    // bias is the accumulator's initial value: lane (m,q) owns channels cbase + 4t + j
    const int cbase = chunk * 16 * NT + q * 4 * NT;
    f4 acc[MT][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f4 bv = *reinterpret_cast<const f4*>(bias + cbase + 4 * t);
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i][t] = bv;
    }

    const int nbatch = (KS32 + UK - 1) / UK;
    uint4 wst[NPASS];   // weight staging registers (global -> regs -> LDS)
    h8 xf[UK][MT];      // activation fragments of the current batch
    auto load_w = [&](int bt) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int f = ps * 4 + wave;                 // fragment index inside the batch: u*NT + t
            const int ks = bt * UK + f / NT;
            uint4 v = {0u, 0u, 0u, 0u};
            if (f < NFRAG && ks < KS32)
                v = *reinterpret_cast<const uint4*>(wsrc + ((size_t)(bt * UK) * NT + f) * 512);
            wst[ps] = v;
        }
    };
    // DG (defer gate; used by the small-M 7x7 layers, which are latency- not occupancy-bound): activation
    // fragments and their fp32 squeeze-excite gates are only LOADED in load_x; the multiply happens in
    // apply_gate() right before the MFMAs that consume them, so the loads of batch bt+1 really overlap the
    // MFMAs of batch bt.  !DG multiplies at load time (waits for the data, but keeps 8 fewer VGPRs per
    // fragment alive, which is what the large-M layers want).
    constexpr int GN = (GATE && DG) ? 2 : 1;
    auto load_x = [&](int bt, h8 (&dst)[UK][MT], f4 (&g)[UK][MT][GN]) {
#pragma unroll
        for (int u = 0; u < UK; ++u) {
            const int k = (bt * UK + u) * 32 + q * 8;
            const bool kok = k < K;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (kok && rok[i]) v = *reinterpret_cast<const h8*>(X + (size_t)row[i] * K + k);
                if (GATE) {
                    f4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0;
                    if (kok && rok[i]) {
                        g0 = *reinterpret_cast<const f4*>(gate + (size_t)gpatch[i] * K + k);
                        g1 = *reinterpret_cast<const f4*>(gate + (size_t)gpatch[i] * K + k + 4);
                    }
                    if (DG) {
                        g[u][i][0] = g0;
                        g[u][i][GN - 1] = g1;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[j] = (_Float16)((float)v[j] * g0[j]);
                            v[4 + j] = (_Float16)((float)v[4 + j] * g1[j]);
                        }
                    }
                }
                dst[u][i] = v;
            }
        }
    };
    auto apply_gate = [&](h8 (&x)[UK][MT], f4 (&g)[UK][MT][GN]) {
        if (!(GATE && DG)) return;
#pragma unroll
        for (int u = 0; u < UK; ++u)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x[u][i][j] = (_Float16)((float)x[u][i][j] * g[u][i][0][j]);
                    x[u][i][4 + j] = (_Float16)((float)x[u][i][4 + j] * g[u][i][GN - 1][j]);
                }
    };
    f4 gf[UK][MT][GN];
    load_w(0);
    load_x(0, xf, gf);
    for (int bt = 0; bt < nbatch; ++bt) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int f = ps * 4 + wave;
            if (f < NFRAG) *reinterpret_cast<uint4*>(wlds + f * 512 + lane * 8) = wst[ps];
        }
        apply_gate(xf, gf);
        __syncthreads();
        h8 xn[UK][MT];
        f4 gn[UK][MT][GN];
        const bool more = bt + 1 < nbatch;
        if (more) {  // next batch's global loads fly during this batch's MFMAs
            load_w(bt + 1);
            load_x(bt + 1, xn, gn);
        }
#pragma unroll
        for (int u = 0; u < UK; ++u) {
            if (bt * UK + u < KS32) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const h8 wf = *reinterpret_cast<const h8*>(wlds + (u * NT + t) * 512 + lane * 8);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[u][i], acc[i][t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (more) {
#pragma unroll
            for (int u = 0; u < UK; ++u)
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    xf[u][i] = xn[u][i];
#pragma unroll
                    for (int e = 0; e < GN; ++e) gf[u][i][e] = gn[u][i][e];
                }
        }
    }
    // epilogue: lane (m,q) holds channels cbase + 4t + j of pixel row[i]; acc already includes the bias
    if (EPI == EPI_GAP) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = row16_sum(rok[0] ? silu_scaled(acc[0][t][j]) : 0.0f);
                if (m == 0) red[wave * 16 * NT + q * 4 * NT + 4 * t + j] = v;
            }
        __syncthreads();
        if (tid < 16 * NT) {
            const float s = ((red[tid] + red[16 * NT + tid]) + (red[32 * NT + tid] + red[48 * NT + tid])) * inv_hw;
            const int c = chunk * 16 * NT + tid;
            if (c < N) gap_out[(size_t)blockIdx.x * N + c] = s;
        }
        return;
    }
    // skip-connection operands: ALL of them requested before the first is used (unconditional, clamped addresses) -- inside the
    // row / channel conditions below every load was followed by its own s_waitcnt vmcnt(0): MT x NT exposed round trips per workgroup
    h4 rres[MT][NT];
    if (RES) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int c = cbase + 4 * t;
                rres[i][t] = *reinterpret_cast<const h4*>(res + (size_t)(rok[i] ? row[i] : 0) * N + (c < N ? c : 0));
            }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        if (!rok[i]) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int c = cbase + 4 * t;
            if (c < N) {  // N is a multiple of 4
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[i][t][j];
                if (EPI == EPI_SILU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = silu_scaled(v[j]);
                }
                if (RES) {
                    const h4 r = rres[i][t];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
                }
                h4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (_Float16)v[j];
                *reinterpret_cast<h4*>(Y + (size_t)row[i] * N + c) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// pw_gemm_fp8_kernel: SE-scale + project conv (+ skip) on fp8 MFMA operands -- BASELINE.json configs[4] ("EfficientNet-B4, fp8
// weights/activations on CDNA4 fp8 MFMA"; not in the reference).  v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands and unit
// block scales: K = 128 per instruction at twice the fp16 rate (tools/ubench/mfma_f8.hip: layout and cycles).
//   weights      e4m3, one fp32 scale per output channel (amax / 448), quantised and packed in fragment order on the host
//   activations  the fp16 depthwise output times its fp32 squeeze-excite gate, one fp32 scale per pixel row (amax / 447 of the
//                gated row), quantised here: pass 1 reads the row for its maximum, pass 2 reads it again (L2 / Infinity Cache),
//                scales, converts (v_cvt_pk_fp8_f32) and feeds the MFMAs
//   epilogue     y = acc * row scale * channel scale + bias (+ skip) -> fp16
// Operands are swapped as in pw_gemm_kernel (A = weights, B = pixels): a lane ends up with 4 consecutive channels of one pixel.
// One workgroup = 4 waves x 16 pixel rows x NT output fragments.
// ---------------------------------------------------------------------------------------------
typedef int v8i __attribute__((ext_vector_type(8)));
template <int NT, bool RES>
__global__ __launch_bounds__(256) void pw_gemm_fp8_kernel(Fp8GemmArgs a)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int row = (blockIdx.x * 4 + wave) * 16 + m;
    const bool rok = row < a.M;
    const int rowc = rok ? row : a.M - 1;
    const _Float16* xr = a.X + (size_t)rowc * a.K;
    const float* gr = a.gate + (size_t)(rowc / a.HW) * a.K;
    const int f0 = blockIdx.y * NT;
    // one k-step of this lane: 32 channels from 128 ks + 32 q, as x * gate in fp32 (zeros past K)
    auto gated = [&](int ks, float (&v)[32]) {
        const int k0 = 128 * ks + 32 * q;
#pragma unroll
        for (int c8 = 0; c8 < 4; ++c8) {
            const int k = k0 + 8 * c8;
            h8 x = {0, 0, 0, 0, 0, 0, 0, 0};
            f4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0;
            if (k < a.K) {
                x = *reinterpret_cast<const h8*>(xr + k);
                g0 = *reinterpret_cast<const f4*>(gr + k);
                g1 = *reinterpret_cast<const f4*>(gr + k + 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[8 * c8 + j] = (float)x[j] * g0[j]; v[8 * c8 + 4 + j] = (float)x[4 + j] * g1[j]; }
        }
    };
    // ---- pass 1: the row's largest magnitude ----
    float mx = 0.f;
    for (int ks = 0; ks < a.KS128; ++ks) {
        float v[32];
        gated(ks, v);
#pragma unroll
        for (int e = 0; e < 32; e += 2) mx = __builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fabsf(v[e]), __builtin_fabsf(v[e + 1])));
    }
    mx = __builtin_fmaxf(mx, __shfl_xor(mx, 16));
    mx = __builtin_fmaxf(mx, __shfl_xor(mx, 32));
    // 447 (not 448): the scaled maximum stays below e4m3's largest finite value after fp32 rounding
    const float inv = mx > 0.f ? 447.0f / mx : 0.f;
    const float sx = mx > 0.f ? mx * (1.0f / 447.0f) : 0.f;
    // ---- pass 2: quantise and multiply ----
    f4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
    const uint8_t* wl = a.W8 + (size_t)lane * 32;
    for (int ks = 0; ks < a.KS128; ++ks) {
        float v[32];
        gated(ks, v);
        v8i bq;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            int w = 0;
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * d] * inv, v[4 * d + 1] * inv, w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * d + 2] * inv, v[4 * d + 3] * inv, w, true);
            bq[d] = w;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const uint8_t* wp = wl + ((size_t)(f0 + t) * a.KS128 + ks) * 2048;
            union { uint4 u[2]; v8i v; } aw;
            aw.u[0] = *reinterpret_cast<const uint4*>(wp);
            aw.u[1] = *reinterpret_cast<const uint4*>(wp + 16);
            acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(aw.v, bq, acc[t], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
    }
    // ---- epilogue: lane (m, q) holds channels 16 (f0 + t) + 4 q + j of pixel row m ----
    if (!rok) return;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c = 16 * (f0 + t) + 4 * q;
        if (c >= a.N) continue;   // N is a multiple of 4
        const f4 sw = *reinterpret_cast<const f4*>(a.sw + c), bv = *reinterpret_cast<const f4*>(a.bias + c);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[t][j] * sx * sw[j] + bv[j];
        if (RES) {
            const h4 r = *reinterpret_cast<const h4*>(a.res + (size_t)row * a.N + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
        }
        h4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (_Float16)v[j];
        *reinterpret_cast<h4*>(a.Y + (size_t)row * a.N + c) = o;
    }
}

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

// ---------------------------------------------------------------------------------------------
// thin_proj_kernel: SE-scale + project conv (+ skip) for layers with K <= 64 and N <= 32 on big images (B4's expand-less
// blocks 0 and 1: 48 -> 24 and 24 -> 24 at 112x112).  pw_gemm_kernel gives such a layer one k-step of work per workgroup
// between two barriers (0.6-1.1 TB/s measured); here a workgroup owns a run of one patch's pixel fragments, keeps the
// 2*KSTEPS weight fragments and the patch's gate in registers, and each wave streams fragments with the next one's loads in
// flight.  Same weight packing (pack_pw, nt = 2) as pw_gemm_kernel<.,2,EPI_LINEAR,GATE,RES>.
// ---------------------------------------------------------------------------------------------
template <int KSTEPS, bool RES>
__global__ __launch_bounds__(256) void thin_proj_kernel(const _Float16* __restrict__ X, int K, const _Float16* __restrict__ Wp,
                                                        const float* __restrict__ bias, _Float16* __restrict__ Y, int N,
                                                        const float* __restrict__ gate, int HW, int frags_per_wg,
                                                        const _Float16* __restrict__ res, int plane_rows)
{   // plane_rows > 0: X is [K / 32 planes][plane_rows][32] (mb1_kernel<true>'s output), k-step ks reads plane ks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int b = blockIdx.y;
    const int nfrag = HW >> 4;   // HW is a multiple of 16
    const int f0 = blockIdx.x * frags_per_wg;
    const int f1 = f0 + frags_per_wg < nfrag ? f0 + frags_per_wg : nfrag;
    // The squeeze-excite gate goes into the WEIGHT fragments, once per workgroup (the A operand's k index of lane quarter q is
    // 8q .. 8q+7: the patch's gate values for those input channels): no gate registers (2 x 4 per k-step) and no per-fragment
    // scaling of the activations (8 conversions + products per k-step and fragment) -- what kept five-k-step layers (block 2's
    // project) slower here than on pw_gemm_kernel.  w * g rounded to fp16 instead of x * g: the same size of rounding error.
    // Round 3: every request of the prologue goes out before anything is consumed -- the wave's first pixel fragment, then all gate
    // values, then all weight fragments (it used to be gate -> wait -> weights -> wait per k-step, KSTEPS exposed round trips for a wave
    // that streams three to seven fragments), all unconditional: columns past K re-read the last eight channels / gate values, finite
    // numbers that meet zero weight rows (K is zero padded to whole k-steps in Wp).
    const int cbase = q * 8;   // lane (m,q) owns channels 8q .. 8q+7 (4t + j) of pixel row m of the fragment
    const int cres = cbase < N ? cbase : 0;
    const _Float16* xb = X + (size_t)b * HW * K;
    auto load = [&](int f, h8 (&dst)[KSTEPS], h8& r) {
        const size_t row = (size_t)f * 16 + m;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int k0 = ks * 32 + q * 8, k = k0 < K ? k0 : K - 8;
            dst[ks] = plane_rows ? *reinterpret_cast<const h8*>(X + (((size_t)(k >> 5) * plane_rows + (size_t)b * HW + row) * 32 + (k & 31)))
                                 : *reinterpret_cast<const h8*>(xb + row * K + k);
        }
        if (RES) r = *reinterpret_cast<const h8*>(res + ((size_t)b * HW + row) * N + cres);
    };
    h8 xc[KSTEPS], xn[KSTEPS], rc = {0, 0, 0, 0, 0, 0, 0, 0}, rn = rc;
    int f = f0 + wave;
    load(f < f1 ? f : f1 - 1, xc, rc);
    h8 wf[KSTEPS][2];
    {
        f4 g0[KSTEPS], g1[KSTEPS];
        uint4 wr[KSTEPS][2];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int k0 = ks * 32 + q * 8, k = k0 < K ? k0 : K - 8;
            g0[ks] = *reinterpret_cast<const f4*>(gate + (size_t)b * K + k);
            g1[ks] = *reinterpret_cast<const f4*>(gate + (size_t)b * K + k + 4);
        }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t) wr[ks][t] = *reinterpret_cast<const uint4*>(Wp + ((size_t)(ks * 2 + t) * 64 + lane) * 8);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint4 gw = gate_h8(wr[ks][t], g0[ks], g1[ks]);
                wf[ks][t] = *reinterpret_cast<const h8*>(&gw);
            }
    }
    const f4 bv0 = *reinterpret_cast<const f4*>(bias + cbase), bv1 = *reinterpret_cast<const f4*>(bias + cbase + 4);
    for (; f < f1; f += 4) {
        const bool more = f + 4 < f1;
        if (more) load(f + 4, xn, rn);
        f4 a0 = bv0, a1 = bv1;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][0], xc[ks], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][1], xc[ks], a1, 0, 0, 0);
        }
        if (cbase < N) {   // N is a multiple of 8 here
            h8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (_Float16)(a0[j] + (RES ? (float)rc[j] : 0.f));
                o[4 + j] = (_Float16)(a1[j] + (RES ? (float)rc[4 + j] : 0.f));
            }
            *reinterpret_cast<h8*>(Y + ((size_t)b * HW + (size_t)f * 16 + m) * N + cbase) = o;
        }
        if (more) {
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) xc[ks] = xn[ks];
            rc = rn;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Depthwise KSxKS convolution, stride ST, TF-same padding, + bias (BN folded) + SiLU, fp16 out,
// plus per-(patch, channel) partial sums of the fp32 SiLU outputs for squeeze-excite.
// Thread = 8 channels x TW consecutive output pixels of one row; channel-group index is the fastest
// thread index so neighbouring lanes read neighbouring 16-byte vectors (coalesced NHWC).
// blockDim.x = CG*S (CG = channel groups of 8 per workgroup = C/8/gridDim.z, S strips per pass).  Partial sums are reduced through
// LDS in a fixed order and written to pool_part[patch][blockIdx.x][C] (deterministic, no atomics).
// ---------------------------------------------------------------------------------------------
template <int KS, int ST, int TW>
__global__ __launch_bounds__(256) void dwconv_kernel(const _Float16* __restrict__ in,  // [B][H][W][C]
                                                     const float* __restrict__ wt,     // [KS*KS][C]
                                                     const float* __restrict__ bias,   // [C]
                                                     _Float16* __restrict__ out,       // [B][Ho][Wo][C]
                                                     float* __restrict__ pool_part,    // [B][gridDim.x][C]
                                                     int H, int W, int C, int Ho, int Wo, int pad_t, int pad_l,
                                                     int CG, int S, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float red[];  // [S][8 CG]
    const int tid = threadIdx.x;
    const int cg = tid % CG, s = tid / CG;
    const int b = blockIdx.y;
    const int strips_per_row = Wo / TW;
    const int nstrips = Ho * strips_per_row;
    const int coff = blockIdx.z * CG * 8;   // layers wider than 2048 channels split their channel groups over z
    const int CL = CG * 8;
    const int c0 = coff + cg * 8;
    const _Float16* inb = in + (size_t)b * H * W * C + c0;
    _Float16* outb = out + (size_t)b * Ho * Wo * C + c0;
    float bs[8];
    {
        const f4 b0 = *reinterpret_cast<const f4*>(bias + c0);
        const f4 b1 = *reinterpret_cast<const f4*>(bias + c0 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { bs[j] = b0[j]; bs[4 + j] = b1[j]; }
    }
    float pooled[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) pooled[j] = 0.f;
    constexpr int NX = (TW - 1) * ST + KS;  // input columns a strip touches
    for (int it = 0; it < iters; ++it) {
        const int strip = (blockIdx.x * iters + it) * S + s;
        if (strip < nstrips) {
            const int oy = strip / strips_per_row;
            const int ox0 = (strip - oy * strips_per_row) * TW;
            float acc[TW][8];
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[t][j] = bs[j];
#pragma unroll 1
            for (int ky = 0; ky < KS; ++ky) {
                const int iy = oy * ST - pad_t + ky;
                if (iy < 0 || iy >= H) continue;
                const _Float16* rowp = inb + (size_t)iy * W * C;
                float wk[KS][8];
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const f4 w0 = *reinterpret_cast<const f4*>(wt + (size_t)(ky * KS + kx) * C + c0);
                    const f4 w1 = *reinterpret_cast<const f4*>(wt + (size_t)(ky * KS + kx) * C + c0 + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { wk[kx][j] = w0[j]; wk[kx][4 + j] = w1[j]; }
                }
#pragma unroll
                for (int xr = 0; xr < NX; ++xr) {
                    const int ix = ox0 * ST - pad_l + xr;
                    h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (ix >= 0 && ix < W) v = *reinterpret_cast<const h8*>(rowp + (size_t)ix * C);
                    float vf[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) vf[j] = (float)v[j];
#pragma unroll
                    for (int t = 0; t < TW; ++t) {
                        const int kx = xr - t * ST;
                        if (kx >= 0 && kx < KS) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[t][j] = __builtin_fmaf(vf[j], wk[kx][j], acc[t][j]);
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < TW; ++t) {
                h8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float y = silu_scaled(acc[t][j]);
                    pooled[j] += y;
                    o[j] = (_Float16)y;
                }
                *reinterpret_cast<h8*>(outb + ((size_t)oy * Wo + ox0 + t) * C) = o;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[s * CL + cg * 8 + j] = pooled[j];
    __syncthreads();
    for (int c = tid; c < CL; c += blockDim.x) {
        float sum = 0.f;
        for (int ss = 0; ss < S; ++ss) sum += red[ss * CL + c];
        pool_part[((size_t)b * gridDim.x + blockIdx.x) * C + coff + c] = sum;
    }
}

// ---------------------------------------------------------------------------------------------
// Squeeze-excite gate in two launches:
//   pooled[c] = inv_hw * sum_p pool_part[b][p][c]
//   r[j]      = silu(b_r[j] + sum_c W_r[j][c] pooled[c])        j < Cs   (wave-reduced dot products)
//   gate[c]   = sigmoid(b_e[c] + sum_j W_e[c][j] r[j])      (W_e stored transposed, [Cs][C])
// ---------------------------------------------------------------------------------------------
// Both squeeze-excite FCs are batch GEMMs over the patches, run on the exact-f32 MFMA
// (v_mfma_f32_16x16x4_f32) so that a weight row is fetched once per 16 patches, not once per patch.
//   XMODE 1: X[row][k] = sum_{p<nslab} Xs[(row*nslab + p)*K + k]        (pool partial sums of a patch)
//   XMODE 2: X[row][k] = silu(xbias[k] + sum_{z<nslab} Xs[(z*M + row)*K + k])   (split-K partials of FC1)
//   ACT 0: Y slab z = partial products over this z's K range (no bias)   ACT 2: sigmoid(acc + bias)
// Lane (i=l&15, q=l>>4) feeds 4 consecutive k per 16-k group (one per MFMA step); outputs land as
// lane (i,q) -> columns n0 + 16t + 4q + j of row i (operands swapped, as in the other GEMMs).
template <int XMODE, int ACT>
__global__ __launch_bounds__(256) void se_gemm_f32_kernel(const float* __restrict__ Xs, int nslab, int M, int K,
                                                          const float* __restrict__ xbias,
                                                          const float* __restrict__ W,   // [N][K]
                                                          const float* __restrict__ bias, float* __restrict__ Y,
                                                          int N, int kz)
{
    constexpr int NT = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int row = (blockIdx.x * 4 + wave) * 16 + i;
    const bool rok = row < M;
    const int n0 = blockIdx.y * 16 * NT;
    const int kbeg = blockIdx.z * kz;
    const int kend = (kbeg + kz) < K ? (kbeg + kz) : K;
    f4 acc[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t][0] = acc[t][1] = (f4){0.f, 0.f, 0.f, 0.f};
    // U k-groups per batch: every load of a batch is issued before its MFMAs (these GEMMs are pure
    // latency chains: tiny, with all operands a fresh L2/HBM round trip away)
    constexpr int U = 3;
    for (int k0 = kbeg; k0 < kend; k0 += 16 * U) {
        f4 xv[U], wv[U][NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 16 * u + 4 * q;
            const bool kok = k < kend;
            f4 x = {0.f, 0.f, 0.f, 0.f};
            if (rok && kok) {
                if (XMODE == 1) {
                    const float* xp = Xs + (size_t)row * nslab * K + k;
                    f4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s0, s3 = s0;
                    int p = 0;
                    for (; p + 3 < nslab; p += 4) {
                        s0 += *reinterpret_cast<const f4*>(xp + (size_t)p * K);
                        s1 += *reinterpret_cast<const f4*>(xp + (size_t)(p + 1) * K);
                        s2 += *reinterpret_cast<const f4*>(xp + (size_t)(p + 2) * K);
                        s3 += *reinterpret_cast<const f4*>(xp + (size_t)(p + 3) * K);
                    }
                    for (; p < nslab; ++p) s0 += *reinterpret_cast<const f4*>(xp + (size_t)p * K);
                    x = (s0 + s1) + (s2 + s3);
                } else {
                    f4 sum = *reinterpret_cast<const f4*>(xbias + k);
#pragma unroll 8
                    for (int z = 0; z < nslab; ++z)
                        sum += *reinterpret_cast<const f4*>(Xs + ((size_t)z * M + row) * K + k);
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j] = silu_f(sum[j]);
                }
            }
            xv[u] = x;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n = n0 + t * 16 + i;
                f4 w = {0.f, 0.f, 0.f, 0.f};
                if (n < N && kok) w = *reinterpret_cast<const f4*>(W + (size_t)n * K + k);
                wv[u][t] = w;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc[t][s & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][t][s], xv[u][s], acc[t][s & 1], 0, 0, 0);
    }
    if (!rok) return;
    float* yo = Y + (ACT == 0 ? (size_t)blockIdx.z * M * N : 0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + t * 16 + 4 * q + j;
            if (n < N) {
                float v = acc[t][0][j] + acc[t][1][j];
                if (ACT == 2) v = sigmoid_f(v + bias[n]);
                yo[(size_t)row * N + n] = v;
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Squeeze-excite in ONE launch: both FCs for 16 patches per workgroup of 16 waves.  The two tiny
// GEMMs are pure latency chains, so the workgroup is wide instead of deep:
//   FC1  r[16][Cs4] = silu(br + P[16][C] . Wr^T): the 16 waves split K (each sums its pool-partial slabs on
//        the fly and issues all its loads before its exact-f32 MFMAs), partials meet in LDS;
//   FC2  gate[16][C] = sigmoid(be + r . We^T): the 16 waves split the C/16 output fragments, r comes from LDS.
// ---------------------------------------------------------------------------------------------
#define SE_MAXG 5   // k-groups (16 k each) a wave may own in FC1: C <= 16 waves * 5 * 16 = 1280
#define SE_MAXT 1   // output fragments a wave may own in FC2 (after the gridDim.y split)
// Weights are fp32 (fp16 storage was tried: the gate error it causes is coherent per channel and
// roughly doubled the end-to-end feature error), packed in MFMA fragment order (16 bytes per lane,
// 1 KB per fragment, contiguous -> perfectly coalesced loads):
//   WrP[(g*3 + t)*64 + lane][4] = Wr[16t + i][16g + 4q .. +4]      (zero for j >= Cs, carries 1/(HW log2e))
//   WeP[(T*3 + g)*64 + lane][4] = We[16T + i][16g + 4q .. +4]      (zero for k >= Cs)
// gridDim = (ceil(M/16), NSPLIT): every y-slice recomputes FC1 (cheap) and owns 1/NSPLIT of FC2's outputs,
// so the weight stream of one patch group is spread over NSPLIT compute units.
template <int MAXG>
__global__ __launch_bounds__(1024) void se_fused_kernel(const float* __restrict__ pool_part, int nslab, int M, int C,
                                                        int Cs4, const float* __restrict__ WrP,
                                                        const float* __restrict__ br,  // [48] zero padded
                                                        const float* __restrict__ WeP,
                                                        const float* __restrict__ be, float* __restrict__ gate)
{
    __shared__ __attribute__((aligned(16))) float part[16][16][48];  // [wave][row][j]
    __shared__ __attribute__((aligned(16))) float rs[16][48];        // [row][j]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * 16 + i;
    const bool rok = row < M;
    const int NG = C >> 4;            // k-groups of FC1 == output fragments of FC2
    const int per_y = (NG + gridDim.y - 1) / gridDim.y;
    const int t_lo = blockIdx.y * per_y;
    const int t_hi = (t_lo + per_y) < NG ? (t_lo + per_y) : NG;
    // FC2's weight fragments and bias do not depend on FC1: issue their loads first so the whole kernel
    // is one memory round trip (pool partials, Wr, We, be all in flight together)
    f4 we[SE_MAXT][3];
    f4 bev[SE_MAXT];
#pragma unroll
    for (int u = 0; u < SE_MAXT; ++u) {
        const int T = t_lo + wave + 16 * u;
        bev[u] = (f4){0.f, 0.f, 0.f, 0.f};
        if (T < t_hi) bev[u] = *reinterpret_cast<const f4*>(be + T * 16 + 4 * q);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            f4 w = {0.f, 0.f, 0.f, 0.f};
            if (T < t_hi) w = *reinterpret_cast<const f4*>(WeP + ((size_t)(T * 3 + g) * 64 + lane) * 4);
            we[u][g] = w;
        }
    }
    // ---- FC1: this wave owns k-groups g = wave, wave+16, ... ----
    {
        f4 xv[MAXG];
        f4 wv[MAXG][3];
#pragma unroll
        for (int u = 0; u < MAXG; ++u) {
            const int g = wave + 16 * u;
            const bool gok = g < NG;
            const int k = g * 16 + 4 * q;
            f4 x = {0.f, 0.f, 0.f, 0.f};
            if (rok && gok) {
                const float* xp = pool_part + (size_t)row * nslab * C + k;
                f4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
                int p = 0;
                if (MAXG <= 2) {   // early blocks: many slabs, few channels -> 16 independent loads per round trip
                    for (; p + 15 < nslab; p += 16) {
                        f4 v[16];
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = *reinterpret_cast<const f4*>(xp + (size_t)(p + e) * C);
#pragma unroll
                        for (int e = 0; e < 16; e += 4) {
                            s0 += v[e];
                            s1 += v[e + 1];
                            s2 += v[e + 2];
                            s3 += v[e + 3];
                        }
                    }
                }
                for (; p + 3 < nslab; p += 4) {
                    s0 += *reinterpret_cast<const f4*>(xp + (size_t)p * C);
                    s1 += *reinterpret_cast<const f4*>(xp + (size_t)(p + 1) * C);
                    s2 += *reinterpret_cast<const f4*>(xp + (size_t)(p + 2) * C);
                    s3 += *reinterpret_cast<const f4*>(xp + (size_t)(p + 3) * C);
                }
                for (; p < nslab; ++p) s0 += *reinterpret_cast<const f4*>(xp + (size_t)p * C);
                x = (s0 + s1) + (s2 + s3);
            }
            xv[u] = x;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                f4 w = {0.f, 0.f, 0.f, 0.f};
                if (gok) w = *reinterpret_cast<const f4*>(WrP + ((size_t)(g * 3 + t) * 64 + lane) * 4);
                wv[u][t] = w;
            }
        }
        f4 acc[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < MAXG; ++u)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][t][s], xv[u][s], acc[t], 0, 0, 0);
        // lane (i,q) holds outputs j = 16t + 4q + jj of row i
#pragma unroll
        for (int t = 0; t < 3; ++t) *reinterpret_cast<f4*>(&part[wave][i][16 * t + 4 * q]) = acc[t];
    }
    __syncthreads();
    if (tid < 16 * 48) {
        const int r = tid / 48, j = tid - r * 48;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) s += part[w][r][j];
        rs[r][j] = (j < Cs4) ? silu_f(s + br[j]) : 0.f;
    }
    __syncthreads();
    // ---- FC2: output fragments T = y*per_y + wave + 16u ----
    {
        f4 xr[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) xr[g] = *reinterpret_cast<const f4*>(&rs[i][g * 16 + 4 * q]);
#pragma unroll
        for (int u = 0; u < SE_MAXT; ++u) {
            const int T = t_lo + wave + 16 * u;
            if (T >= t_hi) break;
            f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(we[u][g][s], xr[g][s], acc, 0, 0, 0);
            if (rok) {
                const int n = T * 16 + 4 * q;
                const f4 bv = bev[u];
                f4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = sigmoid_f(acc[j] + bv[j]);
                *reinterpret_cast<f4*>(gate + (size_t)row * C + n) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// se_small_kernel: squeeze-excite for the early blocks (C <= 256 channels, Cs <= 16 squeeze units), one patch per
// 256-thread workgroup.  The work is tiny (b0: 32x8, b1: 96x4, b2: 144x6 MACs per FC), what matters is that the
// launch gets onto the chip at once while the other lane's big kernels fill it: a 16-wave / 52 KB workgroup of
// se_fused_kernel has to wait for a whole compute unit to drain, a 4-wave / 1 KB one fits anywhere.
// fp32 throughout, natural weight layouts, fixed summation order (slab sums in 4 chains, wave butterfly for FC1).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void se_small_kernel(const float* __restrict__ pool_part, int nparts, int C, int Cs,
                                                       const float* __restrict__ wr,   // [Cs][C], carries 1/(HW log2e)
                                                       const float* __restrict__ br,   // [Cs]
                                                       const float* __restrict__ we,   // [C][Cs]
                                                       const float* __restrict__ be,   // [C]
                                                       float* __restrict__ gate)       // [B][C]
{
    // (Round 3 re-tried both FCs' operands requested at the top, this time unconditionally from clamped addresses: 9.7 / 7.5 / 6.8 us against
    // 6.9 / 6.7 / 7.3 -- the 36 extra requests per thread in front of block 0's 49 pool partials cost more than the two round trips they hide.)
    __shared__ float pooled[256];
    __shared__ float rs[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    if (tid < C) {
        const float* pp = pool_part + (size_t)b * nparts * C + tid;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int p = 0;
#pragma unroll 4   // 16 partials in flight per round trip (block 0 has 49 tiles: twelve dependent round trips otherwise); same sums
        for (; p + 3 < nparts; p += 4) {
            s0 += pp[(size_t)p * C];
            s1 += pp[(size_t)(p + 1) * C];
            s2 += pp[(size_t)(p + 2) * C];
            s3 += pp[(size_t)(p + 3) * C];
        }
        for (; p < nparts; ++p) s0 += pp[(size_t)p * C];
        pooled[tid] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    for (int j = wave; j < Cs; j += 4) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s = __builtin_fmaf(pooled[c], wr[(size_t)j * C + c], s);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) rs[j] = silu_f(s + br[j]);
    }
    __syncthreads();
    if (tid < C) {
        float acc = be[tid];
        for (int j = 0; j < Cs; ++j) acc = __builtin_fmaf(rs[j], we[(size_t)tid * Cs + j], acc);
        gate[(size_t)b * C + tid] = sigmoid_f(acc);
    }
}

// se_wide_kernel: the same computation without the size limits (any C, any Cs; pooled vectors and squeeze units in
// dynamic LDS) -- the squeeze-excite of the generic per-layer schedule (EfficientNet-B4: C <= 2688, Cs <= 112).
// One workgroup takes PB consecutive patches so that a weight element fetched from L2 serves PB patches (one workgroup per
// patch re-read up to 1.2 MB per FC: 19 % of B4's time); each patch's own arithmetic sequence is that of PB = 1, so results
// do not depend on how patches are grouped.
template <int PB>
__global__ __launch_bounds__(1024) void se_wide_kernel(const float* __restrict__ pool_part, int nparts, int nB, int C, int Cs,
                                                      const float* __restrict__ wr,   // [Cs][C], carries 1/(HW log2e)
                                                      const float* __restrict__ br,   // [Cs]
                                                      const float* __restrict__ we,   // [Cs][C] (transposed: lanes read neighbours)
                                                      const float* __restrict__ be,   // [C]
                                                      float* __restrict__ gate)       // [B][C]
{
    extern __shared__ float se_sm[];
    float* pooled = se_sm;            // [PB][C]
    float* rs = se_sm + PB * C;       // [PB][Cs]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * PB;
    const int nb = (nB - b0) < PB ? (nB - b0) : PB;
    for (int pb = 0; pb < nb; ++pb)
        for (int c = tid; c < C; c += 1024) {
            const float* pp = pool_part + (size_t)(b0 + pb) * nparts * C + c;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int p = 0;
            for (; p + 3 < nparts; p += 4) {
                s0 += pp[(size_t)p * C];
                s1 += pp[(size_t)(p + 1) * C];
                s2 += pp[(size_t)(p + 2) * C];
                s3 += pp[(size_t)(p + 3) * C];
            }
            for (; p < nparts; ++p) s0 += pp[(size_t)p * C];
            pooled[pb * C + c] = (s0 + s1) + (s2 + s3);
        }
    for (int pb = nb; pb < PB; ++pb)
        for (int c = tid; c < C; c += 1024) pooled[pb * C + c] = 0.f;
    __syncthreads();
    for (int j = wave; j < Cs; j += 16) {   // 16 waves: the FC1 rows are a latency chain per wave
        float s[PB];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) s[pb] = 0.f;
        // sixteen weights requested per round trip (unconditional, clamped; round 3: the loop was load -> wait -> fma, one exposed L2
        // round trip per 64 channels -- up to 42 per row); the products are summed in the same order as before
        for (int c0 = lane; c0 < C; c0 += 1024) {
            float w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) w[u] = wr[(size_t)j * C + (c0 + 64 * u < C ? c0 + 64 * u : C - 1)];
            // (the pool sums of eight channels x PB patches per LDS round trip: read inside the FMA loop they were one exposed round
            // trip per FMA)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float x[8][PB];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + 64 * (8 * h + u);
                    const int cl = c < C ? c : C - 1;
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) x[u][pb] = pooled[pb * C + cl];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + 64 * (8 * h + u);
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) s[pb] = c < C ? __builtin_fmaf(x[u][pb], w[8 * h + u], s[pb]) : s[pb];
                }
            }
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) s[pb] += __shfl_xor(s[pb], o);
            if (lane == 0) rs[pb * Cs + j] = silu_f(s[pb] + br[j]);
        }
    }
    __syncthreads();
    // excite FC: a thread's (up to three) channels advance together, sixteen squeeze units per round trip: 48 requests in flight
    // (round 3: one request, one wait, one fma -- up to 3 x 112 exposed L2 round trips per thread); same summation order per channel
    {
        constexpr int NC = 3;   // C <= 3072
        float acc[NC][PB];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int c = tid + 1024 * k < C ? tid + 1024 * k : C - 1;
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) acc[k][pb] = be[c];
        }
        for (int j0 = 0; j0 < Cs; j0 += 16) {
            float w[NC][16];
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int c = tid + 1024 * k < C ? tid + 1024 * k : C - 1;
#pragma unroll
                for (int u = 0; u < 16; ++u) w[k][u] = we[(size_t)(j0 + u < Cs ? j0 + u : Cs - 1) * C + c];
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float r[8][PB];   // (eight squeeze units x PB patches per LDS round trip)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + 8 * h + u < Cs ? j0 + 8 * h + u : Cs - 1;
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) r[u][pb] = rs[pb * Cs + j];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                        for (int k = 0; k < NC; ++k)
                            acc[k][pb] = j0 + 8 * h + u < Cs ? __builtin_fmaf(r[u][pb], w[k][8 * h + u], acc[k][pb]) : acc[k][pb];
            }
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int c = tid + 1024 * k;
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
                if (c < C && pb < nb) gate[(size_t)(b0 + pb) * C + c] = sigmoid_f(acc[k][pb]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Calibrated MLP head, fp32 end to end (the reference gate is max|dp| <= 1e-6, inference/export.py:31).
// Y[m][n] = act( sum_k X[m][k] W[n][k] + b[n] ) on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32).
// Lane (i=l&15, q=l>>4) loads 4 consecutive k of its row (16 B) and feeds element s at step s, so
// the 16 k of a group are covered by 4 MFMAs with both operands using the same k permutation.
// Workgroup = 4 waves; wave w owns rows [16*(4*bx+w), +16) and NT fragments of 16 output columns.
// ---------------------------------------------------------------------------------------------
template <int NT, bool RELU>
__global__ __launch_bounds__(256) void mlp_gemm_f32_kernel(const float* __restrict__ X, int M, int K,
                                                           const float* __restrict__ W, const float* __restrict__ bias,
                                                           float* __restrict__ Y, int N)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int row = (blockIdx.x * 4 + wave) * 16 + i;
    const bool rok = row < M;
    const int n0 = blockIdx.y * 16 * NT;
    // four independent accumulation chains per fragment (k-step s feeds chain s): shorter chains
    // than one 1280-long fma sequence -> less fp32 drift against the reference's blocked sgemm,
    // and no MFMA dependent-issue stalls.
    f4 acc[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[t][s] = (f4){0.f, 0.f, 0.f, 0.f};
    const int K16 = K & ~15;
    for (int k0 = 0; k0 < K16; k0 += 16) {
        f4 xv = {0.f, 0.f, 0.f, 0.f};
        if (rok) xv = *reinterpret_cast<const f4*>(X + (size_t)row * K + k0 + 4 * q);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = n0 + t * 16 + i;
            f4 wv = {0.f, 0.f, 0.f, 0.f};
            if (n < N) wv = *reinterpret_cast<const f4*>(W + (size_t)n * K + k0 + 4 * q);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc[t][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[s], xv[s], acc[t][s], 0, 0, 0);
        }
    }
    if (K16 < K) {  // K tail: K is padded to a multiple of 4 by mmc_head_create
        f4 xv = {0.f, 0.f, 0.f, 0.f};
        const int k = K16 + 4 * q;
        if (rok && k < K) xv = *reinterpret_cast<const f4*>(X + (size_t)row * K + k);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = n0 + t * 16 + i;
            f4 wv = {0.f, 0.f, 0.f, 0.f};
            if (n < N && k < K) wv = *reinterpret_cast<const f4*>(W + (size_t)n * K + k);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc[t][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[s], xv[s], acc[t][s], 0, 0, 0);
        }
    }
    // swapped operands: lane (i,q) holds outputs n = n0 + 16t + 4q + j of row `row`
    if (!rok) return;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + t * 16 + 4 * q + j;
            if (n < N) {
                float v = ((acc[t][0][j] + acc[t][1][j]) + (acc[t][2][j] + acc[t][3][j])) + bias[n];
                if (RELU) v = fmaxf(v, 0.f);
                Y[(size_t)row * N + n] = v;
            }
        }
}

// One wave per row: softmax -> Platt sigmoid -> row normalise (uniform row when the sum is 0)
// -> sklearn overshoot clip -> argmax (first maximum, like numpy/torch argmax).   head.py:75-89
__global__ __launch_bounds__(256) void calibrate_kernel(const float* __restrict__ logits, int M, int K,
                                                        const float* __restrict__ a, const float* __restrict__ bcal,
                                                        float* __restrict__ proba, int32_t* __restrict__ argmax_out)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* x = logits + (size_t)row * K;
    float mx = -INFINITY;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, x[k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float se = 0.f;
    for (int k = lane; k < K; k += 64) se += expf(x[k] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
    float cs = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float p = expf(x[k] - mx) / se;
        const float c = 1.0f / (1.0f + expf(a[k] * p + bcal[k]));  // sigmoid(-(a p + b))
        proba[(size_t)row * K + k] = c;
        cs += c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cs += __shfl_xor(cs, o);
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int k = lane; k < K; k += 64) {
        float v = (cs != 0.f) ? proba[(size_t)row * K + k] / cs : 1.0f / (float)K;
        if (v > 1.0f && v <= 1.00001f) v = 1.0f;
        proba[(size_t)row * K + k] = v;
        if (v > best) { best = v; besti = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(besti, o);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if (argmax_out && lane == 0) argmax_out[row] = besti;
}

// ---------------------------------------------------------------------------------------------
// crop_patches: reflect-pad + slice as pure index arithmetic on the resident image.
// numpy 'reflect': index i<0 -> -i ; i>=n -> 2(n-1)-i.  One thread = 4 output pixels (12 bytes).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crop_kernel(const uint8_t* __restrict__ image, int H, int W,
                                                   const int32_t* __restrict__ rowcols, uint8_t* __restrict__ out)
{
    const int p = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;  // over 224*56 groups of 4 pixels
    if (idx >= 224 * 56) return;
    const int y = idx / 56, xg = idx - y * 56;
    // device-resident points have not been seen by the host: clamp them into the image so that a bad point can never
    // turn into an out-of-bounds read (the host-points path rejects such points with MMC_ERR_ARG before launching)
    int row = rowcols[2 * p], col = rowcols[2 * p + 1];
    row = row < 0 ? 0 : (row >= H ? H - 1 : row);
    col = col < 0 ? 0 : (col >= W ? W - 1 : col);
    int sy = row - 112 + y;
    sy = sy < 0 ? -sy : sy;
    sy = sy >= H ? 2 * (H - 1) - sy : sy;
    uint32_t w[3] = {0, 0, 0};
    uint8_t* wb = reinterpret_cast<uint8_t*>(w);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int sx = col - 112 + xg * 4 + e;
        sx = sx < 0 ? -sx : sx;
        sx = sx >= W ? 2 * (W - 1) - sx : sx;
        const uint8_t* src = image + ((size_t)sy * W + sx) * 3;
        wb[3 * e + 0] = src[0];
        wb[3 * e + 1] = src[1];
        wb[3 * e + 2] = src[2];
    }
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + ((size_t)p * 224 * 224 + (size_t)y * 224 + xg * 4) * 3);
    dst[0] = w[0];
    dst[1] = w[1];
    dst[2] = w[2];
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


// ---------------------------------------------------------------------------------------------
// Fused MBConv front half: expand 1x1 (+bias+SiLU) -> LDS -> depthwise KSxKS stride ST (+bias+SiLU)
// -> fp16 NHWC to HBM, plus squeeze-excite partial sums.  The 6x-expanded tensor never leaves the CU.
// One workgroup = (patch, output tile TH x TWo, chunk of CC expanded channels).
//   phase 0: the chunk's depthwise taps go to LDS; every wave issues ALL its input-fragment loads
//            (up to NPAIR pairs of 16-position fragments x KSTEPS) so one HBM latency covers the tile.
//   phase 1: the tile's input window (halo included, clipped to the image) is P positions x Cin;
//            weights = MFMA A operand (prefetched one 16-channel fragment ahead), positions = B
//            operand; silu(acc+bias) is written as fp16 into LDS E[position][CC]
//            (row stride CC*2+16 bytes: 16-B aligned rows, spread over banks).
//   phase 2: the depthwise conv reads E with 16-byte LDS reads (8 channels x TW output pixels per
//            thread) and accumulates in fp32 with v_fma_mix_f32; image borders are handled by tap
//            predication (padding is zero in the expanded domain, so skipped taps are exact).
// ---------------------------------------------------------------------------------------------
// CC (channels per chunk) and TWO (output tile width) are template parameters so that every row
// stride, channel-group split and strip decode is constant arithmetic: the kernel is VALU-bound and
// runtime integer multiplies/divides were ~half of its instruction stream.
// PB > 1 (whole-image tiles only: 7x7 layers): one workgroup takes PB consecutive patches, so the chunk's
// weight fragments are streamed once per PB patches and all four waves have MFMA fragments to work on.
// WLDS: the chunk's expand weights (fragment order, Wfrag) are copied to LDS in one burst at kernel start and
// read back lane-linearly per MFMA; otherwise fragments stream from L2 (Wexp rows), one fragment ahead.
// PRE (block 1 only): the kernel's input is block 0's DEPTHWISE output [B][H][W][32]; block 0's squeeze-excite scale and
// project conv (32 -> 16, one MFMA per 16 positions) run on the freshly loaded fragments, so block 0's output tensor and
// its project launch do not exist.  The project result lands as 4 consecutive channels per lane (4q..4q+3); the expand
// weights are packed with the matching K permutation (slot 8q+j <- channel 4q+j, j < 4) so no lane exchange is needed.
template <int KS, int ST, int TW, int KSTEPS, int NPAIR, int CC, int TWO, int PB, bool WLDS, bool PRE = false>
__global__ __launch_bounds__(256) void mbconv_a_kernel(const _Float16* __restrict__ X,     // [B][H][W][Cin]
                                                       const _Float16* __restrict__ Wexp,  // [Ce][32*KSTEPS] natural rows
                                                       const float* __restrict__ bexp,     // [Ce]
                                                       const float* __restrict__ Wdw,      // [KS*KS][Ce]
                                                       const float* __restrict__ bdw,      // [Ce]
                                                       _Float16* __restrict__ out,         // [B][Ho][Wo][Ce]
                                                       float* __restrict__ pool_part,      // [B][ntiles][Ce]
                                                       int H, int W, int Cin, int Ce, int Ho, int Wo, int pad, int TH,
                                                       int tiles_x, int wl_off, int red_off, int nB,
                                                       const _Float16* __restrict__ Wfrag, int wfr_off,
                                                       const _Float16* __restrict__ pre_w = nullptr,   // [64][8] project fragment
                                                       const float* __restrict__ pre_b = nullptr,      // [16]
                                                       const float* __restrict__ pre_gate = nullptr)   // [B][32]
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert(!PRE || (KSTEPS == 1 && PB == 1), "PRE: one k-step, one patch per workgroup");
    constexpr int Kp = 32 * KSTEPS;
    constexpr int TWo = TWO, CCG = CC / 8, S = 256 / CCG;
    constexpr int ES = CC * 2 + 16;  // bytes per E row
    constexpr int NTC = CC / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int tile = blockIdx.x, chunk = blockIdx.y, b = blockIdx.z * PB;
    const int nb = (nB - b) < PB ? (nB - b) : PB;   // patches this workgroup really has
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TWo;
    // input window of this tile, clipped to the image
    int wy0 = oy0 * ST - pad, wy1 = (oy0 + TH - 1) * ST - pad + KS;
    int wx0 = ox0 * ST - pad, wx1 = (ox0 + TWo - 1) * ST - pad + KS;
    wy0 = wy0 < 0 ? 0 : wy0;
    wx0 = wx0 < 0 ? 0 : wx0;
    wy1 = wy1 > H ? H : wy1;
    wx1 = wx1 > W ? W : wx1;
    const int ww = wx1 - wx0;
    const int P1 = (wy1 - wy0) * ww;                // positions of one patch's window
    const int P = (PB > 1 ? nb : 1) * P1;           // PB > 1: windows are whole images, stacked patch after patch
    const unsigned wmagic = (65536u + ww - 1) / ww;  // p / ww == (p * wmagic) >> 16 for p < 65536 / ww
    float* wl = reinterpret_cast<float*>(smem + wl_off);    // [KS*KS][CC] depthwise taps of this chunk, then bias [CC]
    float* bl = wl + KS * KS * CC;                          // expand bias of this chunk
    float* red = reinterpret_cast<float*>(smem + red_off);  // [S][CC]; aliases E (used after phase 2)
    // ---------------- phase 0: issue every global load this workgroup needs ----------------
    int p[NPAIR][2];
    h8 xf[NPAIR][2][KSTEPS];
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pp = ((pr * 4 + wave) * 2 + i) * 16 + m;
            p[pr][i] = pp;
            const bool ok = pp < P;
            const _Float16* xp;
            if (PB > 1) {   // whole images: position pp of the group is row b*H*W + pp of the NHWC tensor
                xp = X + ((size_t)b * H * W + (ok ? pp : 0)) * Cin + q * 8;
            } else {
                const int py = ok ? (int)(((unsigned)pp * wmagic) >> 16) : 0, px = ok ? pp - py * ww : 0;
                xp = X + (((size_t)b * H + wy0 + py) * W + wx0 + px) * Cin + q * 8;
            }
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok && ks * 32 + q * 8 < Cin) v = *reinterpret_cast<const h8*>(xp + ks * 32);
                xf[pr][i][ks] = v;
            }
        }
    if (PRE) {
        const h8 wpre = *reinterpret_cast<const h8*>(pre_w + lane * 8);
        const f4 bpre = *reinterpret_cast<const f4*>(pre_b + 4 * q);
        const f4 g0 = *reinterpret_cast<const f4*>(pre_gate + (size_t)b * 32 + 8 * q);
        const f4 g1 = *reinterpret_cast<const f4*>(pre_gate + (size_t)b * 32 + 8 * q + 4);
#pragma unroll
        for (int pr = 0; pr < NPAIR; ++pr)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint4 gx = gate_h8(*reinterpret_cast<const uint4*>(&xf[pr][i][0]), g0, g1);
                const f4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wpre, *reinterpret_cast<const h8*>(&gx), bpre, 0, 0, 0);
                h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (_Float16)acc[j];   // block 0's output, rounded to fp16 as the separate path stores it
                xf[pr][i][0] = v;
            }
    }
    if (WLDS) {
        // eight 16-byte pieces per thread and round trip (as a rolled `dst[i] = src[i]` loop every piece was a load, s_waitcnt vmcnt(0),
        // ds_write: B4's 7x7 stage stages 84 KB per workgroup = 21 exposed L2 round trips before the first MFMA)
        const uint4* src = reinterpret_cast<const uint4*>(Wfrag + (size_t)chunk * NTC * KSTEPS * 512);
        uint4* dst = reinterpret_cast<uint4*>(smem + wfr_off);
        constexpr int NPC = NTC * KSTEPS * 64;
        for (int i0 = tid; i0 < NPC; i0 += 8 * 256) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[i0 + 256 * u < NPC ? i0 + 256 * u : NPC - 1];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + 256 * u < NPC) dst[i0 + 256 * u] = v[u];
        }
    }
    {
        constexpr int NTAP = KS * KS * CC, NIT = (NTAP + 255) / 256;
        float tv[NIT];   // (all requests first, see above)
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = tid + 256 * u < NTAP ? tid + 256 * u : NTAP - 1;
            const int tap = i / CC, c = i - tap * CC;
            tv[u] = Wdw[(size_t)tap * Ce + chunk * CC + c];
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u)
            if (tid + 256 * u < NTAP) wl[tid + 256 * u] = tv[u];
    }
    if (tid < CC) bl[tid] = bexp[chunk * CC + tid];
    __syncthreads();
    // ---------------- phase 1: expand GEMM into LDS ----------------
    {
        const _Float16* wbase = Wexp + ((size_t)chunk * CC + m) * Kp + q * 8;
        const _Float16* wfr = reinterpret_cast<const _Float16*>(smem + wfr_off);
        h8 wn[KSTEPS];
        if (!WLDS) {
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) wn[ks] = *reinterpret_cast<const h8*>(wbase + ks * 32);
        }
        for (int t = 0; t < NTC; ++t) {
            h8 wc[KSTEPS];
            if (WLDS) {
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks)
                    wc[ks] = *reinterpret_cast<const h8*>(wfr + ((t * KSTEPS + ks) * 64 + lane) * 8);
                // all KSTEPS fragments in one LDS round trip: left alone the compiler sinks each read next to its two MFMAs behind an
                // s_waitcnt lgkmcnt(0) (B4's 7x7 stage: fourteen exposed round trips per 16 output channels)
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) wc[ks] = wn[ks];
                if (t + 1 < NTC) {
#pragma unroll
                    for (int ks = 0; ks < KSTEPS; ++ks)
                        wn[ks] = *reinterpret_cast<const h8*>(wbase + (size_t)(t + 1) * 16 * Kp + ks * 32);
                }
            }
            const f4 bv = *reinterpret_cast<const f4*>(bl + t * 16 + 4 * q);  // bias = accumulator init
#pragma unroll
            for (int pr = 0; pr < NPAIR; ++pr) {
                if (((pr * 4 + wave) * 2) * 16 >= P) continue;  // wave-uniform: no position in this pair
                f4 a0 = bv, a1 = bv;
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[ks], xf[pr][0][ks], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[ks], xf[pr][1][ks], a1, 0, 0, 0);
                }
                h4 o0, o1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o0[j] = (_Float16)silu_scaled(a0[j]);
                    o1[j] = (_Float16)silu_scaled(a1[j]);
                }
                if (p[pr][0] < P) *reinterpret_cast<h4*>(smem + p[pr][0] * ES + (t * 16 + 4 * q) * 2) = o0;
                if (p[pr][1] < P) *reinterpret_cast<h4*>(smem + p[pr][1] * ES + (t * 16 + 4 * q) * 2) = o1;
            }
        }
    }
    __syncthreads();
    // ---------------- phase 2: depthwise from LDS ----------------
    const bool active = tid < CCG * S;
    const int cg = tid % CCG, s = tid / CCG;
    const int cglob = chunk * CC + cg * 8;
    constexpr int spr = TWo / TW;
    const int nstrips = TH * spr;
    float pooled[PB][8];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int j = 0; j < 8; ++j) pooled[pb][j] = 0.f;
    if (active) {
        float bs[8];
        {
            const f4 b0 = *reinterpret_cast<const f4*>(bdw + cglob);
            const f4 b1 = *reinterpret_cast<const f4*>(bdw + cglob + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { bs[j] = b0[j]; bs[4 + j] = b1[j]; }
        }
        constexpr int NX = (TW - 1) * ST + KS;
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            if (pb >= nb) break;
            _Float16* outb = out + (size_t)(b + pb) * Ho * Wo * Ce + cglob;
            const int ebase = pb * P1;   // first E row of this patch
            for (int strip = s; strip < nstrips; strip += S) {
                const int oyl = strip / spr;
                const int oy = oy0 + oyl, ox = ox0 + (strip - oyl * spr) * TW;
                float acc[TW][8];
#pragma unroll
                for (int t = 0; t < TW; ++t)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[t][j] = bs[j];
#pragma unroll 1
                for (int ky = 0; ky < KS; ++ky) {
                    const int iy = oy * ST - pad + ky;
                    if (iy < 0 || iy >= H) continue;
                    const int rbase = ebase + (iy - wy0) * ww - wx0;  // E row of (iy, ix) is rbase + ix
                    float wk[KS][8];
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        const f4 w0 = *reinterpret_cast<const f4*>(wl + (ky * KS + kx) * CC + cg * 8);
                        const f4 w1 = *reinterpret_cast<const f4*>(wl + (ky * KS + kx) * CC + cg * 8 + 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { wk[kx][j] = w0[j]; wk[kx][4 + j] = w1[j]; }
                    }
                    // the row's NX operands in one LDS round trip (clamped addresses, zeros selected afterwards): read behind
                    // `if (inside)` next to their taps they were NX dependent round trips per kernel row
                    uint4 vrow[NX];
#pragma unroll
                    for (int xr = 0; xr < NX; ++xr) {
                        const int ix = ox * ST - pad + xr;
                        const int ixc = ix < 0 ? 0 : (ix < W ? ix : W - 1);
                        vrow[xr] = *reinterpret_cast<const uint4*>(smem + (rbase + ixc) * ES + cg * 16);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int xr = 0; xr < NX; ++xr) {
                        const int ix = ox * ST - pad + xr;
                        const bool in = ix >= 0 && ix < W;
                        const uint4 v = {in ? vrow[xr].x : 0u, in ? vrow[xr].y : 0u, in ? vrow[xr].z : 0u, in ? vrow[xr].w : 0u};
#pragma unroll
                        for (int t = 0; t < TW; ++t) {
                            const int kx = xr - t * ST;
                            if (kx >= 0 && kx < KS) {
                                acc[t][0] = fma_mix_lo(v.x, wk[kx][0], acc[t][0]);
                                acc[t][1] = fma_mix_hi(v.x, wk[kx][1], acc[t][1]);
                                acc[t][2] = fma_mix_lo(v.y, wk[kx][2], acc[t][2]);
                                acc[t][3] = fma_mix_hi(v.y, wk[kx][3], acc[t][3]);
                                acc[t][4] = fma_mix_lo(v.z, wk[kx][4], acc[t][4]);
                                acc[t][5] = fma_mix_hi(v.z, wk[kx][5], acc[t][5]);
                                acc[t][6] = fma_mix_lo(v.w, wk[kx][6], acc[t][6]);
                                acc[t][7] = fma_mix_hi(v.w, wk[kx][7], acc[t][7]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < TW; ++t) {
                    h8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float y = silu_scaled(acc[t][j]);
                        pooled[pb][j] += y;
                        o[j] = (_Float16)y;
                    }
                    *reinterpret_cast<h8*>(outb + ((size_t)oy * Wo + ox + t) * Ce) = o;
                }
            }
        }
    }
    __syncthreads();  // every wave is done reading E: its space is reused for the pool scratch [PB][S][CC]
    if (active) {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[(pb * S + s) * CC + cg * 8 + j] = pooled[pb][j];
    }
    __syncthreads();
    for (int e = tid; e < PB * CC; e += 256) {
        const int pb = e / CC, c = e - pb * CC;
        if (pb >= nb) continue;
        float sum = 0.f;
        for (int ss = 0; ss < S; ++ss) sum += red[(pb * S + ss) * CC + c];
        pool_part[((size_t)(b + pb) * gridDim.x + tile) * Ce + chunk * CC + c] = sum;
    }
}

// ---------------------------------------------------------------------------------------------
// mbconv_d_kernel: the fused MBConv front half with the depthwise taps on v_dot2c_f32_f16.
// Same decomposition as mbconv_a_kernel (patch x output tile x chunk of CC channels, phase 1 = expand
// GEMM + SiLU into LDS, phase 2 = depthwise + SiLU + pool partials), but the expanded tile is stored
// PAIR-INTERLEAVED: E2[row][xp][c] is one dword = (E[row][2xp][c], E[row][2xp+1][c]) -- two horizontally
// adjacent pixels of one channel, pairs aligned to even absolute x.  One v_dot2c then does TWO taps
// (fp16 x fp16 products, fp32 accumulate): 3 instead of 5 per kernel row for k=5, 2 instead of 3 for k=3,
// and since E2 holds real zeros outside the image no tap needs a bounds test.
//   * phase 1 runs the MFMA un-swapped (positions = A operand rows, channels = B operand columns) so a
//     lane ends up with 4 CONSECUTIVE positions of ONE channel = two ready-made pairs (two ds_write_b32).
//   * a thread owns 8 channels x 2 adjacent outputs (x even); for each kernel row it loads NP pairs
//     (32 B each) and the row's tap-pair weights (fp16 pairs built once per workgroup in LDS).
// Positions enumerate rows [wy0,wy1) x pair columns [xp0,xp1) x 2; a position with x >= W (odd W only)
// is written as zero.
// ---------------------------------------------------------------------------------------------
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

template <int KS, int ST>
struct DwPairs {
    static constexpr int PAD = (ST == 1) ? (KS - 1) / 2 : (KS == 3 ? 0 : 1);   // TF-same "before" pad
    static constexpr int OFF = PAD & 1;                                           // first tap's offset in its pair
    static constexpr int NP = (OFF + ST + KS + 1) / 2;                            // pairs a 2-output strip touches
};

template <int KS, int ST, int KSTEPS, int NPAIR, int CC, int TWO, int PB>
__global__ __launch_bounds__(256) void mbconv_d_kernel(const _Float16* __restrict__ X,     // [B][H][W][Cin]
                                                       const _Float16* __restrict__ Wexp,  // [Ce][32*KSTEPS] natural rows
                                                       const float* __restrict__ bexp,     // [Ce]
                                                       const float* __restrict__ Wdw,      // [KS*KS][Ce] fp32
                                                       const float* __restrict__ bdw,      // [Ce]
                                                       _Float16* __restrict__ out,         // [B][Ho][Wo][Ce]
                                                       float* __restrict__ pool_part,      // [B][ntiles][Ce]
                                                       int H, int W, int Cin, int Ce, int Ho, int Wo, int TH,
                                                       int tiles_x, int wl_off, int red_off, int nB)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using DP = DwPairs<KS, ST>;
    constexpr int PAD = DP::PAD, OFF = DP::OFF, NP = DP::NP;
    constexpr int Kp = 32 * KSTEPS;
    constexpr int CCG = CC / 8, S = 256 / CCG, NTC = CC / 16;
    constexpr int SPR = (TWO + 1) / 2;                     // 2-output strips per tile row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int tile = blockIdx.x, chunk = blockIdx.y, b = blockIdx.z * PB;
    const int nb = (nB - b) < PB ? (nB - b) : PB;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TWO;
    int wy0 = oy0 * ST - PAD, wy1 = (oy0 + TH - 1) * ST - PAD + KS;
    int wx0 = ox0 * ST - PAD, wx1 = (ox0 + TWO - 1) * ST - PAD + KS;
    wy0 = wy0 < 0 ? 0 : wy0;
    wx0 = wx0 < 0 ? 0 : wx0;
    wy1 = wy1 > H ? H : wy1;
    wx1 = wx1 > W ? W : wx1;
    const int xp0 = wx0 >> 1, xp1 = (wx1 + 1) >> 1;       // pair columns [xp0, xp1)
    const int npx = xp1 - xp0, rowlen = 2 * npx;
    const int P1 = (wy1 - wy0) * rowlen;                  // positions of one patch's window (even)
    const int P = (PB > 1 ? nb : 1) * P1;
    const unsigned rmagic = (65536u + rowlen - 1) / rowlen;
    uint32_t* E2 = reinterpret_cast<uint32_t*>(smem);      // [P/2][CC] pair dwords
    uint32_t* wl2 = reinterpret_cast<uint32_t*>(smem + wl_off);   // [KS][2][NP][CC] tap-pair weights (fp16 x2)
    float* bl = reinterpret_cast<float*>(wl2 + KS * 2 * NP * CC); // expand bias [CC]
    float* red = reinterpret_cast<float*>(smem + red_off);
    // ---------------- phase 0: input fragments, tap-pair weights, bias ----------------
    int p[NPAIR][2];
    h8 xf[NPAIR][2][KSTEPS];
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pp = ((pr * 4 + wave) * 2 + i) * 16 + m;
            p[pr][i] = pp;
            bool ok = pp < P;
            int pb = 0, pl = pp;
            if (PB > 1) { pb = pp >= P1 ? (pp >= 2 * P1 ? (pp >= 3 * P1 ? 3 : 2) : 1) : 0; pl = pp - pb * P1; }
            const int py = (int)(((unsigned)pl * rmagic) >> 16), pxx = pl - py * rowlen;
            const int ix = 2 * xp0 + pxx;
            ok = ok && ix < W;
            const _Float16* xp = X + (((size_t)(b + pb) * H + wy0 + py) * W + (ok ? ix : 0)) * Cin + q * 8;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok && ks * 32 + q * 8 < Cin) v = *reinterpret_cast<const h8*>(xp + ks * 32);
                xf[pr][i][ks] = v;
            }
        }
    for (int i = tid; i < KS * 2 * NP * CC; i += 256) {
        const int c = i % CC, r = i / CC;
        const int ip = r % NP, t = (r / NP) % 2, ky = r / (2 * NP);
        const int kx0 = 2 * ip - OFF - t * ST, kx1 = kx0 + 1;
        h2 w;
        w[0] = (kx0 >= 0 && kx0 < KS) ? (_Float16)Wdw[(size_t)(ky * KS + kx0) * Ce + chunk * CC + c] : (_Float16)0.0f;
        w[1] = (kx1 >= 0 && kx1 < KS) ? (_Float16)Wdw[(size_t)(ky * KS + kx1) * Ce + chunk * CC + c] : (_Float16)0.0f;
        wl2[i] = *reinterpret_cast<uint32_t*>(&w);
    }
    if (tid < CC) bl[tid] = bexp[chunk * CC + tid];
    __syncthreads();
    // ---------------- phase 1: expand GEMM (un-swapped) into pair-interleaved LDS ----------------
    {
        const _Float16* wbase = Wexp + ((size_t)chunk * CC + m) * Kp + q * 8;
        h8 wn[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) wn[ks] = *reinterpret_cast<const h8*>(wbase + ks * 32);
        for (int t = 0; t < NTC; ++t) {
            h8 wc[KSTEPS];
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) wc[ks] = wn[ks];
            if (t + 1 < NTC) {
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks)
                    wn[ks] = *reinterpret_cast<const h8*>(wbase + (size_t)(t + 1) * 16 * Kp + ks * 32);
            }
            const float bv = bl[t * 16 + m];   // this lane's channel: bias = accumulator init
#pragma unroll
            for (int pr = 0; pr < NPAIR; ++pr) {
                const int pbase = ((pr * 4 + wave) * 2) * 16;
                if (pbase >= P) continue;  // wave-uniform
                f4 a0 = {bv, bv, bv, bv}, a1 = a0;
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[pr][0][ks], wc[ks], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[pr][1][ks], wc[ks], a1, 0, 0, 0);
                }
                // lane (m = channel, q): positions pbase + 4q + j  (fragment 0) and pbase + 16 + 4q + j (fragment 1)
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const f4 a = f ? a1 : a0;
                    const int pq = pbase + f * 16 + 4 * q;     // first of this lane's 4 positions (multiple of 4)
                    if (pq >= P) continue;
                    int pl = pq;
                    if (PB > 1) { const int pb = pq >= P1 ? (pq >= 2 * P1 ? (pq >= 3 * P1 ? 3 : 2) : 1) : 0; pl = pq - pb * P1; }
                    h2 v0, v1;
                    v0[0] = (_Float16)silu_scaled(a[0]);
                    v0[1] = (_Float16)silu_scaled(a[1]);
                    v1[0] = (_Float16)silu_scaled(a[2]);
                    v1[1] = (_Float16)silu_scaled(a[3]);
                    if (W & 1) {   // odd image width: the pad pixel of the last pair of every row is a real zero
                        const int py = (int)(((unsigned)pl * rmagic) >> 16), pxx = pl - py * rowlen;
                        if (2 * xp0 + pxx + 1 >= W) v0[1] = (_Float16)0.0f;
                        const int pl2 = pl + 2;
                        const int py2 = (int)(((unsigned)pl2 * rmagic) >> 16), pxx2 = pl2 - py2 * rowlen;
                        if (2 * xp0 + pxx2 + 1 >= W) v1[1] = (_Float16)0.0f;
                    }
                    uint32_t* dst = E2 + (size_t)(pq >> 1) * CC + t * 16 + m;
                    dst[0] = *reinterpret_cast<uint32_t*>(&v0);
                    if (pq + 2 < P) dst[CC] = *reinterpret_cast<uint32_t*>(&v1);
                }
            }
        }
    }
    __syncthreads();
    // ---------------- phase 2: depthwise on v_dot2c ----------------
    const bool active = tid < CCG * S;
    const int cg = tid % CCG, s = tid / CCG;
    const int cglob = chunk * CC + cg * 8;
    const int nstrips = TH * SPR;
    float pooled[PB][8];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int j = 0; j < 8; ++j) pooled[pb][j] = 0.f;
    if (active) {
        float bs[8];
        {
            const f4 b0 = *reinterpret_cast<const f4*>(bdw + cglob);
            const f4 b1 = *reinterpret_cast<const f4*>(bdw + cglob + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { bs[j] = b0[j]; bs[4 + j] = b1[j]; }
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            if (pb >= nb) break;
            _Float16* outb = out + (size_t)(b + pb) * Ho * Wo * Ce + cglob;
            const int ebase = pb * (P1 >> 1);   // first pair of this patch
            for (int strip = s; strip < nstrips; strip += S) {
                const int oyl = strip / SPR;
                const int oy = oy0 + oyl, ox = ox0 + (strip - oyl * SPR) * 2;   // ox even
                float acc[2][8];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[t][j] = bs[j];
                const int fp = (ox * ST - PAD - OFF) / 2 - xp0;   // first pair column (window-relative; may be < 0)
#pragma unroll 1
                for (int ky = 0; ky < KS; ++ky) {
                    const int iy = oy * ST - PAD + ky;
                    if (iy < 0 || iy >= H) continue;
                    const uint32_t* erow = E2 + (size_t)(ebase + (iy - wy0) * npx) * CC + cg * 8;
                    const uint32_t* wrow = wl2 + (size_t)ky * 2 * NP * CC + cg * 8;
#pragma unroll
                    for (int ip = 0; ip < NP; ++ip) {
                        const int xpc = fp + ip;
                        if (xpc < 0 || xpc >= npx) continue;   // whole pair outside the image: contributes zero
                        const uint4 d0 = *reinterpret_cast<const uint4*>(erow + (size_t)xpc * CC);
                        const uint4 d1 = *reinterpret_cast<const uint4*>(erow + (size_t)xpc * CC + 4);
                        const uint32_t dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            constexpr int dummy = 0;
                            (void)dummy;
                            const int kx0 = 2 * ip - OFF - t * ST;
                            if (kx0 + 1 < 0 || kx0 >= KS) continue;   // compile-time: this pair carries no tap of output t
                            const uint4 w0 = *reinterpret_cast<const uint4*>(wrow + (size_t)(t * NP + ip) * CC);
                            const uint4 w1 = *reinterpret_cast<const uint4*>(wrow + (size_t)(t * NP + ip) * CC + 4);
                            const uint32_t wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                acc[t][j] = __builtin_amdgcn_fdot2(*reinterpret_cast<const h2*>(&dv[j]),
                                                                   *reinterpret_cast<const h2*>(&wv[j]), acc[t][j], false);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (ox + t >= ox0 + TWO) continue;   // odd tile width: second output of the last strip does not exist
                    h8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float y = silu_scaled(acc[t][j]);
                        pooled[pb][j] += y;
                        o[j] = (_Float16)y;
                    }
                    *reinterpret_cast<h8*>(outb + ((size_t)oy * Wo + ox + t) * Ce) = o;
                }
            }
        }
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[(pb * S + s) * CC + cg * 8 + j] = pooled[pb][j];
    }
    __syncthreads();
    for (int e = tid; e < PB * CC; e += 256) {
        const int pb = e / CC, c = e - pb * CC;
        if (pb >= nb) continue;
        float sum = 0.f;
        for (int ss = 0; ss < S; ++ss) sum += red[(pb * S + ss) * CC + c];
        pool_part[((size_t)(b + pb) * gridDim.x + tile) * Ce + chunk * CC + c] = sum;
    }
}

// ---------------------------------------------------------------------------------------------
// tail7_kernel: the 7x7 MBConv blocks b12..b14 (192 -> 1152 -> 192, k5 s1, squeeze-excite, skip) of ONE
// patch per workgroup, chained inside the CU's 160 KB of LDS: the block input/output X[49][192] and the
// expanded tensor ED[49][1152] never leave the CU, squeeze-excite needs no second launch (the whole
// patch is local), and a 3-block chain is one launch instead of nine.  512 threads = 8 waves.
//   expand   swapped MFMA (A = weight fragments streamed from L2, B = pixel fragments held in registers for
//            the whole phase); wave w owns expanded channels [144w, 144w+144); silu -> ED fp16.
//   dw       thread = one expanded channel: its 49 inputs become 28 pixel-pair dwords in registers, the 49
//            outputs run on v_dot2c (same tap pairs and order as mbconv_d_kernel), silu, written back IN PLACE
//            (a channel's column is private to its thread), pooled sum -> LDS.
//   SE       both FCs are matrix-vector products here: fp32 FMAs on fp16 weights (measured effect on the features
//            3e-5 relative), coalesced loads, fixed order.
// Weight streams are requested one phase ahead (registers) so that the L2 round trips hide behind compute.
//   gate     ED <- fp16(ED * gate) in place (same rounding as pw_gemm_kernel's gate-at-load).
//   project  swapped MFMA (A = weight fragments from L2, 4 k-steps ahead; B = pixels from ED); waves 0..3 own
//            two 16-channel output fragments, waves 4..7 one (12 fragments; every SIMD gets three);
//            + bias + residual -> X in place.
// Pixels enumerate y*7+x; a 16-pixel MFMA fragment past pixel 48 re-reads pixel 48 and is dropped.
// ---------------------------------------------------------------------------------------------
#define T7_PIX 49
#define T7_C 192
#define T7_CE 1152
#define T7_XS 400                                    // X row stride, bytes
#define T7_ES 2320                                   // ED row stride, bytes (580 dwords = 4 mod 64 banks)
#define T7_YS 656                                    // row stride of b15's 320-channel output (parked in ED)
#define T7_DS11 1552                                 // row stride of block 11's depthwise output when it is produced in LDS (24 k-steps + 16)
#define T7_OFF_X (T7_PIX * T7_ES)
#define T7_OFF_POOL (T7_OFF_X + T7_PIX * T7_XS)
#define T7_OFF_RS (T7_OFF_POOL + T7_CE * 4)
#define T7_OFF_GATE (T7_OFF_RS + 48 * 4)
#define T7_OFF_PART (T7_OFF_GATE + T7_CE * 4)
// DW4 (depthwise on 4x4x4 MFMA blocks): eight wave-private planar staging regions [16 channels][12 rows][8 columns] fp16, 200 bytes
// per channel (192 + 8: conflict-free 8-byte stores of the expand, 2-way 8-byte reads of the depthwise).  They alias gate + part (dead
// between a block's project and its squeeze-excite) and run on to the end of the CU's 160 KB: 163 680 of 163 840 bytes.
#define T7_PCS 200
#define T7_OFF_STG T7_OFF_GATE
#define T7_LDS (T7_OFF_STG + 8 * 16 * T7_PCS)
static_assert(T7_LDS >= T7_OFF_PART + 32 * 48 * 4 && T7_LDS <= 163840, "tail7 LDS map");

// 1: the squeeze-excite gate rides on the project's WEIGHT fragments (each wave scales the fragments it streams: 8 v_fma_mix per
// fragment between the MFMAs, product in fp32, one rounding) instead of on the expanded tensor (a pass over ED[49][1152] between two
// barriers: 3.7 k cycles per block).  Measured (round 3, build_variants via MMC_LIBRARY): the gate pass disappears but the project
// grows from 12.6 k to 16-19.6 k cycles -- the extra vector instructions are not free in an MFMA phase that is also LDS- and
// latency-bound at two waves per SIMD -- bench 229.3 k vs 231.3 k patches/s.  0 (the round-2 gate pass) stays the default.
#ifndef T7_GATE_IN_WEIGHTS
#define T7_GATE_IN_WEIGHTS 0
#endif
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

// SiLU of a depthwise output row inside tail7's rounds (staged: all exponentials, packed adds, reciprocals, packed products)
#define DW_SILU(acc) silu_scaled_staged(acc)
// DW4 = true: blocks 12..15 run expand + depthwise as ONE phase per wave and 16-channel group, the depthwise conv on
// v_mfma_f32_4x4x4_16B_f16 (see mid14m_kernel): no barrier between the two, no thread = channel rounds, a third of the vector
// instructions.  DW4 = false: the round-2 phases (expand, barrier, depthwise in place on v_dot2c).
template <bool DW4>
__global__ __launch_bounds__(512) void tail7_kernel(TailArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ED = smem;
    unsigned char* XL = smem + T7_OFF_X;
    float* pooled = reinterpret_cast<float*>(smem + T7_OFF_POOL);
    float* gate = reinterpret_cast<float*>(smem + T7_OFF_GATE);
    float* part = reinterpret_cast<float*>(smem + T7_OFF_PART);
    float* rs = reinterpret_cast<float*>(smem + T7_OFF_RS);
    const int tid0 = threadIdx.x;
    const int b = blockIdx.x;
    const bool clk_on = a.dbg_clk && a.clk_sections;
    long long tkk[6] = {0, 0, 0, 0, 0, 0};   // whole-kernel stamps (production-mode phase clock, MMC_TAIL_CLK=1)
    if (clk_on) tkk[0] = (long long)__builtin_readcyclecounter();
    // ---- project conv pieces shared by the block loop and the b11 pre-block --------------------------------------
    // Output fragments (16 channels) nf0 .. nf0+nfn-1 of this wave; weight image [cout/16][KS][64 lanes][16 B].
    auto proj_prefetch = [&](const GLOBAL_AS _Float16* wproj, const GLOBAL_AS float* bproj, int KS, int nf0, int nfn, int lane, int q,
                             f4 (&pbias)[3], unsigned (&wo)[3], h8 (&wa)[3][4]) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int nf = nf0 + (i < nfn ? i : 0);   // surplus slots alias fragment nf0 (loaded, never used)
            pbias[i] = gload<f4>(bproj, (unsigned)(16 * nf + 4 * q) * 4u);
            wo[i] = (unsigned)((nf * KS * 64 + lane) * 16);
#pragma unroll
            for (int d = 0; d < 4; ++d) wa[i][d] = gload<h8>(wproj, wo[i] + (unsigned)(d * 1024));
        }
    };
    // mode 0: X <- fp16(acc + X) in place (192 outputs, skip); 1: ED <- fp16(acc) as [49][320] (b15); 2: X <- fp16(acc).
    // KS (k-steps, a multiple of 4) pixel fragments come from ED one k-step ahead of their MFMAs, weight fragments four
    // k-steps ahead.  The K order is part of the result: same for every workgroup.
    auto proj_run = [&](const GLOBAL_AS _Float16* wproj, int KS, int mode, int nf0, int nfn, int lane, int m, int q,
                        const int (&pixc)[4], f4 (&pbias)[3], unsigned (&wo)[3], h8 (&wa)[3][4], int es = T7_ES) {
        f4 acc[3][4];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int pf = 0; pf < 4; ++pf) acc[i][pf] = pbias[i];
        const unsigned char* bxp[4];
#pragma unroll
        for (int pf = 0; pf < 4; ++pf) bxp[pf] = ED + pixc[pf] * es + 16 * q;
        // NF = output fragments of this wave: straight-line loops per NF instead of wave-uniform branches inside one.
        // Weight fragments live in TWO register sets of four k-steps each: while the MFMAs of one set run, the other set
        // is refilled for four k-steps later, in consumption order.  (With one set refilled in place inside a rolled loop
        // the compiler loaded into temporaries and copied them back at the loop end behind an s_waitcnt vmcnt(0): the
        // four-k-step prefetch distance collapsed to half an iteration and every iteration exposed an L2 round trip.)
        auto k_loop = [&](auto nf_tag, auto ks_tag) {
            constexpr int NF = decltype(nf_tag)::value, KSC = decltype(ks_tag)::value;
            static_assert(KSC % 4 == 0 && KSC >= 8, "k-steps come in sets of four");
            h8 bx[4], bn[4], wb[NF][4];
#pragma unroll
            for (int pf = 0; pf < 4; ++pf) bx[pf] = *reinterpret_cast<const h8*>(bxp[pf]);
            // gate values of the k-step's eight channels of this lane (A operand: lane (m, q) holds k = 32 ks + 8 q .. + 7), one step ahead
            f4 gc0 = *reinterpret_cast<const f4*>(gate + 8 * q), gc1 = *reinterpret_cast<const f4*>(gate + 8 * q + 4), gn0, gn1;
            auto gated = [&](const h8& w) -> h8 {
                if (!T7_GATE_IN_WEIGHTS) return w;
                const uint4 o = gate_h8(*reinterpret_cast<const uint4*>(&w), gc0, gc1);   // fp32 product, one rounding (wait states padded inside)
                return *reinterpret_cast<const h8*>(&o);
            };
            auto step = [&](int ks, const h8 (&wset)[3][4], int d) {
                const int kn = ks + 1 < KSC ? ks + 1 : KSC - 1;   // last step re-reads itself (unused)
#pragma unroll
                for (int pf = 0; pf < 4; ++pf) bn[pf] = *reinterpret_cast<const h8*>(bxp[pf] + 64 * kn);
                if (T7_GATE_IN_WEIGHTS) {
                    gn0 = *reinterpret_cast<const f4*>(gate + 32 * kn + 8 * q);
                    gn1 = *reinterpret_cast<const f4*>(gate + 32 * kn + 8 * q + 4);
                }
                h8 wg[NF];
#pragma unroll
                for (int i = 0; i < NF; ++i) wg[i] = gated(wset[i][d]);
#pragma unroll
                for (int i = 0; i < NF; ++i)
#pragma unroll
                    for (int pf = 0; pf < 4; ++pf)
                        acc[i][pf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wg[i], bx[pf], acc[i][pf], 0, 0, 0);
#pragma unroll
                for (int pf = 0; pf < 4; ++pf) bx[pf] = bn[pf];
                gc0 = gn0; gc1 = gn1;
            };
            auto stepb = [&](int ks, int d) {
                const int kn = ks + 1 < KSC ? ks + 1 : KSC - 1;
#pragma unroll
                for (int pf = 0; pf < 4; ++pf) bn[pf] = *reinterpret_cast<const h8*>(bxp[pf] + 64 * kn);
                if (T7_GATE_IN_WEIGHTS) {
                    gn0 = *reinterpret_cast<const f4*>(gate + 32 * kn + 8 * q);
                    gn1 = *reinterpret_cast<const f4*>(gate + 32 * kn + 8 * q + 4);
                }
                h8 wg[NF];
#pragma unroll
                for (int i = 0; i < NF; ++i) wg[i] = gated(wb[i][d]);
#pragma unroll
                for (int i = 0; i < NF; ++i)
#pragma unroll
                    for (int pf = 0; pf < 4; ++pf)
                        acc[i][pf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wg[i], bx[pf], acc[i][pf], 0, 0, 0);
#pragma unroll
                for (int pf = 0; pf < 4; ++pf) bx[pf] = bn[pf];
                gc0 = gn0; gc1 = gn1;
            };
#pragma unroll 1
            for (int k0 = 0; k0 + 8 <= KSC; k0 += 8) {
#pragma unroll
                for (int d = 0; d < 4; ++d) {     // set A computes, set B is requested for k0+4 .. k0+7
#pragma unroll
                    for (int i = 0; i < NF; ++i) wb[i][d] = gload<h8>(wproj, wo[i] + (unsigned)((k0 + 4 + d) * 1024));
                    PIN_VMEM();   // the request stays HERE: the scheduler otherwise sinks it next to its use
                    step(k0 + d, wa, d);
                    PIN_VMEM();
                }
#pragma unroll
                for (int d = 0; d < 4; ++d) {     // set B computes, set A is requested for k0+8 .. k0+11 (clamped: unused past the end)
                    const int kw = k0 + 8 + d < KSC ? k0 + 8 + d : KSC - 1;
#pragma unroll
                    for (int i = 0; i < NF; ++i) wa[i][d] = gload<h8>(wproj, wo[i] + (unsigned)(kw * 1024));
                    PIN_VMEM();
                    stepb(k0 + 4 + d, d);
                    PIN_VMEM();
                }
            }
            if (KSC % 8) {
#pragma unroll
                for (int d = 0; d < 4; ++d) step(KSC - 4 + d, wa, d);
            }
        };
        auto run_nf = [&](auto ks_tag) {
            if (nfn == 3) k_loop(std::integral_constant<int, 3>{}, ks_tag);
            else if (nfn == 2) k_loop(std::integral_constant<int, 2>{}, ks_tag);
            else k_loop(std::integral_constant<int, 1>{}, ks_tag);
        };
        if (KS == 36) run_nf(std::integral_constant<int, 36>{});
        else run_nf(std::integral_constant<int, 24>{});
        if (mode == 1) {   // b15: the result replaces ED (all reads of ED are done after the barrier)
            T7_BAR();
#pragma unroll
            for (int pf = 0; pf < 4; ++pf) {
                if (16 * pf + m >= T7_PIX) continue;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if (i >= nfn) continue;
                    h4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (_Float16)acc[i][pf][j];
                    *reinterpret_cast<h4*>(ED + (16 * pf + m) * T7_YS + (16 * (nf0 + i) + 4 * q) * 2) = o;
                }
            }
        } else {           // 192 outputs into X (each lane owns its elements: in place is safe)
#pragma unroll
            for (int pf = 0; pf < 4; ++pf) {
                if (16 * pf + m >= T7_PIX) continue;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (i >= nfn) continue;
                    h4* px = reinterpret_cast<h4*>(XL + (16 * pf + m) * T7_XS + (16 * (nf0 + i) + 4 * q) * 2);
                    h4 r = {0, 0, 0, 0};
                    if (mode == 0) r = *px;
                    h4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (_Float16)(acc[i][pf][j] + (float)r[j]);
                    *px = o;
                }
            }
        }
    };
    if (a.pre_D || a.pre_X) {
        // ---- block 11, second half (its depthwise output D11[49][672] and pool sums come from mbconv_a_kernel, or are
        //      produced right here from the block's input when pre_X is given): squeeze-excite, gate, project
        //      672 -> 192 (no skip) -> X.  Same recipes as in the block loop below. ----
        const int DS = a.pre_X ? T7_DS11 : T7_ES;   // row stride of D11 in LDS
        const int tid = tid0, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int m = lane & 15, q = lane >> 4;
        const GLOBAL_AS _Float16* wr_t = sgpr_ptr<_Float16>(a.pre_wr_t);
        const GLOBAL_AS _Float16* we_t = sgpr_ptr<_Float16>(a.pre_we_t);
        const GLOBAL_AS _Float16* wproj = sgpr_ptr<_Float16>(a.pre_wproj);
        const GLOBAL_AS float* bproj = sgpr_ptr<float>(a.pre_bproj);
        // FC weights first (registers): thread = (4 squeeze outputs j4, one of 48 channel slices) / channels 2t, 2t+1
        const int sl = tid / 7, j4 = tid - sl * 7;
        const bool fc_thr = tid < 336;
        u2v w1[14];
#pragma unroll
        for (int i = 0; i < 14; ++i) w1[i] = gload<u2v>(wr_t, (unsigned)((((fc_thr ? sl : 0) + 48 * i) * 28 + 4 * j4) * 2));
        uint32_t w2[28];
#pragma unroll
        for (int j = 0; j < 28; ++j) w2[j] = gload<uint32_t>(we_t, (unsigned)((j * 672 + (fc_thr ? 2 * tid : 0)) * 2));
        const float be0 = fc_thr ? a.pre_be[2 * tid] : 0.f, be1 = fc_thr ? a.pre_be[2 * tid + 1] : 0.f;
        const float brv = tid < 28 ? a.pre_br[tid] : 0.f;
        if (a.pre_X) {
            // ---- block 11, FIRST half, inside the workgroup (the recipe of mid14_kernel<4,5,672,2>): the block input
            //      X11[196][112] goes to registers as pixel fragments once; per chunk of 96 expanded channels: expand (MFMA,
            //      weight fragments from L2 one ahead) -> silu -> E[196][96] in LDS -> depthwise 5x5 stride 2 (thread =
            //      channel x 2 output rows, 7x7 pixel-pair window in registers, v_dot2c) -> silu -> D11[49][672] compact in
            //      LDS + pool sums.  No launch, no D11 / pool tensor in HBM, no second read of them. ----
            constexpr int CH = 96, ES2 = 416;                  // E2[98 pixel pairs][96 channels], one dword per pair (as in mid14_kernel)
            // expanded chunk, behind the compact D11, with one zero input row above the image and four below it (the rows a stride-2
            // window reaches outside: read as zeros instead of being selected to zero register by register); 131.4 KB in all, X's
            // region (unused until block 11's project) included
            unsigned char* EB = ED + T7_PIX * T7_DS11 + 7 * ES2;
            float* pband = part;                               // [7][96] pool partials of the output rows
            const GLOBAL_AS _Float16* wexp = sgpr_ptr<_Float16>(a.pre_wexp);
            const GLOBAL_AS float* bexp = sgpr_ptr<float>(a.pre_bexp);
            const GLOBAL_AS uint32_t* dwp = sgpr_ptr<uint32_t>(a.pre_dwp);
            const GLOBAL_AS float* bdw = sgpr_ptr<float>(a.pre_bdw);
            const GLOBAL_AS _Float16* xgp = sgpr_ptr<_Float16>(a.pre_X) + (size_t)b * 196 * 112;
            const int npf = wave < 5 ? 2 : 1;
            const int pf0 = wave < 5 ? 2 * wave : wave + 5;
            h8 xb[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pix = 16 * (pf0 + (i < npf ? i : 0)) + m;
                const int pixc = pix < 196 ? pix : 195;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int kk = 32 * ks + 8 * q;
                    // K columns 112..127 re-read channels 104..111: finite values against the zero rows the host packs there
                    xb[i][ks] = gload<h8>(xgp, (unsigned)((pixc * 112 + (kk < 112 ? kk : 104)) * 2));
                }
            }
            for (int e = tid; e < T7_PIX * 12; e += 512) {   // k-steps 21..23 of D11 are zeros
                const int pix = e / 12, oc = e - pix * 12;
                *reinterpret_cast<uint4*>(ED + pix * T7_DS11 + 1344 + oc * 16) = uint4{0u, 0u, 0u, 0u};
            }
            for (int e = tid; e < 35 * (ES2 / 16); e += 512) {
                const int row = e / (ES2 / 16), c16 = e - row * (ES2 / 16);
                *reinterpret_cast<uint4*>(EB + (row < 7 ? row - 7 : row + 91) * ES2 + 16 * c16) = uint4{0u, 0u, 0u, 0u};
            }
            const int band = tid / CH, cd = tid - band * CH;   // depthwise role: channel cd, output row(s) by tid / 96 (see the depthwise phase)
            // Weight fragments AND the bias of the next output fragment are requested one fragment ahead (across the chunk
            // boundary too; fragment 42 = fragment 41 re-read, unused), bias first: a load needed now is never queued behind
            // loads needed later (vmcnt retires in order).  The wave's role (two pixel fragments or one) is a template argument
            // and lanes past pixel 195 store to a scratch word instead of branching, so a chunk's expand is one straight-line
            // block: requests pinned at the top of each fragment, SiLU staged over the fragment's 4 or 8 accumulators.
            h8 wn[4];
            float bsn = gload<float>(bexp, (unsigned)m * 4u);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wn[ks] = gload<h8>(wexp, (unsigned)((ks * 64 + lane) * 16));
            unsigned char* const scratch = reinterpret_cast<unsigned char*>(part) + 3072 + lane * 4;   // beyond pband ([7][96] floats), unread
            auto expand_chunk = [&](auto npf_tag, int chunk) {
                constexpr int NPF = decltype(npf_tag)::value;
#pragma unroll
                for (int nf = 0; nf < 6; ++nf) {
                    const int nfg = 6 * chunk + nf;
                    h8 wc[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) wc[ks] = wn[ks];
                    const float bs = bsn;
                    {
                        const int nxt = nfg + 1 < 42 ? nfg + 1 : 41;
                        bsn = gload<float>(bexp, (unsigned)(16 * nxt + m) * 4u);
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) wn[ks] = gload<h8>(wexp, (unsigned)(((nxt * 4 + ks) * 64 + lane) * 16));
                    }
                    PIN_VMEM();
                    const f4 bv = {bs, bs, bs, bs};
                    f4 acc[NPF];
#pragma unroll
                    for (int i = 0; i < NPF; ++i) acc[i] = bv;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)   // un-swapped: lane (m, q) = channel 16 nf + m of pixels 16 pf + 4q .. +3
#pragma unroll
                        for (int i = 0; i < NPF; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xb[i][ks], wc[ks], acc[i], 0, 0, 0);
                    float t[4 * NPF];
#pragma unroll
                    for (int i = 0; i < NPF; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) t[4 * i + j] = acc[i][j];
                    silu_scaled_staged(t);
#pragma unroll
                    for (int i = 0; i < NPF; ++i) {
                        const int pix0 = 16 * (pf0 + i) + 4 * q;
                        const h2 p0 = {(_Float16)t[4 * i], (_Float16)t[4 * i + 1]};
                        const h2 p1 = {(_Float16)t[4 * i + 2], (_Float16)t[4 * i + 3]};
                        const bool ok = pix0 < 196;
                        unsigned char* dst = EB + (pix0 >> 1) * ES2 + (16 * nf + m) * 4;
                        *reinterpret_cast<h2*>(ok ? dst : scratch) = p0;
                        *reinterpret_cast<h2*>(ok ? dst + ES2 : scratch + 256) = p1;
                    }
                }
            };
#pragma unroll 1
            for (int chunk = 0; chunk < 7; ++chunk) {
                uint32_t raw[15];
                const int cg = chunk * CH + cd;   // (waves 6, 7: some channel of the block, unused)
#pragma unroll
                for (int i = 0; i < 15; ++i) raw[i] = 0u;
                float dbias;
                {   // taps + bias in four 16-byte requests (layout [4][672][4]: see the depthwise rounds below)
                    u4v t4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) t4[j] = gload<u4v>(dwp, (unsigned)((j * 672 + cg) * 16));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        raw[4 * j] = t4[j].x; raw[4 * j + 1] = t4[j].y; raw[4 * j + 2] = t4[j].z;
                        if (j < 3) raw[4 * j + 3] = t4[j].w;
                    }
                    dbias = __builtin_bit_cast(float, (uint32_t)t4[3].w);
                }
                if (npf == 2) expand_chunk(std::integral_constant<int, 2>{}, chunk);
                else expand_chunk(std::integral_constant<int, 1>{}, chunk);
                T7_BAR();
                // Work item = (channel, output row): 7 x 96 = 672 items of 7 outputs over 512 threads, a thread keeping its channel
                // (tid % 96: its taps are in registers) -- pass 0: rows 0 .. 5 (row 5: channels 0 .. 31), pass 1 (waves 0 .. 2): row 6
                // in threads 0 .. 95, the rest of row 5 in threads 96 .. 191 (channels 0 .. 31 repeat pass 0's items: same values, same
                // addresses).  With bands of two rows in waves 0 .. 5, two SIMDs carried two loaded waves and two carried one.
#pragma unroll 1
                for (int pass = 0; pass < 2; ++pass) {
                    if (pass == 1 && wave >= 3) break;   // wave-uniform
                    const int oy = pass == 0 ? band : (band == 0 ? 6 : 5);
                    const unsigned char* col = EB + 4 * cd + ((2 * oy - 1) * 7) * ES2;   // (row 0 starts in the zero row above the image)
                    uint32_t P[5][7];
#pragma unroll
                    for (int r = 0; r < 5; ++r)
#pragma unroll
                        for (int pp = 0; pp < 7; ++pp) P[r][pp] = *reinterpret_cast<const uint32_t*>(col + (r * 7 + pp) * ES2);
                    unsigned char* dcol = ED + (chunk * CH + cd) * 2;
                    float acc[7];
                    bool started[7] = {false, false, false, false, false, false, false};
#pragma unroll
                    for (int ky = 0; ky < 5; ++ky) {
                        const uint32_t r0 = raw[3 * ky], r1 = raw[3 * ky + 1], r2 = raw[3 * ky + 2];
                        const uint32_t wq[3] = {r0 << 16, __builtin_amdgcn_alignbit(r1, r0, 16), __builtin_amdgcn_alignbit(r2, r1, 16)};
#pragma unroll
                        for (int ip = 0; ip < 3; ++ip)
#pragma unroll
                            for (int ox = 0; ox < 7; ++ox) {
                                const int xpc = ox - 1 + ip;
                                if (xpc < 0 || xpc > 6) continue;
                                if (!started[ox]) { acc[ox] = dot2_from(P[ky][xpc], wq[ip], dbias); started[ox] = true; }
                                else acc[ox] = __builtin_amdgcn_fdot2(*reinterpret_cast<const h2*>(&P[ky][xpc]),
                                                                      *reinterpret_cast<const h2*>(&wq[ip]), acc[ox], false);
                            }
                    }
                    silu_scaled_staged(acc);
                    f2 psum2 = {0.f, 0.f};
#pragma unroll
                    for (int ox = 0; ox < 6; ox += 2) {
                        const f2 v = {acc[ox], acc[ox + 1]};
                        psum2 = psum2 + v;
                        const uint32_t hv = cvt_pk_f16(acc[ox], acc[ox + 1]);
                        *reinterpret_cast<uint16_t*>(dcol + (oy * 7 + ox) * T7_DS11) = (uint16_t)hv;
                        *reinterpret_cast<uint16_t*>(dcol + (oy * 7 + ox + 1) * T7_DS11) = (uint16_t)(hv >> 16);
                    }
                    *reinterpret_cast<_Float16*>(dcol + (oy * 7 + 6) * T7_DS11) = (_Float16)acc[6];
                    pband[oy * CH + cd] = (psum2.x + psum2.y) + acc[6];
                }
                T7_BAR();
                if (tid < CH)
                    pooled[chunk * CH + tid] = (((pband[tid] + pband[CH + tid]) + (pband[2 * CH + tid] + pband[3 * CH + tid])) +
                                                (pband[4 * CH + tid] + pband[5 * CH + tid])) + pband[6 * CH + tid];
            }
            if (a.dbg_dw) {   // per-tensor mode: block 11's depthwise output as the separate kernels would have stored it
                T7_BAR();
                for (int e = tid; e < T7_PIX * 84; e += 512) {
                    const int pix = e / 84, oc = e - pix * 84;
                    *reinterpret_cast<uint4*>(a.dbg_dw + ((size_t)b * T7_PIX + pix) * 672 + oc * 8) = *reinterpret_cast<const uint4*>(ED + pix * T7_DS11 + oc * 16);
                }
            }
        } else {
            const _Float16* dg = a.pre_D + (size_t)b * T7_PIX * 672;
            for (int e = tid; e < T7_PIX * 96; e += 512) {   // 84 real 16-byte columns + 12 of zeros (k-steps 21..23)
                const int pix = e / 96, oc = e - pix * 96;
                h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (oc < 84) v = *reinterpret_cast<const h8*>(dg + pix * 672 + oc * 8);
                *reinterpret_cast<h8*>(ED + pix * T7_ES + oc * 16) = v;
            }
            for (int kk = tid; kk < 672; kk += 512) pooled[kk] = a.pre_pool[(size_t)b * 672 + kk];
        }
        T7_BAR();
        if (clk_on) tkk[1] = (long long)__builtin_readcyclecounter();
        if (fc_thr) {
            f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 14; ++i) {
                const float x = pooled[sl + 48 * i];
                acc[0] = fma_mix_lo(w1[i].x, x, acc[0]);
                acc[1] = fma_mix_hi(w1[i].x, x, acc[1]);
                acc[2] = fma_mix_lo(w1[i].y, x, acc[2]);
                acc[3] = fma_mix_hi(w1[i].y, x, acc[3]);
            }
            *reinterpret_cast<f4*>(part + sl * 32 + 4 * j4) = acc;
        }
        T7_BAR();
        if (tid < 28) {
            float s = 0.f;
#pragma unroll
            for (int w0 = 0; w0 < 48; w0 += 16) {
                float pv[16];
#pragma unroll
                for (int w = 0; w < 16; ++w) pv[w] = part[(w0 + w) * 32 + tid];
#pragma unroll
                for (int w = 0; w < 16; ++w) s += pv[w];
            }
            rs[tid] = silu_f(s * (float)(1.0 / (49.0 * 1.4426950408889634)) + brv);
        }
        T7_BAR();
        {
            float a0 = be0, a1 = be1;
#pragma unroll
            for (int j = 0; j < 28; ++j) {
                const float r = rs[j];
                a0 = fma_mix_lo(w2[j], r, a0);
                a1 = fma_mix_hi(w2[j], r, a1);
            }
            if (fc_thr) {
                const float g0 = sigmoid_f(a0), g1 = sigmoid_f(a1);
                gate[2 * tid] = g0;
                gate[2 * tid + 1] = g1;
                if (a.dbg_gate) {
                    a.dbg_gate[(size_t)b * 672 + 2 * tid] = g0;
                    a.dbg_gate[(size_t)b * 672 + 2 * tid + 1] = g1;
                }
            }
        }
        const bool lowh = wave < 4;
        const int nfn = lowh ? 2 : 1, nf0 = lowh ? 2 * wave : wave + 4;
        int pixc[4];
#pragma unroll
        for (int pf = 0; pf < 4; ++pf) pixc[pf] = (16 * pf + m) < T7_PIX ? (16 * pf + m) : (T7_PIX - 1);
        f4 pbias[3];
        unsigned wo[3];
        h8 wa[3][4];
        proj_prefetch(wproj, bproj, 24, nf0, nfn, lane, q, pbias, wo, wa);
        T7_BAR();
        if (clk_on) tkk[2] = (long long)__builtin_readcyclecounter();
        if (T7_GATE_IN_WEIGHTS) {
            if (tid < 96) gate[672 + tid] = 0.f;   // k-steps 21..23 are zero padding: their gate values must be finite
        } else
        if (tid < 504) {   // 84 groups of 8 channels x 6 pixel residues: the gate values stay in registers (see the blocks' gate pass)
            const int r6 = tid / 84, oc = tid - 84 * r6;
            const f4 g0 = *reinterpret_cast<const f4*>(gate + oc * 8);
            const f4 g1 = *reinterpret_cast<const f4*>(gate + oc * 8 + 4);
            unsigned char* pc = ED + oc * 16;
            for (int k0 = 0; k0 < 9; k0 += 3) {   // three pixels in flight
                uint4 v[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int pix = r6 + 6 * (k0 + j);
                    v[j] = *reinterpret_cast<const uint4*>(pc + (pix < T7_PIX ? pix : T7_PIX - 1) * DS);
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int pix = r6 + 6 * (k0 + j);
                    const uint4 o = gate_h8(v[j], g0, g1);
                    if (pix < T7_PIX) *reinterpret_cast<uint4*>(pc + pix * DS) = o;
                }
            }
        }
        T7_BAR();
        if (clk_on) tkk[3] = (long long)__builtin_readcyclecounter();
        proj_run(wproj, 24, 2, nf0, nfn, lane, m, q, pixc, pbias, wo, wa, DS);
        T7_BAR();
        if (clk_on) {
            tkk[4] = (long long)__builtin_readcyclecounter();
            if (tid0 == 0) for (int i = 0; i < 4; ++i) a.dbg_clk[((size_t)b * 8) * 8 + i] = (float)(tkk[i + 1] - tkk[i]);
        }
    } else if (a.in_wide) {
        // head-only use (per-tensor tests): the input is block 15's output [49][320]
        const int tid = tid0;
        const _Float16* xg = a.X + (size_t)b * T7_PIX * 320;
        for (int e = tid; e < T7_PIX * 40; e += 512) {
            const int pix = e / 40, p16 = e - pix * 40;
            *reinterpret_cast<h8*>(ED + pix * T7_YS + p16 * 16) = *reinterpret_cast<const h8*>(xg + pix * 320 + p16 * 8);
        }
        __syncthreads();
    } else {
        const int tid = tid0;
        const _Float16* xg = a.X + (size_t)b * T7_PIX * T7_C;
        for (int e = tid; e < T7_PIX * 24; e += 512) {
            const int pix = e / 24, p16 = e - pix * 24;
            *reinterpret_cast<h8*>(XL + pix * T7_XS + p16 * 16) = *reinterpret_cast<const h8*>(xg + pix * T7_C + p16 * 8);
        }
        __syncthreads();
    }
    bool out_wide = a.in_wide != 0;
#pragma unroll 1
    for (int nb = 0; nb < a.nblk; ++nb) {
        // One table row via scalar loads.  Pointers that come out of memory are "flat" to the compiler; the casts
        // restore the global address space so the weight streams are global_load (vmcnt only), not flat_load.
        const TailBlock Wt = a.blk[nb];
        struct {
            const GLOBAL_AS _Float16* wexp; const GLOBAL_AS float* bexp; const GLOBAL_AS uint32_t* dwp; const GLOBAL_AS float* bdw;
            const GLOBAL_AS _Float16* wr_t; const GLOBAL_AS float* br; const GLOBAL_AS _Float16* we_t; const GLOBAL_AS float* be;
            const GLOBAL_AS _Float16* wproj; const GLOBAL_AS float* bproj;
        } W = {sgpr_ptr<_Float16>(Wt.wexp), sgpr_ptr<float>(Wt.bexp), sgpr_ptr<uint32_t>(Wt.dwp), sgpr_ptr<float>(Wt.bdw),
               sgpr_ptr<_Float16>(Wt.wr_t), sgpr_ptr<float>(Wt.br), sgpr_ptr<_Float16>(Wt.we_t), sgpr_ptr<float>(Wt.be),
               sgpr_ptr<_Float16>(Wt.wproj), sgpr_ptr<float>(Wt.bproj)};
        // Thread indices are re-derived through an opaque move every iteration: otherwise the compiler hoists the
        // ~150 loop-invariant weight-fragment addresses of all phases out of the block loop and spills them.
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int m = lane & 15, q = lane >> 4;
        long long tk[7];
        int ntk = 0;
#define T7_TICK() do { if (a.dbg_clk) tk[ntk] = (long long)__builtin_readcyclecounter(); ++ntk; } while (0)
        T7_TICK();
        int pixc[4];   // this lane's pixel in each of the four pixel fragments (clamped)
#pragma unroll
        for (int pf = 0; pf < 4; ++pf) pixc[pf] = (16 * pf + m) < T7_PIX ? (16 * pf + m) : (T7_PIX - 1);
        // Depthwise taps of this thread's first channel: issued now, they arrive while the expand phase computes.
        // (3 dwords per kernel row ky: (k0,k1), (k2,k3), (k4,0) as fp16 pairs; a 3x3 kernel fills rows 0..2 with
        // (k0,k1), (k2,0), 0.)
        const bool ks3 = Wt.ks == 3;
        uint32_t rawA[15], rawB[15];
        float biasA, biasB;
        // taps + bias of a channel: four 16-byte requests (slots 0..14 = tap pairs, 15 = bias; 1 KB per wave-instruction) instead of 16
        // dword loads -- a timing-only build without the tap loads ran the depthwise phase in 21.0 k cycles instead of 25.5 k
        auto load_taps = [&](int ch, uint32_t (&raw)[15], float& bias) {
            u4v t4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t4[j] = gload<u4v>(W.dwp, (unsigned)((j * T7_CE + ch) * 16));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                raw[4 * j] = t4[j].x; raw[4 * j + 1] = t4[j].y; raw[4 * j + 2] = t4[j].z;
                if (j < 3) raw[4 * j + 3] = t4[j].w;
            }
            bias = __builtin_bit_cast(float, (uint32_t)t4[3].w);
        };
        if constexpr (!DW4) {
        load_taps(tid, rawA, biasA);
        // ---------------- expand: ED = silu(X . Wexp^T + b) ----------------
        {
            h8 xb[4][6];
#pragma unroll
            for (int pf = 0; pf < 4; ++pf)
#pragma unroll
                for (int ks = 0; ks < 6; ++ks)
                    xb[pf][ks] = *reinterpret_cast<const h8*>(XL + pixc[pf] * T7_XS + (32 * ks + 8 * q) * 2);
            // Every workgroup streams the same weights at about the same time; rotating the fragment order by
            // workgroup spreads the requests of the CUs that share an L2 over its channels.
            // (No per-workgroup rotation of the fragment order: the loop is unrolled, and the compiler picks the fused or the two-step
            // f32 -> f16 form of the last multiply per POSITION in it -- with a rotation the same channel came out in different last
            // bits in different workgroups; measured, the rotation bought nothing.)
            constexpr int rot = 0;
            // The weight fragments AND the bias of the next output fragment are requested one fragment ahead, bias first:
            // vmcnt retires in order, so a load that is needed now must never be issued behind loads that are needed later
            // (a bias load issued after the prefetch made every iteration wait for the whole prefetch: s_waitcnt vmcnt(0)).
            //
            // Software pipeline over the wave's nine output fragments: the 24 MFMAs of fragment i are interleaved, ONE MFMA
            // then THREE vector instructions, with the SiLU epilogue of fragment i-1 (staged over its 16 accumulators: all
            // exponentials, then all adds, all reciprocals, all products, then the packing and the four 8-byte LDS stores).
            // Run back to back, a wave's MFMA burst holds the matrix pipe while its vector port idles and its SiLU burst the
            // other way round, and the two waves of a SIMD do so in lockstep (measured: 384 cycles of MFMAs then ~800 of
            // epilogue per fragment and wave, strictly one after the other).  An MFMA occupies the issue port for 8 of its
            // 16 pipe cycles: three vector instructions fit in its shadow.  Same instructions, same values, other order.
            h8 wn[6];
            f4 bvn = gload<f4>(W.bexp, (unsigned)(16 * (9 * wave + rot) + 4 * q) * 4u);
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) wn[ks] = gload<h8>(W.wexp, (unsigned)((((9 * wave + rot) * 6 + ks) * 64 + lane) * 16));
            unsigned char* const scratch = reinterpret_cast<unsigned char*>(part) + lane * 8;   // masked rows store here (part is idle now)
            float tp[16], ep[16];   // previous fragment: accumulators / SiLU intermediates
            int nfp = 0;
            f4 bvn_cur = bvn;
            // vector operation number `op` of the epilogue of the previous fragment (52 in all: the adds and the products run two
            // values per instruction)
            auto epi_op = [&](int op) {
                if (op < 16) ep[op] = __builtin_amdgcn_exp2f(-tp[op]);
                else if (op < 24) {
                    const int i2 = 2 * (op - 16);
                    f2 v = {ep[i2], ep[i2 + 1]};
                    v = v + (f2){1.0f, 1.0f};
                    ep[i2] = v.x; ep[i2 + 1] = v.y;
                } else if (op < 40) ep[op - 24] = __builtin_amdgcn_rcpf(ep[op - 24]);
                else if (op < 48) {
                    const int i2 = 2 * (op - 40);
                    f2 v = {tp[i2], tp[i2 + 1]}, r = {ep[i2], ep[i2 + 1]};
                    v = v * r;
                    tp[i2] = v.x; tp[i2 + 1] = v.y;
                } else if (op < 52) {
                    const int pf = op - 48;
                    h4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (_Float16)tp[4 * pf + j];
                    unsigned char* dst = ED + (16 * pf + m) * T7_ES + (16 * nfp + 4 * q) * 2;
                    *reinterpret_cast<h4*>((16 * pf + m < T7_PIX) ? dst : scratch) = o;
                }
            };
#pragma unroll
            for (int i = 0; i < 10; ++i) {   // i = 9 only drains the last fragment's epilogue
                const int ir = i + rot >= 9 ? i + rot - 9 : i + rot;
                const int nf = 9 * wave + ir;
                h8 wc[6];
                f4 acc[4];
                if (i < 9) {
#pragma unroll
                    for (int ks = 0; ks < 6; ++ks) wc[ks] = wn[ks];
                    bvn_cur = bvn;
                    if (i + 1 < 9) {
                        const int nfn = 9 * wave + (ir + 1 >= 9 ? ir + 1 - 9 : ir + 1);
                        bvn = gload<f4>(W.bexp, (unsigned)(16 * nfn + 4 * q) * 4u);
#pragma unroll
                        for (int ks = 0; ks < 6; ++ks) wn[ks] = gload<h8>(W.wexp, (unsigned)(((nfn * 6 + ks) * 64 + lane) * 16));
                    }
                    PIN_VMEM();
                }
                const f4 bvc = bvn_cur;
#pragma unroll
                for (int slot = 0; slot < 24; ++slot) {
                    // (the first k-step takes the bias vector as its addend: no copy per accumulator)
                    if (i < 9) acc[slot & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[slot >> 2], xb[slot & 3][slot >> 2], slot < 4 ? bvc : acc[slot & 3], 0, 0, 0);
                    if (i > 0) {
#pragma unroll
                        for (int v = 0; v < 3; ++v)
                            if (3 * slot + v < 52) epi_op(3 * slot + v);
                    }
                    if (i > 0 && i < 9) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA ...
                        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // ... then three vector instructions
                    }
                }
                if (i < 9) {
#pragma unroll
                    for (int pf = 0; pf < 4; ++pf)
#pragma unroll
                        for (int j = 0; j < 4; ++j) tp[4 * pf + j] = acc[pf][j];
                    nfp = nf;
                }
            }
        }
        T7_BAR();
        T7_TICK();
        // ---------------- depthwise KSxKS (5 or 3) + silu, in place; pooled sums ----------------
        // Pixels of a row are paired (x even, x+1); a v_dot2c does two taps.  KS = 5: output x even uses
        // (k0,k1)(k2,k3)(k4,0) on the pairs from x-2, output x odd (0,k0)(k1,k2)(k3,k4) on the pairs from x-3 -- the
        // values mbconv_d_kernel keeps in LDS.  KS = 3: x even (0,k0)(k1,k2) on the pairs from x-2, x odd (k0,k1)(k2,0)
        // on the pairs from x-1.  The odd/even variants are derived from the raw row pairs by shifts.
        }
        const bool fc1_thr = tid < 384;
        const int cr = tid / 12, j4 = tid - cr * 12;   // FC1: thread = 4 outputs j x channels cr, cr+32, ...
        // Squeeze-excite weights arrive in 16-byte requests, two rows of the old 8-byte layout per request (the host pairs them:
        // TailBlock::wr_t / we_t): streamed from L2 by every workgroup, a wave-instruction costs ~16-20 cycles of the CU's memory
        // path whether it carries 512 bytes or 1 KB (tools/ubench/l2_stream.hip: 32-34 vs 51-55 B/clk), and the depthwise phase
        // waits on exactly this stream.  Same values in the same order as before.
        u4v fw1[18];   // 2 x 4 fp16 weights each (channels 64p + cr and 64p + 32 + cr), consumed by v_fma_mix_f32 without conversion
        u4v fw2[24];   // excite FC: thread = 4 consecutive channels x all 48 squeeze units (two per register quad)
        const bool fc2_thr = tid < 288;
        const int t2 = fc2_thr ? tid : 0;
        float brv = 0.f;
        if constexpr (DW4) {
        auto dw4_phase = [&](auto ks_tag) __attribute__((always_inline)) {
            // ---------------- expand + depthwise on the matrix pipe, one 16-channel group at a time, wave-private ----------------
            // Wave w owns groups 9w .. 9w+8.  Pixel tiles of the expand are two image rows x 8 columns (column 7 and row 7 do not
            // exist: their slots repeat a neighbour and are written as ZEROS -- they are the right / bottom border of the planar
            // image), un-swapped MFMA: lane (n16, q) gets channel n16 of slots 4q .. 4q+3 = columns 4 (q & 1) .. +3 of row
            // 2t + (q >> 1): one 8-byte store into P[channel][row + 2][column].  Depthwise: block = channel, B = the quads of rows
            // y0 + n + ky - R at columns 0 and 4, A = Toeplitz slices of the taps (host-packed, see mid14m_kernel), output tiles at
            // columns -2, 2, 6: four MFMAs per kernel row and 4-row strip.  SiLU, pool sums, and the outputs go to ED[pixel][channel]
            // as dwords of two channels (v_permlane16_swap pairs channels 2k, 2k+1).
            constexpr int KS = decltype(ks_tag)::value, R = KS / 2;
            unsigned char* SG = smem + T7_OFF_STG + wave * (16 * T7_PCS);
            const int n16 = lane & 15;
            const int blk = lane >> 2, n = lane & 3;
            const int c = 2 * (blk & 3) + ((blk >> 2) & 1) + 8 * (blk >> 3);
            const bool oddrow = (blk >> 2) & 1;
            const GLOBAL_AS _Float16* dwt = sgpr_ptr<_Float16>(Wt.dwtoe);
            for (int e = lane; e < T7_PCS; e += 64) *reinterpret_cast<uint4*>(SG + 16 * e) = uint4{0u, 0u, 0u, 0u};   // (gate / part of the previous block)
            h8 xa[4][6];
            {
                const int sx = n16 & 7, sy = n16 >> 3;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int row = 2 * t + sy;
                    const int pix = (row < 7 ? row : 6) * 7 + (sx < 7 ? sx : 6);
#pragma unroll
                    for (int ks = 0; ks < 6; ++ks) xa[t][ks] = *reinterpret_cast<const h8*>(XL + pix * T7_XS + (32 * ks + 8 * q) * 2);
                }
            }
            unsigned char* est = SG + n16 * T7_PCS + (2 + (q >> 1)) * 16 + (q & 1) * 8;       // + t * 32
            const uint32_t mhi = (q & 1) ? 0x0000ffffu : 0xffffffffu;                          // column 7 -> zero
            const uint32_t mrow3 = q >= 2 ? 0u : 0xffffffffu;                                  // tile 3: row 7 -> zeros
            const unsigned char* dld = SG + c * T7_PCS + (n + 2 - R) * 16;                     // + (4 YT + ky) * 16 + 8 * quad
            unsigned char* dummy = SG + c * T7_PCS + 192;                                      // 8 spare bytes per channel: masked stores land here
            // output stores: a lane of an even 16-lane row writes columns 0 .. 3 of its row, of an odd one columns 4 .. 6 (+ a dummy)
            unsigned char* const dst_lane = ED + (n * 7 + (oddrow ? 4 : 0)) * T7_ES + (c >> 1) * 4;
            h8 wg[6];
            u2v ta[KS][2];
            float be, bd;
            auto request_w = [&](int G) {
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) wg[ks] = gload<h8>(W.wexp, (unsigned)(((G * 6 + ks) * 64 + lane) * 16));
                be = gload<float>(W.bexp, (unsigned)(16 * G + n16) * 4u);
            };
            auto request_t = [&](int G) {
#pragma unroll
                for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                    for (int h = 0; h < 2; ++h) ta[ky][h] = gload<u2v>(dwt, (unsigned)((((G * KS + ky) * 2 + h) * 64 + lane) * 8));
                bd = gload<float>(W.bdw, (unsigned)(16 * G + c) * 4u);
            };
            long long gk[4] = {0, 0, 0, 0};   // phase clock of the group loop (MMC_TAIL_CLK=1): expand MFMAs | SiLU + store | depthwise MFMAs | epilogue
            auto group = [&](int G, auto last_tag) __attribute__((always_inline)) {
                constexpr bool LAST = decltype(last_tag)::value;
                long long g0 = 0, g1 = 0, g2 = 0, g3 = 0;
                if (clk_on) g0 = (long long)__builtin_readcyclecounter();
                request_t(G);
                PIN_VMEM();
                // ---- expand ----
                f4 acc[4];
                {
                    const f4 bev = {be, be, be, be};
#pragma unroll
                    for (int ks = 0; ks < 6; ++ks)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[t][ks], wg[ks], ks == 0 ? bev : acc[t], 0, 0, 0);
                }
                if (!LAST) request_w(G + 1);      // the weight registers are free again: next group's fragments arrive during the depthwise part
                else {
                    // last group: the pixel fragments are dead -- the squeeze-excite weights are requested now (see the round-2 phases)
                    const int crl = fc1_thr ? cr : 0;
#pragma unroll
                    for (int i = 0; i < 18; ++i) fw1[i] = gload<u4v>(W.wr_t, (unsigned)((i * 384 + crl * 12 + j4) * 16));
#pragma unroll
                    for (int k = 0; k < 12; ++k) fw2[k] = gload<u4v>(W.we_t, (unsigned)((k * 288 + t2) * 16));
                    brv = tid < 48 ? gload<float>(W.br, (unsigned)tid * 4u) : 0.f;
                }
                PIN_VMEM();
                if (clk_on) g1 = (long long)__builtin_readcyclecounter();
                {
                    float t16[16];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int j = 0; j < 4; ++j) t16[4 * t + j] = acc[t][j];
                    silu_scaled_staged(t16);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        u2v o = {cvt_pk_f16(t16[4 * t], t16[4 * t + 1]), cvt_pk_f16(t16[4 * t + 2], t16[4 * t + 3]) & mhi};
                        if (t == 3) { o.x &= mrow3; o.y &= mrow3; }
                        *reinterpret_cast<u2v*>(est + t * 32) = o;
                    }
                }
                if (clk_on) g2 = (long long)__builtin_readcyclecounter();
                // ---- depthwise (the wave reads back what it wrote: LDS operations of a wave complete in order) ----
                f4 dacc[2][3];
                {
                    // the quads of kernel rows 0 .. KB-1 are requested at once, the rest behind them while the first batch computes (one
                    // kernel row ahead the 8 MFMAs of a row -- 67 cycles -- did not cover an LDS round trip: 1.2 k cycles per group for 40 MFMAs)
                    const f4 bdv = {bd, bd, bd, bd};
                    constexpr int KB = KS == 5 ? 3 : KS;
                    h4 bq[KS][2][2];
                    auto quads = [&](int k0, int k1) {
#pragma unroll
                        for (int ky = k0; ky < k1; ++ky)
#pragma unroll
                            for (int yt = 0; yt < 2; ++yt)
#pragma unroll
                                for (int xq = 0; xq < 2; ++xq) bq[ky][yt][xq] = *reinterpret_cast<const h4*>(dld + (4 * yt + ky) * 16 + 8 * xq);
                    };
                    auto rows = [&](int k0, int k1) {
#pragma unroll
                        for (int ky = k0; ky < k1; ++ky) {
                            const h4 a0 = __builtin_bit_cast(h4, ta[ky][0]), a1 = __builtin_bit_cast(h4, ta[ky][1]);
                            // output tile xt = columns 4 xt - 2 .. 4 xt + 1: quad xt with the h = 1 slice, quad xt - 1 with the h = 0 slice
#pragma unroll
                            for (int yt = 0; yt < 2; ++yt) {
                                dacc[yt][0] = __builtin_amdgcn_mfma_f32_4x4x4f16(a1, bq[ky][yt][0], ky == 0 ? bdv : dacc[yt][0], 0, 0, 0);
                                dacc[yt][1] = __builtin_amdgcn_mfma_f32_4x4x4f16(a1, bq[ky][yt][1], ky == 0 ? bdv : dacc[yt][1], 0, 0, 0);
                                dacc[yt][2] = __builtin_amdgcn_mfma_f32_4x4x4f16(a0, bq[ky][yt][1], ky == 0 ? bdv : dacc[yt][2], 0, 0, 0);
                            }
#pragma unroll
                            for (int yt = 0; yt < 2; ++yt) dacc[yt][1] = __builtin_amdgcn_mfma_f32_4x4x4f16(a0, bq[ky][yt][0], dacc[yt][1], 0, 0, 0);
                        }
                    };
                    quads(0, KB);
                    __builtin_amdgcn_sched_barrier(0);
                    if (KB < KS) quads(KB, KS);
                    __builtin_amdgcn_sched_barrier(0);
                    rows(0, KB);
                    if (KB < KS) rows(KB, KS);
                }
                if (clk_on) g3 = (long long)__builtin_readcyclecounter();
                {
                    float v[14];   // [strip][column]: column x = 4 xt - 2 + i
#pragma unroll
                    for (int yt = 0; yt < 2; ++yt)
#pragma unroll
                        for (int x = 0; x < 7; ++x) v[7 * yt + x] = dacc[yt][(x + 2) >> 2][(x + 2) & 3];
                    silu_scaled_staged(v);
                    const float s0 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + v[6]);
                    const float s1 = ((v[7] + v[8]) + (v[9] + v[10])) + ((v[11] + v[12]) + v[13]);
                    const float psum = quad_sum(s0 + (n < 3 ? s1 : 0.f));   // row 7 does not exist
                    if (n == 0) pooled[16 * G + c] = psum;
                    unsigned char* d0 = dst_lane + 32 * G;
#pragma unroll
                    for (int yt = 0; yt < 2; ++yt) {
                        const bool norow = yt == 1 && n == 3;   // row 7 does not exist: its lanes store to the spare bytes
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            // even rows end up with (own column j, partner's column j), odd rows with (partner's column 4 + j, own column 4 + j)
                            float lo = v[7 * yt + j], hi = j < 3 ? v[7 * yt + 4 + j] : 0.f;
                            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
                            unsigned char* dp = d0 + (28 * yt + j) * T7_ES;
                            if (norow || (j == 3 && oddrow)) dp = dummy;   // (column 7 does not exist either)
                            *reinterpret_cast<uint32_t*>(dp) = cvt_pk_f16(lo, hi);
                        }
                    }
                }
                if (clk_on) {
                    const long long g4 = (long long)__builtin_readcyclecounter();
                    gk[0] += g1 - g0; gk[1] += g2 - g1; gk[2] += g3 - g2; gk[3] += g4 - g3;
                }
            };
            request_w(9 * wave);
#pragma unroll 1
            for (int g = 0; g < 8; ++g) group(9 * wave + g, std::false_type{});
            group(9 * wave + 8, std::true_type{});
            if (clk_on && nb == 0 && lane == 0 && (wave == 0 || wave == 4))   // section 7: the four sums of waves 0 and 4 (SIMD-mates), first block
                for (int i = 0; i < 4; ++i) a.dbg_clk[((size_t)b * 8 + 7) * 8 + (wave ? 4 : 0) + i] = (float)gk[i];
        };
            if (ks3) dw4_phase(std::integral_constant<int, 3>{});
            else dw4_phase(std::integral_constant<int, 5>{});
            T7_BAR();
            T7_TICK();
            T7_TICK();
        } else {
        auto dw_phase = [&](auto ks_tag) __attribute__((always_inline)) {
            constexpr int KS = decltype(ks_tag)::value, R = KS / 2, NP = KS == 5 ? 3 : 2;
            auto tap_pairs = [&](const uint32_t (&raw)[15], uint32_t (&wp)[2 * KS * NP]) {
#pragma unroll
                for (int ky = 0; ky < KS; ++ky) {
                    const uint32_t r0 = raw[3 * ky], r1 = raw[3 * ky + 1], r2 = raw[3 * ky + 2];
                    if (KS == 5) {
                        wp[(ky * 2 + 0) * NP + 0] = r0;
                        wp[(ky * 2 + 0) * NP + 1] = r1;
                        wp[(ky * 2 + 0) * NP + (NP - 1)] = r2;
                        wp[(ky * 2 + 1) * NP + 0] = r0 << 16;
                        wp[(ky * 2 + 1) * NP + 1] = __builtin_amdgcn_alignbit(r1, r0, 16);
                        wp[(ky * 2 + 1) * NP + (NP - 1)] = __builtin_amdgcn_alignbit(r2, r1, 16);
                    } else {
                        wp[(ky * 2 + 0) * NP + 0] = r0 << 16;
                        wp[(ky * 2 + 0) * NP + 1] = __builtin_amdgcn_alignbit(r1, r0, 16);
                        wp[(ky * 2 + 1) * NP + 0] = r0;
                        wp[(ky * 2 + 1) * NP + 1] = r1;
                    }
                }
            };
            // first pixel pair an output column reads
            auto first_pair = [](int ox) { return (KS == 5 || !(ox & 1)) ? (ox >> 1) - 1 : (ox >> 1); };
            // one full round: thread = one expanded channel c, all 49 pixels
            auto dw_round = [&](int c, const uint32_t (&raw)[15], float bias) __attribute__((always_inline)) {
                uint32_t wp[2 * KS * NP];
                tap_pairs(raw, wp);
                unsigned char* col = ED + 2 * c;
                // All 49 two-byte reads go out before the first pair is assembled: left alone the compiler reads two, waits
                // (s_waitcnt lgkmcnt(0)), shifts and ors, 24 times over -- 24 LDS round trips in front of every round.
                uint32_t P[28];
                {
                    uint16_t px[49];
#pragma unroll
                    for (int i = 0; i < 49; ++i) px[i] = *reinterpret_cast<const uint16_t*>(col + i * T7_ES);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int y = 0; y < 7; ++y)
#pragma unroll
                        for (int pp = 0; pp < 4; ++pp)
                            P[y * 4 + pp] = (uint32_t)px[y * 7 + 2 * pp] | (pp < 3 ? (uint32_t)px[y * 7 + 2 * pp + 1] << 16 : 0u);
                }
                f2 psum2 = {0.f, 0.f};
                float psum1 = 0.f;
#pragma unroll
                for (int oy = 0; oy < 7; ++oy) {
                    // the seven outputs of a row advance together (ox innermost): consecutive v_dot2c go to different
                    // accumulators, so no dependent-issue stalls; the first tap of each takes the bias as its addend
                    float acc[7];
                    bool started[7] = {false, false, false, false, false, false, false};
#pragma unroll
                    for (int ky = 0; ky < KS; ++ky) {
                        const int iy = oy - R + ky;
                        if (iy < 0 || iy >= 7) continue;
#pragma unroll
                        for (int ip = 0; ip < NP; ++ip)
#pragma unroll
                            for (int ox = 0; ox < 7; ++ox) {
                                const int xpc = first_pair(ox) + ip;
                                if (xpc < 0 || xpc > 3) continue;
                                if (!started[ox]) { acc[ox] = dot2_from(P[iy * 4 + xpc], wp[(ky * 2 + (ox & 1)) * NP + ip], bias); started[ox] = true; }
                                else acc[ox] = __builtin_amdgcn_fdot2(*reinterpret_cast<const h2*>(&P[iy * 4 + xpc]),
                                                                      *reinterpret_cast<const h2*>(&wp[(ky * 2 + (ox & 1)) * NP + ip]),
                                                                      acc[ox], false);
                            }
                    }
                    DW_SILU(acc);
#pragma unroll
                    for (int ox = 0; ox < 6; ox += 2) {
                        const f2 v = {acc[ox], acc[ox + 1]};
                        psum2 = psum2 + v;
                        const uint32_t hv = cvt_pk_f16(acc[ox], acc[ox + 1]);
                        *reinterpret_cast<uint16_t*>(col + (oy * 7 + ox) * T7_ES) = (uint16_t)hv;
                        *reinterpret_cast<uint16_t*>(col + (oy * 7 + ox + 1) * T7_ES) = (uint16_t)(hv >> 16);
                    }
                    psum1 += acc[6];
                    *reinterpret_cast<_Float16*>(col + (oy * 7 + 6) * T7_ES) = (_Float16)acc[6];
                }
                pooled[c] = (psum2.x + psum2.y) + psum1;
            };
            // The depthwise phase is VALU / LDS work with the vector-memory path idle: everything the next phases stream is
            // requested here, in the order it will be consumed, and PINNED (sched_barrier) -- left alone the scheduler sinks
            // each request to just before its first use, which put the whole squeeze-excite weight fetch (221 KB per
            // block) on the critical path of the two FCs.  Taps of the next round before the current round computes; the
            // squeeze FC weights (36 x 8 bytes per thread) and the first half of the excite FC weights (24 x 8 bytes) before
            // the last quarter round (earlier the two tap buffers leave no registers for them: spills); the second half
            // once FC1 has consumed the squeeze weights.
            load_taps(512 + tid, rawB, biasB);
            dw_round(tid, rawA, biasA);
            const int cl = tid & 127, p4 = tid >> 7, rb = 2 * p4;
            load_taps(1024 + cl, rawA, biasA);
            dw_round(512 + tid, rawB, biasB);
            {
                const int crl = fc1_thr ? cr : 0;   // idle threads re-read row group 0 (no divergent region around the loads)
#pragma unroll
                for (int i = 0; i < 18; ++i) fw1[i] = gload<u4v>(W.wr_t, (unsigned)((i * 384 + crl * 12 + j4) * 16));
            }
#pragma unroll
            for (int k = 0; k < 12; ++k) fw2[k] = gload<u4v>(W.we_t, (unsigned)((k * 288 + t2) * 16));
            brv = tid < 48 ? gload<float>(W.br, (unsigned)tid * 4u) : 0.f;
            PIN_VMEM();
            {
                // Channels 1024..1151 (a quarter round) are shared by FOUR threads each so that all 8 waves stay busy:
                // thread (channel, p) computes output rows 2p and 2p+1 from input rows 2p-R .. 2p+1+R (zeros outside
                // the image).  In place needs every read of a channel before any write: barrier in between.
                constexpr int NR = 2 + 2 * R;
                const int c = 1024 + cl;
                uint32_t wpA[2 * KS * NP];
                tap_pairs(rawA, wpA);
                unsigned char* col = ED + 2 * c;
                uint32_t P[NR * 4];
                {
                    uint16_t px[NR * 7];   // (all reads first: see dw_round)
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const int iy = rb - R + r;
                        const unsigned char* rowp = col + ((iy >= 0 && iy < 7) ? iy : 0) * (7 * T7_ES);
#pragma unroll
                        for (int x = 0; x < 7; ++x) px[r * 7 + x] = *reinterpret_cast<const uint16_t*>(rowp + x * T7_ES);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const int iy = rb - R + r;
                        const bool rok = iy >= 0 && iy < 7;
#pragma unroll
                        for (int pp = 0; pp < 4; ++pp) {
                            const uint32_t v = (uint32_t)px[r * 7 + 2 * pp] | (pp < 3 ? (uint32_t)px[r * 7 + 2 * pp + 1] << 16 : 0u);
                            P[r * 4 + pp] = rok ? v : 0u;
                        }
                    }
                }
                T7_BAR();
                f2 psum2 = {0.f, 0.f};
                float psum1 = 0.f;
#pragma unroll
                for (int ro = 0; ro < 2; ++ro) {
                    const int oy = rb + ro;
                    if (oy < 7) {
                        float acc[7];
                        bool started[7] = {false, false, false, false, false, false, false};
#pragma unroll
                        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                            for (int ip = 0; ip < NP; ++ip)
#pragma unroll
                                for (int ox = 0; ox < 7; ++ox) {
                                    const int xpc = first_pair(ox) + ip;
                                    if (xpc < 0 || xpc > 3) continue;
                                    if (!started[ox]) { acc[ox] = dot2_from(P[(ro + ky) * 4 + xpc], wpA[(ky * 2 + (ox & 1)) * NP + ip], biasA); started[ox] = true; }
                                    else acc[ox] = __builtin_amdgcn_fdot2(*reinterpret_cast<const h2*>(&P[(ro + ky) * 4 + xpc]),
                                                                          *reinterpret_cast<const h2*>(&wpA[(ky * 2 + (ox & 1)) * NP + ip]),
                                                                          acc[ox], false);
                                }
                        DW_SILU(acc);
#pragma unroll
                        for (int ox = 0; ox < 6; ox += 2) {
                            const f2 v = {acc[ox], acc[ox + 1]};
                            psum2 = psum2 + v;
                            const uint32_t hv = cvt_pk_f16(acc[ox], acc[ox + 1]);
                            *reinterpret_cast<uint16_t*>(col + (oy * 7 + ox) * T7_ES) = (uint16_t)hv;
                            *reinterpret_cast<uint16_t*>(col + (oy * 7 + ox + 1) * T7_ES) = (uint16_t)(hv >> 16);
                        }
                        psum1 += acc[6];
                        *reinterpret_cast<_Float16*>(col + (oy * 7 + 6) * T7_ES) = (_Float16)acc[6];
                    }
                }
                part[p4 * 128 + cl] = (psum2.x + psum2.y) + psum1;
                T7_BAR();
                if (tid < 128) pooled[1024 + tid] = ((part[tid] + part[128 + tid]) + part[256 + tid]) + part[384 + tid];
            }
        };
        if (ks3) dw_phase(std::integral_constant<int, 3>{});
        else dw_phase(std::integral_constant<int, 5>{});
        T7_BAR();
        T7_TICK();
        }
        if (a.dbg_dw) {
            _Float16* dg = a.dbg_dw + (size_t)b * T7_PIX * T7_CE;
            for (int e = tid; e < T7_PIX * 144; e += 512) {
                const int pix = e / 144, oc = e - pix * 144;
                *reinterpret_cast<h8*>(dg + pix * T7_CE + oc * 8) = *reinterpret_cast<const h8*>(ED + pix * T7_ES + oc * 16);
            }
        }
        // ---------------- squeeze-excite FC1: r = silu(br + pooled . Wr^T) ----------------
        // One patch per workgroup makes the two FCs matrix-VECTOR products: fp32 FMAs on fp16 weights (fixed
        // summation order).  The excite weights (48 x 8 bytes per thread) are requested before FC1 computes.
        // (The second half of the excite weights goes out HERE, in front of FC1's arithmetic: behind it, the request had only the
        // 48-thread partial reduction to arrive in, and FC2 started with s_waitcnt vmcnt(0).)
#pragma unroll
        for (int k = 12; k < 24; ++k) fw2[k] = gload<u4v>(W.we_t, (unsigned)((k * 288 + t2) * 16));
        const f4 bev = gload<f4>(W.be, (unsigned)t2 * 16u);
        PIN_VMEM();
        if (fc1_thr) {
            // all 36 pool sums of this thread are requested before the first FMA (left alone the compiler reads two, waits, computes
            // eight FMAs, eighteen times over: 18 LDS round trips = 2.5 k of this phase's 3.7 k cycles)
            float xs[36];
#pragma unroll
            for (int i = 0; i < 18; ++i) { xs[2 * i] = pooled[64 * i + cr]; xs[2 * i + 1] = pooled[64 * i + 32 + cr]; }
            __builtin_amdgcn_sched_barrier(0);
            f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                const float x0 = xs[2 * i], x1 = xs[2 * i + 1];
                acc[0] = fma_mix_lo(fw1[i].x, x0, acc[0]);
                acc[1] = fma_mix_hi(fw1[i].x, x0, acc[1]);
                acc[2] = fma_mix_lo(fw1[i].y, x0, acc[2]);
                acc[3] = fma_mix_hi(fw1[i].y, x0, acc[3]);
                acc[0] = fma_mix_lo(fw1[i].z, x1, acc[0]);
                acc[1] = fma_mix_hi(fw1[i].z, x1, acc[1]);
                acc[2] = fma_mix_lo(fw1[i].w, x1, acc[2]);
                acc[3] = fma_mix_hi(fw1[i].w, x1, acc[3]);
            }
            *reinterpret_cast<f4*>(part + cr * 48 + 4 * j4) = acc;
        }
        T7_BAR();
        if (tid < 48) {
            float pv[32];   // all 32 partials requested at once (one LDS latency, not 32), summed in the fixed order
#pragma unroll
            for (int w = 0; w < 32; ++w) pv[w] = part[w * 48 + tid];
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 32; ++w) s += pv[w];
            // the pooled sums are over 49 pixels of log2(e)-scaled activations (kept out of the fp16 weights)
            rs[tid] = silu_f(s * (float)(1.0 / (49.0 * 1.4426950408889634)) + brv);
        }
        T7_BAR();
        T7_TICK();
        // ---------------- FC2: gate = sigmoid(be + r . We^T) ----------------
        if (fc2_thr) {
            f4 rq[12];   // the 48 squeeze outputs (broadcast reads), all requested before the first FMA
#pragma unroll
            for (int k = 0; k < 12; ++k) rq[k] = *reinterpret_cast<const f4*>(rs + 4 * k);
            __builtin_amdgcn_sched_barrier(0);
            f4 acc = bev;
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                const float r0 = rq[k >> 1][2 * (k & 1)], r1 = rq[k >> 1][2 * (k & 1) + 1];
                acc[0] = fma_mix_lo(fw2[k].x, r0, acc[0]);
                acc[1] = fma_mix_hi(fw2[k].x, r0, acc[1]);
                acc[2] = fma_mix_lo(fw2[k].y, r0, acc[2]);
                acc[3] = fma_mix_hi(fw2[k].y, r0, acc[3]);
                acc[0] = fma_mix_lo(fw2[k].z, r1, acc[0]);
                acc[1] = fma_mix_hi(fw2[k].z, r1, acc[1]);
                acc[2] = fma_mix_lo(fw2[k].w, r1, acc[2]);
                acc[3] = fma_mix_hi(fw2[k].w, r1, acc[3]);
            }
            f4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = sigmoid_f(acc[j]);
            *reinterpret_cast<f4*>(gate + 4 * tid) = o;
        }
        // project: bias and the first four k-steps of weight fragments are requested before the gate pass.
        // Output fragments (16 channels) per wave: 192 outputs = 12 fragments -> waves 0..3 take 2, waves 4..7 take 1;
        // 320 outputs (b15) = 20 fragments -> 3 and 2.  Every SIMD hosts one wave of each kind: equal MFMA work.
        const bool wide = Wt.cout == 320;
        const bool lowh = wave < 4;
        const int nfn = wide ? (lowh ? 3 : 2) : (lowh ? 2 : 1);
        const int nf0 = wide ? (lowh ? 3 * wave : 2 * wave + 4) : (lowh ? 2 * wave : wave + 4);
        f4 pbias[3];
        unsigned wo[3];
        h8 wa[3][4];
        proj_prefetch(W.wproj, W.bproj, 36, nf0, nfn, lane, q, pbias, wo, wa);
        T7_BAR();
        T7_TICK();
        if (a.dbg_gate) {
            for (int e = tid; e < T7_CE; e += 512) a.dbg_gate[(size_t)b * T7_CE + e] = gate[e];
        }
        // ---------------- gate, in place ----------------
        // A thread keeps ONE group of 8 channels (its 8 gate values in registers) and walks every third pixel: 144 groups x 3 = 432
        // threads.  (One (pixel, group) element per thread and step re-read the 32 bytes of gate values for every 16 bytes of
        // data: the pass is LDS traffic, and two thirds of it was gate.)
        if (!T7_GATE_IN_WEIGHTS)
        if (tid < 432) {
            const int r3 = tid >= 288 ? 2 : (tid >= 144 ? 1 : 0), oc = tid - 144 * r3;
            const f4 g0 = *reinterpret_cast<const f4*>(gate + oc * 8);
            const f4 g1 = *reinterpret_cast<const f4*>(gate + oc * 8 + 4);
            unsigned char* pc = ED + oc * 16;
            for (int k0 = 0; k0 < 17; k0 += 4) {   // four pixels in flight (gate_h8 is asm: the compiler does not unroll around it)
                uint4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pix = r3 + 3 * (k0 + j);
                    v[j] = *reinterpret_cast<const uint4*>(pc + (pix < T7_PIX ? pix : T7_PIX - 1) * T7_ES);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pix = r3 + 3 * (k0 + j);
                    const uint4 o = gate_h8(v[j], g0, g1);   // fp32 product, one rounding, one VALU op per element
                    if (pix < T7_PIX) *reinterpret_cast<uint4*>(pc + pix * T7_ES) = o;
                }
            }
        }
        if (!T7_GATE_IN_WEIGHTS) T7_BAR();   // (with the gate on the weights nothing happens between FC2's barrier and the project)
        T7_TICK();
        // ---------------- project + bias (+ residual) ----------------
        proj_run(W.wproj, 36, wide ? 1 : 0, nf0, nfn, lane, m, q, pixc, pbias, wo, wa);
        out_wide = wide;
        T7_BAR();
        T7_TICK();
#undef T7_TICK
        if (a.dbg_clk && tid0 == 0) {   // cycles per phase of this block: expand, dw, fc1, fc2, gate, project
            float* dst = a.clk_sections ? a.dbg_clk + ((size_t)b * 8 + 1 + nb) * 8 : a.dbg_clk + (size_t)b * 8;
            for (int i = 0; i < 6; ++i) dst[i] = (float)(tk[i + 1] - tk[i]);
        }
    }
    if (clk_on) tkk[5] = (long long)__builtin_readcyclecounter();
    if (a.head_w) {
        // ---- head: features[n] = mean over pixels of silu(b[n] + Y15[pixel] . Wh[n]) (1280 x 320), Y15 in ED [49][320].
        //      Wave w owns output fragments 10w .. 10w+9 in two groups of five (accumulators 5 x 4 pixel fragments);
        //      pixel fragments come from LDS per k-step, weight fragments stream from L2 one k-step ahead. ----
        const int tid = tid0, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int m = lane & 15, q = lane >> 4;
        const GLOBAL_AS _Float16* hw = sgpr_ptr<_Float16>(a.head_w);
        const GLOBAL_AS float* hb = sgpr_ptr<float>(a.head_b);
        const unsigned char* bxp[4];
#pragma unroll
        for (int pf = 0; pf < 4; ++pf) bxp[pf] = ED + ((16 * pf + m) < T7_PIX ? (16 * pf + m) : (T7_PIX - 1)) * T7_YS + 16 * q;
#pragma unroll 1
        for (int g = 0; g < 2; ++g) {
            const int nfb = 10 * wave + 5 * g;
            f4 acc[5][4];
            // Weight fragments of a k-step live in one of three register sets, requested TWO k-steps ahead in consumption
            // order; the k loop is straight-line (no register rotation) and the requests are pinned where they are written
            // (left alone, the scheduler sinks each load next to its first use: one exposed L2 round trip per k-step).
            h8 w[3][5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const f4 bv = gload<f4>(hb, (unsigned)(16 * (nfb + i) + 4 * q) * 4u);
#pragma unroll
                for (int pf = 0; pf < 4; ++pf) acc[i][pf] = bv;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < 5; ++i) w[s2][i] = gload<h8>(hw, (unsigned)((((nfb + i) * 10 + s2) * 64 + lane) * 16));
#pragma unroll
            for (int ks = 0; ks < 10; ++ks) {
                if (ks + 2 < 10) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) w[(ks + 2) % 3][i] = gload<h8>(hw, (unsigned)((((nfb + i) * 10 + ks + 2) * 64 + lane) * 16));
                }
                PIN_VMEM();
                h8 bx[4];
#pragma unroll
                for (int pf = 0; pf < 4; ++pf) bx[pf] = *reinterpret_cast<const h8*>(bxp[pf] + 64 * ks);
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int pf = 0; pf < 4; ++pf)
                        acc[i][pf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ks % 3][i], bx[pf], acc[i][pf], 0, 0, 0);
                PIN_VMEM();
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                float t[16];
#pragma unroll
                for (int pf = 0; pf < 4; ++pf)
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[4 * pf + j] = acc[i][pf][j];
                silu_scaled_staged(t);
                f4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int pf = 0; pf < 4; ++pf) {
                    const bool ok = 16 * pf + m < T7_PIX;
#pragma unroll
                    for (int j = 0; j < 4; ++j) sum[j] += ok ? t[4 * pf + j] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sum[j] = row16_sum(sum[j]) * a.inv_hw;
                }
                if (m == 0) *reinterpret_cast<f4*>(a.feat + (size_t)b * 1280 + 16 * (nfb + i) + 4 * q) = sum;
            }
        }
        if (clk_on && tid0 == 0) {
            const long long tend = (long long)__builtin_readcyclecounter();
            a.dbg_clk[((size_t)b * 8 + 5) * 8] = (float)(tend - tkk[5]);
            a.dbg_clk[((size_t)b * 8 + 6) * 8] = (float)(tend - tkk[0]);
        }
    } else if (!out_wide) {
        const int tid = tid0;
        _Float16* yg = a.Y + (size_t)b * T7_PIX * T7_C;
        for (int e = tid; e < T7_PIX * 24; e += 512) {
            const int pix = e / 24, p16 = e - pix * 24;
            *reinterpret_cast<h8*>(yg + pix * T7_C + p16 * 8) = *reinterpret_cast<const h8*>(XL + pix * T7_XS + p16 * 16);
        }
    } else {
        const int tid = tid0;
        _Float16* yg = a.Y + (size_t)b * T7_PIX * 320;
        for (int e = tid; e < T7_PIX * 40; e += 512) {
            const int pix = e / 40, p16 = e - pix * 40;
            *reinterpret_cast<h8*>(yg + pix * 320 + p16 * 8) = *reinterpret_cast<const h8*>(ED + pix * T7_YS + p16 * 16);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// mid14_kernel: the front half (expand 1x1 + SiLU, depthwise KSDxKSD + SiLU, pool sums) of a 14x14 MBConv block for ONE
// patch per workgroup (512 threads).  The tile/chunk kernels (mbconv_a/d) split a patch into 10-14 channel-chunk
// workgroups that each re-load the block input and pay their own load/barrier skeleton; here the input X[196][Cin]
// is loaded once into LDS and the workgroup walks the expanded channels in chunks of 96:
//   expand   78 (16-channel x 16-pixel) tiles per chunk, dealt round-robin to the 8 waves; swapped MFMA, weight
//            fragments from L2 one tile ahead, pixel fragments in registers; un-swapped (pixels = rows), so a lane holds 4
//            consecutive pixels of one channel: silu -> two pixel-pair dwords of E2[98 pairs][96 channels] in LDS;
//   dw       thread = (channel, band of 3 output rows): the 7 input rows it needs are 7x7 pixel-pair dwords (one ds_read_b32
//            each) in registers (zero outside the image), taps on v_dot2c exactly as in tail7_kernel, the 14 outputs of a row
//            advance together; silu; fp16 straight to the depthwise output tensor in HBM (lanes = consecutive
//            channels: 128-byte segments); pool sums per band -> LDS -> one value per channel.
// Output: D[B][196][CE] and pool[B][CE] -- what proj_patch_kernel consumes.
// ---------------------------------------------------------------------------------------------
// ST = 2 (block 11: 5x5 stride 2, 14x14 -> 7x7, TF-same pad 1): the same expand; a depthwise thread = (channel, band of 2
// output rows) with the same 7x7 pixel-pair window (input rows 4*band-1 .. 4*band+5); output x reads pairs x-1, x, x+1 with the
// tap pairs (0,k0), (k1,k2), (k3,k4) -- the odd-x variant of the stride-1 taps; four bands, output D[B][49][CE].
template <int CKS, int KSD, int CE, int ST = 1>
__global__ __launch_bounds__(512) void mid14_kernel(Mid14Args a)
{
    static_assert(ST == 1 || (ST == 2 && KSD == 5), "stride 2 is the 5x5 block 11");
    constexpr int HW = 196, CH = 96, NCHK = CE / CH, NPF = 13, NTILE = 6 * NPF;
    constexpr int HWO = ST == 1 ? 196 : 49, NBAND = ST == 1 ? 5 : 4;
    constexpr int ES2 = 416;                // bytes per row of E2[98 pixel pairs][96 channels] (one dword = pixels 2p, 2p+1 of a
                                            // channel); 104 dwords: the four lane quarters of a store land in disjoint banks
    constexpr int R = KSD / 2, NP = KSD == 5 ? 3 : 2;
    static_assert(CE % CH == 0, "chunking");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // Zero rows above and below the image (ZT / ZB input rows of 7 pairs): a depthwise window row outside the image reads zeros
    // instead of being selected to zero register by register (35-49 v_cndmask per thread and chunk).  Written once, below.
    constexpr int ZT = ST == 1 ? R : 1, ZB = ST == 1 ? R + 1 : 4;
    unsigned char* E = smem + ZT * 7 * ES2;                  // pair row 0 of the image
    float* pband = reinterpret_cast<float*>(smem + (98 + 7 * (ZT + ZB)) * ES2);   // [5][96]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int b = blockIdx.x;
    const int Cin = a.Cin;
    const GLOBAL_AS _Float16* wexp = sgpr_ptr<_Float16>(a.wexp);
    const GLOBAL_AS float* bexp = sgpr_ptr<float>(a.bexp);
    const GLOBAL_AS uint32_t* dwp = sgpr_ptr<uint32_t>(a.dwp);
    const GLOBAL_AS float* bdw = sgpr_ptr<float>(a.bdw);
    // Block input: each wave owns pixel fragments (13 fragments of 16 pixels: waves 0..4 own two, waves 5..7 one) and
    // reads them straight into registers, so LDS only holds the expanded chunk and two workgroups fit a CU.
    const int npf = wave < 5 ? 2 : 1;
    const int pf0 = wave < 5 ? 2 * wave : wave + 5;
    const GLOBAL_AS _Float16* xgp = sgpr_ptr<_Float16>(a.X) + (size_t)b * HW * Cin;
    // depthwise role of this thread: channel cd of the chunk, output rows rb .. rb+2 (band 4: rows 12, 13).  Threads past the
    // last band repeat its work (same values to the same addresses): no store sits behind a branch
    const int band0 = tid / CH, cd = tid - band0 * CH;
    const int band = band0 < NBAND ? band0 : NBAND - 1;
    const int rb = 3 * band;
    for (int e = tid; e < (ZT + ZB) * 7 * (ES2 / 16); e += 512) {
        const int row = e / (ES2 / 16), c16 = e - row * (ES2 / 16);
        unsigned char* zr = smem + (row < ZT * 7 ? row : row + 98) * ES2 + 16 * c16;
        *reinterpret_cast<uint4*>(zr) = uint4{0u, 0u, 0u, 0u};
    }
    constexpr bool KEEP_XB = KSD == 5 && ST == 1;
    h8 xb[2][CKS];
    auto load_xb = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pix = 16 * (pf0 + (i < npf ? i : 0)) + m;
            const int pixc = pix < HW ? pix : HW - 1;
#pragma unroll
            for (int ks = 0; ks < CKS; ++ks) {
                const int kk = 32 * ks + 8 * q;
                // zero-padded K columns re-read the last eight channels: finite values against the zero rows the host packs there
                xb[i][ks] = gload<h8>(xgp, (unsigned)((pixc * Cin + (kk < Cin ? kk : Cin - 8)) * 2));
            }
        }
    };
    if (KEEP_XB) load_xb();
#pragma unroll 1
    for (int chunk = blockIdx.y; chunk < NCHK; chunk += gridDim.y) {   // gridDim.y workgroups share a patch's chunks
        // taps and bias of this thread's channel: requested now, used after the expand phase
        uint32_t raw[15];
        const int cg = chunk * CH + cd;
#pragma unroll
        for (int i = 0; i < 15; ++i) raw[i] = 0u;
        float dbias;
        if (KSD == 3) {   // nine dwords + bias: the 3x3 variant sits at exactly 128 registers (two workgroups per CU) and keeps dword loads
#pragma unroll
            for (int i = 0; i < 9; ++i) raw[i] = gload<uint32_t>(dwp, (unsigned)(((i >> 2) * CE + cg) * 4 + (i & 3)) * 4u);
            dbias = gload<float>(bdw, (unsigned)cg * 4u);
        } else {   // taps + bias in four 16-byte requests (layout [4][CE][4]: slots 0..14 = tap pairs, 15 = bias) instead of 16 dword loads
            u4v t4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t4[j] = gload<u4v>(dwp, (unsigned)((j * CE + cg) * 16));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                raw[4 * j] = t4[j].x; raw[4 * j + 1] = t4[j].y; raw[4 * j + 2] = t4[j].z;
                if (j < 3) raw[4 * j + 3] = t4[j].w;
            }
            dbias = __builtin_bit_cast(float, (uint32_t)t4[3].w);
        }
        // ---------------- expand: this wave's pixel fragments x the chunk's six 16-channel weight fragments ----------------
        {
            // 3x3 (128 registers, two workgroups per CU): the pixel fragments are re-read per chunk from L2 -- holding them across the
            // depthwise phase costs 32 registers.  5x5 (one workgroup per CU, registers to spare): read once, before the chunk loop
            // (a timing-only build without these loads ran the 5x5 variants 7-10 % faster).
            if (!KEEP_XB) load_xb();
            // Weight fragments and bias of the next 16-channel fragment are requested one fragment ahead, bias first and pinned
            // (see tail7_kernel's block 11): the wave's role is a template argument and lanes past the last pixel store to a
            // scratch word, so the six fragments of a chunk are one straight-line block.
            h8 wn[CKS];
            float bsn = gload<float>(bexp, (unsigned)(16 * (6 * chunk) + m) * 4u);
#pragma unroll
            for (int ks = 0; ks < CKS; ++ks) wn[ks] = gload<h8>(wexp, (unsigned)((((6 * chunk) * CKS + ks) * 64 + lane) * 16));
            unsigned char* const scratch = reinterpret_cast<unsigned char*>(pband + NBAND * CH) + lane * 4;
            auto expand_chunk = [&](auto npf_tag) {
                constexpr int NPFW = decltype(npf_tag)::value;
#pragma unroll
                for (int nf = 0; nf < 6; ++nf) {
                    const int nfg = 6 * chunk + nf;
                    h8 wc[CKS];
#pragma unroll
                    for (int ks = 0; ks < CKS; ++ks) wc[ks] = wn[ks];
                    const float bs = bsn;
                    if (nf + 1 < 6) {
                        bsn = gload<float>(bexp, (unsigned)(16 * (nfg + 1) + m) * 4u);
#pragma unroll
                        for (int ks = 0; ks < CKS; ++ks) wn[ks] = gload<h8>(wexp, (unsigned)((((nfg + 1) * CKS + ks) * 64 + lane) * 16));
                    }
                    PIN_VMEM();
                    // un-swapped MFMA (pixels = rows, channels = columns): lane (m, q) gets channel 16 nf + m of pixels
                    // 16 pf + 4q .. +3 = two ready-made pixel pairs (same dot products, same k order as the swapped form)
                    const f4 bv = {bs, bs, bs, bs};   // the first MFMA reads it as its addend: no copy per accumulator
                    f4 acc[NPFW];
#pragma unroll
                    for (int i = 0; i < NPFW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xb[i][0], wc[0], bv, 0, 0, 0);
#pragma unroll
                    for (int ks = 1; ks < CKS; ++ks)
#pragma unroll
                        for (int i = 0; i < NPFW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xb[i][ks], wc[ks], acc[i], 0, 0, 0);
                    float t[4 * NPFW];
#pragma unroll
                    for (int i = 0; i < NPFW; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) t[4 * i + j] = acc[i][j];
                    silu_scaled_staged(t);
#pragma unroll
                    for (int i = 0; i < NPFW; ++i) {
                        const int pix0 = 16 * (pf0 + i) + 4 * q;
                        const h2 p0 = {(_Float16)t[4 * i], (_Float16)t[4 * i + 1]};
                        const h2 p1 = {(_Float16)t[4 * i + 2], (_Float16)t[4 * i + 3]};
                        const bool ok = pix0 < HW;
                        unsigned char* dst = E + (pix0 >> 1) * ES2 + (16 * nf + m) * 4;
                        *reinterpret_cast<h2*>(ok ? dst : scratch) = p0;
                        *reinterpret_cast<h2*>(ok ? dst + ES2 : scratch + 256) = p1;
                    }
                }
            };
            if (npf == 2) expand_chunk(std::integral_constant<int, 2>{});
            else expand_chunk(std::integral_constant<int, 1>{});
        }
        T7_BAR();
        // ---------------- depthwise ----------------
        {
            constexpr int NR = ST == 1 ? 3 + 2 * R : 7;   // input rows of a band (3 output rows; stride 2: 2 output rows)
            const unsigned char* col = E + 4 * cd + ((ST == 1 ? rb - R : 4 * band - 1) * 7) * ES2;   // (may start in the zero rows)
            uint32_t P[NR][7];
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int pp = 0; pp < 7; ++pp) P[r][pp] = *reinterpret_cast<const uint32_t*>(col + (r * 7 + pp) * ES2);
            f2 psum2 = {0.f, 0.f};
            float psum1 = 0.f;
            uint16_t* dg = reinterpret_cast<uint16_t*>(a.D + (size_t)b * HWO * CE + chunk * CH + cd);
            if (ST == 2) {
#pragma unroll
                for (int ro = 0; ro < 2; ++ro) {
                    const int oy = 2 * band + ro;
                    if (oy < 7) {
                        float acc[7];
                        bool started[7] = {false, false, false, false, false, false, false};
#pragma unroll
                        for (int ky = 0; ky < 5; ++ky) {
                            const uint32_t r0 = raw[3 * ky], r1 = raw[3 * ky + 1], r2 = raw[3 * ky + 2];
                            const uint32_t wq[3] = {r0 << 16, __builtin_amdgcn_alignbit(r1, r0, 16), __builtin_amdgcn_alignbit(r2, r1, 16)};
#pragma unroll
                            for (int ip = 0; ip < 3; ++ip)
#pragma unroll
                                for (int ox = 0; ox < 7; ++ox) {
                                    const int xpc = ox - 1 + ip;
                                    if (xpc < 0 || xpc > 6) continue;
                                    if (!started[ox]) { acc[ox] = dot2_from(P[2 * ro + ky][xpc], wq[ip], dbias); started[ox] = true; }
                                    else acc[ox] = __builtin_amdgcn_fdot2(*reinterpret_cast<const h2*>(&P[2 * ro + ky][xpc]),
                                                                          *reinterpret_cast<const h2*>(&wq[ip]), acc[ox], false);
                                }
                        }
                        silu_scaled_staged(acc);
#pragma unroll
                        for (int ox = 0; ox < 6; ox += 2) {
                            const f2 v = {acc[ox], acc[ox + 1]};
                            psum2 = psum2 + v;
                            const uint32_t hv = cvt_pk_f16(acc[ox], acc[ox + 1]);
                            dg[(size_t)(oy * 7 + ox) * CE] = (uint16_t)hv;
                            dg[(size_t)(oy * 7 + ox + 1) * CE] = (uint16_t)(hv >> 16);
                        }
                        psum1 += acc[6];
                        reinterpret_cast<_Float16*>(dg)[(size_t)(oy * 7 + 6) * CE] = (_Float16)acc[6];
                    }
                }
            } else
#pragma unroll
            for (int ro = 0; ro < 3; ++ro) {
                const int oy = rb + ro;
                if (oy < 14) {
                    float acc[14];
                    bool started[14] = {false, false, false, false, false, false, false, false, false, false, false, false, false, false};
#pragma unroll
                    for (int ky = 0; ky < KSD; ++ky) {
                        // tap pairs of this kernel row: [parity of x][pair]; the shifted variants are derived here (per
                        // row) instead of kept for the whole chunk -- 15 registers that decide between 3 and 4 waves/SIMD
                        const uint32_t r0 = raw[3 * ky], r1 = raw[3 * ky + 1], r2 = raw[3 * ky + 2];
                        uint32_t wq[2][3];
                        if (KSD == 5) {
                            wq[0][0] = r0; wq[0][1] = r1; wq[0][2] = r2;
                            wq[1][0] = r0 << 16; wq[1][1] = __builtin_amdgcn_alignbit(r1, r0, 16); wq[1][2] = __builtin_amdgcn_alignbit(r2, r1, 16);
                        } else {
                            wq[0][0] = r0 << 16; wq[0][1] = __builtin_amdgcn_alignbit(r1, r0, 16); wq[0][2] = 0u;
                            wq[1][0] = r0; wq[1][1] = r1; wq[1][2] = 0u;
                        }
#pragma unroll
                        for (int ip = 0; ip < NP; ++ip)
#pragma unroll
                            for (int ox = 0; ox < 14; ++ox) {
                                const int fp = (KSD == 5 || !(ox & 1)) ? (ox >> 1) - 1 : (ox >> 1);
                                const int xpc = fp + ip;
                                if (xpc < 0 || xpc > 6) continue;
                                if (!started[ox]) { acc[ox] = dot2_from(P[ro + ky][xpc], wq[ox & 1][ip], dbias); started[ox] = true; }
                                else acc[ox] = __builtin_amdgcn_fdot2(*reinterpret_cast<const h2*>(&P[ro + ky][xpc]),
                                                                      *reinterpret_cast<const h2*>(&wq[ox & 1][ip]), acc[ox], false);
                            }
                    }
                    silu_scaled_staged(acc);
#pragma unroll
                    for (int ox = 0; ox < 14; ox += 2) {
                        const f2 v = {acc[ox], acc[ox + 1]};
                        psum2 = psum2 + v;
                        const uint32_t hv = cvt_pk_f16(acc[ox], acc[ox + 1]);
                        dg[(size_t)(oy * 14 + ox) * CE] = (uint16_t)hv;
                        dg[(size_t)(oy * 14 + ox + 1) * CE] = (uint16_t)(hv >> 16);
                    }
                }
            }
            pband[band * CH + cd] = (psum2.x + psum2.y) + psum1;
        }
        T7_BAR();
        if (tid < CH)
            a.pool[(size_t)b * CE + chunk * CH + tid] =
                ((pband[tid] + pband[CH + tid]) + pband[2 * CH + tid]) + pband[3 * CH + tid] + (NBAND == 5 ? pband[4 * CH + tid] : 0.f);
        // (the next chunk's expand writes E only after every wave passed the barrier above; pband is rewritten only
        // after the next chunk's first barrier)
    }
}

// ---------------------------------------------------------------------------------------------
// mid14m_kernel: the front half of a 14x14 MBConv block with the DEPTHWISE CONV ON THE MATRIX PIPE (round 3: 4x4x4 form).
//
// The fused expand + depthwise kernels are bound by vector-instruction issue (per output a 5x5 depthwise costs 15 v_dot2c + 4
// SiLU instructions + conversion / store / bookkeeping) while the matrix pipe idles at 2-9 %.  Round 2 put the depthwise conv
// on v_mfma_f32_16x16x32_f16 with a block-diagonal weight matrix: 1/16 of every MFMA useful, 13 LDS fragments of 1 KB per output
// row -- equal in speed.  v_mfma_f32_4x4x4_16B_f16 multiplies 16 INDEPENDENT 4x4x4 blocks, so a block can be a channel with its
// own weights (tools/ubench/mfma4x4.hip: layout confirmed, 8.4 cycles per instruction, one vector instruction of a SIMD-mate wave
// rides along):
//     block = channel c;  D[i][j] = out[c][y0 + j][x0 + i]   (4 output columns x 4 output rows)
//     B[k][j] = in[c][y0 + j + ky - R][xq + k]               (a lane's four k values are 8 contiguous bytes of a planar row)
//     A[i][k] = w[c][ky][xq + k - (x0 + i) + R]              (a Toeplitz slice of kernel row ky; zero outside 0 .. KS-1)
// with the input quads xq = x0 - 2 and x0 + 2: two MFMAs per (kernel row, 4x4 output tile), 20 of their 32 products per output
// useful (5x5).  A 16-channel group's 196 outputs take 4 x 4 tiles x KS rows x 2 = 160 MFMAs (144: the last column tile's
// second quad lies in the zero border) = 1.2 k matrix-pipe cycles against ~3.7 k vector-issue cycles of v_dot2c, read 80 x 512 B
// of LDS (block-diagonal form: 182 x 1 KB), and the vector port keeps SiLU + pack only.  The Toeplitz fragments (A) are packed on
// the host (`dwtoe`: 8 bytes per lane, kernel row and quad), loaded once per channel group.
//
// Work split (as in round 2): a WAVE owns whole 16-channel groups -- it expands its group for all 196 pixels (un-swapped MFMA:
// a lane gets 4 consecutive pixels of one channel) into its PRIVATE planar region E[16 channels][18 rows][20 columns] (two zero
// border rows / columns on every side: nothing is predicated), runs the depthwise MFMAs over it, applies SiLU, transposes a
// 4-row strip through a private [56 pixels][16 channels] tile and stores it to D with 16-byte lanes, pool sums in registers.
// No data crosses waves after the block input has been staged: ONE barrier per kernel.
// Pixel tiles of the expand are image ROWS (14 pixels + 2 repeats): a lane's LDS addresses are base + row * immediate.
// Output: D[B][196][CE], pool[B][CE] -- what mid14_kernel writes (sums in another, equally fixed, order).
// ---------------------------------------------------------------------------------------------
template <int CKS, int KSD, int CE>
__global__ __launch_bounds__(512) void mid14m_kernel(Mid14Args a)
{
    constexpr int HW = 196, NG = CE / 16, R = KSD / 2;
    constexpr int XSTR = 64 * CKS + 16;          // bytes per staged block-input row (k zero padded to 32 CKS)
    constexpr int ERS = 40;                      // bytes per planar row: columns x = 0 .. 15 (quads at 0, 4, 8, 12; 14, 15 stay zero) + 8 spare
    constexpr int ECS = 736;                     // bytes per channel: 18 rows (y = -2 .. 15) + 16: the 16 x 4 lanes of a ds_read_b64 hit 64 banks
    constexpr int EREG = 16 * ECS;               // a wave's planar region
    constexpr int TREG = 56 * 32;                // a wave's transpose tile: [4 rows x 14 pixels][16 channels] fp16
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* XS = smem;                                   // [196][XSTR]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* EW = smem + HW * XSTR + wave * (EREG + TREG);   // this wave's planar region
    unsigned char* TW = EW + EREG;                                 // ... and its transpose tile
    const int n16 = lane & 15, q = lane >> 4;    // expand role: channel n16 of the group, pixels 4q .. 4q+3 of an image row
    // depthwise role: MFMA block lane >> 2 = channel c of the group, output row y0 + n (B / D operand), tap row i = n (A operand).
    // Blocks map to channels so that channels 2k and 2k+1 sit in 16-lane rows r and r+1 at the same position: one
    // v_permlane16_swap pairs their values for the transposing store (two channels of a pixel = one dword).
    const int blk = lane >> 2, n = lane & 3;
    const int c = 2 * (blk & 3) + ((blk >> 2) & 1) + 8 * (blk >> 3);
    const int b = blockIdx.x;
    const int Cin = a.Cin;
    const GLOBAL_AS _Float16* wexp = sgpr_ptr<_Float16>(a.wexp);
    const GLOBAL_AS float* bexp = sgpr_ptr<float>(a.bexp);
    const GLOBAL_AS _Float16* dwt = sgpr_ptr<_Float16>(a.dwdiag);
    const GLOBAL_AS float* bdw = sgpr_ptr<float>(a.bdw);
    const bool clk = a.dbg_clk != nullptr;
    long long ck[5] = {clk ? (long long)__builtin_readcyclecounter() : 0, 0, 0, 0, 0};
    // ---- stage the block input (k zero padded) and zero this wave's planar region (borders stay zero for the whole kernel) ----
    {
        const _Float16* xg = a.X + (size_t)b * HW * Cin;
        constexpr int CPR = 4 * CKS;             // 16-byte chunks per staged row
        // all of a thread's chunks are requested before the first is stored (unconditional loads from clamped addresses: the loop used to
        // be load -> vmcnt(0) -> store, six exposed round trips -- the 6 k cycles of "staging" in the first phase clock)
        constexpr int NIT = (HW * CPR + 511) / 512;
        uint4 xv[NIT];
        for (int e = lane; e < EREG / 16; e += 64) *reinterpret_cast<uint4*>(EW + 16 * e) = uint4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e0 = tid + 512 * it, e = e0 < HW * CPR ? e0 : HW * CPR - 1;
            const int row = e / CPR, cc = e - row * CPR;
            xv[it] = *reinterpret_cast<const uint4*>(xg + (size_t)row * Cin + (8 * cc < Cin ? 8 * cc : Cin - 8));
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + 512 * it;
            const int row = e / CPR, cc = e - row * CPR;
            const uint4 v = 8 * cc < Cin ? xv[it] : uint4{0u, 0u, 0u, 0u};   // zero-padded k columns
            if (e < HW * CPR) *reinterpret_cast<uint4*>(XS + row * XSTR + 16 * cc) = v;
        }
    }
    __syncthreads();
    if (clk) ck[1] = (long long)__builtin_readcyclecounter();
    // lane constants
    const unsigned char* xrow = XS + (n16 < 14 ? n16 : 13) * XSTR + 16 * q;              // A operand of the expand: pixel (y, min(m, 13)), + y * 14 * XSTR
    unsigned char* est = EW + n16 * ECS + 2 * ERS + 8 * q;                                // expand store: pixels x = 4q .. 4q+3 of row y (one 8-byte store), + y * ERS
    const uint32_t m3 = q == 3 ? 0u : 0xffffffffu;                                        // columns 14, 15 do not exist: they are written as zeros (right border)
    const unsigned char* dld = EW + c * ECS + n * ERS + (2 - R) * ERS;                    // B operand rows y0 + n + ky - R (+2 border), + (4 yt + ky) * ERS
    const unsigned char* dld3 = EW + c * ECS + (n < 2 ? n : 1) * ERS + (2 - R) * ERS;     // last strip (rows 12, 13): lanes n >= 2 repeat row 13
    // transpose tile store: after the lane swap a lane of an even 16-lane row holds channels (c, c+1) of pixels x = 0 .. 6 of its
    // row, a lane of an odd row channels (c-1, c) of pixels x = 7 .. 13: one dword per pixel at [pixel (n, x)][channel pair]
    unsigned char* tst = TW + (n * 14 + (((blk >> 2) & 1) ? 7 : 0)) * 32 + (c >> 1) * 4;
    // operands of a channel group: expand weights (B operand, CKS fragments) + bias, Toeplitz depthwise fragments + bias.  The next
    // group's are requested while the current group computes (an exposed L2 round trip otherwise).
    h8 we[CKS], wen[CKS];
    u2v ta[KSD][2], tan[KSD][2];
    float be, bd, ben, bdn;
    auto request_group = [&](int g, h8 (&w1)[CKS], u2v (&w2)[KSD][2], float& b1, float& b2) {
#pragma unroll
        for (int ks = 0; ks < CKS; ++ks) w1[ks] = gload<h8>(wexp, (unsigned)(((g * CKS + ks) * 64 + lane) * 16));
        b1 = gload<float>(bexp, (unsigned)(16 * g + n16) * 4u);
#pragma unroll
        for (int ky = 0; ky < KSD; ++ky)
#pragma unroll
            for (int h = 0; h < 2; ++h) w2[ky][h] = gload<u2v>(dwt, (unsigned)((((g * KSD + ky) * 2 + h) * 64 + lane) * 8));
        b2 = gload<float>(bdw, (unsigned)(16 * g + c) * 4u);
    };
    {
        const int g0 = blockIdx.y * 8 + wave;
        request_group(g0 < NG ? g0 : 0, wen, tan, ben, bdn);
    }
#pragma unroll 1
    for (int g = blockIdx.y * 8 + wave; g < NG; g += 8 * gridDim.y) {
#pragma unroll
        for (int ks = 0; ks < CKS; ++ks) we[ks] = wen[ks];
#pragma unroll
        for (int ky = 0; ky < KSD; ++ky) { ta[ky][0] = tan[ky][0]; ta[ky][1] = tan[ky][1]; }
        be = ben;
        bd = bdn;
        {
            const int gn = g + 8 * (int)gridDim.y;
            request_group(gn < NG ? gn : g, wen, tan, ben, bdn);   // (the last group re-requests itself: unused)
        }
        PIN_VMEM();
        // ---------------- expand: E[n16][y][x] = silu(X[y][x] . W_g[n16] + b) for the 14 image rows ----------------
        // Software pipeline inside the wave (it has ONE partner on its SIMD): the A fragments of the next two rows are requested
        // before the current rows' MFMAs, and the MFMAs of step i are interleaved with the SiLU epilogue of step i - 1.
        auto x_frags = [&](int y, h8 (&xb)[2][CKS]) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int ks = 0; ks < CKS; ++ks) xb[u][ks] = *reinterpret_cast<const h8*>(xrow + (y + u) * 14 * XSTR + 64 * ks);
        };
        auto x_epilogue = [&](int y, const f4 (&acc)[2]) {   // SiLU + store of image rows y, y + 1
            float t[8] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[1][0], acc[1][1], acc[1][2], acc[1][3]};
            silu_scaled_staged(t);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const u2v o = {cvt_pk_f16(t[4 * u], t[4 * u + 1]), cvt_pk_f16(t[4 * u + 2], t[4 * u + 3]) & m3};
                *reinterpret_cast<u2v*>(est + (y + u) * ERS) = o;
            }
        };
        {
            // (requests pinned with sched_barrier: left alone, the scheduler sinks every ds_read next to its MFMA and waits for it
            // there -- one exposed LDS round trip per MFMA, 6.5-11 k cycles per group instead of ~2 k)
            const f4 bev = {be, be, be, be};
            h8 xf[2][2][CKS];       // two register sets: rows of the current step, rows of the next
            f4 accp[2] = {bev, bev};
            x_frags(0, xf[0]);
#pragma unroll
            for (int st = 0; st < 7; ++st) {          // image rows 2 st, 2 st + 1
                if (st + 1 < 7) x_frags(2 * (st + 1), xf[(st + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                f4 acc[2];
#pragma unroll
                for (int ks = 0; ks < CKS; ++ks)
#pragma unroll
                    for (int u = 0; u < 2; ++u)   // un-swapped: rows = pixels, columns = channels; the first k-step takes the bias as its addend
                        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[st & 1][u][ks], we[ks], ks == 0 ? bev : acc[u], 0, 0, 0);
                if (st > 0) x_epilogue(2 * (st - 1), accp);
                if (st > 0) {
#pragma unroll
                    for (int i = 0; i < 2 * CKS; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 2; ++u) accp[u] = acc[u];
            }
            x_epilogue(12, accp);
        }
        if (clk && ck[2] == 0) ck[2] = (long long)__builtin_readcyclecounter();
        // ---------------- depthwise on the matrix pipe (4x4x4 blocks = channels), SiLU, transpose, store, pool sums ----------------
        // One strip of four output rows per step: KSD x 4 quads of B (ds_read_b64), KSD x 7 MFMAs into four 4x4 accumulators; the
        // previous strip's epilogue (SiLU, pack, transpose tile, 16-byte stores) is issued beside the current strip's MFMAs.
        float psum = 0.f;
        unsigned char* dgb = reinterpret_cast<unsigned char*>(a.D + (size_t)b * HW * CE + 16 * g);
        auto d_quads = [&](auto yt_tag, h4 (&bq)[KSD][4]) {   // the KSD x 4 input quads of a strip: rows y0 + n + ky - R, columns 0, 4, 8, 12
            constexpr int YT = decltype(yt_tag)::value;
            const unsigned char* rb = YT == 3 ? dld3 : dld;
#pragma unroll
            for (int ky = 0; ky < KSD; ++ky)
#pragma unroll
                for (int x4 = 0; x4 < 4; ++x4) bq[ky][x4] = *reinterpret_cast<const h4*>(rb + (4 * YT + ky) * ERS + 8 * x4);
        };
        auto d_strip = [&](const h4 (&bq)[KSD][4], f4 (&acc)[4]) {
            const f4 bdv = {bd, bd, bd, bd};
#pragma unroll
            for (int ky = 0; ky < KSD; ++ky) {
                const h4 a0 = __builtin_bit_cast(h4, ta[ky][0]), a1 = __builtin_bit_cast(h4, ta[ky][1]);
                // output tile xt = columns 4 xt - 2 .. 4 xt + 1: quad xt with the h = 1 slice, quad xt - 1 with the h = 0 slice (tile 0's
                // left quad is the zero border: no MFMA).  Four independent accumulators back to back, then the second round.
#pragma unroll
                for (int xt = 0; xt < 4; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_4x4x4f16(a1, bq[ky][xt], ky == 0 ? bdv : acc[xt], 0, 0, 0);
#pragma unroll
                for (int xt = 1; xt < 4; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_4x4x4f16(a0, bq[ky][xt - 1], acc[xt], 0, 0, 0);
            }
        };
        auto d_epilogue = [&](auto yt_tag, const f4 (&acc)[4]) {
            constexpr int YT = decltype(yt_tag)::value;
            float v[14];
#pragma unroll
            for (int x = 0; x < 14; ++x) v[x] = acc[(x + 2) >> 2][(x + 2) & 3];
            silu_scaled_staged(v);
            float s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])) + ((v[8] + v[9]) + (v[10] + v[11])) + (v[12] + v[13]);
            if (YT == 3) s = n < 2 ? s : 0.f;   // rows 14, 15 do not exist
            psum += s;
#ifdef MID14M_NOSWAP
#pragma unroll
            for (int x = 0; x < 14; x += 2) {
                const uint32_t hv = cvt_pk_f16(v[x], v[x + 1]);
                *reinterpret_cast<uint16_t*>(TW + (n * 14 + x) * 32 + c * 2) = (uint16_t)hv;
                *reinterpret_cast<uint16_t*>(TW + (n * 14 + x + 1) * 32 + c * 2) = (uint16_t)(hv >> 16);
            }
#else
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                // even 16-lane rows end up with (own v[j], partner's v[j]), odd rows with (partner's v[7 + j], own v[7 + j])
                // (inline asm: hipcc 7.2 reads BOTH results of __builtin_amdgcn_permlane16_swap from the first register -- seen in the
                // ISA as v_cvt_pk_f16_f32 v26, v27, v27; the s_nop pads are the wait states the compiler cannot see around asm)
                float lo = v[j], hi = v[7 + j];
                asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
                *reinterpret_cast<uint32_t*>(tst + j * 32) = cvt_pk_f16(lo, hi);
            }
#endif
            // the tile [56 pixels][32 B] leaves as 112 sixteen-byte vectors: lane L takes vectors L and L + 64 (the last strip has 56)
            // (lanes past the last vector repeat it: no store behind a branch -- a conditional store makes the compiler wait for
            // vmcnt(0), i.e. for the stores themselves, at the next use of a prefetched operand)
            constexpr int NV = YT == 3 ? 56 : 112;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (64 * r >= NV) break;
                const int v0 = lane + 64 * r, vi = v0 < NV ? v0 : NV - 1;
                const uint4 o = *reinterpret_cast<const uint4*>(TW + vi * 16);
                *reinterpret_cast<uint4*>(dgb + (size_t)(56 * YT + (vi >> 1)) * (CE * 2) + 16 * (vi & 1)) = o;
            }
        };
        {
            // strip yt's quads are requested one strip ahead (pinned), its MFMAs run beside the previous strip's epilogue
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
            h4 bqa[KSD][4], bqb[KSD][4];
            f4 acc0[4], acc1[4];
            d_quads(I0{}, bqa);
            d_quads(I1{}, bqb);
            __builtin_amdgcn_sched_barrier(0);
            d_strip(bqa, acc0);
            __builtin_amdgcn_sched_barrier(0);
            d_quads(I2{}, bqa);
            __builtin_amdgcn_sched_barrier(0);
            d_strip(bqb, acc1);
            d_epilogue(I0{}, acc0);
            __builtin_amdgcn_sched_barrier(0);
            d_quads(I3{}, bqb);
            __builtin_amdgcn_sched_barrier(0);
            d_strip(bqa, acc0);
            d_epilogue(I1{}, acc1);
            __builtin_amdgcn_sched_barrier(0);
            d_strip(bqb, acc1);
            d_epilogue(I2{}, acc0);
            __builtin_amdgcn_sched_barrier(0);
            d_epilogue(I3{}, acc1);
        }
        // pool sum of channel c: the four lanes n = 0 .. 3 hold its four row residues
        psum = quad_sum(psum);
        if (n == 0) a.pool[(size_t)b * CE + 16 * g + c] = psum;
        if (clk && ck[3] == 0) ck[3] = (long long)__builtin_readcyclecounter();
    }
    if (clk && lane == 0 && (wave == 0 || wave == 4 || wave == 7)) {   // staging | first group: expand | depthwise | all remaining groups
        ck[4] = (long long)__builtin_readcyclecounter();
        float* dst = a.dbg_clk + ((size_t)b * 8 + blockIdx.y) * 16 + (wave == 0 ? 0 : (wave == 4 ? 4 : 8));
        dst[0] = (float)(ck[1] - ck[0]); dst[1] = (float)(ck[2] - ck[1]); dst[2] = (float)(ck[3] - ck[2]); dst[3] = (float)(ck[4] - ck[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// mb1_kernel: block 1 (112x112 -> 56x56: block 0's SE scale + project 32 -> 16, expand 16 -> 96, depthwise 3x3 stride 2)
// in the thread = channel x window-in-registers style of mid14_kernel.  One workgroup (512 threads) = (patch, output
// tile of 8 rows x 28 columns, chunk of 32 expanded channels); its input window is 17 x 57 positions.
//   expand   wave w owns the 16-position fragments w, w+8, ...: block 0's depthwise output is read straight into
//            registers (all loads first: one round trip), scaled by block 0's gate, projected by ONE MFMA (K-permuted
//            expand weights, as in mbconv_a_kernel PRE), expanded by two un-swapped MFMAs (a lane gets 4 consecutive
//            positions of one channel), silu -> pixel-pair dwords E2[17 x 29 pairs][32] in LDS; positions outside the image
//            are written as zeros (padding lives in the expanded domain).
//   dw       thread = (channel, output row, half of the 28 columns): 3 input rows x 15 pixel pairs (one ds_read_b32 each) in registers,
//            two v_dot2c per kernel row and output ((k0,k1) on pair j, (k2,0) on pair j+1), silu, fp16 to HBM,
//            pool sums through LDS -> pool[patch][tile][96].
// ---------------------------------------------------------------------------------------------
// PLANAR: the depthwise output goes to D as three planes [chunk][B * 56 * 56][32] instead of [B][56][56][96].  A workgroup
// produces its three 32-channel chunks ~25 us apart; interleaved, the 64 bytes it writes per pixel and chunk are a third of a
// 192-byte pixel, the 128-byte lines stay partial until another chunk (long evicted) completes them, and WRITE_SIZE was 1.7 x the
// tensor (131 MB vs 77 MB per 128 patches).  In a plane the two 64-byte halves of a line are consecutive outputs of one thread.
// thin_proj_kernel reads the planes (a k-step of its MFMA is exactly one plane).
template <bool PLANAR>
__global__ __launch_bounds__(512, 4) void mb1_kernel(Mb1Args a)   // 128 VGPRs: two workgroups per CU (one: 105 vs 94 us)
{
    // The window is enumerated with 58 columns (29 pixel pairs; the 58th column is one more real pixel, or zero past the image):
    // a 4-position group of the un-swapped expand MFMA is two whole pairs, stored as E2[17 x 29 pairs][32 channels] dwords.
    constexpr int WC = 58, NPOS = 17 * WC, NPF = (NPOS + 15) / 16, ES2 = 160;   // 986 positions, 62 fragments, bytes per pair row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // (40 dwords: the lane quarters of a store hit disjoint banks)
    unsigned char* E = smem;
    float* pred = reinterpret_cast<float*>(smem + NPF * 8 * ES2);    // [16][32] pool partials
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int tile = blockIdx.x, b = blockIdx.z;
    const int ty = tile >> 1, tx = tile & 1;
    const int oy0 = 8 * ty, ox0 = 28 * tx, iy0 = 16 * ty, ix0 = 56 * tx;
    const GLOBAL_AS _Float16* xg = sgpr_ptr<_Float16>(a.X) + (size_t)b * 112 * 112 * 32;
    // ---------------- block 0's gate + project, ONCE per tile: the projected fragments (4 fp16 per lane and fragment) stay in
    // registers and feed the expand of all three 32-channel chunks.  (One workgroup per (tile, chunk) read block 0's depthwise output
    // three times -- 320 MB fetched per 128 patches against 103 MB -- and redid the gate + project MFMA per chunk.) ----------------
    // Positions outside the image (row 112 / column 112: TF-same pads bottom and right only) are NOT zeroed in E: they hold the
    // expand of some in-image pixel, and the only depthwise taps that read them -- kernel row 2 of output row 55, kernel column 2
    // of output column 55 -- get zero weights in the threads that own those outputs (five selects per thread and chunk instead of
    // a select per stored pair and the bookkeeping of which pairs lie outside).
    uint2 xbp[8];     // block 0's output fragment (k = 4q .. 4q+3 of 16), rounded as the separate path stores it
    {
        u4v xr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pf = wave + 8 * i;
            const int p = 16 * pf + m;
            const int r = p / WC, c = p - r * WC;
            const int iy = iy0 + r, ix = ix0 + c;
            const bool ok = pf < NPF && p < NPOS && iy < 112 && ix < 112;
            xr[i] = gload<u4v>(xg, (unsigned)((((ok ? iy : 0) * 112 + (ok ? ix : 0)) * 32 + 8 * q) * 2));
        }
        // block 0's squeeze-excite gate goes into the project's weight fragment (lane quarter q holds input channels 8q .. 8q+7 of
        // both operands), once per tile, instead of into every pixel fragment (8 conversions + products each)
        const uint4 wraw = *reinterpret_cast<const uint4*>(a.pre_w + lane * 8);
        const f4 bpre = *reinterpret_cast<const f4*>(a.pre_b + 4 * q);
        const f4 g0 = *reinterpret_cast<const f4*>(a.pre_gate + (size_t)b * 32 + 8 * q);
        const f4 g1 = *reinterpret_cast<const f4*>(a.pre_gate + (size_t)b * 32 + 8 * q + 4);
        const uint4 wgated = gate_h8(wraw, g0, g1);
        const h8 wpre = *reinterpret_cast<const h8*>(&wgated);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f4 x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wpre, *reinterpret_cast<const h8*>(&xr[i]), bpre, 0, 0, 0);
            h4 xh;
#pragma unroll
            for (int j = 0; j < 4; ++j) xh[j] = (_Float16)x1[j];
            xbp[i] = *reinterpret_cast<const uint2*>(&xh);
        }
    }
    // operands of a chunk: two expand weight fragments + biases, this thread's nine depthwise taps + bias; the next chunk's are
    // requested in front of the current chunk's depthwise phase
    h4 wexp[2];
    float bexp[2];
    float kdw[9], dbias;
    auto request_chunk = [&](int chunk) {
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
            wexp[nf] = *reinterpret_cast<const h4*>(a.wexp + ((size_t)(chunk * 32 + 16 * nf + m) * 32 + 8 * q));   // slots 8q .. 8q+3 = channels 4q .. 4q+3
            bexp[nf] = a.bexp[chunk * 32 + 16 * nf + m];
        }
    };
    auto request_taps = [&](int chunk) {
        const int cg_dw = chunk * 32 + (tid & 31);
#pragma unroll
        for (int i = 0; i < 9; ++i) kdw[i] = a.wdw[(size_t)i * 96 + cg_dw];
        dbias = a.bdw[cg_dw];
    };
    request_chunk(0);
    request_taps(0);
#pragma unroll 1
    for (int chunk = 0; chunk < 3; ++chunk) {
        // ---------------- expand ----------------
        f4 bexp4[2];   // the bias as a ready accumulator operand, built once per chunk
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) bexp4[nf] = f4{bexp[nf], bexp[nf], bexp[nf], bexp[nf]};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (wave + 8 * i >= NPF) continue;   // wave-uniform
            // K = 16 MFMA: block 0's 16 output channels are exactly one k-step of it, and the projected fragment (k = 4q .. 4q+3 in
            // lane quarter q) is its A operand as it stands -- no zero-padded upper half to assemble per fragment and chunk
            const h4 xb = *reinterpret_cast<const h4*>(&xbp[i]);
            float t[8];   // both output fragments' accumulators: SiLU staged over all eight
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) {
                // un-swapped: lane (m, q) = channel 16 nf + m of positions 16 pf + 4q .. +3 = two pixel pairs
                const f4 acc = __builtin_amdgcn_mfma_f32_16x16x16f16(xb, wexp[nf], bexp4[nf], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) t[4 * nf + j] = acc[j];
            }
            silu_scaled_staged(t);
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) {
                const h2 p0 = {(_Float16)t[4 * nf], (_Float16)t[4 * nf + 1]};
                const h2 p1 = {(_Float16)t[4 * nf + 2], (_Float16)t[4 * nf + 3]};
                unsigned char* dst = E + (8 * (wave + 8 * i) + 2 * q) * ES2 + (16 * nf + m) * 4;   // first output pair: (16 pf + 4 q) / 2
                *reinterpret_cast<h2*>(dst) = p0;
                *reinterpret_cast<h2*>(dst + ES2) = p1;
            }
        }
        if (chunk + 1 < 3) request_chunk(chunk + 1);   // lands during the depthwise phase
        T7_BAR();
        // ---------------- depthwise 3x3 stride 2 ----------------
        {
            const int c = tid & 31, orow = (tid >> 5) & 7, half = tid >> 8;
            const int cg = chunk * 32 + c;
            uint32_t wq[3][2], wql[3];   // wql: the (k2, 0) pair as output column 13 of the half sees it
            const bool last_row = oy0 + orow == 55, last_col = tx == 1 && half == 1;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float k0 = kdw[ky * 3 + 0], k1 = kdw[ky * 3 + 1], k2 = kdw[ky * 3 + 2];
                h2 w0 = {(_Float16)k0, (_Float16)k1}, w1 = {(_Float16)k2, (_Float16)0.0f};
                wq[ky][0] = *reinterpret_cast<uint32_t*>(&w0);
                wq[ky][1] = *reinterpret_cast<uint32_t*>(&w1);
                if (ky == 2) { wq[ky][0] = last_row ? 0u : wq[ky][0]; wq[ky][1] = last_row ? 0u : wq[ky][1]; }
                wql[ky] = last_col ? 0u : wq[ky][1];
            }
            const float dbias_c = dbias;
            if (chunk + 1 < 3) request_taps(chunk + 1);
            const unsigned char* col = E + 4 * c;
            uint32_t P[3][15];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const unsigned char* rowp = col + ((2 * orow + ky) * 29 + 14 * half) * ES2;
#pragma unroll
                for (int pp = 0; pp < 15; ++pp) P[ky][pp] = *reinterpret_cast<const uint32_t*>(rowp + pp * ES2);
            }
            float acc[14];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int ip = 0; ip < 2; ++ip)
#pragma unroll
                    for (int j = 0; j < 14; ++j) {
                        const uint32_t wt = (ip == 1 && j == 13) ? wql[ky] : wq[ky][ip];
                        if (ky == 0 && ip == 0) acc[j] = dot2_from(P[ky][j + ip], wt, dbias_c);
                        else acc[j] = __builtin_amdgcn_fdot2(*reinterpret_cast<const h2*>(&P[ky][j + ip]), *reinterpret_cast<const h2*>(&wt),
                                                             acc[j], false);
                    }
            f2 psum2 = {0.f, 0.f};
            constexpr int PS = PLANAR ? 32 : 96;   // elements between consecutive pixels
            const size_t pix = ((size_t)b * 56 + oy0 + orow) * 56 + ox0 + 14 * half;
            uint16_t* dg = reinterpret_cast<uint16_t*>(PLANAR ? a.D + ((size_t)chunk * a.B * 3136 + pix) * 32 + c : a.D + pix * 96 + cg);
            silu_scaled_staged(acc);
#pragma unroll
            for (int j = 0; j < 14; j += 2) {
                const f2 v = {acc[j], acc[j + 1]};
                psum2 = psum2 + v;
                const uint32_t hv = cvt_pk_f16(acc[j], acc[j + 1]);
                dg[(size_t)j * PS] = (uint16_t)hv;
                dg[(size_t)(j + 1) * PS] = (uint16_t)(hv >> 16);
            }
            pred[(tid >> 5) * 32 + c] = psum2.x + psum2.y;
        }
        T7_BAR();   // E and pred are free again behind this barrier (the pool sums below only read pred, rewritten two barriers on)
        if (tid < 32) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) s += pred[w * 32 + tid];
            a.pool[((size_t)b * 14 + tile) * 96 + chunk * 32 + tid] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// mbt_kernel: front half (expand + depthwise stride 1 + pool sums) of the 56x56 and 28x28 blocks (b2, b4) in the
// window-in-registers style of mid14_kernel, with spatial tiling.  One workgroup (512 threads) = (patch, output tile of
// 14 rows x 28 columns, chunk of 48 expanded channels).  The tile's input window (14 + 2R rows, 28 or 30 columns, even
// aligned) is expanded into E[window position][48] in LDS -- positions outside the image as zeros -- from pixel
// fragments read straight into registers; then thread = (channel, band of 3 output rows, half of the 28 columns) holds
// its (3 + 2R) x 9 pixel-pair window in registers and runs the taps on v_dot2c as tail7/mid14 do.
// Template: KSD depthwise size, CKS k-steps of the block input, CE expanded channels, HIMG image size.
// ---------------------------------------------------------------------------------------------
template <int KSD, int CKS, int CE, int HIMG>
__global__ __launch_bounds__(512) void mbt_kernel(MbtArgs a)
{
    constexpr int R = KSD / 2, NP = KSD == 5 ? 3 : 2, NROWS = 14 + 2 * R, CH = 48;
    constexpr int ES2 = 224;                              // bytes per row of E2[window pixel pairs][48 channels] (56 dwords: the
                                                          // four lane quarters of a store hit disjoint banks)
    constexpr int WW = HIMG == 28 ? 28 : 30;              // window columns (even aligned)
    constexpr int NPOS = NROWS * WW, NPF = (NPOS + 15) / 16;
    static_assert(NPF <= 32, "four fragments per wave");
    constexpr int TX = HIMG / 28, NR = 3 + 2 * R;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* E = smem;                                            // [NPF*8][ES2]
    float* pred = reinterpret_cast<float*>(smem + NPF * 8 * ES2);       // [10][48]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int tile = blockIdx.x, chunk = blockIdx.y, b = blockIdx.z;
    const int ty = tile / TX, tx = tile - ty * TX;
    const int oy0 = 14 * ty, ox0 = 28 * tx;
    const int wx0 = ox0 >= 2 ? ox0 - 2 : 0;               // even
    const int Cin = a.Cin;
    const GLOBAL_AS _Float16* xg = sgpr_ptr<_Float16>(a.X) + (size_t)b * HIMG * HIMG * Cin;
    const GLOBAL_AS _Float16* wexp = sgpr_ptr<_Float16>(a.wexp);
    uint32_t raw[15];
    float dbias;
    const int cg_dw = chunk * CH + tid % CH;
    // ---------------- expand ----------------
    {
        u4v xr[4][CKS];
        // validity of this lane's two output pairs (positions 16 pf + 4q .. +3): window and image.  The window's columns always lie
        // inside the image (wx0 .. wx0 + WW - 1 <= HIMG - 1), only its first R / last R rows can fall outside (top / bottom tiles).
        // 3x3: those rows are NOT zeroed in E -- the one tap that reads them (kernel row 0 of output row 0 at the top, kernel row 2
        // of output row 13 at the bottom) gets zero weights in the depthwise threads that own those rows: 8 selects per thread instead
        // of 6 per stored fragment plus the bookkeeping of which pairs lie outside.  5x5 has six such (row, kernel row) pairs: it keeps E zeroed.
        constexpr bool MASK_E = KSD != 3;
        bool okp[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = 16 * (wave + 8 * i) + m;
            const int r = p / WW, c = p - r * WW;
            const int iy = oy0 - R + r, ix = wx0 + c;
            const bool ok = p < NPOS && iy >= 0 && iy < HIMG && ix < HIMG;
            const int row = ok ? iy * HIMG + ix : 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pp = 16 * (wave + 8 * i) + 4 * q + 2 * h;      // even: the pair lies in one window row, and is in or out
                const int pr = pp / WW, pc = pp - pr * WW;               // of the image as a whole (wx0 and HIMG are even)
                const int py = oy0 - R + pr;
                okp[i][h] = !MASK_E || (pp < NPOS && py >= 0 && py < HIMG && wx0 + pc < HIMG);
            }
#pragma unroll
            for (int ks = 0; ks < CKS; ++ks) {
                const int kk = 32 * ks + 8 * q;
                xr[i][ks] = gload<u4v>(xg, (unsigned)((row * Cin + (kk < Cin ? kk : Cin - 8)) * 2));
            }
        }
        h8 wa[3][CKS];
        float ba[3];
#pragma unroll
        for (int nf = 0; nf < 3; ++nf) {
            const int nfg = 3 * chunk + nf;
#pragma unroll
            for (int ks = 0; ks < CKS; ++ks) wa[nf][ks] = gload<h8>(wexp, (unsigned)(((nfg * CKS + ks) * 64 + lane) * 16));
            ba[nf] = a.bexp[16 * nfg + m];
        }
        // depthwise taps and bias of this thread's channel: requested behind the expand's operands (in-order return: they
        // delay nothing) and pinned here, so their round trip is hidden by the expand instead of opening the depthwise phase
#pragma unroll
        for (int i = 0; i < 3 * KSD; ++i) raw[i] = a.dwp[(size_t)i * CE + cg_dw];
        dbias = a.bdw[cg_dw];
        PIN_VMEM();
        f4 ba4[3];   // the bias as a ready accumulator operand (built once: per fragment it was three v_mov per 16 channels)
#pragma unroll
        for (int nf = 0; nf < 3; ++nf) ba4[nf] = f4{ba[nf], ba[nf], ba[nf], ba[nf]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (wave + 8 * i >= NPF) continue;   // wave-uniform
            h8 xb[CKS];
#pragma unroll
            for (int ks = 0; ks < CKS; ++ks) {
                xb[ks] = *reinterpret_cast<const h8*>(&xr[i][ks]);   // (zero-padded K columns: finite re-reads x zero weight rows)
            }
            const int pair0 = 8 * (wave + 8 * i) + 2 * q;
            float t[12];   // the three output fragments' accumulators: SiLU staged over all twelve
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) {
                // un-swapped: lane (m, q) = channel 16 nf + m of positions 16 pf + 4q .. +3 = two pixel pairs
                f4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xb[0], wa[nf][0], ba4[nf], 0, 0, 0);
#pragma unroll
                for (int ks = 1; ks < CKS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xb[ks], wa[nf][ks], acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) t[4 * nf + j] = acc[j];
            }
            silu_scaled_staged(t);
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) {
                const h2 p0 = {(_Float16)t[4 * nf], (_Float16)t[4 * nf + 1]};
                const h2 p1 = {(_Float16)t[4 * nf + 2], (_Float16)t[4 * nf + 3]};
                const h2 z = {(_Float16)0.0f, (_Float16)0.0f};
                unsigned char* dst = E + pair0 * ES2 + (16 * nf + m) * 4;
                *reinterpret_cast<h2*>(dst) = okp[i][0] ? p0 : z;
                *reinterpret_cast<h2*>(dst + ES2) = okp[i][1] ? p1 : z;
            }
        }
    }
    T7_BAR();
    // ---------------- depthwise ----------------
    {
        // rest: 0..9 for the 480 working threads; threads 480..511 repeat rest 9's work (same values to the same addresses), so
        // that no store sits behind a branch: 42 exec-mask branches per thread kept the compiler from staging anything
        const int c = tid % CH, rest = tid < 10 * CH ? tid / CH : 9;
        const int band = rest % 5, half = rest / 5;
        const int rb = 3 * band;
        const int cg = chunk * CH + c;
        // window pair columns of this half: local pair l (0..8) <-> window pair (cbase/2 - 1 + l); cbase is even.  Only l = 0
        // (left image border) and l = 8 (right border) can fall outside the window, and then the whole pair is zero padding
        const int cbase = ox0 + 14 * half - wx0;
        const int pb = (cbase >> 1) - 1;
        const bool lok = pb >= 0, rok = pb + 8 < WW / 2;
        const unsigned char* col = E + 4 * c + (lok ? pb : 0) * ES2;
        const int o0 = lok ? 0 : -ES2;                      // byte offset of pair l relative to col: o0 + l * ES2 (l >= 1)
        uint32_t P[NR][9];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int wr = rb + r < NROWS ? rb + r : NROWS - 1;     // (band 4 has two output rows: its last window row is unused)
            const unsigned char* rowp = col + (wr * (WW / 2)) * ES2;
            const uint32_t v0 = *reinterpret_cast<const uint32_t*>(rowp);
            P[r][0] = lok ? v0 : 0u;
#pragma unroll
            for (int l = 1; l < 8; ++l) P[r][l] = *reinterpret_cast<const uint32_t*>(rowp + o0 + l * ES2);
            const uint32_t v8 = *reinterpret_cast<const uint32_t*>(rowp + o0 + (rok ? 8 : 7) * ES2);
            P[r][8] = rok ? v8 : 0u;
        }
        f2 psum2 = {0.f, 0.f};
        _Float16* dg = a.D + (((size_t)b * HIMG + oy0 + rb) * HIMG + ox0 + 14 * half) * CE + cg;
#pragma unroll
        for (int ro = 0; ro < 3; ++ro) {
            if (rb + ro < 14) {
                float acc[14];
#pragma unroll
                for (int ky = 0; ky < KSD; ++ky) {
                    const uint32_t r0 = raw[3 * ky], r1 = raw[3 * ky + 1], r2 = raw[3 * ky + 2];
                    uint32_t wq[2][3];
                    if (KSD == 5) {
                        wq[0][0] = r0; wq[0][1] = r1; wq[0][2] = r2;
                        wq[1][0] = r0 << 16; wq[1][1] = __builtin_amdgcn_alignbit(r1, r0, 16); wq[1][2] = __builtin_amdgcn_alignbit(r2, r1, 16);
                    } else {
                        wq[0][0] = r0 << 16; wq[0][1] = __builtin_amdgcn_alignbit(r1, r0, 16); wq[0][2] = 0u;
                        wq[1][0] = r0; wq[1][1] = r1; wq[1][2] = 0u;
                        // window row ro + ky of this band lies outside the image: top tile, output row 0, kernel row 0; bottom tile,
                        // output row 13 (band 4, ro = 1), kernel row 2
                        if ((ro == 0 && ky == 0) || (ro == 1 && ky == 2)) {
                            const bool out = ro == 0 ? (oy0 + rb == 0) : (oy0 + rb + 1 == HIMG - 1);
                            wq[0][0] = out ? 0u : wq[0][0]; wq[0][1] = out ? 0u : wq[0][1];
                            wq[1][0] = out ? 0u : wq[1][0]; wq[1][1] = out ? 0u : wq[1][1];
                        }
                    }
#pragma unroll
                    for (int ip = 0; ip < NP; ++ip)
#pragma unroll
                        for (int j = 0; j < 14; ++j) {
                            // local pair of output column j: k5 -> (j>>1) + ip; k3 -> even j: (j>>1) + ip, odd j: (j>>1) + 1 + ip
                            const int l = (KSD == 5 || !(j & 1)) ? (j >> 1) + ip : (j >> 1) + 1 + ip;
                            if (ky == 0 && ip == 0) acc[j] = dot2_from(P[ro + ky][l], wq[j & 1][ip], dbias);
                            else acc[j] = __builtin_amdgcn_fdot2(*reinterpret_cast<const h2*>(&P[ro + ky][l]),
                                                                 *reinterpret_cast<const h2*>(&wq[j & 1][ip]), acc[j], false);
                        }
                }
                silu_scaled_staged(acc);
#pragma unroll
                for (int j = 0; j < 14; j += 2) {
                    const f2 v = {acc[j], acc[j + 1]};
                    psum2 = psum2 + v;
                    const uint32_t hv = cvt_pk_f16(acc[j], acc[j + 1]);     // one conversion per two outputs, low / high half stores
                    reinterpret_cast<uint16_t*>(dg)[((size_t)ro * HIMG + j) * CE] = (uint16_t)hv;
                    reinterpret_cast<uint16_t*>(dg)[((size_t)ro * HIMG + j + 1) * CE] = (uint16_t)(hv >> 16);
                }
            }
        }
        pred[rest * CH + c] = psum2.x + psum2.y;
    }
    T7_BAR();
    if (tid < CH) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 10; ++w) s += pred[w * CH + tid];
        a.pool[((size_t)b * gridDim.x + tile) * CE + chunk * CH + tid] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// mbt4_kernel: mbt_kernel's 5x5 stride-1 layer at 28x28 (b4; B4's b7..b9) with the DEPTHWISE CONV ON THE MATRIX PIPE
// (v_mfma_f32_4x4x4_16B_f16, block = channel: see mid14m_kernel).  Same workgroup = (patch, 14 x 28 output tile, chunk of 48
// channels), same expand (window of 18 x 28 positions, pixel fragments straight into registers, un-swapped MFMA: a lane gets
// four consecutive positions of one channel = ONE aligned quad of a window row, 28 being a multiple of 4) -- but the quad goes
// to a PLANAR image E[48 channels][18 rows][28 columns] (64-byte rows, 1160 bytes per channel: the 8-byte stores of 16 channels
// and the 8-byte reads of 8 channels x 4 rows are both conflict-free), rows outside the image as zeros, and the depthwise phase is
// 24 items = 3 channel groups x 4 strips of four output rows x 2 column halves, three per wave: 5 kernel rows x (4 quad reads +
// 7 MFMAs: output tiles at columns -2, 2, 6, 10 of the half, the zero border quad skipped) -> SiLU -> v_permlane16_swap ->
// wave-private [56 pixels][16 channels] tile -> 16-byte stores.  The 630 v_dot2c of a depthwise thread become 35 MFMAs per item.
// Template: CKS k-steps of the block input, CE expanded channels.
// ---------------------------------------------------------------------------------------------
template <int CKS, int CE>
__global__ __launch_bounds__(512, 4) void mbt4_kernel(MbtArgs a)
{
    constexpr int HIMG = 28, R = 2, KSD = 5, NROWS = 18, WW = 28, CH = 48, NPOS = NROWS * WW, NPF = 32;
    constexpr int ERS = 64, ECS = 1160;            // planar row / channel stride (18 x 64 + 8: the 8 spare bytes take masked stores)
    constexpr int TREG = 56 * 32;                  // a wave's transpose tile [4 rows x 14 pixels][16 channels]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* E = smem;                                            // [48][ECS]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* TW = smem + CH * ECS + wave * TREG;
    float* pred = reinterpret_cast<float*>(smem + CH * ECS + 8 * TREG);   // [8 regions][48]
    const int m = lane & 15, q = lane >> 4;
    const int tile = blockIdx.x, chunk = blockIdx.y, b = blockIdx.z;
    const int oy0 = 14 * tile;
    const int Cin = a.Cin;
    const GLOBAL_AS _Float16* xg = sgpr_ptr<_Float16>(a.X) + (size_t)b * HIMG * HIMG * Cin;
    const GLOBAL_AS _Float16* wexp = sgpr_ptr<_Float16>(a.wexp);
    const GLOBAL_AS _Float16* dwt = sgpr_ptr<_Float16>(a.dwtoe);
    // ---------------- expand ----------------
    {
        u4v xr[4][CKS];
        int eoff[4];      // byte offset of this lane's quad inside a channel's planar image (or the spare bytes)
        bool okq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = 16 * (wave + 8 * i) + m;
            const int r = (p * 2341) >> 16, c = p - r * WW;              // p / 28 (exact below 896)
            const int iy = oy0 - R + r;
            const bool ok = p < NPOS && iy >= 0 && iy < HIMG;
            const int row = ok ? iy * HIMG + c : 0;
            const int pq = 16 * (wave + 8 * i) + 4 * q;                  // this lane's output quad: positions pq .. pq+3 of one window row
            const int qr = (pq * 2341) >> 16, qc = pq - qr * WW;
            const int qy = oy0 - R + qr;
            okq[i] = pq < NPOS && qy >= 0 && qy < HIMG;
            eoff[i] = pq < NPOS ? qr * ERS + qc * 2 : NROWS * ERS;
#pragma unroll
            for (int ks = 0; ks < CKS; ++ks) {
                const int kk = 32 * ks + 8 * q;
                xr[i][ks] = gload<u4v>(xg, (unsigned)((row * Cin + (kk < Cin ? kk : Cin - 8)) * 2));
            }
        }
        h8 wa[3][CKS];
        float ba[3];
#pragma unroll
        for (int nf = 0; nf < 3; ++nf) {
            const int nfg = 3 * chunk + nf;
#pragma unroll
            for (int ks = 0; ks < CKS; ++ks) wa[nf][ks] = gload<h8>(wexp, (unsigned)(((nfg * CKS + ks) * 64 + lane) * 16));
            ba[nf] = a.bexp[16 * nfg + m];
        }
        PIN_VMEM();
        f4 ba4[3];
#pragma unroll
        for (int nf = 0; nf < 3; ++nf) ba4[nf] = f4{ba[nf], ba[nf], ba[nf], ba[nf]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h8 xb[CKS];
#pragma unroll
            for (int ks = 0; ks < CKS; ++ks) xb[ks] = *reinterpret_cast<const h8*>(&xr[i][ks]);
            float t[12];
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) {
                f4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xb[0], wa[nf][0], ba4[nf], 0, 0, 0);
#pragma unroll
                for (int ks = 1; ks < CKS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xb[ks], wa[nf][ks], acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) t[4 * nf + j] = acc[j];
            }
            silu_scaled_staged(t);
            const uint32_t mk = okq[i] ? 0xffffffffu : 0u;   // rows outside the image are the depthwise conv's zero padding
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) {
                const u2v o = {cvt_pk_f16(t[4 * nf], t[4 * nf + 1]) & mk, cvt_pk_f16(t[4 * nf + 2], t[4 * nf + 3]) & mk};
                *reinterpret_cast<u2v*>(E + (16 * nf + m) * ECS + eoff[i]) = o;
            }
        }
    }
    T7_BAR();
    // ---------------- depthwise on 4x4x4 MFMA blocks ----------------
    {
        const int blk = lane >> 2, n = lane & 3;
        const int c = 2 * (blk & 3) + ((blk >> 2) & 1) + 8 * (blk >> 3);     // channel of the group this lane's block holds (mid14m_kernel)
        const bool oddrow = (blk >> 2) & 1;
        const int yt = wave & 3, xh = wave >> 2;                               // this wave's region: output rows 4 yt .. +3, columns 14 xh .. +13
        const int nn = (yt == 3 && n >= 2) ? 1 : n;                           // strip 3 has rows 12, 13 only: lanes n >= 2 repeat row 13
        const bool rowok = !(yt == 3 && n >= 2);
        const unsigned char* dld = E + c * ECS + (4 * yt + nn) * ERS + 24 * xh;   // + 16 g * ECS + ky * ERS + 8 * local quad (quads 0..3 / 3..6)
        unsigned char* tst = TW + (n * 14 + (oddrow ? 7 : 0)) * 32 + (c >> 1) * 4;
        // the tile's 112 (strip 3: 56) sixteen-byte vectors leave through lanes L and L + 64: pixel tp = v >> 1 -> row tp / 14, column tp % 14
        // (lanes past the last vector repeat it -- same bytes to the same address: a store behind a branch makes the compiler wait for
        // vmcnt(0), i.e. for the previous item's stores to reach memory, before the next item's first MFMA)
        const int nv = yt == 3 ? 56 : 112;
        unsigned goff[2], toff[2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int v0 = lane + 64 * rr, vi = v0 < nv ? v0 : nv - 1, tp = vi >> 1;
            const int tr = (tp * 147) >> 11, tx = tp - 14 * tr;
            toff[rr] = (unsigned)(vi * 16);
            goff[rr] = (unsigned)((((oy0 + 4 * yt + tr) * HIMG + 14 * xh + tx) * CE + chunk * CH) * 2 + 16 * (vi & 1));
        }
        unsigned char* dgb = reinterpret_cast<unsigned char*>(a.D + (size_t)b * HIMG * HIMG * CE);
        u2v ta[KSD][2], tan[KSD][2];
        float bd, bdn;
        auto request_t = [&](int g, u2v (&w2)[KSD][2], float& b2) {
            const int G = 3 * chunk + g;
#pragma unroll
            for (int ky = 0; ky < KSD; ++ky)
#pragma unroll
                for (int h = 0; h < 2; ++h) w2[ky][h] = gload<u2v>(dwt, (unsigned)((((G * KSD + ky) * 2 + h) * 64 + lane) * 8));
            b2 = a.bdw[16 * G + c];
        };
        request_t(0, tan, bdn);
        auto item = [&](int g, auto xh_tag) __attribute__((always_inline)) {
            constexpr int XH = decltype(xh_tag)::value;
#pragma unroll
            for (int ky = 0; ky < KSD; ++ky) { ta[ky][0] = tan[ky][0]; ta[ky][1] = tan[ky][1]; }
            bd = bdn;
            request_t(g + 1 < 3 ? g + 1 : g, tan, bdn);
            PIN_VMEM();
            const f4 bdv = {bd, bd, bd, bd};
            f4 acc[4];
            const unsigned char* rb = dld + 16 * g * ECS;
            // Quads two kernel rows ahead of their MFMAs (rows 0, 1 before the first MFMA, row ky + 2 into the slot row ky has just freed),
            // pinned: left alone the compiler sinks every read next to its MFMAs -- a read, s_waitcnt lgkmcnt(1), three MFMAs, twenty times
            // per item.  (All five rows up front: 13 registers spilled at the 128 this kernel may use; three ahead: 3.)
            h4 ql[2][4];
            auto quads = [&](int ky) {
#pragma unroll
                for (int k = 0; k < 4; ++k) ql[ky & 1][k] = *reinterpret_cast<const h4*>(rb + ky * ERS + 8 * k);
            };
            quads(0); quads(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ky = 0; ky < KSD; ++ky) {
                const h4 a0 = __builtin_bit_cast(h4, ta[ky][0]), a1 = __builtin_bit_cast(h4, ta[ky][1]);
                const h4 (&qk)[4] = ql[ky & 1];
                if (XH == 0) {   // tiles at columns -2, 2, 6, 10: quad t with the h = 1 slice, quad t - 1 with the h = 0 slice (tile 0's left quad is the zero border)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_4x4x4f16(a1, qk[t], ky == 0 ? bdv : acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 1; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_4x4x4f16(a0, qk[t - 1], acc[t], 0, 0, 0);
                } else {         // tiles at columns 14, 18, 22, 26: local quad t (= quad 3 + t) with h = 0, local quad t + 1 with h = 1 (tile 3's right quad is the zero border)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_4x4x4f16(a0, qk[t], ky == 0 ? bdv : acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_4x4x4f16(a1, qk[t + 1], acc[t], 0, 0, 0);
                }
                if (ky + 2 < KSD) {
                    __builtin_amdgcn_sched_barrier(0);
                    quads(ky + 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            float v[14];
#pragma unroll
            for (int x = 0; x < 14; ++x) v[x] = XH == 0 ? acc[(x + 2) >> 2][(x + 2) & 3] : acc[x >> 2][x & 3];
            silu_scaled_staged(v);
            float s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])) + ((v[8] + v[9]) + (v[10] + v[11])) + (v[12] + v[13]);
            s = quad_sum(rowok ? s : 0.f);
            if (n == 0) pred[(2 * yt + xh) * CH + 16 * g + c] = s;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                float lo = v[j], hi = v[7 + j];
                asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
                *reinterpret_cast<uint32_t*>(tst + j * 32) = cvt_pk_f16(lo, hi);
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const uint4 o = *reinterpret_cast<const uint4*>(TW + toff[rr]);
                *reinterpret_cast<uint4*>(dgb + goff[rr] + 32 * g) = o;
            }
        };
        if (xh == 0) {
#pragma unroll
            for (int g = 0; g < 3; ++g) item(g, std::integral_constant<int, 0>{});
        } else {
#pragma unroll
            for (int g = 0; g < 3; ++g) item(g, std::integral_constant<int, 1>{});
        }
    }
    T7_BAR();
    if (tid < CH) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += pred[w * CH + tid];
        a.pool[((size_t)b * gridDim.x + tile) * CE + chunk * CH + tid] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// mbt2_kernel: the same recipe for the STRIDE-2 blocks at 56x56 and 28x28 (b3: 5x5, 24 -> 144; b5: 3x3, 40 -> 240).
// One workgroup (512 threads) = (patch, output tile of 7 rows x 14 columns, chunk of 48 expanded channels).  The tile's
// input window (12 + KSD rows, 32 or 30 columns starting on an even column) is expanded un-swapped into pixel-pair dwords
// E2[window pairs][48] in LDS (zeros outside the image); a depthwise thread = (channel, output row) holds its KSD rows x
// 16 pixel pairs in registers and runs v_dot2c: output x reads the pairs pbase + x + ip with the tap pairs (0,k0), (k1,k2),
// (k3,k4) for 5x5 (TF-same pad 1) and (k0,k1), (k2,0) for 3x3 (pad 0).  Output D[B][H/2][H/2][CE], pool[B][tiles][CE].
// ---------------------------------------------------------------------------------------------
template <int KSD, int CKS, int CE, int HIMG>
__global__ __launch_bounds__(512) void mbt2_kernel(MbtArgs a)
{
    constexpr int PADB = KSD == 5 ? 1 : 0, NIP = KSD == 5 ? 3 : 2, NROWS = 12 + KSD, CH = 48, ES2 = 224;
    constexpr int WW = KSD == 5 ? 32 : 30, NPOS = NROWS * WW, NPF = (NPOS + 15) / 16, NS = (NPF + 7) / 8;
    constexpr int HOUT = HIMG / 2, TX = HOUT / 14, NPR = KSD == 5 ? 16 : 15;   // pixel pairs per window row a thread needs
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* E = smem;                                            // [NS*8*8][ES2]
    float* pred = reinterpret_cast<float*>(smem + NS * 64 * ES2);       // [14][48]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int tile = blockIdx.x, chunk = blockIdx.y, b = blockIdx.z;
    const int ty = tile / TX, tx = tile - ty * TX;
    const int oy0 = 7 * ty, ox0 = 14 * tx;
    const int wy0 = 2 * oy0 - PADB;                                     // may be -1: that row is padding
    int wx0 = (2 * ox0 - PADB) & ~1;
    wx0 = wx0 < 0 ? 0 : wx0;
    const int pbase = (2 * ox0 - PADB - wx0 - (KSD == 5 ? 1 : 0)) >> 1;   // window pair of output column 0's first tap pair (-1: left padding)
    const int Cin = a.Cin;
    const GLOBAL_AS _Float16* xg = sgpr_ptr<_Float16>(a.X) + (size_t)b * HIMG * HIMG * Cin;
    const GLOBAL_AS _Float16* wexp = sgpr_ptr<_Float16>(a.wexp);
    uint32_t raw[15];
    float dbias;
    const int cg_dw = chunk * CH + tid % CH;
    // ---------------- expand ----------------
    {
        u4v xr[NS][CKS];
        // 3x3 (28 -> 14, TF-same pads bottom / right only): image row 28 and columns 28, 29 of the window are NOT zeroed in E -- the
        // taps that read them (kernel row 2 of output row 13; kernel column 2 of output column 13, always the tile's last) get a zero
        // weight / are left out in the depthwise phase.  5x5 (pads on all four sides) keeps E zeroed.
        constexpr bool MASK_E = KSD != 3;
        static_assert(KSD != 3 || HIMG == 28, "the 3x3 stride-2 shortcut assumes one tile column");
        bool okp[NS][2];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int p = 16 * (wave + 8 * i) + m;
            const int r = p / WW, c = p - r * WW;
            const int iy = wy0 + r, ix = wx0 + c;
            const bool ok = p < NPOS && iy >= 0 && iy < HIMG && ix < HIMG;
            const int row = ok ? iy * HIMG + ix : 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pp = 16 * (wave + 8 * i) + 4 * q + 2 * h;
                const int pr = pp / WW, pc = pp - pr * WW;
                const int py = wy0 + pr;
                okp[i][h] = !MASK_E || (pp < NPOS && py >= 0 && py < HIMG && wx0 + pc < HIMG);
            }
#pragma unroll
            for (int ks = 0; ks < CKS; ++ks) {
                const int kk = 32 * ks + 8 * q;
                xr[i][ks] = gload<u4v>(xg, (unsigned)((row * Cin + (kk < Cin ? kk : Cin - 8)) * 2));
            }
        }
        h8 wa[3][CKS];
        float ba[3];
#pragma unroll
        for (int nf = 0; nf < 3; ++nf) {
            const int nfg = 3 * chunk + nf;
#pragma unroll
            for (int ks = 0; ks < CKS; ++ks) wa[nf][ks] = gload<h8>(wexp, (unsigned)(((nfg * CKS + ks) * 64 + lane) * 16));
            ba[nf] = a.bexp[16 * nfg + m];
        }
        // depthwise taps and bias of this thread's channel, behind the expand's operands and pinned (see mbt_kernel)
#pragma unroll
        for (int i = 0; i < 3 * KSD; ++i) raw[i] = a.dwp[(size_t)i * CE + cg_dw];
        dbias = a.bdw[cg_dw];
        PIN_VMEM();
        f4 ba4[3];   // the bias as a ready accumulator operand (see mbt_kernel)
#pragma unroll
        for (int nf = 0; nf < 3; ++nf) ba4[nf] = f4{ba[nf], ba[nf], ba[nf], ba[nf]};
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (wave + 8 * i >= NPF) continue;   // wave-uniform
            h8 xb[CKS];
#pragma unroll
            for (int ks = 0; ks < CKS; ++ks) {
                xb[ks] = *reinterpret_cast<const h8*>(&xr[i][ks]);   // (zero-padded K columns: finite re-reads x zero weight rows)
            }
            const int pair0 = 8 * (wave + 8 * i) + 2 * q;
            float t[12];   // the three output fragments' accumulators: SiLU staged over all twelve
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) {
                f4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xb[0], wa[nf][0], ba4[nf], 0, 0, 0);
#pragma unroll
                for (int ks = 1; ks < CKS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xb[ks], wa[nf][ks], acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) t[4 * nf + j] = acc[j];
            }
            silu_scaled_staged(t);
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) {
                const h2 p0 = {(_Float16)t[4 * nf], (_Float16)t[4 * nf + 1]};
                const h2 p1 = {(_Float16)t[4 * nf + 2], (_Float16)t[4 * nf + 3]};
                const h2 z = {(_Float16)0.0f, (_Float16)0.0f};
                unsigned char* dst = E + pair0 * ES2 + (16 * nf + m) * 4;
                *reinterpret_cast<h2*>(dst) = okp[i][0] ? p0 : z;
                *reinterpret_cast<h2*>(dst + ES2) = okp[i][1] ? p1 : z;
            }
        }
    }
    T7_BAR();
    // ---------------- depthwise, stride 2 ----------------
    // Work item = (channel, output row, half of the 14 columns): 672 items of 7 outputs over 512 threads -- a first pass of all eight
    // waves, a second of waves 0 .. 2 (a few threads repeat an item: same values, same addresses, no branch around a store).
    // One item per (channel, output row) was 336 threads: waves 0 .. 5 carried 14 outputs each while 6 and 7 idled -- two SIMDs with
    // two loaded waves, two with one; now the busiest SIMD issues three half-items' worth instead of four, and the window is 9 pairs
    // per row instead of 16.
    constexpr int NPH = KSD == 5 ? 9 : 8;   // window pairs of a half row: output 7 half + j reads the pairs 7 half + j + ip
    static_assert(KSD == 5 ? (7 + NPH - 1 < WW / 2 + 1) : (7 + NPH - 1 < WW / 2), "the last pair of the right half lies inside the window");
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && wave >= 3) break;   // wave-uniform: items 512 .. 671 are waves 0 .. 2
        // A thread keeps its channel (tid % 48: its taps are in registers): pass 0 takes rows 0 .. 10 of the 14 (row 10: channels
        // 0 .. 31), pass 1 rows 11, 12, 13 in threads 0 .. 143 and the rest of row 10 in threads 144 .. 191 (whose channels 0 .. 31
        // repeat pass 0's items).
        const int c = tid % CH, r1 = tid / CH;
        const int rest = pass == 0 ? r1 : (r1 < 3 ? 11 + r1 : 10);        // rest = 7 half + output row
        const int half = rest >= 7 ? 1 : 0, orow = rest - 7 * half;
        const int cg = chunk * CH + c;
        // only window pair 0 of the left half can fall outside the window (left image border: pbase = -1): zero padding
        const int pb = pbase + 7 * half;
        const bool lok = pb >= 0;
        const unsigned char* col = E + 4 * c + (lok ? pb : 0) * ES2;
        const int o0 = lok ? 0 : -ES2;
        uint32_t P[KSD][NPH];
#pragma unroll
        for (int ky = 0; ky < KSD; ++ky) {
            const unsigned char* rowp = col + ((2 * orow + ky) * (WW / 2)) * ES2;
            const uint32_t v0 = *reinterpret_cast<const uint32_t*>(rowp);
            P[ky][0] = lok ? v0 : 0u;
#pragma unroll
            for (int l = 1; l < NPH; ++l) P[ky][l] = *reinterpret_cast<const uint32_t*>(rowp + o0 + l * ES2);
        }
        float acc[7];
#pragma unroll
        for (int ky = 0; ky < KSD; ++ky) {
            const uint32_t r0 = raw[3 * ky], r1 = raw[3 * ky + 1], r2 = raw[3 * ky + 2];
            uint32_t wq[3], wql = 0u;
            if (KSD == 5) { wq[0] = r0 << 16; wq[1] = __builtin_amdgcn_alignbit(r1, r0, 16); wq[2] = __builtin_amdgcn_alignbit(r2, r1, 16); }
            else {
                wq[0] = r0; wq[1] = r1; wq[2] = 0u;
                if (ky == 2) {   // window row 2 orow + 2 is image row 28 for output row 13
                    const bool out = oy0 + orow == HOUT - 1;
                    wq[0] = out ? 0u : wq[0]; wq[1] = out ? 0u : wq[1];
                }
                wql = half ? 0u : wq[1];   // (k2, 0) of output column 13 falls on columns 28, 29: outside the image
            }
#pragma unroll
            for (int ip = 0; ip < NIP; ++ip)
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const uint32_t wt = (KSD == 3 && j == 6 && ip == 1) ? wql : wq[ip];
                    if (ky == 0 && ip == 0) acc[j] = dot2_from(P[ky][j + ip], wt, dbias);
                    else acc[j] = __builtin_amdgcn_fdot2(*reinterpret_cast<const h2*>(&P[ky][j + ip]), *reinterpret_cast<const h2*>(&wt), acc[j], false);
                }
        }
        f2 psum2 = {0.f, 0.f};
        uint16_t* dg = reinterpret_cast<uint16_t*>(a.D + (((size_t)b * HOUT + oy0 + orow) * HOUT + ox0 + 7 * half) * CE + cg);
        silu_scaled_staged(acc);
#pragma unroll
        for (int j = 0; j < 6; j += 2) {
            const f2 v = {acc[j], acc[j + 1]};
            psum2 = psum2 + v;
            const uint32_t hv = cvt_pk_f16(acc[j], acc[j + 1]);
            dg[(size_t)j * CE] = (uint16_t)hv;
            dg[(size_t)(j + 1) * CE] = (uint16_t)(hv >> 16);
        }
        reinterpret_cast<_Float16*>(dg)[(size_t)6 * CE] = (_Float16)acc[6];
        pred[rest * CH + c] = (psum2.x + psum2.y) + acc[6];
    }
    T7_BAR();
    if (tid < CH) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 14; ++w) s += pred[w * CH + tid];
        a.pool[((size_t)b * gridDim.x + tile) * CE + chunk * CH + tid] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// proj_patch_kernel: squeeze-excite + project conv (+ residual) of ONE patch per workgroup, for the 28x28 and
// 14x14 blocks (b3..b10).  There the separate path is launch- and latency-bound (an SE launch of ~10 us plus a
// project GEMM whose workgroups each do a few dozen MFMAs); with the whole patch in one workgroup
//   * the squeeze-excite gate is computed in the prologue from the depthwise kernel's pool partials (two
//     matrix-vector products on fp16 weights, fp32 accumulate) -- no second launch, no gate tensor in HBM;
//   * the project weights (N x K, <= 147 KB) are fetched once per patch, parked in LDS in MFMA fragment order,
//     and every wave streams its pixel fragments of the patch's depthwise output (the B operand, gated in
//     registers with v_fma_mixlo/hi_f16) against them: swapped MFMA, two pixel fragments per weight-fragment read.
// 512 threads; wave w owns the pixel-fragment pairs w, w+8, ...; K is walked in chunks of CK k-steps with the
// next chunk's pixel fragments in flight.  Template: KS = k-steps of 32 (K zero-padded), NF = 16-channel output
// fragments (N zero-padded), HW = pixels per patch, RES = skip connection.
// ---------------------------------------------------------------------------------------------
template <int KS, int NF, int HW, bool RES>
__global__ __launch_bounds__(512) void proj_patch_kernel(ProjPatchArgs a)
{
    // k-steps per chunk of pixel fragments (the next chunk is in flight while one computes).  Small chunks put fewer bytes in front
    // of the first MFMA (the prologue is bound by them): KS = 21 -> 3 (22.2 -> 21.4 us), KS = 15 -> 3 with five output fragments
    // (16.0 -> 15.5 us) but 5 with seven (3: 16.6 -> 17.2 us), KS = 12 -> 6.
    constexpr int CK = (KS <= 8) ? KS : (KS % 7 == 0 ? 3 : (KS % 5 == 0 ? (NF <= 5 ? 3 : 5) : 6));
    constexpr int NCH = KS / CK;
    static_assert(KS % CK == 0, "chunking");
    constexpr int NPF = (HW + 15) / 16, NPAIR = (NPF + 1) / 2;
    constexpr int NWCH = NF * KS * 64;                 // 16-byte chunks of the weight image
    constexpr int WPT = (NWCH + 511) / 512;            // chunks per thread
    constexpr int KP = 32 * KS;
    constexpr bool EARLY_RES = KS < 21;   // the 672-channel blocks (b9, b10) are at the register limit
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wl = smem;                                           // [NF][KS][64 lanes][16 B]
    float* pooled = reinterpret_cast<float*>(smem + NF * KS * 1024);    // [KP]
    float* gate = pooled + KP;                                          // [KP]
    float* rs = gate + KP;                                              // [32] squeeze activations
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int b = blockIdx.x;
    const int K = a.K, CSP = a.CSP;
    long long tk0 = 0, tk1 = 0, tk2 = 0;
    if (a.dbg_clk) tk0 = (long long)__builtin_readcyclecounter();
    const GLOBAL_AS _Float16* wfrag = sgpr_ptr<_Float16>(a.wfrag);
    const GLOBAL_AS _Float16* wr_g = sgpr_ptr<_Float16>(a.wr_g);
    const GLOBAL_AS _Float16* we_t = sgpr_ptr<_Float16>(a.we_t);
    const GLOBAL_AS float* pp = sgpr_ptr<float>(a.pool_part);
    constexpr int FC1_IT = (KP + 63) / 64, KPAD = 64 * FC1_IT;   // FC1: channel slices of 64 per lane (see below)
    const int G = CSP >> 2;
    const bool fc1_wave = wave < G;
    const GLOBAL_AS _Float16* xg = sgpr_ptr<_Float16>(a.X) + (size_t)b * HW * K;
    auto load_chunk = [&](int pr, int ch, h8 (&dst)[2][CK]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = 32 * pr + 16 * i + m;
            const int pix = p < HW ? p : HW - 1;
#pragma unroll
            for (int u = 0; u < CK; ++u) {
                // Columns beyond K (zero-padded k-steps) re-read the row's last 8 channels: their gate and weights are
                // zero, so no lane predicate is needed -- a per-lane branch around the load would make the compiler
                // drain ALL outstanding loads (vmcnt(0)) at every chunk boundary and undo the prefetch.
                // (No masking either: the re-read values are finite depthwise outputs and meet a zero gate, so they contribute an
                // exact zero.  The select that used to zero them consumed each load at once -- at the register limit the compiler
                // then issued the loads one at a time, each behind an s_waitcnt vmcnt(0): 14 exposed round trips in the prologue.)
                const int k = 32 * (ch * CK + u) + 8 * q;
                dst[i][u] = gload<h8>(xg, (unsigned)((pix * K + (k < K ? k : K - 8)) * 2));
            }
        }
    };
    // Bulk loads, behind the chain's inputs in every wave's queue.  The project weights (<= 147 KB, the same for every workgroup) go
    // from L2 STRAIGHT into LDS (global_load_lds_dwordx4: wave-uniform LDS base + lane x 16 bytes = the lane-linear fragment image),
    // ALL of them issued by wave 7, which has no part in FC1: no 76 staging registers, no ds_write pass, and no other wave has a DMA
    // in flight (hipcc waits for vmcnt(0) at the next use of a plain load while one is).  Staged through registers by all waves
    // between FC1 and the reduce, the requests' issue alone (264 KB per workgroup at ~32 B/clk) put 8 k cycles between those two
    // barriers, and parking the weights cost another 2 k behind FC2.  The first pixel fragments are requested here as well.
    // (A wave issues in order and the memory pipe pushes back: with every wave's first pixel fragments up here too -- 264 KB per
    // workgroup in front of the chain -- the pooled barrier came at 15 k cycles.  They go out behind FC1, in the waves that idle there.)
    // A CU's memory pipe serves requests in issue order, whichever wave they come from: the barrier puts every wave's chain inputs in
    // the queue ahead of the first DMA piece (without it the pool sums came back behind the weights: pooled barrier at 11 k cycles).
    // The DMA goes out in three slices, one in front of each barrier of the chain: a wave's memory queue holds ~64 requests and takes
    // ~40 cycles per 1-KB piece, so wave 7 issuing all of them (up to 147) in front of the pooled barrier held every other wave there
    // for up to 6 k cycles (phase clock: pooled barrier at 10-11 k cycles for K = 672, 5 k for K = 144).  Slice sizes are what the
    // other waves' work between two barriers covers; wave 7 does nothing else in the prologue (the reduce moved to wave 0).
    constexpr int NW = NF * KS;
    constexpr int DS0 = NW < 44 ? NW : 44, DS1 = NW - DS0 < 50 ? NW - DS0 : 50;
    // (Every workgroup fetches the same image in the same order at about the same time; starting each at a different sixteenth of it
    // -- so that the workgroups of an XCD do not ask one L2 channel for the same line together -- changed nothing: measured.  Larger
    // first slices (72 / 25 / 35) moved the pooled barrier out by what they took.)
    auto dma = [&](auto i0_tag, auto i1_tag) {
        constexpr int I0 = decltype(i0_tag)::value, I1 = decltype(i1_tag)::value;
#pragma unroll
        for (int i = I0; i < I1; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.wfrag + ((size_t)i * 64 + lane) * 8),
                                             (__attribute__((address_space(3))) void*)(wl + i * 1024), 16, 0, 0);
        PIN_VMEM();
    };
    h8 xc[2][CK], xn[2][CK];
    // Wave 7 runs its own straight-line path with the chain's four barriers in it: hipcc cannot count vmcnt across a DMA (LDS-DMA
    // pieces retire out of order with plain loads), so in a shared path every wave would meet s_waitcnt vmcnt(0) at each use of a
    // loaded register behind a point where a DMA MAY be in flight -- FC2 would wait for the first pixel fragments, and so on.
    if (wave == 7) {
        __builtin_amdgcn_s_barrier();   // (the other waves' chain inputs are in the memory queue)
        dma(std::integral_constant<int, 0>{}, std::integral_constant<int, DS0>{});
        T7_BAR();   // pooled
        dma(std::integral_constant<int, DS0>{}, std::integral_constant<int, DS0 + DS1>{});
        T7_BAR();   // FC1
        dma(std::integral_constant<int, DS0 + DS1>{}, std::integral_constant<int, NW>{});
        if (7 < NPAIR) { load_chunk(7, 0, xc); PIN_VMEM(); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA has landed
        T7_BAR();   // FC2 / gate
    } else {
        // ---- The prologue is a dependent chain (pool sums -> FC1 -> FC2 -> gate) beside ~300 KB of bulk loads (project
        //      weights, first pixel fragments).  Loads return in order and a wave cannot pass a barrier before it has
        //      ISSUED its loads (the memory pipe takes 64 B/clk), so: chain inputs first (pool partials, then the FC
        //      weights), all unpredicated and straight-line; the bulk loads go out after the first barrier and stream
        //      in while the FCs compute.
        // Pool channels: thread t takes channel t and channel t + 512 -- except wave 7 (the DMA path above): its channels 448 .. 511
        // ride in the second slot of threads 160 .. 223, which is free (K <= 672).
        static_assert(KP <= 672, "channel 448..511 reassignment assumes no channel t + 512 for t >= 160");
        float ps0 = 0.f, ps1 = 0.f;
        const int pk0 = tid, pk1 = (tid >= 160 && tid < 224) ? tid + 288 : tid + 512;
        {
            const int k0 = pk0 < K ? pk0 : 0, k1 = pk1 < K ? pk1 : 0;
            if (a.nparts == 1) {
                ps0 = gload<float>(pp, (unsigned)((b * K + k0) * 4));
                ps1 = gload<float>(pp, (unsigned)((b * K + k1) * 4));
            } else {   // up to 16 tiles per patch (b3: 14, b4: 4, b5: 7): every load issued before the first add
                float v0[16], v1[16];
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    const int pc = p < a.nparts ? p : 0;
                    v0[p] = gload<float>(pp, (unsigned)(((b * a.nparts + pc) * K + k0) * 4));
                    v1[p] = gload<float>(pp, (unsigned)(((b * a.nparts + pc) * K + k1) * 4));
                }
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    ps0 += p < a.nparts ? v0[p] : 0.f;
                    ps1 += p < a.nparts ? v1[p] : 0.f;
                }
            }
        }
        // FC1: wave = 4 outputs (group j4 = wave < CSP / 4), lane = one of 64 channel slices (k = lane, lane + 64, ...): the 64 partial
        // sums of an output are the lanes of ONE wave and add up on DPP row operations + four v_readlane -- no partials in LDS, no
        // reduce stage, one barrier less in the chain (that stage took 2.1-3.0 k of the prologue's 10-14.7 k cycles).  Weights
        // host-packed [group][64 * FC1_IT channels][4] (zero beyond K): a lane's request is 8 bytes next to its neighbours'.
        const int wj = fc1_wave ? wave : 0;   // (idle waves re-read group 0: no branch around the loads)
        u2v w1[FC1_IT];
#pragma unroll
        for (int i = 0; i < FC1_IT; ++i) w1[i] = gload<u2v>(wr_g, (unsigned)(((wj * KPAD + lane + 64 * i) * 4) * 2));
        // FC2: thread = channels 2*tid, 2*tid + 1 (one dword of We^T per squeeze unit)
        const int k2 = 2 * tid;
        const bool fc2_thr = k2 < K;
        uint32_t w2[28];
#pragma unroll
        for (int j = 0; j < 28; ++j) w2[j] = gload<uint32_t>(we_t, (unsigned)(((j < CSP ? j : 0) * K + (fc2_thr ? k2 : 0)) * 2));
        const float be0 = fc2_thr ? a.be[k2] : 0.f, be1 = fc2_thr ? a.be[k2 + 1] : 0.f;
        const float brv = a.br[4 * wj + (lane & 3)];   // squeeze bias of output 4 * wave + lane (lanes 0 .. 3 finish FC1; br is padded to 32)
    PIN_VMEM();
    __builtin_amdgcn_s_barrier();
    if (pk0 < KP) pooled[pk0] = pk0 < K ? ps0 : 0.f;
    if (pk1 < KP) pooled[pk1] = pk1 < K ? ps1 : 0.f;
    T7_BAR();
    if (a.dbg_clk && tid == 0) a.dbg_clk[(size_t)b * 8 + 3] = (float)((long long)__builtin_readcyclecounter() - tk0);
    // ---- FC1: r = silu(br + psc * pooled . Wr^T) ----
    // (The bulk loads -- 147 KB of project weights and the first pixel fragments, 33 x 16 bytes per thread -- used to be issued
    // HERE, in front of FC1: their address processing alone takes ~4 k cycles per workgroup and FC1's barrier came 9-10 k cycles
    // after the pooled one.  They are needed only after FC2, so they now go out behind FC1 and stream in under the reduce and FC2.)
    if (fc1_wave) {
        // the lane's pool sums in ONE LDS round trip (clamped addresses: slots beyond the padded K meet zero weights)
        float xs[FC1_IT];
#pragma unroll
        for (int i = 0; i < FC1_IT; ++i) {
            const int k = lane + 64 * i;
            xs[i] = pooled[k < KP ? k : 0];
        }
        __builtin_amdgcn_sched_barrier(0);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < FC1_IT; ++i) {
            acc[0] = fma_mix_lo(w1[i].x, xs[i], acc[0]);
            acc[1] = fma_mix_hi(w1[i].x, xs[i], acc[1]);
            acc[2] = fma_mix_lo(w1[i].y, xs[i], acc[2]);
            acc[3] = fma_mix_hi(w1[i].y, xs[i], acc[3]);
        }
        float tot[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // fixed order: rows of 16 lanes on DPP, then (row 0 + row 1) + (row 2 + row 3)
            const int v = __builtin_bit_cast(int, row16_sum(acc[j]));
            tot[j] = (__builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 16))) +
                     (__builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 48)));
        }
        if (lane < 4) {
            const float sv = lane == 0 ? tot[0] : (lane == 1 ? tot[1] : (lane == 2 ? tot[2] : tot[3]));
            rs[4 * wave + lane] = silu_f(sv * a.psc + brv);
        }
    }
    T7_BAR();
    if (a.dbg_clk && tid == 0) {
        a.dbg_clk[(size_t)b * 8 + 4] = (float)((long long)__builtin_readcyclecounter() - tk0);
        a.dbg_clk[(size_t)b * 8 + 5] = a.dbg_clk[(size_t)b * 8 + 4];   // (the reduce stage is gone: same stamp)
    }
    // first pixel fragments of every wave (wave 7's behind the last DMA slice), unconditional: every layer has at least seven
    // pairs of pixel fragments, and a load behind a branch would cost the chain its counted waits.  Behind FC1's barrier (a wave
    // passes a barrier only once its loads are ISSUED -- in front of it, block 4's 112 KB held the barrier for 4 k cycles); FC2
    // waits with a counted vmcnt for its own operands only.
    static_assert(NPAIR >= 7, "waves 0..6 all own a pair of pixel fragments");
    load_chunk(wave, 0, xc);
    PIN_VMEM();
    // ---- FC2: gate = sigmoid(be + r . We^T) ----
    if (k2 < KP) {   // (whole waves beyond K skip it: wave 7 must not wait here for operands queued behind its DMA)
        float a0 = be0, a1 = be1;
        f4 rq[7];   // all 28 squeeze slots in one round trip (slots >= CSP hold whatever: selected to zero below)
#pragma unroll
        for (int j = 0; j < 7; ++j) rq[j] = *reinterpret_cast<const f4*>(rs + 4 * j);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 28; ++j) {
            const float r = j < CSP ? rq[j >> 2][j & 3] : 0.f;
            a0 = fma_mix_lo(w2[j], r, a0);
            a1 = fma_mix_hi(w2[j], r, a1);
        }
        {   // zero beyond K: the zero-padded x columns stay zero
            const float g0 = fc2_thr ? sigmoid_f(a0) : 0.f, g1 = fc2_thr ? sigmoid_f(a1) : 0.f;
            gate[k2] = g0;
            gate[k2 + 1] = g1;
            if (a.dbg_gate && fc2_thr) {
                a.dbg_gate[(size_t)b * K + k2] = g0;
                a.dbg_gate[(size_t)b * K + k2 + 1] = g1;
            }
        }
    }
    if (a.dbg_clk && tid == 0) a.dbg_clk[(size_t)b * 8 + 6] = (float)((long long)__builtin_readcyclecounter() - tk0);
    T7_BAR();
    }
    if (a.dbg_clk && tid == 0) a.dbg_clk[(size_t)b * 8 + 7] = (float)((long long)__builtin_readcyclecounter() - tk0);
    if (a.dbg_clk) tk1 = (long long)__builtin_readcyclecounter();
    // ---- project: Y[pixel][n] = sum_k (X[pixel][k] * gate[k]) W[n][k] + bias (+ residual) ----
    f4 bv[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) bv[nf] = *reinterpret_cast<const f4*>(a.bias + 16 * nf + 4 * q);
#pragma unroll 1
    for (int pr = wave; pr < NPAIR; pr += 8) {
        f4 acc[2][NF];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[i][nf] = bv[nf];
        // skip-connection input of this pair (unpredicated, clamped addresses: one round trip for all of it): requested
        // before the k-loop where registers allow (EARLY_RES), else at the start of the epilogue
        h4 rv[2][NF];
        auto load_res = [&]() {
            if (!RES) return;
            const GLOBAL_AS _Float16* rg = sgpr_ptr<_Float16>(a.res) + (size_t)b * HW * a.N;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int p = 32 * pr + 16 * i + m;
                const int pc = p < HW ? p : HW - 1;
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
                    const int c = 16 * nf + 4 * q;
                    rv[i][nf] = gload<h4>(rg, (unsigned)((pc * a.N + (c < a.N ? c : 0)) * 2));
                }
            }
        };
        if (EARLY_RES) load_res();
        h8 wcur[NF];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) wcur[nf] = *reinterpret_cast<const h8*>(wl + ((nf * KS) * 64 + lane) * 16);
        f4 gc0 = *reinterpret_cast<const f4*>(gate + 8 * q), gc1 = *reinterpret_cast<const f4*>(gate + 8 * q + 4);
        long long tka = 0;
        if (a.dbg_clk) tka = (long long)__builtin_readcyclecounter();
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            // the next chunk (of this pair, or the first one of the wave's next pair) is in flight while this one computes
            // (always issued -- the wave's last pair re-reads its own first chunk -- so that the load count at the wait
            // below is a compile-time constant; a conditional load makes the compiler wait for vmcnt(0))
            if (ch + 1 < NCH) load_chunk(pr, ch + 1, xn);
            else load_chunk(pr + 8 < NPAIR ? pr + 8 : pr, 0, xn);
#pragma unroll
            for (int u = 0; u < CK; ++u) {
                const int ks = ch * CK + u;
                // weight fragments (LDS) and gates of the NEXT k-step are read while this one's MFMAs run: otherwise
                // every fragment's LDS latency sits in front of its two MFMAs
                const int kn = ks + 1 < KS ? ks + 1 : 0;
                h8 wnx[NF];
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) wnx[nf] = *reinterpret_cast<const h8*>(wl + ((nf * KS + kn) * 64 + lane) * 16);
                const f4 gn0 = *reinterpret_cast<const f4*>(gate + 32 * kn + 8 * q);
                const f4 gn1 = *reinterpret_cast<const f4*>(gate + 32 * kn + 8 * q + 4);
                PIN_VMEM();   // keep those reads AHEAD of this k-step's MFMAs (the scheduler sinks them otherwise)
                h8 xb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint4 o = gate_h8(*reinterpret_cast<const uint4*>(&xc[i][u]), gc0, gc1);
                    xb[i] = *reinterpret_cast<const h8*>(&o);
                }
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
                    acc[0][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wcur[nf], xb[0], acc[0][nf], 0, 0, 0);
                    acc[1][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wcur[nf], xb[1], acc[1][nf], 0, 0, 0);
                }
                PIN_VMEM();
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) wcur[nf] = wnx[nf];
                gc0 = gn0;
                gc1 = gn1;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int u = 0; u < CK; ++u) xc[i][u] = xn[i][u];
        }
        if (a.dbg_clk && tid == 0) a.dbg_clk[(size_t)b * 8 + 2] = (float)((long long)__builtin_readcyclecounter() - tka);
        if (!EARLY_RES) load_res();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = 32 * pr + 16 * i + m;
            if (p >= HW) continue;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const int c = 16 * nf + 4 * q;
                if (c >= a.N) continue;   // N is a multiple of 4 (padding fragments are dropped)
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[i][nf][j];
                if (RES) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += (float)rv[i][nf][j];
                }
                h4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (_Float16)v[j];
                *reinterpret_cast<h4*>(a.Y + ((size_t)b * HW + p) * a.N + c) = o;
            }
        }
    }
    if (a.dbg_clk) {
        T7_BAR();
        tk2 = (long long)__builtin_readcyclecounter();
        if (tid == 0) { a.dbg_clk[(size_t)b * 8] = (float)(tk1 - tk0); a.dbg_clk[(size_t)b * 8 + 1] = (float)(tk2 - tk1); }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused stem + block-0 depthwise: u8 patch -> [stem conv3x3s2 + bias + SiLU] -> LDS -> [depthwise 3x3 s1 +
// bias + SiLU] -> fp16 NHWC (112x112x32) + squeeze-excite partial sums.  The 112x112x32 stem output (the
// largest tensor of the net after the expanded ones) never goes to HBM.  Same two-phase structure as
// mbconv_a_kernel: one workgroup = (patch, 16x16 output tile); phase 1 builds MFMA operands from the
// staged u8 tile exactly like stem_conv_kernel (K packing and normalisation folding are shared).
// ---------------------------------------------------------------------------------------------
#define SD_T 16                       // output tile edge
#define SD_WIN (SD_T + 2)             // stem window edge (halo 1)
#define SD_IN (2 * SD_WIN + 1)        // input tile edge (37)
#define SD_ROWH 128                   // halves per staged input row (123 used)
#define SD_ES 80                      // bytes per E row: 32 ch * 2 + 16
__global__ __launch_bounds__(256) void stem_dw_kernel(const uint8_t* __restrict__ patches,  // [B][224][224][3]
                                                      const _Float16* __restrict__ w,        // [32][32] stem (n, kslot)
                                                      const float* __restrict__ bias,        // [32] stem
                                                      const float* __restrict__ padval,      // [3]
                                                      const float* __restrict__ Wdw,         // [9][32]
                                                      const float* __restrict__ bdw,         // [32]
                                                      _Float16* __restrict__ out,            // [B][112][112][32]
                                                      float* __restrict__ pool_part)         // [B][49][32]
{
    __shared__ __attribute__((aligned(16))) _Float16 tile[SD_IN * SD_ROWH];                    // 8.3 KB
    __shared__ __attribute__((aligned(16))) unsigned char E[((SD_WIN * SD_WIN + 15) / 16 * 16) * SD_ES];  // 26.9 KB
    __shared__ __attribute__((aligned(16))) float wl[9 * 32];
    const int tx = blockIdx.x, ty = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int oy0 = ty * SD_T, ox0 = tx * SD_T;
    int wy0 = oy0 - 1, wy1 = oy0 + SD_T + 1, wx0 = ox0 - 1, wx1 = ox0 + SD_T + 1;
    wy0 = wy0 < 0 ? 0 : wy0;
    wx0 = wx0 < 0 ? 0 : wx0;
    wy1 = wy1 > 112 ? 112 : wy1;
    wx1 = wx1 > 112 ? 112 : wx1;
    const int wh = wy1 - wy0;
    // The stem window is enumerated as a FIXED 18 x 18 grid from (oy0 - 1, ox0 - 1); positions outside the image hold zeros in E
    // (the depthwise conv's padding) and are skipped by phase 1.  Phase 2 then needs no bounds test, no skipped kernel row and no
    // address arithmetic (every E read is lane base + immediate), and a depthwise accumulator starts as its first tap's addend.
    constexpr int P = SD_WIN * SD_WIN;
    const int dy = wy0 - (oy0 - 1);                  // 1 for the top tiles (window row 0 is above the image)
    // ---- stage the input tile as exact fp16 (u8-128): rows 2*wy0 .. 2*wy1, cols from 2*(16tx-2) so that every
    //      row segment starts on a dword (6*(16tx-2) bytes); 37 rows x 31 dwords (41 pixels), one dword per thread-step
    const int icol0 = 2 * (ox0 - 2);                 // may be -4 for the leftmost tiles (those pixels are never used)
    constexpr int cshift = 2;                        // staged column of window column 0's first input pixel: 2 (ox0 - 1) - icol0
    // Everything the later phases read from global memory is requested HERE, in front of the input tile (round 3: the stem weights and
    // biases used to be loaded behind the first barrier and the depthwise biases behind the second -- one exposed L2 round trip each).
    h8 wf[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) wf[t] = *reinterpret_cast<const h8*>(w + (t * 16 + m) * 32 + q * 8);
    f4 bsv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) bsv[t] = *reinterpret_cast<const f4*>(bias + q * 8 + t * 4);
    const f4 bdw0 = *reinterpret_cast<const f4*>(bdw + (tid & 3) * 8), bdw1 = *reinterpret_cast<const f4*>(bdw + (tid & 3) * 8 + 4);
    const float wl0 = Wdw[tid], wl1 = Wdw[256 + (tid & 31)];
    {
        const uint8_t* img = patches + (size_t)b * (224 * 224 * 3);
        const int iy0 = 2 * wy0;
        const int rows = 2 * wh + 1;
        const float pv0 = padval[0], pv1 = padval[1], pv2 = padval[2];
        // Tiles that touch neither the right nor the bottom image edge need no padding values: four bytes -> four exact fp16
        // (u8 - 128) with v_cvt_f32_ubyteN + packed converts and ONE 8-byte LDS store (the per-byte path below spends ~12
        // instructions per byte on pixel/channel bookkeeping).  Same values either way.
        const bool interior = tx < 6 && ty < 6;   // workgroup-uniform
        if (interior) {
            // ALL of a thread's dwords are requested before the first is converted (round 3: the loop used to be load -> s_waitcnt
            // vmcnt(0) -> convert -> store, four to five exposed HBM round trips in front of everything else the workgroup does).
            // Unconditional loads from clamped addresses: a branch around a load makes the compiler wait for it at once.  Dwords left
            // of the image (boff < 0, leftmost tiles) and past the last row hold some other pixel's bytes: never used / never stored.
            constexpr int NIT = (37 * 31 + 255) / 256;
            uint32_t wd[NIT];
            const int nd = rows * 31;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i0 = tid + 256 * it, i = i0 < nd ? i0 : nd - 1;
                const int r = i / 31, d = i - r * 31;
                const int boff = icol0 * 3 + d * 4;
                wd[it] = *reinterpret_cast<const uint32_t*>(img + (size_t)(iy0 + r) * 672 + (boff >= 0 ? boff : 0));
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = tid + 256 * it;
                const int r = i / 31, d = i - r * 31;
                const uint32_t word = wd[it];
                h4 v = {(_Float16)((float)(word & 0xffu) - 128.0f), (_Float16)((float)((word >> 8) & 0xffu) - 128.0f),
                        (_Float16)((float)((word >> 16) & 0xffu) - 128.0f), (_Float16)((float)(word >> 24) - 128.0f)};
                if (i < nd) *reinterpret_cast<h4*>(tile + r * SD_ROWH + d * 4) = v;   // (row stride 128 halves: the 124th half is spare)
            }
        } else {
        // right / bottom edge tiles: the same, all requests first (unconditional, clamped to the image; what lies outside is replaced
        // by the padding values below)
        constexpr int NIT = (37 * 31 + 255) / 256;
        uint32_t wd[NIT];
        const int nd = rows * 31;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i0 = tid + 256 * it, i = i0 < nd ? i0 : nd - 1;
            const int r = i / 31, d = i - r * 31;
            const int iy = iy0 + r < 224 ? iy0 + r : 223;
            int boff = icol0 * 3 + d * 4;
            boff = boff < 0 ? 0 : (boff > 668 ? 668 : boff);
            wd[it] = *reinterpret_cast<const uint32_t*>(img + (size_t)iy * 672 + boff);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + 256 * it;
            if (i >= nd) break;
            const int r = i / 31, d = i - r * 31;
            const int iy = iy0 + r;
            const uint32_t word = wd[it];
            const bool row_ok = iy < 224;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int bb = d * 4 + e;             // byte inside the staged row (0..123)
                const int pix = bb / 3, c = bb - pix * 3;
                const int col = icol0 + pix;
                float v;
                if (row_ok && col >= 0 && col < 224) v = (float)((word >> (8 * e)) & 0xffu) - 128.0f;
                else v = (c == 0) ? pv0 : (c == 1 ? pv1 : pv2);
                if (bb < 123) tile[r * SD_ROWH + bb] = (_Float16)v;
            }
        }
        }
        wl[tid] = wl0;
        if (tid < 32) wl[256 + tid] = wl1;
        if (tx == 0 || ty == 0 || tx == 6 || ty == 6)   // border tiles: zero the window positions outside the image
            for (int pz = tid; pz < P; pz += 256) {
                const int py = pz / SD_WIN, px = pz - py * SD_WIN;
                const int sy = oy0 - 1 + py, sx = ox0 - 1 + px;
                if (sy < 0 || sy >= 112 || sx < 0 || sx >= 112) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) *reinterpret_cast<uint4*>(E + pz * SD_ES + 16 * v) = uint4{0u, 0u, 0u, 0u};
                }
            }
    }
    __syncthreads();
    // ---- phase 1: stem conv on the window -> E[p][32] ----
    {
        constexpr int MTn = (P + 15) >> 4;
        for (int mt = wave; mt < MTn; mt += 4) {
            const int p = mt * 16 + m;
            const int py0 = p / SD_WIN, px0 = p - py0 * SD_WIN;
            const int sy = oy0 - 1 + py0, sx = ox0 - 1 + px0;
            const bool ok = p < P && sy >= 0 && sy < 112 && sx >= 0 && sx < 112;
            const int py = ok ? py0 - dy : 0, px = ok ? px0 : 1;   // staged rows start at stem row wy0 = oy0 - 1 + dy
            h8 a;
            if (q < 3) {
                // 8 consecutive halves starting on a dword boundary (6*px + 3*cshift is even): four 32-bit LDS reads
                const uint32_t* src = reinterpret_cast<const uint32_t*>(tile + (2 * py + q) * SD_ROWH + 3 * cshift + 6 * px);
                union { uint32_t u[4]; h8 v; } cv;
#pragma unroll
                for (int j = 0; j < 4; ++j) cv.u[j] = src[j];
                a = cv.v;
            } else {
                union { uint32_t u[4]; h8 v; } cv;
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    cv.u[j] = *reinterpret_cast<const uint32_t*>(tile + (2 * py + j) * SD_ROWH + 3 * cshift + 6 * px + 8) & 0xffffu;
                cv.u[3] = 0u;
                // slots 0,1,2 take the three values; repack: (v0,v1),(v2,0),(0,0),(0,0)
                cv.u[0] = cv.u[0] | (cv.u[1] << 16);
                cv.u[1] = cv.u[2];
                cv.u[2] = 0u;
                a = cv.v;
            }
            h8 o;
            float tv[8];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t], a, bsv[t], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) tv[t * 4 + j] = acc[j];
            }
            silu_scaled_staged(tv);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (_Float16)tv[j];
            if (ok) *reinterpret_cast<h8*>(E + p * SD_ES + q * 16) = o;   // lane (m,q): channels 8q..8q+7
        }
    }
    __syncthreads();
    // ---- phase 2: depthwise 3x3 stride 1 from E; thread = (4 channel groups) x (64 strips of 4 pixels) ----
    const int cg = tid & 3, s = tid >> 2;
    const int oyl = s >> 2, oxl = (s & 3) * 4;
    const int oy = oy0 + oyl, ox = ox0 + oxl;
    float bs[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { bs[j] = bdw0[j]; bs[4 + j] = bdw1[j]; }   // (cg = tid & 3: requested at the top of the kernel)
    float acc[4][8];
    const unsigned char* ebase = E + ((oyl * SD_WIN + oxl) * SD_ES + cg * 16);   // window position (oyl + ky, oxl + xr) = output (oy - 1 + ky, ox - 1 + xr)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        float wk[3][8];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const f4 w0 = *reinterpret_cast<const f4*>(wl + (ky * 3 + kx) * 32 + cg * 8);
            const f4 w1 = *reinterpret_cast<const f4*>(wl + (ky * 3 + kx) * 32 + cg * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { wk[kx][j] = w0[j]; wk[kx][4 + j] = w1[j]; }
        }
#pragma unroll
        for (int xr = 0; xr < 6; ++xr) {
            const uint4 v = *reinterpret_cast<const uint4*>(ebase + (ky * SD_WIN + xr) * SD_ES);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int kx = xr - t;
                if (kx >= 0 && kx < 3) {
                    const bool first = ky == 0 && kx == 0;   // the accumulator's first tap takes the bias as its addend
                    acc[t][0] = fma_mix_lo(v.x, wk[kx][0], first ? bs[0] : acc[t][0]);
                    acc[t][1] = fma_mix_hi(v.x, wk[kx][1], first ? bs[1] : acc[t][1]);
                    acc[t][2] = fma_mix_lo(v.y, wk[kx][2], first ? bs[2] : acc[t][2]);
                    acc[t][3] = fma_mix_hi(v.y, wk[kx][3], first ? bs[3] : acc[t][3]);
                    acc[t][4] = fma_mix_lo(v.z, wk[kx][4], first ? bs[4] : acc[t][4]);
                    acc[t][5] = fma_mix_hi(v.z, wk[kx][5], first ? bs[5] : acc[t][5]);
                    acc[t][6] = fma_mix_lo(v.w, wk[kx][6], first ? bs[6] : acc[t][6]);
                    acc[t][7] = fma_mix_hi(v.w, wk[kx][7], first ? bs[7] : acc[t][7]);
                }
            }
        }
    }
    float pooled[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) pooled[j] = 0.f;
    _Float16* outb = out + (size_t)b * 112 * 112 * 32 + cg * 8;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        h8 o;
        silu_scaled_staged(acc[t]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            pooled[j] += acc[t][j];
            o[j] = (_Float16)acc[t][j];
        }
        *reinterpret_cast<h8*>(outb + ((size_t)oy * 112 + ox + t) * 32) = o;
    }
    __syncthreads();  // E is free: reuse it for the pool scratch [64 strips][32]
    float* red = reinterpret_cast<float*>(E);
#pragma unroll
    for (int j = 0; j < 8; ++j) red[s * 32 + cg * 8 + j] = pooled[j];
    __syncthreads();
    if (tid < 32) {
        float sum = 0.f;   // 16 partials requested at a time (one LDS latency per 16, not per partial), summed in the fixed order
#pragma unroll
        for (int s0 = 0; s0 < 64; s0 += 16) {
            float pv[16];
#pragma unroll
            for (int ss = 0; ss < 16; ++ss) pv[ss] = red[(s0 + ss) * 32 + tid];
#pragma unroll
            for (int ss = 0; ss < 16; ++ss) sum += pv[ss];
        }
        pool_part[((size_t)b * 49 + ty * 7 + tx) * 32 + tid] = sum;
    }
}

// =============================================================================================
// Host-side launchers (plain C++ signatures declared in kernels.h)
// =============================================================================================
#define LAUNCH_CHECK()                          \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

int launch_stem(const uint8_t* patches, const _Float16* w, const float* bias, const float* padval, _Float16* out, int B,
                int channels, hipStream_t st)
{
    dim3 grid(7, 7, B);
    if (channels == 32) hipLaunchKernelGGL(stem_conv_kernel<2>, grid, dim3(256), 0, st, patches, w, bias, padval, out);
    else if (channels == 48) hipLaunchKernelGGL(stem_conv_kernel<3>, grid, dim3(256), 0, st, patches, w, bias, padval, out);
    else return -13;
    LAUNCH_CHECK();
    return 0;
}

template <int MT, int NT, int UK, bool DG>
static int launch_gemm_uk(const GemmArgs& a, hipStream_t st)
{
    const int rows_per_wg = 64 * MT;
    dim3 grid((a.M + rows_per_wg - 1) / rows_per_wg, a.n_chunks, 1);
    dim3 block(256);
#define GEMM_GO(EPI, GATE, RES)                                                                                    \
    hipLaunchKernelGGL((pw_gemm_kernel<MT, NT, EPI, GATE, RES, UK, DG>), grid, block, 0, st, a.X, a.M, a.K, a.Wp, a.Kp / 32,   \
                       a.bias, a.Y, a.N, a.gate, a.HW, a.res, a.gap_out, a.inv_hw)
    if (a.epi == EPI_SILU) GEMM_GO(EPI_SILU, false, false);
    else if (a.epi == EPI_LINEAR) {
        if (a.gate && a.res) GEMM_GO(EPI_LINEAR, true, true);
        else if (a.gate) GEMM_GO(EPI_LINEAR, true, false);
        else if (a.res) GEMM_GO(EPI_LINEAR, false, true);
        else GEMM_GO(EPI_LINEAR, false, false);
    } else return -1;
#undef GEMM_GO
    LAUNCH_CHECK();
    return 0;
}

template <int MT, int NT>
static int launch_gemm_nt(const GemmArgs& a, hipStream_t st)
{
    // k-steps per LDS batch (UK) is bounded by registers; 7x7 project layers use the deferred-gate variant
    if (MT * NT <= 4) {
        if (MT == 1 && a.defer_gate) return launch_gemm_uk<MT, (MT * NT <= 4 ? NT : 1), 4, (MT == 1)>(a, st);
        return launch_gemm_uk<MT, (MT * NT <= 4 ? NT : 1), 4, false>(a, st);
    }
    return launch_gemm_uk<MT, NT, 2, false>(a, st);
}

template <int NT>
static int launch_gap_nt(const GemmArgs& a, hipStream_t st)
{
    dim3 grid(a.M / a.HW, a.n_chunks, 1);
    hipLaunchKernelGGL((pw_gemm_kernel<1, NT, EPI_GAP, false, false, (NT <= 4 ? 4 : 2), false>), grid, dim3(256), 0, st, a.X, a.M, a.K, a.Wp,
                       a.Kp / 32, a.bias, a.Y, a.N, a.gate, a.HW, a.res, a.gap_out, a.inv_hw);
    LAUNCH_CHECK();
    return 0;
}

int launch_pw_gemm(const GemmArgs& a, hipStream_t st)
{
    if (a.epi == EPI_GAP) {
        if (a.HW > 64) return -2;
        switch (a.nt) {
            case 4: return launch_gap_nt<4>(a, st);
            case 5: return launch_gap_nt<5>(a, st);
            case 8: return launch_gap_nt<8>(a, st);
            default: return -3;
        }
    }
#define CASE_NT(n)                                              \
    case n:                                                     \
        return a.mt == 2 ? launch_gemm_nt<2, n>(a, st) : launch_gemm_nt<1, n>(a, st);
    switch (a.nt) {
        CASE_NT(1)
        CASE_NT(2)
        CASE_NT(3)
        CASE_NT(4)
        CASE_NT(5)
        CASE_NT(6)
        CASE_NT(7)
        CASE_NT(8)
        default: return -3;
    }
#undef CASE_NT
}

int launch_pw_gemm_fp8(const Fp8GemmArgs& a, hipStream_t st)
{
    if (a.M < 1 || (a.K & 7) || (a.N & 3) || a.KS128 * 128 < a.K || a.NFp % 7 || 16 * a.NFp < a.N || !a.gate || a.HW < 1) return -17;
    dim3 grid((a.M + 63) / 64, a.NFp / 7);
    if (a.res) hipLaunchKernelGGL((pw_gemm_fp8_kernel<7, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((pw_gemm_fp8_kernel<7, false>), grid, dim3(256), 0, st, a);
    LAUNCH_CHECK();
    return 0;
}

int thin_proj_has(int ksteps) { return ksteps >= 1 && ksteps <= 6; }   // the k-step counts launch_thin_proj instantiates

int launch_thin_proj(const GemmArgs& a, int patches, hipStream_t st)
{
    // pack_pw layout with nt = 2, one chunk; whole 16-pixel fragments; 8-channel lanes
    if (a.nt != 2 || a.n_chunks != 1 || a.K > 192 || (a.K & 7) || a.N > 32 || (a.N & 7) || (a.HW & 15) || !a.gate || a.epi != EPI_LINEAR ||
        a.M != patches * a.HW || (a.x_plane_rows && (a.x_plane_rows != a.M || (a.K & 31))))
        return -15;
    const int nfrag = a.HW / 16;
    int per = (int)(((long)nfrag * patches / 2048 + 3) / 4 * 4);   // fragments per workgroup: ~2048 workgroups, whole rounds of 4 waves
    if (per < 8) per = 8;
    if (per > 112) per = 112;
    if (nfrag < per) per = nfrag;
    dim3 grid((nfrag + per - 1) / per, patches);
    const int ks = a.Kp / 32;
#define TP_GO(KS_, RES_) hipLaunchKernelGGL((thin_proj_kernel<KS_, RES_>), grid, dim3(256), 0, st, a.X, a.K, a.Wp, a.bias, a.Y, a.N, a.gate, a.HW, per, a.res, a.x_plane_rows)
    if (ks == 1 && a.res) TP_GO(1, true);
    else if (ks == 1) TP_GO(1, false);
    else if (ks == 2 && a.res) TP_GO(2, true);
    else if (ks == 2) TP_GO(2, false);
    else if (ks == 3 && a.res) TP_GO(3, true);
    else if (ks == 3) TP_GO(3, false);
    else if (ks == 4 && a.res) TP_GO(4, true);
    else if (ks == 4) TP_GO(4, false);
    else if (ks == 5 && a.res) TP_GO(5, true);
    else if (ks == 5) TP_GO(5, false);
    else if (ks == 6 && a.res) TP_GO(6, true);
    else if (ks == 6) TP_GO(6, false);
    else return -15;
#undef TP_GO
    LAUNCH_CHECK();
    return 0;
}

template <int KS, int ST, int TW>
static int launch_dw_t(const DwArgs& a, hipStream_t st)
{
    const int nz = a.nz > 0 ? a.nz : 1;
    if (a.CG * nz * 8 != a.C || a.CG * a.S > 256 || a.CG * a.S < 1) return -14;
    dim3 grid(a.parts, a.B, nz);
    dim3 block(a.CG * a.S);
    const size_t shm = (size_t)a.S * a.CG * 8 * sizeof(float);
    hipLaunchKernelGGL((dwconv_kernel<KS, ST, TW>), grid, block, shm, st, a.in, a.wt, a.bias, a.out, a.pool_part, a.H,
                       a.W, a.C, a.Ho, a.Wo, a.pad_t, a.pad_l, a.CG, a.S, a.iters);
    LAUNCH_CHECK();
    return 0;
}

int launch_dwconv(const DwArgs& a, hipStream_t st)
{
#define DW_CASE(KS, ST, TW) \
    if (a.ks == KS && a.stride == ST && a.tw == TW) return launch_dw_t<KS, ST, TW>(a, st);
    DW_CASE(3, 1, 4)
    DW_CASE(3, 1, 2)
    DW_CASE(3, 1, 7)
    DW_CASE(3, 2, 4)
    DW_CASE(3, 2, 2)
    DW_CASE(3, 2, 7)
    DW_CASE(5, 1, 4)
    DW_CASE(5, 1, 2)
    DW_CASE(5, 1, 7)
    DW_CASE(5, 2, 4)
    DW_CASE(5, 2, 2)
    DW_CASE(5, 2, 7)
#undef DW_CASE
    return -4;
}

int launch_se_small(const float* pool_part, int nparts, int B, int C, int Cs, const float* wr, const float* br,
                    const float* we, const float* be, float* gate, hipStream_t st)
{
    if (C > 256 || Cs > 16 || B < 1) return -12;
    hipLaunchKernelGGL(se_small_kernel, dim3(B), dim3(256), 0, st, pool_part, nparts, C, Cs, wr, br, we, be, gate);
    LAUNCH_CHECK();
    return 0;
}

int launch_se_wide(const float* pool_part, int nparts, int B, int C, int Cs, const float* wr, const float* br,
                   const float* we_t, const float* be, float* gate, hipStream_t st)
{
    static const int PB = [] { const char* e = getenv("MMC_SE_PB"); const int v = e ? atoi(e) : 1; return v == 2 || v == 4 ? v : 1; }();
    if (B < 1 || C < 1 || C > 3072 || Cs < 1 || (size_t)PB * (C + Cs) * 4 > 64000) return -12;   // (3072: three channels per thread in the excite FC)
    const dim3 grid((B + PB - 1) / PB);
    const size_t shm = (size_t)PB * (C + Cs) * sizeof(float);
    if (PB == 1) hipLaunchKernelGGL(se_wide_kernel<1>, grid, dim3(1024), shm, st, pool_part, nparts, B, C, Cs, wr, br, we_t, be, gate);
    else if (PB == 2) hipLaunchKernelGGL(se_wide_kernel<2>, grid, dim3(1024), shm, st, pool_part, nparts, B, C, Cs, wr, br, we_t, be, gate);
    else hipLaunchKernelGGL(se_wide_kernel<4>, grid, dim3(1024), shm, st, pool_part, nparts, B, C, Cs, wr, br, we_t, be, gate);
    LAUNCH_CHECK();
    return 0;
}

int launch_se_gate(const float* pool_part, int nparts, int B, int C, int Cs4, const float* WrP, const float* br,
                   const float* WeP, const float* be, float* gate, hipStream_t st)
{
    if (C > 16 * SE_MAXG * 16 || Cs4 > 48 || (C & 15)) return -6;
    const int ng = C / 16;
    int nsplit = (ng + 16 * SE_MAXT - 1) / (16 * SE_MAXT);   // each y-slice covers <= 16 waves * SE_MAXT fragments
    if (ng >= 30 && nsplit < 4) nsplit = 4;                  // big layers: spread the weight stream over 4 CUs
    const dim3 grid((B + 15) / 16, nsplit);
    const int maxg = (ng + 15) / 16;                         // FC1 k-groups per wave
#define SE_LAUNCH(G) hipLaunchKernelGGL((se_fused_kernel<G>), grid, dim3(1024), 0, st, pool_part, nparts, B, C, Cs4, WrP, br, WeP, be, gate)
    if (maxg <= 1) SE_LAUNCH(1);
    else if (maxg <= 2) SE_LAUNCH(2);
    else if (maxg <= 3) SE_LAUNCH(3);
    else SE_LAUNCH(SE_MAXG);
#undef SE_LAUNCH
    LAUNCH_CHECK();
    return 0;
}

int launch_mlp_layer(const float* X, int M, int K, const float* W, const float* bias, float* Y, int N, bool relu,
                     hipStream_t st)
{
    constexpr int NT = 4;
    dim3 grid((M + 63) / 64, (N + 16 * NT - 1) / (16 * NT), 1);
    if (relu)
        hipLaunchKernelGGL((mlp_gemm_f32_kernel<NT, true>), grid, dim3(256), 0, st, X, M, K, W, bias, Y, N);
    else
        hipLaunchKernelGGL((mlp_gemm_f32_kernel<NT, false>), grid, dim3(256), 0, st, X, M, K, W, bias, Y, N);
    LAUNCH_CHECK();
    return 0;
}

int launch_calibrate(const float* logits, int M, int K, const float* a, const float* b, float* proba, int32_t* argmax,
                     hipStream_t st)
{
    hipLaunchKernelGGL(calibrate_kernel, dim3((M + 3) / 4), dim3(256), 0, st, logits, M, K, a, b, proba, argmax);
    LAUNCH_CHECK();
    return 0;
}

int launch_crop(const uint8_t* image, int H, int W, const int32_t* rowcols, int n, uint8_t* out, hipStream_t st)
{
    dim3 grid((224 * 56 + 255) / 256, n, 1);
    hipLaunchKernelGGL(crop_kernel, grid, dim3(256), 0, st, image, H, W, rowcols, out);
    LAUNCH_CHECK();
    return 0;
}

template <int KS, int ST, int TW, int KSTEPS, int NPAIR, int CC, int TWO, int PB>
static int launch_mbconv_t(const MbArgs& a, hipStream_t st)
{
    dim3 grid(a.tiles_x * a.tiles_y, a.Ce / a.CC, (a.B + PB - 1) / PB);
    if (a.wlds) {
        static bool attr_done = false;  // more than the default 64 KB of dynamic LDS
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(
                reinterpret_cast<const void*>(&mbconv_a_kernel<KS, ST, TW, KSTEPS, NPAIR, CC, TWO, PB, true>),
                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            attr_done = true;
        }
        hipLaunchKernelGGL((mbconv_a_kernel<KS, ST, TW, KSTEPS, NPAIR, CC, TWO, PB, true>), grid, dim3(256), a.lds_bytes,
                           st, a.X, a.Wexp, a.bexp, a.Wdw, a.bdw, a.out, a.pool_part, a.H, a.W, a.Cin, a.Ce, a.Ho, a.Wo,
                           a.pad, a.TH, a.tiles_x, a.wl_off, a.red_off, a.B, a.Wfrag, a.wfr_off);
    } else {
        static bool attr_done2 = false;
        if (!attr_done2 && a.lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(
                reinterpret_cast<const void*>(&mbconv_a_kernel<KS, ST, TW, KSTEPS, NPAIR, CC, TWO, PB, false>),
                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            attr_done2 = true;
        }
        hipLaunchKernelGGL((mbconv_a_kernel<KS, ST, TW, KSTEPS, NPAIR, CC, TWO, PB, false>), grid, dim3(256), a.lds_bytes,
                           st, a.X, a.Wexp, a.bexp, a.Wdw, a.bdw, a.out, a.pool_part, a.H, a.W, a.Cin, a.Ce, a.Ho, a.Wo,
                           a.pad, a.TH, a.tiles_x, a.wl_off, a.red_off, a.B, a.Wfrag, a.wfr_off);
    }
    LAUNCH_CHECK();
    return 0;
}

int launch_mbconv_pre(const MbArgs& a, const _Float16* pre_w, const float* pre_b, const float* pre_gate, hipStream_t st)
{
    // block 1 with block 0's squeeze-excite scale + project conv folded in (mbconv_a_kernel, PRE)
    if (!(a.ks == 3 && a.stride == 2 && a.tw == 2 && a.ksteps == 1 && a.npair == 3 && a.CC == 48 && a.TWo == 8 && a.pb == 1 &&
          a.Cin == 32 && !a.wlds))
        return -13;
    dim3 grid(a.tiles_x * a.tiles_y, a.Ce / a.CC, a.B);
    hipLaunchKernelGGL((mbconv_a_kernel<3, 2, 2, 1, 3, 48, 8, 1, false, true>), grid, dim3(256), a.lds_bytes, st, a.X, a.Wexp,
                       a.bexp, a.Wdw, a.bdw, a.out, a.pool_part, a.H, a.W, a.Cin, a.Ce, a.Ho, a.Wo, a.pad, a.TH, a.tiles_x,
                       a.wl_off, a.red_off, a.B, a.Wfrag, a.wfr_off, pre_w, pre_b, pre_gate);
    LAUNCH_CHECK();
    return 0;
}

int launch_mbconv_a(const MbArgs& a, hipStream_t st)
{
    if (a.pb > 1 && (a.tiles_x * a.tiles_y != 1 || a.TH != a.Ho || a.TWo != a.Wo)) return -8;
#define MB_CASE(KS_, ST_, TW_, KSTEPS_, NPAIR_, CC_, TWO_, PB_)                                              \
    if (a.ks == KS_ && a.stride == ST_ && a.tw == TW_ && a.ksteps == KSTEPS_ && a.npair == NPAIR_ &&         \
        a.CC == CC_ && a.TWo == TWO_ && a.pb == PB_)                                                         \
        return launch_mbconv_t<KS_, ST_, TW_, KSTEPS_, NPAIR_, CC_, TWO_, PB_>(a, st);
    MB_CASE(3, 2, 2, 1, 3, 48, 8, 1)     // b1
    MB_CASE(3, 1, 2, 1, 2, 48, 14, 1)    // b2
    MB_CASE(5, 2, 2, 1, 3, 48, 14, 1)    // b3
    MB_CASE(5, 1, 2, 2, 3, 48, 14, 1)    // b4
    MB_CASE(3, 2, 2, 2, 2, 80, 14, 1)    // b5
    MB_CASE(3, 1, 2, 3, 2, 96, 14, 1)    // b6, b7
    MB_CASE(5, 1, 1, 6, 1, 192, 7, 1)    // b12-b14, one patch per workgroup
    MB_CASE(3, 1, 1, 6, 1, 192, 7, 1)    // b15
    MB_CASE(5, 1, 1, 6, 1, 96, 7, 2)     // b12-b14, two patches per workgroup
    MB_CASE(5, 1, 2, 3, 2, 48, 14, 1)    // b8
    MB_CASE(5, 1, 2, 4, 2, 48, 14, 1)    // b9, b10
    MB_CASE(5, 2, 1, 4, 2, 48, 7, 1)     // b11
    MB_CASE(3, 1, 1, 6, 1, 96, 7, 2)     // b15
    MB_CASE(3, 2, 2, 2, 4, 80, 14, 1)    // b5 with a 7x14 output tile: less halo, full depthwise passes (28.4 vs 32.8 us)
    // EfficientNet-B4 (generic_fuse_cfg): blocks 2-9 and 16 reuse the instantiations above
    MB_CASE(3, 2, 2, 2, 2, 48, 14, 1)    // B4 b10
    MB_CASE(3, 1, 2, 4, 2, 96, 14, 1)    // B4 b11-b15
    MB_CASE(5, 1, 2, 5, 2, 48, 14, 1)    // B4 b17-b21
    MB_CASE(5, 1, 2, 5, 2, 96, 14, 1)    // (MMC_B4_CC14=96)
    MB_CASE(5, 1, 2, 4, 2, 96, 14, 1)
    MB_CASE(5, 2, 1, 5, 2, 48, 7, 1)     // B4 b22
    MB_CASE(5, 1, 1, 9, 1, 96, 7, 2)     // B4 b23-b29
    MB_CASE(3, 1, 1, 9, 1, 96, 7, 2)     // B4 b30
    MB_CASE(3, 1, 1, 14, 1, 96, 7, 2)    // B4 b31
#undef MB_CASE
    return -5;
}

int launch_stem_dw(const uint8_t* patches, const _Float16* w, const float* bias, const float* padval, const float* Wdw,
                   const float* bdw, _Float16* out, float* pool_part, int B, hipStream_t st)
{
    hipLaunchKernelGGL(stem_dw_kernel, dim3(7, 7, B), dim3(256), 0, st, patches, w, bias, padval, Wdw, bdw, out, pool_part);
    LAUNCH_CHECK();
    return 0;
}

template <int KS, int ST, int KSTEPS, int NPAIR, int CC, int TWO, int PB>
static int launch_mbconv_d_t(const MbArgs& a, hipStream_t st)
{
    dim3 grid(a.tiles_x * a.tiles_y, a.Ce / a.CC, (a.B + PB - 1) / PB);
    static bool attr_done = false;
    if (!attr_done && a.lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mbconv_d_kernel<KS, ST, KSTEPS, NPAIR, CC, TWO, PB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL((mbconv_d_kernel<KS, ST, KSTEPS, NPAIR, CC, TWO, PB>), grid, dim3(256), a.lds_bytes, st, a.X, a.Wexp,
                       a.bexp, a.Wdw, a.bdw, a.out, a.pool_part, a.H, a.W, a.Cin, a.Ce, a.Ho, a.Wo, a.TH, a.tiles_x,
                       a.wl_off, a.red_off, a.B);
    LAUNCH_CHECK();
    return 0;
}

int launch_mbconv_d(const MbArgs& a, hipStream_t st)
{
    if (a.pb > 1 && (a.tiles_x * a.tiles_y != 1 || a.TH != a.Ho || a.TWo != a.Wo)) return -8;
#define MD_CASE(KS_, ST_, KSTEPS_, NPAIR_, CC_, TWO_, PB_)                                                   \
    if (a.ks == KS_ && a.stride == ST_ && a.ksteps == KSTEPS_ && a.npair == NPAIR_ && a.CC == CC_ &&         \
        a.TWo == TWO_ && a.pb == PB_)                                                                        \
        return launch_mbconv_d_t<KS_, ST_, KSTEPS_, NPAIR_, CC_, TWO_, PB_>(a, st);
    MD_CASE(5, 1, 2, 3, 48, 14, 1)    // b4
    MD_CASE(5, 1, 4, 2, 48, 14, 1)    // b9, b10
    MD_CASE(5, 1, 6, 1, 96, 7, 2)     // b12-b14
#undef MD_CASE
    return -5;
}

int launch_tail7(const TailArgs& a, hipStream_t st)
{
    if (a.nblk < 0 || a.nblk > 4 || a.B < 1) return -9;
    if (a.nblk == 0 && !a.pre_D && !a.pre_X && !a.head_w) return -9;
    if (a.pre_X && (!a.pre_wexp || !a.pre_bexp || !a.pre_dwp || !a.pre_bdw)) return -9;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tail7_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, T7_LDS);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tail7_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, T7_LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    static int tune[4] = {-1, -1, -1, -1};
    if (tune[0] < 0)
        for (int i = 0; i < 4; ++i) {
            char nm[24];
            snprintf(nm, sizeof nm, "MMC_T7_TUNE%d", i);
            const char* e = getenv(nm);
            tune[i] = e ? atoi(e) : 0;
        }
    TailArgs aa = a;
    for (int i = 0; i < 4; ++i) aa.tune[i] = tune[i];
    if (a.dw4) hipLaunchKernelGGL(tail7_kernel<true>, dim3(a.B), dim3(512), T7_LDS, st, aa);
    else hipLaunchKernelGGL(tail7_kernel<false>, dim3(a.B), dim3(512), T7_LDS, st, aa);
    LAUNCH_CHECK();
    return 0;
}

// k-steps of 32 the weight image of proj_patch_kernel is packed with: K rounded up, and 11 -> 12 (the kernel walks K in equal chunks)
int proj_patch_ksteps(int K)
{
    const int ks = (K + 31) / 32;
    return ks == 11 ? 12 : ks;
}

int proj_patch_fc1_rows(int K) { return 64 * ((32 * proj_patch_ksteps(K) + 63) / 64); }

template <int KS, int NF, int HW, bool RES>
static int launch_proj_patch_t(const ProjPatchArgs& a, hipStream_t st)
{
    const int lds = NF * KS * 1024 + 2 * 32 * KS * 4 + 128;   // project weights, pooled + gate vectors, squeeze activations
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&proj_patch_kernel<KS, NF, HW, RES>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL((proj_patch_kernel<KS, NF, HW, RES>), dim3(a.B), dim3(512), lds, st, a);
    LAUNCH_CHECK();
    return 0;
}

int launch_proj_patch(const ProjPatchArgs& a, hipStream_t st)
{
    // CSP <= 28: FC1's output groups are waves 0 .. CSP / 4 - 1 and wave 7 (the DMA path) must not be one of them
    if (a.B < 1 || a.CSP < 4 || a.CSP > 28 || (a.CSP & 3) || a.nparts < 1) return -11;
    const int ks = proj_patch_ksteps(a.K), nf = (a.N + 15) / 16;
#define PP_CASE(KS_, NF_, HW_, RES_) \
    if (ks == KS_ && nf == NF_ && a.HW == HW_ && (a.res != nullptr) == RES_) return launch_proj_patch_t<KS_, NF_, HW_, RES_>(a, st);
    PP_CASE(5, 3, 784, false)    // b3: 144 -> 40 @ 28x28
    PP_CASE(8, 3, 784, true)     // b4: 240 -> 40
    PP_CASE(8, 5, 196, false)    // b5: 240 -> 80 @ 14x14
    PP_CASE(15, 5, 196, true)    // b6, b7: 480 -> 80
    PP_CASE(15, 7, 196, false)   // b8: 480 -> 112
    PP_CASE(21, 7, 196, true)    // b9, b10: 672 -> 112 (and B4 b11-b15)
    PP_CASE(6, 4, 784, false)    // B4 b6: 192 -> 56 @ 28x28
    PP_CASE(12, 4, 784, true)    // B4 b7-b9: 336 -> 56 (10.5 k-steps, padded to 12 = two chunks of 6)
    PP_CASE(12, 7, 196, false)   // B4 b10: 336 -> 112 @ 14x14
#undef PP_CASE
    return -5;
}

// the shapes launch_proj_patch has an instantiation for (the schedule asks before packing a layer for it)
int proj_patch_has(int K, int N, int HW, int res)
{
    const int ks = proj_patch_ksteps(K), nf = (N + 15) / 16;
    static const int T[][4] = {{5, 3, 784, 0}, {8, 3, 784, 1}, {8, 5, 196, 0}, {15, 5, 196, 1}, {15, 7, 196, 0}, {21, 7, 196, 1},
                               {6, 4, 784, 0}, {12, 4, 784, 1}, {12, 7, 196, 0}};
    for (auto& t : T)
        if (t[0] == ks && t[1] == nf && t[2] == HW && t[3] == (res ? 1 : 0)) return 1;
    return 0;
}

template <int CKS, int KSD, int CE, int ST = 1>
static int launch_mid14_t(const Mid14Args& a, hipStream_t st)
{
    // E2 chunk with its zero rows above and below the image (mid14_kernel's ZT + ZB), pool partials of the row bands, scratch
    // words of the expand's masked stores
    constexpr int ZROWS = ST == 1 ? 2 * (KSD / 2) + 1 : 5;
    const int lds = (98 + 7 * ZROWS) * 416 + 5 * 96 * 4 + 512;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mid14_kernel<CKS, KSD, CE, ST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL((mid14_kernel<CKS, KSD, CE, ST>), dim3(a.B, a.nsplit < 1 ? 1 : a.nsplit), dim3(512), lds, st, a);
    LAUNCH_CHECK();
    return 0;
}

template <int CKS, int KSD, int CE>
static int launch_mid14m_t(const Mid14Args& a, hipStream_t st)
{
    const int lds = 196 * (64 * CKS + 16) + 8 * (16 * 736 + 56 * 32);   // staged block input + eight wave-private planar regions and transpose tiles
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mid14m_kernel<CKS, KSD, CE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    // two workgroups per patch: 16 waves share the CE / 16 channel groups (256 workgroups per 128-patch lane: one round)
    hipLaunchKernelGGL((mid14m_kernel<CKS, KSD, CE>), dim3(a.B, a.nsplit < 1 ? 1 : a.nsplit), dim3(512), lds, st, a);
    LAUNCH_CHECK();
    return 0;
}

int launch_mid14(const Mid14Args& a, hipStream_t st)
{
    if (a.B < 1) return -14;
    const int cks = (a.Cin + 31) / 32;
    if (a.dwdiag && a.stride == 1) {   // depthwise on the matrix pipe (mid14m_kernel)
        if (cks == 4 && a.ks == 5 && a.Ce == 672) return launch_mid14m_t<4, 5, 672>(a, st);   // b9, b10
        if (cks == 3 && a.ks == 5 && a.Ce == 480) return launch_mid14m_t<3, 5, 480>(a, st);   // b8
        if (cks == 3 && a.ks == 3 && a.Ce == 480) return launch_mid14m_t<3, 3, 480>(a, st);   // b6, b7
        return -5;
    }
    if (a.stride == 2) {
        if (cks == 4 && a.ks == 5 && a.Ce == 672) return launch_mid14_t<4, 5, 672, 2>(a, st);   // b11
        return -5;
    }
    if (cks == 4 && a.ks == 5 && a.Ce == 672) return launch_mid14_t<4, 5, 672>(a, st);   // b9, b10
    if (cks == 3 && a.ks == 5 && a.Ce == 480) return launch_mid14_t<3, 5, 480>(a, st);   // b8
    if (cks == 3 && a.ks == 3 && a.Ce == 480) return launch_mid14_t<3, 3, 480>(a, st);   // b6, b7
    return -5;
}

int launch_mb1(const Mb1Args& a, hipStream_t st)
{
    if (a.B < 1) return -15;
    const int lds = 62 * 8 * 160 + 16 * 32 * 4;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mb1_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mb1_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    // one workgroup per (tile, patch): walks the three channel chunks
    if (a.planar) hipLaunchKernelGGL(mb1_kernel<true>, dim3(14, 1, a.B), dim3(512), lds, st, a);
    else hipLaunchKernelGGL(mb1_kernel<false>, dim3(14, 1, a.B), dim3(512), lds, st, a);
    LAUNCH_CHECK();
    return 0;
}

template <int KSD, int CKS, int CE, int HIMG>
static int launch_mbt_t(const MbtArgs& a, hipStream_t st)
{
    constexpr int NROWS = 14 + 2 * (KSD / 2), WW = HIMG == 28 ? 28 : 30, NPF = (NROWS * WW + 15) / 16;
    const int lds = NPF * 16 * 112 + 10 * 48 * 4;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mbt_kernel<KSD, CKS, CE, HIMG>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL((mbt_kernel<KSD, CKS, CE, HIMG>), dim3((HIMG / 14) * (HIMG / 28), CE / 48, a.B), dim3(512), lds, st, a);
    LAUNCH_CHECK();
    return 0;
}

template <int CKS, int CE>
static int launch_mbt4_t(const MbtArgs& a, hipStream_t st)
{
    const int lds = 48 * 1160 + 8 * 56 * 32 + 8 * 48 * 4;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mbt4_kernel<CKS, CE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL((mbt4_kernel<CKS, CE>), dim3(2, CE / 48, a.B), dim3(512), lds, st, a);
    LAUNCH_CHECK();
    return 0;
}

template <int KSD, int CKS, int CE, int HIMG>
static int launch_mbt2_t(const MbtArgs& a, hipStream_t st)
{
    constexpr int NROWS = 12 + KSD, WW = KSD == 5 ? 32 : 30, NPF = (NROWS * WW + 15) / 16, NS = (NPF + 7) / 8, HOUT = HIMG / 2;
    const int lds = NS * 64 * 224 + 14 * 48 * 4;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mbt2_kernel<KSD, CKS, CE, HIMG>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL((mbt2_kernel<KSD, CKS, CE, HIMG>), dim3((HOUT / 7) * (HOUT / 14), CE / 48, a.B), dim3(512), lds, st, a);
    LAUNCH_CHECK();
    return 0;
}

int launch_mbt(const MbtArgs& a, hipStream_t st)
{
    if (a.B < 1) return -16;
    if (a.stride == 2) {
        if (a.H == 56 && a.ks == 5 && a.Cin == 24 && a.Ce == 144) return launch_mbt2_t<5, 1, 144, 56>(a, st);   // b3
        if (a.H == 28 && a.ks == 3 && a.Cin == 40 && a.Ce == 240) return launch_mbt2_t<3, 2, 240, 28>(a, st);   // b5
        if (a.H == 56 && a.ks == 5 && a.Cin == 32 && a.Ce == 192) return launch_mbt2_t<5, 1, 192, 56>(a, st);   // B4 b6
        if (a.H == 28 && a.ks == 3 && a.Cin == 56 && a.Ce == 336) return launch_mbt2_t<3, 2, 336, 28>(a, st);   // B4 b10
        return -5;
    }
    if (a.H == 56 && a.ks == 3 && a.Cin == 24 && a.Ce == 144) return launch_mbt_t<3, 1, 144, 56>(a, st);   // b2
    if (a.dwtoe && a.H == 28 && a.ks == 5 && a.Cin == 40 && a.Ce == 240) return launch_mbt4_t<2, 240>(a, st);   // b4, depthwise on 4x4x4 MFMA blocks
    if (a.dwtoe && a.H == 28 && a.ks == 5 && a.Cin == 56 && a.Ce == 336) return launch_mbt4_t<2, 336>(a, st);   // B4 b7-b9
    if (a.H == 28 && a.ks == 5 && a.Cin == 40 && a.Ce == 240) return launch_mbt_t<5, 2, 240, 28>(a, st);   // b4
    if (a.H == 56 && a.ks == 3 && a.Cin == 32 && a.Ce == 192) return launch_mbt_t<3, 1, 192, 56>(a, st);   // B4 b3-b5
    if (a.H == 28 && a.ks == 5 && a.Cin == 56 && a.Ce == 336) return launch_mbt_t<5, 2, 336, 28>(a, st);   // B4 b7-b9
    return -5;
}

// the layer shapes launch_mbt has an instantiation for
int mbt_has(int H, int ks, int stride, int Cin, int Ce)
{
    static const int T[][5] = {{56, 3, 1, 24, 144}, {28, 5, 1, 40, 240}, {56, 5, 2, 24, 144}, {28, 3, 2, 40, 240},
                               {56, 3, 1, 32, 192}, {28, 5, 1, 56, 336}, {56, 5, 2, 32, 192}, {28, 3, 2, 56, 336}};
    for (auto& t : T)
        if (t[0] == H && t[1] == ks && t[2] == stride && t[3] == Cin && t[4] == Ce) return 1;
    return 0;
}
