// k_generic.hip -- hand-written CDNA4 (gfx950) kernels for the EfficientNet-B0 patch
// feature-extraction path.  gfx950 only: 64-wide wavefronts, v_mfma_f32_16x16x32_f16,
// v_mfma_f32_16x16x4_f32, LDS.  No CUDA compatibility paths.
//
// k_generic.hip: the per-layer kernels of the generic schedule (EfficientNet-B4, MMC_FUSE=0) and the head: stem_conv, pw_gemm,
// pw_gemm_fp8, dwconv, squeeze-excite, the calibrated MLP head, crop (mbconv_a / mbconv_d: k_mbconv.hip).
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
#include "device_common.h"

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
                silu_scaled_staged(acc[t]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float y = acc[t][j];
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

// =============================================================================================
// Host-side launchers (plain C++ signatures declared in kernels.h)
// =============================================================================================
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
