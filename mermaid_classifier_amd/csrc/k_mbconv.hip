// k_mbconv.hip -- the generic fused MBConv front halves (expand 1x1 + depthwise + pool sums per output tile and channel chunk):
// mbconv_a_kernel (v_fma_mix taps; EfficientNet-B4 and the MMC_FUSE fallbacks of B0) and mbconv_d_kernel (v_dot2c taps).  gfx950 only.
#include "device_common.h"

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
                {
                    float sv[8];   // staged over the pair's eight values (packed adds / products, no dependent chain per value): same bits
#pragma unroll
                    for (int j = 0; j < 4; ++j) { sv[j] = a0[j]; sv[4 + j] = a1[j]; }
                    silu_scaled_staged(sv);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { o0[j] = (_Float16)sv[j]; o1[j] = (_Float16)sv[4 + j]; }
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
                    silu_scaled_staged(acc[t]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float y = acc[t][j];
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

// =============================================================================================
// Host-side launchers (plain C++ signatures declared in kernels.h)
// =============================================================================================
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


