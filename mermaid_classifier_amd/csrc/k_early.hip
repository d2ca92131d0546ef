// k_early.hip -- the tiled kernels of EfficientNet-B0 blocks 0..5 (112x112 .. 28x28): stem_dw, mb1, mbt, mbt2, mbt4, thin_proj.
// gfx950 only.
#include "device_common.h"

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

int launch_stem_dw(const uint8_t* patches, const _Float16* w, const float* bias, const float* padval, const float* Wdw,
                   const float* bdw, _Float16* out, float* pool_part, int B, hipStream_t st)
{
    hipLaunchKernelGGL(stem_dw_kernel, dim3(7, 7, B), dim3(256), 0, st, patches, w, bias, padval, Wdw, bdw, out, pool_part);
    LAUNCH_CHECK();
    return 0;
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

