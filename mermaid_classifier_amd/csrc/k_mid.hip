// k_mid.hip -- one patch per workgroup at 28x28 / 14x14 (blocks 3..10): mid14, mid14m (front halves), proj_patch (squeeze-excite +
// project).  gfx950 only.
#include "device_common.h"

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
    // other waves' work between two barriers covers; wave 7 does nothing else in the prologue (FC1's reduction happens inside the FC1 waves).
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
    f4 bv[NF];   // project bias of this lane's four output channels per fragment
    const GLOBAL_AS float* biasg = sgpr_ptr<float>(a.bias);
    auto load_bias = [&]() {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) bv[nf] = gload<f4>(biasg, (unsigned)(16 * nf + 4 * q) * 4u);
    };
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
        load_bias();
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
        load_bias();   // (the project's bias, behind the chain's inputs: requested at the start of the GEMM it was an exposed round trip)
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

// =============================================================================================
// Host-side launchers (plain C++ signatures declared in kernels.h)
// =============================================================================================
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

