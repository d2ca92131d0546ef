// k_tail.hip -- tail7_kernel: block 11's back half, blocks 12..15 and the head conv of one patch per workgroup, chained in LDS.
// gfx950 only.
#include "device_common.h"

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

// =============================================================================================
// Host-side launchers (plain C++ signatures declared in kernels.h)
// =============================================================================================
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

