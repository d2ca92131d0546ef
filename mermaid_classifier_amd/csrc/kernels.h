// kernels.h -- launcher interface between mmc_api.cpp (network schedule, C ABI) and the kernel translation units (k_generic.hip, k_mbconv.hip, k_early.hip, k_mid.hip, k_tail.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { EPI_SILU = 0, EPI_LINEAR = 1, EPI_GAP = 2 };

// sets the thread-local message mmc_last_error() returns and hands back `code` (defined in mmc_api.cpp)
int mmc_fail(int code, const char* fmt, ...);

struct GemmArgs {
    const _Float16* X;   // [M][K] activations (NHWC rows)
    int M, K;
    const _Float16* Wp;  // [n_chunks*16*nt][Kp] row-permuted, zero padded
    int Kp;
    const float* bias;   // [n_chunks*16*nt] natural channel order, zero padded
    _Float16* Y;         // [M][N]
    int N;
    int mt, nt, n_chunks, defer_gate;
    int epi;
    const float* gate;   // [patch][K] or null
    int HW;              // rows per patch
    const _Float16* res; // [M][N] or null
    float* gap_out;      // EPI_GAP: [patches][N]
    int x_plane_rows;    // thin_proj only: X is [K/32][x_plane_rows][32] planes (0: rows of K)
    float inv_hw;
};

// Project conv on fp8 (OCP e4m3) MFMA operands (pw_gemm_fp8_kernel; BASELINE configs[4]): weights quantised on the host with one
// scale per output channel, activations (x * squeeze-excite gate) quantised in the kernel with one scale per pixel row.
struct Fp8GemmArgs {
    const _Float16* X;   // [M][K] depthwise output (NHWC rows), K a multiple of 8
    int M, K;
    const uint8_t* W8;   // [NFp][KS128][64 lanes][32] e4m3: fragment f = channels 16 f .. +15, lane (i, q) holds k = 128 ks + 32 q .. +31 of channel 16 f + i
    int KS128, NFp;      // k-steps of 128 (K zero padded), fragments (padded to whole workgroups of 7)
    const float* sw;     // [16 NFp] weight scales (0 for padding channels)
    const float* bias;   // [16 NFp]
    _Float16* Y;         // [M][N]
    int N;
    const float* gate;   // [patch][K]
    int HW;              // rows per patch
    const _Float16* res; // [M][N] or null
};
int launch_pw_gemm_fp8(const Fp8GemmArgs& a, hipStream_t st);

struct DwArgs {
    const _Float16* in;
    const float* wt;     // [ks*ks][C]
    const float* bias;   // [C]
    _Float16* out;
    float* pool_part;    // [B][parts][C]
    int B, H, W, C, Ho, Wo, pad_t, pad_l;
    int ks, stride, tw;
    int CG, S, iters, parts;   // CG = channel groups (of 8) per workgroup
    int nz;                    // channel splits (gridDim.z): CG * nz * 8 == C
};

struct MbArgs {
    const _Float16* X;      // [B][H][W][Cin]
    const _Float16* Wexp;   // [Ce][32*ksteps] natural rows, zero-padded K
    const _Float16* Wfrag;  // the same weights in MFMA fragment order [Ce/16][ksteps][64][8] (wlds)
    const float* bexp;
    const float* Wdw;       // [ks*ks][Ce]
    const float* bdw;
    _Float16* out;          // [B][Ho][Wo][Ce]
    float* pool_part;       // [B][tiles][Ce]
    int B, H, W, Cin, Ce, Ho, Wo, pad;
    int ks, stride, tw, ksteps, npair, pb;
    int TH, TWo, tiles_x, tiles_y, CC, CCG, S;
    int wl_off, red_off, lds_bytes, wlds, wfr_off;
};

// One 7x7 MBConv block (192 -> 1152 -> cout, depthwise 5x5 or 3x3, stride 1) packed for tail7_kernel.
struct TailBlock {
    const _Float16* wexp;   // [72][6][64][8] expand weights, MFMA fragment order
    const float* bexp;      // [1152]
    const uint32_t* dwp;    // [4][1152][4] dwords: slots 0..14 = depthwise taps as fp16 pairs (slot 3*ky + d; 5x5: (k0,k1),(k2,k3),(k4,0);
                            // 3x3: (k0,k1),(k2,0),0), slot 15 = the bias (fp32 bits); request j of a channel = slots 4j .. 4j+3
    const float* bdw;       // [1152]
    const _Float16* wr_t;   // [18][384][8] squeeze FC: request p of thread (cr, j4) = Wr^T[64p + cr][4 j4 .. +3], Wr^T[64p + 32 + cr][4 j4 .. +3]
                            // (Wr^T = [1152][48], channel-major; unscaled: 1/(49*log2e) is applied in fp32)
    const float* br;        // [48]
    const _Float16* we_t;   // [24][288][8] excite FC: request p of thread t = We^T[2p][4t .. +3], We^T[2p + 1][4t .. +3] (We^T = [48][1152])
    const float* be;        // [1152]
    const _Float16* wproj;  // [cout/16][36][64][8] project weights, MFMA fragment order
    const float* bproj;     // [cout]
    int cout;               // 192 (skip connection, b12..b14) or 320 (b15, no skip; must be the last block)
    int ks;                 // depthwise kernel size: 5 or 3
    const _Float16* dwtoe;  // [72][ks][2][64][4] Toeplitz depthwise fragments of v_mfma_f32_4x4x4_16B_f16 (TailArgs::dw4), or null
};
struct TailArgs {
    const _Float16* X;      // [B][49][192] block input (or [B][49][320] when in_wide); unused with pre_D
    _Float16* Y;            // [B][49][cout of the last block]; unused with head_w
    int B, nblk;            // nblk blocks from the table (0..4)
    const TailBlock* blk;   // device table, nblk consecutive rows
    _Float16* dbg_dw;       // optional [B][49][1152]: depthwise output of the last block run (nblk == 1)
    float* dbg_gate;        // optional [B][1152] (or [B][672] for the pre-block)
    float* dbg_clk;         // optional [B][8]: shader cycles per phase (expand, dw, fc1, fc2, gate, project)
    int clk_sections;       // 0: dbg_clk is [B][8] (one block per launch); 1: [B][8 sections][8] -- section 0 = block 11 (front half, SE,
                            // gate, project), 1..4 = blocks 12..15 (expand, dw, fc1, fc2, gate, project), 5 = head, 6 = whole kernel
    // optional pre-block: second half of block 11 (squeeze-excite + project 672 -> 192, no skip) on its depthwise output
    const _Float16* pre_D;      // [B][49][672] or null
    const float* pre_pool;      // [B][672] pool sums (one tile per patch)
    const _Float16* pre_wr_t;   // [672][28]
    const float* pre_br;        // [28+]
    const _Float16* pre_we_t;   // [28][672]
    const float* pre_be;        // [672]
    // ... or the whole of block 11 from its INPUT (pre_X instead of pre_D / pre_pool): expand 112 -> 672 + depthwise 5x5 stride 2 in LDS
    const _Float16* pre_X;      // [B][196][112] block 11's input, or null
    const _Float16* pre_wexp;   // [42][4][64][8] expand weights, MFMA fragment order
    const float* pre_bexp;      // [672]
    const uint32_t* pre_dwp;    // [4][672][4] depthwise taps as fp16 pairs + bias (layout of TailBlock::dwp)
    const float* pre_bdw;       // [672]
    const _Float16* pre_wproj;  // [12][24][64][8] (k-steps 21..23 zero)
    const float* pre_bproj;     // [192]
    // optional head: conv 320 -> 1280 + swish + average pool -> feat
    const _Float16* head_w;     // [80][10][64][8] or null
    const float* head_b;        // [1280]
    float* feat;                // [B][1280]
    float inv_hw;               // 1 / (49 log2 e)
    int in_wide;                // input X is [B][49][320] (head-only launches)
    int tune[4];                // experiment knobs (env MMC_T7_TUNE0..3, read by launch_tail7); 0 = off
    int dw4;                    // 1: blocks run expand + depthwise as one wave-private phase with the depthwise conv on 4x4x4 MFMA blocks
                                // (every TailBlock::dwtoe set); 0: the round-2 phases
};
int launch_tail7(const TailArgs& a, hipStream_t st);

// Block 1 front half with block 0's SE scale + project folded in (mb1_kernel)
struct Mb1Args {
    const _Float16* X;        // [B][112][112][32] block 0's depthwise output
    const _Float16* pre_w;    // [64][8] block 0's project conv as one MFMA fragment
    const float* pre_b;       // [16]
    const float* pre_gate;    // [B][32] block 0's squeeze-excite gate
    const _Float16* wexp;     // [96][32] expand weights, natural rows, K-permuted (slot 8q+j <- channel 4q+j, j < 4)
    const float* bexp;        // [96]
    const float* wdw;         // [9][96] depthwise taps (fp32, tap-major)
    const float* bdw;         // [96]
    _Float16* D;              // [B][56][56][96], or planar: [3][B * 56 * 56][32]
    int planar;
    float* pool;              // [B][14][96]
    int B;
};
int launch_mb1(const Mb1Args& a, hipStream_t st);

// Front half of a stride-1 MBConv block on 14x28 output tiles (mbt_kernel: b2, b4)
struct MbtArgs {
    const _Float16* X;        // [B][H][H][Cin]
    const _Float16* wexp;     // [Ce/16][ceil(Cin/32)][64][8] expand weights, MFMA fragment order
    const float* bexp;        // [Ce]
    const uint32_t* dwp;      // [15][Ce] depthwise taps as fp16 pairs (layout of TailBlock::dwp)
    const float* bdw;         // [Ce]
    _Float16* D;              // [B][H][H][Ce]
    float* pool;              // [B][tiles][Ce]
    int B, H, Cin, Ce, ks;
    int stride;               // 1 (b2, b4: D is [B][H][H][Ce]) or 2 (b3, b5: D is [B][H/2][H/2][Ce], mbt2_kernel)
    const _Float16* dwtoe;    // optional [Ce/16][ks][2][64][4]: Toeplitz depthwise fragments -> mbt4_kernel (5x5 stride 1 at 28x28)
};
int launch_mbt(const MbtArgs& a, hipStream_t st);
int thin_proj_has(int ksteps);                          // 1 when launch_thin_proj has an instantiation for this many k-steps
int mbt_has(int H, int ks, int stride, int Cin, int Ce);   // 1 when launch_mbt has an instantiation for this layer

// Front half of a 14x14 MBConv block for one patch per workgroup (mid14_kernel)
struct Mid14Args {
    const _Float16* X;        // [B][196][Cin]
    const _Float16* wexp;     // [Ce/16][ceil(Cin/32)][64][8] expand weights, MFMA fragment order (K zero padded)
    const float* bexp;        // [Ce]
    const uint32_t* dwp;      // [4][Ce][4] depthwise taps as fp16 pairs + bias in 16-byte requests (layout of TailBlock::dwp)
    const float* bdw;         // [Ce]
    _Float16* D;              // [B][196][Ce] depthwise output
    float* pool;              // [B][Ce] pool sums
    int B, Cin, Ce, ks;
    int nsplit;               // workgroups per patch (each takes every nsplit-th chunk of 96 channels)
    int stride;               // 1, or 2 (block 11: D is [B][49][Ce])
    float* dbg_clk;           // optional [B][8 workgroups][16]: shader cycles of the first chunk's phases (MMC_TAIL_CLK=1)
    const _Float16* dwdiag;   // optional [Ce/16][ks][2][64][4]: Toeplitz depthwise fragments of v_mfma_f32_4x4x4_16B_f16 -> mid14m_kernel
};
int launch_mid14(const Mid14Args& a, hipStream_t st);

// Squeeze-excite + project conv of one patch per workgroup (proj_patch_kernel)
struct ProjPatchArgs {
    const _Float16* X;        // [B][HW][K] depthwise output
    const float* pool_part;   // [B][nparts][K] depthwise pool partials
    const _Float16* wr_g;     // [CSP/4][proj_patch_fc1_rows(K)][4] squeeze FC (fp16; group of four outputs, channel, output; zero beyond K)
    const float* br;          // [CSP]
    const _Float16* we_t;     // [CSP][K] excite FC (fp16)
    const float* be;          // [K]
    const _Float16* wfrag;    // [N/16 up][K/32 up][64][8] project weights, MFMA fragment order, zero padded
    const float* bias;        // [16 * ceil(N/16)]
    const _Float16* res;      // [B][HW][N] skip input or null
    _Float16* Y;              // [B][HW][N]
    float* dbg_gate;          // optional [B][K]
    float* dbg_clk;           // optional [B][8]: shader cycles of prologue, GEMM
    int B, HW, K, N, CSP, nparts;
    float psc;                // 1 / (HW * log2 e)
};
int launch_proj_patch(const ProjPatchArgs& a, hipStream_t st);
int proj_patch_ksteps(int K);                        // k-steps (of 32) its weight image must be packed with
int proj_patch_fc1_rows(int K);                      // channel rows per output group of ProjPatchArgs::wr_g (64 x the kernel's FC1 iterations)
int proj_patch_has(int K, int N, int HW, int res);   // 1 when launch_proj_patch has an instantiation for this layer shape

int launch_mbconv_a(const MbArgs& a, hipStream_t st);
// block 1 reading block 0's depthwise output, with block 0's SE scale + project conv folded in
int launch_mbconv_pre(const MbArgs& a, const _Float16* pre_w, const float* pre_b, const float* pre_gate, hipStream_t st);
int launch_mbconv_d(const MbArgs& a, hipStream_t st);   // dot2 depthwise variant (pair-interleaved LDS tile)
int launch_stem(const uint8_t* patches, const _Float16* w, const float* bias, const float* padval, _Float16* out, int B,
                int channels, hipStream_t st);
int launch_stem_dw(const uint8_t* patches, const _Float16* w, const float* bias, const float* padval, const float* Wdw,
                   const float* bdw, _Float16* out, float* pool_part, int B, hipStream_t st);
int launch_pw_gemm(const GemmArgs& a, hipStream_t st);
// SE-scale + project (+ skip) for K <= 64, N <= 32 (pack_pw weights with nt = 2): one patch's pixel fragments streamed per workgroup
int launch_thin_proj(const GemmArgs& a, int patches, hipStream_t st);
int launch_dwconv(const DwArgs& a, hipStream_t st);
int launch_se_gate(const float* pool_part, int nparts, int B, int C, int Cs4, const float* WrP, const float* br,
                   const float* WeP, const float* be, float* gate, hipStream_t st);
int launch_se_small(const float* pool_part, int nparts, int B, int C, int Cs, const float* wr, const float* br,
                    const float* we, const float* be, float* gate, hipStream_t st);
int launch_se_wide(const float* pool_part, int nparts, int B, int C, int Cs, const float* wr, const float* br,
                   const float* we_t /* [Cs][C] */, const float* be, float* gate, hipStream_t st);
int launch_mlp_layer(const float* X, int M, int K, const float* W, const float* bias, float* Y, int N, bool relu,
                     hipStream_t st);
int launch_calibrate(const float* logits, int M, int K, const float* a, const float* b, float* proba, int32_t* argmax,
                     hipStream_t st);
int launch_crop(const uint8_t* image, int H, int W, const int32_t* rowcols, int n, uint8_t* out, hipStream_t st);
