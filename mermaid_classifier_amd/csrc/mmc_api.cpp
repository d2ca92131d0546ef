// mmc_api.cpp -- C ABI (include/mmc.h) and the EfficientNet-B0 launch schedule.
// Host C++ only; kernels live in k_*.hip.  No torch, no CUDA shims.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/mmc.h"
#include "kernels.h"

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(MMC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));    \
    } while (0)
#define KTRY(expr)                                                                                 \
    do {                                                                                           \
        int r_ = (expr);                                                                           \
        if (r_ != 0) return fail(MMC_ERR_HIP, "%s failed (%d: %s)", #expr, r_,                     \
                                 r_ > 0 ? hipGetErrorString((hipError_t)r_) : "unsupported shape"); \
    } while (0)

// error plumbing for the other translation units (trainer.hip)
int mmc_fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* mmc_last_error(void) { return g_err; }
extern "C" int mmc_version(void) { return 1; }
extern "C" int mmc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------------------
// EfficientNet-B0 stage table (the published architecture; mirrors oracle/efficientnet_b0_ref.py)
// ------------------------------------------------------------------------------------------
struct BlockDef { int k, s, e, cin, cout; };
static const BlockDef B0_BLOCKS[16] = {
    {3, 1, 1, 32, 16},  {3, 2, 6, 16, 24},  {3, 1, 6, 24, 24},   {5, 2, 6, 24, 40},
    {5, 1, 6, 40, 40},  {3, 2, 6, 40, 80},  {3, 1, 6, 80, 80},   {3, 1, 6, 80, 80},
    {5, 1, 6, 80, 112}, {5, 1, 6, 112, 112}, {5, 1, 6, 112, 112}, {5, 2, 6, 112, 192},
    {5, 1, 6, 192, 192}, {5, 1, 6, 192, 192}, {5, 1, 6, 192, 192}, {3, 1, 6, 192, 320}};
static const int IMG = 224;
// The published family: B0's stage table under compound scaling (widths to multiples of 8, never below 90 % of the scaled
// value; repeats rounded up).  B0 is the network the reference path runs; B4 (width 1.4, depth 1.8, BASELINE.json
// configs[4]) does not exist in the reference and runs on the generic per-layer kernels.
struct ArchDef { int stem = 32, head_in = 320, feat = 1280; std::vector<BlockDef> blocks; };
static int round_filters(int c, double width)
{
    const double v = c * width;
    int n = (int)(v + 4) / 8 * 8;
    if (n < 8) n = 8;
    if (n < 0.9 * v) n += 8;
    return n;
}
static ArchDef make_arch(int arch)
{
    ArchDef A;
    if (arch == MMC_ARCH_B0) {
        A.blocks.assign(B0_BLOCKS, B0_BLOCKS + 16);
        return A;
    }
    const double width = 1.4, depth = 1.8;   // MMC_ARCH_B4
    static const int STAGES[7][6] = {{1, 3, 1, 1, 32, 16},  {2, 3, 2, 6, 16, 24},   {2, 5, 2, 6, 24, 40}, {3, 3, 2, 6, 40, 80},
                                     {3, 5, 1, 6, 80, 112}, {4, 5, 2, 6, 112, 192}, {1, 3, 1, 6, 192, 320}};
    A.stem = round_filters(32, width);
    for (auto& st : STAGES) {
        const int cin = round_filters(st[4], width), cout = round_filters(st[5], width);
        const int reps = (int)std::ceil(depth * st[0] - 1e-9);
        for (int r = 0; r < reps; ++r) A.blocks.push_back({st[1], r == 0 ? st[2] : 1, st[3], r == 0 ? cin : cout, cout});
    }
    A.head_in = A.blocks.back().cout;
    A.feat = round_filters(1280, width);
    return A;
}
// Scaled activation domain (device_common.h, silu_scaled): every SiLU output is stored times log2(e).
// Producers (stem, expand, depthwise, head) get weights/bias times LOG2E, consumers times 1/LOG2E;
// for the depthwise taps the two cancel, so only its bias is scaled.
static const double LOG2E = 1.4426950408889634;

static void same_pad(int size, int k, int s, int* before, int* out)
{
    *out = (size + s - 1) / s;
    int pad = (*out - 1) * s + k - size;
    if (pad < 0) pad = 0;
    *before = pad / 2;
}

// A 1x1 convolution packed for pw_gemm_kernel
struct PwLayer {
    int N = 0, K = 0, Kp = 0, nt = 0, n_chunks = 0;
    _Float16* w = nullptr;  // device [n_chunks*16*nt][Kp]
    float* b = nullptr;     // device [n_chunks*16*nt]
};

// A project conv with e4m3 weights for pw_gemm_fp8_kernel (MMC_PRECISION_FP8)
struct Fp8Layer {
    int N = 0, K = 0, KS128 = 0, NFp = 0;
    uint8_t* w8 = nullptr;  // device [NFp][KS128][64][32]
    float* sw = nullptr;    // device [16 NFp] per-output-channel scales
    float* b = nullptr;     // device [16 NFp]
};

// fp32 -> OCP e4m3fn (4 exponent bits, bias 7, 3 mantissa bits, no infinities, largest finite 448), round to nearest even,
// saturating.  The device side uses v_cvt_pk_fp8_f32; the host needs the same encoding for the weights.
static uint8_t e4m3_encode(float x)
{
    if (x != x) return 0x7F;
    const uint8_t sign = std::signbit(x) ? 0x80 : 0x00;
    float a = std::fabs(x);
    if (a >= 448.0f) return sign | 0x7E;
    if (a < 0.0009765625f) return sign;                        // below half of the smallest subnormal (2^-9): zero (a tie goes to even = 0)
    int e;
    (void)std::frexp(a, &e);                                   // a = f * 2^e, f in [0.5, 1)
    int ex = e - 1;                                            // a in [2^ex, 2^(ex+1))
    if (ex < -6) ex = -6;                                      // subnormals share the exponent of the smallest normal
    const float quantum = std::ldexp(1.0f, ex - 3);
    float qf = std::nearbyint(a / quantum);                    // round to nearest even (default rounding mode)
    int qi = (int)qf;
    if (qi >= 16) { qi = 8; ++ex; }
    if (ex > 8) return sign | 0x7E;
    if (ex == 8 && qi > 14) return sign | 0x7E;                // 448 = 1.75 * 2^8 is the largest finite value
    if (qi < 8) return sign | (uint8_t)qi;                     // subnormal (only with ex == -6)
    return sign | (uint8_t)(((ex + 7) << 3) | (qi - 8));
}

extern "C" int mmc_fp8_e4m3_encode(const float* in, uint8_t* out, size_t n)
{
    if (!in || !out) return fail(MMC_ERR_ARG, "NULL argument");
    for (size_t i = 0; i < n; ++i) out[i] = e4m3_encode(in[i]);
    return MMC_OK;
}

// Channel fragments (16 wide) per workgroup.  Big-M layers (early blocks) take the widest chunk that
// divides N (X is read once per chunk).  Small-M layers (14x14 and 7x7 blocks, head) take narrow
// chunks: more workgroups and registers left for a 4-step-deep fragment prefetch; X re-reads hit L2.
static int pick_nt(int N, bool small_m)
{
    const int tiles = (N + 15) / 16;
    if (small_m) {
        int best = 4, waste = 1 << 30;
        for (int nt = 4; nt >= 2; --nt) {
            const int w = (tiles + nt - 1) / nt * nt - tiles;
            if (w < waste) { waste = w; best = nt; }
        }
        return tiles <= 4 ? tiles : best;
    }
    for (int nt = 8; nt >= 1; --nt)
        if (tiles % nt == 0) return nt;
    return 1;
}

struct BlockW {
    BlockDef d;
    int H = 0, Ho = 0, ce = 0, cs = 0, cs4 = 0, pad = 0;
    bool has_expand = false, skip = false;
    PwLayer expand, project;
    Fp8Layer p8;             // the project conv on fp8 operands (MMC_PRECISION_FP8, blocks from fp8_from on)
    float *dw_w = nullptr, *dw_b = nullptr;                    // [k*k][ce], [ce]
    float *se_br = nullptr, *se_be = nullptr;
    float *se_wrp = nullptr, *se_wep = nullptr;   // fragment-ordered fp32 squeeze-excite weights
    float *se_wr_nat = nullptr, *se_we_nat = nullptr, *se_br_nat = nullptr;   // natural fp32 copies for se_small_kernel (C <= 256)
    // depthwise launch geometry
    int tw = 0, CG = 0, S = 0, iters = 0, parts = 0, nz = 1;
    // fused expand+depthwise (mbconv_a_kernel) geometry; fused == false -> separate GEMM + dwconv
    bool fused = false;
    _Float16* exp_nat = nullptr;  // [ce][32*f_ksteps] natural rows
    int f_TH = 0, f_TWo = 0, f_CC = 0, f_tw = 0, f_ksteps = 0, f_CCG = 0, f_S = 0, f_tiles_x = 0, f_tiles_y = 0,
        f_red_off = 0, f_lds = 0, f_npair = 0, f_wl_off = 0, f_pb = 1, f_wlds = 0, f_wfr_off = 0;
    // geometry of the dot2 variant (mbconv_d_kernel): pair-aligned windows are a little wider
    bool use_d = false;
    int d_npair = 0, d_wl_off = 0, d_red_off = 0, d_lds = 0;
    _Float16* exp_frag = nullptr;  // expand weights in MFMA fragment order
    // tail7_kernel packing (7x7 blocks 12..14): project weights in plain fragment order, depthwise tap pairs
    _Float16* t_wproj = nullptr;
    uint32_t* t_dwp = nullptr;
    uint32_t* t_dwp4 = nullptr;  // taps + bias, [4][ce][4] dwords (16-byte requests)
    _Float16* dw_diag = nullptr;  // mid14m_kernel: Toeplitz depthwise fragments [ce/16][k][2][64][4] for v_mfma_f32_4x4x4_16B_f16 (depthwise on the matrix pipe)
    _Float16 *t_wr = nullptr, *t_we = nullptr;   // squeeze-excite FCs transposed (fp16) for matrix-vector use
    _Float16* t_wrg = nullptr;                   // ... the squeeze FC as proj_patch_kernel reads it (ProjPatchArgs::wr_g)
    _Float16 *t_wr2 = nullptr, *t_we2 = nullptr; // ... blocks 12-15: paired rows for tail7_kernel's 16-byte requests
    // proj_patch_kernel packing (blocks 3..10): project weights/bias padded to whole fragments, SE FCs as above with
    // Cs padded to a multiple of 4
    bool pp = false;
    _Float16* pp_w = nullptr;
    float *pp_b = nullptr, *pp_br = nullptr;
    int pp_csp = 0;
};

// Output tile (TH x TWo) and channel chunk CC of the fused kernel, per B0 block (index 1..15):
// chosen so that E[P][CC] + the pool scratch stay <= 64 KB of LDS (>= 2 workgroups per CU) while
// the halo recompute and the per-chunk re-read of the (small) block input stay low.
struct FuseCfg { int TH, TWo, CC, TW, PB, WLDS; };
static const FuseCfg B0_FUSE[16] = {
    {0, 0, 0, 0, 1, 0},     {8, 8, 48, 2, 1, 0},    {14, 14, 48, 2, 1, 0},  {4, 14, 48, 2, 1, 0},   {14, 14, 48, 2, 1, 0},
    {7, 14, 80, 2, 1, 0},   {14, 14, 96, 2, 1, 0},  {14, 14, 96, 2, 1, 0},  {14, 14, 48, 2, 1, 0},  {14, 14, 48, 2, 1, 0},
    {14, 14, 48, 2, 1, 0},  {7, 7, 48, 1, 1, 0},    {7, 7, 96, 1, 2, 0},    {7, 7, 96, 1, 2, 0},    {7, 7, 96, 1, 2, 0},
    {7, 7, 96, 1, 2, 0}};

// The same choices by layer geometry for the other members of the family (B4): the tile and chunk B0 uses at that
// resolution / stride / kernel size, with CC a divisor of the expanded width.
static FuseCfg generic_fuse_cfg(int H, int k, int s, int ce, int cc5)
{
    FuseCfg fc{0, 0, 0, 0, 1, 0};
    if (H == 112 && s == 2) fc = {8, 8, 48, 2, 1, 0};
    else if (H == 56 && s == 1) fc = {14, 14, 48, 2, 1, 0};
    else if (H == 56 && s == 2) fc = {4, 14, 48, 2, 1, 0};
    else if (H == 28 && s == 1) fc = {14, 14, 48, 2, 1, 0};
    else if (H == 28 && s == 2) fc = {2, 14, 48, 2, 1, 0};
    else if (H == 14 && s == 1) fc = {14, 14, k == 3 ? 96 : cc5, 2, 1, 0};
    else if (H == 14 && s == 2) fc = {7, 7, 48, 1, 1, 0};
    else if (H == 7 && s == 1) fc = {7, 7, 96, 1, 2, 0};
    if (fc.CC && ce % fc.CC) fc.TH = 0;
    return fc;
}

struct Saved {
    void* dev = nullptr;
    size_t bytes = 0;
    size_t elems = 0;
    bool is_half = true;
};

#define MMC_GRAPH_CACHE 32   // captured (input, output, n) combinations kept per handle

struct mmc_backbone {
    int device = 0, max_batch = 0;
    int arch = MMC_ARCH_B0, nblk = 16, stem_ch = 32, head_in = 320, feat = 1280;
    _Float16* stem_w = nullptr;
    float *stem_b = nullptr, *stem_pad = nullptr;
    std::vector<BlockW> blk;
    PwLayer head;
    // workspace: one lane per internal stream.  A pass over n patches is split into `nlanes` independent
    // sub-batches that run concurrently on their own HIP streams, so the latency-bound small launches of
    // one lane (squeeze-excite FCs, 7x7 layers) overlap the bandwidth/VALU-bound launches of the other.
    struct Lane {
        _Float16 *act0 = nullptr, *act1 = nullptr, *expbuf = nullptr, *dwbuf = nullptr;
        float *pool_part = nullptr, *gate = nullptr;
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
    };
    Lane lanes[4];
    // optional (MMC_GRAPH=1): a pass over device-resident buffers is captured once per (input, output, n) into a HIP graph
    // and replayed -- the ~26 launches per lane then cost one graph launch
    // (least recently used entry evicted beyond MMC_GRAPH_CACHE combinations)
    struct GraphEntry { const void* in; float* out; int n; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;
    bool use_graph = false;
    int graph_warm = 0;
    // a (buffers, n) combination is captured the second time it is seen (ring of recent combinations): callers that
    // reuse their buffers (bench, BatchedExtractor's chunks, torch's caching allocator) get graphs, others plain launches
    struct SeenKey { const void* in; float* out; int n; };
    std::vector<SeenKey> seen;
    // combinations whose graph was evicted are NOT captured again (they run as plain launches from then on): a caller that
    // cycles through more combinations than the cache holds would otherwise pay capture + instantiate + destroy on every call.
    // After one full turnover of the cache (MMC_GRAPH_CACHE evictions) no new combination is captured at all.
    std::vector<SeenKey> evicted;
    int evictions = 0;
    long long captures = 0;
    hipStream_t gstream = nullptr;
    int nlanes = 1, lane_cap = 0;
    hipEvent_t fork = nullptr;
    uint8_t* in_stage = nullptr;
    float* out_stage = nullptr;
    size_t ws_bytes = 0;
    std::vector<void*> allocs;
    bool keep = false, fuse_stem = false;
    float* dbg_clk = nullptr;        // keep mode: per-patch phase cycle counts of the patch-resident kernels
    float* mid_clk = nullptr;        // MMC_TAIL_CLK=1: [max_batch][8 workgroups][16] phase cycle counts of block 10's mid14 launch
    float* tail_clk = nullptr;       // MMC_TAIL_CLK=1: [max_batch][8 sections][8] phase cycle counts of the production tail7 launch
    // tail7 extensions: block 11's squeeze-excite + project (pre-block) and the head conv inside the same launch
    // block 0's SE scale + project conv folded into block 1's fused kernel (mbconv_a_kernel PRE): no b0 output tensor
    bool fuse_b0b1 = false;
    bool mbt = false;                // blocks 2 and 4 on mbt_kernel (tiled, window-in-registers depthwise)
    bool mbt2 = false;               // ... and the stride-2 blocks 3 and 5 on mbt2_kernel
    bool mb1 = false;                // block 1 on mb1_kernel (window-in-registers depthwise) instead of mbconv_a PRE
    bool mid14 = false;              // 14x14 blocks: per-patch front half (mid14_kernel) instead of tile/chunk workgroups
    int mid14_last = 8;              // ... for blocks 6..mid14_last
    bool mid14_b11 = false;          // ... and block 11 (5x5 stride 2) on mid14_kernel<4,5,672,2>
    bool b1_planar = true;           // block 1's depthwise output as 32-channel planes between mb1 and thin_proj (MMC_B1_PLANAR=0)
    bool thin_proj = true;           // B4 blocks 0/1: thin_proj_kernel instead of pw_gemm for the tiny-K project convs (MMC_THIN_PROJ=0)
    bool se_small = true;            // light per-patch squeeze-excite kernel for the early blocks (MMC_SE_SMALL=0: se_fused)
    _Float16 *b0_pre_w = nullptr, *b1_exp_pre = nullptr;
    bool tail_full = false;
    bool tail_b11 = false;           // block 11's front half (expand + depthwise stride 2) inside tail7_kernel too: no b11 launch at all
    _Float16 *pre_wproj = nullptr, *head_wfrag = nullptr;
    bool fp8 = false;                // MMC_PRECISION_FP8: project convs of blocks fp8_from .. on e4m3 MFMA operands
    int fp8_from = 0;
    TailBlock* tail_tab = nullptr;   // device table for tail7_kernel (blocks 12..14), null = separate launches
    bool tail_dw4 = true;            // every tail block has its Toeplitz depthwise fragments (only with MMC_TAIL_DW4=1)
    std::map<std::string, Saved> saved;
    int last_n = 0;
    // One pass at a time per handle (lane workspaces, fork/done events and the staging buffers are shared): calls are
    // serialised on the host by `mu`, and a call on a different stream than the previous one waits for that one's work.
    std::mutex mu;
    hipEvent_t last_done = nullptr;
    hipStream_t last_stream = nullptr;
    bool have_last = false;
};

// serialise against the previous call's device work when the caller switches streams; record this call's end
struct PassOrder {
    mmc_backbone* bb; hipStream_t st; bool armed = false;
    int begin()
    {
        if (bb->have_last && bb->last_stream != st) HIP_TRY(hipStreamWaitEvent(st, bb->last_done, 0));
        armed = true;
        return 0;
    }
    ~PassOrder()
    {
        if (!armed) return;
        if (!bb->last_done && hipEventCreateWithFlags(&bb->last_done, hipEventDisableTiming) != hipSuccess) { bb->last_done = nullptr; return; }
        if (hipEventRecord(bb->last_done, st) == hipSuccess) { bb->last_stream = st; bb->have_last = true; }
    }
};

struct ProfEntry { std::string name; hipEvent_t e0, e1; };
struct Prof { std::vector<ProfEntry> entries; };

// ------------------------------------------------------------------------------------------
// blob parsing: header{magic,version,arch,n_tensors} + table{offset,nbytes} + fp32 tensors
// ------------------------------------------------------------------------------------------
struct BlobReader {
    const uint8_t* base;
    size_t nbytes;
    uint32_t n_tensors;
    const uint64_t* table;
    uint32_t next = 0;
    const float* take(size_t n_floats, const char* what, int* err)
    {
        if (next >= n_tensors) { *err = fail(MMC_ERR_WEIGHTS, "weights blob ended before %s", what); return nullptr; }
        const uint64_t off = table[2 * next], nb = table[2 * next + 1];
        if (off + nb > nbytes || nb != n_floats * sizeof(float)) {
            *err = fail(MMC_ERR_WEIGHTS, "tensor %u (%s): expected %zu bytes, blob has %llu", next, what,
                        n_floats * sizeof(float), (unsigned long long)nb);
            return nullptr;
        }
        ++next;
        return reinterpret_cast<const float*>(base + off);
    }
};

template <typename T>
static int dev_alloc(mmc_backbone* bb, T** p, size_t count)
{
    void* d = nullptr;
    hipError_t e = hipMalloc(&d, count * sizeof(T) + 256);  // +256: vector tail reads stay in-bounds
    if (e != hipSuccess) return fail(MMC_ERR_NOMEM, "hipMalloc(%zu bytes): %s", count * sizeof(T), hipGetErrorString(e));
    bb->allocs.push_back(d);
    bb->ws_bytes += count * sizeof(T);
    *p = reinterpret_cast<T*>(d);
    return 0;
}

template <typename T>
static int dev_upload(mmc_backbone* bb, T** p, const std::vector<T>& host)
{
    int r = dev_alloc(bb, p, host.size());
    if (r) return r;
    HIP_TRY(hipMemcpy(*p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

// Pack a natural [N][K] fp32 1x1-conv weight into the fragment-ordered fp16 layout of pw_gemm_kernel:
// fragment (chunk, kstep, t) = 1 KB at ((chunk*KS32 + kstep)*nt + t)*512 halves; inside it lane
// (q*16 + m) holds W[channel(chunk, t, m)][kstep*32 + q*8 .. +8], with the row permutation
// fragment row (t*16 + 4qr + jr) <- channel (chunk*16nt + qr*4nt + 4t + jr).
static int pack_pw(mmc_backbone* bb, PwLayer* L, const float* w, const float* b, int N, int K, int force_nt,
                   double wscale, double bscale)
{
    L->N = N;
    L->K = K;
    L->Kp = (K + 31) / 32 * 32;
    L->nt = force_nt > 0 ? force_nt : pick_nt(N, false);
    const int cw = 16 * L->nt;
    L->n_chunks = (N + cw - 1) / cw;
    const int Np = L->n_chunks * cw;
    const int ks32 = L->Kp / 32;
    std::vector<_Float16> wp((size_t)Np * L->Kp, (_Float16)0.0f);
    std::vector<float> bp(Np, 0.0f);
    for (int ch = 0; ch < L->n_chunks; ++ch)
        for (int ks = 0; ks < ks32; ++ks)
            for (int t = 0; t < L->nt; ++t)
                for (int m = 0; m < 16; ++m) {
                    const int c = ch * cw + (m >> 2) * 4 * L->nt + 4 * t + (m & 3);
                    if (c >= N) continue;
                    for (int q = 0; q < 4; ++q)
                        for (int j = 0; j < 8; ++j) {
                            const int k = ks * 32 + q * 8 + j;
                            if (k >= K) continue;
                            const size_t off = ((((size_t)ch * ks32 + ks) * L->nt + t) * 64 + (q * 16 + m)) * 8 + j;
                            wp[off] = (_Float16)(float)(w[(size_t)c * K + k] * wscale);
                        }
                }
    for (int c = 0; c < N; ++c) bp[c] = (float)(b[c] * bscale);
    int r = dev_upload(bb, &L->w, wp);
    if (r) return r;
    return dev_upload(bb, &L->b, bp);
}

extern "C" void mmc_backbone_destroy(mmc_backbone* bb)
{
    if (!bb) return;
    hipSetDevice(bb->device);
    for (void* p : bb->allocs) hipFree(p);
    for (auto& kv : bb->saved) hipFree(kv.second.dev);
    for (int l = 0; l < 4; ++l) {
        if (bb->lanes[l].stream) hipStreamDestroy(bb->lanes[l].stream);
        if (bb->lanes[l].done) hipEventDestroy(bb->lanes[l].done);
    }
    if (bb->fork) hipEventDestroy(bb->fork);
    if (bb->last_done) hipEventDestroy(bb->last_done);
    for (auto& g : bb->graphs) hipGraphExecDestroy(g.exec);
    if (bb->gstream) hipStreamDestroy(bb->gstream);
    delete bb;
}

extern "C" int mmc_backbone_create(const void* packed, size_t nbytes, int arch, int device, int max_batch,
                                   mmc_backbone** out)
{
    return mmc_backbone_create_ex(packed, nbytes, arch, device, max_batch, 0u, out);
}

extern "C" int mmc_backbone_create_ex(const void* packed, size_t nbytes, int arch, int device, int max_batch, unsigned flags,
                                      mmc_backbone** out)
{
    if (!out) return fail(MMC_ERR_ARG, "out is NULL");
    if (flags & ~(unsigned)MMC_PRECISION_FP8) return fail(MMC_ERR_ARG, "unknown flags 0x%x", flags);
    if ((flags & MMC_PRECISION_FP8) && arch != MMC_ARCH_B4)
        return fail(MMC_ERR_ARG, "MMC_PRECISION_FP8 is implemented for MMC_ARCH_B4 (BASELINE configs[4]): B0's late blocks run fused kernels without a separate project GEMM");
    *out = nullptr;
    if (!packed || nbytes < 16) return fail(MMC_ERR_WEIGHTS, "weights blob is empty");
    if (arch != MMC_ARCH_B0 && arch != MMC_ARCH_B4) return fail(MMC_ERR_ARG, "unsupported arch %d (MMC_ARCH_B0 or MMC_ARCH_B4)", arch);
    if (max_batch < 1 || max_batch > 4096) return fail(MMC_ERR_ARG, "max_batch %d out of range [1,4096]", max_batch);
    const uint8_t* base = static_cast<const uint8_t*>(packed);
    uint32_t hdr[4];
    memcpy(hdr, base, 16);
    if (memcmp(base, "MMCW", 4) != 0) return fail(MMC_ERR_WEIGHTS, "bad magic in weights blob");
    if (hdr[1] != 1) return fail(MMC_ERR_WEIGHTS, "weights blob version %u, expected 1", hdr[1]);
    if ((int)hdr[2] != arch) return fail(MMC_ERR_WEIGHTS, "weights blob arch %u != requested %d", hdr[2], arch);
    const uint32_t nt = hdr[3];
    if (16 + (size_t)nt * 16 > nbytes) return fail(MMC_ERR_WEIGHTS, "weights blob truncated (table)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(MMC_ERR_HIP, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(MMC_ERR_ARG, "device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));

    mmc_backbone* bb = new mmc_backbone();
    bb->device = device;
    bb->max_batch = max_batch;
    bb->fp8 = (flags & MMC_PRECISION_FP8) != 0;
    const ArchDef AD = make_arch(arch);
    const bool is_b0 = arch == MMC_ARCH_B0;
    const int STEM_CH = AD.stem, HEAD_IN = AD.head_in, FEAT = AD.feat, NBLK = (int)AD.blocks.size();
    bb->arch = arch; bb->nblk = NBLK; bb->stem_ch = STEM_CH; bb->head_in = HEAD_IN; bb->feat = FEAT;
    bb->blk.resize(NBLK);
    const char* keep = getenv("MMC_KEEP_ACTIVATIONS");
    bb->keep = keep && keep[0] == '1';
    { const char* e = getenv("MMC_GRAPH"); bb->use_graph = !(e && e[0] == '0') && !bb->keep; }
    { const char* e = getenv("MMC_SE_SMALL"); bb->se_small = !(e && e[0] == '0'); }
    if (bb->keep) { int r__ = dev_alloc(bb, &bb->dbg_clk, (size_t)max_batch * 8); if (r__) { mmc_backbone_destroy(bb); return r__; } }
    { const char* e = getenv("MMC_TAIL_CLK");
      if (e && e[0] == '1' && !bb->keep) {
          // rows are indexed lane * lane_cap + row with lane_cap = ceil(max_batch / lanes): up to lanes - 1 <= 3 rows more than max_batch
          const size_t clk_rows = (size_t)max_batch + 4;
          int r__ = dev_alloc(bb, &bb->tail_clk, clk_rows * 64); if (r__) { mmc_backbone_destroy(bb); return r__; }
          hipMemset(bb->tail_clk, 0, clk_rows * 64 * sizeof(float));
          r__ = dev_alloc(bb, &bb->mid_clk, clk_rows * 128); if (r__) { mmc_backbone_destroy(bb); return r__; }
          hipMemset(bb->mid_clk, 0, clk_rows * 128 * sizeof(float));
          bb->use_graph = false;
      } }
    std::vector<uint64_t> table(2 * (size_t)nt);
    memcpy(table.data(), base + 16, (size_t)nt * 16);
    BlobReader rd{base, nbytes, nt, table.data()};
    int err = 0;
#define TAKE(var, n, what)                         \
    const float* var = rd.take((n), what, &err);   \
    if (!var) { mmc_backbone_destroy(bb); return err; }
#define TRY_OR_FREE(expr)                          \
    do { int r__ = (expr); if (r__) { mmc_backbone_destroy(bb); return r__; } } while (0)

    // ---- stem: [Cstem][27] folded (ky,kx,c), bias[Cstem], padval[3] ----
    {
        TAKE(w, (size_t)STEM_CH * 27, "stem.weight");
        TAKE(b, STEM_CH, "stem.bias");
        TAKE(pv, 3, "stem.padval");
        const int snt = STEM_CH / 16;   // output fragments: lane quarter q owns channels q*4*snt + 4t + j
        std::vector<_Float16> wp((size_t)STEM_CH * 32, (_Float16)0.0f);
        for (int t = 0; t < snt; ++t)
            for (int q = 0; q < 4; ++q)
                for (int j = 0; j < 4; ++j) {
                    const int prow = t * 16 + 4 * q + j;
                    const int c = q * 4 * snt + 4 * t + j;
                    const float* wc = w + (size_t)c * 27;
                    for (int qq = 0; qq < 3; ++qq)           // kernel row qq, bytes 0..7 of its 9-byte run
                        for (int jj = 0; jj < 8; ++jj) wp[prow * 32 + qq * 8 + jj] = (_Float16)(float)(wc[qq * 9 + jj] * LOG2E);
                    for (int jj = 0; jj < 3; ++jj) wp[prow * 32 + 24 + jj] = (_Float16)(float)(wc[jj * 9 + 8] * LOG2E);
                }
        TRY_OR_FREE(dev_upload(bb, &bb->stem_w, wp));
        std::vector<float> sb(STEM_CH);
        for (int c = 0; c < STEM_CH; ++c) sb[c] = (float)(b[c] * LOG2E);
        TRY_OR_FREE(dev_upload(bb, &bb->stem_b, sb));
        std::vector<float> pvv = {pv[0], pv[1], pv[2], 0.f};
        TRY_OR_FREE(dev_upload(bb, &bb->stem_pad, pvv));
    }
    // ---- blocks ----
    const char* fuse_env = getenv("MMC_FUSE");
    const bool fuse_generic_early = !is_b0 && !(fuse_env && fuse_env[0] == '0');   // (= fuse_generic, needed before its definition)
    const bool fuse_enabled = is_b0 && !(fuse_env && fuse_env[0] == '0');   // the fused kernels are shaped for B0's layers
    const char* dot2_env = getenv("MMC_MB_DOT2");
    const bool dot2_enabled = !(dot2_env && dot2_env[0] == '0');
    const bool fuse_generic = !is_b0 && !(fuse_env && fuse_env[0] == '0');   // B4: fused expand+depthwise where an instantiation fits
    const int b4_cc14 = [] { const char* e = getenv("MMC_B4_CC14"); return e && atoi(e) == 48 ? 48 : 96; }();   // 5x5 layers at 14x14: 96 measured +1.3 %
    { const char* e = getenv("MMC_THIN_PROJ"); bb->thin_proj = !(e && e[0] == '0'); }
    { const char* e = getenv("MMC_B1_PLANAR"); bb->b1_planar = !(e && e[0] == '0'); }
    bb->fuse_stem = fuse_enabled;
    const char* pp_env = getenv("MMC_PROJSE");
    const bool projse_enabled = fuse_enabled && !(pp_env && pp_env[0] == '0');
    const char* mid_env = getenv("MMC_MID14");
    // MMC_MID14: 0 = off (tile/chunk kernels), 1 (default) = blocks 6..10, 2 = blocks 6..8 only
    const int mid14_mode = !(fuse_enabled && projse_enabled) ? 0 : (mid_env ? atoi(mid_env) : 1);
    const bool mid14_enabled = mid14_mode != 0;
    bb->mid14 = mid14_enabled;
    bb->mid14_last = mid14_mode == 1 ? 10 : 8;
    { const char* e = getenv("MMC_MID14_B11"); bb->mid14_b11 = mid14_mode == 1 && e && e[0] == '1'; }   // block 11's front half (stride 2) too: measured equal (35.7 vs 33.0 us), opt-in
    // MMC_MID14M (default 2): mid14m_kernel -- depthwise conv on 4x4x4 MFMA blocks (block = channel), wave-private channel groups, one
    // barrier per kernel -- 2: on the 5x5 blocks 8..10 (13-24 % ahead of mid14_kernel alone on the chip, one workgroup per patch: bench
    // 228.4 k vs 227.1 k patches/s), 1: on blocks 6..10 (the 3x3 blocks are equal alone and lose in the bench), 0: mid14_kernel everywhere
    // (MMC_MID14M=2: only the 5x5 blocks 8..10, where it is 13-24 % ahead alone on the chip; the 3x3 blocks stay on mid14_kernel)
    const int mid14m_mode = [] { const char* e = getenv("MMC_MID14M"); return e ? atoi(e) : 2; }();
    const bool mid14m_enabled = mid14_enabled && mid14m_mode >= 1;
    // MMC_TAIL_DW4=1 (default 0): tail7_kernel's blocks 12..15 with the depthwise conv on 4x4x4 MFMA blocks, fused with the expand into one
    // wave-private phase.  Parity-tested; measured 44-45 k cycles per block for expand + depthwise against 40.6 k for the round-2 phases
    // (45 % fewer vector instructions, but the 4x4x4 MFMAs hold the issue port half their time and the phase does not
    // overlap its matrix and vector halves at two waves per SIMD), so it stays opt-in.
    const bool tail_dw4 = [] { const char* e = getenv("MMC_TAIL_DW4"); return e && e[0] == '1'; }();
    const char* mbt_env = getenv("MMC_MBT");
    const bool mbt_enabled = (fuse_enabled || fuse_generic_early) && !(mbt_env && mbt_env[0] == '0');
    // MMC_MBT4 (default 1): the 5x5 stride-1 layers at 28x28 (b4; B4's b7..b9) on mbt4_kernel (depthwise conv on 4x4x4 MFMA blocks)
    const bool mbt4_enabled = [] { const char* e = getenv("MMC_MBT4"); return !(e && e[0] == '0'); }();   // default since the pair-interleaved tile: b2 66 vs 78.5 us, b4 42.6 vs 59.5
    bb->mbt = mbt_enabled;
    { const char* e = getenv("MMC_MBT2"); bb->mbt2 = mbt_enabled && !(e && e[0] == '0'); }
    const char* tail_env = getenv("MMC_TAIL");
    const bool tail_enabled = fuse_enabled && !(tail_env && tail_env[0] == '0');
    int H = IMG / 2;
    // MMC_PRECISION_FP8: the project convs of the blocks whose OUTPUT is at most fp8_maxh pixels wide run on e4m3 operands.  Default 7
    // (the 7x7 stage: the CPU study of the operand format puts feature cosine >= 0.997 there); MMC_FP8_MAXH=14 adds the 14x14 stages.
    const int fp8_maxh = [] { const char* e = getenv("MMC_FP8_MAXH"); const int v = e ? atoi(e) : 7; return v < 7 ? 7 : v; }();
    bb->fp8_from = NBLK;
    size_t max_act = (size_t)H * H * STEM_CH, max_exp = 0, max_dw = 0, max_pool = 0;
    int max_c = 0;
    for (int i = 0; i < NBLK; ++i) {
        BlockW& B = bb->blk[i];
        B.d = AD.blocks[i];
        B.H = H;
        B.ce = B.d.cin * B.d.e;
        B.cs = B.d.cin / 4 > 1 ? B.d.cin / 4 : 1;
        B.has_expand = B.d.e != 1;
        B.skip = B.d.s == 1 && B.d.cin == B.d.cout;
        same_pad(H, B.d.k, B.d.s, &B.pad, &B.Ho);
        char nm[64];
        const float* exp_w_host = nullptr;
        if (B.has_expand) {
            snprintf(nm, sizeof nm, "b%d.expand", i);
            TAKE(w, (size_t)B.ce * B.d.cin, nm);
            TAKE(b, B.ce, nm);
            TRY_OR_FREE(pack_pw(bb, &B.expand, w, b, B.ce, B.d.cin, 0, LOG2E, LOG2E));
            exp_w_host = w;
        }
        {
            snprintf(nm, sizeof nm, "b%d.dw", i);
            TAKE(w, (size_t)B.ce * B.d.k * B.d.k, nm);  // [ce][k][k]
            TAKE(b, B.ce, nm);
            const int kk = B.d.k * B.d.k;
            // the taps see a log2(e)-scaled input (stem or expand output) and produce a scaled output: the factors cancel --
            // except for an expand-less block fed by a project conv (B4's block 1), whose input is in the plain domain
            const double tsc = (B.has_expand || i == 0) ? 1.0 : LOG2E;
            std::vector<float> wt((size_t)kk * B.ce);
            for (int c = 0; c < B.ce; ++c)
                for (int t = 0; t < kk; ++t) wt[(size_t)t * B.ce + c] = (float)(w[(size_t)c * kk + t] * tsc);
            TRY_OR_FREE(dev_upload(bb, &B.dw_w, wt));
            std::vector<float> db(B.ce);
            for (int c = 0; c < B.ce; ++c) db[c] = (float)(b[c] * LOG2E);
            TRY_OR_FREE(dev_upload(bb, &B.dw_b, db));
            if ((tail_enabled && i >= 11 && i <= 15) || (mid14_enabled && i >= 6 && i <= 11) || (mbt_enabled && mbt_has(H, B.d.k, B.d.s, B.d.cin, B.ce))) {
                // taps of tail7_kernel / mid14_kernel / mbt_kernel as fp16 pairs: kernel row ky = (k0,k1), (k2,k3), (k4,0); the kernel derives the
                // odd-output pairs by shifts, giving the same values as mbconv_d_kernel's wl2 table
                std::vector<uint32_t> dp((size_t)15 * B.ce, 0u);
                auto tap = [&](int c, int ky, int kx) -> _Float16 {
                    return kx < B.d.k ? (_Float16)w[(size_t)c * kk + ky * B.d.k + kx] : (_Float16)0.0f;
                };
                for (int c = 0; c < B.ce; ++c)
                    for (int ky = 0; ky < B.d.k; ++ky)
                        for (int d = 0; d < 3; ++d) {   // slot 3*ky + d holds taps (2d, 2d+1) of kernel row ky
                            _Float16 h[2] = {tap(c, ky, 2 * d), tap(c, ky, 2 * d + 1)};
                            uint32_t u;
                            memcpy(&u, h, 4);
                            dp[(size_t)(ky * 3 + d) * B.ce + c] = u;
                        }
                TRY_OR_FREE(dev_upload(bb, &B.t_dwp, dp));
                // the same 15 dwords + the bias (as the 16th) in 16-byte requests: [4][ce][4] -- request j of channel c holds slots
                // 4j .. 4j+3; a wave's request is 1 KB contiguous (16 dword loads per thread were 16 wave-instructions of 256 bytes)
                std::vector<uint32_t> dp4((size_t)16 * B.ce, 0u);
                for (int c = 0; c < B.ce; ++c)
                    for (int t = 0; t < 16; ++t) {
                        uint32_t v = 0u;
                        if (t < 15) v = dp[(size_t)t * B.ce + c];
                        else memcpy(&v, &db[c], 4);   // the bias exactly as dw_b holds it
                        dp4[((size_t)(t / 4) * B.ce + c) * 4 + (t % 4)] = v;
                    }
                TRY_OR_FREE(dev_upload(bb, &B.t_dwp4, dp4));
            }
            if (((mid14m_enabled && i >= (mid14m_mode == 2 ? 8 : 6) && i <= 10 && H == 14) || (mbt4_enabled && mbt_enabled && H == 28 && B.d.k == 5 && mbt_has(H, B.d.k, B.d.s, B.d.cin, B.ce)) || (tail_dw4 && tail_enabled && i >= 12 && i <= 15 && H == 7)) && B.d.s == 1 && B.ce % 16 == 0) {
                // Depthwise on the matrix pipe (mid14m_kernel, v_mfma_f32_4x4x4_16B_f16: 16 independent blocks = 16 channels).  A operand
                // of block c, kernel row ky, input quad h (columns x0 - 2 + 4h .. +3 of an output tile x0 .. x0+3): the Toeplitz slice
                // A[i][k] = w[c][ky][k - i + 4h - 2 + R] (zero outside 0 .. K-1), lane 4 blk + i holding k = 0 .. 3, block blk = channel
                // 2 (blk & 3) + ((blk >> 2) & 1) + 8 (blk >> 3) of the group (channels 2k, 2k+1 in neighbouring 16-lane rows: the kernel
                // pairs them with v_permlane16_swap).
                const int K = B.d.k, R = K / 2, ngr = B.ce / 16;
                std::vector<_Float16> dd((size_t)ngr * K * 2 * 64 * 4, (_Float16)0.0f);
                for (int g = 0; g < ngr; ++g)
                    for (int ky = 0; ky < K; ++ky)
                        for (int h = 0; h < 2; ++h)
                            for (int ln = 0; ln < 64; ++ln) {
                                const int blk = ln >> 2, ii = ln & 3;
                                const int cc = 2 * (blk & 3) + ((blk >> 2) & 1) + 8 * (blk >> 3);
                                for (int k = 0; k < 4; ++k) {
                                    const int t = k - ii + 4 * h - 2 + R;
                                    if (t < 0 || t >= K) continue;
                                    dd[((((size_t)g * K + ky) * 2 + h) * 64 + ln) * 4 + k] = (_Float16)(float)(w[(size_t)(16 * g + cc) * kk + ky * K + t] * tsc);
                                }
                            }
                TRY_OR_FREE(dev_upload(bb, &B.dw_diag, dd));
            }
        }
        {
            snprintf(nm, sizeof nm, "b%d.se", i);
            TAKE(wr, (size_t)B.cs * B.ce, nm);
            TAKE(br, B.cs, nm);
            TAKE(we, (size_t)B.ce * B.cs, nm);
            TAKE(be, B.ce, nm);
            // Squeeze-excite weights, fp32 in MFMA fragment order (se_fused_kernel): 3 output/k fragments of 16
            // cover Cs <= 48.  FC1 carries 1/(HW*log2e): the pooled sums are over HW pixels of log2(e)-scaled
            // activations.
            B.cs4 = (B.cs + 3) / 4 * 4;
            const double psc = 1.0 / ((double)B.Ho * B.Ho * LOG2E);
            const int ng = is_b0 ? B.ce / 16 : 0;   // (se_fused_kernel's packing: B0 only, Cs <= 48)
            std::vector<float> wrp((size_t)ng * 3 * 64 * 4 + 4, 0.f), wep((size_t)ng * 3 * 64 * 4 + 4, 0.f);
            for (int g = 0; g < ng; ++g)
                for (int t = 0; t < 3; ++t)
                    for (int ln = 0; ln < 64; ++ln)
                        for (int e = 0; e < 4; ++e) {
                            const int ii = ln & 15, qq = ln >> 4;
                            const size_t off = (((size_t)g * 3 + t) * 64 + ln) * 4 + e;
                            const int j = 16 * t + ii, c = 16 * g + 4 * qq + e;         // FC1: Wr[j][c]
                            if (j < B.cs) wrp[off] = (float)(wr[(size_t)j * B.ce + c] * psc);
                            const int n = 16 * g + ii, k = 16 * t + 4 * qq + e;         // FC2: We[n][k] (T = g, k-group = t)
                            if (k < B.cs) wep[off] = we[(size_t)n * B.cs + k];
                        }
            // B4: squeeze-excite + project per patch wherever proj_patch_kernel has the shape (blocks 6-15) and Cs fits its 32 slots
            const bool fp8_blk = bb->fp8 && B.Ho <= fp8_maxh;
            const bool b4_pp = !fp8_blk && fuse_generic && !(pp_env && pp_env[0] == '0') && B.has_expand && B.cs <= 28 &&
                               proj_patch_has(B.ce, B.d.cout, B.Ho * B.Ho, B.skip ? 1 : 0);
            const bool pp_blk = (projse_enabled && i >= 3 && i <= 10) || (tail_enabled && i == 11) || b4_pp;
            if ((tail_enabled && i >= 12 && i <= 15 && B.cs == 48) || pp_blk) {
                // fp16, transposed for matrix-vector use: Wr^T [ce][csp], We^T [csp][ce] (csp = Cs padded to 4)
                const int csp = (B.cs + 3) / 4 * 4;
                std::vector<_Float16> wrt((size_t)B.ce * csp, (_Float16)0.0f), wet((size_t)csp * B.ce, (_Float16)0.0f);
                for (int c = 0; c < B.ce; ++c)
                    for (int j = 0; j < B.cs; ++j) {
                        wrt[(size_t)c * csp + j] = (_Float16)wr[(size_t)j * B.ce + c];
                        wet[(size_t)j * B.ce + c] = (_Float16)we[(size_t)c * B.cs + j];
                    }
                TRY_OR_FREE(dev_upload(bb, &B.t_wr, wrt));
                TRY_OR_FREE(dev_upload(bb, &B.t_we, wet));
                if (pp_blk && csp <= 28) {
                    // proj_patch_kernel's FC1: [group of four outputs][channel row][4] (ProjPatchArgs::wr_g), zero beyond ce
                    const int rows = proj_patch_fc1_rows(B.ce);
                    std::vector<_Float16> wrg((size_t)(csp / 4) * rows * 4, (_Float16)0.0f);
                    for (int g = 0; g < csp / 4; ++g)
                        for (int c = 0; c < B.ce; ++c)
                            for (int e = 0; e < 4; ++e) wrg[((size_t)g * rows + c) * 4 + e] = wrt[(size_t)c * csp + 4 * g + e];
                    TRY_OR_FREE(dev_upload(bb, &B.t_wrg, wrg));
                }
                if (tail_enabled && i >= 12 && i <= 15 && B.cs == 48 && B.ce == 1152) {
                    // tail7_kernel's 16-byte request layout (TailBlock::wr_t / we_t): two rows of the transposed matrices per request
                    std::vector<_Float16> wr2((size_t)18 * 384 * 8), we2((size_t)24 * 288 * 8);
                    for (int p = 0; p < 18; ++p)
                        for (int t = 0; t < 384; ++t)
                            for (int e = 0; e < 8; ++e) {
                                const int cr = t / 12, j4 = t % 12, c = 64 * p + (e < 4 ? 0 : 32) + cr;
                                wr2[((size_t)p * 384 + t) * 8 + e] = wrt[(size_t)c * csp + 4 * j4 + (e & 3)];
                            }
                    for (int p = 0; p < 24; ++p)
                        for (int t = 0; t < 288; ++t)
                            for (int e = 0; e < 8; ++e) we2[((size_t)p * 288 + t) * 8 + e] = wet[(size_t)(2 * p + (e < 4 ? 0 : 1)) * B.ce + 4 * t + (e & 3)];
                    TRY_OR_FREE(dev_upload(bb, &B.t_wr2, wr2));
                    TRY_OR_FREE(dev_upload(bb, &B.t_we2, we2));
                }
                std::vector<float> brp(csp > 32 ? csp : 32, 0.f);
                for (int j = 0; j < B.cs; ++j) brp[j] = br[j];
                TRY_OR_FREE(dev_upload(bb, &B.pp_br, brp));
                B.pp_csp = csp;
            }
            if ((B.ce <= 256 && B.cs <= 16) || !is_b0) {   // early blocks: light per-patch squeeze-excite kernel; generic schedule: se_wide
                std::vector<float> wrn((size_t)B.cs * B.ce);
                for (size_t e = 0; e < wrn.size(); ++e) wrn[e] = (float)(wr[e] * psc);
                TRY_OR_FREE(dev_upload(bb, &B.se_wr_nat, wrn));
                if (is_b0) {
                    TRY_OR_FREE(dev_upload(bb, &B.se_we_nat, std::vector<float>(we, we + (size_t)B.ce * B.cs)));
                } else {   // se_wide_kernel reads the excite FC transposed ([Cs][C]): neighbouring lanes, neighbouring words
                    std::vector<float> wet((size_t)B.cs * B.ce);
                    for (int c = 0; c < B.ce; ++c)
                        for (int j = 0; j < B.cs; ++j) wet[(size_t)j * B.ce + c] = we[(size_t)c * B.cs + j];
                    TRY_OR_FREE(dev_upload(bb, &B.se_we_nat, wet));
                }
                TRY_OR_FREE(dev_upload(bb, &B.se_br_nat, std::vector<float>(br, br + B.cs)));
            }
            TRY_OR_FREE(dev_upload(bb, &B.se_wrp, wrp));
            TRY_OR_FREE(dev_upload(bb, &B.se_wep, wep));
            std::vector<float> brs(B.cs > 48 ? B.cs : 48, 0.f);
            for (int j = 0; j < B.cs; ++j) brs[j] = br[j];
            TRY_OR_FREE(dev_upload(bb, &B.se_br, brs));
            TRY_OR_FREE(dev_upload(bb, &B.se_be, std::vector<float>(be, be + B.ce)));
        }
        {
            snprintf(nm, sizeof nm, "b%d.project", i);
            TAKE(w, (size_t)B.d.cout * B.ce, nm);
            TAKE(b, B.d.cout, nm);
            TRY_OR_FREE(pack_pw(bb, &B.project, w, b, B.d.cout, B.ce, B.Ho <= 14 ? pick_nt(B.d.cout, true) : 0, 1.0 / LOG2E, 1.0));
            if (bb->fp8 && B.Ho <= fp8_maxh) {
                // e4m3 weights for pw_gemm_fp8_kernel: per-output-channel scale amax / 448 (the 1 / log2 e of the scaled domain rides in the
                // scale), fragment f = channels 16 f .. +15, k-step of 128, lane (i, q) holds k = 128 ks + 32 q .. +31 of channel 16 f + i
                Fp8Layer& L = B.p8;
                L.N = B.d.cout; L.K = B.ce; L.KS128 = (B.ce + 127) / 128; L.NFp = ((B.d.cout + 15) / 16 + 6) / 7 * 7;
                std::vector<uint8_t> w8((size_t)L.NFp * L.KS128 * 64 * 32, (uint8_t)0);
                std::vector<float> sw((size_t)16 * L.NFp, 0.0f), bp((size_t)16 * L.NFp, 0.0f);
                for (int nn = 0; nn < L.N; ++nn) {
                    float amax = 0.f;
                    for (int k = 0; k < L.K; ++k) amax = std::max(amax, std::fabs(w[(size_t)nn * L.K + k]));
                    const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
                    sw[nn] = (float)(sc / LOG2E);
                    bp[nn] = b[nn];
                    for (int k = 0; k < L.K; ++k) {
                        const int f = nn / 16, ii = nn % 16, ks = k / 128, qq = (k % 128) / 32, e = k % 32;
                        w8[((((size_t)f * L.KS128 + ks) * 64) + qq * 16 + ii) * 32 + e] = e4m3_encode(w[(size_t)nn * L.K + k] / sc);
                    }
                }
                TRY_OR_FREE(dev_upload(bb, &L.w8, w8));
                TRY_OR_FREE(dev_upload(bb, &L.sw, sw));
                TRY_OR_FREE(dev_upload(bb, &L.b, bp));
                if (i < bb->fp8_from) bb->fp8_from = i;
            }
            if (i == 0 && fuse_enabled && B.ce == 32 && B.d.cout == 16) {
                // block 0's project conv as ONE MFMA fragment (16 outputs x 32 inputs) for mbconv_a_kernel PRE
                std::vector<_Float16> wf(512, (_Float16)0.0f);
                for (int c = 0; c < 16; ++c)
                    for (int k = 0; k < 32; ++k) wf[((k / 8) * 16 + c) * 8 + (k % 8)] = (_Float16)(float)(w[(size_t)c * 32 + k] * (1.0 / LOG2E));
                TRY_OR_FREE(dev_upload(bb, &bb->b0_pre_w, wf));
            }
            if ((projse_enabled && i >= 3 && i <= 10) ||
                (fuse_generic && !(pp_env && pp_env[0] == '0') && B.has_expand && B.cs <= 28 &&
                 proj_patch_has(B.ce, B.d.cout, B.Ho * B.Ho, B.skip ? 1 : 0))) {
                // proj_patch_kernel: fragment order as below, K and N zero-padded to whole fragments
                const int ks32 = proj_patch_ksteps(B.ce), nf = (B.d.cout + 15) / 16;
                std::vector<_Float16> wf((size_t)nf * ks32 * 512, (_Float16)0.0f);
                for (int c = 0; c < B.d.cout; ++c)
                    for (int k = 0; k < B.ce; ++k)
                        wf[((((size_t)(c / 16) * ks32 + k / 32) * 64) + ((k % 32) / 8) * 16 + (c % 16)) * 8 + (k % 8)] =
                            (_Float16)(float)(w[(size_t)c * B.ce + k] * (1.0 / LOG2E));
                TRY_OR_FREE(dev_upload(bb, &B.pp_w, wf));
                std::vector<float> bp((size_t)16 * nf, 0.f);
                for (int c = 0; c < B.d.cout; ++c) bp[c] = b[c];
                TRY_OR_FREE(dev_upload(bb, &B.pp_b, bp));
                B.pp = true;
            }
            if (tail_enabled && i == 11 && B.ce == 672 && B.d.cout == 192) {
                // tail7 pre-block: fragment order [12][24][64][8], k-steps 21..23 zero (its k-loop runs 4 steps at a time)
                std::vector<_Float16> wf((size_t)12 * 24 * 512, (_Float16)0.0f);
                for (int c = 0; c < B.d.cout; ++c)
                    for (int k = 0; k < B.ce; ++k)
                        wf[((((size_t)(c / 16) * 24 + k / 32) * 64) + ((k % 32) / 8) * 16 + (c % 16)) * 8 + (k % 8)] =
                            (_Float16)(float)(w[(size_t)c * B.ce + k] * (1.0 / LOG2E));
                TRY_OR_FREE(dev_upload(bb, &bb->pre_wproj, wf));
            }
            if (tail_enabled && i >= 12 && i <= 15) {
                // plain MFMA fragment order [cout/16][ce/32][64 lanes][8]: lane (q*16 + m) holds W[16nf + m][32ks + 8q ..+8]
                const int ks32 = B.ce / 32;
                std::vector<_Float16> wf((size_t)B.d.cout * B.ce);
                for (int c = 0; c < B.d.cout; ++c)
                    for (int k = 0; k < B.ce; ++k)
                        wf[((((size_t)(c / 16) * ks32 + k / 32) * 64) + ((k % 32) / 8) * 16 + (c % 16)) * 8 + (k % 8)] =
                            (_Float16)(float)(w[(size_t)c * B.ce + k] * (1.0 / LOG2E));
                TRY_OR_FREE(dev_upload(bb, &B.t_wproj, wf));
            }
        }
        // depthwise geometry
        B.tw = (B.Ho % 4 == 0) ? 4 : (B.Ho % 7 == 0 && B.Ho <= 7 ? 7 : 2);
        B.CG = B.ce / 8;
        B.nz = 1;
        while (B.CG / B.nz > 256 || B.CG % B.nz) ++B.nz;   // layers wider than 2048 channels: split the channel groups
        B.CG /= B.nz;
        B.S = 256 / B.CG > 0 ? 256 / B.CG : 1;
        const int strips = B.Ho * (B.Ho / B.tw);
        const int passes = (strips + B.S - 1) / B.S;
        B.iters = passes >= 8 ? 4 : 1;
        B.parts = (passes + B.iters - 1) / B.iters;
        if (B.has_expand && ((fuse_enabled && i < 16) || fuse_generic)) {
            FuseCfg fc = is_b0 ? B0_FUSE[i] : generic_fuse_cfg(H, B.d.k, B.d.s, B.ce, b4_cc14);
            if (const char* ov = getenv("MMC_FUSE_CFG")) {   // "i:TH,TWo,CC;..." experiment override
                char key[16];
                snprintf(key, sizeof key, "%d:", i);
                const char* hit = strstr(ov, key);
                while (hit && hit != ov && hit[-1] != ';') hit = strstr(hit + 1, key);
                if (hit) sscanf(hit + strlen(key), "%d,%d,%d,%d,%d,%d", &fc.TH, &fc.TWo, &fc.CC, &fc.TW, &fc.PB, &fc.WLDS);
            }
            if (fc.TH > 0 && B.Ho % fc.TH == 0 && B.Ho % fc.TWo == 0 && B.ce % fc.CC == 0 && fc.CC % 16 == 0) {
                B.f_TH = fc.TH; B.f_TWo = fc.TWo; B.f_CC = fc.CC;
                B.f_tw = fc.TW;
                B.f_ksteps = (B.d.cin + 31) / 32;
                B.f_CCG = fc.CC / 8;
                B.f_S = 256 / B.f_CCG;
                B.f_tiles_x = B.Ho / fc.TWo; B.f_tiles_y = B.Ho / fc.TH;
                int wh = (fc.TH - 1) * B.d.s + B.d.k, wwid = (fc.TWo - 1) * B.d.s + B.d.k;
                if (wh > H) wh = H;
                if (wwid > H) wwid = H;
                B.f_pb = fc.PB < 1 ? 1 : fc.PB;
                const int ppad = (B.f_pb * wh * wwid + 15) / 16 * 16;
                B.f_npair = (ppad / 16 + 7) / 8;
                B.f_wlds = fc.WLDS ? 1 : 0;
                B.f_wfr_off = ppad * (fc.CC * 2 + 16);                       // E | [weight fragments] | taps+bias
                B.f_wl_off = B.f_wfr_off + (B.f_wlds ? (fc.CC / 16) * B.f_ksteps * 1024 : 0);
                B.f_lds = B.f_wl_off + (B.d.k * B.d.k + 1) * fc.CC * 4;
                B.f_red_off = 0;  // pool scratch [PB][S][CC] aliases E when it fits, else gets its own space
                if (B.f_pb * B.f_S * fc.CC * 4 > B.f_wfr_off) { B.f_red_off = B.f_lds; B.f_lds += B.f_pb * B.f_S * fc.CC * 4; }
                const int kp = 32 * B.f_ksteps;
                if (B.f_lds <= 128 * 1024 && fc.TWo % B.f_tw == 0) {
                    std::vector<_Float16> wn((size_t)B.ce * kp, (_Float16)0.0f);
                    for (int c = 0; c < B.ce; ++c)
                        for (int k = 0; k < B.d.cin; ++k) wn[(size_t)c * kp + k] = (_Float16)(float)(exp_w_host[(size_t)c * B.d.cin + k] * LOG2E);
                    TRY_OR_FREE(dev_upload(bb, &B.exp_nat, wn));
                    if (i == 1 && B.d.cin == 16 && kp == 32) {
                        // K-permuted copy for mbconv_a_kernel PRE: MFMA slot 8q+j holds input channel 4q+j (j < 4), the
                        // rest zero -- the layout block 0's in-kernel project leaves in the lanes
                        std::vector<_Float16> wq((size_t)B.ce * 32, (_Float16)0.0f);
                        for (int c = 0; c < B.ce; ++c)
                            for (int qq = 0; qq < 4; ++qq)
                                for (int j = 0; j < 4; ++j) wq[(size_t)c * 32 + 8 * qq + j] = wn[(size_t)c * kp + 4 * qq + j];
                        TRY_OR_FREE(dev_upload(bb, &bb->b1_exp_pre, wq));
                    }
                    {   // fragment order: ((c/16 * ksteps + k/32) * 64 + (k%32)/8 * 16 + c%16) * 8 + k%8
                        std::vector<_Float16> wf((size_t)B.ce * kp, (_Float16)0.0f);
                        for (int c = 0; c < B.ce; ++c)
                            for (int k = 0; k < B.d.cin; ++k)
                                wf[((((size_t)(c / 16) * B.f_ksteps + k / 32) * 64) + ((k % 32) / 8) * 16 + (c % 16)) * 8 + (k % 8)] = wn[(size_t)c * kp + k];
                        TRY_OR_FREE(dev_upload(bb, &B.exp_frag, wf));
                    }
                    B.fused = true;
                    // dot2 depthwise variant: only where it measured faster (5x5 stride-1 blocks; MI355X, batch 128/256)
                    static const bool B0_DOT2[16] = {0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 1, 0, 1, 1, 1, 0};
                    if (is_b0 && dot2_enabled && (B0_DOT2[i] || (dot2_env && dot2_env[0] == '2'))) {
                        // window of mbconv_d_kernel: rows as above, columns widened to whole pixel pairs (even absolute x)
                        const int padb = B.pad;
                        int rowlen = 0;
                        for (int tx = 0; tx < B.f_tiles_x; ++tx) {
                            int x0 = tx * fc.TWo * B.d.s - padb, x1 = (tx * fc.TWo + fc.TWo - 1) * B.d.s - padb + B.d.k;
                            x0 = x0 < 0 ? 0 : x0;
                            x1 = x1 > H ? H : x1;
                            const int rl = 2 * (((x1 + 1) >> 1) - (x0 >> 1));
                            if (rl > rowlen) rowlen = rl;
                        }
                        const int dpad = (B.f_pb * wh * rowlen + 15) / 16 * 16;
                        const int np = (B.pad % 2 + B.d.s + B.d.k + 1) / 2;
                        B.d_npair = (dpad / 16 + 7) / 8;
                        B.d_wl_off = dpad * fc.CC * 2;
                        B.d_lds = B.d_wl_off + B.d.k * 2 * np * fc.CC * 4 + fc.CC * 4;
                        B.d_red_off = 0;
                        if (B.f_pb * B.f_S * fc.CC * 4 > B.d_wl_off) { B.d_red_off = B.d_lds; B.d_lds += B.f_pb * B.f_S * fc.CC * 4; }
                        B.use_d = B.d_lds <= 128 * 1024;
                    }
                    const size_t pp = (size_t)B.f_tiles_x * B.f_tiles_y * B.ce;
                    if (pp > max_pool) max_pool = pp;
                }
            }
        }
        if ((size_t)H * H * B.ce > max_exp && B.has_expand && !B.fused) max_exp = (size_t)H * H * B.ce;
        if ((size_t)B.Ho * B.Ho * B.ce > max_dw) max_dw = (size_t)B.Ho * B.Ho * B.ce;
        if ((size_t)B.Ho * B.Ho * B.d.cout > max_act) max_act = (size_t)B.Ho * B.Ho * B.d.cout;
        if ((size_t)B.parts * B.ce > max_pool) max_pool = (size_t)B.parts * B.ce;
        if (i == 0 && (size_t)49 * B.ce > max_pool) max_pool = (size_t)49 * B.ce;
        if (B.ce > max_c) max_c = B.ce;
        H = B.Ho;
    }
    {
        TAKE(w, (size_t)FEAT * HEAD_IN, "head.weight");
        TAKE(b, FEAT, "head.bias");
        TRY_OR_FREE(pack_pw(bb, &bb->head, w, b, FEAT, HEAD_IN, 4, LOG2E, LOG2E));
        if (tail_enabled && HEAD_IN % 32 == 0) {   // the same weights in plain fragment order [80][10][64][8] for tail7's head phase
            std::vector<_Float16> wf((size_t)FEAT * HEAD_IN);
            for (int c = 0; c < FEAT; ++c)
                for (int k = 0; k < HEAD_IN; ++k)
                    wf[((((size_t)(c / 16) * (HEAD_IN / 32) + k / 32) * 64) + ((k % 32) / 8) * 16 + (c % 16)) * 8 + (k % 8)] =
                        (_Float16)(float)(w[(size_t)c * HEAD_IN + k] * LOG2E);
            TRY_OR_FREE(dev_upload(bb, &bb->head_wfrag, wf));
        }
    }
    if (tail_enabled) {
        bool ok = NBLK == 16;
        for (int i = 12; i <= 15 && ok; ++i) {
            const BlockW& B = bb->blk[i];
            ok = ok && B.exp_frag && B.t_wr2 && B.t_we2 && B.t_dwp && B.t_wproj && B.H == 7 && B.d.s == 1 && B.d.cin == 192 && B.ce == 1152;
        }
        if (ok) {
            std::vector<TailBlock> tab(4);
            for (int j = 0; j < 4; ++j) {
                const BlockW& B = bb->blk[12 + j];
                tab[j] = TailBlock{B.exp_frag, B.expand.b, B.t_dwp4, B.dw_b, B.t_wr2, B.se_br, B.t_we2, B.se_be, B.t_wproj, B.project.b,
                                   B.d.cout, B.d.k, B.dw_diag};
                bb->tail_dw4 = bb->tail_dw4 && B.dw_diag != nullptr;
            }
            TRY_OR_FREE(dev_upload(bb, &bb->tail_tab, tab));
            const char* tf = getenv("MMC_TAIL_FULL");
            const BlockW& B11 = bb->blk[11];
            bb->tail_full = !(tf && tf[0] == '0') && bb->pre_wproj && bb->head_wfrag && B11.t_wr && B11.pp_csp == 28 && B11.fused &&
                            B11.f_tiles_x * B11.f_tiles_y == 1;
            const char* tb = getenv("MMC_TAIL_B11");
            bb->tail_b11 = bb->tail_full && !(tb && tb[0] == '0') && B11.exp_frag && B11.t_dwp && B11.f_ksteps == 4 && B11.ce == 672 &&
                           B11.d.k == 5 && B11.d.s == 2 && B11.H == 14;
        }
    }
    {
        const char* e = getenv("MMC_FUSE_B0");
        const BlockW& B1 = bb->blk[1];
        { const char* m1 = getenv("MMC_MB1"); bb->mb1 = !(m1 && m1[0] == '0'); }
        bb->fuse_b0b1 = !(e && e[0] == '0') && bb->fuse_stem && bb->b0_pre_w && bb->b1_exp_pre && B1.fused && !B1.use_d && B1.f_TH == 8 &&
                        B1.f_TWo == 8 && B1.f_CC == 48 && B1.f_tw == 2 && B1.f_pb == 1 && !B1.f_wlds && B1.f_npair == 3;
    }
    if (rd.next != nt) {
        mmc_backbone_destroy(bb);
        return fail(MMC_ERR_WEIGHTS, "weights blob has %u tensors, expected %u", nt, rd.next);
    }
    const size_t mb = (size_t)max_batch;
    {
        const char* ls = getenv("MMC_LANES");
        int nl = ls ? atoi(ls) : 2;
        if (nl < 1) nl = 1;
        if (nl > 4) nl = 4;
        if (bb->keep || max_batch < 2 * nl) nl = 1;
        bb->nlanes = nl;
        bb->lane_cap = (max_batch + nl - 1) / nl;
        const size_t lc = (size_t)bb->lane_cap;
        for (int l = 0; l < nl; ++l) {
            mmc_backbone::Lane& L = bb->lanes[l];
            TRY_OR_FREE(dev_alloc(bb, &L.act0, lc * max_act));
            TRY_OR_FREE(dev_alloc(bb, &L.act1, lc * max_act));
            TRY_OR_FREE(dev_alloc(bb, &L.expbuf, lc * (max_exp ? max_exp : 64)));
            TRY_OR_FREE(dev_alloc(bb, &L.dwbuf, lc * max_dw));
            TRY_OR_FREE(dev_alloc(bb, &L.pool_part, lc * max_pool));
            TRY_OR_FREE(dev_alloc(bb, &L.gate, lc * (size_t)max_c));
            if (nl > 1) {
                if (hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) != hipSuccess ||
                    hipEventCreateWithFlags(&L.done, hipEventDisableTiming) != hipSuccess) {
                    mmc_backbone_destroy(bb);
                    return fail(MMC_ERR_HIP, "cannot create internal stream/event");
                }
            }
        }
        if (nl > 1 && hipEventCreateWithFlags(&bb->fork, hipEventDisableTiming) != hipSuccess) {
            mmc_backbone_destroy(bb);
            return fail(MMC_ERR_HIP, "cannot create fork event");
        }
    }
    TRY_OR_FREE(dev_alloc(bb, &bb->in_stage, mb * (size_t)IMG * IMG * 3));
    TRY_OR_FREE(dev_alloc(bb, &bb->out_stage, mb * (size_t)FEAT));
#undef TAKE
#undef TRY_OR_FREE
    *out = bb;
    return MMC_OK;
}

extern "C" int mmc_feature_dim(const mmc_backbone* bb) { return bb ? bb->feat : 0; }
extern "C" int mmc_backbone_max_batch(const mmc_backbone* bb) { return bb ? bb->max_batch : 0; }
extern "C" int mmc_backbone_lanes(const mmc_backbone* bb) { return bb ? bb->nlanes : 0; }
extern "C" size_t mmc_backbone_workspace_bytes(const mmc_backbone* bb) { return bb ? bb->ws_bytes : 0; }

// ------------------------------------------------------------------------------------------
// the launch schedule for one pass over n <= max_batch resident patches
// ------------------------------------------------------------------------------------------
static int save_act(mmc_backbone* bb, const char* name, const void* dev, size_t elems, bool is_half, hipStream_t st)
{
    Saved& s = bb->saved[name];
    const size_t bytes = elems * (is_half ? 2 : 4);
    if (s.bytes < bytes) {
        if (s.dev) hipFree(s.dev);
        HIP_TRY(hipMalloc(&s.dev, bytes));
        s.bytes = bytes;
    }
    s.elems = elems;
    s.is_half = is_half;
    HIP_TRY(hipMemcpyAsync(s.dev, dev, bytes, hipMemcpyDeviceToDevice, st));
    return 0;
}

static int gemm_mt(const PwLayer& L, int M)
{
    // two row fragments per wave when that still leaves >= ~4 workgroups per CU
    const long wgs2 = ((long)M + 127) / 128 * L.n_chunks;
    return wgs2 >= 1024 ? 2 : 1;
}

static std::string gemm_label(const PwLayer& L, int M, int epi, bool gate, bool res)
{
    char b[64];
    snprintf(b, sizeof b, "pw_gemm<%d,%d,%d,%d,%d>", epi == EPI_GAP ? 1 : gemm_mt(L, M), L.nt, epi, gate ? 1 : 0, res ? 1 : 0);
    return b;
}

static int run_gemm(const PwLayer& L, const _Float16* X, int M, _Float16* Y, int epi, const float* gate, int HW,
                    const _Float16* res, float* gap_out, hipStream_t st)
{
    GemmArgs a{};
    a.X = X; a.M = M; a.K = L.K; a.Wp = L.w; a.Kp = L.Kp; a.bias = L.b; a.Y = Y; a.N = L.N;
    a.nt = L.nt; a.n_chunks = L.n_chunks; a.epi = epi; a.gate = gate; a.HW = HW; a.res = res;
    a.gap_out = gap_out; a.inv_hw = (float)(1.0 / ((double)HW * LOG2E));  // GAP input is log2(e)-scaled
    a.mt = gemm_mt(L, M);
    a.defer_gate = (gate && HW <= 49) ? 1 : 0;   // 7x7 layers: few workgroups, deep K -> overlap loads with MFMAs
    return launch_pw_gemm(a, st);
}

static int forward_lane(mmc_backbone* bb, mmc_backbone::Lane& ws, const uint8_t* patches_dev, int n, float* out_dev,
                        hipStream_t st, Prof* prof)
{
#define STEP(nm, label, expr)                                                           \
    do {                                                                                \
        ProfEntry pe_;                                                                  \
        if (prof) {                                                                     \
            pe_.name = std::string(nm) + "|" + std::string(label);                     \
            HIP_TRY(hipEventCreate(&pe_.e0));                                           \
            HIP_TRY(hipEventCreate(&pe_.e1));                                           \
            HIP_TRY(hipEventRecord(pe_.e0, st));                                        \
        }                                                                               \
        KTRY(expr);                                                                     \
        if (prof) {                                                                     \
            HIP_TRY(hipEventRecord(pe_.e1, st));                                        \
            prof->entries.push_back(pe_);                                               \
        }                                                                               \
    } while (0)
    char nm[64];
    _Float16* x = ws.act0;
    _Float16* y = ws.act1;
    const bool stem_fused = bb->fuse_stem;   // then the stem tensor never exists in HBM (no "stem" activation to keep)
    if (!stem_fused) {
        STEP("stem", "stem_conv", launch_stem(patches_dev, bb->stem_w, bb->stem_b, bb->stem_pad, x, n, bb->stem_ch, st));
        if (bb->keep) { int r = save_act(bb, "stem", x, (size_t)n * 112 * 112 * bb->stem_ch, true, st); if (r) return r; }
    }
    bool tail_done = false;
    const bool se_small_enabled = bb->se_small;
    const int FEAT = bb->feat;
    (void)FEAT;
    for (int i = 0; i < bb->nblk; ++i) {
        if (i == 12 && bb->tail_tab) {
            // blocks 12..15 in one launch, one patch per workgroup, tensors resident in LDS (tail7_kernel)
            if (!bb->keep) {
                TailArgs ta{};
                ta.X = x; ta.Y = y; ta.B = n; ta.nblk = 4; ta.blk = bb->tail_tab; ta.dw4 = bb->tail_dw4;
                STEP("b12-15.tail", "tail7", launch_tail7(ta, st));
                _Float16* t = x; x = y; y = t;
            } else {
                for (int j = 0; j < 4; ++j) {   // block at a time so every intermediate tensor can be read back
                    TailArgs ta{};
                    ta.X = x; ta.Y = y; ta.B = n; ta.nblk = 1; ta.blk = bb->tail_tab + j; ta.dw4 = bb->tail_dw4;
                    ta.dbg_dw = ws.dwbuf; ta.dbg_gate = ws.gate; ta.dbg_clk = ws.pool_part;
                    snprintf(nm, sizeof nm, "b%d.tail", 12 + j);
                    STEP(nm, "tail7", launch_tail7(ta, st));
                    int r;
                    snprintf(nm, sizeof nm, "b%d.dw", 12 + j);
                    if ((r = save_act(bb, nm, ws.dwbuf, (size_t)n * 49 * bb->blk[12 + j].ce, true, st))) return r;
                    snprintf(nm, sizeof nm, "b%d.gate", 12 + j);
                    if ((r = save_act(bb, nm, ws.gate, (size_t)n * bb->blk[12 + j].ce, false, st))) return r;
                    snprintf(nm, sizeof nm, "b%d.out", 12 + j);
                    if ((r = save_act(bb, nm, y, (size_t)n * 49 * bb->blk[12 + j].d.cout, true, st))) return r;
                    snprintf(nm, sizeof nm, "b%d.clk", 12 + j);
                    if ((r = save_act(bb, nm, ws.pool_part, (size_t)n * 8, false, st))) return r;
                    _Float16* t = x; x = y; y = t;
                }
            }
            i = 15;
            continue;
        }
        BlockW& B = bb->blk[i];
        const int HWi = B.H * B.H, HWo = B.Ho * B.Ho;
        int nparts = B.parts;
        // small-K, small-N project on a big image: thin_proj_kernel (up to 5 k-steps since the gate is folded into the weight
        // fragments: B0's b2 30.8 vs 43.7 us on pw_gemm; MMC_THIN_PROJ_KS moves the limit, clamped to what is instantiated, and a
        // shape without an instantiation stays on pw_gemm)
        static const int thin_max = [] { const char* e = getenv("MMC_THIN_PROJ_KS"); const int v = e ? atoi(e) : 5; return v < 0 ? 0 : (v > 6 ? 6 : v); }();
        const bool use_thin = bb->thin_proj && B.project.nt == 2 && B.project.n_chunks == 1 && B.project.Kp / 32 <= thin_max &&
                              thin_proj_has(B.project.Kp / 32) && B.project.N <= 32 &&
                              (B.project.N & 7) == 0 && (HWo & 15) == 0 && HWo >= 3136;
        // block 1's depthwise output as three 32-channel planes between mb1_kernel and thin_proj_kernel (see mb1_kernel); per-tensor
        // mode keeps the interleaved tensor it hands out (MMC_B1_PLANAR=0 switches the planes off)
        const bool b1_planar = i == 1 && use_thin && bb->b1_planar && !bb->keep && B.project.K == 96;
        bool d_planar = false;   // set where mb1_kernel is the producer
        if (i == 11 && bb->tail_full && bb->tail_b11 && !bb->keep) {
            // the whole of block 11, blocks 12..15 and the head conv in ONE launch: from block 10's output to the feature vector
            TailArgs ta{};
            ta.B = n; ta.blk = bb->tail_tab; ta.nblk = 4; ta.dw4 = bb->tail_dw4;
            ta.pre_X = x; ta.pre_wexp = B.exp_frag; ta.pre_bexp = B.expand.b; ta.pre_dwp = B.t_dwp4; ta.pre_bdw = B.dw_b;
            ta.pre_wr_t = B.t_wr; ta.pre_br = B.pp_br; ta.pre_we_t = B.t_we; ta.pre_be = B.se_be; ta.pre_wproj = bb->pre_wproj;
            ta.pre_bproj = B.project.b; ta.inv_hw = (float)(1.0 / (49.0 * LOG2E));
            ta.head_w = bb->head_wfrag; ta.head_b = bb->head.b; ta.feat = out_dev;
            if (bb->tail_clk) {   // debug clock: rows of this lane's patches (out_dev is the lane's slice of the caller's matrix)
                ta.dbg_clk = bb->tail_clk + (size_t)(&ws - bb->lanes) * bb->lane_cap * 64; ta.clk_sections = 1;
            }
            STEP("b11all-head.tail", "tail7", launch_tail7(ta, st));
            tail_done = true;
            break;
        }
        if (i == 0 && stem_fused) {
            // with block 0's project folded into block 1's kernel the depthwise output goes to the spare activation
            // buffer (block 1 reads it while writing its own depthwise output to dwbuf)
            STEP("stem+b0.dw", "stem_dw", launch_stem_dw(patches_dev, bb->stem_w, bb->stem_b, bb->stem_pad, B.dw_w, B.dw_b,
                                                          bb->fuse_b0b1 ? y : ws.dwbuf, ws.pool_part, n, st));
            nparts = 49;
        } else if (B.fused && B.t_dwp && B.exp_frag && (B.d.s == 1 ? bb->mbt : bb->mbt2) && mbt_has(B.H, B.d.k, B.d.s, B.d.cin, B.ce)) {
            MbtArgs ta{};
            ta.X = x; ta.wexp = B.exp_frag; ta.bexp = B.expand.b; ta.dwp = B.t_dwp; ta.bdw = B.dw_b; ta.D = ws.dwbuf;
            ta.pool = ws.pool_part; ta.B = n; ta.H = B.H; ta.Cin = B.d.cin; ta.Ce = B.ce; ta.ks = B.d.k; ta.stride = B.d.s;
            ta.dwtoe = (B.d.s == 1 && B.H == 28 && B.d.k == 5) ? B.dw_diag : nullptr;
            nparts = B.d.s == 2 ? (B.Ho / 7) * (B.Ho / 14) : (B.H / 14) * (B.H / 28);
            snprintf(nm, sizeof nm, "b%d.mbconv", i);
            char ml[48];   // the instantiation's template arguments, as rocprofv3 names it
            snprintf(ml, sizeof ml, "%s<%d,%d,%d,%d>", B.d.s == 2 ? "mbt2" : "mbt", B.d.k, (B.d.cin + 31) / 32, B.ce, B.H);
            if (ta.dwtoe) snprintf(ml, sizeof ml, "mbt4<%d,%d>", (B.d.cin + 31) / 32, B.ce);
            STEP(nm, ml, launch_mbt(ta, st));
        } else if (B.fused && bb->mid14 && ((i >= 6 && i <= bb->mid14_last) || (i == 11 && bb->mid14_b11)) && B.t_dwp && B.exp_frag) {
            Mid14Args ma{};
            ma.X = x; ma.wexp = B.exp_frag; ma.bexp = B.expand.b; ma.dwp = B.t_dwp4; ma.bdw = B.dw_b; ma.D = ws.dwbuf;
            ma.pool = ws.pool_part; ma.B = n; ma.Cin = B.d.cin; ma.Ce = B.ce; ma.ks = B.d.k;
            ma.dwdiag = (B.d.s == 1) ? B.dw_diag : nullptr;
            { const char* e = getenv("MMC_MID14_SPLIT"); ma.nsplit = e ? atoi(e) : (ma.dwdiag ? 1 : (B.d.s == 2 ? 7 : 4)); }
            ma.stride = B.d.s;
            if (bb->mid_clk && i == 10) ma.dbg_clk = bb->mid_clk + (size_t)(&ws - bb->lanes) * bb->lane_cap * 128;
            nparts = 1;
            snprintf(nm, sizeof nm, "b%d.mbconv", i);
            char ml[48];
            snprintf(ml, sizeof ml, ma.dwdiag ? "mid14m<%d,%d,%d,%d>" : "mid14<%d,%d,%d,%d>", (B.d.cin + 31) / 32, B.d.k, B.ce, B.d.s);
            STEP(nm, ml, launch_mid14(ma, st));
        } else if (B.fused) {
            MbArgs a{};
            a.X = x; a.Wexp = B.exp_nat; a.bexp = B.expand.b; a.Wdw = B.dw_w; a.bdw = B.dw_b; a.out = ws.dwbuf;
            a.pool_part = ws.pool_part; a.B = n; a.H = B.H; a.W = B.H; a.Cin = B.d.cin; a.Ce = B.ce; a.Ho = B.Ho;
            a.Wo = B.Ho; a.pad = B.pad; a.ks = B.d.k; a.stride = B.d.s; a.tw = B.f_tw; a.ksteps = B.f_ksteps;
            a.TH = B.f_TH; a.TWo = B.f_TWo; a.tiles_x = B.f_tiles_x; a.tiles_y = B.f_tiles_y; a.CC = B.f_CC;
            a.CCG = B.f_CCG; a.S = B.f_S; a.red_off = B.f_red_off; a.lds_bytes = B.f_lds; a.npair = B.f_npair; a.pb = B.f_pb; a.wlds = B.f_wlds; a.wfr_off = B.f_wfr_off; a.Wfrag = B.exp_frag;
            a.wl_off = B.f_wl_off;
            nparts = B.f_tiles_x * B.f_tiles_y;
            snprintf(nm, sizeof nm, "b%d.mbconv", i);
            char fl[48];
            if (B.use_d) {
                a.npair = B.d_npair; a.wl_off = B.d_wl_off; a.red_off = B.d_red_off; a.lds_bytes = B.d_lds;
                snprintf(fl, sizeof fl, "mbconv_d<%d,%d,%d,%d,%d,%d,%d>", a.ks, a.stride, a.ksteps, a.npair, a.CC, a.TWo, a.pb);
                STEP(nm, fl, launch_mbconv_d(a, st));
            } else {
                snprintf(fl, sizeof fl, "mbconv_a<%d,%d,%d,%d,%d,%d,%d,%d>", a.ks, a.stride, a.tw, a.ksteps, a.npair, a.CC, a.pb, a.wlds);
                if (i == 1 && bb->fuse_b0b1 && bb->mb1) {
                    Mb1Args m1{};
                    m1.X = y; m1.pre_w = bb->b0_pre_w; m1.pre_b = bb->blk[0].project.b; m1.pre_gate = ws.gate; m1.wexp = bb->b1_exp_pre;
                    m1.bexp = B.expand.b; m1.wdw = B.dw_w; m1.bdw = B.dw_b; m1.D = ws.dwbuf; m1.pool = ws.pool_part; m1.B = n;
                    m1.planar = b1_planar ? 1 : 0;
                    d_planar = b1_planar;
                    nparts = 14;
                    snprintf(nm, sizeof nm, "b0.project+b1.mbconv");
                    STEP(nm, "mb1", launch_mb1(m1, st));
                } else if (i == 1 && bb->fuse_b0b1) {
                    a.X = y; a.Cin = 32; a.Wexp = bb->b1_exp_pre;   // block 0's depthwise output; its gate is in ws.gate
                    snprintf(nm, sizeof nm, "b0.project+b1.mbconv");
                    STEP(nm, "mbconv_a_pre", launch_mbconv_pre(a, bb->b0_pre_w, bb->blk[0].project.b, ws.gate, st));
                } else
                STEP(nm, fl, launch_mbconv_a(a, st));
            }
        } else {
            const _Float16* dw_in = x;
            if (B.has_expand) {
                snprintf(nm, sizeof nm, "b%d.expand", i);
                STEP(nm, gemm_label(B.expand, n * HWi, EPI_SILU, false, false),
                     run_gemm(B.expand, x, n * HWi, ws.expbuf, EPI_SILU, nullptr, HWi, nullptr, nullptr, st));
                if (bb->keep) { int r = save_act(bb, nm, ws.expbuf, (size_t)n * HWi * B.ce, true, st); if (r) return r; }
                dw_in = ws.expbuf;
            }
            DwArgs d{};
            d.in = dw_in; d.wt = B.dw_w; d.bias = B.dw_b; d.out = ws.dwbuf; d.pool_part = ws.pool_part;
            d.B = n; d.H = B.H; d.W = B.H; d.C = B.ce; d.Ho = B.Ho; d.Wo = B.Ho; d.pad_t = B.pad; d.pad_l = B.pad;
            d.ks = B.d.k; d.stride = B.d.s; d.tw = B.tw; d.CG = B.CG; d.S = B.S; d.iters = B.iters; d.parts = B.parts; d.nz = B.nz;
            snprintf(nm, sizeof nm, "b%d.dw", i);
            char dl[48];
            snprintf(dl, sizeof dl, "dwconv<%d,%d,%d>", d.ks, d.stride, d.tw);
            STEP(nm, dl, launch_dwconv(d, st));
        }
        snprintf(nm, sizeof nm, "b%d.dw", i);
        if (bb->keep) {
            int r = save_act(bb, nm, (i == 0 && bb->fuse_b0b1 && stem_fused) ? y : ws.dwbuf, (size_t)n * HWo * B.ce, true, st);
            if (r) return r;
        }
        if (i == 11 && bb->tail_full) {
            // block 11's squeeze-excite + project, blocks 12..15 and the head conv in ONE launch (tail7_kernel): from
            // the last 14x14 depthwise output straight to the feature vector.  Per-tensor mode runs it in pieces.
            const BlockW& B11 = B;
            TailArgs ta{};
            ta.B = n; ta.blk = bb->tail_tab; ta.dw4 = bb->tail_dw4;
            ta.pre_D = ws.dwbuf; ta.pre_pool = ws.pool_part; ta.pre_wr_t = B11.t_wr; ta.pre_br = B11.pp_br; ta.pre_we_t = B11.t_we;
            ta.pre_be = B11.se_be; ta.pre_wproj = bb->pre_wproj; ta.pre_bproj = B11.project.b;
            ta.inv_hw = (float)(1.0 / (49.0 * LOG2E));
            if (!bb->keep) {
                ta.nblk = 4; ta.head_w = bb->head_wfrag; ta.head_b = bb->head.b; ta.feat = out_dev;
                STEP("b11-head.tail", "tail7", launch_tail7(ta, st));
                tail_done = true;
                break;
            }
            int r;
            ta.nblk = 0; ta.Y = y; ta.dbg_gate = ws.gate;
            STEP("b11.tail", "tail7", launch_tail7(ta, st));
            if ((r = save_act(bb, "b11.gate", ws.gate, (size_t)n * B.ce, false, st))) return r;
            if ((r = save_act(bb, "b11.out", y, (size_t)n * 49 * 192, true, st))) return r;
            _Float16* t = x; x = y; y = t;
            continue;
        }
        if (B.pp && B.fused && B.t_wrg) {
            // squeeze-excite + project, one patch per workgroup (proj_patch_kernel): no gate tensor, one launch
            ProjPatchArgs pa{};
            pa.X = ws.dwbuf; pa.pool_part = ws.pool_part; pa.wr_g = B.t_wrg; pa.br = B.pp_br; pa.we_t = B.t_we; pa.be = B.se_be;
            pa.wfrag = B.pp_w; pa.bias = B.pp_b; pa.res = B.skip ? x : nullptr; pa.Y = y;
            pa.dbg_gate = bb->keep ? ws.gate : nullptr;
            pa.dbg_clk = bb->keep ? bb->dbg_clk : nullptr;
            pa.B = n; pa.HW = HWo; pa.K = B.ce; pa.N = B.d.cout; pa.CSP = B.pp_csp; pa.nparts = nparts;
            pa.psc = (float)(1.0 / ((double)HWo * LOG2E));
            snprintf(nm, sizeof nm, "b%d.projse", i);
            char pl[48];
            snprintf(pl, sizeof pl, "proj_patch<%d,%d,%d,%d>", proj_patch_ksteps(B.ce), (B.d.cout + 15) / 16, HWo, B.skip ? 1 : 0);
            STEP(nm, pl, launch_proj_patch(pa, st));
            if (bb->keep) {
                int r;
                snprintf(nm, sizeof nm, "b%d.gate", i);
                if ((r = save_act(bb, nm, ws.gate, (size_t)n * B.ce, false, st))) return r;
                snprintf(nm, sizeof nm, "b%d.out", i);
                if ((r = save_act(bb, nm, y, (size_t)n * HWo * B.d.cout, true, st))) return r;
                snprintf(nm, sizeof nm, "b%d.clk", i);
                if ((r = save_act(bb, nm, bb->dbg_clk, (size_t)n * 8, false, st))) return r;
            }
            _Float16* t = x; x = y; y = t;
            continue;
        }
        snprintf(nm, sizeof nm, "b%d.gate", i);
        if (bb->arch != MMC_ARCH_B0)
            STEP(nm, "se_wide", launch_se_wide(ws.pool_part, nparts, n, B.ce, B.cs, B.se_wr_nat, B.se_br_nat, B.se_we_nat, B.se_be,
                                               ws.gate, st));
        else if (B.se_wr_nat && se_small_enabled)
            STEP(nm, "se_small", launch_se_small(ws.pool_part, nparts, n, B.ce, B.cs, B.se_wr_nat, B.se_br_nat, B.se_we_nat, B.se_be,
                                                 ws.gate, st));
        else
        STEP(nm, "se_fused", launch_se_gate(ws.pool_part, nparts, n, B.ce, B.cs4, B.se_wrp, B.se_br, B.se_wep, B.se_be,
                                            ws.gate, st));
        if (bb->keep) { int r = save_act(bb, nm, ws.gate, (size_t)n * B.ce, false, st); if (r) return r; }
        if (i == 0 && bb->fuse_b0b1 && stem_fused) continue;   // block 0's project runs inside block 1's kernel
        snprintf(nm, sizeof nm, "b%d.project", i);
        const int pks = B.project.Kp / 32;
        if (use_thin) {
            // small-K, small-N project on a big image (B4 blocks 0, 1; B0 block 1): stream one patch's fragments per workgroup
            GemmArgs a{};
            a.X = ws.dwbuf; a.M = n * HWo; a.K = B.project.K; a.Wp = B.project.w; a.Kp = B.project.Kp; a.bias = B.project.b; a.Y = y;
            a.N = B.project.N; a.nt = B.project.nt; a.n_chunks = B.project.n_chunks; a.epi = EPI_LINEAR; a.gate = ws.gate; a.HW = HWo;
            a.res = B.skip ? x : nullptr;
            a.x_plane_rows = d_planar ? n * HWo : 0;
            STEP(nm, "thin_proj", launch_thin_proj(a, n, st));
        } else if (B.p8.w8) {
            Fp8GemmArgs fa{};
            fa.X = ws.dwbuf; fa.M = n * HWo; fa.K = B.p8.K; fa.W8 = B.p8.w8; fa.KS128 = B.p8.KS128; fa.NFp = B.p8.NFp; fa.sw = B.p8.sw;
            fa.bias = B.p8.b; fa.Y = y; fa.N = B.p8.N; fa.gate = ws.gate; fa.HW = HWo; fa.res = B.skip ? x : nullptr;
            STEP(nm, "pw_gemm_fp8", launch_pw_gemm_fp8(fa, st));
        } else
        STEP(nm, gemm_label(B.project, n * HWo, EPI_LINEAR, true, B.skip), run_gemm(B.project, ws.dwbuf, n * HWo, y, EPI_LINEAR, ws.gate, HWo, B.skip ? x : nullptr, nullptr, st));
        snprintf(nm, sizeof nm, "b%d.out", i);
        if (bb->keep) { int r = save_act(bb, nm, y, (size_t)n * HWo * B.d.cout, true, st); if (r) return r; }
        _Float16* t = x; x = y; y = t;
    }
    const int HWh = bb->blk[bb->nblk - 1].Ho * bb->blk[bb->nblk - 1].Ho;
    if (tail_done) {
        // features already written by tail7_kernel
    } else if (bb->tail_full) {   // per-tensor mode: the head phase of tail7_kernel on its own
        TailArgs ta{};
        ta.X = x; ta.B = n; ta.nblk = 0; ta.blk = bb->tail_tab; ta.in_wide = 1;
        ta.head_w = bb->head_wfrag; ta.head_b = bb->head.b; ta.feat = out_dev; ta.inv_hw = (float)(1.0 / (49.0 * LOG2E));
        STEP("head.tail", "tail7", launch_tail7(ta, st));
    } else
        STEP("head", gemm_label(bb->head, n * HWh, EPI_GAP, false, false), run_gemm(bb->head, x, n * HWh, nullptr, EPI_GAP, nullptr, HWh, nullptr, out_dev, st));
    bb->last_n = n;
#undef STEP
    return 0;
}

// One pass over n resident patches: split into lanes, fork from / join to the caller's stream ONCE.  n may exceed
// max_batch: the pass is then a sequence of max_batch-sized chunks, and each lane walks its sub-batch of every chunk on its
// own stream without waiting for the other lanes -- the drain of one chunk (the last lane's per-patch kernels, which fill
// half the chip) overlaps the first kernels of the next chunk.  Sub-batch shapes are those of separate calls, so results are
// bitwise the same.
static int forward_pass(mmc_backbone* bb, const uint8_t* patches_dev, int n, float* out_dev, hipStream_t st, Prof* prof)
{
    // Profiling records HIP events on each lane's own stream, i.e. durations as they are with the lanes running
    // concurrently (what rocprofv3 sees); MMC_PROFILE_SERIAL=1 profiles one lane at a time instead (isolated kernels).
    static const bool serial_prof = [] { const char* e = getenv("MMC_PROFILE_SERIAL"); return e && e[0] == '1'; }();
    const size_t psz = (size_t)IMG * IMG * 3, FEAT = (size_t)bb->feat;
    if (bb->nlanes == 1 || (prof && serial_prof) || n < 2 * bb->nlanes) {
        for (int off = 0; off < n; off += bb->lane_cap) {
            const int cur = n - off < bb->lane_cap ? n - off : bb->lane_cap;
            int r = forward_lane(bb, bb->lanes[0], patches_dev + (size_t)off * psz, cur, out_dev + (size_t)off * FEAT, st, prof);
            if (r) return r;
        }
        return 0;
    }
    HIP_TRY(hipEventRecord(bb->fork, st));
    for (int l = 0; l < bb->nlanes; ++l) HIP_TRY(hipStreamWaitEvent(bb->lanes[l].stream, bb->fork, 0));
    for (int base = 0; base < n; base += bb->max_batch) {          // chunk-major enqueue: the lanes' launches interleave on the host
        const int cn = n - base < bb->max_batch ? n - base : bb->max_batch;
        const int per = (cn + bb->nlanes - 1) / bb->nlanes;
        for (int l = 0; l < bb->nlanes; ++l) {
            const int off = l * per;
            const int cur = cn - off < per ? cn - off : per;
            if (cur <= 0) break;
            mmc_backbone::Lane& L = bb->lanes[l];
            int r = forward_lane(bb, L, patches_dev + (size_t)(base + off) * psz, cur, out_dev + (size_t)(base + off) * FEAT, L.stream, prof);
            if (r) return r;
        }
    }
    for (int l = 0; l < bb->nlanes; ++l) {
        mmc_backbone::Lane& L = bb->lanes[l];
        HIP_TRY(hipEventRecord(L.done, L.stream));
        HIP_TRY(hipStreamWaitEvent(st, L.done, 0));
    }
    return 0;
}

// forward_pass through a cached HIP graph (device-resident buffers only); falls back to plain launches on any error
static int run_pass(mmc_backbone* bb, const uint8_t* pin, int n, float* pout, hipStream_t st)
{
    if (!bb->use_graph || n > 8 * bb->max_batch) return forward_pass(bb, pin, n, pout, st, nullptr);   // (bounded graph size)
    for (size_t gi = 0; gi < bb->graphs.size(); ++gi) {
        const mmc_backbone::GraphEntry g = bb->graphs[gi];
        if (g.in == pin && g.out == pout && g.n == n) {
            if (gi + 1 != bb->graphs.size()) {   // most recently used last
                bb->graphs.erase(bb->graphs.begin() + (long)gi);
                bb->graphs.push_back(g);
            }
            HIP_TRY(hipGraphLaunch(g.exec, st));
            return 0;
        }
    }
    if (bb->evictions >= MMC_GRAPH_CACHE) return forward_pass(bb, pin, n, pout, st, nullptr);   // cache frozen: see `evicted`
    for (auto& k : bb->evicted)
        if (k.in == pin && k.out == pout && k.n == n) return forward_pass(bb, pin, n, pout, st, nullptr);
    bool repeat = false;
    for (auto& k : bb->seen) repeat = repeat || (k.in == pin && k.out == pout && k.n == n);
    if (!repeat) {
        if (bb->seen.size() >= 2 * MMC_GRAPH_CACHE) bb->seen.erase(bb->seen.begin());
        bb->seen.push_back({pin, pout, n});
    }
    if (bb->graph_warm < 2 || !repeat) {   // first passes un-captured: lazy per-kernel attribute setup happens there
        ++bb->graph_warm;
        return forward_pass(bb, pin, n, pout, st, nullptr);
    }
    if (!bb->gstream && hipStreamCreateWithFlags(&bb->gstream, hipStreamNonBlocking) != hipSuccess) {
        bb->use_graph = false;
        return forward_pass(bb, pin, n, pout, st, nullptr);
    }
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool ok = hipStreamBeginCapture(bb->gstream, hipStreamCaptureModeRelaxed) == hipSuccess;
    int r = ok ? forward_pass(bb, pin, n, pout, bb->gstream, nullptr) : 0;
    if (ok) ok = hipStreamEndCapture(bb->gstream, &graph) == hipSuccess && r == 0 && graph;
    if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph) hipGraphDestroy(graph);
    if (!ok) {
        (void)hipGetLastError();
        bb->use_graph = false;
        return forward_pass(bb, pin, n, pout, st, nullptr);
    }
    if (bb->graphs.size() >= MMC_GRAPH_CACHE) {
        // least recently used graph goes; it may still be executing on a stream this handle was given earlier: passes of a
        // handle are serialised (PassOrder), so waiting for the last pass's completion event covers every earlier replay
        if (bb->have_last && hipEventSynchronize(bb->last_done) != hipSuccess) {
            (void)hipGetLastError();
            hipGraphExecDestroy(exec);
            return forward_pass(bb, pin, n, pout, st, nullptr);
        }
        const mmc_backbone::GraphEntry old = bb->graphs.front();
        hipGraphExecDestroy(old.exec);
        bb->graphs.erase(bb->graphs.begin());
        if (bb->evicted.size() >= 8 * MMC_GRAPH_CACHE) bb->evicted.erase(bb->evicted.begin());
        bb->evicted.push_back({old.in, old.out, old.n});
        ++bb->evictions;
    }
    bb->graphs.push_back({pin, pout, n, exec});
    ++bb->captures;
    HIP_TRY(hipGraphLaunch(exec, st));
    return 0;
}

extern "C" int mmc_backbone_graph_stats(mmc_backbone* bb, int64_t stats[3])
{
    if (!bb || !stats) return fail(MMC_ERR_ARG, "backbone handle / stats is NULL");
    std::lock_guard<std::mutex> lock(bb->mu);
    stats[0] = bb->captures; stats[1] = bb->evictions; stats[2] = (int64_t)bb->graphs.size();
    return MMC_OK;
}

extern "C" int mmc_backbone_extract(mmc_backbone* bb, const void* patches, int64_t n, float* out_features,
                                    unsigned flags, void* hip_stream)
{
    if (!bb) return fail(MMC_ERR_ARG, "backbone handle is NULL");
    if (n < 0) return fail(MMC_ERR_ARG, "n = %lld is negative", (long long)n);
    if (n == 0) return MMC_OK;
    if (!patches || !out_features) return fail(MMC_ERR_ARG, "patches/out_features is NULL");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    std::lock_guard<std::mutex> lock(bb->mu);
    HIP_TRY(hipSetDevice(bb->device));
    PassOrder order{bb, st};
    { int r = order.begin(); if (r) return r; }
    const size_t psz = (size_t)IMG * IMG * 3;
    const uint8_t* in = static_cast<const uint8_t*>(patches);
    if (!(flags & (MMC_IN_HOST | MMC_OUT_HOST)) && !bb->keep) {
        // device-resident buffers: the whole call is one pass (pipelined over max_batch-sized chunks when n is larger)
        for (int64_t off = 0; off < n; off += (1 << 20)) {
            const int cur = (int)((n - off) < (1 << 20) ? (n - off) : (1 << 20));
            int r = run_pass(bb, in + (size_t)off * psz, cur, out_features + (size_t)off * bb->feat, st);
            if (r) return r;
        }
        return MMC_OK;
    }
    for (int64_t off = 0; off < n; off += bb->max_batch) {
        const int cur = (int)((n - off) < bb->max_batch ? (n - off) : bb->max_batch);
        const uint8_t* pin = in + (size_t)off * psz;
        if (flags & MMC_IN_HOST) {
            HIP_TRY(hipMemcpyAsync(bb->in_stage, pin, (size_t)cur * psz, hipMemcpyHostToDevice, st));
            pin = bb->in_stage;
        }
        const size_t FEAT = (size_t)bb->feat;
        float* pout = (flags & MMC_OUT_HOST) ? bb->out_stage : out_features + (size_t)off * FEAT;
        int r = run_pass(bb, pin, cur, pout, st);   // (host buffers go through the fixed staging buffers: same graph every call)
        if (r) return r;
        if (flags & MMC_OUT_HOST)
            HIP_TRY(hipMemcpyAsync(out_features + (size_t)off * FEAT, bb->out_stage, (size_t)cur * FEAT * sizeof(float),
                                   hipMemcpyDeviceToHost, st));
    }
    if (flags & MMC_OUT_HOST) HIP_TRY(hipStreamSynchronize(st));
    return MMC_OK;
}

extern "C" int mmc_backbone_read_activation(mmc_backbone* bb, const char* name, float* out, size_t capacity,
                                            size_t* n_written)
{
    if (!bb || !name || !out) return fail(MMC_ERR_ARG, "NULL argument");
    if (bb->mid_clk && strcmp(name, "mid14.clk") == 0) {
        const size_t ne = (size_t)bb->max_batch * 128;
        if (ne > capacity) return fail(MMC_ERR_ARG, "'mid14.clk' has %zu elements, capacity %zu", ne, capacity);
        HIP_TRY(hipSetDevice(bb->device));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out, bb->mid_clk, ne * 4, hipMemcpyDeviceToHost));
        if (n_written) *n_written = ne;
        return MMC_OK;
    }
    if (bb->tail_clk && strcmp(name, "tail.clk") == 0) {   // MMC_TAIL_CLK=1: phase clock of the last production pass
        const size_t ne = (size_t)bb->max_batch * 64;
        if (ne > capacity) return fail(MMC_ERR_ARG, "'tail.clk' has %zu elements, capacity %zu", ne, capacity);
        HIP_TRY(hipSetDevice(bb->device));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out, bb->tail_clk, ne * 4, hipMemcpyDeviceToHost));
        if (n_written) *n_written = ne;
        return MMC_OK;
    }
    if (!bb->keep) return fail(MMC_ERR_ARG, "activations are not kept: set MMC_KEEP_ACTIVATIONS=1 before create");
    auto it = bb->saved.find(name);
    if (it == bb->saved.end()) return fail(MMC_ERR_ARG, "no saved activation named '%s'", name);
    const Saved& s = it->second;
    if (s.elems > capacity) return fail(MMC_ERR_ARG, "'%s' has %zu elements, capacity %zu", name, s.elems, capacity);
    HIP_TRY(hipSetDevice(bb->device));
    HIP_TRY(hipDeviceSynchronize());
    if (s.is_half) {
        std::vector<_Float16> tmp(s.elems);
        HIP_TRY(hipMemcpy(tmp.data(), s.dev, s.elems * 2, hipMemcpyDeviceToHost));
        const size_t ln = strlen(name);
        const bool scaled = strcmp(name, "stem") == 0 || (ln > 3 && strcmp(name + ln - 3, ".dw") == 0) ||
                            (ln > 7 && strcmp(name + ln - 7, ".expand") == 0);   // SiLU outputs live times log2(e)
        const float inv = scaled ? (float)(1.0 / LOG2E) : 1.0f;
        for (size_t i = 0; i < s.elems; ++i) out[i] = (float)tmp[i] * inv;
    } else {
        HIP_TRY(hipMemcpy(out, s.dev, s.elems * 4, hipMemcpyDeviceToHost));
    }
    if (n_written) *n_written = s.elems;
    return MMC_OK;
}

extern "C" int mmc_backbone_profile(mmc_backbone* bb, const void* patches_dev, int64_t n, float* out_features_dev,
                                    void* hip_stream, char (*names)[64], float* ms, int* launches, int cap, int* n_out)
{
    if (!bb || !patches_dev || !out_features_dev || !names || !ms || !n_out)
        return fail(MMC_ERR_ARG, "NULL argument");
    if (n < 1 || n > bb->max_batch) return fail(MMC_ERR_ARG, "n must be in [1, max_batch]");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    std::lock_guard<std::mutex> lock(bb->mu);
    HIP_TRY(hipSetDevice(bb->device));
    PassOrder order{bb, st};
    { int r = order.begin(); if (r) return r; }
    Prof prof;
    int r = forward_pass(bb, static_cast<const uint8_t*>(patches_dev), (int)n, out_features_dev, st, &prof);
    if (r) return r;
    HIP_TRY(hipStreamSynchronize(st));
    int cnt = 0;
    for (auto& e : prof.entries) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, e.e0, e.e1));
        hipEventDestroy(e.e0);
        hipEventDestroy(e.e1);
        if (cnt < cap) {
            snprintf(names[cnt], 64, "%s", e.name.c_str());
            ms[cnt] = t;
            if (launches) launches[cnt] = 1;
            ++cnt;
        }
    }
    *n_out = cnt;
    return MMC_OK;
}

// ------------------------------------------------------------------------------------------
// crop front-end
// ------------------------------------------------------------------------------------------
// Host-side cut for sparse points: pinned ring of 4 slots; a slot is reused only after the H2D copy that read it has
// completed (event), so calls stay asynchronous on the caller's stream.  One thread at a time per process (mutex).
namespace {
struct PinnedSlot { void* host = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool busy = false; };
PinnedSlot g_slots[4];
int g_slot_next = 0;
std::mutex g_slot_mu;
// upload path of mmc_crop_patches (dense points on a host image): per-device staging buffers kept between calls
struct CropStage { void* img = nullptr; size_t img_cap = 0; void* rc = nullptr; size_t rc_cap = 0; hipEvent_t ev = nullptr; };
CropStage g_crop_stage[16];
}  // namespace

static int crop_on_host(const uint8_t* img, int H, int W, const int32_t* rowcols, int64_t n, void* out_dev, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(g_slot_mu);
    PinnedSlot& s = g_slots[g_slot_next];
    g_slot_next = (g_slot_next + 1) & 3;
    const size_t psz = (size_t)IMG * IMG * 3, need = (size_t)n * psz;
    if (s.busy) { HIP_TRY(hipEventSynchronize(s.ev)); s.busy = false; }
    if (!s.ev) HIP_TRY(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
    if (s.cap < need) {
        if (s.host) hipHostFree(s.host);
        s.host = nullptr; s.cap = 0;
        HIP_TRY(hipHostMalloc(&s.host, need, hipHostMallocDefault));
        s.cap = need;
    }
    auto refl = [](int i, int nn) { i = i < 0 ? -i : i; return i >= nn ? 2 * (nn - 1) - i : i; };   // numpy 'reflect'
    uint8_t* dst = static_cast<uint8_t*>(s.host);
    auto cut = [&](int64_t p0, int64_t p1) {
        for (int64_t p = p0; p < p1; ++p) {
            const int row = rowcols[2 * p], col = rowcols[2 * p + 1];
            const int sx0 = col - IMG / 2;
            const bool inside = sx0 >= 0 && sx0 + IMG <= W;
            for (int y = 0; y < IMG; ++y) {
                const int sy = refl(row - IMG / 2 + y, H);
                uint8_t* d = dst + (size_t)p * psz + (size_t)y * IMG * 3;
                if (inside) {
                    memcpy(d, img + ((size_t)sy * W + sx0) * 3, (size_t)IMG * 3);
                } else {
                    for (int x = 0; x < IMG; ++x) {
                        const uint8_t* src = img + ((size_t)sy * W + refl(sx0 + x, W)) * 3;
                        d[3 * x] = src[0]; d[3 * x + 1] = src[1]; d[3 * x + 2] = src[2];
                    }
                }
            }
        }
    };
    static const int max_threads = [] { const char* e = getenv("MMC_CROP_THREADS"); const int v = e ? atoi(e) : 4; return v < 1 ? 1 : (v > 4 ? 4 : v); }();
    const int nthreads = n >= 8 ? max_threads : 1;   // the cut is memcpy-bound: a few threads saturate what one core cannot
    if (nthreads == 1) cut(0, n);
    else {
        std::thread th[3];
        const int64_t per = (n + nthreads - 1) / nthreads;
        for (int t = 1; t < nthreads; ++t) th[t - 1] = std::thread(cut, t * per < n ? t * per : n, (t + 1) * per < n ? (t + 1) * per : n);
        cut(0, per < n ? per : n);
        for (int t = 1; t < nthreads; ++t) th[t - 1].join();
    }
    HIP_TRY(hipMemcpyAsync(out_dev, s.host, need, hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(s.ev, st));
    s.busy = true;
    return MMC_OK;
}

extern "C" int mmc_crop_patches(const void* image, int height, int width, const int32_t* rowcols, int64_t n,
                                void* patches_out_dev, unsigned flags, int device, void* hip_stream)
{
    if (!image || !rowcols || !patches_out_dev) return fail(MMC_ERR_ARG, "NULL argument");
    if (n < 0 || n > 65535) return fail(MMC_ERR_ARG, "n = %lld out of range [0,65535]", (long long)n);
    if (n == 0) return MMC_OK;
    if (height <= IMG || width <= IMG)
        return fail(MMC_ERR_ARG, "image %dx%d must exceed the crop size %d in both dimensions", height, width, IMG);
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    HIP_TRY(hipSetDevice(device));
    const uint8_t* img = static_cast<const uint8_t*>(image);
    const int32_t* rc = rowcols;
    if (flags & MMC_IN_HOST) {
        for (int64_t i = 0; i < n; ++i) {
            const int r = rowcols[2 * i], c = rowcols[2 * i + 1];
            if (r < 0 || r >= height || c < 0 || c >= width)
                return fail(MMC_ERR_ARG, "point %lld (%d,%d) outside the %dx%d image", (long long)i, r, c, height, width);
        }
        const size_t ib = (size_t)height * width * 3;
        // Sparse points on a big host image (the reference's data: 10-25 points on a 27 MP image): uploading the image moves
        // 81 MB to cut 3.8 MB of patches.  Cut them on the host instead -- same index arithmetic, row memcpys into a pinned
        // ring slot -- and upload only the patches.  Dense points keep the upload + crop_kernel path.  MMC_CROP_HOST=0/1 forces.
        static const int force = [] { const char* e = getenv("MMC_CROP_HOST"); return e ? atoi(e) : -1; }();
        const bool host_crop = force >= 0 ? force != 0 : (size_t)n * IMG * IMG * 3 * 6 <= ib;
        if (host_crop) return crop_on_host(img, height, width, rowcols, n, patches_out_dev, st);
        // dense points: upload the image once into a per-device staging buffer that is kept between calls (grown on demand)
        std::lock_guard<std::mutex> lock(g_slot_mu);
        CropStage& cs = g_crop_stage[device & 15];
        const size_t rb = (size_t)n * 8;
        if (cs.img_cap < ib) {
            if (cs.img) { hipStreamSynchronize(st); hipFree(cs.img); }
            cs.img = nullptr; cs.img_cap = 0;
            HIP_TRY(hipMalloc(&cs.img, ib));
            cs.img_cap = ib;
        }
        if (cs.rc_cap < rb) {
            if (cs.rc) { hipStreamSynchronize(st); hipFree(cs.rc); }
            cs.rc = nullptr; cs.rc_cap = 0;
            HIP_TRY(hipMalloc(&cs.rc, rb));
            cs.rc_cap = rb;
        }
        // the previous call's kernel may still be reading the staging buffers on another stream
        if (cs.ev) HIP_TRY(hipStreamWaitEvent(st, cs.ev, 0));
        else HIP_TRY(hipEventCreateWithFlags(&cs.ev, hipEventDisableTiming));
        HIP_TRY(hipMemcpyAsync(cs.img, image, ib, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(cs.rc, rowcols, rb, hipMemcpyHostToDevice, st));
        int r = launch_crop(static_cast<const uint8_t*>(cs.img), height, width, static_cast<const int32_t*>(cs.rc), (int)n,
                            static_cast<uint8_t*>(patches_out_dev), st);
        HIP_TRY(hipEventRecord(cs.ev, st));
        // the host image / rowcols are pageable caller memory the caller may free or overwrite right after this call
        HIP_TRY(hipStreamSynchronize(st));
        if (r) return fail(MMC_ERR_HIP, "crop kernel launch failed (%d)", r);
        return MMC_OK;
    }
    // device-resident image and points: the kernel clamps each point into the image (see crop_kernel)
    int r = launch_crop(img, height, width, rc, (int)n, static_cast<uint8_t*>(patches_out_dev), st);
    if (r) return fail(MMC_ERR_HIP, "crop kernel launch failed (%d)", r);
    return MMC_OK;
}

// ------------------------------------------------------------------------------------------
// calibrated MLP head
// ------------------------------------------------------------------------------------------
struct mmc_head {
    int device = 0, n_layers = 0, K = 0, input_dim = 0, in_pad = 0;
    std::vector<int> dims_pad;        // padded widths (multiples of 4), last = K (unpadded)
    std::vector<float*> W, b;         // device
    float *a = nullptr, *bc = nullptr;
    float *buf0 = nullptr, *buf1 = nullptr, *in_stage = nullptr, *proba_stage = nullptr;
    int32_t* arg_stage = nullptr;
    int64_t cap_rows = 0;
};

extern "C" void mmc_head_destroy(mmc_head* h)
{
    if (!h) return;
    hipSetDevice(h->device);
    for (float* p : h->W) hipFree(p);
    for (float* p : h->b) hipFree(p);
    hipFree(h->a); hipFree(h->bc); hipFree(h->buf0); hipFree(h->buf1);
    hipFree(h->in_stage); hipFree(h->proba_stage); hipFree(h->arg_stage);
    delete h;
}

extern "C" int mmc_head_create(const float* const* W, const float* const* b, const int* dims, int n_layers,
                               const float* a, const float* bcal, int K, int device, mmc_head** out)
{
    if (!out) return fail(MMC_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!W || !b || !dims || !a || !bcal) return fail(MMC_ERR_ARG, "NULL argument");
    if (n_layers < 1 || n_layers > 16) return fail(MMC_ERR_ARG, "n_layers %d out of range [1,16]", n_layers);
    if (K <= 2) return fail(MMC_ERR_ARG, "CalibratedHead only supports the multiclass (K > 2) path; got K=%d", K);
    if (dims[n_layers] != K) return fail(MMC_ERR_ARG, "dims[n_layers]=%d != K=%d", dims[n_layers], K);
    for (int l = 0; l <= n_layers; ++l)
        if (dims[l] < 1) return fail(MMC_ERR_ARG, "dims[%d]=%d must be positive", l, dims[l]);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(MMC_ERR_HIP, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(MMC_ERR_ARG, "device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    mmc_head* h = new mmc_head();
    h->device = device; h->n_layers = n_layers; h->K = K; h->input_dim = dims[0];
    h->dims_pad.resize(n_layers + 1);
    for (int l = 0; l <= n_layers; ++l) h->dims_pad[l] = (l == n_layers) ? dims[l] : (dims[l] + 3) / 4 * 4;
    h->in_pad = h->dims_pad[0];
    for (int l = 0; l < n_layers; ++l) {
        const int kin = dims[l], kp = h->dims_pad[l], nout = dims[l + 1], np = h->dims_pad[l + 1];
        std::vector<float> wp((size_t)np * kp, 0.f), bp(np, 0.f);
        for (int n = 0; n < nout; ++n) {
            memcpy(&wp[(size_t)n * kp], W[l] + (size_t)n * kin, (size_t)kin * sizeof(float));
            bp[n] = b[l][n];
        }
        float *dw = nullptr, *db = nullptr;
        if (hipMalloc((void**)&dw, wp.size() * 4 + 256) != hipSuccess || hipMalloc((void**)&db, bp.size() * 4 + 256) != hipSuccess) {
            mmc_head_destroy(h);
            return fail(MMC_ERR_NOMEM, "hipMalloc failed for head layer %d", l);
        }
        h->W.push_back(dw); h->b.push_back(db);
        hipMemcpy(dw, wp.data(), wp.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(db, bp.data(), bp.size() * 4, hipMemcpyHostToDevice);
    }
    if (hipMalloc((void**)&h->a, K * 4 + 256) != hipSuccess || hipMalloc((void**)&h->bc, K * 4 + 256) != hipSuccess) {
        mmc_head_destroy(h);
        return fail(MMC_ERR_NOMEM, "hipMalloc failed for calibration parameters");
    }
    hipMemcpy(h->a, a, K * 4, hipMemcpyHostToDevice);
    hipMemcpy(h->bc, bcal, K * 4, hipMemcpyHostToDevice);
    *out = h;
    return MMC_OK;
}

extern "C" int mmc_head_input_dim(const mmc_head* h) { return h ? h->input_dim : 0; }
extern "C" int mmc_head_num_classes(const mmc_head* h) { return h ? h->K : 0; }

static int head_reserve(mmc_head* h, int64_t rows)
{
    if (rows <= h->cap_rows) return 0;
    hipFree(h->buf0); hipFree(h->buf1); hipFree(h->in_stage); hipFree(h->proba_stage); hipFree(h->arg_stage);
    h->buf0 = h->buf1 = h->in_stage = h->proba_stage = nullptr; h->arg_stage = nullptr; h->cap_rows = 0;
    int wmax = h->K;
    for (int d : h->dims_pad) wmax = d > wmax ? d : wmax;
    const size_t nb = (size_t)rows * wmax * 4 + 256;
    HIP_TRY(hipMalloc((void**)&h->buf0, nb));
    HIP_TRY(hipMalloc((void**)&h->buf1, nb));
    HIP_TRY(hipMalloc((void**)&h->in_stage, (size_t)rows * h->in_pad * 4 + 256));
    HIP_TRY(hipMalloc((void**)&h->proba_stage, (size_t)rows * h->K * 4 + 256));
    HIP_TRY(hipMalloc((void**)&h->arg_stage, (size_t)rows * 4 + 256));
    h->cap_rows = rows;
    return 0;
}

extern "C" int mmc_head_predict(mmc_head* h, const float* feats, int64_t n, float* proba, int32_t* argmax,
                                unsigned flags, void* hip_stream)
{
    if (!h) return fail(MMC_ERR_ARG, "head handle is NULL");
    if (n < 0) return fail(MMC_ERR_ARG, "n = %lld is negative", (long long)n);
    if (n == 0) return MMC_OK;
    if (!feats || !proba) return fail(MMC_ERR_ARG, "feats/proba is NULL");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    HIP_TRY(hipSetDevice(h->device));
    const int64_t chunk = 65536;
    for (int64_t off = 0; off < n; off += chunk) {
        const int cur = (int)((n - off) < chunk ? (n - off) : chunk);
        int r = head_reserve(h, cur);
        if (r) return r;
        const float* x = feats + (size_t)off * h->input_dim;
        const hipMemcpyKind kin = (flags & MMC_IN_HOST) ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
        if ((flags & MMC_IN_HOST) || h->in_pad != h->input_dim) {
            if (h->in_pad != h->input_dim) HIP_TRY(hipMemsetAsync(h->in_stage, 0, (size_t)cur * h->in_pad * 4, st));
            HIP_TRY(hipMemcpy2DAsync(h->in_stage, (size_t)h->in_pad * 4, x, (size_t)h->input_dim * 4,
                                     (size_t)h->input_dim * 4, cur, kin, st));
            x = h->in_stage;
        }
        float* pa = h->buf0;
        float* pb = h->buf1;
        for (int l = 0; l < h->n_layers; ++l) {
            const bool last = l == h->n_layers - 1;
            KTRY(launch_mlp_layer(x, cur, h->dims_pad[l], h->W[l], h->b[l], pa, h->dims_pad[l + 1], !last, st));
            x = pa;
            float* t = pa; pa = pb; pb = t;
        }
        float* pout = (flags & MMC_OUT_HOST) ? h->proba_stage : proba + (size_t)off * h->K;
        int32_t* aout = argmax ? ((flags & MMC_OUT_HOST) ? h->arg_stage : argmax + off) : nullptr;
        KTRY(launch_calibrate(x, cur, h->K, h->a, h->bc, pout, aout, st));
        if (flags & MMC_OUT_HOST) {
            HIP_TRY(hipMemcpyAsync(proba + (size_t)off * h->K, pout, (size_t)cur * h->K * 4, hipMemcpyDeviceToHost, st));
            if (argmax) HIP_TRY(hipMemcpyAsync(argmax + off, aout, (size_t)cur * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
    }
    return MMC_OK;
}
