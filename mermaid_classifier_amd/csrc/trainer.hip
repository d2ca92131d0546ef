// trainer.hip -- mini-batch Adam training of the MLP classifier on precomputed feature vectors (gfx950).
//
// Replaces the arithmetic of TorchMLPClassifier.partial_fit (reference
// mermaid_classifier/pyspacer/torch_classifier.py:226-303: per mini-batch zero_grad -> logits -> weighted
// cross-entropy (mean over sum of weights) + (0.5*alpha/mb) * sum(W^2) -> backward -> torch.optim.Adam.step), as driven
// by the trainer's batch loop (mermaid_classifier/pyspacer/trainer.py:138-145).  fp32 end to end on the exact-f32 MFMA
// (v_mfma_f32_16x16x4_f32); every reduction runs in a fixed order (no atomics), so a run is bit-reproducible.
//
// One step on a resident mini-batch H0 [mb][d0]:
//   forward   H(l+1) = relu(H(l) W(l)^T + b(l)), logits = H(L) without relu           gemm<TA=0,TB=1>, EPI_BIAS(_RELU)
//   loss      per-row weighted CE, dZ(L) = w(y)/sum_w * (softmax - onehot)             ce_grad_kernel
//   backward  dW(l) = dZ(l+1)^T H(l) + (alpha/mb) W(l)                                 gemm<TA=1,TB=0>, EPI_L2
//             db(l) = column sums of dZ(l+1)                                           colsum_kernel
//             dZ(l) = (dZ(l+1) W(l)) * (H(l) > 0)                                      gemm<TA=0,TB=0>, EPI_MASK
//   update    Adam exactly in torch's order of operations (lerp, addcmul, addcdiv)     adam_kernel
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <vector>

#include "../../include/mmc.h"
#include "kernels.h"

namespace {

enum { TEPI_NONE = 0, TEPI_BIAS_RELU = 1, TEPI_BIAS = 2, TEPI_MASK = 3, TEPI_L2 = 4 };

// C[M][N] = op(A)[M][K] . op(B)[K][N] (+ epilogue).  TA: A is stored [K][M]; TB: B is stored [N][K].
// 64x64 output tile per 256-thread workgroup, K in steps of 16 through LDS (k-major tiles, so every transpose flavour is
// index arithmetic at staging time); wave w owns the 32x32 quadrant (w>>1, w&1) as 2x2 MFMA fragments.
template <bool TA, bool TB, int EPI>
__global__ __launch_bounds__(256) void tgemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                        float* __restrict__ C, int M, int N, int K,
                                                        const float* __restrict__ aux,   // bias[N] | mask[M][N] | W[M][N]
                                                        float scale)                     // EPI_L2: alpha / mb
{
    __shared__ float As[16][68];
    __shared__ float Bs[16][68];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int li = lane & 15, lq = lane >> 4;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int idx = tid + 256 * r;
            {
                const int m = TA ? (idx & 63) : (idx >> 4), k = TA ? (idx >> 6) : (idx & 15);
                const int gm = m0 + m, gk = k0 + k;
                float v = 0.f;
                if (gm < M && gk < K) v = TA ? A[(size_t)gk * M + gm] : A[(size_t)gm * K + gk];
                As[k][m] = v;
            }
            {
                const int n = TB ? (idx >> 4) : (idx & 63), k = TB ? (idx & 15) : (idx >> 6);
                const int gn = n0 + n, gk = k0 + k;
                float v = 0.f;
                if (gn < N && gk < K) v = TB ? B[(size_t)gn * K + gk] : B[(size_t)gk * N + gn];
                Bs[k][n] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
            float av[2], bv[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                av[f] = As[kk + lq][wm + f * 16 + li];
                bv[f] = Bs[kk + lq][wn + f * 16 + li];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }
    // lane (li, lq) holds C[m0 + wm + 16a + 4 lq + r][n0 + wn + 16b + li]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + 16 * a + 4 * lq + r, n = n0 + wn + 16 * b + li;
                if (m >= M || n >= N) continue;
                float v = acc[a][b][r];
                if (EPI == TEPI_BIAS_RELU) v = fmaxf(v + aux[n], 0.f);
                if (EPI == TEPI_BIAS) v = v + aux[n];
                if (EPI == TEPI_MASK) v = aux[(size_t)m * N + n] > 0.f ? v : 0.f;
                if (EPI == TEPI_L2) v = v + scale * aux[(size_t)m * N + n];
                C[(size_t)m * N + n] = v;
            }
}

// One wave per row: log-softmax, weighted negative log-likelihood, gradient of the weighted-mean loss w.r.t. the logits.
// F.cross_entropy(logits, y, weight=w) = sum_i w[y_i] * nll_i / sum_i w[y_i]   (torch_classifier.py:278)
__global__ __launch_bounds__(256) void ce_grad_kernel(const float* __restrict__ logits, const int32_t* __restrict__ y, int M, int K,
                                                      const float* __restrict__ cw,   // [K] or null
                                                      float inv_wsum, float* __restrict__ dlogits, float* __restrict__ row_loss)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* z = logits + (size_t)row * K;
    float mx = -INFINITY;
    for (int c = lane; c < K; c += 64) mx = fmaxf(mx, z[c]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float s = 0.f;
    for (int c = lane; c < K; c += 64) s += expf(z[c] - mx);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const int yi = y[row];
    const float w = cw ? cw[yi] : 1.0f;
    const float lse = logf(s);
    const float g = w * inv_wsum;
    for (int c = lane; c < K; c += 64) {
        const float p = expf(z[c] - mx - lse);
        dlogits[(size_t)row * K + c] = g * (p - (c == yi ? 1.0f : 0.0f));
    }
    if (lane == 0) row_loss[row] = w * (lse - (z[yi] - mx)) * inv_wsum;
}

// db[n] = sum_m dZ[m][n]; thread per column, rows in order (four chains)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dZ, int M, int N, float* __restrict__ db)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int m = 0;
    for (; m + 3 < M; m += 4) {
        s0 += dZ[(size_t)m * N + n];
        s1 += dZ[(size_t)(m + 1) * N + n];
        s2 += dZ[(size_t)(m + 2) * N + n];
        s3 += dZ[(size_t)(m + 3) * N + n];
    }
    for (; m < M; ++m) s0 += dZ[(size_t)m * N + n];
    db[n] = (s0 + s1) + (s2 + s3);
}

// partial[blockIdx.x] = sum of x[i]^2 over this block's grid-stride slice (fixed order: per-thread chain, then LDS tree)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, size_t n, float* __restrict__ partial)
{
    __shared__ float red[256];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += x[i] * x[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// loss_out[0] = sum(row_loss[0..M)) + reg_scale * sum(partials[0..P))   (one workgroup, fixed order)
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ row_loss, int M, const float* __restrict__ partials,
                                                            int P, float reg_scale, float* __restrict__ loss_out)
{
    __shared__ float red[256];
    float s = 0.f, q = 0.f;
    for (int i = threadIdx.x; i < M; i += 256) s += row_loss[i];
    for (int i = threadIdx.x; i < P; i += 256) q += partials[i];
    red[threadIdx.x] = s + reg_scale * q;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_out[0] = red[0];
}

// torch.optim.Adam (single-tensor path), in its order of operations:
//   exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
//   denom = exp_avg_sq.sqrt() / sqrt(1 - beta2^t) + eps;  param.addcdiv_(exp_avg, denom, value=-(lr / (1 - beta1^t)))
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float om_beta1, float beta2, float om_beta2, float eps,
                                                   float step_size, float bc2_sqrt)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    // om_beta = (float)(1.0 - (double)beta): torch forms 1 - beta in double and casts the scalar, which is not 1.0f - (float)beta
    const float mi = m[i] + om_beta1 * (gi - m[i]);
    const float vi = v[i] * beta2 + om_beta2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
}

// Xo[i][:] = Xn[order[i]][:], yo[i] = yn[order[i]]: the visiting order of a pass applied on the device (one float4 per thread)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ Xn, const int32_t* __restrict__ yn,
                                                          const int64_t* __restrict__ order, int64_t n, int d4, float* __restrict__ Xo,
                                                          int32_t* __restrict__ yo)
{
    const int64_t i = blockIdx.x;
    const int64_t src = order[i];
    const float4* s = reinterpret_cast<const float4*>(Xn + (size_t)src * d4 * 4);
    float4* d = reinterpret_cast<float4*>(Xo + (size_t)i * d4 * 4);
    for (int k = threadIdx.x; k < d4; k += 256) d[k] = s[k];
    if (threadIdx.x == 0) yo[i] = yn[src];
    (void)n;
}

template <bool TA, bool TB>
int launch_tgemm(const float* A, const float* B, float* C, int M, int N, int K, int epi, const float* aux, float scale, hipStream_t st)
{
    dim3 grid((M + 63) / 64, (N + 63) / 64);
#define GO(E) hipLaunchKernelGGL((tgemm_f32_kernel<TA, TB, E>), grid, dim3(256), 0, st, A, B, C, M, N, K, aux, scale)
    switch (epi) {
        case TEPI_NONE: GO(TEPI_NONE); break;
        case TEPI_BIAS_RELU: GO(TEPI_BIAS_RELU); break;
        case TEPI_BIAS: GO(TEPI_BIAS); break;
        case TEPI_MASK: GO(TEPI_MASK); break;
        case TEPI_L2: GO(TEPI_L2); break;
        default: return -1;
    }
#undef GO
    return (int)hipGetLastError();
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
struct mmc_trainer {
    int device = 0, L = 0, K = 0;
    std::vector<int> dims;
    std::vector<float*> W, b, gW, gb, mW, vW, mb, vb;   // device
    std::vector<float*> H;                              // H[0] = mini-batch input, H[l+1] = layer l's output; size L+1
    std::vector<float*> dZ;                             // dZ[l] = gradient w.r.t. layer l-1's pre-activation (size L+1, [0] unused)
    float* cw = nullptr;                                // class weights or null
    float *X = nullptr, *row_loss = nullptr, *partials = nullptr, *losses = nullptr;
    int32_t* y = nullptr;
    float* Xn = nullptr;            // natural-order staging of a pass (device-side shuffle)
    int32_t* yn = nullptr;
    int64_t* order = nullptr;
    int64_t cap_n = 0, cap_nn = 0;
    int cap_mb = 0, cap_steps = 0;
    double lr = 1e-3, beta1 = 0.9, beta2 = 0.999, eps = 1e-8, alpha = 1e-4;   // kept in double: torch derives its fp32 scalars from python floats
    long long t = 0;                                    // Adam step count
    std::vector<float> cw_host;
};

#define T_TRY(expr)                                                                                   \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return mmc_fail(MMC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)
#define T_K(expr)                                                                   \
    do {                                                                            \
        int r_ = (expr);                                                            \
        if (r_ != 0) return mmc_fail(MMC_ERR_HIP, "%s failed (%d)", #expr, r_);     \
    } while (0)

static void free_all(std::vector<float*>& v)
{
    for (float* p : v) hipFree(p);
    v.clear();
}

extern "C" void mmc_trainer_destroy(mmc_trainer* t)
{
    if (!t) return;
    hipSetDevice(t->device);
    free_all(t->W); free_all(t->b); free_all(t->gW); free_all(t->gb); free_all(t->mW); free_all(t->vW); free_all(t->mb); free_all(t->vb);
    for (size_t l = 1; l < t->H.size(); ++l) hipFree(t->H[l]);
    for (size_t l = 1; l < t->dZ.size(); ++l) hipFree(t->dZ[l]);
    hipFree(t->cw); hipFree(t->X); hipFree(t->row_loss); hipFree(t->partials); hipFree(t->losses); hipFree(t->y);
    hipFree(t->Xn); hipFree(t->yn); hipFree(t->order);
    delete t;
}

extern "C" int mmc_trainer_create(const float* const* W, const float* const* b, const int* dims, int n_layers, double lr, double beta1,
                                  double beta2, double eps, double alpha, const float* class_weight, int device, mmc_trainer** out)
{
    if (!out) return mmc_fail(MMC_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!W || !b || !dims) return mmc_fail(MMC_ERR_ARG, "NULL argument");
    if (n_layers < 1 || n_layers > 16) return mmc_fail(MMC_ERR_ARG, "n_layers %d out of range [1,16]", n_layers);
    for (int l = 0; l <= n_layers; ++l)
        if (dims[l] < 1) return mmc_fail(MMC_ERR_ARG, "dims[%d]=%d must be positive", l, dims[l]);
    if (!(lr > 0.) || !(beta1 >= 0. && beta1 < 1.) || !(beta2 >= 0. && beta2 < 1.) || !(eps >= 0.) || !(alpha >= 0.))
        return mmc_fail(MMC_ERR_ARG, "bad optimizer hyper-parameter (lr %g, betas %g %g, eps %g, alpha %g)", lr, beta1, beta2, eps, alpha);
    const int K = dims[n_layers];
    if (class_weight)
        for (int c = 0; c < K; ++c)
            if (!(class_weight[c] >= 0.f)) return mmc_fail(MMC_ERR_ARG, "class_weight[%d] = %g is negative", c, class_weight[c]);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return mmc_fail(MMC_ERR_HIP, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return mmc_fail(MMC_ERR_ARG, "device %d out of range (%d visible)", device, ndev);
    T_TRY(hipSetDevice(device));
    mmc_trainer* t = new mmc_trainer();
    t->device = device; t->L = n_layers; t->K = K;
    t->dims.assign(dims, dims + n_layers + 1);
    t->lr = lr; t->beta1 = beta1; t->beta2 = beta2; t->eps = eps; t->alpha = alpha;
    auto up = [&](std::vector<float*>& dst, const float* src, size_t n, bool zero) -> bool {
        float* d = nullptr;
        if (hipMalloc((void**)&d, n * 4 + 256) != hipSuccess) return false;
        dst.push_back(d);
        return (zero ? hipMemset(d, 0, n * 4) : hipMemcpy(d, src, n * 4, hipMemcpyHostToDevice)) == hipSuccess;
    };
    bool ok = true;
    for (int l = 0; l < n_layers && ok; ++l) {
        const size_t nw = (size_t)dims[l + 1] * dims[l], nb = (size_t)dims[l + 1];
        ok = up(t->W, W[l], nw, false) && up(t->b, b[l], nb, false) && up(t->gW, nullptr, nw, true) && up(t->gb, nullptr, nb, true) &&
             up(t->mW, nullptr, nw, true) && up(t->vW, nullptr, nw, true) && up(t->mb, nullptr, nb, true) && up(t->vb, nullptr, nb, true);
    }
    if (ok && class_weight) {
        ok = hipMalloc((void**)&t->cw, (size_t)K * 4 + 256) == hipSuccess &&
             hipMemcpy(t->cw, class_weight, (size_t)K * 4, hipMemcpyHostToDevice) == hipSuccess;
        t->cw_host.assign(class_weight, class_weight + K);
    }
    if (ok) ok = hipMalloc((void**)&t->partials, 256 * 4 * 16 + 256) == hipSuccess;
    if (!ok) {
        mmc_trainer_destroy(t);
        return mmc_fail(MMC_ERR_NOMEM, "hipMalloc/hipMemcpy failed while creating the trainer");
    }
    t->H.assign(n_layers + 1, nullptr);
    t->dZ.assign(n_layers + 1, nullptr);
    *out = t;
    return MMC_OK;
}

static int trainer_reserve(mmc_trainer* t, int64_t n, int mb, int steps)
{
    if (n > t->cap_n) {
        hipFree(t->X); hipFree(t->y);
        t->X = nullptr; t->y = nullptr; t->cap_n = 0;
        T_TRY(hipMalloc((void**)&t->X, (size_t)n * t->dims[0] * 4 + 256));
        T_TRY(hipMalloc((void**)&t->y, (size_t)n * 4 + 256));
        t->cap_n = n;
    }
    if (mb > t->cap_mb) {
        for (int l = 1; l <= t->L; ++l) { hipFree(t->H[l]); hipFree(t->dZ[l]); t->H[l] = t->dZ[l] = nullptr; }
        hipFree(t->row_loss); t->row_loss = nullptr; t->cap_mb = 0;
        for (int l = 1; l <= t->L; ++l) {
            T_TRY(hipMalloc((void**)&t->H[l], (size_t)mb * t->dims[l] * 4 + 256));
            T_TRY(hipMalloc((void**)&t->dZ[l], (size_t)mb * t->dims[l] * 4 + 256));
        }
        T_TRY(hipMalloc((void**)&t->row_loss, (size_t)mb * 4 + 256));
        t->cap_mb = mb;
    }
    if (steps > t->cap_steps) {
        hipFree(t->losses); t->losses = nullptr; t->cap_steps = 0;
        T_TRY(hipMalloc((void**)&t->losses, (size_t)steps * 4 + 256));
        t->cap_steps = steps;
    }
    return 0;
}

// one optimizer step on rows [start, start+mb) of the resident X / y
static int trainer_step(mmc_trainer* t, int64_t start, int mb, float inv_wsum, float* loss_slot, hipStream_t st)
{
    const int L = t->L;
    t->H[0] = t->X + (size_t)start * t->dims[0];
    for (int l = 0; l < L; ++l)
        T_K((launch_tgemm<false, true>(t->H[l], t->W[l], t->H[l + 1], mb, t->dims[l + 1], t->dims[l], l == L - 1 ? TEPI_BIAS : TEPI_BIAS_RELU,
                                      t->b[l], 0.f, st)));
    hipLaunchKernelGGL(ce_grad_kernel, dim3((mb + 3) / 4), dim3(256), 0, st, t->H[L], t->y + start, mb, t->K, t->cw, inv_wsum, t->dZ[L],
                       t->row_loss);
    // regularised loss of this mini-batch (before the update): data + (0.5 alpha / mb) sum W^2
    for (int l = 0; l < L; ++l)
        hipLaunchKernelGGL(sumsq_kernel, dim3(256), dim3(256), 0, st, t->W[l], (size_t)t->dims[l + 1] * t->dims[l], t->partials + 256 * l);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, t->row_loss, mb, t->partials, 256 * L, (float)(0.5 * t->alpha / (double)mb),
                       loss_slot);
    for (int l = L - 1; l >= 0; --l) {
        const int no = t->dims[l + 1], ni = t->dims[l];
        T_K((launch_tgemm<true, false>(t->dZ[l + 1], t->H[l], t->gW[l], no, ni, mb, TEPI_L2, t->W[l], (float)(t->alpha / (double)mb), st)));
        hipLaunchKernelGGL(colsum_kernel, dim3((no + 255) / 256), dim3(256), 0, st, t->dZ[l + 1], mb, no, t->gb[l]);
        if (l > 0) T_K((launch_tgemm<false, false>(t->dZ[l + 1], t->W[l], t->dZ[l], mb, ni, no, TEPI_MASK, t->H[l], 0.f, st)));
    }
    ++t->t;
    const double bc1 = 1.0 - std::pow(t->beta1, (double)t->t), bc2 = 1.0 - std::pow(t->beta2, (double)t->t);
    const float step_size = (float)(t->lr / bc1), bc2_sqrt = (float)std::sqrt(bc2);
    const float om1 = (float)(1.0 - t->beta1), om2 = (float)(1.0 - t->beta2), b2f = (float)t->beta2, epsf = (float)t->eps;
    for (int l = 0; l < L; ++l) {
        const size_t nw = (size_t)t->dims[l + 1] * t->dims[l], nb = (size_t)t->dims[l + 1];
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, t->W[l], t->gW[l], t->mW[l], t->vW[l], nw,
                           om1, b2f, om2, epsf, step_size, bc2_sqrt);
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, t->b[l], t->gb[l], t->mb[l], t->vb[l], nb,
                           om1, b2f, om2, epsf, step_size, bc2_sqrt);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mmc_fail(MMC_ERR_HIP, "trainer step launch failed: %s", hipGetErrorString(e));
    return 0;
}

// `order` (n int64 row indices, or NULL = rows are already in visiting order): the shuffle of torch_classifier.py:251-257 applied
// on the device -- the natural-order matrix is uploaded as it is and a gather kernel builds the visiting order (when the
// feature width is a multiple of 4; otherwise, and without `order`, the rows are taken as given).
extern "C" int mmc_trainer_partial_fit_ordered(mmc_trainer* t, const float* X, const int32_t* y, const int64_t* order, int64_t n,
                                               int batch_size, double* avg_loss, void* hip_stream)
{
    if (!t) return mmc_fail(MMC_ERR_ARG, "trainer handle is NULL");
    if (!X || !y) return mmc_fail(MMC_ERR_ARG, "X/y is NULL");
    if (n < 1) return mmc_fail(MMC_ERR_ARG, "n = %lld must be positive", (long long)n);
    if (batch_size < 1) return mmc_fail(MMC_ERR_ARG, "batch_size = %d must be positive", batch_size);
    if (order && (t->dims[0] & 3)) return mmc_fail(MMC_ERR_ARG, "device-side ordering needs a feature width that is a multiple of 4 (got %d)", t->dims[0]);
    for (int64_t i = 0; i < n; ++i) {
        if (y[i] < 0 || y[i] >= t->K) return mmc_fail(MMC_ERR_ARG, "label index y[%lld] = %d outside [0, %d)", (long long)i, y[i], t->K);
        if (order && (order[i] < 0 || order[i] >= n)) return mmc_fail(MMC_ERR_ARG, "order[%lld] = %lld outside [0, %lld)", (long long)i, (long long)order[i], (long long)n);
    }
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    T_TRY(hipSetDevice(t->device));
    const int mb = (int)(batch_size < n ? batch_size : n);
    const int steps = (int)((n + mb - 1) / mb);
    // every mini-batch's weight sum (mean reduction of the weighted CE) is checked BEFORE anything is uploaded or launched, so
    // a rejected pass leaves the weights, the Adam moments and the step counter exactly as they were
    std::vector<double> wsums(steps);
    for (int s = 0; s < steps; ++s) {
        const int64_t start = (int64_t)s * mb;
        const int cur = (int)((n - start) < mb ? (n - start) : mb);
        double wsum = 0.0;
        if (t->cw) { for (int i = 0; i < cur; ++i) wsum += t->cw_host[y[order ? order[start + i] : start + i]]; } else wsum = cur;
        if (!(wsum > 0.0)) return mmc_fail(MMC_ERR_ARG, "mini-batch %d has zero total class weight", s);
        wsums[s] = wsum;
    }
    int r = trainer_reserve(t, n, mb, steps);
    if (r) return r;
    if (order) {
        if (n > t->cap_nn) {
            hipFree(t->Xn); hipFree(t->yn); hipFree(t->order);
            t->Xn = nullptr; t->yn = nullptr; t->order = nullptr; t->cap_nn = 0;
            T_TRY(hipMalloc((void**)&t->Xn, (size_t)n * t->dims[0] * 4 + 256));
            T_TRY(hipMalloc((void**)&t->yn, (size_t)n * 4 + 256));
            T_TRY(hipMalloc((void**)&t->order, (size_t)n * 8 + 256));
            t->cap_nn = n;
        }
        T_TRY(hipMemcpyAsync(t->Xn, X, (size_t)n * t->dims[0] * 4, hipMemcpyHostToDevice, st));
        T_TRY(hipMemcpyAsync(t->yn, y, (size_t)n * 4, hipMemcpyHostToDevice, st));
        T_TRY(hipMemcpyAsync(t->order, order, (size_t)n * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n), dim3(256), 0, st, t->Xn, t->yn, t->order, n, t->dims[0] / 4, t->X, t->y);
    } else {
        T_TRY(hipMemcpyAsync(t->X, X, (size_t)n * t->dims[0] * 4, hipMemcpyHostToDevice, st));
        T_TRY(hipMemcpyAsync(t->y, y, (size_t)n * 4, hipMemcpyHostToDevice, st));
    }
    std::vector<int> sizes(steps);
    for (int s = 0; s < steps; ++s) {
        const int64_t start = (int64_t)s * mb;
        const int cur = (int)((n - start) < mb ? (n - start) : mb);
        sizes[s] = cur;
        r = trainer_step(t, start, cur, (float)(1.0 / wsums[s]), t->losses + s, st);
        if (r) return r;
    }
    std::vector<float> h(steps);
    T_TRY(hipMemcpyAsync(h.data(), t->losses, (size_t)steps * 4, hipMemcpyDeviceToHost, st));
    T_TRY(hipStreamSynchronize(st));
    double tot = 0.0;
    for (int s = 0; s < steps; ++s) tot += (double)h[s] * sizes[s];   // torch_classifier.py:293-298: loss.item() * mb_size
    if (avg_loss) *avg_loss = tot / (double)n;
    return MMC_OK;
}

extern "C" int mmc_trainer_partial_fit(mmc_trainer* t, const float* X, const int32_t* y, int64_t n, int batch_size, double* avg_loss,
                                       void* hip_stream)
{
    return mmc_trainer_partial_fit_ordered(t, X, y, nullptr, n, batch_size, avg_loss, hip_stream);
}

extern "C" int mmc_trainer_get_params(mmc_trainer* t, float* const* W, float* const* b)
{
    if (!t || !W || !b) return mmc_fail(MMC_ERR_ARG, "NULL argument");
    T_TRY(hipSetDevice(t->device));
    T_TRY(hipDeviceSynchronize());
    for (int l = 0; l < t->L; ++l) {
        T_TRY(hipMemcpy(W[l], t->W[l], (size_t)t->dims[l + 1] * t->dims[l] * 4, hipMemcpyDeviceToHost));
        T_TRY(hipMemcpy(b[l], t->b[l], (size_t)t->dims[l + 1] * 4, hipMemcpyDeviceToHost));
    }
    return MMC_OK;
}

// Optimizer state for pickling (torch_classifier.py:404-415 serialises module + optimizer state dicts):
// which = 0: exp_avg, 1: exp_avg_sq.  `set` != 0 uploads instead.  *step is read / written alike.
extern "C" int mmc_trainer_adam_state(mmc_trainer* t, int which, int set, float* const* W, float* const* b, long long* step)
{
    if (!t || !W || !b || !step) return mmc_fail(MMC_ERR_ARG, "NULL argument");
    if (which != 0 && which != 1) return mmc_fail(MMC_ERR_ARG, "which must be 0 (exp_avg) or 1 (exp_avg_sq)");
    T_TRY(hipSetDevice(t->device));
    T_TRY(hipDeviceSynchronize());
    for (int l = 0; l < t->L; ++l) {
        float* dw = which == 0 ? t->mW[l] : t->vW[l];
        float* db = which == 0 ? t->mb[l] : t->vb[l];
        const size_t nw = (size_t)t->dims[l + 1] * t->dims[l] * 4, nb = (size_t)t->dims[l + 1] * 4;
        if (set) { T_TRY(hipMemcpy(dw, W[l], nw, hipMemcpyHostToDevice)); T_TRY(hipMemcpy(db, b[l], nb, hipMemcpyHostToDevice)); }
        else { T_TRY(hipMemcpy(W[l], dw, nw, hipMemcpyDeviceToHost)); T_TRY(hipMemcpy(b[l], db, nb, hipMemcpyDeviceToHost)); }
    }
    if (set) t->t = *step; else *step = t->t;
    return MMC_OK;
}

// logits of the current parameters (eval mode): X n x dims[0] host -> logits n x K host
extern "C" int mmc_trainer_logits(mmc_trainer* t, const float* X, int64_t n, float* logits, void* hip_stream)
{
    if (!t) return mmc_fail(MMC_ERR_ARG, "trainer handle is NULL");
    if (n < 0) return mmc_fail(MMC_ERR_ARG, "n = %lld is negative", (long long)n);
    if (n == 0) return MMC_OK;
    if (!X || !logits) return mmc_fail(MMC_ERR_ARG, "X/logits is NULL");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    T_TRY(hipSetDevice(t->device));
    const int64_t chunk = 16384;
    for (int64_t off = 0; off < n; off += chunk) {
        const int cur = (int)((n - off) < chunk ? (n - off) : chunk);
        int r = trainer_reserve(t, cur, cur, 1);
        if (r) return r;
        T_TRY(hipMemcpyAsync(t->X, X + (size_t)off * t->dims[0], (size_t)cur * t->dims[0] * 4, hipMemcpyHostToDevice, st));
        t->H[0] = t->X;
        for (int l = 0; l < t->L; ++l)
            T_K((launch_tgemm<false, true>(t->H[l], t->W[l], t->H[l + 1], cur, t->dims[l + 1], t->dims[l],
                                          l == t->L - 1 ? TEPI_BIAS : TEPI_BIAS_RELU, t->b[l], 0.f, st)));
        T_TRY(hipMemcpyAsync(logits + (size_t)off * t->K, t->H[t->L], (size_t)cur * t->K * 4, hipMemcpyDeviceToHost, st));
        T_TRY(hipStreamSynchronize(st));
    }
    return MMC_OK;
}
