/*
 * mmc.h -- C ABI of libmermaid_mi355.so: the MI355X (gfx950) implementation of the
 * PySpacer EfficientNet patch feature-extraction path of data-mermaid/mermaid-classifier.
 *
 * The reference has no FFI; its "plugin API" for this path is a Python class contract
 * plus three file artifacts.  Every entry point below names the reference
 * interface it replaces.  Plain C: no exceptions cross the boundary, every call returns an
 * int status (0 = MMC_OK) and mmc_last_error() returns a thread-local message.
 * All buffers are caller-owned; handles are opaque; streams are explicit (hipStream_t
 * passed as void*, NULL = the default stream).  No torch types appear here.
 */
#ifndef MMC_H
#define MMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMC_OK 0
#define MMC_ERR_ARG 1      /* bad argument / shape (ValueError on the Python side) */
#define MMC_ERR_WEIGHTS 2  /* packed weights blob malformed (KeyError/ValueError) */
#define MMC_ERR_HIP 3      /* HIP runtime failure (RuntimeError) */
#define MMC_ERR_NOMEM 4

#define MMC_ARCH_B0 0   /* efficientnet-b0: the network pyspacer's EfficientNetExtractor builds (the reference path) */
#define MMC_ARCH_B4 1   /* efficientnet-b4 (width 1.4, depth 1.8) on 224x224 patches: BASELINE.json configs[4]; not in the
                         * reference; fused expand+depthwise kernels plus the shape-generic squeeze-excite / project / head kernels, feature_dim 1792 */

/* memory-kind flags for mmc_backbone_extract / mmc_head_predict / mmc_crop_patches */
#define MMC_IN_DEVICE 0u
#define MMC_IN_HOST 1u   /* `patches`/`feats`/`image` is host memory: staged with hipMemcpyAsync */
#define MMC_OUT_DEVICE 0u
#define MMC_OUT_HOST 2u  /* outputs are host memory; the call synchronises the stream before returning */

#define MMC_FEATURE_DIM_B0 1280
#define MMC_FEATURE_DIM_B4 1792
#define MMC_PATCH 224

typedef struct mmc_backbone mmc_backbone;
typedef struct mmc_head mmc_head;

/* ---- library ------------------------------------------------------------------------ */
const char* mmc_last_error(void);
int mmc_version(void);          /* ABI version, currently 1 */
int mmc_device_count(void);     /* number of visible HIP devices (0 when none) */

/* ---- backbone ------------------------------------------------------------------------
 * Replaces: EfficientNetExtractor.load_weights(stream) + net.to(device).eval()
 *   (reference scripts/build_feature_bucket.py:402-413) and
 *   net.extract_features(batch) (scripts/build_feature_bucket.py:430-437), including the
 *   per-patch transformation() (ToTensor + Normalize, :420-423), which is folded into the
 *   stem kernel: the library takes raw u8 HWC patches.
 *
 * `packed` is the blob produced by mermaid_classifier_amd.weights.pack_backbone():
 *   header  { char magic[4]="MMCW"; u32 version=1; u32 arch; u32 n_tensors; }
 *   table   n_tensors x { u64 offset; u64 nbytes; }          (offsets from blob start, 256-B aligned)
 *   tensors in the fixed order documented in mermaid_classifier_amd/weights.py
 *   (BN folded: fp16 GEMM weights in MFMA-row-permuted [Np][Kp] layout, fp32 biases,
 *    fp32 depthwise taps [k*k][C], fp32 squeeze-excite matrices).
 * The blob is copied to device memory; the caller may free it after the call.
 * `max_batch` bounds the activation workspace (patches processed per internal pass).
 */
int mmc_backbone_create(const void* packed, size_t nbytes, int arch, int device, int max_batch,
                        mmc_backbone** out);
/* The same with `flags`.  MMC_PRECISION_FP8 (MMC_ARCH_B4 only; BASELINE.json configs[4], not in the reference): the squeeze-excite
 * gated project convs of the 7x7 stage (env MMC_FP8_MAXH=14: of the 14x14 stages too) run on OCP e4m3 operands
 * (v_mfma_scale_f32_16x16x128_f8f6f4, fp32 accumulation): weights quantised at create time with one scale per output channel,
 * activations in the kernel with one scale per pixel.  Everything else stays fp16 storage / fp32 accumulation.  Accuracy is that of
 * the operand format, not the fp16 gates: feature cosine >= 0.997 against the fp32 oracle on image-like patches (tests). */
#define MMC_PRECISION_FP8 1u
int mmc_backbone_create_ex(const void* packed, size_t nbytes, int arch, int device, int max_batch, unsigned flags,
                           mmc_backbone** out);
/* fp32 -> OCP e4m3fn bytes (round to nearest even, saturating at +-448): the encoding the fp8 weights are stored in.  Host code,
 * no device needed; exported so that tests can check it against an independent implementation. */
int mmc_fp8_e4m3_encode(const float* in, uint8_t* out, size_t n);
void mmc_backbone_destroy(mmc_backbone* bb);
int mmc_feature_dim(const mmc_backbone* bb);            /* 1280 for B0, 1792 for B4 */
int mmc_backbone_max_batch(const mmc_backbone* bb);
/* A pass is split into this many sub-batches that run concurrently on internal HIP streams (forked from and
 * joined to `hip_stream`); env MMC_LANES overrides the default of 2.  Results do not depend on it. */
int mmc_backbone_lanes(const mmc_backbone* bb);
size_t mmc_backbone_workspace_bytes(const mmc_backbone* bb);

/* patches: n x 224 x 224 x 3 u8 (HWC, RGB).  out_features: n x feature_dim fp32, row i = patch i.
 * flags: MMC_IN_HOST | MMC_OUT_HOST select host pointers; default both device pointers.
 * Asynchronous on `hip_stream` unless MMC_OUT_HOST is set.  One pass at a time per handle (the reference's forward path
 * is single-threaded: scripts/build_feature_bucket.py, "one extractor instance per process"): concurrent calls on one
 * handle are serialised by a per-handle mutex, and a call that comes in on a different stream than the previous one first
 * waits (on the device) for that call's work, because the workspace is shared.  For concurrency create one handle per
 * stream.
 * When the same (patches, out_features, n) combination comes in repeatedly the pass is captured into a HIP graph
 * once and replayed on `hip_stream` afterwards (env MMC_GRAPH=0 disables); results are identical either way.  The 32
 * most recently used combinations keep their graph.  A combination whose graph was evicted runs as plain launches from
 * then on (it is not captured a second time), and after 32 evictions -- one full turnover of the cache -- no further
 * combination is captured: a caller that cycles through more buffers than the cache holds pays plain launches, never a
 * capture per call. */
int mmc_backbone_extract(mmc_backbone* bb, const void* patches, int64_t n, float* out_features,
                         unsigned flags, void* hip_stream);

/* Debug hook for the graph cache above: stats[0] = graphs captured so far, stats[1] = graphs evicted so far,
 * stats[2] = graphs cached now.  (No reference counterpart: the reference has no launch graphs.) */
int mmc_backbone_graph_stats(mmc_backbone* bb, int64_t stats[3]);

/* Debug/parity hook: copy one intermediate activation of the LAST internal pass to host.
 * name: "stem", "b<i>.expand", "b<i>.dw", "b<i>.gate", "b<i>.out".  fp16 NHWC tensors are returned
 * as fp32 NHWC; gates as fp32 (n,C).  `capacity` = number of floats available in `out`;
 * *n_written receives the element count.  Requires MMC_KEEP_ACTIVATIONS=1 at create time
 * (separate buffers per layer); used only by the parity tests. */
int mmc_backbone_read_activation(mmc_backbone* bb, const char* name, float* out, size_t capacity,
                                 size_t* n_written);

/* Kernel timing hook for bench.py: runs `iters` passes over `n` resident patches and returns the
 * HIP-event elapsed milliseconds of every launch ("<layer>|<kernel instantiation>"), measured on `hip_stream`.
 * Sub-batches (see mmc_backbone_lanes) are run one after the other here, so each layer appears once per lane
 * with ceil(n / lanes) patches.
 * names/ms arrays have `cap` slots; *n_out receives the number filled. */
int mmc_backbone_profile(mmc_backbone* bb, const void* patches_dev, int64_t n, float* out_features_dev,
                         void* hip_stream, char (*names)[64], float* ms, int* launches, int cap, int* n_out);

/* ---- GPU crop front-end ---------------------------------------------------------------
 * Replaces: pyspacer crop_patches(image, rowcols, 224) as called by FeatureExtractor.__call__
 *   (reached from scripts/build_feature_bucket.py:775 and
 *    mermaid_classifier/pyspacer/annotation.py:241): reflect-pad by 224, slice 224x224 around
 *   each (row,col).  Implemented as index arithmetic on the resident image (no padded copy).
 * image: H x W x 3 u8; rowcols: n x 2 int32 (row, col); patches_out: n x 224 x 224 x 3 u8 (device).
 * Points must lie inside the image: host points (MMC_IN_HOST) outside it are rejected with MMC_ERR_ARG; device-resident
 * points cannot be inspected by the host and are clamped into the image by the kernel (never an out-of-bounds read).
 * With MMC_IN_HOST and few points on a big image (n * 150528 * 6 <= image bytes -- the reference's data: 10-25 points on a
 * 27 MP image) the patches are cut on the host (same index arithmetic, up to 4 threads, into a pinned ring slot) and only
 * they are uploaded; otherwise the image is uploaded and crop_kernel cuts them.  Same bytes either way.  The host image is
 * fully consumed before the call returns; the upload is asynchronous on `hip_stream`.  Env MMC_CROP_HOST=0/1 forces a path. */
int mmc_crop_patches(const void* image, int height, int width, const int32_t* rowcols, int64_t n,
                     void* patches_out_dev, unsigned flags, int device, void* hip_stream);

/* ---- calibrated MLP head -------------------------------------------------------------
 * Replaces: CalibratedHead.forward (mermaid_classifier/pyspacer/inference/head.py:66-89) as run by
 *   Predictor.predict_proba (mermaid_classifier/pyspacer/inference/loader.py:30-35).
 * W[l]: (dims[l+1], dims[l]) row-major fp32 (torch nn.Linear layout); b[l]: (dims[l+1]);
 * dims has n_layers+1 entries, dims[n_layers] == K; a, bcal: (K) Platt parameters.  All host pointers.
 */
int mmc_head_create(const float* const* W, const float* const* b, const int* dims, int n_layers,
                    const float* a, const float* bcal, int K, int device, mmc_head** out);
void mmc_head_destroy(mmc_head* h);
int mmc_head_input_dim(const mmc_head* h);
int mmc_head_num_classes(const mmc_head* h);
/* feats: n x input_dim fp32; proba: n x K fp32; argmax: n int32 (may be NULL). */
int mmc_head_predict(mmc_head* h, const float* feats, int64_t n, float* proba, int32_t* argmax,
                     unsigned flags, void* hip_stream);

/* ---- MLP classifier training on precomputed feature vectors --------------------------------
 * Replaces: the arithmetic of TorchMLPClassifier.partial_fit (mermaid_classifier/pyspacer/torch_classifier.py:226-303)
 *   as driven by the trainer's batch loop (mermaid_classifier/pyspacer/trainer.py:138-145): per mini-batch
 *   logits -> F.cross_entropy(weight=class_weight) + (0.5*alpha/mb)*sum(W^2) -> backward -> torch.optim.Adam.step().
 * The host side keeps what the reference keeps on the host: classes_/label lookup, Glorot initialisation (torch RNG, so
 * the same random_state gives the same initial weights), the shuffle order, loss_curve_/n_iter_ bookkeeping.
 * W[l]: (dims[l+1], dims[l]) row-major fp32, b[l]: (dims[l+1]) -- the initial parameters (host pointers);
 * class_weight: K floats in classes_ order or NULL; Adam moments start at zero, step count at 0.
 * Hyper-parameters are doubles because torch derives its fp32 scalars (1 - beta, lr / bias_correction, alpha / mb) from
 * python floats: (float)(1.0 - 0.9) is not 1.0f - 0.9f. */
typedef struct mmc_trainer mmc_trainer;
int mmc_trainer_create(const float* const* W, const float* const* b, const int* dims, int n_layers, double lr, double beta1,
                       double beta2, double eps, double alpha, const float* class_weight, int device, mmc_trainer** out);
void mmc_trainer_destroy(mmc_trainer* t);
/* One pass over n samples ALREADY IN VISITING ORDER (the caller applies the shuffle, torch_classifier.py:251-257):
 * X n x dims[0] fp32 and y n int32 class indices, host pointers; mini-batches of `batch_size` rows (last one ragged),
 * one Adam step each.  *avg_loss = sum(loss_i * mb_i) / n, the value the reference appends to loss_curve_ (:293-300).
 * Synchronises `hip_stream` before returning. */
int mmc_trainer_partial_fit(mmc_trainer* t, const float* X, const int32_t* y, int64_t n, int batch_size, double* avg_loss,
                            void* hip_stream);
/* The same with the visiting order applied on the device: X / y in their natural order plus `order` (n int64 row indices:
 * position i of the pass visits row order[i]); NULL order = mmc_trainer_partial_fit.  Needs dims[0] % 4 == 0. */
int mmc_trainer_partial_fit_ordered(mmc_trainer* t, const float* X, const int32_t* y, const int64_t* order, int64_t n,
                                    int batch_size, double* avg_loss, void* hip_stream);
/* Current parameters to host buffers shaped like the create-time ones. */
int mmc_trainer_get_params(mmc_trainer* t, float* const* W, float* const* b);
/* Adam moments (which = 0: exp_avg, 1: exp_avg_sq) and step count, read (set = 0) or written (set = 1): what the
 * reference pickles as the optimizer state dict (torch_classifier.py:404-415). */
int mmc_trainer_adam_state(mmc_trainer* t, int which, int set, float* const* W, float* const* b, long long* step);
/* Raw logits of the current parameters: X n x dims[0] host -> logits n x K host (the softmax / float64 renormalisation of
 * _forward_probs, torch_classifier.py:332-376, stays on the host). */
int mmc_trainer_logits(mmc_trainer* t, const float* X, int64_t n, float* logits, void* hip_stream);

/* ---- multi-GPU: the gather of the sharded path --------------------------------------------------------------------
 * The path shards by patches (contiguous blocks of the row range per rank, weights replicated, no exchange during compute);
 * its one exchange step is the all-gather of the ranks' (n_r, 1280) feature blocks.  Replaces: nothing in the reference's
 * code -- it scales out as one job per source id (scripts/launch_processing.py:59-66, 199-233) and the matrices meet on S3;
 * SURVEY.md section 8(b) sketches this entry.  The Python package does the same through torch.distributed
 * (mermaid_classifier_amd/dist.py: FeatureGatherer); these entries give a non-Python host the same step through this
 * header alone.  One process per GPU; librccl is resolved at the first call (dlopen: the copy the process already has,
 * else ROCm's; MMC_RCCL_LIBRARY overrides), so single-GPU users never load it.
 *   mmc_dist_unique_id  rank 0 makes the 128-byte id and hands it to the other ranks out of band (file, env, TCP, MPI)
 *   mmc_dist_create     collective: every rank calls it with the same id, its rank, the world size and its device
 *   mmc_gather_features collective, asynchronous on `hip_stream`: local = this rank's n_local x dim fp32 block (device),
 *                       all = sum(counts) x dim (device) on EVERY rank, rank r's block at row sum(counts[0..r));
 *                       counts = host array of `world` block heights, or NULL when every rank holds n_local rows. */
#define MMC_DIST_ID_BYTES 128
typedef struct mmc_dist mmc_dist;
int mmc_dist_unique_id(unsigned char id[MMC_DIST_ID_BYTES]);
int mmc_dist_create(const unsigned char id[MMC_DIST_ID_BYTES], int rank, int world, int device, mmc_dist** out);
void mmc_dist_destroy(mmc_dist* d);
int mmc_gather_features(mmc_dist* d, const float* local, int64_t n_local, int dim, const int64_t* counts, float* all,
                        void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* MMC_H */
