/* c_host.c -- the extraction path from a plain C program: nothing but include/mmc.h and libmermaid_mi355.so (no Python, no torch,
 * no HIP calls in the host: MMC_IN_HOST | MMC_OUT_HOST hands the library host buffers).  This is the shape of the binding a
 * non-Python host (cgo, JNI, N-API ...) would wrap around the reference's patches_to_features
 * (scripts/build_feature_bucket.py:415-446): packed weights in, u8 patches in, (n, 1280) fp32 features out.
 *
 *   gcc -std=c99 -O2 -I include examples/c_host.c -o c_host -L mermaid_classifier_amd -lmermaid_mi355 -Wl,-rpath,$PWD/mermaid_classifier_amd
 *   ./c_host weights.mmcw patches.u8 n features.f32      (weights.mmcw: mermaid_classifier_amd.weights.pack_backbone(...))
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "mmc.h"

static void* slurp(const char* path, size_t* nbytes)
{
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void* p = malloc((size_t)n);
    if (p && fread(p, 1, (size_t)n, f) != (size_t)n) { free(p); p = NULL; }
    fclose(f);
    *nbytes = (size_t)n;
    return p;
}

int main(int argc, char** argv)
{
    if (argc != 5) { fprintf(stderr, "usage: %s weights.mmcw patches.u8 n features.f32\n", argv[0]); return 2; }
    const int64_t n = atoll(argv[3]);
    size_t wbytes = 0, pbytes = 0;
    void* blob = slurp(argv[1], &wbytes);
    void* patches = slurp(argv[2], &pbytes);
    if (!blob || !patches || n < 1 || pbytes != (size_t)n * MMC_PATCH * MMC_PATCH * 3) {
        fprintf(stderr, "bad inputs (%zu weight bytes, %zu patch bytes for n = %lld)\n", wbytes, pbytes, (long long)n);
        return 2;
    }
    if (mmc_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 3; }
    mmc_backbone* bb = NULL;
    if (mmc_backbone_create(blob, wbytes, MMC_ARCH_B0, 0, 64, &bb) != MMC_OK) { fprintf(stderr, "create: %s\n", mmc_last_error()); return 1; }
    free(blob);   /* the library has copied it */
    const int dim = mmc_feature_dim(bb);
    float* feats = (float*)malloc((size_t)n * (size_t)dim * sizeof(float));
    if (mmc_backbone_extract(bb, patches, n, feats, MMC_IN_HOST | MMC_OUT_HOST, NULL) != MMC_OK) {
        fprintf(stderr, "extract: %s\n", mmc_last_error());
        return 1;
    }
    FILE* out = fopen(argv[4], "wb");
    if (!out || fwrite(feats, sizeof(float), (size_t)n * (size_t)dim, out) != (size_t)n * (size_t)dim) { fprintf(stderr, "cannot write %s\n", argv[4]); return 1; }
    fclose(out);
    double sum = 0.0;
    for (int64_t i = 0; i < n * dim; ++i) sum += feats[i];
    printf("%lld patches -> (%lld, %d) features, sum %.6f, lanes %d, workspace %.1f MiB\n", (long long)n, (long long)n, dim, sum,
           mmc_backbone_lanes(bb), (double)mmc_backbone_workspace_bytes(bb) / (1024.0 * 1024.0));
    mmc_backbone_destroy(bb);
    free(feats);
    free(patches);
    return 0;
}
