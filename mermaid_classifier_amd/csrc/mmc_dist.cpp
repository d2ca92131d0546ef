// mmc_dist.cpp -- the one exchange step of the sharded path behind the C ABI: an all-gather of the ranks' feature blocks over
// RCCL (xGMI inside a node), for hosts that are not Python (the Python package does the same through torch.distributed,
// mermaid_classifier_amd/dist.py).  Replaces nothing in the reference's code: its scale-out is one job per source id
// (scripts/launch_processing.py:59-66, 199-233) and the feature matrices meet on S3; SURVEY.md section 8(b) sketches this entry.
//
// librccl is resolved at the first call with dlopen (the library the process already has -- torch ships its own copy -- else the
// ROCm one), so libmermaid_mi355.so itself does not depend on it and single-GPU users never load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/mmc.h"

int mmc_fail(int code, const char* fmt, ...);   // mmc_api.cpp

namespace {
struct NcclId { char internal[128]; };
typedef void* NcclComm;
enum { NCCL_FLOAT = 7 };   // ncclFloat32 (rccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4, ncclUint64 5, ncclFloat16 6, ncclFloat32 7)
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

int load_rccl()
{
    if (g_rccl.lib) return MMC_OK;
    const char* override_path = getenv("MMC_RCCL_LIBRARY");
    const char* names[] = {override_path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (int pass = 0; pass < 2 && !h; ++pass)   // pass 0: a copy the process has already mapped; pass 1: load one
        for (const char* n : names) {
            if (!n || !*n) continue;
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) break;
        }
    if (!h) return mmc_fail(MMC_ERR_HIP, "librccl not found (set MMC_RCCL_LIBRARY): %s", dlerror());
#define SYM(field, name)                                                                      \
    do {                                                                                      \
        *reinterpret_cast<void**>(&g_rccl.field) = dlsym(h, name);                            \
        if (!g_rccl.field) return mmc_fail(MMC_ERR_HIP, "librccl lacks %s", name);            \
    } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(AllGather, "ncclAllGather");
    SYM(Broadcast, "ncclBroadcast");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.lib = h;
    return MMC_OK;
}
#define RCCL_TRY(expr)                                                                                       \
    do {                                                                                                     \
        int r_ = (expr);                                                                                     \
        if (r_ != 0) return mmc_fail(MMC_ERR_HIP, "%s: %s", #expr, g_rccl.GetErrorString(r_));               \
    } while (0)
}   // namespace

struct mmc_dist {
    NcclComm comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

extern "C" int mmc_dist_unique_id(unsigned char id[MMC_DIST_ID_BYTES])
{
    if (!id) return mmc_fail(MMC_ERR_ARG, "mmc_dist_unique_id: null id");
    int r = load_rccl();
    if (r) return r;
    NcclId u;
    RCCL_TRY(g_rccl.GetUniqueId(&u));
    static_assert(sizeof u == MMC_DIST_ID_BYTES, "id size");
    memcpy(id, &u, sizeof u);
    return MMC_OK;
}

extern "C" int mmc_dist_create(const unsigned char id[MMC_DIST_ID_BYTES], int rank, int world, int device, mmc_dist** out)
{
    if (!id || !out) return mmc_fail(MMC_ERR_ARG, "mmc_dist_create: null argument");
    if (world < 1 || rank < 0 || rank >= world) return mmc_fail(MMC_ERR_ARG, "mmc_dist_create: rank %d of %d", rank, world);
    int r = load_rccl();
    if (r) return r;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return mmc_fail(MMC_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    NcclId u;
    memcpy(&u, id, sizeof u);
    mmc_dist* d = new mmc_dist;
    d->rank = rank; d->world = world; d->device = device;
    int rr = g_rccl.CommInitRank(&d->comm, world, u, rank);   // collective: every rank calls it with the same id
    if (rr != 0) {
        delete d;
        return mmc_fail(MMC_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.GetErrorString(rr));
    }
    *out = d;
    return MMC_OK;
}

extern "C" void mmc_dist_destroy(mmc_dist* d)
{
    if (!d) return;
    if (d->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(d->comm);
    delete d;
}

extern "C" int mmc_gather_features(mmc_dist* d, const float* local, int64_t n_local, int dim, const int64_t* counts, float* all,
                                   void* hip_stream)
{
    if (!d || !all || dim < 1 || n_local < 0 || (n_local > 0 && !local)) return mmc_fail(MMC_ERR_ARG, "mmc_gather_features: bad argument");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if (counts) {
        if (counts[d->rank] != n_local)
            return mmc_fail(MMC_ERR_ARG, "mmc_gather_features: counts[%d] = %lld but n_local = %lld", d->rank, (long long)counts[d->rank],
                            (long long)n_local);
        for (int r = 0; r < d->world; ++r)
            if (counts[r] < 0) return mmc_fail(MMC_ERR_ARG, "mmc_gather_features: counts[%d] < 0", r);
    }
    hipError_t e = hipSetDevice(d->device);
    if (e != hipSuccess) return mmc_fail(MMC_ERR_HIP, "hipSetDevice(%d): %s", d->device, hipGetErrorString(e));
    if (!counts) {   // equal blocks: one all-gather (rank r's block lands at row r * n_local)
        if (n_local > 0) RCCL_TRY(g_rccl.AllGather(local, all, (size_t)n_local * dim, NCCL_FLOAT, d->comm, st));
        return MMC_OK;
    }
    // ragged blocks (contiguous sharding of n rows over `world` ranks leaves the last blocks one row short): one broadcast per
    // rank inside a group -- a single fused operation on the wire, each block written straight to its rows, no padding pass
    RCCL_TRY(g_rccl.GroupStart());
    int64_t off = 0;
    int rr = 0;
    for (int r = 0; r < d->world && rr == 0; ++r) {
        float* dst = all + (size_t)off * dim;
        if (counts[r] > 0) rr = g_rccl.Broadcast(r == d->rank ? (const void*)local : (const void*)dst, dst, (size_t)counts[r] * dim, NCCL_FLOAT, r, d->comm, st);
        off += counts[r];
    }
    int re = g_rccl.GroupEnd();
    if (rr != 0) return mmc_fail(MMC_ERR_HIP, "ncclBroadcast: %s", g_rccl.GetErrorString(rr));
    if (re != 0) return mmc_fail(MMC_ERR_HIP, "ncclGroupEnd: %s", g_rccl.GetErrorString(re));
    return MMC_OK;
}
