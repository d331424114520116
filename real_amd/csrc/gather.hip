// gather.hip -- the one collective of the path behind the C ABI: the shards' results to the root over
// RCCL (xGMI point-to-point links), one process per GPU.  Reads are sharded contiguously over the ranks, every rank
// matches its shard against its own replica of the index, and nothing is reduced -- no read is seen by two ranks --
// so the collective is a concatenation in rank order: counts first (one all-gather of four words per rank), then the
// payload (grouped ncclSend / ncclRecv: every peer's bytes arrive on its own link, sizes may differ).  matchUnique
// gathers the 12-byte records, matchAll the variable-length hit lists with read indices and offsets rebased to the
// whole batch (the reference emits the unified hit list of every read of a block, matchAllImplementation.cpp:451-535).
//
// librccl is opened at run time (dlopen): the library has no link dependency on it, single-GPU users never load it,
// and a process that already carries a librccl (PyTorch's) shares that one.
#include "real_hip_internal.h"

#include <dlfcn.h>
#include <cstring>
#include <string>
#include <vector>
#include <rccl/rccl.h>

namespace {
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

RcclApi *rccl()
{
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names)
            if ((api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (api.lib) {
#define RH_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, name))
            RH_SYM(GetUniqueId, "ncclGetUniqueId"); RH_SYM(CommInitRank, "ncclCommInitRank"); RH_SYM(CommDestroy, "ncclCommDestroy");
            RH_SYM(AllGather, "ncclAllGather"); RH_SYM(Send, "ncclSend"); RH_SYM(Recv, "ncclRecv");
            RH_SYM(GroupStart, "ncclGroupStart"); RH_SYM(GroupEnd, "ncclGroupEnd"); RH_SYM(GetErrorString, "ncclGetErrorString");
#undef RH_SYM
            if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.Send || !api.Recv || !api.GroupStart || !api.GroupEnd)
                api.lib = nullptr;
        }
    }
    return api.lib ? &api : nullptr;
}

int rccl_fail(real_hip_ctx *ctx, const char *what, ncclResult_t r)
{
    RcclApi *R = rccl();
    std::string msg = std::string(what) + ": " + ((R && R->GetErrorString) ? R->GetErrorString(r) : "rccl error");
    return rh_fail(ctx, REAL_HIP_E_DEVICE, msg.c_str(), hipSuccess);
}
#define RH_NCCL(ctx, call)                                          \
    do {                                                            \
        ncclResult_t _r = (call);                                   \
        if (_r != ncclSuccess) return rccl_fail((ctx), #call, _r);  \
    } while (0)

// the read index of the hits of one rank, and its offsets, rebased to the whole batch
__global__ void rebase_hits_kernel(real_hip_hit *hits, uint64_t n, uint32_t first_read)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) hits[i].read += first_read;
}
__global__ void rebase_offsets_kernel(const uint64_t *__restrict__ in, uint64_t n_plus_1, uint64_t hits_before, uint64_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_plus_1) out[i] = in[i] + hits_before;
}

// counts first: {n_reads, n_hits, capacity for reads, capacity for hits, status} of every rank, known to every rank
// afterwards.  status = what the rank found wrong on its own side BEFORE the exchange (bad arguments, a scratch buffer it
// could not reserve: RH_ST_ERR | code) and which receive arrays it was given (RH_ST_HAS_*).  Every decision to give up is
// taken from the exchanged tuples alone, so it is the same on ALL ranks and is taken before any rank posts a send or a
// receive: nobody is left waiting in a send to a root that has returned.  (A rank with a local error still takes part in
// the exchange -- its peers are in it.)
#define RH_CNT 5
#define RH_ST_ERR 0x8000ull       /* low 15 bits: -(status code) of the rank's local error */
#define RH_ST_HAS_A 0x10000ull    /* root: info_all / hits_all given */
#define RH_ST_HAS_B 0x20000ull    /* root: score_all / offsets_all given */
int exchange_counts(real_hip_ctx *ctx, uint64_t n_local, uint64_t n_hits, uint64_t cap_reads, uint64_t cap_hits, uint64_t status, std::vector<uint64_t> &all)
{
    RcclApi *R = rccl();
    const int nr = ctx->comm_size;
    int rc = rh_reserve(ctx, ctx->comm_counts, (size_t)(nr + 1) * RH_CNT * 8);
    if (rc) return rc; // (a few hundred bytes: if this fails the device is gone and the peers' collective fails with it)
    uint64_t *d = (uint64_t *)ctx->comm_counts.p;
    const uint64_t mine[RH_CNT] = {n_local, n_hits, cap_reads, cap_hits, status};
    RH_HIP(ctx, hipMemcpyAsync(d + RH_CNT * (size_t)nr, mine, sizeof mine, hipMemcpyHostToDevice, ctx->stream));
    RH_NCCL(ctx, R->AllGather(d + RH_CNT * (size_t)nr, d, RH_CNT, ncclUint64, (ncclComm_t)ctx->comm, ctx->stream));
    all.assign((size_t)nr * RH_CNT, 0);
    RH_HIP(ctx, hipMemcpyAsync(all.data(), d, (size_t)nr * RH_CNT * 8, hipMemcpyDeviceToHost, ctx->stream));
    RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return REAL_HIP_OK;
}
// the first rank that reported a local error, or -1
int first_failed_rank(const std::vector<uint64_t> &all, int nr, int *code)
{
    for (int p = 0; p < nr; ++p)
        if (all[RH_CNT * (size_t)p + 4] & RH_ST_ERR) { *code = -(int)(all[RH_CNT * (size_t)p + 4] & 0x7fffull); return p; }
    return -1;
}
int peer_failed(real_hip_ctx *ctx, int me, int p, int code, const char *mine)
{
    if (p == me) return rh_fail(ctx, code, mine, hipSuccess);
    char msg[160];
    snprintf(msg, sizeof msg, "rank %d gave up before the gather (%s); no data was exchanged", p, real_hip_strerror(code));
    return rh_fail(ctx, code, msg, hipSuccess);
}

// payload: rank r's `bytes[r]` bytes from `send` to the root's `recv + offset[r]`; one group, every peer on its own link
int gather_bytes(real_hip_ctx *ctx, int root, const void *send, const std::vector<uint64_t> &bytes, void *recv, const std::vector<uint64_t> &offset)
{
    RcclApi *R = rccl();
    const int nr = ctx->comm_size, me = ctx->comm_rank;
    RH_NCCL(ctx, R->GroupStart());
    ncclResult_t r = ncclSuccess;
    if (bytes[(size_t)me]) r = R->Send(send, bytes[(size_t)me], ncclUint8, root, (ncclComm_t)ctx->comm, ctx->stream);
    if (me == root)
        for (int p = 0; p < nr && r == ncclSuccess; ++p)
            if (bytes[(size_t)p]) r = R->Recv((uint8_t *)recv + offset[(size_t)p], bytes[(size_t)p], ncclUint8, p, (ncclComm_t)ctx->comm, ctx->stream);
    const ncclResult_t e = R->GroupEnd();
    if (r != ncclSuccess) return rccl_fail(ctx, "ncclSend/ncclRecv", r);
    if (e != ncclSuccess) return rccl_fail(ctx, "ncclGroupEnd", e);
    return REAL_HIP_OK;
}
} // namespace

extern "C" int real_hip_comm_id(uint8_t id[REAL_HIP_COMM_ID_BYTES])
{
    RcclApi *R = rccl();
    if (!R || !id) return REAL_HIP_E_DEVICE;
    static_assert(REAL_HIP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId u;
    if (R->GetUniqueId(&u) != ncclSuccess) return REAL_HIP_E_DEVICE;
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return REAL_HIP_OK;
}

extern "C" int real_hip_comm_init(real_hip_ctx *ctx, const uint8_t id[REAL_HIP_COMM_ID_BYTES], int rank, int n_ranks)
{
    if (!ctx || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return REAL_HIP_E_INVALID;
    RH_HIP(ctx, hipSetDevice(ctx->device));
    RcclApi *R = rccl();
    if (!R) return rh_fail(ctx, REAL_HIP_E_DEVICE, "librccl could not be loaded", hipSuccess);
    if (ctx->comm) return rh_fail(ctx, REAL_HIP_E_STATE, "communicator already initialised", hipSuccess);
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    RH_NCCL(ctx, R->CommInitRank(&c, n_ranks, u, rank));
    ctx->comm = c; ctx->comm_rank = rank; ctx->comm_size = n_ranks;
    return REAL_HIP_OK;
}

void rh_comm_destroy(real_hip_ctx *ctx)
{
    RcclApi *R = rccl();
    if (ctx->comm && R) (void)R->CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr; ctx->comm_size = 1; ctx->comm_rank = 0;
}
extern "C" int real_hip_comm_destroy(real_hip_ctx *ctx)
{
    if (!ctx) return REAL_HIP_E_INVALID;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    rh_comm_destroy(ctx);
    return REAL_HIP_OK;
}

extern "C" int real_hip_gather_records(real_hip_ctx *ctx, int root, const uint64_t *info, const float *score, uint64_t n_local,
                                       uint64_t *info_all, float *score_all, uint64_t cap_all, uint64_t *n_all)
{
    if (!ctx) return REAL_HIP_E_INVALID;
    RH_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->comm) return rh_fail(ctx, REAL_HIP_E_STATE, "real_hip_comm_init first", hipSuccess);
    if (root < 0 || root >= ctx->comm_size) return rh_fail(ctx, REAL_HIP_E_INVALID, "gather arguments: root (must be the same valid rank on every rank)", hipSuccess);
    const int nr = ctx->comm_size, me = ctx->comm_rank;
    const char *why = "gather arguments";
    uint64_t status = (n_local && !info) ? (RH_ST_ERR | (uint64_t)(-REAL_HIP_E_INVALID)) : 0;
    if (me == root) status |= (info_all ? RH_ST_HAS_A : 0) | (score_all ? RH_ST_HAS_B : 0);
    std::vector<uint64_t> all;
    int rc = exchange_counts(ctx, n_local, score ? 1 : 0, me == root ? cap_all : 0, 0, status, all);
    if (rc) return rc;
    int code = 0;
    const int bad = first_failed_rank(all, nr, &code);
    if (bad >= 0) return peer_failed(ctx, me, bad, code, why);
    uint64_t total = 0;
    std::vector<uint64_t> bi((size_t)nr), oi((size_t)nr), bs((size_t)nr), os((size_t)nr);
    for (int p = 0; p < nr; ++p) { oi[(size_t)p] = total * 8; os[(size_t)p] = total * 4; bi[(size_t)p] = all[RH_CNT * (size_t)p] * 8; bs[(size_t)p] = all[RH_CNT * (size_t)p] * 4; total += all[RH_CNT * (size_t)p]; }
    if (n_all) *n_all = total;
    if (total > all[RH_CNT * (size_t)root + 2]) return rh_fail(ctx, REAL_HIP_E_OVERFLOW, "the root's record arrays are too small (every rank reports this)", hipSuccess);
    // (every rank sees what the root was given and whether the root gathers scores: the same verdict everywhere)
    const uint64_t rs = all[RH_CNT * (size_t)root + 4];
    const bool root_scores = all[RH_CNT * (size_t)root + 1] != 0;
    if (total && (!(rs & RH_ST_HAS_A) || (root_scores && !(rs & RH_ST_HAS_B))))
        return rh_fail(ctx, REAL_HIP_E_INVALID, "null receive arrays on the root (every rank reports this)", hipSuccess);
    if ((score != nullptr) != root_scores) return rh_fail(ctx, REAL_HIP_E_INVALID, "score arrays on some ranks only", hipSuccess);
    if ((rc = gather_bytes(ctx, root, info, bi, info_all, oi))) return rc;
    if (score && (rc = gather_bytes(ctx, root, score, bs, score_all, os))) return rc;
    RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return REAL_HIP_OK;
}

extern "C" int real_hip_gather_hits(real_hip_ctx *ctx, int root, const real_hip_hit *hits, const uint64_t *hit_offsets, uint64_t n_local,
                                    uint64_t n_hits_local, real_hip_hit *hits_all, uint64_t cap_hits, uint64_t *offsets_all, uint64_t cap_reads,
                                    uint64_t *n_reads_all, uint64_t *n_hits_all)
{
    if (!ctx) return REAL_HIP_E_INVALID;
    RH_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->comm) return rh_fail(ctx, REAL_HIP_E_STATE, "real_hip_comm_init first", hipSuccess);
    if (root < 0 || root >= ctx->comm_size) return rh_fail(ctx, REAL_HIP_E_INVALID, "gather arguments: root (must be the same valid rank on every rank)", hipSuccess);
    const int nr = ctx->comm_size, me = ctx->comm_rank;
    const char *why = "gather arguments";
    uint64_t status = (!hit_offsets || (n_hits_local && !hits)) ? (RH_ST_ERR | (uint64_t)(-REAL_HIP_E_INVALID)) : 0;
    uint64_t *scratch = nullptr;
    if (me == root) {
        status |= (hits_all ? RH_ST_HAS_A : 0) | (offsets_all ? RH_ST_HAS_B : 0);
        // the shards' offset arrays side by side: reserved for what the caller's offsets_all can hold, BEFORE the exchange,
        // so that a failure here is every rank's knowledge before anyone sends
        const int rr = rh_reserve(ctx, ctx->comm_scratch, (size_t)(cap_reads + (uint64_t)nr + 1) * 8);
        if (rr && !(status & RH_ST_ERR)) { status |= RH_ST_ERR | (uint64_t)(-rr); why = "scratch for the gathered offsets"; }
        scratch = (uint64_t *)ctx->comm_scratch.p;
    }
    std::vector<uint64_t> all;
    int rc = exchange_counts(ctx, n_local, n_hits_local, me == root ? cap_reads : 0, me == root ? cap_hits : 0, status, all);
    if (rc) return rc;
    int code = 0;
    const int bad = first_failed_rank(all, nr, &code);
    if (bad >= 0) return peer_failed(ctx, me, bad, code, why);
    uint64_t reads = 0, nh = 0;
    std::vector<uint64_t> bh((size_t)nr), oh((size_t)nr), bo((size_t)nr), oo((size_t)nr), first((size_t)nr), before((size_t)nr);
    for (int p = 0; p < nr; ++p) {
        const uint64_t n = all[RH_CNT * (size_t)p], h = all[RH_CNT * (size_t)p + 1];
        first[(size_t)p] = reads; before[(size_t)p] = nh;
        oh[(size_t)p] = nh * sizeof(real_hip_hit); bh[(size_t)p] = h * sizeof(real_hip_hit);
        oo[(size_t)p] = (reads + (uint64_t)p) * 8; bo[(size_t)p] = (n + 1) * 8; // the shards' offset arrays side by side in a scratch
        reads += n; nh += h;
    }
    if (n_reads_all) *n_reads_all = reads;
    if (n_hits_all) *n_hits_all = nh;
    if (reads > all[RH_CNT * (size_t)root + 2] || nh > all[RH_CNT * (size_t)root + 3])
        return rh_fail(ctx, REAL_HIP_E_OVERFLOW, "the root's hit / offset arrays are too small (every rank reports this)", hipSuccess);
    if (reads > 0xffffffffull) return rh_fail(ctx, REAL_HIP_E_INVALID, "more than 2^32 reads in the gathered batch (real_hip_hit.read is 32 bit)", hipSuccess);
    const uint64_t rs = all[RH_CNT * (size_t)root + 4];
    if ((nh && !(rs & RH_ST_HAS_A)) || !(rs & RH_ST_HAS_B))
        return rh_fail(ctx, REAL_HIP_E_INVALID, "null receive arrays on the root (every rank reports this)", hipSuccess);
    if ((rc = gather_bytes(ctx, root, hits, bh, hits_all, oh))) return rc;
    if ((rc = gather_bytes(ctx, root, hit_offsets, bo, scratch, oo))) return rc;
    if (me == root) {
        for (int p = 0; p < nr; ++p) {
            const uint64_t n = all[RH_CNT * (size_t)p], h = all[RH_CNT * (size_t)p + 1];
            if (h && first[(size_t)p])
                hipLaunchKernelGGL(rebase_hits_kernel, dim3((unsigned)((h + 255) / 256)), dim3(256), 0, ctx->stream, hits_all + before[(size_t)p], h, (uint32_t)first[(size_t)p]);
            // (n + 1 entries: the last one of rank p is overwritten by the first one of rank p + 1 -- the same value)
            hipLaunchKernelGGL(rebase_offsets_kernel, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const uint64_t *)(scratch + first[(size_t)p] + (uint64_t)p), n + 1, before[(size_t)p], offsets_all + first[(size_t)p]);
        }
        RH_HIP(ctx, hipGetLastError());
    }
    RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return REAL_HIP_OK;
}
