// real_hip_api.hip -- the C ABI of include/real_hip.h: context, uploads, staging,
// launches.  No CPU fallback: every entry point either runs the HIP path or fails.
#include "real_hip_internal.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

int rh_count_wild(real_hip_ctx *ctx, uint64_t n);

// ---------------------------------------------------------------------------
// plumbing
// ---------------------------------------------------------------------------
int rh_fail(real_hip_ctx *ctx, int status, const char *what, hipError_t e)
{
    if (ctx) {
        char buf[512];
        snprintf(buf, sizeof buf, "%s: %s%s%s", real_hip_strerror(status), what, e != hipSuccess ? ": " : "",
                 e != hipSuccess ? hipGetErrorString(e) : "");
        ctx->last_error = buf;
    }
    (void)hipGetLastError(); // clear sticky state
    return status;
}

double rh_now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
int rh_reserve(real_hip_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return REAL_HIP_OK;
    rh_release(ctx, b);
    const double t0 = rh_now_ms();
    hipError_t e = hipMalloc(&b.p, bytes);
    if (ctx) { ctx->alloc_ms += rh_now_ms() - t0; ctx->alloc_calls++; ctx->alloc_bytes += bytes; }
    if (e != hipSuccess) { b.p = nullptr; return rh_fail(ctx, REAL_HIP_E_NOMEM, "hipMalloc", e); }
    b.cap = bytes;
    return REAL_HIP_OK;
}
void rh_release(real_hip_ctx *ctx, DevBuf &b)
{
    if (b.p) {
        const double t0 = rh_now_ms();
        (void)hipFree(b.p); // (waits for the device to go idle)
        if (ctx) { ctx->free_ms += rh_now_ms() - t0; ctx->free_calls++; }
    }
    b.p = nullptr; b.cap = 0;
}
void rh_release(DevBuf &b) { rh_release(nullptr, b); }

RhTimer::RhTimer(real_hip_ctx *ctx, int w) : c(ctx), which(w)
{
    if (c->timing) (void)hipEventRecord(c->ev0, c->stream);
}
RhTimer::~RhTimer()
{
    if (!c->timing) return;
    (void)hipEventRecord(c->ev1, c->stream);
    if (hipEventSynchronize(c->ev1) == hipSuccess) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) { c->k_ms[which] += ms; c->k_n[which] += 1; }
    }
}

// asynchronous timing: one event pair per launch, resolved (elapsed time read) after the next sync
static hipEvent_t rh_event(real_hip_ctx *c)
{
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void rh_time_begin(real_hip_ctx *c, hipStream_t st, int which)
{
    if (!c->timing) return;
    real_hip_ctx::Pending p; p.a = rh_event(c); p.b = nullptr; p.which = which;
    (void)hipEventRecord(p.a, st);
    c->pending.push_back(p);
}
void rh_time_end(real_hip_ctx *c, hipStream_t st)
{
    if (!c->timing || c->pending.empty()) return;
    for (size_t i = c->pending.size(); i-- > 0;)
        if (!c->pending[i].b) { c->pending[i].b = rh_event(c); (void)hipEventRecord(c->pending[i].b, st); break; }
}
void rh_time_resolve(real_hip_ctx *c)
{
    // (pairs whose end has not executed yet -- a batch still in flight in the other slot -- stay pending)
    size_t keep = 0;
    for (auto &p : c->pending) {
        if (p.a && p.b && hipEventQuery(p.b) == hipSuccess) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { c->k_ms[p.which] += ms; c->k_n[p.which] += 1; }
            c->ev_pool.push_back(p.a);
            c->ev_pool.push_back(p.b);
        } else {
            c->pending[keep++] = p;
        }
    }
    (void)hipGetLastError(); // (hipErrorNotReady of the queries)
    c->pending.resize(keep);
}

extern "C" const char *real_hip_strerror(int s)
{
    switch (s) {
    case REAL_HIP_OK: return "ok";
    case REAL_HIP_E_INVALID: return "invalid argument";
    case REAL_HIP_E_NOMEM: return "out of memory";
    case REAL_HIP_E_DEVICE: return "HIP runtime error";
    case REAL_HIP_E_OVERFLOW: return "output capacity too small";
    case REAL_HIP_E_STATE: return "text or index not set";
    case REAL_HIP_E_UNSUPPORTED: return "unsupported";
    default: return "unknown status";
    }
}
extern "C" const char *real_hip_last_error(const real_hip_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }
extern "C" int real_hip_abi_version(void) { return REAL_HIP_ABI_VERSION; }

// ---------------------------------------------------------------------------
// scoring table: Scoring::init + Scoring::getScore(char,char,int)
// (Scoring.cpp:28-36, 61-133, 155-171).  Plain IEEE double arithmetic in the
// reference's operation order; built once on the host.
// ---------------------------------------------------------------------------
static const double kQPrb[65] = {
    1.0000000, 0.7943282, 0.6309573, 0.5011872, 0.3981072, 0.3162278, 0.2511886, 0.1995262, 0.1584893, 0.1258925,
    0.1000000, 0.0794328, 0.0630957, 0.0501187, 0.0398107, 0.0316228, 0.0251189, 0.0199526, 0.0158489, 0.0125893,
    0.0100000, 0.0079433, 0.0063096, 0.0050119, 0.0039811, 0.0031623, 0.0025119, 0.0019953, 0.0015849, 0.0012589,
    0.0010000, 0.0007943, 0.0006310, 0.0005012, 0.0003981, 0.0003162, 0.0002512, 0.0001995, 0.0001585, 0.0001259,
    0.0001000, 0.0000794, 0.0000631, 0.0000501, 0.0000398, 0.0000316, 0.0000251, 0.0000200, 0.0000158, 0.0000126,
    0.0000100, 0.0000079, 0.0000063, 0.0000050, 0.0000040, 0.0000032, 0.0000025, 0.0000020, 0.0000016, 0.0000013,
    0.0000010, 0.0000008, 0.0000006, 0.0000005, 0.0000004};

extern "C" void real_hip_scoring_table(double similarity, double gc, double trans, double err, double bias, double LL[1024])
{
    volatile double R[4][4]; // the reference stores every intermediate (-ffloat-store, src/Makefile.am:95-96)
    volatile double t1 = trans * (1 - similarity);
    volatile double t2 = (1 - trans) * (1 - similarity);
    const double transit = t1, transver = t2;
    double bg[4] = {(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2};
    bias = bias * (1 - gc) / gc;
    R[0][2] = transit / (bias + 1) / (1 - gc);
    R[3][1] = transit / (bias + 1) / (1 - gc);
    R[2][0] = transit / (bias + 1) / gc * bias;
    R[1][3] = transit / (bias + 1) / gc * bias;
    R[0][1] = transver / 2 / (bias + 1) / (1 - gc);
    R[3][2] = transver / 2 / (bias + 1) / (1 - gc);
    R[0][3] = transver / 2 / (bias + 1) / (1 - gc);
    R[3][0] = transver / 2 / (bias + 1) / (1 - gc);
    R[1][0] = transver / 2 / (bias + 1) / gc * bias;
    R[2][3] = transver / 2 / (bias + 1) / gc * bias;
    R[1][2] = transver / 2 / (bias + 1) / gc * bias;
    R[2][1] = transver / 2 / (bias + 1) / gc * bias;
    R[0][0] = 1 - R[0][1] - R[0][2] - R[0][3];
    R[3][3] = 1 - R[3][0] - R[3][1] - R[3][2];
    R[2][2] = 1 - R[2][0] - R[2][1] - R[2][3];
    R[1][1] = 1 - R[1][0] - R[1][2] - R[1][3];
    for (int x = 0; x < 4; ++x)
        for (int y = 0; y < 4; ++y) {
            R[x][y] *= 1 - err;
            R[x][y] /= bg[y];
        }
    for (unsigned c0 = 0; c0 < 4; ++c0)
        for (unsigned c1 = 0; c1 < 4; ++c1)
            for (unsigned q = 0; q < 64; ++q)
                LL[(c0 << 8) | (c1 << 6) | q] = std::log(R[c0][c1]) / std::log(2.0) * (1 - kQPrb[q]);
}

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
extern "C" int real_hip_create(real_hip_ctx **out, const real_hip_params *p)
{
    if (!out || !p || p->struct_size != sizeof(real_hip_params)) return REAL_HIP_E_INVALID;
    // RealOptions.cpp:434-453 clamps these; at the ABI they are errors
    if (p->seedl < 4 || p->seedl > 64 || (p->seedl % 4) || p->seedkmax > 2 || p->totalkmax > 15) return REAL_HIP_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || p->device < 0 || p->device >= ndev) {
        (void)hipGetLastError();
        return REAL_HIP_E_DEVICE; // no GPU: the product path fails loudly, there is no CPU fallback
    }
    real_hip_ctx *c = new (std::nothrow) real_hip_ctx();
    if (!c) return REAL_HIP_E_NOMEM;
    c->prm = *p;
    c->device = p->device;
    int rc = REAL_HIP_OK;
    do {
        if (hipSetDevice(c->device) != hipSuccess) { rc = REAL_HIP_E_DEVICE; break; }
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { rc = REAL_HIP_E_DEVICE; break; }
        if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) { rc = REAL_HIP_E_DEVICE; break; }
        if (hipHostMalloc((void **)&c->h_state, 3 * 2 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) { rc = REAL_HIP_E_NOMEM; break; }
        memset(c->h_state, 0, 3 * 2 * sizeof(unsigned long long));
        if ((rc = rh_reserve(c, c->LL, 1024 * sizeof(double)))) break;
        if ((rc = rh_reserve(c, c->counters, (size_t)(RH_CSTRIPES + 1) * 16 * sizeof(uint64_t)))) break;
        if (hipMemcpy(c->LL.p, p->LL, 1024 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { rc = REAL_HIP_E_DEVICE; break; }
        if (hipMemset(c->counters.p, 0, (size_t)(RH_CSTRIPES + 1) * 16 * sizeof(uint64_t)) != hipSuccess) { rc = REAL_HIP_E_DEVICE; break; }
    } while (0);
    if (rc) { real_hip_destroy(c); return rc; }
    *out = c;
    return REAL_HIP_OK;
}

extern "C" void real_hip_destroy(real_hip_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    rh_comm_destroy(c);
    DevBuf *all[] = {&c->comm_counts, &c->comm_scratch, &c->text, &c->wild, &c->frag, &c->LL, &c->counters, &c->s_bases, &c->s_qual, &c->s_off, &c->s_info,
                     &c->s_score, &c->maxpatl, &c->ovf_list, &c->ovf2_list, &c->ovf_count, &c->raw, &c->raw_count, &c->hit_cnt, &c->big_list, &c->all_cursor, &c->p_text, &c->p_nl, &c->p_scal, &c->p_spans, &c->p_off,
                     &c->p_len1, &c->p_bases, &c->p_qual, &c->keys_a,
                     &c->keys_b, &c->vals_a, &c->vals_b, &c->sort_tmp, &c->hit_off, &c->s_hits, &c->s_nflags};
    for (DevBuf *b : all) rh_release(*b);
    for (int k = 0; k < 6; ++k) { rh_release(c->ent[k]); rh_release(c->bkt[k]); }
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->down_stream) (void)hipStreamSynchronize(c->down_stream);
    for (int i = 0; i < REAL_HIP_SLOTS; ++i) {
        RhSlot &S = c->slot[i];
        DevBuf *sb[] = {&S.bases, &S.qual, &S.off, &S.nflags, &S.info, &S.score};
        for (DevBuf *b : sb) rh_release(*b);
        if (S.up) (void)hipEventDestroy(S.up);
        if (S.matched) (void)hipEventDestroy(S.matched);
        if (S.done) (void)hipEventDestroy(S.done);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->down_stream) (void)hipStreamDestroy(c->down_stream);
    if (c->h_state) (void)hipHostFree(c->h_state);
    rh_time_resolve(c);
    for (auto &p : c->pending) { if (p.a) (void)hipEventDestroy(p.a); if (p.b) (void)hipEventDestroy(p.b); }
    c->pending.clear();
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

#define RH_ENTER(ctx)                                   \
    if (!(ctx)) return REAL_HIP_E_INVALID;              \
    RH_HIP((ctx), hipSetDevice((ctx)->device));

extern "C" int real_hip_set_match_params(real_hip_ctx *ctx, uint32_t seedkmax, uint32_t totalkmax, uint32_t scores, double filter_mult)
{
    if (!ctx) return REAL_HIP_E_INVALID;
    if (seedkmax > 2 || totalkmax > 15) return rh_fail(ctx, REAL_HIP_E_INVALID, "seedkmax <= 2, totalkmax <= 15 (RealOptions.cpp:172-180, 449-453)", hipSuccess);
    ctx->prm.seedkmax = seedkmax; ctx->prm.totalkmax = totalkmax; ctx->prm.scores = scores ? 1u : 0u; ctx->prm.filter_mult = filter_mult;
    return REAL_HIP_OK;
}

extern "C" int real_hip_wait_event(real_hip_ctx *ctx, void *hip_event)
{
    if (!ctx || !hip_event) return REAL_HIP_E_INVALID;
    RH_HIP(ctx, hipSetDevice(ctx->device));
    RH_HIP(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)hip_event, 0));
    return REAL_HIP_OK;
}

extern "C" int real_hip_device_memory(real_hip_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes)
{
    RH_ENTER(ctx);
    size_t f = 0, t = 0;
    RH_HIP(ctx, hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return REAL_HIP_OK;
}

// ---------------------------------------------------------------------------
// text
// ---------------------------------------------------------------------------
static int set_frag(real_hip_ctx *ctx, uint32_t fileid, uint64_t n, const uint64_t *frag_start, uint32_t n_frag)
{
    if (!frag_start || !n_frag || frag_start[0] != 0 || frag_start[n_frag] != n)
        return rh_fail(ctx, REAL_HIP_E_INVALID, "frag_start must begin at 0 and end at n_bases", hipSuccess);
    for (uint32_t i = 0; i < n_frag; ++i)
        if (frag_start[i + 1] <= frag_start[i])
            return rh_fail(ctx, REAL_HIP_E_INVALID, "fragment starts must be strictly increasing (empty FASTA records are not representable in the reference's RangeVector)", hipSuccess);
    // UniqueMatchInfo.hpp:31-32: 6 bits of file id, 16 bits of fragment id; positions are u32
    if (fileid >= 64 || n_frag > 65536 || n > 0xffffffffull)
        return rh_fail(ctx, REAL_HIP_E_INVALID, "fileid/fragment/length exceeds the UniqueMatchInfo record", hipSuccess);
    int rc = rh_reserve(ctx, ctx->frag, ((size_t)n_frag + 1) * 8);
    if (rc) return rc;
    RH_HIP(ctx, hipMemcpyAsync(ctx->frag.p, frag_start, ((size_t)n_frag + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_frag = n_frag; ctx->fileid = fileid; ctx->n_bases = n;
    return REAL_HIP_OK;
}

static int alloc_text(real_hip_ctx *ctx, uint64_t n)
{
    // padded: kernels write whole 64-symbol groups, and the matcher requests the words of text[pos, pos + patl) of a
    // candidate before it has checked that the read ends inside the text (cand_load: up to RH_MAXW + 2 words from the
    // last window start on)
    size_t tw = (size_t)((n + 63) / 64) * 2 + RH_MAXW + 6, ww = (size_t)((n + 63) / 64) + 4;
    int rc = rh_reserve(ctx, ctx->text, tw * 8);
    if (rc) return rc;
    if ((rc = rh_reserve(ctx, ctx->wild, ww * 8))) return rc;
    RH_HIP(ctx, hipMemsetAsync(ctx->text.p, 0, tw * 8, ctx->stream));
    RH_HIP(ctx, hipMemsetAsync(ctx->wild.p, 0, ww * 8, ctx->stream));
    return REAL_HIP_OK;
}

extern "C" int real_hip_set_text(real_hip_ctx *ctx, uint32_t fileid, const uint64_t *text2bit, const uint64_t *wildbits,
                                 uint64_t n, const uint64_t *frag_start, uint32_t n_frag)
{
    RH_ENTER(ctx);
    if (!text2bit || !wildbits) return rh_fail(ctx, REAL_HIP_E_INVALID, "null text", hipSuccess);
    ctx->have_text = false; ctx->have_index = false;
    int rc = set_frag(ctx, fileid, n, frag_start, n_frag);
    if (rc) return rc;
    if ((rc = alloc_text(ctx, n))) return rc;
    RH_HIP(ctx, hipMemcpyAsync(ctx->text.p, text2bit, (size_t)((2 * n + 63) / 64) * 8, hipMemcpyHostToDevice, ctx->stream));
    RH_HIP(ctx, hipMemcpyAsync(ctx->wild.p, wildbits, (size_t)((n + 63) / 64) * 8, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = rh_count_wild(ctx, n))) return rc;
    ctx->have_text = true;
    return REAL_HIP_OK;
}

extern "C" int real_hip_set_text_symbols(real_hip_ctx *ctx, uint32_t fileid, const uint8_t *sym, uint64_t n, int on_device,
                                         const uint64_t *frag_start, uint32_t n_frag)
{
    RH_ENTER(ctx);
    if (!sym && n) return rh_fail(ctx, REAL_HIP_E_INVALID, "null symbols", hipSuccess);
    ctx->have_text = false; ctx->have_index = false;
    int rc = set_frag(ctx, fileid, n, frag_start, n_frag);
    if (rc) return rc;
    if ((rc = alloc_text(ctx, n))) return rc;
    const uint8_t *d_sym = sym;
    ScopedBuf tmp(ctx);
    if (!on_device) {
        if ((rc = rh_reserve(ctx, tmp, n ? n : 1))) return rc;
        RH_HIP(ctx, hipMemcpyAsync(tmp.p, sym, n, hipMemcpyHostToDevice, ctx->stream));
        d_sym = (const uint8_t *)tmp.p;
    }
    if ((rc = rh_pack_text(ctx, d_sym, n))) return rc; // (synchronous: tmp may go)
    ctx->have_text = true;
    return REAL_HIP_OK;
}

// ---------------------------------------------------------------------------
// index
// ---------------------------------------------------------------------------
extern "C" int real_hip_set_index_block(real_hip_ctx *ctx, uint64_t n, const void *const sign[6], const uint32_t *const pos[6])
{
    RH_ENTER(ctx);
    if (!ctx->have_text) return rh_fail(ctx, REAL_HIP_E_STATE, "set the text first", hipSuccess);
    if (!sign || !pos || n > 0xffffffffull) return rh_fail(ctx, REAL_HIP_E_INVALID, "bad index block", hipSuccess);
    const unsigned sb = ctx->prm.seedl <= 32 ? 4 : 8; // real.cpp:219-229
    ctx->have_index = false;
    for (int k = 0; k < 6; ++k)
        if (n && (!sign[k] || !pos[k])) return rh_fail(ctx, REAL_HIP_E_INVALID, "null list", hipSuccess);
    ctx->n_entries = n;
    rh_choose_tables(ctx, n);
    int rc = rh_index_from_host_lists(ctx, n, sign, pos, sb);
    if (rc) return rc;
    ctx->have_index = true;
    return REAL_HIP_OK;
}

extern "C" int real_hip_build_index_block(real_hip_ctx *ctx, uint64_t first_window, uint64_t max_entries,
                                          uint64_t *n_entries, int *have_next)
{
    RH_ENTER(ctx);
    if (!ctx->have_text) return rh_fail(ctx, REAL_HIP_E_STATE, "set the text first", hipSuccess);
    ctx->have_index = false;
    return rh_index_build_device(ctx, first_window, max_entries, n_entries, have_next);
}

extern "C" int real_hip_index_build_stats(real_hip_ctx *ctx, real_hip_build_stats *out, int reset)
{
    if (!ctx || !out || out->struct_size != sizeof(real_hip_build_stats)) return REAL_HIP_E_INVALID;
    out->wall_ms = ctx->build_wall_ms; out->kernel_ms = ctx->k_ms[REAL_HIP_K_INDEX];
    out->alloc_ms = ctx->alloc_ms; out->free_ms = ctx->free_ms;
    out->alloc_bytes = ctx->alloc_bytes; out->alloc_calls = ctx->alloc_calls; out->free_calls = ctx->free_calls;
    if (reset) {
        ctx->build_wall_ms = ctx->alloc_ms = ctx->free_ms = 0; ctx->alloc_bytes = ctx->alloc_calls = ctx->free_calls = 0;
        ctx->k_ms[REAL_HIP_K_INDEX] = 0; ctx->k_n[REAL_HIP_K_INDEX] = 0;
    }
    return REAL_HIP_OK;
}

extern "C" int real_hip_index_info(const real_hip_ctx *ctx, uint64_t *n_entries, uint32_t *prefix_bits)
{
    if (!ctx || !ctx->have_index) return REAL_HIP_E_STATE;
    if (n_entries) *n_entries = ctx->n_entries;
    if (prefix_bits) *prefix_bits = ctx->pb;
    return REAL_HIP_OK;
}

extern "C" int real_hip_index_table_kind(const real_hip_ctx *ctx, uint32_t *kind)
{
    if (!ctx || !kind) return REAL_HIP_E_INVALID;
    if (!ctx->have_index) return REAL_HIP_E_STATE;
    *kind = (uint32_t)ctx->fine;
    return REAL_HIP_OK;
}

extern "C" int real_hip_index_download(real_hip_ctx *ctx, int list, uint32_t *entries, uint32_t *bucket)
{
    RH_ENTER(ctx);
    if (!ctx->have_index) return rh_fail(ctx, REAL_HIP_E_STATE, "no index", hipSuccess);
    if (list < 0 || list > 5) return rh_fail(ctx, REAL_HIP_E_INVALID, "list", hipSuccess);
    const uint64_t n = ctx->n_entries;
    if (ctx->fine == 3) { // bucket rows: entries and bucket starts by an ordered traversal of the rows
        ScopedBuf e(ctx), st(ctx);
        int rc;
        if ((rc = rh_reserve(ctx, e, (n ? n : 1) * sizeof(uint2)))) return rc;
        if ((rc = rh_reserve(ctx, st, (((size_t)1 << ctx->pb) + 1) * 4))) return rc;
        rc = rh_rows_unpack(ctx, list, (uint2 *)e.p, (uint32_t *)st.p);
        hipError_t he = hipSuccess;
        if (!rc && n && entries) he = hipMemcpy(entries, e.p, n * sizeof(uint2), hipMemcpyDeviceToHost);
        if (!rc && he == hipSuccess && bucket) he = hipMemcpy(bucket, st.p, (((size_t)1 << ctx->pb) + 1) * 4, hipMemcpyDeviceToHost);
        if (rc) return rc;
        if (he != hipSuccess) return rh_fail(ctx, REAL_HIP_E_DEVICE, "index download", he);
        return REAL_HIP_OK;
    }
    if (n && entries) RH_HIP(ctx, hipMemcpy(entries, ctx->ent[list].p, n * sizeof(uint2), hipMemcpyDeviceToHost));
    if (bucket) {
        const size_t nbk = ((size_t)1 << ctx->pb) + 1;
        if (ctx->fine) RH_HIP(ctx, hipMemcpy2D(bucket, 4, ctx->bkt[list].p, 16, 4, nbk, hipMemcpyDeviceToHost)); // the .x of every uint4 (kinds 1, 2)
        else RH_HIP(ctx, hipMemcpy(bucket, ctx->bkt[list].p, nbk * 4, hipMemcpyDeviceToHost));
    }
    return REAL_HIP_OK;
}

int rh_index_export(real_hip_ctx *ctx, int list, void *h_sign, uint32_t *h_pos);
extern "C" int real_hip_index_export(real_hip_ctx *ctx, int list, void *sign, uint32_t *pos)
{
    RH_ENTER(ctx);
    if (!ctx->have_index) return rh_fail(ctx, REAL_HIP_E_STATE, "no index", hipSuccess);
    if (list < 0 || list > 5) return rh_fail(ctx, REAL_HIP_E_INVALID, "list", hipSuccess);
    int rc = rh_index_export(ctx, list, sign, pos);
    rh_release(ctx->keys_a); rh_release(ctx->vals_a);
    return rc;
}

// ---------------------------------------------------------------------------
// batches
// ---------------------------------------------------------------------------
struct Staged {
    const uint8_t *bases = nullptr, *qual = nullptr;
    const uint64_t *off = nullptr;
    uint32_t upatl = 0, maxpatl = 0, W = 0;
    bool maxpatl_declared = false; // device offsets with the caller's bound: longer reads may exist (they get a wave each)
    uint32_t packed = 0;             // the matcher reads the 2-bit packed bases itself
    const uint8_t *nflags = nullptr;
};
// device buffers a host batch is copied into: the ctx's own (synchronous calls) or those of a slot (submit / wait)
struct StageBufs {
    DevBuf *bases, *qual, *off, *nflags;
};

// the batch struct of ABI version 1 ended behind max_patl
#define RH_BATCH_V1_SIZE 48u
static int batch_view(real_hip_ctx *ctx, const real_hip_batch *b, real_hip_batch &v)
{
    if (!b || (b->struct_size != sizeof(real_hip_batch) && b->struct_size != RH_BATCH_V1_SIZE))
        return rh_fail(ctx, REAL_HIP_E_INVALID, "batch struct_size", hipSuccess);
    memset(&v, 0, sizeof v);
    memcpy(&v, b, b->struct_size);
    return REAL_HIP_OK;
}

// Uploads (host batches; on `up`, which is ctx->stream for the synchronous calls and the copy stream for submitted ones).
// After it the arrays of `s` are valid for kernels on ctx->stream.
static int stage_batch(real_hip_ctx *ctx, const real_hip_batch &b, Staged &s, const StageBufs &sb, hipStream_t up, hipEvent_t up_done)
{
    if (!ctx->have_text || !ctx->have_index) return rh_fail(ctx, REAL_HIP_E_STATE, "text and index must be set", hipSuccess);
    if (b.n_reads > 0xffffffffull) return rh_fail(ctx, REAL_HIP_E_INVALID, "more than 2^32 reads in one batch", hipSuccess);
    const uint64_t n = b.n_reads;
    if (!n) return REAL_HIP_OK;
    if (!b.bases) return rh_fail(ctx, REAL_HIP_E_INVALID, "null bases", hipSuccess);
    if (b.nflags && !b.packed) return rh_fail(ctx, REAL_HIP_E_INVALID, "nflags belong to packed batches (unpacked ones carry symbol 4)", hipSuccess);
    int rc;
    uint64_t total = 0;
    if (b.offsets) {
        if (b.on_device) {
            s.off = b.offsets;
            s.maxpatl = b.max_patl;
            s.maxpatl_declared = b.max_patl != 0;
            if (!s.maxpatl && (rc = rh_max_patl(ctx, s.off, n, &s.maxpatl))) return rc;
        } else {
            for (uint64_t i = 0; i < n; ++i) {
                if (b.offsets[i + 1] < b.offsets[i]) return rh_fail(ctx, REAL_HIP_E_INVALID, "offsets not monotone", hipSuccess);
                uint64_t len = b.offsets[i + 1] - b.offsets[i];
                if (len > s.maxpatl) s.maxpatl = (uint32_t)(len > 0xffffffffull ? 0xffffffffull : len);
            }
            total = b.offsets[n];
            if ((rc = rh_reserve(ctx, *sb.off, (n + 1) * 8))) return rc;
            RH_HIP(ctx, hipMemcpyAsync(sb.off->p, b.offsets, (n + 1) * 8, hipMemcpyHostToDevice, up));
            s.off = (const uint64_t *)sb.off->p;
        }
    } else {
        s.upatl = b.patl; s.maxpatl = b.patl;
        total = n * (uint64_t)b.patl;
    }
    // Reads longer than the register budget of the lane-per-read kernels are matched by a wave each (match_wave.hip, from
    // LDS); beyond REAL_HIP_MAX_PATL_LONG nothing can: the reference has no such limit (RestWordBuffer grows), so that is
    // an explicit, loud error and not a skip.
    if (s.maxpatl > REAL_HIP_MAX_PATL_LONG) return rh_fail(ctx, REAL_HIP_E_UNSUPPORTED, "read longer than REAL_HIP_MAX_PATL_LONG", hipSuccess);
    const uint64_t base_bytes = b.packed ? (total + 3) / 4 : total;
    const uint8_t *d_bases = b.bases, *d_flags = b.nflags;
    if (b.on_device) {
        s.qual = b.qual;
    } else {
        if ((rc = rh_reserve(ctx, *sb.bases, (base_bytes ? base_bytes : 1) + 16))) return rc;
        RH_HIP(ctx, hipMemcpyAsync(sb.bases->p, b.bases, base_bytes, hipMemcpyHostToDevice, up));
        d_bases = (const uint8_t *)sb.bases->p;
        if (b.qual) {
            if ((rc = rh_reserve(ctx, *sb.qual, total ? total : 1))) return rc;
            RH_HIP(ctx, hipMemcpyAsync(sb.qual->p, b.qual, total, hipMemcpyHostToDevice, up));
            s.qual = (const uint8_t *)sb.qual->p;
        }
        if (b.nflags) {
            if ((rc = rh_reserve(ctx, *sb.nflags, (n + 7) / 8))) return rc;
            RH_HIP(ctx, hipMemcpyAsync(sb.nflags->p, b.nflags, (n + 7) / 8, hipMemcpyHostToDevice, up));
            d_flags = (const uint8_t *)sb.nflags->p;
        }
    }
    if (up != ctx->stream) { // the kernels wait for the upload, the host does not
        RH_HIP(ctx, hipEventRecord(up_done, up));
        RH_HIP(ctx, hipStreamWaitEvent(ctx->stream, up_done, 0));
    }
    if (b.packed) {
        // the matcher packs its words straight from the packed bytes (25 instead of 100 bytes of HBM per 100 bp read);
        // a read may start anywhere inside a byte
        s.bases = d_bases; s.packed = 1; s.nflags = d_flags;
    } else {
        s.bases = d_bases;
    }
    s.W = (s.maxpatl + 31) / 32;
    if (s.W < 1) s.W = 1;
    if (s.W > RH_MAXW) s.W = RH_MAXW; // (longer reads: handed over to the wave-per-read kernel)
    return REAL_HIP_OK;
}

static void fill_args(real_hip_ctx *ctx, const Staged &s, uint64_t n, MatchArgs &a)
{
    memset(&a, 0, sizeof a);
    a.t.text = (const uint64_t *)ctx->text.p; a.t.wild = (const uint64_t *)ctx->wild.p;
    a.t.frag_start = (const uint64_t *)ctx->frag.p; a.t.n = ctx->n_bases; a.t.n_frag = ctx->n_frag;
    a.t.has_wild = ctx->n_wild ? 1 : 0; a.t.fileid = ctx->fileid;
    const uint32_t l = ctx->prm.seedl, pb = ctx->pb;
    for (int k = 0; k < 6; ++k) { a.ix.ent[k] = (const uint2 *)ctx->ent[k].p; a.ix.bkt[k] = (const uint32_t *)ctx->bkt[k].p; }
    a.ix.n = ctx->n_entries; a.ix.pb = pb;
    rh_index_geometry(l, pb, &a.ix.pshift, &a.ix.fshift, &a.ix.fbits, &a.ix.pbits);
    a.ix.fine = (uint32_t)ctx->fine;
    a.b.bases = s.bases; a.b.qual = ctx->prm.scores ? s.qual : nullptr; a.b.off = s.off;
    a.b.n_reads = n; a.b.upatl = s.upatl; a.b.W = s.W; a.b.maxpatl = s.maxpatl_declared ? REAL_HIP_MAX_PATL_LONG : s.maxpatl;
    a.b.packed = s.packed; a.b.nflags = s.nflags;
    a.LL = (const double *)ctx->LL.p;
    a.counters = (unsigned long long *)ctx->counters.p;
    a.filter_mult = ctx->prm.filter_mult;
    a.l = l; a.q = l / 4; a.b_bits = 2 * (l / 4); a.seedkmax = ctx->prm.seedkmax; a.totalkmax = ctx->prm.totalkmax;
}

extern "C" int real_hip_match_unique(real_hip_ctx *ctx, const real_hip_batch *b, uint64_t *info, float *score)
{
    RH_ENTER(ctx);
    Staged s;
    real_hip_batch bv;
    int rc = batch_view(ctx, b, bv);
    if (rc) return rc;
    b = &bv;
    if ((rc = stage_batch(ctx, bv, s, StageBufs{&ctx->s_bases, &ctx->s_qual, &ctx->s_off, &ctx->s_nflags}, ctx->stream, nullptr))) return rc;
    const uint64_t n = b->n_reads;
    if (!n) return REAL_HIP_OK;
    const bool sc = ctx->prm.scores != 0;
    if (!info || (sc && !score)) return rh_fail(ctx, REAL_HIP_E_INVALID, "null info/score", hipSuccess);
    uint64_t *d_info = info;
    float *d_score = score;
    if (b->on_device != 1) { // outputs in host memory
        if ((rc = rh_reserve(ctx, ctx->s_info, n * 8))) return rc;
        if (!b->fresh) RH_HIP(ctx, hipMemcpyAsync(ctx->s_info.p, info, n * 8, hipMemcpyHostToDevice, ctx->stream));
        d_info = (uint64_t *)ctx->s_info.p;
        if (sc) {
            if ((rc = rh_reserve(ctx, ctx->s_score, n * 4))) return rc;
            if (!b->fresh) RH_HIP(ctx, hipMemcpyAsync(ctx->s_score.p, score, n * 4, hipMemcpyHostToDevice, ctx->stream));
            d_score = (float *)ctx->s_score.p;
        }
    }
    MatchArgs a;
    fill_args(ctx, s, n, a);
    a.info = d_info; a.score = d_score; a.b.fresh = b->fresh ? 1u : 0u;
    if ((rc = rh_launch_match(ctx, a, false, 2))) return rc;
    if (b->on_device != 1) { // outputs in host memory
        RH_HIP(ctx, hipMemcpyAsync(info, d_info, n * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (sc) RH_HIP(ctx, hipMemcpyAsync(score, d_score, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rh_time_resolve(ctx);
    return rh_match_finish(ctx, 2);
}

// ---------------------------------------------------------------------------
// pipelined host batches: submit / wait over two slots.  The upload of batch k+1 (copy stream) and the download of
// the records of batch k-1 (download stream) run beside the kernels of batch k (ctx stream); the counterpart of the
// reference's producer / consumer block ring (AsynchronousReader.hpp:181-259).
// ---------------------------------------------------------------------------
__global__ void fill_f32_kernel(float *p, uint64_t n, float v)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

static int pipeline_init(real_hip_ctx *ctx)
{
    if (ctx->copy_stream) return REAL_HIP_OK;
    RH_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    RH_HIP(ctx, hipStreamCreateWithFlags(&ctx->down_stream, hipStreamNonBlocking));
    for (int i = 0; i < REAL_HIP_SLOTS; ++i) {
        RH_HIP(ctx, hipEventCreateWithFlags(&ctx->slot[i].up, hipEventDisableTiming));
        RH_HIP(ctx, hipEventCreateWithFlags(&ctx->slot[i].matched, hipEventDisableTiming));
        RH_HIP(ctx, hipEventCreateWithFlags(&ctx->slot[i].done, hipEventDisableTiming));
    }
    return REAL_HIP_OK;
}

// everything of a submit that touches the streams; a failure half way leaves copies of the caller's memory in flight,
// which the wrapper below drains before it reports the error
static int submit_unique(real_hip_ctx *ctx, RhSlot &S, int slot, const real_hip_batch &bv, uint64_t *info, float *score, int fresh)
{
    const uint64_t n = bv.n_reads;
    const bool sc = ctx->prm.scores != 0;
    int rc;
    // records: uploaded (they are in/out: folds compose across genome blocks), or initialised on the device (fresh:
    // uniqueinfo(numpat), matchUniqueImplementation.cpp:1094-1097 -- NoMatch, score -FLT_MAX)
    if ((rc = rh_reserve(ctx, S.info, n * 8))) return rc;
    if (sc && (rc = rh_reserve(ctx, S.score, n * 4))) return rc;
    if (!fresh) { // (fresh: the kernel starts every record itself)
        RH_HIP(ctx, hipMemcpyAsync(S.info.p, info, n * 8, hipMemcpyHostToDevice, ctx->copy_stream));
        if (sc) RH_HIP(ctx, hipMemcpyAsync(S.score.p, score, n * 4, hipMemcpyHostToDevice, ctx->copy_stream));
    }
    Staged s;
    if ((rc = stage_batch(ctx, bv, s, StageBufs{&S.bases, &S.qual, &S.off, &S.nflags}, ctx->copy_stream, S.up))) return rc;
    MatchArgs a;
    fill_args(ctx, s, n, a);
    a.info = (uint64_t *)S.info.p; a.score = (float *)S.score.p; a.b.fresh = (fresh || bv.fresh) ? 1u : 0u;
    if ((rc = rh_launch_match(ctx, a, false, slot))) return rc;
    RH_HIP(ctx, hipEventRecord(S.matched, ctx->stream));
    RH_HIP(ctx, hipStreamWaitEvent(ctx->down_stream, S.matched, 0));
    RH_HIP(ctx, hipMemcpyAsync(info, S.info.p, n * 8, hipMemcpyDeviceToHost, ctx->down_stream));
    if (sc) RH_HIP(ctx, hipMemcpyAsync(score, S.score.p, n * 4, hipMemcpyDeviceToHost, ctx->down_stream));
    RH_HIP(ctx, hipEventRecord(S.done, ctx->down_stream));
    return REAL_HIP_OK;
}

extern "C" int real_hip_match_unique_submit(real_hip_ctx *ctx, const real_hip_batch *b, uint64_t *info, float *score, uint32_t slot, int fresh)
{
    RH_ENTER(ctx);
    if (slot >= REAL_HIP_SLOTS) return rh_fail(ctx, REAL_HIP_E_INVALID, "slot", hipSuccess);
    RhSlot &S = ctx->slot[slot];
    if (S.busy) return rh_fail(ctx, REAL_HIP_E_STATE, "slot in flight: real_hip_wait first", hipSuccess);
    real_hip_batch bv;
    int rc = batch_view(ctx, b, bv);
    if (rc) return rc;
    if (bv.on_device) return rh_fail(ctx, REAL_HIP_E_INVALID, "submit takes host batches (device batches: the synchronous calls)", hipSuccess);
    const uint64_t n = bv.n_reads;
    if (n && (!info || (ctx->prm.scores && !score))) return rh_fail(ctx, REAL_HIP_E_INVALID, "null info/score", hipSuccess);
    if ((rc = pipeline_init(ctx))) return rc;
    S.n = n; S.status = REAL_HIP_OK;
    if (!n) { S.busy = true; S.empty = true; return REAL_HIP_OK; }
    S.empty = false;
    if ((rc = submit_unique(ctx, S, (int)slot, bv, info, score, fresh))) {
        // nothing of the caller's memory may stay in flight behind an error
        const std::string msg = ctx->last_error;
        (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamSynchronize(ctx->stream); (void)hipStreamSynchronize(ctx->down_stream);
        (void)hipGetLastError();
        ctx->last_error = msg;
        return rc;
    }
    S.busy = true;
    return REAL_HIP_OK;
}

extern "C" int real_hip_wait(real_hip_ctx *ctx, uint32_t slot)
{
    RH_ENTER(ctx);
    if (slot >= REAL_HIP_SLOTS) return rh_fail(ctx, REAL_HIP_E_INVALID, "slot", hipSuccess);
    RhSlot &S = ctx->slot[slot];
    if (!S.busy) return REAL_HIP_OK;
    S.busy = false;
    if (S.empty) return REAL_HIP_OK;
    RH_HIP(ctx, hipEventSynchronize(S.done));
    rh_time_resolve(ctx);
    return rh_match_finish(ctx, (int)slot);
}

extern "C" void *real_hip_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
extern "C" void real_hip_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

extern "C" int real_hip_match_all(real_hip_ctx *ctx, const real_hip_batch *b, real_hip_hit *out, uint64_t cap,
                                  uint64_t *n_out, uint64_t *hit_offsets)
{
    RH_ENTER(ctx);
    Staged s;
    real_hip_batch bv;
    int rc = batch_view(ctx, b, bv);
    if (rc) return rc;
    b = &bv;
    if ((rc = stage_batch(ctx, bv, s, StageBufs{&ctx->s_bases, &ctx->s_qual, &ctx->s_off, &ctx->s_nflags}, ctx->stream, nullptr))) return rc;
    const uint64_t n = b->n_reads;
    if (n_out) *n_out = 0;
    if (cap > 0xffffffffull) cap = 0xffffffffull; // record indices are 32 bit inside the post-pass
    if ((rc = rh_reserve(ctx, ctx->raw_count, 8))) return rc;
    RH_HIP(ctx, hipMemsetAsync(ctx->raw_count.p, 0, 8, ctx->stream));
    if ((rc = rh_reserve(ctx, ctx->raw, (cap ? cap : 1) * sizeof(uint4)))) return rc;
    unsigned long long n_raw = 0;
    if (n) {
        MatchArgs a;
        fill_args(ctx, s, n, a);
        if ((rc = rh_reserve(ctx, ctx->hit_cnt, (n + 1) * 4))) return rc;
        a.raw = (uint4 *)ctx->raw.p; a.raw_count = (unsigned long long *)ctx->raw_count.p; a.raw_cap = cap;
        a.hit_cnt = (uint32_t *)ctx->hit_cnt.p;
        if ((rc = rh_launch_match(ctx, a, true, 2))) return rc;
        RH_HIP(ctx, hipMemcpyAsync(&n_raw, ctx->raw_count.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if ((rc = rh_match_finish(ctx, 2))) return rc;
    }
    if (n_out) *n_out = n_raw;
    if (n_raw > cap) return rh_fail(ctx, REAL_HIP_E_OVERFLOW, "hit buffer too small", hipSuccess);
    real_hip_hit *d_out = out;
    uint64_t *d_off = hit_offsets;
    if (b->on_device != 1) { // outputs in host memory
        if ((rc = rh_reserve(ctx, ctx->s_hits, (n_raw ? n_raw : 1) * sizeof(real_hip_hit)))) return rc;
        d_out = (real_hip_hit *)ctx->s_hits.p;
        if (hit_offsets) {
            if ((rc = rh_reserve(ctx, ctx->hit_off, (n + 1) * 8))) return rc;
            d_off = (uint64_t *)ctx->hit_off.p;
        }
    }
    if ((n_raw && !out) ) return rh_fail(ctx, REAL_HIP_E_INVALID, "null hit buffer", hipSuccess);
    if ((rc = rh_all_finish(ctx, n_raw, n, d_out, d_off))) return rc;
    if (b->on_device != 1) { // outputs in host memory
        if (n_raw) RH_HIP(ctx, hipMemcpyAsync(out, d_out, n_raw * sizeof(real_hip_hit), hipMemcpyDeviceToHost, ctx->stream));
        if (hit_offsets) RH_HIP(ctx, hipMemcpyAsync(hit_offsets, d_off, (n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rh_time_resolve(ctx);
    return REAL_HIP_OK;
}

// ---------------------------------------------------------------------------
// read ingestion
// ---------------------------------------------------------------------------
extern "C" int real_hip_parse_reads(real_hip_ctx *ctx, const char *text, uint64_t n_bytes, int text_on_device, int fastq,
                                    int quality_offset, real_hip_parsed *out)
{
    RH_ENTER(ctx);
    if (!out || (n_bytes && !text)) return rh_fail(ctx, REAL_HIP_E_INVALID, "null text / out", hipSuccess);
    const char *d_text = text;
    if (!text_on_device && n_bytes) {
        int rc = rh_reserve(ctx, ctx->p_text, n_bytes);
        if (rc) return rc;
        RH_HIP(ctx, hipMemcpyAsync(ctx->p_text.p, text, n_bytes, hipMemcpyHostToDevice, ctx->stream));
        d_text = (const char *)ctx->p_text.p;
    }
    RhTimer tm(ctx, REAL_HIP_K_PARSE);
    return rh_parse_reads(ctx, d_text, n_bytes, fastq, quality_offset, out);
}

extern "C" int real_hip_download(real_hip_ctx *ctx, const void *device_ptr, void *host_ptr, size_t bytes)
{
    RH_ENTER(ctx);
    if (bytes && (!device_ptr || !host_ptr)) return rh_fail(ctx, REAL_HIP_E_INVALID, "null pointer", hipSuccess);
    if (bytes) RH_HIP(ctx, hipMemcpyAsync(host_ptr, device_ptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return REAL_HIP_OK;
}

// ---------------------------------------------------------------------------
// counters / timing
// ---------------------------------------------------------------------------
extern "C" int real_hip_counters_get(real_hip_ctx *ctx, real_hip_counters *out, int reset)
{
    RH_ENTER(ctx);
    std::vector<uint64_t> all((size_t)RH_CSTRIPES * 16);
    RH_HIP(ctx, hipMemcpyAsync(all.data(), ctx->counters.p, all.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (reset) RH_HIP(ctx, hipMemsetAsync(ctx->counters.p, 0, all.size() * 8, ctx->stream));
    RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t st = 0; st < RH_CSTRIPES; ++st)
        for (int k = 0; k < 8; ++k) h[k] += all[st * 16 + k];
    if (out) {
        out->reads = h[0]; out->lookups = h[1]; out->probes = h[2]; out->candidates = h[3];
        out->seedpass = h[4]; out->hits = h[5]; out->verified = h[6]; out->handed_over = h[7];
    }
    return REAL_HIP_OK;
}

extern "C" int real_hip_kernel_time(real_hip_ctx *ctx, int which, double *total_ms, uint64_t *launches, int reset)
{
    if (!ctx || which < 0 || which >= REAL_HIP_K_COUNT) return REAL_HIP_E_INVALID;
    if (total_ms) *total_ms = ctx->k_ms[which];
    if (launches) *launches = ctx->k_n[which];
    if (reset) { ctx->k_ms[which] = 0; ctx->k_n[which] = 0; }
    return REAL_HIP_OK;
}

extern "C" int real_hip_timing_enable(real_hip_ctx *ctx, int on)
{
    if (!ctx) return REAL_HIP_E_INVALID;
    ctx->timing = on != 0;
    return REAL_HIP_OK;
}
