// real_hip_internal.h -- shared between the translation units of libreal_hip.so.
// gfx950 only; no portability layers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "real_hip.h"

#define RH_MAXW 10 /* 64-bit words per oriented read: REAL_HIP_MAX_PATL / 32 */
static_assert(RH_MAXW * 32 == REAL_HIP_MAX_PATL, "one lane-per-read instance per 32 bases up to REAL_HIP_MAX_PATL");
// Work counters are striped over this many 128-byte lines: every wave ends with a handful of atomics,
// and 781 k waves hammering ONE line serialise at the L2 channel that owns it (measured: the whole
// kernel then runs at the atomic rate, independent of the genome size).
#define RH_CSTRIPES 1024

// ---- device-side views ------------------------------------------------------
struct DevText {
    const uint64_t *text;       // 2-bit text, MSB first, padded by 4 zero words
    const uint64_t *wild;       // N bit vector, MSB first, padded
    const uint64_t *frag_start; // n_frag + 1
    uint64_t n;                 // bases
    uint32_t n_frag;
    uint32_t has_wild;
    uint32_t fileid;
};

// One entry of a sorted list: {fingerprint, window start}.  The fingerprint is
// 32 bits of the signature just below the bucket prefix; entries of a bucket
// are in the reference's list order (signature ascending, stable => position
// ascending), so equal fingerprints are contiguous and the members of the
// reference's equal range appear in the reference's order.
struct DevIndex {
    const uint2    *ent[6];
    const uint32_t *bkt[6]; // 2^pb + 1 bucket starts (u32), or in fine mode uint4 {start, 8 x {size:4, partner digest:8}}
    uint64_t n;
    uint32_t pb;     // prefix bits
    uint32_t pshift; // sig_bits - pb: prefix = sign >> pshift
    uint32_t fshift; // entry.x = [fbits of (sign >> fshift)] [pbits of the partner signature's top bits]
    uint32_t fbits;  // signature bits kept in the entry (all sig_bits - pb of them when that is <= 30)
    uint32_t pbits;  // partner-signature bits kept in the entry (even; 0 when the signature needs all 32)
    uint32_t fine;   // bucket table kind: 0 u32 starts; 1 "fine": size + partner digest of every key group of the bucket
                     // (fbits <= 3); 2 fingerprints of the bucket's first entries (pbits == 0, wide signatures);
                     // 3 bucket rows: bkt[] holds one 128-byte row per bucket (directory + entries), ent[] the overflow
};

// "fine" bucket tables: the prefix is all signature bits but one to three, so a bucket has at most eight key
// groups (= signature values) and its 16-byte table entry describes each of them
static inline bool rh_is_fine(uint32_t l, uint32_t pb) { return l >= pb && l - pb >= 1 && l - pb <= 3; }
#define RH_FINE_SAT 15u /* group size field: 15 = "15 or more", bounds by binary search */
// fingerprint tables: uint4 {start, count:8 | 8 x fingerprint:11}: the first eight entries of the bucket by an
// 11-bit hash of their 32-bit key; a lookup whose own fingerprint is not among them reads no entry at all
#define RH_ROW_CAP 20u /* entries a bucket row holds (table kind 3) */
#define RH_FP_SLOTS 8u
static inline __host__ __device__ uint32_t rh_fp11(uint32_t key) { return (key * 0x9E3779B1u) >> 21; }
// bucket rows of wide signatures (entries hold a 32-bit key): key group = the key's leading four bits, the row keeps a
// 16-bit fingerprint of the other 28
static inline __host__ __device__ uint32_t rh_fp16(uint32_t key) { return ((key & 0x0fffffffu) * 0x9E3779B1u) >> 16; }

// Bucket rows are addressed through a bijection of the signature space: mixed = sign * odd constant mod 2^seedl.  The row
// of a signature is the leading bits of the MIXED value, its key group the bits below.  A genome's signatures are far from
// uniform when its base composition is skewed (60 % A+T, the human genome's share: the 16 signature values that share a
// 14-base prefix rich in A/T hold ten times the mean, a third of all entries sits in rows that overflow); the leading bits
// of the product depend on every bit of the signature, so a row collects 16 unrelated signature values and the loads
// concentrate around the mean (2.7 % of the entries in overflowing rows at 60 % A+T).  Being a
// bijection it keeps the layout exact: (row, key group) <-> one signature value, the entries of a group are the
// reference's equal range in its order (the sort is stable: ascending position inside equal signatures; the order of
// DIFFERENT signatures in the device list is the mixed one, real_hip_index_export sorts it back).
#define RH_MIX32 0x9E3779B1u
#define RH_MIX64 0x9E3779B97F4A7C15ull
static inline __host__ __device__ uint32_t rh_mix32(uint32_t sign, uint32_t l) { return (sign * RH_MIX32) & (l >= 32 ? 0xffffffffu : ((1u << l) - 1u)); }
static inline __host__ __device__ uint64_t rh_mix64(uint64_t sign, uint32_t l) { return (sign * RH_MIX64) & (l >= 64 ? ~0ull : ((1ull << l) - 1ull)); }

// entry geometry shared by the index build and the matcher
static inline void rh_index_geometry(uint32_t l, uint32_t pb, uint32_t *pshift, uint32_t *fshift, uint32_t *fbits, uint32_t *pbits)
{
    const uint32_t R = l - pb; // signature bits below the bucket prefix
    *pshift = R;
    if (R <= 30) {
        uint32_t pbts = (32 - R) & ~1u;
        if (pbts > l) pbts = l;
        if (pbts > 30) pbts = 30;
        *fshift = 0; *fbits = R; *pbits = pbts;
    } else {
        *fshift = R - 32; *fbits = 32; *pbits = 0;
    }
}

// a batch of reads as the caller holds it (real_hip_batch): the matcher reads these arrays itself
struct DevBatch {
    const uint8_t  *bases; // concatenated mapped symbols 0..4
    const uint8_t  *qual;  // concatenated qualities 0..63, nullptr => 30
    const uint64_t *off;   // n_reads + 1 start offsets, nullptr => uniform length upatl
    uint64_t n_reads;
    uint32_t upatl;
    uint32_t W;            // 64-bit words per oriented read = ceil(max read length / 32)
    uint32_t gl;           // reads a wave stages through its LDS region at a time (power of two <= 64)
    // 2-bit packed bases: base g of the batch at bits 7-2(g%4)-1.. of byte g/4, MSB first -- the layout of a word of the
    // oriented read; a read may start inside a byte.  nflags (nullable): bit r%8 of byte r/8 set => read r holds a symbol
    // > 3 and is skipped
    uint32_t packed;
    const uint8_t *nflags;
    uint32_t fresh;        // matchUnique: the records are not read, every read of the batch starts as NoMatch / -FLT_MAX
    uint32_t maxpatl;      // the longest read of the batch (declared or measured): beyond 32 * RH_MAXW the wave matcher keeps room for long reads
};

struct MatchArgs {
    DevText  t;
    DevIndex ix;
    DevBatch b;
    const double *LL;      // device copy of the 4x4x64 table
    uint64_t *info;        // in/out records
    float    *score;       // in/out scores (scores mode)
    unsigned long long *counters; // RH_CSTRIPES stripes x 16 u64 (one 128-B line per stripe), summed on the host
    // matchAll
    uint4 *raw;            // raw hit records
    unsigned long long *raw_count;
    uint64_t raw_cap;
    uint32_t *hit_cnt;     // [n_reads] hits appended per read (every read of the batch is written)
    // reads handed from the matcher to the repeat kernel (scores on)
    uint32_t *ovf_list;    // first pass -> second pass (bucket rows, parked hits): reads that need more room than a lane of the first has
    unsigned long long *ovf_count;
    uint32_t *ovf2_list;   // -> wave-per-read kernel: long equal ranges, long reads, what the second pass could not hold either
    unsigned long long *ovf2_count;
    uint32_t *tile_ctr2;   // the second pass' tile counter
    uint32_t *err_flags;   // bit 0: a read longer than the declared bound / offsets not monotone (nothing was staged for it)
    uint32_t *tile_ctr;    // the next tile of 64 reads to be handed out (match_kernel: the waves of a resident grid take tiles as they get done)
    double   filter_mult;
    uint32_t l, q, b_bits, seedkmax, totalkmax;
};

// ---- host context ------------------------------------------------------------
struct DevBuf {
    void  *p = nullptr;
    size_t cap = 0;
};

// one slot of the submit / wait pipeline: device copies of a host batch and its records
struct RhSlot {
    DevBuf bases, qual, off, nflags, info, score;
    hipEvent_t up = nullptr, matched = nullptr, done = nullptr;
    uint64_t n = 0;
    int status = 0;
    bool busy = false, empty = false;
};

struct real_hip_ctx {
    real_hip_params prm;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string last_error;

    // text
    DevBuf text, wild, frag;
    uint64_t n_bases = 0;
    uint64_t n_wild = 0;
    uint32_t n_frag = 0, fileid = 0;
    bool have_text = false;

    // index
    DevBuf ent[6], bkt[6];
    uint64_t n_entries = 0;
    uint32_t pb = 0;
    int fine = 0;      // bucket table kind, see DevIndex::fine
    bool no_rows = false; // bucket rows did not fit the device memory once: stay with directory tables
    bool have_index = false;

    // tables / counters
    DevBuf LL, counters;

    // batch staging (host batches), hand-over list of the repeat kernel
    DevBuf s_bases, s_qual, s_off, s_info, s_score, s_nflags;
    RhSlot slot[REAL_HIP_SLOTS];                    // submit / wait
    hipStream_t copy_stream = nullptr, down_stream = nullptr;
    DevBuf maxpatl, ovf_list, ovf2_list, ovf_count;
    unsigned long long *h_state = nullptr; // pinned: {reads handed over, error flags} of the last launch of slot 0, slot 1, the synchronous calls
    // read ingestion (read_parse.hip)
    DevBuf p_text, p_nl, p_scal, p_spans, p_off, p_len1, p_bases, p_qual;
    // matchAll workspace
    DevBuf raw, raw_count, hit_cnt, big_list, all_cursor, keys_a, keys_b, vals_a, vals_b, sort_tmp, hit_off, s_hits;

    // where the wall time of an index build goes (real_hip_index_build_stats)
    double   alloc_ms = 0, free_ms = 0, build_wall_ms = 0;
    uint64_t alloc_bytes = 0, alloc_calls = 0, free_calls = 0;

    // multi-GPU (gather.hip): an RCCL communicator over the ranks' devices, one process per GPU
    void *comm = nullptr;
    int comm_rank = 0, comm_size = 1;
    DevBuf comm_counts, comm_scratch;

    // timing
    bool timing = true;
    double   k_ms[REAL_HIP_K_COUNT] = {};
    uint64_t k_n[REAL_HIP_K_COUNT] = {};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // asynchronous timing of pipelined launches: event pairs resolved at the next synchronisation point
    struct Pending { hipEvent_t a, b; int which; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> ev_pool;
};

// error plumbing: never throw across the ABI
int rh_fail(real_hip_ctx *ctx, int status, const char *what, hipError_t e);
#define RH_HIP(ctx, call)                                                          \
    do {                                                                           \
        hipError_t _e = (call);                                                    \
        if (_e != hipSuccess) return rh_fail((ctx), REAL_HIP_E_DEVICE, #call, _e); \
    } while (0)
int rh_reserve(real_hip_ctx *ctx, DevBuf &b, size_t bytes);
void rh_release(DevBuf &b);
void rh_release(real_hip_ctx *ctx, DevBuf &b); // (timed: real_hip_index_build_stats)
double rh_now_ms();
// a function-local device buffer: released on every exit path
struct ScopedBuf : DevBuf {
    real_hip_ctx *c;
    explicit ScopedBuf(real_hip_ctx *ctx) : c(ctx) {}
    ~ScopedBuf() { rh_release(c, *this); }
    ScopedBuf(const ScopedBuf &) = delete;
    ScopedBuf &operator=(const ScopedBuf &) = delete;
};

struct RhTimer { // HIP events on the ctx stream around a group of launches
    real_hip_ctx *c;
    int which;
    RhTimer(real_hip_ctx *ctx, int w);
    ~RhTimer();
};

// ---- kernels launchers (match_kernels.hip) -----------------------------------
void rh_comm_destroy(real_hip_ctx *ctx);
int rh_launch_match(real_hip_ctx *ctx, const MatchArgs &a, bool all, int state_slot);
int rh_match_finish(real_hip_ctx *ctx, int state_slot); // after the launch has completed: errors the kernels flagged

// asynchronous kernel timing (no host synchronisation at the launch site)
void rh_time_begin(real_hip_ctx *ctx, hipStream_t st, int which);
void rh_time_end(real_hip_ctx *ctx, hipStream_t st);
void rh_time_resolve(real_hip_ctx *ctx); // call after the streams were synchronised
int rh_max_patl(real_hip_ctx *ctx, const uint64_t *d_off, uint64_t n_reads, uint32_t *out);
int rh_parse_reads(real_hip_ctx *ctx, const char *d_text, uint64_t n_bytes, int fastq, int qoff, real_hip_parsed *out);
int rh_all_finish(real_hip_ctx *ctx, uint64_t n_raw, uint64_t n_reads, real_hip_hit *d_out,
                  uint64_t *d_hit_offsets);

// ---- text + index (index_build.hip) ------------------------------------------
int rh_pack_text(real_hip_ctx *ctx, const uint8_t *d_sym, uint64_t n);
int rh_index_from_host_lists(real_hip_ctx *ctx, uint64_t n, const void *const sign[6], const uint32_t *const pos[6], unsigned sig_bytes);
int rh_index_build_device(real_hip_ctx *ctx, uint64_t first_window, uint64_t max_entries,
                          uint64_t *n_entries, int *have_next);
void rh_choose_tables(real_hip_ctx *ctx, uint64_t n_entries); // sets ctx->pb and ctx->fine
int rh_rows_unpack(real_hip_ctx *ctx, int list, uint2 *d_entries, uint32_t *d_starts);
