/*
 * real_hip.h -- C ABI of the MI355X (gfx950) read-matching hot path of REAL.
 *
 * The reference (solonas13/REAL) has no plugin / FFI interface: the hot path
 * sits behind C++ template seams inside one binary.  This
 * header is the drop-in boundary a maintainer would bind instead of the
 * per-read loops
 *
 *     for z in block: UM.match(pattern, uniqueinfo[patid], fi, RWB, handled)
 *                                   matchUniqueImplementation.cpp:1268-1295
 *     for z in block: AM.match(pattern, fi, RWB, handled, localmatches);
 *                     unifyMatches(localmatches)
 *                                   matchAllImplementation.cpp:459-533
 *
 * Plain pointers and sizes only; no C++ / torch types; nothing throws across
 * the boundary: every entry point returns 0 or a negative real_hip_status.
 * Each entry point cites the reference interface it replaces.
 *
 * Threading: one submitting thread per ctx (calls on one ctx are not
 * re-entrant); distinct ctx (one per GPU) are independent.
 *
 * Streams: every ctx owns one non-blocking HIP stream and runs all its copies
 * and kernels there; it knows nothing about the caller's streams.  Device
 * memory handed to a call (on_device / text_on_device / sym_on_device inputs
 * and the in/out records of on_device = 1) must therefore be COMPLETE before
 * the call -- the producing stream synchronised, or an event of it waited for
 * with real_hip_wait_event() -- and must not be touched until the call (or,
 * for the _submit forms, the matching real_hip_wait) has returned.  Results
 * are complete when the synchronous entry points return.
 */
#ifndef REAL_HIP_H
#define REAL_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REAL_HIP_ABI_VERSION 2

typedef struct real_hip_ctx real_hip_ctx;

typedef enum real_hip_status {
    REAL_HIP_OK            =  0,
    REAL_HIP_E_INVALID     = -1,  /* bad argument / struct_size                       */
    REAL_HIP_E_NOMEM       = -2,  /* host or device allocation failed (std::bad_alloc
                                     in the reference, matchUniqueImplementation.cpp:1215-1219) */
    REAL_HIP_E_DEVICE      = -3,  /* HIP runtime error (see real_hip_last_error)      */
    REAL_HIP_E_OVERFLOW    = -4,  /* output capacity too small; *n_out = needed size  */
    REAL_HIP_E_STATE       = -5,  /* text / index not set                             */
    REAL_HIP_E_UNSUPPORTED = -6   /* e.g. read longer than REAL_HIP_MAX_PATL_LONG     */
} real_hip_status;

#define REAL_HIP_MAX_PATL 320u        /* longest read the lane-per-read kernels hold in registers                */
#define REAL_HIP_MAX_PATL_LONG 16384u /* longest read at all: reads beyond REAL_HIP_MAX_PATL get a wave each and are
                                         read from LDS (slower, and only those reads); longer ones are refused with
                                         REAL_HIP_E_UNSUPPORTED -- the reference has no limit (RestMatch.hpp:34-37)  */

/* ---- parameters: what RealOptions (RealOptions.hpp:27-77) hands the matcher */
typedef struct real_hip_params {
    uint32_t struct_size;   /* = sizeof(real_hip_params)                              */
    uint32_t seedl;         /* -l : 4..64, multiple of 4 (RealOptions.cpp:434-447)    */
    uint32_t seedkmax;      /* -s : <= 2 (RealOptions.cpp:449-453)                    */
    uint32_t totalkmax;     /* -e : <= 15 (RealOptions.cpp:172-180)                   */
    uint32_t scores;        /* -q : 0/1                                               */
    uint32_t prefix_bits;   /* device bucket-table width; 0 = auto from index size.
                               (The reference's 22-bit getSampleBits() table is a
                               host-layout detail; results do not depend on it.)      */
    int32_t  device;        /* HIP device ordinal                                     */
    uint32_t table_kind;    /* device bucket tables: 0 = auto from index size and device memory, 1 = bucket starts only,
                               2 = directory entries (group sizes + partner digests for seedl <= 32,
                               key fingerprints for wider signatures), 3 = bucket rows (seedl <= 32: one 128-byte
                               row per bucket holds directory and entries, lookups by groups of eight
                               lanes); a host-layout detail like
                               prefix_bits, results do not depend on it                   */
    double   filter_mult;   /* RealOptions.cpp:455-463; epsilon=(float)(filter_mult*patl),
                               RealOptions.hpp:74-77, matchUniqueImplementation.cpp:405 */
    double   LL[1024];      /* Scoring::getRawLogScoreTable, index (ref<<8)|(read<<6)|q
                               (Scoring.hpp:70-73); ignored if !scores                */
} real_hip_params;

/* Scoring::init + getScore (Scoring.cpp:61-133,155-171): fills LL for the
 * -similarity -gc -trans -err -gcmut_bias flags; host-side helper.            */
void real_hip_scoring_table(double similarity, double gc, double trans, double err,
                            double gcmut_bias, double LL[1024]);

int  real_hip_create(real_hip_ctx **out, const real_hip_params *p);
void real_hip_destroy(real_hip_ctx *ctx);
/* make the ctx's stream wait for a hipEvent_t the caller recorded on a stream of his own (the asynchronous way to
 * satisfy the "Streams" contract above); returns at once                                                       */
int  real_hip_wait_event(real_hip_ctx *ctx, void *hip_event);
/* -s / -e / -q / -filter_level for the calls that follow, with the resident text and index kept (they depend on
 * -l only): what constructing another UniqueMatcher / AllMatcher over the same ListSet is to the reference
 * (matchUniqueImplementation.cpp:348-367, matchAllImplementation.cpp:240-259).                                  */
int  real_hip_set_match_params(real_hip_ctx *ctx, uint32_t seedkmax, uint32_t totalkmax, uint32_t scores, double filter_mult);
const char *real_hip_strerror(int status);
const char *real_hip_last_error(const real_hip_ctx *ctx);
int  real_hip_abi_version(void);

/* free / total HBM of the ctx's device in bytes: what getPhysicalMemory() * -f is to the
 * reference's block sizing (matchUniqueImplementation.cpp:1208-1244)           */
int real_hip_device_memory(real_hip_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);

/* ---- genome text: replaces what getText<sse4>() + RangeVector hand the
 * matcher (getText.hpp:31-55, AutoTextArray.hpp:63-109, RangeVector.hpp:46-58).
 * text2bit: 2 bits/base, base i at bits 63-2(i%32)-1.. of word i/32 (MSB
 * first, N stored as 0); wildbits: bit i (MSB first) set iff base i is N;
 * frag_start[n_frag] = n_bases ("terminal", countReads.cpp:81).  Host pointers;
 * copied to HBM.                                                              */
int real_hip_set_text(real_hip_ctx *ctx, uint32_t fileid,
                      const uint64_t *text2bit, const uint64_t *wildbits, uint64_t n_bases,
                      const uint64_t *frag_start, uint32_t n_frag);
/* same from mapped symbols 0..4 (countReads.cpp:83-125 readFile); packed on
 * the device.  sym_on_device != 0: sym is a device pointer.                   */
int real_hip_set_text_symbols(real_hip_ctx *ctx, uint32_t fileid,
                              const uint8_t *sym, uint64_t n_bases, int sym_on_device,
                              const uint64_t *frag_start, uint32_t n_frag);

/* ---- genome index block: what ListSetBlockReader::readNextBlock() exposes
 * (ListSetBlockReader.hpp:24-52, ListSet.hpp:23-31).
 * Host-built form: six lists in the host's sorted order (stable => ascending
 * position inside equal signatures); sign[k] has n_entries elements of
 * sig_bytes (4 if seedl <= 32 else 8, real.cpp:219-229); pos[k][j] = window
 * start of entry j (Mask::getPos, Mask.hpp:36-40,55-59).                      */
int real_hip_set_index_block(real_hip_ctx *ctx, uint64_t n_entries,
                             const void *const sign[6], const uint32_t *const pos[6]);
/* Device-built form: enumerates the N-free windows
 * [first_window, first_window+max_entries) of the resident text
 * (MapTextFile.hpp:118-230), sorts the six lists on the GPU.  Same device
 * arrays as the host-built form.                                              */
int real_hip_build_index_block(real_hip_ctx *ctx, uint64_t first_window, uint64_t max_entries,
                               uint64_t *n_entries, int *have_next);
/* where the wall time of the index builds of this ctx went, accumulated since the last reset: the kernels of the
 * build (HIP events), hipMalloc and hipFree (host clock; the driver maps and clears every page), the rest is
 * synchronisation and host code.  The counterpart of the reference's "Sorting fragments..." clock
 * (ListSetBlockReader.hpp:42-48).                                                                             */
typedef struct real_hip_build_stats {
    uint32_t struct_size, reserved;
    double   wall_ms, kernel_ms, alloc_ms, free_ms;
    uint64_t alloc_bytes, alloc_calls, free_calls;
} real_hip_build_stats;
int real_hip_index_build_stats(real_hip_ctx *ctx, real_hip_build_stats *out, int reset);
/* introspection (tests, CPU baseline): device layout of list k               */
int real_hip_index_info(const real_hip_ctx *ctx, uint64_t *n_entries, uint32_t *prefix_bits);
/* kind of the resident bucket tables: 0 bucket starts, 1 group sizes + partner digests, 2 key fingerprints, 3 bucket rows */
int real_hip_index_table_kind(const real_hip_ctx *ctx, uint32_t *kind);
int real_hip_index_download(real_hip_ctx *ctx, int list,
                            uint32_t *entries      /* n_entries x {key, pos} (raw device layout), nullable */,
                            uint32_t *bucket_start /* 2^prefix_bits + 1, nullable                        */);
/* list k in the reference's own form: sign[j] (sig_bytes each) and pos[j] of
 * the sorted Mask entries (Mask.hpp:22-64); either pointer may be NULL.       */
int real_hip_index_export(real_hip_ctx *ctx, int list, void *sign, uint32_t *pos);

/* ---- read batch: a decoded pattern block (PatternBlock / FastSubDecoder::
 * fillPatternBlock, FastSubDecoder.hpp:107-161): mapped symbols A,C,G,T->0..3,
 * other->4 (acgtnMap.hpp:39-50) and quality = ASCII - offset in 0..63.        */
typedef struct real_hip_batch {
    uint32_t        struct_size;
    uint32_t        on_device;  /* 0: all pointers of the call are host memory (copied);
                                   1: all are device pointers (bases, qual, offsets and the
                                      info / score / hit outputs);
                                   2: bases, qual, offsets are device pointers (e.g. the arrays
                                      real_hip_parse_reads returned), the outputs host memory   */
    uint64_t        n_reads;
    const uint8_t  *bases;      /* concatenated mapped symbols                          */
    const uint8_t  *qual;       /* concatenated qualities; NULL => 30 (Pattern.hpp:42-45) */
    const uint64_t *offsets;    /* n_reads+1 start offsets; NULL => uniform length patl  */
    uint32_t        patl;       /* uniform read length if offsets == NULL               */
    uint32_t        max_patl;   /* upper bound of read length when offsets != NULL and
                                   on_device (0: library computes it); a read that turns out
                                   longer fails the call with REAL_HIP_E_INVALID          */
    /* -- since ABI version 2 (struct_size tells; a version-1 struct of 48 bytes is still accepted) -- */
    uint32_t        packed;     /* 1: bases holds 2 bits per base instead of a byte: base g of the
                                   concatenated batch at bits 7-2(g%4)-1.. of byte g/4 (MSB first, the
                                   packing of TemporaryFile.hpp:335-373); offsets / patl still count bases */
    uint32_t        fresh;      /* matchUnique: 1 = the records start as uniqueinfo(numpat) does (NoMatch, score -FLT_MAX,
                                   matchUniqueImplementation.cpp:1094-1097, UniqueMatchInfo.hpp:191) and info / score are
                                   outputs only -- the first genome block of a run; 0 = in/out, folds compose          */
    const uint8_t  *nflags;     /* packed only, nullable: bit (i%8) of byte i/8 set => read i holds a symbol
                                   > 3 (it cannot be packed) and is skipped as the reference skips it
                                   (matchUniqueImplementation.cpp:376-394)                              */
} real_hip_batch;

/* matchUnique: replaces the loop over UniqueMatcher::match
 * (matchUniqueImplementation.cpp:369-500) including the best/unique fold
 * UpdateUniqueInfo<scores>::update (:97-160, :179-248).  info[i] is the 64-bit
 * UniqueMatchInfo record {state:3@61, fragment:16@45, errors:4@41, fileid:6@35,
 * pos:35@0} (UniqueMatchInfo.hpp:29-39), score[i] its float (init -FLT_MAX,
 * :191); both in/out so folds compose across genome blocks and files exactly
 * as uniqueinfo[] does (matchUniqueImplementation.cpp:1094-1097).  Reads
 * shorter than seedl or containing a symbol > 3 are skipped (:376-394).
 * score may be NULL iff !scores.  Synchronous on return.                      */
int real_hip_match_unique(real_hip_ctx *ctx, const real_hip_batch *b,
                          uint64_t *info, float *score);

/* Pipelined form for host batches -- the producer/consumer block ring of AsynchronousReader.hpp:181-259 as a
 * two-slot ring: submit queues upload, kernels and download of one batch and returns; the upload of the next
 * batch and the download of the previous records run beside the kernels.  A slot's batch, info and score must
 * stay untouched until real_hip_wait(slot) has returned (its status is the batch's).  For the copies to be
 * asynchronous the host memory has to be pinned: real_hip_host_alloc, or the caller's own hipHostRegister.
 * fresh != 0: the records are initialised on the device (NoMatch, score -FLT_MAX: the state of
 * uniqueinfo(numpat), matchUniqueImplementation.cpp:1094-1097) instead of being uploaded.                   */
#define REAL_HIP_SLOTS 2
int real_hip_match_unique_submit(real_hip_ctx *ctx, const real_hip_batch *b, uint64_t *info, float *score,
                                 uint32_t slot, int fresh);
int real_hip_wait(real_hip_ctx *ctx, uint32_t slot);
void *real_hip_host_alloc(size_t bytes);   /* pinned host memory; NULL on failure */
void  real_hip_host_free(void *p);

/* matchAll: replaces AllMatcher::match + unifyMatches
 * (matchAllImplementation.cpp:261-355, :150-161) for the resident block.      */
typedef struct real_hip_hit {       /* MatchPosAndError, matchAllImplementation.cpp:99-120 */
    uint32_t read;                  /* index inside the batch                              */
    uint32_t pos;                   /* 0-based position in the text of fileid              */
    float    score;                 /* 1.0f if !scores (ComputeScore.hpp:31-45)            */
    uint16_t frag;
    uint8_t  k;                     /* mismatches                                          */
    uint8_t  inverted;              /* 0 '+', 1 '-'                                        */
} real_hip_hit;
/* out[hit_offsets[i] .. hit_offsets[i+1]) = hits of read i in unifyMatches order
 * (k, pos, file, frag, score, inverted; duplicates removed).  cap = capacity of
 * out; on REAL_HIP_E_OVERFLOW *n_out is the size needed.  hit_offsets has
 * n_reads+1 entries (may be NULL).                                            */
int real_hip_match_all(real_hip_ctx *ctx, const real_hip_batch *b,
                       real_hip_hit *out, uint64_t cap, uint64_t *n_out, uint64_t *hit_offsets);

/* ---- multi-GPU: one process per GPU, reads sharded contiguously over the ranks, the index replicated.
 * The path has ONE collective: the shards' results to the root, over RCCL (xGMI point-to-point links) -- a
 * concatenation in rank order, nothing is reduced because no read is seen by two ranks.  The reference is a single
 * process; what this replaces is the hand-over of results from its OpenMP threads to the output loop
 * (uniqueinfo[] shared by the threads, matchUniqueImplementation.cpp:1268-1295; the per-block hit emission of
 * matchAllImplementation.cpp:451-535).  Bootstrap: rank 0 calls real_hip_comm_id and hands the 128 bytes to the other
 * ranks by the launcher's own means (MPI_Bcast, a file, the rendezvous store); every rank then calls
 * real_hip_comm_init on its ctx.  All array arguments are DEVICE pointers of the ctx's device; the receive arrays
 * matter on the root only.  Counts travel first: every rank learns every rank's sizes, the root's capacities, which
 * receive arrays the root was given, and whether a rank found something wrong on its own side (a null send array, a
 * scratch buffer it could not reserve).  Every decision to give up is taken from those exchanged tuples alone -- a too
 * small receive array is REAL_HIP_E_OVERFLOW on ALL ranks (with the needed sizes in *n_..._all), null receive arrays on
 * the root or a failed peer are the same error on ALL ranks -- and is taken before any rank posts a send or a receive:
 * nobody is left waiting for a root that has returned.  Then the payload.  `root` must be the same valid rank everywhere.
 * A single process that drives several GPUs through several ctx (real -gpus N) needs none of this: its calls write
 * the shards' results into the caller's host arrays directly.                                                       */
#define REAL_HIP_COMM_ID_BYTES 128
int real_hip_comm_id(uint8_t id[REAL_HIP_COMM_ID_BYTES]);
int real_hip_comm_init(real_hip_ctx *ctx, const uint8_t id[REAL_HIP_COMM_ID_BYTES], int rank, int n_ranks);
int real_hip_comm_destroy(real_hip_ctx *ctx);
/* matchUnique: info_all / score_all (root) = the shards' records in rank order; score may be NULL iff !scores      */
int real_hip_gather_records(real_hip_ctx *ctx, int root, const uint64_t *info, const float *score, uint64_t n_local,
                            uint64_t *info_all, float *score_all, uint64_t cap_all, uint64_t *n_all);
/* matchAll: the shards' unified hit lists as real_hip_match_all returned them (hit_offsets: n_local + 1 entries);
 * on the root hits_all holds all hits with .read rebased to the whole batch and offsets_all (n_reads_all + 1) the
 * rebased offsets                                                                                                   */
int real_hip_gather_hits(real_hip_ctx *ctx, int root, const real_hip_hit *hits, const uint64_t *hit_offsets, uint64_t n_local,
                         uint64_t n_hits_local, real_hip_hit *hits_all, uint64_t cap_hits, uint64_t *offsets_all, uint64_t cap_reads,
                         uint64_t *n_reads_all, uint64_t *n_hits_all);

/* ---- read ingestion on the device: FASTA / FASTQ text -> the arrays of a batch.
 * Replaces FastQReader / FastAReader::getNextPatternUnlocked (FastQReader.hpp:130-180,
 * FastAReader.hpp:107-138), Pattern::computeMapped (Pattern.hpp:105-128, acgtnMap.hpp:39-50) and the
 * quality offset subtraction (FastQReader.hpp:165-173) for text in canonical form: whole records, every
 * field (id, sequence, '+', quality) on one line; "\r\n" accepted.  Anything else -- wrapped sequences,
 * white space inside a field, a chunk that ends inside a record -- is refused with
 * REAL_HIP_E_UNSUPPORTED and left to the caller's reader.  text: n_bytes < 4 GiB, host memory
 * (copied) or device memory (text_on_device).  The returned arrays are device memory owned by ctx,
 * valid until the next real_hip_parse_reads on it; id_start/id_len locate each record's id inside
 * the text (behind its '@' / '>').                                                            */
typedef struct real_hip_parsed {
    uint32_t        struct_size;
    uint32_t        max_patl;    /* longest read of the chunk                              */
    uint64_t        n_reads;
    uint64_t        n_symbols;   /* = offsets[n_reads]                                     */
    const uint8_t  *bases;       /* mapped symbols 0..4, concatenated                      */
    const uint8_t  *qual;        /* quality character - offset; NULL for FASTA             */
    const uint64_t *offsets;     /* n_reads + 1                                            */
    const uint32_t *id_start;    /* n_reads                                                */
    const uint32_t *id_len;      /* n_reads                                                */
} real_hip_parsed;
int real_hip_parse_reads(real_hip_ctx *ctx, const char *text, uint64_t n_bytes, int text_on_device,
                         int fastq, int quality_offset, real_hip_parsed *out);
/* copy of a device array the library returned (the spans of real_hip_parsed: what the output formatter needs
 * beside the records) to host memory; synchronous                                                          */
int real_hip_download(real_hip_ctx *ctx, const void *device_ptr, void *host_ptr, size_t bytes);

/* ---- work counters: accumulated since the last reset ---------- */
typedef struct real_hip_counters {
    uint64_t reads;       /* R  reads matched (not skipped)                           */
    uint64_t lookups;     /* L  ::match calls (12 per read, 7 with the uni0 early-out) */
    uint64_t probes;      /* P  index entries examined inside the buckets             */
    uint64_t candidates;  /* C  entries of the reference's equal range                */
    uint64_t seedpass;    /* S  candidates with seedk <= seedkmax                     */
    uint64_t hits;        /* H  updater::update calls                                 */
    uint64_t verified;    /* distinct (read,strand,pos) actually verified on text     */
    uint64_t handed_over; /* reads (among R) the first pass of the lane matcher did not finish itself: matched by its second
                             pass (many locations, long equal ranges) or by the wave-per-read kernel */
} real_hip_counters;
int real_hip_counters_get(real_hip_ctx *ctx, real_hip_counters *out, int reset);

/* ---- timing: HIP events recorded on the ctx's own stream around every kernel
 * of the path; times are accumulated per kernel since the last reset.         */
enum { REAL_HIP_K_MATCH_UNIQUE = 0, REAL_HIP_K_MATCH_ALL = 1, REAL_HIP_K_ALL_SORT = 2, REAL_HIP_K_INDEX = 3,
       REAL_HIP_K_MATCH_REPEAT = 4, /* second pass over the repeat-rich reads the matcher hands over (scores on) */
       REAL_HIP_K_PARSE = 5,        /* real_hip_parse_reads                                                      */
       REAL_HIP_K_COUNT = 6 };
int real_hip_kernel_time(real_hip_ctx *ctx, int which, double *total_ms, uint64_t *launches, int reset);
int real_hip_timing_enable(real_hip_ctx *ctx, int on);

#ifdef __cplusplus
}
#endif
#endif
