/*
 * real_oracle.h -- CPU restatement of REAL's read-matching hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * Every function cites the reference file:line (relative to the reference's
 * src/ directory) whose behaviour it restates.  The restatement is pinned
 * against the reference's own header-only hot path compiled from
 * /root/reference by oracle/Makefile (oracle/_ref/ref_harness), see
 * oracle/README.md for what is pinned and what is not.
 */
#ifndef REAL_ORACLE_H
#define REAL_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- packed genome text (AutoTextArray.hpp:28-109, ERank222B.hpp:512-566,
 *      RangeVector.hpp:24-80) ------------------------------------------- */
typedef struct ora_genome {
    uint64_t  n;          /* number of symbols (A,C,G,T,N)                    */
    uint64_t  n_words;    /* u64 words of 2-bit text (MSB first)              */
    uint64_t *text;       /* 2-bit text, N stored as 0                        */
    uint64_t  n_wwords;   /* u64 words of the wildcard bit vector             */
    uint64_t *wild;       /* bit i (MSB first) set iff symbol i is N          */
    uint64_t *wild_S;     /* rank superblocks (every 2^16 bits)               */
    uint16_t *wild_M;     /* rank miniblocks (every 64 bits)                  */
    uint64_t  n_wild;     /* number of N symbols                              */
    uint32_t  n_frag;     /* fragments (FASTA records)                        */
    uint64_t *frag_start; /* n_frag+1 entries, last = n ("terminal")          */
    uint64_t  n_fwords;
    uint64_t *fbits;      /* fragment start bit vector, n+1 bits              */
    uint64_t *frag_S;
    uint16_t *frag_M;
} ora_genome;

/* sym: 0..3 = ACGT, 4 = N  (countReads.cpp:83-125) */
ora_genome *ora_genome_create(const uint8_t *sym, uint64_t n,
                              const uint64_t *frag_start, uint32_t n_frag);
void        ora_genome_free(ora_genome *g);
uint64_t    ora_get_text_word(const ora_genome *g, uint64_t i, unsigned l);
uint64_t    ora_rank1(const uint64_t *bits, const uint64_t *S, const uint16_t *M, uint64_t i);
int         ora_is_position_valid(const ora_genome *g, uint64_t pos, unsigned patl);
int         ora_is_dontcare_free(const ora_genome *g, uint64_t pos, unsigned patl);
unsigned    ora_position_to_range(const ora_genome *g, uint64_t pos);

/* ---- genome index in the reference's layout (Mask.hpp, ListSet.hpp,
 *      MapTextFile.hpp:181-230, u_sort.hpp, getLookupTable.hpp:26-51) ----- */
#define ORA_SAMPLE_BITS 22
typedef struct ora_index {
    unsigned  seedl;
    unsigned  sig_bits;      /* = seedl (two segments of seedl/4 symbols)     */
    unsigned  shift;         /* max(sig_bits - 22, 0), same for all six lists */
    uint64_t  n;             /* entries ("masks") in this block               */
    uint64_t  first_window;  /* ordinal of the block's first N-free window    */
    int       have_next;     /* more windows after this block                 */
    uint64_t *sign[6];       /* sorted signatures of list k                   */
    uint32_t *ptr[6];        /* index of the same window in list 5-k          */
    uint32_t *pos[6];        /* window start; lists 0..2 own it, 3..5 = NULL  */
    uint64_t *lookup[6];     /* 2 * 2^22 entries: [low,high) per prefix       */
    const uint32_t *csign[6]; /* compact form only: borrowed sorted signatures  */
    const uint32_t *cpos[6];  /* compact form only: borrowed window starts      */
    int       compact;
} ora_index;

/* CPU-baseline form (bench.py cpu_baseline only): the same sorted lists handed
 * over as {sign, pos} pairs (seedl <= 32), e.g. downloaded from the device;
 * `ptr` is absent, so the partner signature list_b[p->ptr].sign of
 * match.hpp:386 is re-read from the text window at pos (same value).  The
 * 22-bit lookup tables are rebuilt with getLookupTable's semantics.           */
struct ora_index *ora_index_from_entries(const ora_genome *g, unsigned seedl, uint64_t n,
                                         const uint32_t *const sign[6], const uint32_t *const pos[6]);

/* block = windows [first_window, first_window + max_entries) in text order   */
ora_index *ora_index_build(const ora_genome *g, unsigned seedl,
                           uint64_t first_window, uint64_t max_entries);
void       ora_index_free(ora_index *ix);
uint32_t   ora_index_getpos(const ora_index *ix, int list, uint64_t j);

/* ---- signatures (SignatureConstruction.hpp) ----------------------------- */
/* returns 0 if a symbol > 3 is met inside the seed                          */
int  ora_signature_mapped(unsigned seedl, const uint8_t *mapped, uint32_t m[4]);
int  ora_reverse_mapped_signature(unsigned seedl, const uint8_t *mapped, uint32_t m[4]);
void ora_signatures(unsigned seedl, const uint32_t m[4], uint64_t s[6]);

/* ---- scoring (Scoring.cpp:61-171, ComputeScore.hpp:50-190) --------------- */
void  ora_scoring_table(double similarity, double gc, double trans, double err,
                        double gcmut_bias, double LL[1024], double odds[16]);
void  ora_scoring_defaults(double *similarity, double *gc, double *trans,
                           double *err, double *gcmut_bias);
float ora_compute_score(const ora_genome *g, const double *LL, int inverted,
                        const uint8_t *mapped, const uint8_t *qual,
                        uint32_t pos, unsigned patl);

/* ---- record (UniqueMatchInfo.hpp:24-203) -------------------------------- */
enum { ORA_NOMATCH = 0, ORA_STRAIGHT = 1, ORA_REVERSE = 2, ORA_GAPPED = 3, ORA_NONUNIQUE = 4 };
uint64_t ora_record_pack(unsigned state, unsigned frag, unsigned errors, unsigned fileid, uint64_t pos);
void     ora_record_unpack(uint64_t rec, unsigned *state, unsigned *frag,
                           unsigned *errors, unsigned *fileid, uint64_t *pos);

/* ---- matching ----------------------------------------------------------- */
typedef struct ora_params {
    uint32_t seedl;
    uint32_t seedkmax;
    uint32_t totalkmax;
    uint32_t scores;       /* 0/1 */
    uint32_t fileid;
    uint32_t threads;      /* OpenMP threads for the batch calls, 0 = default */
    double   filter_mult;  /* epsilon = (float)(filter_mult * patl)           */
    double   LL[1024];
} ora_params;

/* work counters of SURVEY 8(d) */
typedef struct ora_counters {
    uint64_t reads;      /* R: reads that reached the matcher (not skipped)   */
    uint64_t lookups;    /* L: ::match calls                                  */
    uint64_t probes;     /* P: binary-search probes inside [low,high)         */
    uint64_t candidates; /* C: entries in the equal range                     */
    uint64_t seedpass;   /* S: candidates with seedk <= seedkmax              */
    uint64_t hits;       /* H: updater::update calls                          */
} ora_counters;

/* one hit event = one updater::update call (match.hpp:404-411) */
typedef struct ora_event {
    uint64_t read;
    uint32_t pos;
    uint32_t frag;
    float    score;
    uint8_t  inverted;
    uint8_t  list;
    uint8_t  totalk;
    uint8_t  seedk;
} ora_event;

/* matchUnique over a batch (matchUniqueImplementation.cpp:369-500, fold
 * :97-160 / :179-248).  info/score are in/out so calls compose across genome
 * blocks and files.  events (nullable) receives the ordered update stream.   */
int ora_match_unique(const ora_genome *g, const ora_index *ix, const ora_params *p,
                     const uint8_t *bases, const uint8_t *qual,
                     const uint64_t *offsets, uint64_t n_reads,
                     uint64_t *info, float *score, ora_counters *ctr,
                     ora_event *events, uint64_t event_cap, uint64_t *n_events);

/* the raw update() stream of all 12 ::match calls per read, no early-out and
 * no fold (what AllMatcher::match hands to unifyMatches, and what
 * oracle/_ref/ref_harness logs from the reference); e.list = call index
 * 0..11 = inverted*6 + list.                                                */
int ora_match_events(const ora_genome *g, const ora_index *ix, const ora_params *p,
                     const uint8_t *bases, const uint8_t *qual,
                     const uint64_t *offsets, uint64_t n_reads,
                     ora_event *events, uint64_t event_cap, uint64_t *n_events,
                     ora_counters *ctr);

/* matchAll hit (matchAllImplementation.cpp:99-120), unifyMatches order       */
typedef struct ora_hit {
    uint64_t read;
    uint32_t pos;
    uint32_t frag;
    float    score;
    uint8_t  inverted;
    uint8_t  k;
    uint16_t fileid;
} ora_hit;

/* matchAll over a batch (matchAllImplementation.cpp:261-355, :150-161).
 * hit_offsets has n_reads+1 entries.  returns -1 if cap is too small (n_out
 * then holds the needed size).                                              */
int ora_match_all(const ora_genome *g, const ora_index *ix, const ora_params *p,
                  const uint8_t *bases, const uint8_t *qual,
                  const uint64_t *offsets, uint64_t n_reads,
                  ora_hit *out, uint64_t cap, uint64_t *n_out,
                  uint64_t *hit_offsets, ora_counters *ctr);

/* fold one event into a record; exported so tests can drive the state machine */
void ora_update_unique(int scores, int inverted, unsigned fileid, uint32_t pos,
                       unsigned totalk, float score, float epsilon, unsigned fragid,
                       uint64_t *info, float *info_score);

unsigned ora_diffcountpair32(uint32_t a, uint32_t b);
unsigned ora_diffcountpair64(uint64_t a, uint64_t b);

#ifdef __cplusplus
}
#endif
#endif
