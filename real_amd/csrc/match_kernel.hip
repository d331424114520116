// match_kernel.hip -- the per-read matcher of REAL as one hand-written gfx950 kernel
// (SURVEY 8a rows a1..a12):
//   read symbols, eligibility, signatures   Pattern.hpp:105-128, matchUniqueImplementation.cpp:376-394,
//                                           SignatureConstruction.hpp:62-67, 218-280, 347-410
//   bucket lookup + in-bucket search        match.hpp:376-381
//   seed popcount filter                    match.hpp:383-388, PopCountTable.hpp:113-131
//   position / fragment / N checks          match.hpp:390-398
//   Hamming verify                          RestMatch.hpp:39-81
//   quality aware score                     ComputeScore.hpp:50-190
//   best/unique fold                        matchUniqueImplementation.cpp:97-160, 179-248
//   or hit append of matchAll               matchAllImplementation.cpp:172-184
//
// Work decomposition: one lane per read, a wave per tile of 64 reads.  The grid is as many 256-thread workgroups as the
// device holds at a time (three per CU: three waves per SIMD); a wave takes tiles from a counter until none is left.
//   front  the wave copies the bases of its 64 reads (one contiguous range of the caller's array) into its
//          LDS region; every lane packs its own read into registers (32 bases per word), derives the seed
//          halves of both strands; the reverse complement is computed in registers.
//   match  per strand: the equal ranges of the six lists are enumerated in list order and the entries that survive
//          the partner filter go to the lane's queue in LDS (list-major, entry order = the reference's candidate
//          order; a window reached through consecutive lists is one entry with a list mask); the queue is drained
//          in order: seed window and the words of the whole read from the 2-bit text in one go (the next candidate's
//          are requested before this one is looked at), popcount filters, whole-read Hamming distance.  With scores
//          on a verified window and its update() events are parked.  Three table kinds feed this stage (match_lists:
//          bucket starts; match_lists_fine: directory entries with partner digests or key fingerprints) and a
//          fourth replaces it by lookups of lane groups (match_lists_rows: one 128-byte row = one line of HBM per
//          lookup, the next list's rows in flight while this one is decoded).
//   back   the qualities of the wave's reads come to LDS by LDS-DMA while the queues are drained for the last time
//          (bucket rows; otherwise the wave copies them as it did the bases); all lanes score their parked windows
//          together and replay their events into the fold / append them for matchAll.
// The update() events of a read reach the fold in the canonical order (strand, list, position), which
// matters because the fold is order dependent when scores are on.  Every index access is a
// dependent random read of a 128-byte line of an HBM-resident table: the kernel is bound by the HBM lines it
// moves (13.6 per read with bucket rows on the C2 workload), not by arithmetic (no MFMA:
// XOR/popcount and a short FP64 add chain).
#include "match_common.h"
#include <cstdlib>

// RH_ABLATE (experimental builds only, make variant): bit 0 no scoring, bit 1 no verification of queued candidates,
// bit 2 no entries examined, bit 3 bucket rows fetched but not decoded -- results are wrong, the time tells what a stage costs
#ifndef RH_ABLATE
#define RH_ABLATE 0
#endif
// RH_PHASE_TIMING (experimental builds only): the work counters L,P,C,S,H,V and handed_over are replaced by the time
// (ticks of 10 ns, summed over the waves) spent in: front, waiting for rows, decoding rows, draining queues (verification),
// staging qualities, scoring + delivering, and the whole wave
#ifndef RH_PHASE_TIMING
#define RH_PHASE_TIMING 0
#endif

#if RH_PHASE_TIMING
#define PH_NOW() ((unsigned)__builtin_amdgcn_s_memrealtime())
#else
#define PH_NOW() 0u
#endif
#define EB 8  // index entries a lane requests per round trip while enumerating equal ranges
#define MQ 16 // candidate queue slots per lane (LDS) of the table kinds 0..2; a full queue is drained and refilled
// LDS bytes of one wave: its candidate queue (MQ x 64 positions + lists, 6 x 64 cursors; the first 6.5 KiB) while
// it matches; before and after, the staging area through which it reads the bases / qualities of its reads from
// the batch.  9.5 KiB hold the 64 reads of a wave up to 151 bases each in one go; with the 8 KiB score table that
// is 46 KiB per workgroup = three workgroups per CU, which is what the registers allow anyway.
#define QUEUE_BYTES (MQ * 64u * 5u + 6u * 64u * 4u)
__host__ __device__ constexpr uint32_t stg_bytes(int W, int TK) { return TK >= 3 ? 11008u : (W <= 4 ? QUEUE_BYTES : 9728u); }
// bucket rows (table kind 3): a queue of MQR slots per lane (a read that needs more for one strand is handed over),
// then the 64 rows of a list, 128 bytes each
#define MQR 8u
#define ROWBUF_OFF (MQR * 64u * 5u)
#define BKX_OFF (ROWBUF_OFF + 64u * 128u) /* 64 bucket numbers, transposed for the piece loaders */
static_assert(BKX_OFF + 256u <= stg_bytes(4, 3) && ROWBUF_OFF % 16 == 0 && QUEUE_BYTES % 16 == 0, "row staging fits the wave's LDS region");
// The second pass (bucket rows; scores on or matchAll): the reads whose locations or queue entries outgrew a lane of the first
// pass -- reads on five and more copies of a locus -- are matched once more by the same lane-per-read code with room for
// NPEND2 parked locations and MQR2 queue slots per strand (registers and LDS of an instance that only sees a few per cent
// of the reads).  What outgrows that too, long equal ranges and long reads go to the wave-per-read kernel.
#define NPEND2 24
#define MQR2 24u
__host__ __device__ constexpr uint32_t stg_bytes2() { return MQR2 * 64u * 5u + 64u * 128u + 256u; }
#define STG_PAD 16u
#ifndef RH_KEEP_TW_MAXW
#define RH_KEEP_TW_MAXW 4
#endif
#define NPEND 4      // verified locations a lane parks until their scores are computed (flush_pending), each with the
                     // set of lists through which it was reached (= its update() events)
#define SLOT_NONE 31u
#define SLOT_BIG 30u   // cslot of a read that is handed over because of a long equal range (not for want of room: the second pass would be no help)
#define PEND_OVF 0xffu // p_n of a read that is handed over to the wave-cooperative matcher (match_wave.hip)
#define BIG_T 48u      // an equal range / bucket scan longer than this many entries is not walked by one lane: hand-over
#define BIG_T2 1024u   // ... of the second pass: its waves hold handed-over reads only, and a long range whose entries nearly all fail the
                       // partner filter (a skewed base composition is full of them) is cheap to walk sixteen entries per round trip

// bits 9..14 of a parked location's meta word: the lists whose update() call it has had
#define PM_LM_SHIFT 9
#define PM_LM_MASK (63u << PM_LM_SHIFT)
template <int W, bool SCORES, bool ALL, int NP = NPEND>
struct LaneState {
    // read
    uint64_t O[W];      // oriented read, 32 bases per word
    uint64_t shi, slo;  // seed halves (m0|m1), (m2|m3) of the oriented read
    uint32_t patl;
    int inv;
    // (derived from patl / inv where they are needed instead of being carried along: registers)
    __device__ __forceinline__ uint32_t nw() const { return (patl + 31) >> 5; }                                            // words of 32 bases
    __device__ __forceinline__ uint64_t lastmask() const { return ~0ull << (64 - 2 * (patl - 32 * (nw() - 1))); }          // the bases of the last word
    __device__ __forceinline__ uint32_t so(uint32_t l) const { return inv ? (patl - l) : 0u; }                             // RestMatch::getMatchOffset, RestMatch.hpp:84-89
    __device__ __forceinline__ float eps(double filter_mult) const { return (float)(filter_mult * (double)patl); }         // RealOptions.hpp:74-77
    uint64_t r, o0, o1; // read index; offsets of its first byte and of the byte behind its last one in the batch arrays
    // result
    uint64_t info;
    float iscore;
    // memo of the last verified position of this strand: the same window is reached through
    // up to six lists; its verdict is a function of (strand,pos) only
    uint32_t cpos, ck, cfrag;
    float cscore;
    bool cok;
    uint32_t crpos, ckk; // last seed window looked at and its per-segment mismatch counts (4 x 8 bits)
    // SCORES: verified locations whose score is still to be computed, and the update() events that refer
    // to them, in event order (see flush_pending)
    uint32_t p_pos[NP], p_meta[NP]; // text position; k | strand << 8 | lists << 9 | fragment << 16
    uint64_t p_tw[W <= RH_KEEP_TW_MAXW ? W : 1]; // aligned text words of pending location 0 (kept for reads of up to 128 bases;
                                          // longer ones read the text again when they score: the registers are worth more)
    uint32_t p_n, cslot;                  // locations, slot of the memo
    uint32_t nhit; // matchAll: hits appended for this read
    // work counters of this read, packed (a register each would cost six of the 168): cA = L:4 | V:12 | S:12, cB = P:11 | C:11 | H:10.
    // No field overflows without the read being handed over -- a lane walks at most BIG_T entries of each of its 12 equal
    // ranges (plus a binary search) -- and a read that is handed over counts for nothing here (the wave matcher counts it).
    // (the second pass walks up to BIG_T2 entries per range: P and C get words of their own there)
    unsigned cA, cB, cPw, cCw;
    __device__ __forceinline__ void addL(unsigned n) { cA += n; }
    __device__ __forceinline__ void addV(unsigned n) { cA += n << 4; }
    __device__ __forceinline__ void addS(unsigned n) { cA += n << 16; }
    __device__ __forceinline__ void addP(unsigned n) { if (NP > NPEND) cPw += n; else cB += n; }
    __device__ __forceinline__ void addC(unsigned n) { if (NP > NPEND) cCw += n; else cB += n << 11; }
    __device__ __forceinline__ unsigned getP() const { return NP > NPEND ? cPw : (cB & 2047u); }
    __device__ __forceinline__ unsigned getC() const { return NP > NPEND ? cCw : ((cB >> 11) & 2047u); }
    __device__ __forceinline__ void addH(unsigned n) { cB += n << 22; }
#if RH_PHASE_TIMING
    unsigned tW, tD, tR; // ticks of 10 ns this wave spent waiting for rows, decoding them, draining its queues
#endif
};


// element j of a small register array, j a run-time value (selects over static indices: a run-time index would send the
// array through scratch memory)
template <int N>
__device__ __forceinline__ uint32_t sel(const uint32_t (&v)[N], uint32_t j)
{
    // (an OR of masked words, not a chain of selects: the optimiser turns such a chain over the elements of one array into a
    // load at a run-time index, and the array -- with it the whole lane state -- then lives in scratch memory)
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) r |= j == (uint32_t)i ? v[i] : 0u;
    return r;
}
template <int N>
__device__ __forceinline__ float sel(const float (&v)[N], uint32_t j)
{
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) r |= j == (uint32_t)i ? __float_as_uint(v[i]) : 0u;
    return __uint_as_float(r);
}
template <int N, class T>
__device__ __forceinline__ void put(T (&v)[N], uint32_t j, T x)
{
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = j == (uint32_t)i ? x : v[i];
}

// the update() call itself: the best/unique fold, or the matchAll append
template <int W, bool SCORES, bool ALL, int NP>
__device__ __forceinline__ void deliver(const MatchArgs &a, LaneState<W, SCORES, ALL, NP> &s, uint32_t pos, uint32_t meta, float score)
{
    if (ALL) {
        unsigned long long slot = wave_append_slot(a.raw_count);
        if (slot < a.raw_cap) a.raw[slot] = make_uint4((uint32_t)s.r, pos, __float_as_uint(score), meta & ~PM_LM_MASK);
        s.nhit++;
    } else {
        fold_update<SCORES>((meta >> 8) & 1, a.t.fileid, pos, meta & 0xff, score, s.eps(a.filter_mult), meta >> 16, s.info, s.iscore);
    }
}

// Scores of the parked locations, then their update() events in the reference's order.  The score is
// ~1000 instructions of which every lane of a wave needs one or two per read, but at different points
// of its candidate loop: computed where the hit is found, the wave would run that code once per
// distinct point (5-6 times per wave, mostly idle lanes).  Parked, the lanes score together at the end
// of the read.  A read that needs more slots (repeat-rich) is handed to the repeat kernel, which scores in place.
//
// Order of the events (the fold is order dependent when scores are on).  The reference calls ::match once
// per strand and list (matchUniqueImplementation.cpp:407-497) and walks the equal range of that list in entry order =
// ascending position (match.hpp:383-413): its update() calls are the triples (strand, list, position) that pass, in
// lexicographic order.  A parked location carries the set of lists through which it passed, so that sequence is rebuilt
// here from the sets alone, whatever the order in which the queue was filled and drained (a window reached again through
// a later list, after other windows, has its later events at their own lists' turns -- not at its first one's).
// Two consecutive update() calls for the same location are one: the second finds the record either at that location
// (nothing differs), NonUnique with the same score comparison that just failed, or as untouched as the first left it.
template <int W, bool SCORES, bool ALL, class Row, int NP>
__device__ __forceinline__ void flush_pending(const MatchArgs &a, LaneState<W, SCORES, ALL, NP> &s, const double *sLL, const Row &qrow)
{
    float sc[NP]; // (static indices only: sel / put)
#pragma unroll
    for (int i = 0; i < NP; ++i) sc[i] = 1.0f;
#pragma unroll 1
    for (uint32_t j = 0; j < s.p_n; ++j) {
        if (!SCORES || (RH_ABLATE & 1)) break; // ComputeScore<...,false>: 1.0f (ComputeScore.hpp:31-45)
        const uint32_t pos = sel<NP>(s.p_pos, j);
        const uint32_t meta = sel<NP>(s.p_meta, j);
        const uint32_t inv = (meta >> 8) & 1;
        uint64_t Ow[W], tw[W];
        if ((int)inv == s.inv) {
#pragma unroll
            for (int i = 0; i < W; ++i) Ow[i] = s.O[i];
        } else {
            revcomp_words<W>(s.O, Ow, s.patl);
        }
        if (j == 0 && W <= RH_KEEP_TW_MAXW) {
#pragma unroll
            for (int i = 0; i < W; ++i) tw[i] = s.p_tw[W <= RH_KEEP_TW_MAXW ? i : 0];
        } else { // a further location of a read (any location of a long read): the text is read again
            const uint64_t wi = pos >> 5;
            const unsigned sh = 2u * (pos & 31);
            uint64_t t[W + 2];
#pragma unroll
            for (int i = 0; i <= W; i += 2) {
                U64x2 p2 = {0ull, 0ull};
                if ((uint32_t)i <= s.nw()) p2 = load2(a.t.text + wi + i);
                t[i] = p2.a; t[i + 1] = p2.b;
            }
#pragma unroll
            for (int i = 0; i < W; ++i) tw[i] = sh ? ((t[i] << sh) | (t[i + 1] >> (64 - sh))) : t[i];
        }
        const float v = score_location<W>(sLL, Ow, tw, s.patl, qrow, a.b.qual != nullptr, inv);
        put<NP>(sc, j, v);
    }
    auto lm_of = [&](uint32_t meta) { return (meta >> PM_LM_SHIFT) & 63u; };
    if (s.p_n == 1) {
        if (lm_of(s.p_meta[0])) deliver<W, SCORES, ALL>(a, s, s.p_pos[0], s.p_meta[0], sc[0]); // (all events of the read are this location's: one call)
        return;
    }
    if (ALL) {
        // matchAll keeps one hit per location (process_loaded); unifyMatches orders them afterwards (rh_all_finish)
#pragma unroll 1
        for (uint32_t j = 0; j < s.p_n; ++j) {
            const uint32_t meta = sel<NP>(s.p_meta, j);
            if (lm_of(meta)) deliver<W, SCORES, ALL>(a, s, sel<NP>(s.p_pos, j), meta, sel<NP>(sc, j));
        }
        return;
    }
    if (s.p_n < 2) return;
    // rank of every location by (strand, position); ord = the locations in that order, 5 bits each, twelve to a word
    static_assert(NP <= 24, "two words of twelve ranks");
    uint64_t ord0 = 0, ord1 = 0;
    auto ord_at = [&](uint32_t rk) { return (uint32_t)((rk < 12 ? ord0 >> (5 * rk) : ord1 >> (5 * (rk - 12))) & 31u); };
#pragma unroll
    for (uint32_t j = 0; j < (uint32_t)NP; ++j) {
        const uint64_t kj = ((uint64_t)((s.p_meta[j] >> 8) & 1u) << 32) | s.p_pos[j];
        uint32_t rank = 0;
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)NP; ++i) {
            const uint64_t ki = ((uint64_t)((s.p_meta[i] >> 8) & 1u) << 32) | s.p_pos[i];
            if (i != j && i < s.p_n && (ki < kj || (ki == kj && i < j))) rank++;
        }
        if (j < s.p_n && rank < 12) ord0 |= (uint64_t)j << (5 * rank);
        if (j < s.p_n && rank >= 12) ord1 |= (uint64_t)j << (5 * (rank - 12));
    }
    if (NP <= 4) {
        // the event sequence first (2 bits per event = its location; at most 4 x 6 events), then the calls: the fold's code
        // exists once and runs as often as the lane with the most events needs it
        uint64_t ev = 0;
        uint32_t nev = 0, last = SLOT_NONE;
#pragma unroll 1
        for (uint32_t sl = 0; sl < 12; ++sl) { // strand, list
            const uint32_t inv = sl >= 6 ? 1u : 0u, la = sl - 6 * inv;
#pragma unroll
            for (uint32_t rk = 0; rk < (uint32_t)NP; ++rk) {
                const uint32_t j = ord_at(rk);
                const uint32_t meta = sel<NP>(s.p_meta, j);
                if (rk < s.p_n && ((meta >> 8) & 1u) == inv && ((meta >> (PM_LM_SHIFT + la)) & 1u) && j != last) {
                    ev |= (uint64_t)j << (2 * nev); nev++; last = j;
                }
            }
        }
#pragma unroll 1
        for (uint32_t e = 0; e < nev; ++e) {
            const uint32_t j = (uint32_t)(ev >> (2 * e)) & 3u;
            deliver<W, SCORES, ALL>(a, s, sel<NP>(s.p_pos, j), sel<NP>(s.p_meta, j), sel<NP>(sc, j));
        }
    } else {
        // (the second pass over the reads with many locations: the calls where they are found)
        uint32_t last = SLOT_NONE;
#pragma unroll 1
        for (uint32_t sl = 0; sl < 12; ++sl) {
            const uint32_t inv = sl >= 6 ? 1u : 0u, la = sl - 6 * inv;
#pragma unroll 1
            for (uint32_t rk = 0; rk < s.p_n; ++rk) {
                const uint32_t j = ord_at(rk);
                const uint32_t meta = sel<NP>(s.p_meta, j);
                if (((meta >> 8) & 1u) == inv && ((meta >> (PM_LM_SHIFT + la)) & 1u) && j != last) {
                    deliver<W, SCORES, ALL>(a, s, sel<NP>(s.p_pos, j), meta, sel<NP>(sc, j));
                    last = j;
                }
            }
        }
    }
}

// one member of a bucket whose fingerprint equals the read's: the body of the candidate loop
// of ::match (match.hpp:383-413)
// A survivor of the partner filter goes to the lane's queue.  A window that is in the queue already -- the true locus is
// found through 4.4 lists per read, the copies of a repeat through as many each -- only gets that list's bit set in its
// entry: the drain then runs once per window instead of once per (window, list).
// `anywhere`: the window is looked for in the whole queue, not only in its last entry.  The order in which the fold sees
// the update() calls does not depend on the queue when they are parked (scores on, matchAll): flush_pending rebuilds the
// reference's order from the parked lists.  Without scores they are delivered as the queue is drained, the fold's result
// is order-free but for the position bits a NonUnique record keeps (those of the first hit at its error level), and the
// queue must hold the windows in the order of their first update(): with entries whose being enumerated means being a
// member of the list's equal range (32-bit signatures: the entry carries the whole signature) a window's first entry IS
// its first update(), if it has any, and it may be merged anywhere; with wider signatures an entry is enumerated on a
// prefix or fingerprint and may turn out not to be a member (process_loaded) -- merged into such an earlier entry, the
// window's first real update() would come ahead of windows that were reached before it -- so there only the last entry
// is merged with (a new entry would take the same place).
__device__ __forceinline__ void queue_push(uint32_t *q_pos, uint8_t *q_la, uint32_t &qn, uint32_t pos, int la, bool anywhere)
{
    for (uint32_t k = qn; k-- > 0;) { // (from the last entry: that is where the true locus of the previous list is)
        if (q_pos[k * 64] == pos) { q_la[k * 64] |= (uint8_t)(1u << la); return; }
        if (!anywhere) break;
    }
    q_pos[qn * 64] = pos; q_la[qn * 64] = (uint8_t)(1u << la); qn++;
}

// The text a candidate needs: the two (three) words of its seed window and the words of text[pos, pos + patl).  They are
// requested together, before anything is decided -- one round trip per candidate instead of two (a queued candidate has
// passed the partner filter: it nearly always gets as far as the Hamming distance) -- and the drain of the bucket-row
// matcher requests those of the next candidate before it works on this one.
template <int W>
struct CandText {
    U64x2 tt;
    uint64_t t2;
    uint64_t t[W + 2];
};
template <int W, bool SCORES, bool ALL, int NP>
__device__ __forceinline__ void cand_load(const MatchArgs &a, const LaneState<W, SCORES, ALL, NP> &s, uint32_t rpos, CandText<W> &c)
{
    const uint64_t *__restrict__ T = a.t.text;
    const uint64_t wi0 = rpos >> 5;
    c.tt = load2(T + wi0); // the seed window (2*l bits at bit offset 2*rpos) spans two words, three when l > 32
    c.t2 = (a.l > 32) ? T[wi0 + 2] : 0ull;
    const uint32_t so = s.so(a.l), nw = s.nw();
    const uint64_t wi = (rpos - so) >> 5;
#pragma unroll
    for (int j = 0; j <= W; j += 2) { // 16-byte requests, all in flight together
        U64x2 p2 = {0ull, 0ull};
        if (rpos >= so && (uint32_t)j <= nw) p2 = load2(T + wi + j);
        c.t[j] = p2.a; c.t[j + 1] = p2.b;
    }
}

// lmask = the lists (bit la) through which this window was reached one right after the other: the window is looked at
// once, its update() events are delivered list by list
template <int W, bool SCORES, bool ALL, bool DEFER, int NP>
__device__ __forceinline__ void process_loaded(const MatchArgs &a, LaneState<W, SCORES, ALL, NP> &s, const double *sLL,
                                               uint32_t rpos, uint32_t lmask, const CandText<W> &c)
{
    if (s.p_n == PEND_OVF) return; // handed over
    if (RH_ABLATE & 2) return;
    const uint32_t bb = a.b_bits;
    const uint64_t mb = (bb >= 64) ? ~0ull : ((1ull << bb) - 1);
    const uint32_t so = s.so(a.l);
    const uint32_t nw = s.nw();
    // seed window of the genome at rpos, as the two halves (m0|m1), (m2|m3); per-segment mismatch
    // counts are a function of (strand, rpos) only and are memoised: the true locus is reached
    // through up to six lists in a row
    if (rpos != s.crpos) {
        const unsigned sh0 = 2u * (rpos & 31);
        const uint64_t xhi = extract_bits(c.tt.a, c.tt.b, c.t2, sh0, a.l) ^ s.shi;
        const uint64_t xlo = extract_bits(c.tt.a, c.tt.b, c.t2, sh0 + a.l, a.l) ^ s.slo;
        const uint64_t dhi = ((xhi >> 1) | xhi) & M55, dlo = ((xlo >> 1) | xlo) & M55;
        s.ckk = __popcll(dhi >> bb) | (__popcll(dhi & mb) << 8) | (__popcll(dlo >> bb) << 16) | (__popcll(dlo & mb) << 24);
        s.crpos = rpos;
    }
    const unsigned k0 = s.ckk & 0xff, k1 = (s.ckk >> 8) & 0xff, k2 = (s.ckk >> 16) & 0xff, k3 = s.ckk >> 24;
    // the lists of lmask of whose equal range the window is a member: both segments the list is keyed on are
    // mismatch free (s0..s5 = (0,1),(0,2),(0,3),(1,2),(1,3),(2,3)); otherwise the signature is wider than prefix+32
    const bool z0 = !k0, z1 = !k1, z2 = !k2, z3 = !k3;
    const uint32_t members = lmask & ((z0 && z1 ? 1u : 0u) | (z0 && z2 ? 2u : 0u) | (z0 && z3 ? 4u : 0u) | (z1 && z2 ? 8u : 0u) |
                                      (z1 && z3 ? 16u : 0u) | (z2 && z3 ? 32u : 0u));
    if (!members) return;
    const uint32_t nm = __popc(members);
    if (!a.ix.pbits) s.addC(nm); // (with partner bits the entry holds the whole signature: counted at the scan)
    const unsigned seedk = k0 + k1 + k2 + k3; // = diffcountpair(s_b, list_b[p->ptr].sign), match.hpp:386
    if (seedk > a.seedkmax) return;
    s.addS(nm);
    if (rpos < so) return; // match.hpp:393
    const uint32_t pos = rpos - so;
    bool reg = false; // a location verified here for the first time
    uint32_t parked = SLOT_NONE; // ... or one that is parked already (reached again after other windows: a repeat)
    if (DEFER && pos != s.cpos) {
#pragma unroll
        for (uint32_t j = 0; j < (uint32_t)NP; ++j)
            if (j < s.p_n && s.p_pos[j] == pos && ((s.p_meta[j] >> 8) & 1u) == (uint32_t)s.inv) parked = j;
    }
    if (parked != SLOT_NONE) {
        const uint32_t meta0 = sel<NP>(s.p_meta, parked);
        s.cpos = pos; s.cok = true; s.ck = meta0 & 0xffu; s.cscore = 1.0f; s.cfrag = meta0 >> 16; s.cslot = parked;
    } else if (pos != s.cpos) {
        s.cpos = pos;
        s.cok = false;
        s.addV(1u);
        uint32_t frag;
        if (!frag_valid(a.t, pos, s.patl, frag)) return;
        if (a.t.has_wild && !wild_free(a.t.wild, pos, s.patl)) return;
        // Hamming distance of the whole oriented read against text[pos, pos+patl)
        // = seedk + RestMatch::computeDistance (RestMatch.hpp:39-81)
        const unsigned sh = 2u * (pos & 31);
        uint64_t tw[W];
        unsigned total = 0;
        const uint64_t lastmask = s.lastmask();
        {
#pragma unroll
            for (int j = 0; j < W; ++j) {
                uint64_t al = sh ? ((c.t[j] << sh) | (c.t[j + 1] >> (64 - sh))) : c.t[j];
                tw[j] = al;
                uint64_t x = al ^ s.O[j];
                uint64_t d = ((x >> 1) | x) & M55;
                if ((uint32_t)j + 1 == nw) d &= lastmask;
                if ((uint32_t)j < nw) total += __popcll(d);
            }
        }
        if (total > a.totalkmax) return;
        const float sc = 1.0f; // ComputeScore<...,false>, ComputeScore.hpp:31-45 (scores on: computed by flush_pending)
        s.cok = true; s.ck = total; s.cscore = sc; s.cfrag = frag; s.cslot = SLOT_NONE;
        if (DEFER) { // the score is computed later (flush_pending): park the location
            if (s.p_n == (uint32_t)NP) { s.p_n = PEND_OVF; return; } // out of room => the read is handed over
            const uint32_t meta0 = total | ((uint32_t)s.inv << 8) | (frag << 16);
            put<NP>(s.p_pos, s.p_n, pos);
            put<NP>(s.p_meta, s.p_n, meta0);
            if (s.p_n == 0 && SCORES && W <= RH_KEEP_TW_MAXW) {
#pragma unroll
                for (int j = 0; j < W; ++j) s.p_tw[W <= RH_KEEP_TW_MAXW ? j : 0] = tw[j];
            }
            s.cslot = s.p_n++;
            reg = true;
        }
    }
    (void)reg;
    if (!s.cok) return;
    s.addH(nm); // one updater::update call per list, match.hpp:411
    uint32_t events = members;
    if (ALL) {
        // unifyMatches (matchAllImplementation.cpp:150-161) only removes exact duplicates: the same
        // (strand,pos) reached through a later list.  A hit is kept iff it comes from the first list whose
        // two segments are mismatch free.
        const int first = (z0 && z1) ? 0 : (z0 && z2) ? 1 : (z0 && z3) ? 2 : (z1 && z2) ? 3 : (z1 && z3) ? 4 : 5;
        events &= 1u << first;
    }
    const uint32_t meta = s.ck | ((uint32_t)s.inv << 8) | (s.cfrag << 16);
    const uint32_t ne = __popc(events);
    (void)meta; (void)ne;
    if (!DEFER) {
        for (uint32_t e = 0; e < ne; ++e) deliver<W, SCORES, ALL>(a, s, s.cpos, meta, s.cscore);
        return;
    }
    // park the events: the lists of this location's update() calls (flush_pending puts them in order)
    // (every word is written, with static indices: a conditional store would be turned into one store at a run-time index,
    // and the parked locations -- with them the whole lane state -- would live in scratch memory)
#pragma unroll
    for (uint32_t j = 0; j < (uint32_t)NP; ++j) s.p_meta[j] |= (s.cslot == j ? events : 0u) << PM_LM_SHIFT;
}

template <int W, bool SCORES, bool ALL, bool DEFER, int NP>
__device__ __forceinline__ void process_candidate(const MatchArgs &a, LaneState<W, SCORES, ALL, NP> &s, const double *sLL,
                                                  uint32_t rpos, uint32_t lmask)
{
    if (s.p_n == PEND_OVF) return; // handed over
    CandText<W> c;
    cand_load<W, SCORES, ALL>(a, s, rpos, c);
    process_loaded<W, SCORES, ALL, DEFER>(a, s, sLL, rpos, lmask, c);
}

// Scan of the buckets of lists [LA0, LA1) of one strand; pushes the entries that survive the key
// (and partner) comparison into the lane's LDS queue, list-major and in entry order = the reference's
// candidate order.  All bucket-table loads are issued together, then the first two entries of every bucket
// together.  A scan longer than BIG_T entries, or more survivors than the queue holds (repeat-rich loci only),
// hands the read over to the wave-cooperative matcher: no lane walks a long range alone.
template <int W, bool SCORES, bool ALL, int LA0, int LA1, int NP>
__device__ __forceinline__ void scan_lists(const MatchArgs &a, LaneState<W, SCORES, ALL, NP> &s, uint32_t *q_pos, uint8_t *q_la, uint32_t &qn)
{
    const uint32_t bb = a.b_bits;
    const uint64_t mb = (bb >= 64) ? ~0ull : ((1ull << bb) - 1);
    const uint64_t m[4] = {s.shi >> bb, s.shi & mb, s.slo >> bb, s.slo & mb};
    constexpr int NL = LA1 - LA0;
    const uint32_t pbits = a.ix.pbits;
    const uint32_t pmask = pbits ? ((1u << pbits) - 1) : 0u;
    uint32_t fp[NL], lo[NL], hi[NL];
    // 1. bucket starts of all lists of the strand: 2*NL independent loads
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int la = LA0 + i;
        const int xa = (la < 3) ? 0 : (la < 5) ? 1 : 2, xc = (la == 0) ? 1 : (la == 1 || la == 3) ? 2 : 3;
        const uint64_t sa = (m[xa] << bb) | m[xc]; // s_a of list la, SignatureConstruction.hpp:62-67
        const uint32_t prefix = (uint32_t)(sa >> a.ix.pshift);
        fp[i] = (uint32_t)((sa >> a.ix.fshift) & ((a.ix.fbits >= 32) ? 0xffffffffull : ((1ull << a.ix.fbits) - 1)));
        {
            const uint32_t *__restrict__ bk = a.ix.bkt[la];
            lo[i] = bk[prefix];
            hi[i] = bk[prefix + 1];
        }
    }
    // 2. first two entries of every bucket: 2*NL independent loads
    uint2 e0[NL], e1[NL];
    uint32_t cur[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const uint2 *__restrict__ E = a.ix.ent[LA0 + i];
        e0[i] = (lo[i] < hi[i]) ? E[lo[i]] : make_uint2(0xffffffffu, 0);
        e1[i] = (lo[i] + 1 < hi[i]) ? E[lo[i] + 1] : make_uint2(0xffffffffu, 0);
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        s.addL(1u);
        cur[i] = lo[i];
        if (hi[i] - lo[i] > 16) { // large bucket: lower_bound on the key first
            const uint2 *__restrict__ E = a.ix.ent[LA0 + i];
            uint32_t x = lo[i], y = hi[i];
            while (x < y) {
                uint32_t mid = x + ((y - x) >> 1);
                s.addP(1u);
                if ((E[mid].x >> pbits) < fp[i]) x = mid + 1; else y = mid;
            }
            cur[i] = x;
        }
    }
    // 3. scan in list order, queue the survivors
    qn = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        if (s.p_n != PEND_OVF) {
            const uint2 *__restrict__ E = a.ix.ent[LA0 + i];
            const uint32_t f = fp[i], h = hi[i];
            // top pbits of the read's partner signature s_b (the signature of list 5-la)
            const int lb = 5 - (LA0 + i);
            const int xb = (lb < 3) ? 0 : (lb < 5) ? 1 : 2, xd = (lb == 0) ? 1 : (lb == 1 || lb == 3) ? 2 : 3;
            const uint32_t rp = pbits ? (uint32_t)(((m[xb] << bb) | m[xd]) >> (a.l - pbits)) : 0u;
            uint32_t j = cur[i];
            const uint32_t jstop = j + BIG_T;
            while (j < h) {
                if (j == jstop || qn == MQ) { s.p_n = PEND_OVF; break; } // a whole wave walks this one (match_wave.hip)
                uint2 e;
                if (j == lo[i]) e = e0[i];
                else if (j == lo[i] + 1) e = e1[i];
                else e = E[j];
                s.addP(1u);
                const uint32_t ek = e.x >> pbits;
                if (ek > f) break;
                if (ek == f) {
                    bool keep = true;
                    if (pbits) {
                        // member of the reference's equal range; seed popcount filter (match.hpp:386) on the
                        // partner symbols the entry carries: more than seedkmax known mismatches => rejected
                        // without touching the text (exact: the full count can only be larger)
                        s.addC(1u);
                        const uint32_t x = (e.x & pmask) ^ rp;
                        keep = __popc(((x >> 1) | x) & 0x55555555u) <= a.seedkmax;
                    }
                    if (keep) queue_push(q_pos, q_la, qn, e.y, LA0 + i, pbits != 0 || SCORES || ALL);
                }
                j++;
            }
        }
    }
}

// Fine tables (32-bit signatures, large index): the 16-byte bucket table entry {start, size and partner digest
// of each of the (at most eight) key groups} of a prefix gives the reference's equal range of every list
// directly, and for a range of one entry enough of its partner signature to reject most chance candidates
// without reading them.  The lane enumerates the equal ranges of all lists in list order -- the canonical
// candidate order -- eight entries per round trip, applies the partner filter and queues the survivors; a
// full queue is drained and refilled (the only state across a drain is the enumeration offset).
template <int W, bool SCORES, bool ALL, bool DEFER, int LA0, int LA1, int NP>
__device__ __forceinline__ void match_lists_fine(const MatchArgs &a, LaneState<W, SCORES, ALL, NP> &s, const double *sLL,
                                                 uint32_t *q_pos, uint8_t *q_la)
{
    const uint32_t bb = a.b_bits;
    const uint64_t mb = (bb >= 64) ? ~0ull : ((1ull << bb) - 1);
    const uint64_t m[4] = {s.shi >> bb, s.shi & mb, s.slo >> bb, s.slo & mb};
    constexpr int NL = LA1 - LA0;
    const bool fpk = a.ix.fine == 2; // fingerprint tables: the entries hold a 32-bit key, membership is settled on the text
    const uint32_t pbits = a.ix.pbits, pmask = (1u << pbits) - 1, G = fpk ? 1u : 1u << a.ix.fbits;
    uint32_t lo[NL], cum[NL], rp[NL], total = 0, counted = 0;
    const uint32_t dbits = pbits < 8 ? pbits : 8;
    {
        uint4 t[NL];
        uint32_t f[NL], prefix[NL];
        // 1. one 16-byte table entry per list, all in flight together
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int la = LA0 + i;
            const int xa = (la < 3) ? 0 : (la < 5) ? 1 : 2, xc = (la == 0) ? 1 : (la == 1 || la == 3) ? 2 : 3;
            const uint64_t sa = (m[xa] << bb) | m[xc]; // s_a of list la, SignatureConstruction.hpp:62-67
            prefix[i] = (uint32_t)(sa >> a.ix.pshift);
            f[i] = (uint32_t)sa & (G - 1);
            const int lb = 5 - la; // partner signature s_b = signature of list 5-la
            const int xb = (lb < 3) ? 0 : (lb < 5) ? 1 : 2, xd = (lb == 0) ? 1 : (lb == 1 || lb == 3) ? 2 : 3;
            rp[i] = fpk ? (uint32_t)(sa >> a.ix.fshift) : (uint32_t)(((m[xb] << bb) | m[xd]) >> ((a.l - pbits) & 63u));
            t[i] = reinterpret_cast<const uint4 *>(a.ix.bkt[la])[prefix[i]];
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const uint64_t flo = (uint64_t)t[i].y | ((uint64_t)t[i].z << 32);
            const uint32_t fhi = t[i].w;
            uint32_t start, sz;
            if (fpk) {
                // the entry's 96 bits = count:8, then the 11-bit fingerprints of the bucket's first eight keys
                const uint32_t cnt = (uint32_t)flo & 255u, myfp = rh_fp11(rp[i]);
                uint32_t first = RH_FP_SLOTS, last = 0;
#pragma unroll
                for (int j = 0; j < (int)RH_FP_SLOTS; ++j) {
                    const int b = 8 + 11 * j;
                    const uint32_t fld = (b + 11 <= 64) ? ((uint32_t)(flo >> b) & 0x7ffu)
                                       : (b < 64) ? (((uint32_t)(flo >> b) | (fhi << (64 - b))) & 0x7ffu) : ((fhi >> (b - 64)) & 0x7ffu);
                    if ((uint32_t)j < cnt && fld == myfp) { if (first == RH_FP_SLOTS) first = j; last = j; }
                }
                start = t[i].x + first;
                sz = (first < RH_FP_SLOTS) ? last - first + 1 : 0u;
                if (cnt > RH_FP_SLOTS && cnt < 255u) {
                    // more than eight entries: everything from the first matching slot (or from the ninth entry) to
                    // the end of the bucket is enumerated, the key comparison below sorts it out
                    const uint32_t b0 = first; // RH_FP_SLOTS if none of the first eight matched
                    start = t[i].x + b0;
                    sz = cnt - b0;
                } else if (cnt == 255u) { // a huge bucket: the equal range of the key by binary search
                    const uint2 *__restrict__ E = a.ix.ent[LA0 + i];
                    const uint32_t end = reinterpret_cast<const uint4 *>(a.ix.bkt[LA0 + i])[prefix[i] + 1].x;
                    uint32_t x = t[i].x, y = end;
                    while (x < y) { uint32_t mid = x + ((y - x) >> 1); if (E[mid].x < rp[i]) x = mid + 1; else y = mid; }
                    start = x; y = end;
                    while (x < y) { uint32_t mid = x + ((y - x) >> 1); if (E[mid].x <= rp[i]) x = mid + 1; else y = mid; }
                    sz = x - start;
                }
                s.addP(sz);
            } else {
            // the entry's 96 bits = 8 fields {size:4, digest:8}, field g = key group g of the bucket
            const uint32_t fi = f[i];
            uint32_t off = 0, mine = 0;
            bool sat = false;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const uint32_t fld = (g < 5) ? ((uint32_t)(flo >> (12 * g)) & 0xfffu)
                                   : (g == 5) ? (((uint32_t)(flo >> 60) | (fhi << 4)) & 0xfffu) : ((fhi >> (12 * g - 64)) & 0xfffu);
                if ((uint32_t)g < fi) { off += fld & 15u; sat = sat || ((fld & 15u) == RH_FINE_SAT); }
                if ((uint32_t)g == fi) mine = fld;
            }
            sz = mine & 15u;
            sat = sat || (sz == RH_FINE_SAT);
            start = t[i].x + off;
            if (sat) { // a group of 15 or more entries in front of / at the key: bounds by binary search
                const uint2 *__restrict__ E = a.ix.ent[LA0 + i];
                const uint32_t end = reinterpret_cast<const uint4 *>(a.ix.bkt[LA0 + i])[prefix[i] + 1].x;
                uint32_t x = t[i].x, y = end;
                while (x < y) { uint32_t mid = x + ((y - x) >> 1); if ((E[mid].x >> pbits) < fi) x = mid + 1; else y = mid; }
                start = x; y = end;
                while (x < y) { uint32_t mid = x + ((y - x) >> 1); if ((E[mid].x >> pbits) <= fi) x = mid + 1; else y = mid; }
                sz = x - start;
            }
            counted += sz;
            if (sz == 1 && !sat) {
                // a range of one entry: seed popcount filter (match.hpp:386) on the partner symbols the table
                // knows; more than seedkmax known mismatches => the entry is never read (exact: the full
                // count can only be larger)
                const uint32_t x = (mine >> 4) ^ (rp[i] >> (pbits - dbits));
                if (__popc(((x >> 1) | x) & 0x55u) > a.seedkmax) sz = 0;
            }
            }
            if (sz > BIG_T) s.p_n = PEND_OVF; // a long equal range is walked by a whole wave (match_wave.hip)
            lo[i] = start; cum[i] = total; total += sz;
            s.addL(1u);
        }
    }
    s.addC(counted); s.addP(counted);
    // 2. enumerate, filter, queue, drain
    for (uint32_t kb = 0; kb < total && !(s.p_n == PEND_OVF); kb += MQ) {
        const uint32_t kend = min(total, kb + (uint32_t)MQ);
        uint32_t qn = 0;
        for (uint32_t k0 = kb; k0 < kend; k0 += EB) {
            uint2 e[EB];
            uint32_t li[EB];
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const uint32_t k = k0 + u;
                if (k < kend) {
                    uint32_t i = 0, base = cum[0], l0 = lo[0];
                    const uint2 *E = a.ix.ent[LA0];
#pragma unroll
                    for (int j = 1; j < NL; ++j)
                        if (k >= cum[j]) { i = j; base = cum[j]; l0 = lo[j]; E = a.ix.ent[LA0 + j]; }
                    li[u] = i;
                    e[u] = E[l0 + (k - base)];
                }
            }
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                if (k0 + u < kend) {
                    uint32_t r = rp[0];
#pragma unroll
                    for (int j = 1; j < NL; ++j) if (li[u] == (uint32_t)j) r = rp[j];
                    // seed popcount filter (match.hpp:386) on the partner symbols the entry carries: more than
                    // seedkmax known mismatches => rejected without touching the text (exact: the full count can
                    // only be larger)
                    const uint32_t x = (e[u].x & pmask) ^ r;
                    if (fpk ? (e[u].x == r) : (__popc(((x >> 1) | x) & 0x55555555u) <= a.seedkmax)) {
                        queue_push(q_pos, q_la, qn, e[u].y, LA0 + (int)li[u], !fpk || DEFER);
                    }
                }
            }
        }
        // 3. verify / score / fold in candidate order
        for (uint32_t k = 0; k < qn; ++k)
            process_candidate<W, SCORES, ALL, DEFER>(a, s, sLL, q_pos[k * 64], (uint32_t)q_la[k * 64]);
    }
}

// lists [LA0, LA1) of one strand
template <int W, bool SCORES, bool ALL, bool FINE, bool DEFER, int LA0, int LA1, int NP>
__device__ __forceinline__ void match_lists(const MatchArgs &a, LaneState<W, SCORES, ALL, NP> &s, const double *sLL,
                                            uint32_t *q_pos, uint8_t *q_la)
{
    if (FINE) { match_lists_fine<W, SCORES, ALL, DEFER, LA0, LA1>(a, s, sLL, q_pos, q_la); return; }
    uint32_t qn = 0;
    scan_lists<W, SCORES, ALL, LA0, LA1>(a, s, q_pos, q_la, qn);
    // 4. verify / score / fold in candidate order
    for (uint32_t k = 0; k < qn && s.p_n != PEND_OVF; ++k)
        process_candidate<W, SCORES, ALL, DEFER>(a, s, sLL, q_pos[k * 64], (uint32_t)q_la[k * 64]);
}

// the wave copies bytes [src, src+nbytes) of the batch into its LDS region with 16-byte loads that are
// aligned in global memory and never touch a byte outside the range; returns the LDS offset of src[0]
__device__ __forceinline__ uint32_t stage_wave(uint8_t *stg, const uint8_t *src, uint64_t nbytes, uint32_t lane)
{
    const uintptr_t a0 = (uintptr_t)src, a1 = a0 + nbytes, c0 = a0 & ~(uintptr_t)15;
    for (uintptr_t c = c0 + 16u * lane; c < a1; c += 16u * 64u) {
        const uint32_t lo = STG_PAD + (uint32_t)(c - c0);
        if (c >= a0 && c + 16 <= a1) {
            *reinterpret_cast<uint4 *>(stg + lo) = *reinterpret_cast<const uint4 *>(c);
        } else {
            for (int b = 0; b < 16; ++b)
                if (c + b >= a0 && c + b < a1) stg[lo + b] = *reinterpret_cast<const uint8_t *>(c + b);
        }
    }
    return STG_PAD + (uint32_t)(a0 - c0);
}
// LDS traffic between the lanes of one wave: the hardware keeps a wave's LDS operations in order, the
// compiler must too
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------
// bucket rows (table kind 3): lookups by groups of eight lanes
// ---------------------------------------------------------------------------
// Lists [LA0, LA1) of one strand for ALL lanes of the wave (act = this lane takes part).  A lookup is one
// 128-byte row = ONE line of HBM; it is fetched by eight lanes with one coalesced request (sixteen bytes
// each), eight lookups per load instruction.  The wave works through its 64 lookups of a list in two halves
// of 32: four load instructions, the rows go to LDS, then the 32 owner lanes read their row's directory and
// -- from the same row -- the entries of their key group, apply the partner filter and queue the survivors
// with their positions.  The loads of the next list are in flight while this one is decoded.  The queue is
// drained by every lane for itself, in list order, behind the last list.
// Qualities ahead of time (bucket rows, scores on): a wave whose 64 reads are one staging group, whose quality bytes start
// at a multiple of 16 and fit the place of the rows, has them brought there by LDS-DMA while it drains its queues for the
// last time.  Decided from the offsets of the reads alone, at both places that need to know (nothing is carried along:
// the scalar registers are all in use).
__device__ __forceinline__ bool quals_ahead(const MatchArgs &a, uint64_t o0, uint64_t o1, uint32_t cap, const uint8_t *&src, uint32_t &nbytes, uint32_t &late,
                                            uint32_t &skew)
{
    if (!a.b.qual || a.b.gl < 64) return false;
    const uint64_t gb = __shfl(o0, 0), ge = __shfl(o1, 63);
    // the pieces are 16 bytes at multiples of 16: they start with the `skew` bytes in front of the wave's first quality
    // (bytes of the reads before it; a wave at the very start of an array that is not aligned itself stages as before)
    skew = (uint32_t)((uintptr_t)(a.b.qual + gb) & 15u);
    src = a.b.qual + gb - skew;
    nbytes = (uint32_t)(ge - gb) + skew;
    // up to 8 KiB: everything into the place of the rows.  More (reads of 129 .. 170 bases): the bytes lie from STG_PAD on,
    // as stage_wave() puts them; what comes to lie under the queues -- the first `late` bytes -- follows when the queues
    // have been drained
    late = nbytes <= 64u * 128u ? 0u : ROWBUF_OFF - STG_PAD;
    return ge >= gb && gb >= skew && ge - gb + skew <= cap;
}
// where the first quality of the wave lies in its region
__device__ __forceinline__ uint32_t quals_at(uint32_t late, uint32_t skew) { return (late ? STG_PAD : ROWBUF_OFF) + skew; }

// every byte that quals_ahead()'s LDS-DMA brought is in LDS (and nothing of it can arrive after the wave has ended)
__device__ __forceinline__ void quals_landed()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// qlast (the last call of a read, scores on): while the queues are drained for the last time, the qualities of the wave's
// reads travel into the wave's region if quals_ahead() says so (at quals_at() when the function returns and
// quals_landed() has waited).
template <int W, bool SCORES, bool ALL, bool DEFER, bool WIDE, int LA0, int LA1, int NP>
__device__ __forceinline__ void match_lists_rows(const MatchArgs &a, LaneState<W, SCORES, ALL, NP> &s, const double *sLL, uint8_t *stg, bool act,
                                                 bool qlast = false)
{
    // (queue slots per lane and strand, and where the rows lie behind the queue: the first pass' or the second's)
    constexpr uint32_t MQRn = NP == NPEND ? MQR : MQR2, ROWBUF_OFFn = MQRn * 64u * 5u, BKX_OFFn = ROWBUF_OFFn + 64u * 128u;
    constexpr bool has_pass2_rows = SCORES || ALL; // (bucket rows with parked hits: a second pass stands behind the first)
    constexpr uint32_t BIGn = NP == NPEND ? BIG_T : BIG_T2; // entries of one equal range a lane walks
    constexpr uint32_t OVN = NP == NPEND ? 4u : 16u;         // overflow entries of a complex row requested together
    constexpr int NL = LA1 - LA0;
    uint32_t lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane)); // (not to be carried across the tile loop of the kernel)
    // the wave's LDS region in this mode: MQR x 64 queued positions, MQR x 64 lists, 64 rows of 128 bytes
    uint32_t *q_pos = reinterpret_cast<uint32_t *>(stg) + lane;
    uint8_t *q_la = stg + MQRn * 64 * 4 + lane;
    uint8_t *rowbuf = stg + ROWBUF_OFFn;
    const uint32_t bb = a.b_bits;
    const uint32_t gbits = a.ix.fbits, pbits = a.ix.pbits, p16 = pbits < 16 ? pbits : 16;
    const uint32_t pmask = (1u << pbits) - 1;
    // The signature of list la is two of the four seed segments (SignatureConstruction.hpp:62-67), picked with shifts
    // from two words -- an array indexed by the (uniform, run-time) list number would live in scratch memory.
    // narrow (seedl <= 32): the segments in 16-bit fields of M; bucket = signature >> gbits, key group = its low gbits,
    //   the row key = the leading 16 bits of the partner signature (list 5-la), filtered by symbol mismatches.
    // wide (entries hold a 32-bit key, pbits == 0): the segments in 32-bit fields of MA|MB; bucket = the signature's
    //   leading pb bits, key group = the next four, the row key = a 16-bit fingerprint of the 28 key bits below, compared
    //   for equality (membership in the equal range is settled on the text).
    constexpr bool wide = WIDE; // (a compile-time property of the kernel instance: as a run-time flag it cost the 32-bit path registers)
    uint64_t M = 0, MA = 0, MB = 0;
    if (!wide) {
        const uint32_t mb = (1u << bb) - 1;
        M = ((uint64_t)(uint32_t)(s.shi >> bb) << 48) | ((uint64_t)((uint32_t)s.shi & mb) << 32) | ((uint64_t)(uint32_t)(s.slo >> bb) << 16) |
            (uint64_t)((uint32_t)s.slo & mb);
    } else {
        const uint64_t mb = (bb >= 32) ? 0xffffffffull : ((1ull << bb) - 1);
        MA = ((s.shi >> bb) << 32) | (s.shi & mb);
        MB = ((s.slo >> bb) << 32) | (s.slo & mb);
    }
    auto sig_of = [&](int la) { // s0..s5 = segments (0,1),(0,2),(0,3),(1,2),(1,3),(2,3); narrow
        const uint32_t xa = (0x940u >> (2 * la)) & 3u, xc = (0xfb9u >> (2 * la)) & 3u;
        return (((uint32_t)(M >> (48 - 16 * xa)) & 0xffffu) << bb) | ((uint32_t)(M >> (48 - 16 * xc)) & 0xffffu);
    };
    auto sig_wide = [&](int la) {
        const uint32_t xa = (0x940u >> (2 * la)) & 3u, xc = (0xfb9u >> (2 * la)) & 3u;
        const uint64_t sa_ = ((xa < 2 ? MA : MB) >> (32 * (1 - (xa & 1)))) & 0xffffffffull;
        const uint64_t sc_ = ((xc < 2 ? MA : MB) >> (32 * (1 - (xc & 1)))) & 0xffffffffull;
        return (sa_ << bb) | sc_;
    };
    // the rows are addressed by the MIXED signature (real_hip_internal.h: rh_mix32 / rh_mix64): row = its leading bits,
    // key group = the bits below; the partner key of an entry is plain
    auto msig_of = [&](int la) { return rh_mix32(sig_of(la), a.l); };
    auto msig_wide = [&](int la) { return rh_mix64(sig_wide(la), a.l); };
    auto bucket_of = [&](int la) { return wide ? (uint32_t)(msig_wide(la) >> a.ix.pshift) : (msig_of(la) >> gbits); };
    uint32_t qn = 0, q_last = 0, q_lastla = 0; // queue fill; position and list bits of the entry pushed last
    uint4 va0, va1, va2, va3, va4, va5, va6, va7; // (eight scalars, not an array: the array went through scratch memory)
    uint32_t *bkx = reinterpret_cast<uint32_t *>(stg + BKX_OFFn);
    // the eight loads of list la: lane (8g+j) reads piece j of the row of owner 8*it+g.  The owners' bucket numbers go
    // through LDS transposed, so that a loader finds its eight at bkx[8g .. 8g+7] (an owner that takes no part: row 0)
    // (a macro, not a lambda taking the array by reference: that sent the eight rows through scratch memory)
#define ISSUE_ROWS(LA)                                                                                                      \
    do {                                                                                                                    \
        bkx[(lane & 7) * 8 + (lane >> 3)] = act ? bucket_of(LA) : 0u;                                                       \
        __builtin_amdgcn_wave_barrier(); /* LDS operations of a wave execute in order; a fence would also wait for loads */ \
        const uint4 b0_ = *reinterpret_cast<const uint4 *>(bkx + (lane >> 3) * 8);                                          \
        const uint4 b1_ = *reinterpret_cast<const uint4 *>(bkx + (lane >> 3) * 8 + 4);                                      \
        const uint4 *__restrict__ R_ = reinterpret_cast<const uint4 *>(a.ix.bkt[LA]) + (lane & 7);                          \
        va0 = R_[(uint64_t)b0_.x * 8]; va1 = R_[(uint64_t)b0_.y * 8]; va2 = R_[(uint64_t)b0_.z * 8];                        \
        va3 = R_[(uint64_t)b0_.w * 8]; va4 = R_[(uint64_t)b1_.x * 8]; va5 = R_[(uint64_t)b1_.y * 8];                        \
        va6 = R_[(uint64_t)b1_.z * 8]; va7 = R_[(uint64_t)b1_.w * 8];                                                       \
    } while (0)
    // the last list of the read is decoded, the rows are dead: the qualities of the wave's reads go straight into their place
    // (LDS-DMA: lane i's 16 bytes land at base + 16 i, no registers) while the queues are drained for the last time.
    // (inline assembly: the compiler must not count these among its loads -- it would hold every LDS access of the loop
    // until they have landed; quals_landed() is the wait)
#define QUAL_DMA(SRC_OFF, LDS_ADDR, NBYTES, NCHUNK)                                                                                           \
    _Pragma("unroll") for (uint32_t i = 0; i < (NCHUNK); ++i)                                                                                 \
        if (16u * lane + 1024u * i + 16u <= (NBYTES)) {                                                                                       \
            const uint8_t *g_ = qsrc + (SRC_OFF) + 16u * lane + 1024u * i;                                                                    \
            uint32_t keep_;                                                                                                                   \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"         \
                         : "=&s"(keep_) : "v"(g_), "s"((LDS_ADDR) + 1024u * i) : "memory");                                                   \
        }
#define ISSUE_QUALS()                                                                                                                         \
    do {                                                                                                                                      \
        const uint8_t *qsrc;                                                                                                                  \
        uint32_t qbytes, qlate, qskew;                                                                                                        \
        if (quals_ahead(a, s.o0, s.o1, stg_bytes(W, 3) - STG_PAD - 32u, qsrc, qbytes, qlate, qskew)) {                                             \
            wave_lds_sync();                                                                                                                  \
            const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)rowbuf);   \
            QUAL_DMA(qlate, lds0, qbytes - qlate, 9u)                                                                                         \
            const uint32_t tail0 = qbytes & ~15u; /* (what the last whole 16 bytes leave over) */                                             \
            if (lane < (qbytes & 15u)) rowbuf[tail0 - qlate + lane] = qsrc[tail0 + lane];                                                     \
        }                                                                                                                                     \
    } while (0)
#define ISSUE_QUALS_LATE() /* the queues are drained: what lies in their place */                                                             \
    do {                                                                                                                                      \
        const uint8_t *qsrc;                                                                                                                  \
        uint32_t qbytes, qlate, qskew;                                                                                                        \
        if (quals_ahead(a, s.o0, s.o1, stg_bytes(W, 3) - STG_PAD - 32u, qsrc, qbytes, qlate, qskew) && qlate) {                                     \
            wave_lds_sync();                                                                                                                  \
            const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)stg);      \
            QUAL_DMA(0u, lds0 + STG_PAD, qlate, 3u)                                                                                           \
        }                                                                                                                                     \
    } while (0)
    ISSUE_ROWS(LA0);
#pragma unroll 1
    for (int li = 0; li < NL; ++li) { // (a real loop: the drain below must exist once, not NL times)
        const int la = LA0 + li;
#if RH_PHASE_TIMING
        const unsigned ph0 = PH_NOW();
#endif
        // rows -> LDS: piece j of row r at r * 128 + ((j ^ (r & 7)) * 16) (16-byte stores; the swizzle spreads the owners'
        // reads of the same dword of different rows over eight bank groups)
        wave_lds_sync();
        {
            // lane (8g+j) holds piece j of the rows 8*it+g: (8*it+g) & 7 == g, so the swizzle does not depend on it
            uint8_t *d = rowbuf + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) * 16);
            *reinterpret_cast<uint4 *>(d + 0 * 1024) = va0; *reinterpret_cast<uint4 *>(d + 1 * 1024) = va1;
            *reinterpret_cast<uint4 *>(d + 2 * 1024) = va2; *reinterpret_cast<uint4 *>(d + 3 * 1024) = va3;
            *reinterpret_cast<uint4 *>(d + 4 * 1024) = va4; *reinterpret_cast<uint4 *>(d + 5 * 1024) = va5;
            *reinterpret_cast<uint4 *>(d + 6 * 1024) = va6; *reinterpret_cast<uint4 *>(d + 7 * 1024) = va7;
        }
        wave_lds_sync();
#if RH_PHASE_TIMING
        unsigned ph1 = PH_NOW();
        s.tW += ph1 - ph0;
#endif
        // owners: directory of the row, then the entries of their key group
        const bool mine = act && !(s.p_n == PEND_OVF) && !(RH_ABLATE & 8);
        const uint8_t *rowb = rowbuf + lane * 128;
        const uint32_t sw = lane & 7;
        auto row = [&](uint32_t d) { return *reinterpret_cast<const uint32_t *>(rowb + ((((d >> 2) ^ sw) << 4) | ((d & 3) << 2))); };
        uint32_t e_cnt = 0, e_j = 0, e_base = 0;
        bool e_ovf = false;
        uint32_t g, r; // key group; what an entry's key is compared with (narrow: leading pbits of s_b; wide: the 32-bit key)
        if (!wide) {
            g = msig_of(la) & ((1u << gbits) - 1);
            r = sig_of(5 - la) >> (a.l - pbits);
        } else {
            r = (uint32_t)(msig_wide(la) >> a.ix.fshift);
            g = r >> 28;
        }
        if (mine) {
            s.addL(1u);
            const uint32_t h0 = row(0), h1 = row(1);
            if ((h0 & h1) != 0xffffffffu) {
                // sixteen 4-bit counts: mine, and the sum of those in front of it
                e_cnt = ((g < 8 ? h0 : h1) >> (4 * (g & 7))) & 15u;
                const uint32_t m0 = g < 8 ? (g ? (h0 & (0xffffffffu >> (32 - 4 * g))) : 0u) : h0;
                const uint32_t m1 = g > 8 ? (h1 & (0xffffffffu >> (32 - 4 * (g - 8)))) : 0u;
                const uint32_t b0 = (m0 & 0x0f0f0f0fu) + ((m0 >> 4) & 0x0f0f0f0fu), b1 = (m1 & 0x0f0f0f0fu) + ((m1 >> 4) & 0x0f0f0f0fu);
                e_base = __builtin_amdgcn_sad_u8(b0, 0u, __builtin_amdgcn_sad_u8(b1, 0u, 0u));
            } else { // complex bucket: its entries are in the overflow array; 8-bit group counts
                e_ovf = true;
                // sixteen 8-bit counts in four words: the sum of those in front of mine, mine, and whether any of them
                // is saturated (byte-parallel: this branch runs whenever one of the 64 lanes meets such a bucket)
                uint32_t off = 0, sat255 = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t w = row(4 + q);
                    const int nb = (int)g - 4 * q;                                     // counts of this word in front of mine
                    const uint32_t below = nb <= 0 ? 0u : (nb >= 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1u));
                    const uint32_t upto = nb < 0 ? 0u : (nb >= 3 ? 0xffffffffu : ((1u << (8 * (nb + 1))) - 1u)); // ... and mine
                    off = __builtin_amdgcn_sad_u8(w & below, 0u, off);
                    if ((g >> 2) == (uint32_t)q) e_cnt = (w >> (8 * (g & 3))) & 255u;
                    const uint32_t y = ~w | ~upto;                                     // a zero byte = a count of 255 among them
                    sat255 |= (y - 0x01010101u) & ~y & 0x80808080u;
                }
                const bool sat = sat255 != 0;
                const uint32_t o0 = row(2), tot = row(3);
                e_base = o0 + off;
                if (sat) { // a group of 255 or more entries in front of / at the key: bounds by binary search
                    const uint2 *__restrict__ E = a.ix.ent[la];
                    uint32_t x = o0, y = o0 + tot;
                    const uint32_t end = y;
                    const uint32_t gs = wide ? 28u : pbits;
                    while (x < y) { uint32_t mid = x + ((y - x) >> 1); if ((E[mid].x >> gs) < g) x = mid + 1; else y = mid; }
                    e_base = x; y = end;
                    while (x < y) { uint32_t mid = x + ((y - x) >> 1); if ((E[mid].x >> gs) <= g) x = mid + 1; else y = mid; }
                    e_cnt = x - e_base;
                }
            }
            if (!wide) s.addC(e_cnt); // (wide: counted when the text confirms the membership)
            s.addP(e_cnt);
            // a long equal range is not walked by a lane of the first pass (it would hold its wave): the second pass takes it up to
            // BIG_T2 entries, beyond that a whole wave does (match_wave.hip)
            if (e_cnt > BIGn) { s.p_n = PEND_OVF; s.cslot = (NP == NPEND && has_pass2_rows && DEFER && e_cnt <= BIG_T2) ? SLOT_NONE : SLOT_BIG; e_cnt = 0; }
            if (RH_ABLATE & 4) e_cnt = 0;
        }
        const uint32_t rk = wide ? rh_fp16(r) : (r >> (pbits - p16)); // what the 16 key bits of a row entry are compared with
        // the entries of the lanes' key groups, one per lane and step, without branches on the way: a survivor of the
        // partner filter goes to the lane's queue; a window that is in the queue already only gets this list's bit set
        // (the lane's last queue entry is kept in registers: for it nothing is read back).  A lane that needs
        // more than MQR queue slots for one strand (repeat-rich loci only) hands its read over: the queues are drained
        // once, behind the lists, where the registers that hold the rows in flight are free again.
        // A lane whose row is complex finds its entries in the overflow array: their first two are requested here and looked
        // at behind the lanes with simple rows -- the round trip is hidden behind their work instead of standing in
        // front of every step of the whole wave (with a skewed base composition nearly every wave has such a lane in every
        // list).  A lane is of one kind for the whole list, so the order of its own entries is what it was.
        U64x2 pre = {0ull, 0ull};
        if (e_ovf && e_cnt) pre = load2(reinterpret_cast<const uint64_t *>(a.ix.ent[la] + e_base)); // (two entries; the array is padded by one)
        // the next list's rows are in flight while this one is decoded and drained (requested behind the two entries above:
        // loads come back in the order they were issued; in front of them it measured the same)
        if (li + 1 < NL) ISSUE_ROWS(la + 1);
        // a survivor of the partner filter goes to the lane's queue; a window that is in the queue already only gets this
        // list's bit set (the lane's last queue entry is kept in registers: for it nothing is read back).  A lane that needs
        // more than MQRn queue slots for one strand (repeat-rich loci only) hands its read over: the queues are drained
        // once, behind the lists, where the registers that hold the rows in flight are free again.
        auto take = [&](bool pass, uint32_t pos) {
            const bool merge = pass && qn && pos == q_last;
            if ((!wide || DEFER) && pass && !merge && qn >= 2) { // (repeats: a window that stands further up in the queue gets this list's bit there; see queue_push)
                for (uint32_t k = 0; k + 1 < qn; ++k)
                    if (q_pos[k * 64] == pos) { q_la[k * 64] |= (uint8_t)(1u << la); pass = false; break; }
            }
            if (pass && !merge && qn == MQRn) { s.p_n = PEND_OVF; e_cnt = 0; pass = false; } // (the read is handed over)
            const uint32_t slot = merge ? qn - 1 : qn;
            const uint32_t lav = (merge ? q_lastla : 0u) | (1u << la);
            if (pass) { q_pos[slot * 64] = pos; q_la[slot * 64] = (uint8_t)lav; q_last = pos; q_lastla = lav; qn = slot + 1; }
        };
        // seed popcount filter (match.hpp:386) on the partner symbols the entry carries: more than seedkmax
        // known mismatches => rejected without touching the text (exact: the full count can only be larger)
        auto passes = [&](uint32_t x) { return wide ? (x == 0u) : (__popc(((x >> 1) | x) & 0x55555555u) <= a.seedkmax); };
        // Simple rows.  Nearly every entry that is no copy of the read's locus fails the filter (8 partner symbols, at most
        // seedkmax mismatches: 0.4 % of the chance entries pass), so the entries are filtered first -- four keys of the lane's
        // group per step, their LDS reads in flight together, nothing but a bit per entry kept -- and only the survivors,
        // one or two per lane, take the path through the queue.  (The steps of the wave are those of its lane with the most
        // entries: a quarter of them this way, which is what a skewed base composition, with its long groups, needs.)
        uint32_t passm = 0; // bit j: entry j of the lane's group survives (a group of a simple row has at most 15 entries)
        while (__any(!e_ovf && e_j < e_cnt)) {
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                const uint32_t jj = e_j + u;
                const bool ok = !e_ovf && jj < e_cnt;
                const uint32_t h = ok ? 4 + 3 * (e_base + jj) : 4u; // the key: halfword 4 + 3 * entry of the row
                const uint32_t d0 = row(h >> 1);
                const uint32_t key = (h & 1) ? (d0 >> 16) : (d0 & 0xffffu);
                passm |= (ok && passes(key ^ rk) ? 1u : 0u) << (jj & 31u);
            }
            e_j += 4;
        }
        // more survivors in one list than a lane of the first pass parks locations: a read on that many copies of its
        // locus.  It is handed on here, before anything of it is verified for nothing (handing over is always allowed: who
        // takes the read does all of it).
        if (NP == NPEND && has_pass2_rows && DEFER && __popc(passm) > NPEND) { s.p_n = PEND_OVF; passm = 0; e_cnt = 0; }
        while (__any(passm != 0)) {
            const bool step = passm != 0;
            const uint32_t jj = step ? (uint32_t)__ffs((int)passm) - 1u : 0u;
            passm &= passm - 1u;
            const uint32_t h = step ? 4 + 3 * (e_base + jj) : 4u; // the position: the two halfwords behind the key
            const uint32_t d0 = row(h >> 1), d1 = row((h >> 1) + 1);
            take(step, (h & 1) ? d1 : ((d0 >> 16) | (d1 << 16)));
        }
        // complex rows: {key, pos} in the overflow array.  The first two entries are there (requested above); the rest comes
        // OVN at a time (four; sixteen in the second pass), their loads in flight together -- one round trip per OVN entries, not one per entry
        auto take_ovf = [&](bool ok, uint2 e) { take(ok && passes(wide ? (e.x ^ r) : ((e.x & pmask) ^ r)), e.y); };
        if (__any(e_ovf && e_cnt)) {
            take_ovf(e_ovf && e_cnt > 0, make_uint2((uint32_t)pre.a, (uint32_t)(pre.a >> 32)));
            take_ovf(e_ovf && e_cnt > 1, make_uint2((uint32_t)pre.b, (uint32_t)(pre.b >> 32)));
        }
        e_j = 2;
        while (__any(e_ovf && e_j < e_cnt)) {
            uint2 e[OVN];
#pragma unroll
            for (uint32_t u = 0; u < OVN; ++u) e[u] = (e_ovf && e_j + u < e_cnt) ? a.ix.ent[la][e_base + e_j + u] : make_uint2(0u, 0u);
#pragma unroll
            for (uint32_t u = 0; u < OVN; ++u) take_ovf(e_ovf && e_j + u < e_cnt, e[u]);
            e_j += OVN;
        }
#if RH_PHASE_TIMING
        s.tD += PH_NOW() - ph1;
#endif
    }
    if (SCORES && qlast) ISSUE_QUALS();
#if RH_PHASE_TIMING
    const unsigned ph2 = PH_NOW();
#endif
    // verify / score / fold in candidate order; the text of the next candidate is on its way while this one is looked at
    if (s.p_n == PEND_OVF) qn = 0;
    {
        CandText<W> c0;
        uint32_t rp0 = qn ? q_pos[0] : 0u;
        if (qn) cand_load<W, SCORES, ALL>(a, s, rp0, c0);
        for (uint32_t k = 0; k < qn; ++k) {
            CandText<W> c1;
            uint32_t rp1 = 0;
            if (k + 1 < qn) { rp1 = q_pos[(k + 1) * 64]; cand_load<W, SCORES, ALL>(a, s, rp1, c1); }
            process_loaded<W, SCORES, ALL, DEFER>(a, s, sLL, rp0, (uint32_t)q_la[k * 64], c0);
            c0 = c1; rp0 = rp1;
        }
    }
#if RH_PHASE_TIMING
    s.tR += PH_NOW() - ph2;
#endif
    if (SCORES && qlast) ISSUE_QUALS_LATE();
#undef ISSUE_QUALS
#undef ISSUE_QUALS_LATE
#undef QUAL_DMA
}

// both strands of one read with bucket rows: every lane of the wave comes along, `act` tells which ones have a read
template <int W, bool SCORES, bool ALL, bool DEFER, bool WIDE, int NP>
__device__ __forceinline__ void match_read_rows(const MatchArgs &a, LaneState<W, SCORES, ALL, NP> &s, const double *sLL, uint8_t *stg, bool act, bool quals_dma)
{
    const uint32_t patl = act ? s.patl : 32u * W;
    s.p_n = 0; s.cslot = SLOT_NONE;
#pragma unroll
    for (int j = 0; j < NP; ++j) { s.p_pos[j] = 0; s.p_meta[j] = 0; }
    uint64_t rhi = 0, rlo = 0;
    if (act) seed_halves<W>(s.O, a.l, s.shi, s.slo, rhi, rlo);
    else { s.shi = s.slo = 0; }
    for (int inv = 0; inv < 2; ++inv) {
        if (inv && act) { // transposed pattern, Pattern.hpp:105-128
            uint64_t R[W];
            revcomp_words<W>(s.O, R, patl);
#pragma unroll
            for (int j = 0; j < W; ++j) s.O[j] = R[j];
            s.shi = rhi; s.slo = rlo;
        }
        s.inv = inv;
        s.cpos = 0xffffffffu; s.ck = 0; s.cfrag = 0; s.cscore = 1.0f; s.cok = false;
        s.crpos = 0xffffffffu; s.ckk = 0;
        const bool go = act && !(s.p_n == PEND_OVF);
        if (!ALL && !SCORES) {
            // uni0s / uni0r early-out (matchUniqueImplementation.cpp:434-436, 470-472): lists 1..5 of a
            // strand are skipped when list 0 left the record in this strand's state with 0 errors
            match_lists_rows<W, SCORES, ALL, DEFER, WIDE, 0, 1>(a, s, sLL, stg, go);
            const unsigned st = (unsigned)(s.info >> ST_SHIFT), er = (unsigned)(s.info >> ER_SHIFT) & 15;
            match_lists_rows<W, SCORES, ALL, DEFER, WIDE, 1, 6>(a, s, sLL, stg, go && !(st == (unsigned)(inv ? ST_REVERSE : ST_STRAIGHT) && er == 0));
        } else {
            match_lists_rows<W, SCORES, ALL, DEFER, WIDE, 0, 6>(a, s, sLL, stg, go, quals_dma && inv != 0);
        }
    }
}

// both strands of one read (UniqueMatcher::match / AllMatcher::match, matchUniqueImplementation.cpp:396-500,
// matchAllImplementation.cpp:261-355): s.O holds the read as given on entry, its reverse complement on exit
template <int W, bool SCORES, bool ALL, bool FINE, bool DEFER, int NP>
__device__ __forceinline__ void match_read(const MatchArgs &a, LaneState<W, SCORES, ALL, NP> &s, const double *sLL, uint32_t *q_pos,
                                           uint8_t *q_la)
{
    const uint32_t patl = s.patl;
    s.p_n = 0; s.cslot = SLOT_NONE;
#pragma unroll
    for (int j = 0; j < NP; ++j) { s.p_pos[j] = 0; s.p_meta[j] = 0; }
    uint64_t rhi, rlo;
    seed_halves<W>(s.O, a.l, s.shi, s.slo, rhi, rlo);
    for (int inv = 0; inv < 2; ++inv) {
        if (s.p_n == PEND_OVF) break;
        if (inv) { // transposed pattern, Pattern.hpp:105-128
            uint64_t R[W];
            revcomp_words<W>(s.O, R, patl);
#pragma unroll
            for (int j = 0; j < W; ++j) s.O[j] = R[j];
            s.shi = rhi; s.slo = rlo;
        }
        s.inv = inv;
        s.cpos = 0xffffffffu; s.ck = 0; s.cfrag = 0; s.cscore = 1.0f; s.cok = false;
        s.crpos = 0xffffffffu; s.ckk = 0;
        if (!ALL && !SCORES) {
            // uni0s / uni0r early-out (matchUniqueImplementation.cpp:434-436, 470-472): lists 1..5 of a
            // strand are skipped when list 0 left the record in this strand's state with 0 errors
            match_lists<W, SCORES, ALL, FINE, DEFER, 0, 1>(a, s, sLL, q_pos, q_la);
            const unsigned st = (unsigned)(s.info >> ST_SHIFT), er = (unsigned)(s.info >> ER_SHIFT) & 15;
            if (!(st == (unsigned)(inv ? ST_REVERSE : ST_STRAIGHT) && er == 0))
                match_lists<W, SCORES, ALL, FINE, DEFER, 1, 6>(a, s, sLL, q_pos, q_la);
        } else {
            match_lists<W, SCORES, ALL, FINE, DEFER, 0, 6>(a, s, sLL, q_pos, q_la);
        }
    }
}

// The matcher proper: lane i of the grid takes read i of the batch.  A wave reads the bases of its 64 reads -- one
// contiguous byte range of the caller's array -- through LDS, every lane packs its own read into registers, matches
// both strands, and parks its hits (DEFER: scores on, or matchAll); then the wave reads the qualities of its reads the
// same way and the lanes score and deliver together.  A read that needs more than NPEND locations or more queue slots
// than a lane has is left untouched and its index appended to a.ovf_list: the second pass (PASS2: this kernel again, over
// that list, with NPEND2 locations and MQR2 queue slots per lane; its reads are scattered, so every lane fetches the bytes
// of its own read) does it all and counts it.  What outgrows that as well, a read that meets an equal range of more than
// BIG_T entries, and a read longer than a lane's registers go to a.ovf2_list: the wave-per-read matcher (match_wave.hip).
// TK = kind of the bucket tables: 0 bucket starts, 1 directory entries (digests / fingerprints), 3 bucket rows,
// 4 bucket rows of signatures wider than 32 bits
template <int TK, bool SCORES, bool ALL>
__host__ __device__ constexpr bool has_pass2() { return TK >= 3 && (SCORES || ALL); }

template <int W, bool SCORES, bool ALL, int TK, bool PASS2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PASS2 ? 2 : (W <= 5 ? 3 : 2)))) void match_kernel(MatchArgs a_)
{
    constexpr bool FINE = TK != 0;
    constexpr bool DEFER = SCORES || ALL;
    constexpr int NP = PASS2 ? NPEND2 : NPEND;
    constexpr uint32_t STG = PASS2 ? stg_bytes2() : stg_bytes(W, TK);
    static_assert(!PASS2 || has_pass2<TK, SCORES, ALL>(), "the second pass exists for bucket rows with parked hits only");
    __shared__ double sLL[SCORES ? RH_LL_SLOTS : 1];
    __shared__ __attribute__((aligned(16))) uint8_t smem[4 * STG];
    if (SCORES) { // the score table, once per workgroup
        for (int i = threadIdx.x; i < 1024; i += 256) sLL[i] = a_.LL[i];
        if (threadIdx.x == 0) sLL[RH_LL_ZERO] = 0.0;
        __syncthreads();
    }
    // The grid is as many workgroups as the device holds at a time.  Every wave works through tiles of 64 reads on its own (no
    // workgroup barrier below this line) and takes its next tile from a counter when it is done -- the number of the tile
    // after this one is requested at the start of the tile, it is there at its end.  With one workgroup per 256 reads the
    // wave slots of a workgroup stay empty until its slowest wave has finished and the next workgroup has been
    // dispatched (16 % of the slot time on the C2 workload); with a fixed share of tiles per wave the grid waits for the
    // slowest wave at the end (11 %).
    const uint64_t n_items = PASS2 ? (uint64_t)*a_.ovf_count : a_.b.n_reads; // (second pass: the reads the first one listed)
    const uint64_t n_tiles = (n_items + 63) / 64;
    uint32_t *const tile_ctr = PASS2 ? a_.tile_ctr2 : a_.tile_ctr;
    uint32_t next_raw = 0; // (lane 0 holds the answer of the counter)
    if ((threadIdx.x & 63) == 0) next_raw = atomicAdd(tile_ctr, 1u);
    while (true) {
    const uint64_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)next_raw);
    if (tile >= n_tiles) break;
    if ((threadIdx.x & 63) == 0) next_raw = atomicAdd(tile_ctr, 1u);
    // (what depends on the lane number only is derived again for every tile: carried across the loop it would take
    // registers of all the rest)
    uint32_t tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    // ... and the kernel's arguments are read from the kernel argument segment where they are used (scalar loads that hit
    // the scalar cache), tile by tile: all of them hoisted in front of the loop do not fit the scalar registers
    // (MatchArgs is the kernel's only parameter: it lies at offset 0 of the segment)
    const __attribute__((address_space(4))) MatchArgs *pa = (const __attribute__((address_space(4))) MatchArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(pa));
    const MatchArgs &a = *(const MatchArgs *)pa;
    const uint32_t lane = tid & 63;
    uint8_t *stg = smem + (tid >> 6) * STG;
    uint32_t *q_pos = reinterpret_cast<uint32_t *>(stg) + lane;
    uint8_t *q_la = stg + MQ * 64 * 4 + lane;
    const unsigned ph_start = PH_NOW();
    LaneState<W, SCORES, ALL, NP> s;
    s.cA = s.cB = s.cPw = s.cCw = 0;
#if RH_PHASE_TIMING
    s.tW = s.tD = s.tR = 0;
    unsigned tQ = 0, tS = 0;
#endif
    s.info = 0; s.iscore = 0.f; s.o0 = s.o1 = 0;
    unsigned cR = 0;
    const uint64_t n = a.b.n_reads;

    const uint64_t item = tile * 64 + lane;
    const bool have = item < n_items;
    const uint64_t r = PASS2 ? (have ? (uint64_t)a.ovf_list[item] : n) : item;
    const uint64_t rc = r < n ? r : n; // lanes behind the batch: an empty range at its end
    const uint64_t o0 = a.b.off ? a.b.off[rc] : rc * (uint64_t)a.b.upatl;
    const uint64_t o1 = r < n ? (a.b.off ? a.b.off[r + 1] : o0 + a.b.upatl) : o0;
    // (the span in 64 bits first: offsets that run backwards or jump by 2^32 and more must not alias to a short read)
    const bool span_bad = o1 < o0 || o1 - o0 > (uint64_t)REAL_HIP_MAX_PATL_LONG;
    const uint32_t patl = span_bad ? 0u : (uint32_t)(o1 - o0);
    const uint32_t bsh = a.b.packed ? 2u : 0u; // packed bases: four per byte
    const uint32_t GL = a.b.gl; // reads the wave stages at a time: GL * max_patl fits its LDS region
    const uint32_t stg_cap = STG - STG_PAD - 32u; // bytes of a group the region holds (skew, pad, over-read)
    bool elig = false, toolong = false, give = false;
    if (PASS2) {
        // ---- bases: every lane reads the bytes of its own read (the first pass found it eligible: it fits the registers and
        // holds no symbol > 3 that a flag or the packing would have shown)
        if (r < n) {
            if (a.b.packed) {
                pack_read_packed<W>(GlobalRow{a.b.bases + (o0 >> 2)}, patl, (uint32_t)o0 & 3u, s.O);
                elig = true;
            } else {
                elig = pack_read<W>(GlobalRow{a.b.bases + o0}, patl, s.O);
            }
        }
    } else {
    // ---- bases: global -> LDS -> registers
    // A read longer than the registers of this instance hold (32 W bases: longer than REAL_HIP_MAX_PATL, or than the
    // bound the caller declared), and every read of a group whose bytes do not fit the wave's LDS region because of such
    // a neighbour, is not staged at all: it is given to the wave-per-read matcher, which reads it from LDS words and
    // checks its eligibility itself.  Only offsets that run backwards or a read beyond REAL_HIP_MAX_PATL_LONG are errors.
    for (uint32_t g = 0; g < 64; g += GL) {
        const uint64_t gb = __shfl(o0, (int)g), ge = __shfl(o1, (int)(g + GL - 1));
        const bool fits = ge >= gb && ge - gb <= stg_cap;
        wave_lds_sync();
        uint32_t l0 = 0;
        if (fits) l0 = stage_wave(stg, a.b.bases + (gb >> bsh), a.b.packed ? ((ge + 3) >> 2) - (gb >> 2) : ge - gb, lane);
        wave_lds_sync();
        const bool in_group = lane >= g && lane < g + GL && r < n;
        if (in_group && span_bad) toolong = true;
        else if (in_group && (!fits || patl > 32u * W)) give = patl >= a.l;
        if (in_group && fits && !span_bad && patl >= a.l && patl <= 32u * W) { // matchUniqueImplementation.cpp:376-394
            if (a.b.packed) {
                pack_read_packed<W>(LdsRow{stg, l0 + (uint32_t)((o0 >> 2) - (gb >> 2))}, patl, (uint32_t)o0 & 3u, s.O);
                elig = !(a.b.nflags && ((a.b.nflags[r >> 3] >> (r & 7)) & 1)); // a read with a symbol > 3 is flagged, not packed
            } else {
                elig = pack_read<W>(LdsRow{stg, l0 + (uint32_t)(o0 - gb)}, patl, s.O);
            }
        }
    }
    }
    if (toolong) atomicOr(a.err_flags, 1u); // the host turns this into REAL_HIP_E_INVALID
    wave_lds_sync();
    // ---- match
    s.r = r; s.patl = patl; s.p_n = 0; s.nhit = 0;
    const bool fresh = !PASS2 && a.b.fresh; // (the first pass has started the records of the reads it hands on)
    if (elig && !ALL && !fresh) {
        s.info = a.info[r];
        if (SCORES) s.iscore = a.score[r];
    }
    if (!ALL && fresh) s.iscore = -3.402823466e+38f; // uniqueinfo(numpat): NoMatch (info 0), score -FLT_MAX (UniqueMatchInfo.hpp:191)
    const unsigned ph_front = PH_NOW();
    s.o0 = o0; s.o1 = o1;
    if (TK >= 3) // bucket rows: lookups by lane groups, the whole wave comes along
        match_read_rows<W, SCORES, ALL, DEFER, TK == 4>(a, s, sLL, stg, elig, !PASS2);
    else if (elig)
        match_read<W, SCORES, ALL, FINE, DEFER>(a, s, sLL, q_pos, q_la);
    const bool ovf = give || (elig && s.p_n == PEND_OVF);
    if (ovf) {
        // nothing of this read has been delivered: whoever takes it does it all and counts it.  The first pass gives the
        // reads that were too much for a lane to the second (where there is one); long equal ranges are not walked by lanes
        // at all, and reads a lane cannot hold, or the second pass cannot either, are the wave-per-read matcher's.
        const bool to_wave = PASS2 || !has_pass2<TK, SCORES, ALL>() || give || s.cslot == SLOT_BIG;
        if (to_wave) {
            const unsigned long long slot = wave_append_slot(a.ovf2_count);
            a.ovf2_list[slot] = (uint32_t)r;
        } else {
            const unsigned long long slot = wave_append_slot(a.ovf_count);
            a.ovf_list[slot] = (uint32_t)r;
        }
        s.cA = s.cB = s.cPw = s.cCw = 0;
    }
    // ---- qualities: global -> LDS; score the parked hits and deliver them
    const uint8_t *qsrc_ = nullptr;
    uint32_t qbytes_ = 0, qlate_ = 0, qskew_ = 0;
    const bool q_ahead = !PASS2 && TK >= 3 && SCORES && quals_ahead(a, o0, o1, stg_cap, qsrc_, qbytes_, qlate_, qskew_); // (then match_lists_rows has started them)
    if (q_ahead) quals_landed();
    if (DEFER && PASS2) {
        // The reads of a tile are scattered: every lane copies the qualities of its own read into its 128-byte slot of the row
        // buffer (free now), whole aligned dwords (what they hold besides the read's bytes lies in the same mapped words of
        // the caller's array and is not looked at); a read that does not fit its slot is scored from global memory.
        const bool want = elig && !ovf && s.p_n;
        const bool in_lds = !SCORES || !a.b.qual || patl <= 104u;
        uint8_t *slot = stg + MQR2 * 64u * 5u + lane * 128u; // 16 bytes of slack in front (LdsRow reads whole dwords around a byte), the bytes, slack
        uint32_t head = 0;
        wave_lds_sync();
        if (want && SCORES && a.b.qual && in_lds) {
            const uint8_t *q0 = a.b.qual + o0;
            head = (uint32_t)((uintptr_t)q0 & 3u);
            const uint32_t *__restrict__ src = reinterpret_cast<const uint32_t *>(q0 - head);
            const uint32_t nd = (head + patl + 3u) >> 2;
            for (uint32_t k = 0; k < nd; ++k) reinterpret_cast<uint32_t *>(slot + 16)[k] = src[k];
        }
        wave_lds_sync();
        if (want && in_lds) flush_pending<W, SCORES, ALL>(a, s, sLL, LdsRow{stg, (uint32_t)(slot - stg) + 16u + head});
        else if (want) flush_pending<W, SCORES, ALL>(a, s, sLL, GlobalRow{a.b.qual + o0});
    } else if (DEFER) {
        for (uint32_t g = 0; g < 64; g += GL) {
            const uint64_t gb = __shfl(o0, (int)g), ge = __shfl(o1, (int)(g + GL - 1));
            const bool mine = lane >= g && lane < g + GL && elig && !ovf && s.p_n;
            if (!__any(mine)) continue;
            uint32_t l0 = 0;
#if RH_PHASE_TIMING
            const unsigned pq0 = PH_NOW();
#endif
            if (q_ahead) {
                wave_lds_sync();
                l0 = quals_at(qlate_, qskew_);
            } else {
                wave_lds_sync();
                if (SCORES && a.b.qual) l0 = stage_wave(stg, a.b.qual + gb, ge - gb, lane); // (fits: the bases of this group did)
                wave_lds_sync();
            }
#if RH_PHASE_TIMING
            const unsigned pq1 = PH_NOW();
            tQ += pq1 - pq0;
#endif
            if (mine) flush_pending<W, SCORES, ALL>(a, s, sLL, LdsRow{stg, l0 + (uint32_t)(o0 - gb)});
#if RH_PHASE_TIMING
            tS += PH_NOW() - pq1;
#endif
        }
    }
    if (elig && !ovf) {
        cR = 1;
        if (!ALL) {
            a.info[r] = s.info;
            if (SCORES) a.score[r] = s.iscore;
        }
    } else if (!ALL && fresh && r < n) { // skipped or handed over: the record starts here (whoever takes the read folds into it)
        a.info[r] = 0;
        if (SCORES) a.score[r] = -3.402823466e+38f;
    }
    if (ALL && r < n && (!PASS2 || (elig && !ovf))) a.hit_cnt[r] = (elig && !ovf) ? s.nhit : 0u; // (a handed-over read: who takes it writes it)

    // work counters: wave reduction, one atomic per wave and counter
#if RH_PHASE_TIMING
    const bool l0_ = lane == 0;
    unsigned c[8] = {cR, l0_ ? ph_front - ph_start : 0u, l0_ ? s.tW : 0u, l0_ ? s.tD : 0u, l0_ ? s.tR : 0u, l0_ ? tQ : 0u, l0_ ? tS : 0u, l0_ ? PH_NOW() - ph_start : 0u};
    constexpr int NC = 8;
#else
    // ([7]: reads matched outside the first pass -- real_hip_counters.handed_over)
    unsigned c[8] = {cR, s.cA & 15u, s.getP(), s.getC(), s.cA >> 16, s.cB >> 22, (s.cA >> 4) & 4095u, PASS2 ? cR : 0u};
    constexpr int NC = PASS2 ? 8 : 7;
    (void)ph_start; (void)ph_front;
#endif
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const unsigned v = wave_sum(c[k]);
        if (lane == 0 && v)
            atomicAdd(a.counters + (size_t)(tile & (RH_CSTRIPES - 1)) * 16 + k, (unsigned long long)v);
    }
    } // tiles
}

// ---------------------------------------------------------------------------
// launcher of the instances of ONE read width W = RH_W (this file is compiled once per width, see the Makefile:
// eight translation units build in parallel instead of one for minutes)
// ---------------------------------------------------------------------------
#ifndef RH_W
#error "compile with -DRH_W=<1..10>"
#endif
// as many workgroups as the device holds of this instance at a time (its registers and LDS decide), at most one per 256 reads
static void launch_resident(void (*kernel)(MatchArgs), real_hip_ctx *ctx, const MatchArgs &a)
{
    int per_cu = 0, dev = 0, n_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 1) n_cu = 256;
    static const bool full_grid = getenv("REAL_HIP_FULL_GRID") != nullptr; // (experiments: one workgroup per 256 reads, every wave does one tile)
    const uint64_t want = (a.b.n_reads + 255) / 256, have = full_grid ? want : (uint64_t)per_cu * (uint64_t)n_cu; // (second pass: its reads are a subset)
    dim3 grid((unsigned)(want < have ? want : have)), block(256);
    hipLaunchKernelGGL(kernel, grid, block, 0, ctx->stream, a);
}

template <int W, int TK, bool PASS2>
static void launch_match_wt(real_hip_ctx *ctx, const MatchArgs &a, bool all)
{
    const bool sc = ctx->prm.scores != 0;
    if (all) {
        if (sc) launch_resident(match_kernel<W, true, true, TK, PASS2>, ctx, a);
        else if (!PASS2 || has_pass2<TK, false, true>()) launch_resident(match_kernel<W, false, true, TK, PASS2 && has_pass2<TK, false, true>()>, ctx, a);
    } else {
        if (sc) launch_resident(match_kernel<W, true, false, TK, PASS2>, ctx, a);
        else if (!PASS2) launch_resident(match_kernel<W, false, false, TK, false>, ctx, a);
    }
}
#define RH_CAT2(a, b) a##b
#define RH_CAT(a, b) RH_CAT2(a, b)
void RH_CAT(rh_launch_match_w, RH_W)(real_hip_ctx *ctx, const MatchArgs &a, bool all)
{
    if (a.ix.fine == 3 && !a.ix.pbits) launch_match_wt<RH_W, 4, false>(ctx, a, all);
    else if (a.ix.fine == 3) launch_match_wt<RH_W, 3, false>(ctx, a, all);
    else if (a.ix.fine) launch_match_wt<RH_W, 1, false>(ctx, a, all);
    else launch_match_wt<RH_W, 0, false>(ctx, a, all);
}
// the second pass over the reads the first one listed (bucket rows, parked hits: scores on or matchAll); nothing otherwise
void RH_CAT(rh_launch_match2_w, RH_W)(real_hip_ctx *ctx, const MatchArgs &a, bool all)
{
    if (a.ix.fine != 3 || !(ctx->prm.scores || all)) return;
    if (!a.ix.pbits) launch_match_wt<RH_W, 4, true>(ctx, a, all);
    else launch_match_wt<RH_W, 3, true>(ctx, a, all);
}
uint32_t RH_CAT(rh_stage_bytes_w, RH_W)(int tk) { return stg_bytes(RH_W, tk); }
