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
// Work decomposition: one lane per read, 256-thread workgroups, three waves per SIMD.
//   front  the wave copies the bases of its 64 reads (one contiguous range of the caller's array) into its
//          LDS region; every lane packs its own read into registers (32 bases per word), derives the seed
//          halves of both strands; the reverse complement is computed in registers.
//   match  per strand: the bucket-table entries of the six lists are requested together; the equal ranges
//          are enumerated in list order and the entries that survive the partner filter go to the lane's
//          queue in LDS (list-major, entry order = the reference's candidate order; a window reached
//          through consecutive lists is one entry with a list mask); the queue is drained in order: seed
//          window from the 2-bit text, popcount filters, whole-read Hamming distance.  With scores on a
//          verified window and its update() events are parked.  Three table kinds feed this stage
//          (match_lists: bucket starts; match_lists_fine: directory entries with partner digests or key
//          fingerprints) and a fourth replaces it by lookups of lane groups (match_lists_rows).
//   back   the wave copies the qualities of its reads through LDS the same way; all lanes score their
//          parked windows together and replay their events into the fold / append them for matchAll.
// The update() events of a read reach the fold in the canonical order (strand, list, position), which
// matters because the fold is order dependent when scores are on.  Every index access is a
// dependent random 8..48-byte read of a 128-byte line of an HBM-resident table: the kernel is bound by
// the HBM lines it moves (20.7 per read with digest tables, 95 % of the sustainable line rate), not by
// arithmetic (no MFMA: XOR/popcount and a short FP64 add chain).
#include "kernel_common.h"

#define EB 8  // index entries a lane requests per round trip while enumerating equal ranges
#define MQ 16 // candidate queue slots per lane (LDS); a full queue is drained and refilled
// LDS bytes of one wave: its candidate queue (MQ x 64 positions + lists, 6 x 64 cursors; the first 6.5 KiB) while
// it matches; before and after, the staging area through which it reads the bases / qualities of its reads from
// the batch.  9.5 KiB hold the 64 reads of a wave up to 151 bases each in one go; with the 8 KiB score table that
// is 46 KiB per workgroup = three workgroups per CU, which is what the registers allow anyway.
#define QUEUE_BYTES (MQ * 64u * 5u + 6u * 64u * 4u)
__host__ __device__ constexpr uint32_t stg_bytes(int W, int TK) { return TK >= 3 ? 11008u : (W <= 4 ? QUEUE_BYTES : 9728u); }
// bucket rows (table kind 3): a queue of MQR slots per lane, then the 64 rows of a list, 128 bytes each
#define MQR 8u
#define ROWBUF_OFF (MQR * 64u * 5u)
#define BKX_OFF (ROWBUF_OFF + 64u * 128u) /* 64 bucket numbers, transposed for the piece loaders */
static_assert(BKX_OFF + 256u <= stg_bytes(4, 3) && ROWBUF_OFF % 16 == 0 && QUEUE_BYTES % 16 == 0, "row staging fits the wave's LDS region");
#define STG_PAD 16u
#define NPEND 2      // verified locations a lane parks until their scores are computed (flush_pending)
#define PEND_EV 32   // update() events parked with them
#define SLOT_NONE 3u
#define PEND_OVF 0xffu // p_n of a read that needs more slots: matched again by the repeat kernel

template <int W, bool SCORES, bool ALL>
struct LaneState {
    // read
    uint64_t O[W];      // oriented read, 32 bases per word
    uint64_t shi, slo;  // seed halves (m0|m1), (m2|m3) of the oriented read
    uint32_t patl, nw, so;
    uint64_t lastmask;
    float eps;
    int inv;
    uint64_t r, o0;     // read index; offset of its bytes in the batch arrays
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
    uint32_t p_pos[NPEND], p_meta[NPEND]; // text position; k | strand << 8 | fragment << 16
    uint64_t p_tw[W];                     // aligned text words of pending location 0
    uint32_t p_n, p_nev, p_ev, cslot;     // locations, events, 1 bit per event (= location), slot of the memo
    uint32_t nhit; // matchAll: hits appended for this read
    // work counters
    unsigned cL, cP, cC, cS, cH, cV;
};


// revcomp of a read held as W words of 32 bases: out[i] = 3 - in[patl-1-i] (Pattern.hpp:105-128)
template <int W>
__device__ __forceinline__ void revcomp_words(const uint64_t *in, uint64_t *out, uint32_t patl)
{
    const uint32_t nw = (patl + 31) >> 5;
    const uint32_t pad = 64 * nw - 2 * patl; // 0..62
#pragma unroll
    for (int j = 0; j < W; ++j) {
        uint64_t x = 0, y = 0;
#pragma unroll
        for (int k = 0; k < W; ++k) { // rev2(in[nw-1-j]), rev2(in[nw-2-j]) with register-static indices
            if ((uint32_t)k + (uint32_t)j + 1 == nw) x = rev2(in[k]);
            if ((uint32_t)k + (uint32_t)j + 2 == nw) y = rev2(in[k]);
        }
        const uint64_t v = pad ? ((x << pad) | (y >> (64 - pad))) : x;
        const uint64_t valid = ((uint32_t)j + 1 < nw) ? ~0ull : ((uint32_t)j + 1 == nw ? (~0ull << pad) : 0ull);
        out[j] = ~v & valid;
    }
}

// ---- the bytes of a read (mapped symbols or qualities) -------------------------------------------------
// staged by the wave into its LDS region (row at byte offset lb; the region has STG_PAD bytes of slack in
// front and 4 behind, so a dword that straddles either end of the row is a harmless over-read) ...
struct LdsRow {
    const uint8_t *stg;
    uint32_t lb;
    __device__ __forceinline__ uint32_t dword(int byteoff, uint32_t) const
    {
        const uint32_t o = lb + (uint32_t)byteoff;
        const uint32_t *p = reinterpret_cast<const uint32_t *>(stg + (o & ~3u));
        return __builtin_amdgcn_alignbyte(p[1], p[0], o & 3u);
    }
};
// ... or read in place, byte by byte and never outside the row (repeat kernel: few reads, scattered)
struct GlobalRow {
    const uint8_t *row;
    __device__ __forceinline__ uint32_t dword(int byteoff, uint32_t patl) const
    {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = byteoff + b;
            if (i >= 0 && i < (int)patl) v |= (uint32_t)row[i] << (8 * b);
        }
        return v;
    }
};

// mapped symbols -> 32 bases per word, MSB first (what Pattern::mapped holds, Pattern.hpp:60-103); false
// if the read holds a symbol > 3 (matchUniqueImplementation.cpp:376-394)
template <int W, class Row>
__device__ __forceinline__ bool pack_read(const Row &row, uint32_t patl, uint64_t *O)
{
    bool ok = true;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        uint64_t w = 0;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const uint32_t bo = 32u * j + 4u * d;
            if (bo < patl) {
                uint32_t x = row.dword((int)bo, patl);
                const uint32_t rem = patl - bo;
                if (rem < 4) x &= (1u << (8 * rem)) - 1u;
                if (x & 0xfcfcfcfcu) ok = false;
                // bytes b0 b1 b2 b3 (2 bits each) -> b0<<6 | b1<<4 | b2<<2 | b3 in bits 24..31 of the product
                w |= (uint64_t)(((x & 0x03030303u) * 0x40100401u) >> 24) << (56 - 8 * d);
            }
        }
        O[j] = w;
    }
    return ok;
}

// seed halves (m0|m1), (m2|m3) of read[0..l) and of its reverse complement
// (SignatureConstruction.hpp:347-410)
template <int W>
__device__ __forceinline__ void seed_halves(const uint64_t *O, uint32_t l, uint64_t &shi, uint64_t &slo, uint64_t &rhi, uint64_t &rlo)
{
    const uint32_t h = l >> 1; // 2..32 bases
    const uint64_t hm = (h == 32) ? ~0ull : ((1ull << (2 * h)) - 1);
    shi = O[0] >> (64 - 2 * h);
    if (2 * h <= 32) slo = (O[0] >> (64 - 4 * h)) & hm;
    else slo = ((h == 32) ? O[W > 1 ? 1 : 0] : (((O[0] << (2 * h)) | (O[W > 1 ? 1 : 0] >> (64 - 2 * h))) >> (64 - 2 * h)));
    rhi = (rev2(slo) >> (64 - 2 * h)) ^ hm; // the revcomp of the second half comes first
    rlo = (rev2(shi) >> (64 - 2 * h)) ^ hm;
}

// ComputeScore<...,true>::computeScore, ComputeScore.hpp:50-190: sequential FP64 sum in base order
// starting at 1.0, cast to float once.  Ow = oriented read, tw = text aligned to the read, qrow = the
// read's qualities as given (oriented here: base i of the reversed read has quality[patl-1-i],
// Pattern.hpp:105-128); no qualities => 30 (Pattern.hpp:42-45).
template <int W, class Row>
__device__ __forceinline__ float score_location(const double *sLL, const uint64_t *Ow, const uint64_t *tw, uint32_t patl,
                                                const Row &qrow, bool has_q, uint32_t inv)
{
    double raw = 1.0;
    // 16 bases per step in a real (not unrolled) loop: the adds are one dependent chain, and a fully
    // unrolled body lets the scheduler hoist every table read in front of it (1 wave per SIMD).  The
    // per-step operands sit in registers and are rotated down by one slot per step, which keeps all
    // register indices static.
    constexpr int NQ = 2 * W;
    uint32_t th[NQ], oh[NQ];
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
        th[c] = (uint32_t)(tw[c >> 1] >> ((c & 1) ? 0 : 32));
        oh[c] = (uint32_t)(Ow[c >> 1] >> ((c & 1) ? 0 : 32));
    }
    const uint32_t nchunk = (patl + 15) >> 4;
#pragma unroll 1
    for (uint32_t c = 0; c < nchunk; ++c) {
        uint32_t qa[4] = {0x1e1e1e1eu, 0x1e1e1e1eu, 0x1e1e1e1eu, 0x1e1e1e1eu};
        if (has_q) {
            if (!inv) {
#pragma unroll
                for (int i = 0; i < 4; ++i) qa[i] = qrow.dword((int)(16 * c) + 4 * i, patl);
            } else { // bytes [p-15, p] with p = patl-1-16c, last one first
                const int p0 = (int)patl - 16 - (int)(16 * c);
#pragma unroll
                for (int i = 0; i < 4; ++i) qa[i] = __builtin_bswap32(qrow.dword(p0 + 4 * (3 - i), patl));
            }
        }
        const uint32_t lim = min(16u, patl - 16u * c);
        const uint32_t tr = th[0], rr = oh[0];
#pragma unroll
        for (uint32_t u = 0; u < 16; ++u) {
            if (u < lim) {
                const uint32_t ref = (tr >> (30 - 2 * u)) & 3;
                const uint32_t rb = (rr >> (30 - 2 * u)) & 3;
                const uint32_t q = (qa[u >> 2] >> (8 * (u & 3))) & 0xff;
                raw += sLL[((ref << 8) | (rb << 6) | q) & 1023];
            }
        }
#pragma unroll
        for (int i = 0; i + 1 < NQ; ++i) { th[i] = th[i + 1]; oh[i] = oh[i + 1]; }
    }
    return (float)raw;
}

// the update() call itself: the best/unique fold, or the matchAll append
template <int W, bool SCORES, bool ALL>
__device__ __forceinline__ void deliver(const MatchArgs &a, LaneState<W, SCORES, ALL> &s, uint32_t pos, uint32_t meta, float score)
{
    if (ALL) {
        unsigned long long slot = wave_append_slot(a.raw_count);
        if (slot < a.raw_cap) a.raw[slot] = make_uint4((uint32_t)s.r, pos, __float_as_uint(score), meta);
        s.nhit++;
    } else {
        fold_update<SCORES>((meta >> 8) & 1, a.t.fileid, pos, meta & 0xff, score, s.eps, meta >> 16, s.info, s.iscore);
    }
}

// Scores of the parked locations, then their update() events in the order they occurred.  The score is
// ~1000 instructions of which every lane of a wave needs one or two per read, but at different points
// of its candidate loop: computed where the hit is found, the wave would run that code once per
// distinct point (5-6 times per wave, mostly idle lanes).  Parked, the lanes score together at the end
// of the read; the fold sees the same events in the same order.  A read that needs more slots
// (repeat-rich) is handed to the repeat kernel, which scores in place.
template <int W, bool SCORES, bool ALL, class Row>
__device__ __forceinline__ void flush_pending(const MatchArgs &a, LaneState<W, SCORES, ALL> &s, const double *sLL, const Row &qrow)
{
    float sc[NPEND] = {1.0f, 1.0f};
#pragma unroll 1
    for (uint32_t j = 0; j < s.p_n; ++j) {
        const uint32_t pos = j ? s.p_pos[1] : s.p_pos[0], meta = j ? s.p_meta[1] : s.p_meta[0];
        const uint32_t inv = (meta >> 8) & 1;
        uint64_t Ow[W], tw[W];
        if ((int)inv == s.inv) {
#pragma unroll
            for (int i = 0; i < W; ++i) Ow[i] = s.O[i];
        } else {
            revcomp_words<W>(s.O, Ow, s.patl);
        }
        if (j == 0) {
#pragma unroll
            for (int i = 0; i < W; ++i) tw[i] = s.p_tw[i];
        } else { // second location of a read: the text is read again
            const uint64_t wi = pos >> 5;
            const unsigned sh = 2u * (pos & 31);
            uint64_t t[W + 2];
#pragma unroll
            for (int i = 0; i <= W; i += 2) {
                U64x2 p2 = {0ull, 0ull};
                if ((uint32_t)i <= s.nw) p2 = load2(a.t.text + wi + i);
                t[i] = p2.a; t[i + 1] = p2.b;
            }
#pragma unroll
            for (int i = 0; i < W; ++i) tw[i] = sh ? ((t[i] << sh) | (t[i + 1] >> (64 - sh))) : t[i];
        }
        const float v = score_location<W>(sLL, Ow, tw, s.patl, qrow, a.b.qual != nullptr, inv);
        if (j) sc[1] = v; else sc[0] = v;
    }
#pragma unroll 1
    for (uint32_t e = 0; e < s.p_nev; ++e) {
        const uint32_t j = (s.p_ev >> e) & 1u;
        deliver<W, SCORES, ALL>(a, s, j ? s.p_pos[1] : s.p_pos[0], j ? s.p_meta[1] : s.p_meta[0], j ? sc[1] : sc[0]);
    }
}

// one member of a bucket whose fingerprint equals the read's: the body of the candidate loop
// of ::match (match.hpp:383-413)
// A survivor of the partner filter goes to the lane's queue.  The same window reached through the next list right
// after (the true locus is found through 4.4 lists per read) only sets that list's bit in the entry: the drain then
// runs once per window instead of once per (window, list), and the order of the update() events stays the same.
__device__ __forceinline__ void queue_push(uint32_t *q_pos, uint8_t *q_la, uint32_t &qn, uint32_t pos, int la)
{
    if (qn && q_pos[(qn - 1) * 64] == pos) q_la[(qn - 1) * 64] |= (uint8_t)(1u << la);
    else { q_pos[qn * 64] = pos; q_la[qn * 64] = (uint8_t)(1u << la); qn++; }
}

// lmask = the lists (bit la) through which this window was reached one right after the other: the window is looked at
// once, its update() events are delivered list by list
template <int W, bool SCORES, bool ALL, bool DEFER>
__device__ __forceinline__ void process_candidate(const MatchArgs &a, LaneState<W, SCORES, ALL> &s, const double *sLL,
                                                  uint32_t rpos, uint32_t lmask)
{
    if (SCORES && DEFER && s.p_n == PEND_OVF) return; // handed to the repeat kernel
    const uint64_t *__restrict__ T = a.t.text;
    const uint32_t bb = a.b_bits;
    const uint64_t mb = (bb >= 64) ? ~0ull : ((1ull << bb) - 1);
    // seed window of the genome at rpos, as the two halves (m0|m1), (m2|m3); per-segment mismatch
    // counts are a function of (strand, rpos) only and are memoised: the true locus is reached
    // through up to six lists in a row
    if (rpos != s.crpos) {
        // the seed window (2*l bits at bit offset 2*rpos) spans two words, three when l > 32
        const uint64_t wi0 = rpos >> 5;
        const U64x2 tt = load2(T + wi0);
        const uint64_t t2 = (a.l > 32) ? T[wi0 + 2] : 0ull;
        const unsigned sh0 = 2u * (rpos & 31);
        const uint64_t xhi = extract_bits(tt.a, tt.b, t2, sh0, a.l) ^ s.shi;
        const uint64_t xlo = extract_bits(tt.a, tt.b, t2, sh0 + a.l, a.l) ^ s.slo;
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
    if (!a.ix.pbits) s.cC += nm; // (with partner bits the entry holds the whole signature: counted at the scan)
    const unsigned seedk = k0 + k1 + k2 + k3; // = diffcountpair(s_b, list_b[p->ptr].sign), match.hpp:386
    if (seedk > a.seedkmax) return;
    s.cS += nm;
    if (rpos < s.so) return; // match.hpp:393
    const uint32_t pos = rpos - s.so;
    bool reg = false; // a location verified here for the first time
    if (pos != s.cpos) {
        s.cpos = pos;
        s.cok = false;
        s.cV++;
        uint32_t frag;
        if (!frag_valid(a.t, pos, s.patl, frag)) return;
        if (a.t.has_wild && !wild_free(a.t.wild, pos, s.patl)) return;
        // Hamming distance of the whole oriented read against text[pos, pos+patl)
        // = seedk + RestMatch::computeDistance (RestMatch.hpp:39-81)
        const uint64_t wi = pos >> 5;
        const unsigned sh = 2u * (pos & 31);
        uint64_t tw[W];
        unsigned total = 0;
        {
            uint64_t t[W + 2];
#pragma unroll
            for (int j = 0; j <= W; j += 2) { // 16-byte requests, all in flight together
                U64x2 p2 = {0ull, 0ull};
                if ((uint32_t)j <= s.nw) p2 = load2(T + wi + j);
                t[j] = p2.a; t[j + 1] = p2.b;
            }
#pragma unroll
            for (int j = 0; j < W; ++j) {
                uint64_t al = sh ? ((t[j] << sh) | (t[j + 1] >> (64 - sh))) : t[j];
                tw[j] = al;
                uint64_t x = al ^ s.O[j];
                uint64_t d = ((x >> 1) | x) & M55;
                if ((uint32_t)j + 1 == s.nw) d &= s.lastmask;
                if ((uint32_t)j < s.nw) total += __popcll(d);
            }
        }
        if (total > a.totalkmax) return;
        float sc = 1.0f; // ComputeScore<...,false>, ComputeScore.hpp:31-45
        if (SCORES && !DEFER)
            sc = score_location<W>(sLL, s.O, tw, s.patl, GlobalRow{a.b.qual + s.o0}, a.b.qual != nullptr, (uint32_t)s.inv);
        s.cok = true; s.ck = total; s.cscore = sc; s.cfrag = frag; s.cslot = SLOT_NONE;
        if (SCORES && DEFER) { // the score is computed later (flush_pending): park the location
            if (s.p_n == NPEND) { s.p_n = PEND_OVF; return; } // out of room => the read goes to the repeat kernel
            const uint32_t meta0 = total | ((uint32_t)s.inv << 8) | (frag << 16);
            if (s.p_n) { s.p_pos[1] = pos; s.p_meta[1] = meta0; }
            else {
                s.p_pos[0] = pos; s.p_meta[0] = meta0;
#pragma unroll
                for (int j = 0; j < W; ++j) s.p_tw[j] = tw[j];
            }
            s.cslot = s.p_n++;
            reg = true;
        }
    }
    (void)reg;
    if (!s.cok) return;
    s.cH += nm; // one updater::update call per list, match.hpp:411
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
    if (!(SCORES && DEFER)) {
        for (uint32_t e = 0; e < ne; ++e) deliver<W, SCORES, ALL>(a, s, s.cpos, meta, s.cscore);
        return;
    }
    // park the events (they all refer to the memo's slot); out of room => the read goes to the repeat kernel
    if (s.p_nev + ne > PEND_EV) { s.p_n = PEND_OVF; return; }
    if (s.cslot) s.p_ev |= ((ne >= 32 ? 0xffffffffu : ((1u << ne) - 1u)) << s.p_nev);
    s.p_nev += ne;
}

// Scan of the buckets of lists [LA0, LA1) of one strand; pushes the entries that survive the key
// (and partner) comparison into the lane's LDS queue, list-major and in entry order = the reference's
// candidate order.  FIRST = the normal, only pass: all bucket-table loads are issued together, then
// the first two entries of every bucket together.  !FIRST = continuation after a full queue was
// drained (repeat-rich loci only): everything is recomputed from the seed halves and the cursors
// parked in LDS, so that no scan state has to stay in registers across the drain (occupancy).
// Returns true if the queue filled up (again).
template <int W, bool SCORES, bool ALL, bool FINE, int LA0, int LA1, bool FIRST>
__device__ __forceinline__ bool scan_lists(const MatchArgs &a, LaneState<W, SCORES, ALL> &s, uint32_t *q_pos, uint8_t *q_la,
                                           uint32_t *q_cur, uint32_t &donemask, uint32_t &qn)
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
        if (FIRST) {
            e0[i] = (lo[i] < hi[i]) ? E[lo[i]] : make_uint2(0xffffffffu, 0);
            e1[i] = (lo[i] + 1 < hi[i]) ? E[lo[i] + 1] : make_uint2(0xffffffffu, 0);
        } else {
            e0[i] = e1[i] = make_uint2(0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        if (FIRST) {
            s.cL++;
            cur[i] = lo[i];
            if (hi[i] - lo[i] > 16) { // large bucket: lower_bound on the key first
                const uint2 *__restrict__ E = a.ix.ent[LA0 + i];
                uint32_t x = lo[i], y = hi[i];
                while (x < y) {
                    uint32_t mid = x + ((y - x) >> 1);
                    s.cP++;
                    if ((E[mid].x >> pbits) < fp[i]) x = mid + 1; else y = mid;
                }
                cur[i] = x;
            }
        } else {
            cur[i] = q_cur[i * 64];
        }
    }
    // 3. scan in list order, queue the survivors
    bool again = false;
    qn = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        if (!again && !(donemask & (1u << i))) {
            const uint2 *__restrict__ E = a.ix.ent[LA0 + i];
            const uint32_t f = fp[i], h = hi[i];
            // top pbits of the read's partner signature s_b (the signature of list 5-la)
            const int lb = 5 - (LA0 + i);
            const int xb = (lb < 3) ? 0 : (lb < 5) ? 1 : 2, xd = (lb == 0) ? 1 : (lb == 1 || lb == 3) ? 2 : 3;
            const uint32_t rp = pbits ? (uint32_t)(((m[xb] << bb) | m[xd]) >> (a.l - pbits)) : 0u;
            uint32_t j = cur[i];
            while (j < h) {
                if (qn == MQ) { again = true; break; }
                uint2 e;
                if (FIRST && j == lo[i]) e = e0[i];
                else if (FIRST && j == lo[i] + 1) e = e1[i];
                else e = E[j];
                s.cP++;
                const uint32_t ek = e.x >> pbits;
                if (ek > f) { j = h; break; }
                if (ek == f) {
                    bool keep = true;
                    if (pbits) {
                        // member of the reference's equal range; seed popcount filter (match.hpp:386) on the
                        // partner symbols the entry carries: more than seedkmax known mismatches => rejected
                        // without touching the text (exact: the full count can only be larger)
                        s.cC++;
                        const uint32_t x = (e.x & pmask) ^ rp;
                        keep = __popc(((x >> 1) | x) & 0x55555555u) <= a.seedkmax;
                    }
                    if (keep) queue_push(q_pos, q_la, qn, e.y, LA0 + i);
                }
                j++;
            }
            cur[i] = j;
            if (!again) donemask |= 1u << i;
        }
    }
    if (again) { // park the cursors in LDS; the continuation pass reloads them
#pragma unroll
        for (int i = 0; i < NL; ++i) q_cur[i * 64] = cur[i];
    }
    return again;
}

// Fine tables (32-bit signatures, large index): the 16-byte bucket table entry {start, size and partner digest
// of each of the (at most eight) key groups} of a prefix gives the reference's equal range of every list
// directly, and for a range of one entry enough of its partner signature to reject most chance candidates
// without reading them.  The lane enumerates the equal ranges of all lists in list order -- the canonical
// candidate order -- eight entries per round trip, applies the partner filter and queues the survivors; a
// full queue is drained and refilled (the only state across a drain is the enumeration offset).
template <int W, bool SCORES, bool ALL, bool DEFER, int LA0, int LA1>
__device__ __forceinline__ void match_lists_fine(const MatchArgs &a, LaneState<W, SCORES, ALL> &s, const double *sLL,
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
                s.cP += sz;
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
            lo[i] = start; cum[i] = total; total += sz;
            s.cL++;
        }
    }
    s.cC += counted; s.cP += counted;
    // 2. enumerate, filter, queue, drain
    for (uint32_t kb = 0; kb < total && !(DEFER && s.p_n == PEND_OVF); kb += MQ) {
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
                        queue_push(q_pos, q_la, qn, e[u].y, LA0 + (int)li[u]);
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
template <int W, bool SCORES, bool ALL, bool FINE, bool DEFER, int LA0, int LA1>
__device__ __forceinline__ void match_lists(const MatchArgs &a, LaneState<W, SCORES, ALL> &s, const double *sLL,
                                            uint32_t *q_pos, uint8_t *q_la, uint32_t *q_cur)
{
    if (FINE) { match_lists_fine<W, SCORES, ALL, DEFER, LA0, LA1>(a, s, sLL, q_pos, q_la); return; }
    uint32_t donemask = 0, qn = 0;
    bool again = scan_lists<W, SCORES, ALL, FINE, LA0, LA1, true>(a, s, q_pos, q_la, q_cur, donemask, qn);
    // 4. verify / score / fold in candidate order
    for (uint32_t k = 0; k < qn; ++k)
        process_candidate<W, SCORES, ALL, DEFER>(a, s, sLL, q_pos[k * 64], (uint32_t)q_la[k * 64]);
    while (again) {
        again = scan_lists<W, SCORES, ALL, FINE, LA0, LA1, false>(a, s, q_pos, q_la, q_cur, donemask, qn);
        for (uint32_t k = 0; k < qn; ++k)
            process_candidate<W, SCORES, ALL, DEFER>(a, s, sLL, q_pos[k * 64], (uint32_t)q_la[k * 64]);
    }
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
// drained by every lane for itself, in list order, when one is full and at the end.
template <int W, bool SCORES, bool ALL, bool DEFER, bool WIDE, int LA0, int LA1>
__device__ __forceinline__ void match_lists_rows(const MatchArgs &a, LaneState<W, SCORES, ALL> &s, const double *sLL, uint8_t *stg, bool act)
{
    constexpr int NL = LA1 - LA0;
    const uint32_t lane = threadIdx.x & 63;
    // the wave's LDS region in this mode: MQR x 64 queued positions, MQR x 64 lists, 64 rows of 128 bytes
    uint32_t *q_pos = reinterpret_cast<uint32_t *>(stg) + lane;
    uint8_t *q_la = stg + MQR * 64 * 4 + lane;
    uint8_t *rowbuf = stg + ROWBUF_OFF;
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
    auto bucket_of = [&](int la) { return wide ? (uint32_t)(sig_wide(la) >> a.ix.pshift) : (sig_of(la) >> gbits); };
    uint32_t qn = 0;
    uint4 va0, va1, va2, va3, va4, va5, va6, va7; // (eight scalars, not an array: the array went through scratch memory)
    uint32_t *bkx = reinterpret_cast<uint32_t *>(stg + BKX_OFF);
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
    ISSUE_ROWS(LA0);
#pragma unroll 1
    for (int li = 0; li < NL; ++li) { // (a real loop: the drain below must exist once, not NL times)
        const int la = LA0 + li;
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
        if (li + 1 < NL) ISSUE_ROWS(la + 1); // the next list's rows are in flight while this one is decoded and drained
        // owners: directory of the row, then the entries of their key group
        const bool mine = act && !(DEFER && s.p_n == PEND_OVF);
        const uint8_t *rowb = rowbuf + lane * 128;
        const uint32_t sw = lane & 7;
        auto row = [&](uint32_t d) { return *reinterpret_cast<const uint32_t *>(rowb + ((((d >> 2) ^ sw) << 4) | ((d & 3) << 2))); };
        uint32_t e_cnt = 0, e_j = 0, e_base = 0;
        bool e_ovf = false;
        uint32_t g, r; // key group; what an entry's key is compared with (narrow: leading pbits of s_b; wide: the 32-bit key)
        if (!wide) {
            g = sig_of(la) & ((1u << gbits) - 1);
            r = sig_of(5 - la) >> (a.l - pbits);
        } else {
            r = (uint32_t)(sig_wide(la) >> a.ix.fshift);
            g = r >> 28;
        }
        if (mine) {
            s.cL++;
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
            if (!wide) s.cC += e_cnt; // (wide: counted when the text confirms the membership)
            s.cP += e_cnt;
        }
        while (true) {
            while (e_j < e_cnt && qn < MQR) {
                uint32_t pos;
                bool pass;
                if (!e_ovf) { // 6 bytes at halfword 4 + 3 * (e_base + e_j) of the row
                    const uint32_t h = 4 + 3 * (e_base + e_j);
                    const uint32_t d0 = row(h >> 1), d1 = row((h >> 1) + 1);
                    const uint32_t key = (h & 1) ? (d0 >> 16) : (d0 & 0xffffu);
                    pos = (h & 1) ? d1 : ((d0 >> 16) | (d1 << 16));
                    const uint32_t x = key ^ (r >> (pbits - p16));
                    pass = wide ? (key == rh_fp16(r)) : (__popc(((x >> 1) | x) & 0x5555u) <= a.seedkmax);
                } else {
                    const uint2 e = a.ix.ent[la][e_base + e_j];
                    pos = e.y;
                    const uint32_t x = (e.x & pmask) ^ r;
                    pass = wide ? (e.x == r) : (__popc(((x >> 1) | x) & 0x55555555u) <= a.seedkmax);
                }
                // seed popcount filter (match.hpp:386) on the partner symbols the entry carries: more than seedkmax
                // known mismatches => rejected without touching the text (exact: the full count can only be larger)
                if (pass) queue_push(q_pos, q_la, qn, pos, la);
                e_j++;
            }
            if (!__any(e_j < e_cnt) && li + 1 < NL) break;
            // a full queue somewhere, or the end of the lists: verify / score / fold in candidate order
            for (uint32_t k = 0; k < qn; ++k)
                process_candidate<W, SCORES, ALL, DEFER>(a, s, sLL, q_pos[k * 64], (uint32_t)q_la[k * 64]);
            qn = 0;
            if (DEFER && s.p_n == PEND_OVF) e_j = e_cnt;
            if (!__any(e_j < e_cnt)) break;
        }
    }
}

// both strands of one read with bucket rows: every lane of the wave comes along, `act` tells which ones have a read
template <int W, bool SCORES, bool ALL, bool DEFER, bool WIDE>
__device__ __forceinline__ void match_read_rows(const MatchArgs &a, LaneState<W, SCORES, ALL> &s, const double *sLL, uint8_t *stg, bool act)
{
    const uint32_t patl = act ? s.patl : 32u * W;
    s.nw = (patl + 31) >> 5;
    s.lastmask = ~0ull << (64 - 2 * (patl - 32 * (s.nw - 1)));
    s.eps = (float)(a.filter_mult * (double)patl); // RealOptions.hpp:74-77
    s.p_n = s.p_nev = s.p_ev = 0; s.cslot = SLOT_NONE;
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
        s.so = inv ? (patl - a.l) : 0u; // RestMatch::getMatchOffset, RestMatch.hpp:84-89
        s.cpos = 0xffffffffu; s.ck = 0; s.cfrag = 0; s.cscore = 1.0f; s.cok = false;
        s.crpos = 0xffffffffu; s.ckk = 0;
        const bool go = act && !(DEFER && s.p_n == PEND_OVF);
        if (!ALL && !SCORES) {
            // uni0s / uni0r early-out (matchUniqueImplementation.cpp:434-436, 470-472): lists 1..5 of a
            // strand are skipped when list 0 left the record in this strand's state with 0 errors
            match_lists_rows<W, SCORES, ALL, DEFER, WIDE, 0, 1>(a, s, sLL, stg, go);
            const unsigned st = (unsigned)(s.info >> ST_SHIFT), er = (unsigned)(s.info >> ER_SHIFT) & 15;
            match_lists_rows<W, SCORES, ALL, DEFER, WIDE, 1, 6>(a, s, sLL, stg, go && !(st == (unsigned)(inv ? ST_REVERSE : ST_STRAIGHT) && er == 0));
        } else {
            match_lists_rows<W, SCORES, ALL, DEFER, WIDE, 0, 6>(a, s, sLL, stg, go);
        }
    }
}

// both strands of one read (UniqueMatcher::match / AllMatcher::match, matchUniqueImplementation.cpp:396-500,
// matchAllImplementation.cpp:261-355): s.O holds the read as given on entry, its reverse complement on exit
template <int W, bool SCORES, bool ALL, bool FINE, bool DEFER>
__device__ __forceinline__ void match_read(const MatchArgs &a, LaneState<W, SCORES, ALL> &s, const double *sLL, uint32_t *q_pos,
                                           uint8_t *q_la, uint32_t *q_cur)
{
    const uint32_t patl = s.patl;
    s.nw = (patl + 31) >> 5;
    s.lastmask = ~0ull << (64 - 2 * (patl - 32 * (s.nw - 1)));
    s.eps = (float)(a.filter_mult * (double)patl); // RealOptions.hpp:74-77
    s.p_n = s.p_nev = s.p_ev = 0; s.cslot = SLOT_NONE;
    uint64_t rhi, rlo;
    seed_halves<W>(s.O, a.l, s.shi, s.slo, rhi, rlo);
    for (int inv = 0; inv < 2; ++inv) {
        if (DEFER && s.p_n == PEND_OVF) break;
        if (inv) { // transposed pattern, Pattern.hpp:105-128
            uint64_t R[W];
            revcomp_words<W>(s.O, R, patl);
#pragma unroll
            for (int j = 0; j < W; ++j) s.O[j] = R[j];
            s.shi = rhi; s.slo = rlo;
        }
        s.inv = inv;
        s.so = inv ? (patl - a.l) : 0u; // RestMatch::getMatchOffset, RestMatch.hpp:84-89
        s.cpos = 0xffffffffu; s.ck = 0; s.cfrag = 0; s.cscore = 1.0f; s.cok = false;
        s.crpos = 0xffffffffu; s.ckk = 0;
        if (!ALL && !SCORES) {
            // uni0s / uni0r early-out (matchUniqueImplementation.cpp:434-436, 470-472): lists 1..5 of a
            // strand are skipped when list 0 left the record in this strand's state with 0 errors
            match_lists<W, SCORES, ALL, FINE, DEFER, 0, 1>(a, s, sLL, q_pos, q_la, q_cur);
            const unsigned st = (unsigned)(s.info >> ST_SHIFT), er = (unsigned)(s.info >> ER_SHIFT) & 15;
            if (!(st == (unsigned)(inv ? ST_REVERSE : ST_STRAIGHT) && er == 0))
                match_lists<W, SCORES, ALL, FINE, DEFER, 1, 6>(a, s, sLL, q_pos, q_la, q_cur);
        } else {
            match_lists<W, SCORES, ALL, FINE, DEFER, 0, 6>(a, s, sLL, q_pos, q_la, q_cur);
        }
    }
}

// REPEAT = false: the matcher proper, lane i of the grid takes read i of the batch.  A wave reads the bases
// of its 64 reads -- one contiguous byte range of the caller's array -- through LDS, every lane packs its
// own read into registers, matches both strands, and parks its hits (DEFER, scores on); then the wave reads
// the qualities of its reads the same way and the lanes score and deliver together.  A read that needs more
// than NPEND locations / PEND_EV events is left untouched and its index appended to a.ovf_list.
// REPEAT = true: the same matcher with in-place scoring over the reads of a.ovf_list (grid-stride; the list
// length is read from device memory, no host round trip); a lane fetches the bytes of its read itself.
// TK = kind of the bucket tables: 0 bucket starts, 1 directory entries (digests / fingerprints), 3 bucket rows,
// 4 bucket rows of signatures wider than 32 bits
template <int W, bool SCORES, bool ALL, int TK, bool REPEAT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W <= 5 ? 3 : 2))) void match_kernel(MatchArgs a)
{
    constexpr bool FINE = TK != 0;
    constexpr bool DEFER = SCORES && !REPEAT;
    __shared__ double sLL[SCORES ? 1024 : 1];
    __shared__ __attribute__((aligned(16))) uint8_t smem[4 * stg_bytes(W, TK)];
    if (SCORES) {
        for (int i = threadIdx.x; i < 1024; i += 256) sLL[i] = a.LL[i];
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63;
    uint8_t *stg = smem + (threadIdx.x >> 6) * stg_bytes(W, TK);
    uint32_t *q_pos = reinterpret_cast<uint32_t *>(stg) + lane;
    uint8_t *q_la = stg + MQ * 64 * 4 + lane;
    uint32_t *q_cur = reinterpret_cast<uint32_t *>(stg + MQ * 64 * 5) + lane;
    LaneState<W, SCORES, ALL> s;
    s.cL = s.cP = s.cC = s.cS = s.cH = s.cV = 0;
    s.info = 0; s.iscore = 0.f; s.o0 = 0;
    unsigned cR = 0;
    const uint64_t n = a.b.n_reads;

    if (!REPEAT) {
        const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
        const uint64_t rc = r < n ? r : n; // lanes behind the batch: an empty range at its end
        const uint64_t o0 = a.b.off ? a.b.off[rc] : rc * (uint64_t)a.b.upatl;
        const uint64_t o1 = r < n ? (a.b.off ? a.b.off[r + 1] : o0 + a.b.upatl) : o0;
        const uint32_t patl = (uint32_t)(o1 - o0);
        const uint32_t GL = a.b.gl; // reads the wave stages at a time: GL * max_patl fits its LDS region
        // ---- bases: global -> LDS -> registers
        bool elig = false;
        for (uint32_t g = 0; g < 64; g += GL) {
            const uint64_t gb = __shfl(o0, (int)g), ge = __shfl(o1, (int)(g + GL - 1));
            wave_lds_sync();
            const uint32_t l0 = stage_wave(stg, a.b.bases + gb, ge - gb, lane);
            wave_lds_sync();
            if (lane >= g && lane < g + GL && patl >= a.l && patl <= 32u * W) // matchUniqueImplementation.cpp:376-394
                elig = pack_read<W>(LdsRow{stg, l0 + (uint32_t)(o0 - gb)}, patl, s.O);
        }
        wave_lds_sync();
        // ---- match
        s.r = r; s.patl = patl; s.p_n = 0; s.nhit = 0;
        if (elig && !ALL) {
            s.info = a.info[r];
            if (SCORES) s.iscore = a.score[r];
        }
        if (TK >= 3) // bucket rows: lookups by lane groups, the whole wave comes along
            match_read_rows<W, SCORES, ALL, DEFER, TK == 4>(a, s, sLL, stg, elig);
        else if (elig)
            match_read<W, SCORES, ALL, FINE, DEFER>(a, s, sLL, q_pos, q_la, q_cur);
        const bool ovf = DEFER && elig && s.p_n == PEND_OVF;
        if (ovf) {
            // nothing of this read has been delivered: the repeat kernel does it all and counts it
            const unsigned long long slot = wave_append_slot(a.ovf_count);
            a.ovf_list[slot] = (uint32_t)r;
            s.cL = s.cP = s.cC = s.cS = s.cH = s.cV = 0;
        }
        // ---- qualities: global -> LDS; score the parked hits and deliver them
        if (DEFER) {
            for (uint32_t g = 0; g < 64; g += GL) {
                const uint64_t gb = __shfl(o0, (int)g), ge = __shfl(o1, (int)(g + GL - 1));
                const bool mine = lane >= g && lane < g + GL && elig && !ovf && s.p_n;
                if (!__any(mine)) continue;
                uint32_t l0 = 0;
                wave_lds_sync();
                if (a.b.qual) l0 = stage_wave(stg, a.b.qual + gb, ge - gb, lane);
                wave_lds_sync();
                if (mine) flush_pending<W, SCORES, ALL>(a, s, sLL, LdsRow{stg, l0 + (uint32_t)(o0 - gb)});
            }
        }
        if (elig && !ovf) {
            cR = 1;
            if (!ALL) {
                a.info[r] = s.info;
                if (SCORES) a.score[r] = s.iscore;
            }
        }
        if (ALL && r < n) a.hit_cnt[r] = (elig && !ovf) ? s.nhit : 0u; // (a handed-over read: the repeat kernel writes it)
    } else {
        const uint64_t n_items = (uint64_t)*a.ovf_count;
        // (every lane of a wave makes the same number of trips: the row lookups are wave-wide)
        for (uint64_t it0 = (uint64_t)blockIdx.x * 256; it0 < n_items; it0 += (uint64_t)gridDim.x * 256) {
            const uint64_t it = it0 + threadIdx.x;
            const bool have = it < n_items;
            const uint64_t r = have ? a.ovf_list[it] : 0;
            const uint64_t o0 = a.b.off ? a.b.off[r] : r * (uint64_t)a.b.upatl;
            const uint32_t patl = a.b.off ? (uint32_t)(a.b.off[r + 1] - o0) : a.b.upatl;
            s.r = r; s.o0 = o0; s.patl = patl; s.nhit = 0;
            if (have) {
                pack_read<W>(GlobalRow{a.b.bases + o0}, patl, s.O); // (eligible: the matcher handed it over)
                if (!ALL) {
                    s.info = a.info[r];
                    if (SCORES) s.iscore = a.score[r];
                }
            }
            if (TK >= 3)
                match_read_rows<W, SCORES, ALL, DEFER, TK == 4>(a, s, sLL, stg, have);
            else if (have)
                match_read<W, SCORES, ALL, FINE, DEFER>(a, s, sLL, q_pos, q_la, q_cur);
            if (!have) continue;
            cR++;
            if (!ALL) {
                a.info[r] = s.info;
                if (SCORES) a.score[r] = s.iscore;
            } else {
                a.hit_cnt[r] = s.nhit;
            }
        }
    }

    // work counters: wave reduction, one atomic per wave and counter
    unsigned c[7] = {cR, s.cL, s.cP, s.cC, s.cS, s.cH, s.cV};
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        unsigned v = c[k];
        for (int d = 32; d; d >>= 1) v += __shfl_xor((int)v, d);
        if ((threadIdx.x & 63) == 0 && v)
            atomicAdd(a.counters + (size_t)((blockIdx.x * 4u + (threadIdx.x >> 6)) & (RH_CSTRIPES - 1)) * 16 + k, (unsigned long long)v);
    }
}

// ---------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------
template <int W, int FINE>
static void launch_match_wf(real_hip_ctx *ctx, const MatchArgs &a, bool all)
{
    dim3 grid((unsigned)((a.b.n_reads + 255) / 256)), block(256);
    const bool sc = ctx->prm.scores != 0;
    if (all) {
        if (sc) hipLaunchKernelGGL((match_kernel<W, true, true, FINE, false>), grid, block, 0, ctx->stream, a);
        else    hipLaunchKernelGGL((match_kernel<W, false, true, FINE, false>), grid, block, 0, ctx->stream, a);
    } else {
        if (sc) hipLaunchKernelGGL((match_kernel<W, true, false, FINE, false>), grid, block, 0, ctx->stream, a);
        else    hipLaunchKernelGGL((match_kernel<W, false, false, FINE, false>), grid, block, 0, ctx->stream, a);
    }
}
// reads the matcher handed over (scores on only): far fewer than the batch, so a fixed grid strides over them
template <int W, int FINE>
static void launch_repeat_wf(real_hip_ctx *ctx, const MatchArgs &a, bool all)
{
    const uint64_t blocks = (a.b.n_reads + 255) / 256;
    dim3 grid((unsigned)(blocks < 512 ? blocks : 512)), block(256); // (two workgroups per CU: the hand-over list is short)
    if (all) hipLaunchKernelGGL((match_kernel<W, true, true, FINE, true>), grid, block, 0, ctx->stream, a);
    else     hipLaunchKernelGGL((match_kernel<W, true, false, FINE, true>), grid, block, 0, ctx->stream, a);
}
template <int W>
static void launch_match_w(real_hip_ctx *ctx, const MatchArgs &a, bool all, bool repeat)
{
    if (repeat) {
        if (a.ix.fine == 3 && !a.ix.pbits) launch_repeat_wf<W, 4>(ctx, a, all);
        else if (a.ix.fine == 3) launch_repeat_wf<W, 3>(ctx, a, all);
        else if (a.ix.fine) launch_repeat_wf<W, 1>(ctx, a, all);
        else launch_repeat_wf<W, 0>(ctx, a, all);
        return;
    }
    if (a.ix.fine == 3 && !a.ix.pbits) launch_match_wf<W, 4>(ctx, a, all);
    else if (a.ix.fine == 3) launch_match_wf<W, 3>(ctx, a, all);
    else if (a.ix.fine) launch_match_wf<W, 1>(ctx, a, all);
    else launch_match_wf<W, 0>(ctx, a, all);
}

static int launch_match_any(real_hip_ctx *ctx, const MatchArgs &a, bool all, bool repeat)
{
    switch (a.b.W) {
    case 1: launch_match_w<1>(ctx, a, all, repeat); break;
    case 2: launch_match_w<2>(ctx, a, all, repeat); break;
    case 3: launch_match_w<3>(ctx, a, all, repeat); break;
    case 4: launch_match_w<4>(ctx, a, all, repeat); break;
    case 5: launch_match_w<5>(ctx, a, all, repeat); break;
    case 6: launch_match_w<6>(ctx, a, all, repeat); break;
    case 7: launch_match_w<7>(ctx, a, all, repeat); break;
    case 8: launch_match_w<8>(ctx, a, all, repeat); break;
    default: return rh_fail(ctx, REAL_HIP_E_UNSUPPORTED, "read longer than REAL_HIP_MAX_PATL", hipSuccess);
    }
    return REAL_HIP_OK;
}

int rh_launch_match(real_hip_ctx *ctx, const MatchArgs &args, bool all)
{
    if (!args.b.n_reads) return REAL_HIP_OK;
    MatchArgs a = args;
    const bool sc = ctx->prm.scores != 0;
    int rc;
    { // reads a wave stages at a time: their bytes (+ alignment skew, pad, one dword of over-read) fit its LDS region
        const uint32_t maxlen = a.b.off ? 32u * a.b.W : a.b.upatl;
        uint32_t gl = 64;
        while (gl > 1 && (uint64_t)gl * maxlen + STG_PAD + 16 + 16 > stg_bytes((int)a.b.W, (int)a.ix.fine)) gl >>= 1;
        a.b.gl = gl;
    }
    if (sc) { // hand-over list of the reads the matcher leaves to the repeat kernel
        if ((rc = rh_reserve(ctx, ctx->ovf_list, a.b.n_reads * 4))) return rc;
        if ((rc = rh_reserve(ctx, ctx->ovf_count, 8))) return rc;
        a.ovf_list = (uint32_t *)ctx->ovf_list.p;
        a.ovf_count = (unsigned long long *)ctx->ovf_count.p;
        RH_HIP(ctx, hipMemsetAsync(ctx->ovf_count.p, 0, 8, ctx->stream));
    }
    rh_time_begin(ctx, ctx->stream, all ? REAL_HIP_K_MATCH_ALL : REAL_HIP_K_MATCH_UNIQUE);
    if ((rc = launch_match_any(ctx, a, all, false))) return rc;
    rh_time_end(ctx, ctx->stream);
    RH_HIP(ctx, hipGetLastError());
    if (sc) {
        rh_time_begin(ctx, ctx->stream, REAL_HIP_K_MATCH_REPEAT);
        if ((rc = launch_match_any(ctx, a, all, true))) return rc;
        rh_time_end(ctx, ctx->stream);
        RH_HIP(ctx, hipGetLastError());
    }
    return REAL_HIP_OK;
}
