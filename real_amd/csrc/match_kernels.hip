// match_kernels.hip -- the per-read hot path of REAL as hand-written gfx950 kernels.
//
//   pack_kernel   a1/a3/a4/a5 of SURVEY 8(a): 2-bit packing of both orientations
//                 of a read (Pattern::computeMapped, Pattern.hpp:105-128;
//                 RestMatch::fillRestWordArray{,Reverse}Mapped, RestMatch.hpp:214-318)
//                 and the seed halves s0/s5 of SignatureConstruction::signatureMapped /
//                 reverseMappedSignature (SignatureConstruction.hpp:218-280,347-410).
//   match_kernel  a6..a12: bucket lookup + in-bucket search (match.hpp:376-381),
//                 seed popcount filter (:383-388), position / N checks (:390-398),
//                 Hamming verify (RestMatch.hpp:39-81), ComputeScore (ComputeScore.hpp:50-190),
//                 and either the best/unique fold (matchUniqueImplementation.cpp:97-160,
//                 179-248) or the hit append of matchAll (matchAllImplementation.cpp:172-184).
//
// Work decomposition: one lane per read, 256-thread workgroups.  The six lookups
// of each strand run in the reference's order, so the update() events of a read
// reach the fold in the canonical order (strand, list, position); the fold is
// order dependent when scores are on.  All memory traffic of the
// kernel is dependent random 8..40-byte reads into HBM-resident tables: this is a
// latency/sector-throughput-bound integer kernel, no MFMA.
#include "real_hip_internal.h"

#define M55 0x5555555555555555ull

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------
// number of differing 2-bit symbols (PopCountTable.hpp:103-131 diffcountpair)
__device__ __forceinline__ unsigned pairdiff(uint64_t x) { return __popcll(((x >> 1) | x) & M55); }

// nb (1..32) bases starting at base i, right aligned (AutoTextArray::getTextWord(i,l),
// AutoTextArray.hpp:122-125 -> Rank::getBits64, ERank222B.hpp:55-85)
__device__ __forceinline__ uint64_t text_bits(const uint64_t *__restrict__ T, uint64_t i, unsigned nb)
{
    uint64_t w = i >> 5;
    unsigned sh = 2u * (unsigned)(i & 31);
    uint64_t v = T[w] << sh;
    if (sh + 2 * nb > 64) v |= T[w + 1] >> (64 - sh);
    return v >> (64 - 2 * nb);
}

// RangeVector::isPositionValid / positionToRange (RangeVector.hpp:59-80): fragment of
// pos = (number of fragment starts <= pos) - 1; valid iff the read ends inside it.
__device__ __forceinline__ bool frag_valid(const DevText &t, uint32_t pos, uint32_t patl, uint32_t &frag)
{
    if (t.n_frag == 1) {
        frag = 0;
        return (uint64_t)pos + patl <= t.n;
    }
    uint32_t lo = 0, hi = t.n_frag + 1;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (t.frag_start[mid] <= (uint64_t)pos) lo = mid + 1; else hi = mid;
    }
    frag = lo - 1;
    return (uint64_t)pos + patl <= t.frag_start[lo];
}

// AutoTextArray::isDontCareFree (AutoTextArray.hpp:167-172): no N in [pos,pos+patl).
// The reference takes a rank difference; reading the bits themselves is the same predicate.
__device__ __forceinline__ bool wild_free(const uint64_t *__restrict__ Wd, uint32_t pos, uint32_t patl)
{
    uint64_t a = pos, e = (uint64_t)pos + patl - 1;
    uint64_t wa = a >> 6, we = e >> 6;
    for (uint64_t w = wa; w <= we; ++w) {
        uint64_t m = ~0ull;
        if (w == wa) m &= ~0ull >> (a & 63);
        if (w == we) m &= ~0ull << (63 - (e & 63));
        if (Wd[w] & m) return false;
    }
    return true;
}

// UniqueMatchInfo bit layout (UniqueMatchInfo.hpp:29-39)
#define ST_SHIFT 61
#define FR_SHIFT 45
#define ER_SHIFT 41
#define FI_SHIFT 35
#define POS_MASK ((1ull << 35) - 1)
enum { ST_NOMATCH = 0, ST_STRAIGHT = 1, ST_REVERSE = 2, ST_GAPPED = 3, ST_NONUNIQUE = 4 };

__device__ __forceinline__ uint64_t pack_record(unsigned st, unsigned frag, unsigned err, unsigned file, uint32_t pos)
{
    return ((uint64_t)st << ST_SHIFT) | ((uint64_t)(frag & 0xffff) << FR_SHIFT) | ((uint64_t)(err & 15) << ER_SHIFT) |
           ((uint64_t)(file & 63) << FI_SHIFT) | (uint64_t)pos;
}

// UpdateUniqueInfo<false>::update (matchUniqueImplementation.cpp:97-160) and
// UpdateUniqueInfo<true>::update (:179-248)
template <bool SCORES>
__device__ __forceinline__ void fold_update(bool inv, unsigned fileid, uint32_t pos, unsigned totalk, float score,
                                            float eps, unsigned frag, uint64_t &info, float &iscore)
{
    unsigned st = (unsigned)(info >> ST_SHIFT);
    if (st > 4) st = 4;
    unsigned ifrag = (unsigned)(info >> FR_SHIFT) & 0xffff, ierr = (unsigned)(info >> ER_SHIFT) & 15,
             ifile = (unsigned)(info >> FI_SHIFT) & 63;
    uint64_t ipos = info & POS_MASK;
    bool differs = ((uint64_t)pos != ipos) || (fileid != ifile) || (frag != ifrag);
    bool take = false, nonu = false;
    if (st == ST_NOMATCH || st == ST_GAPPED) {
        take = true;
    } else if (SCORES) {
        if (score > iscore + eps) take = true;
        else if (st != ST_NONUNIQUE && (score > iscore - eps) && differs) nonu = true;
    } else {
        if (totalk < ierr) take = true;
        else if (st != ST_NONUNIQUE && totalk == ierr && differs) nonu = true;
    }
    if (take) {
        info = pack_record(inv ? ST_REVERSE : ST_STRAIGHT, frag, totalk, fileid, pos);
        if (SCORES) iscore = score;
    } else if (nonu) {
        info = (info & ~(7ull << ST_SHIFT)) | ((uint64_t)ST_NONUNIQUE << ST_SHIFT);
    }
}

// wave-aggregated append: the lanes of the wave that are active here take
// consecutive slots behind one atomic (ballot + prefix popcount).
__device__ __forceinline__ unsigned long long wave_append_slot(unsigned long long *counter)
{
    unsigned long long mask = __ballot(1);
    unsigned lane = threadIdx.x & 63;
    unsigned rank = __popcll(mask & ((1ull << lane) - 1ull));
    unsigned long long base = 0;
    if (rank == 0) base = atomicAdd(counter, (unsigned long long)__popcll(mask));
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base);
    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
    return (((unsigned long long)hi << 32) | lo) + rank;
}

// ---------------------------------------------------------------------------
// pack kernel
// ---------------------------------------------------------------------------
// One 256-thread workgroup packs 256 consecutive reads.  Their mapped symbols are one
// contiguous byte range of the batch: it is staged into LDS with coalesced 16-byte loads
// (reads are 36..256 bytes, so per-lane row reads from HBM would waste most of every
// sector), each lane then packs its own read out of LDS (row stride in dwords is odd for
// the common lengths => conflict free).  The qualities go through the same LDS buffer in a
// second phase and are written back as two oriented, 16-byte aligned rows per read.
#define PACK_RB 256

// reverse the order of the 2-bit symbols of a word
__device__ __forceinline__ uint64_t rev2(uint64_t x)
{
    x = __brevll(x);
    return ((x >> 1) & M55) | ((x & M55) << 1);
}

__device__ __forceinline__ void stage_bytes(uint8_t *lds, const uint8_t *src, uint64_t nbytes, uint32_t &skew)
{
    const uintptr_t a0 = (uintptr_t)src, a1 = a0 + nbytes, c0 = a0 & ~(uintptr_t)15;
    skew = (uint32_t)(a0 - c0);
    for (uintptr_t c = c0 + 16u * threadIdx.x; c < a1; c += 16u * PACK_RB) {
        const uint32_t lo = (uint32_t)(c - c0);
        if (c >= a0 && c + 16 <= a1) {
            *reinterpret_cast<uint4 *>(lds + lo) = *reinterpret_cast<const uint4 *>(c);
        } else { // partial chunk at either end of the range: never touch bytes outside it
            for (int b = 0; b < 16; ++b)
                if (c + b >= a0 && c + b < a1) lds[lo + b] = *reinterpret_cast<const uint8_t *>(c + b);
        }
    }
}

template <int W>
__global__ __launch_bounds__(PACK_RB) void pack_kernel(const uint8_t *__restrict__ bases, const uint8_t *__restrict__ qual,
                                                       const uint64_t *__restrict__ off, uint32_t upatl, uint64_t n,
                                                       uint32_t l, uint32_t QS, int want_q, uint64_t *__restrict__ words,
                                                       uint64_t *__restrict__ seeds, uint8_t *__restrict__ qrows,
                                                       uint32_t *__restrict__ patl_out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ uint32_t s_base[PACK_RB], s_patl[PACK_RB];
    const uint64_t r0 = (uint64_t)blockIdx.x * PACK_RB;
    const uint32_t nr = (uint32_t)min((uint64_t)PACK_RB, n - r0);
    const uint64_t o_begin = off ? off[r0] : r0 * (uint64_t)upatl;
    const uint64_t o_end = off ? off[r0 + nr] : (r0 + nr) * (uint64_t)upatl;
    const uint32_t t = threadIdx.x;
    const uint64_t r = r0 + t;
    uint64_t o0 = 0;
    uint32_t patl = 0;
    if (t < nr) {
        o0 = off ? off[r] : r * (uint64_t)upatl;
        patl = off ? (uint32_t)(off[r + 1] - o0) : upatl;
    }
    uint32_t skew;
    stage_bytes(lds, bases + o_begin, o_end - o_begin, skew);
    __syncthreads();

    // eligibility: matchUniqueImplementation.cpp:376-394
    bool ok = (t < nr) && (patl >= l) && (patl <= 32u * W);
    const uint32_t lb = skew + (uint32_t)(o0 - o_begin);
    if (ok) {
        const uint8_t *s = lds + lb;
        uint64_t ws[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            uint64_t w = 0;
            if (32u * j < patl) {
                const uint32_t nb = min(32u, patl - 32u * j);
                for (uint32_t b = 0; b < nb; ++b) {
                    const uint32_t c = s[32 * j + b];
                    if (c > 3) ok = false;
                    w |= (uint64_t)(c & 3) << (62 - 2 * b);
                }
            }
            ws[j] = w;
        }
        if (ok) {
            // transposed[i] = 3 - mapped[patl-1-i] (Pattern.hpp:105-128): reverse the 2-bit symbols of
            // the whole word string, shift out the pad, complement the valid symbols
            const uint32_t nw = (patl + 31) >> 5;
            const uint32_t pad = 64 * nw - 2 * patl; // 0..62
            uint64_t wr[W];
#pragma unroll
            for (int j = 0; j < W; ++j) {
                uint64_t a = 0, b2 = 0;
#pragma unroll
                for (int k = 0; k < W; ++k) { // R[j] = rev2(ws[nw-1-j]) with a register-static index
                    if ((uint32_t)k + (uint32_t)j + 1 == nw) a = rev2(ws[k]);
                    if ((uint32_t)k + (uint32_t)j + 2 == nw) b2 = rev2(ws[k]);
                }
                uint64_t v = pad ? ((a << pad) | (b2 >> (64 - pad))) : a;
                uint64_t valid = ((uint32_t)j + 1 < nw) ? ~0ull : ((uint32_t)j + 1 == nw ? (~0ull << pad) : 0ull);
                wr[j] = (v ^ ~0ull) & valid;
            }
            uint64_t *wo = words + r * (2 * W);
#pragma unroll
            for (int j = 0; j < W; ++j) { wo[j] = ws[j]; wo[W + j] = wr[j]; }
            // seed halves (s0 | s5) of both orientations.  straight: read[0..l); reverse:
            // revcomp(read[0..l)) (SignatureConstruction.hpp:347-410)
            const uint32_t h = l >> 1; // 2..32 bases
            const uint64_t hm = (h == 32) ? ~0ull : ((1ull << (2 * h)) - 1);
            const uint64_t shi = ws[0] >> (64 - 2 * h);
            uint64_t slo;
            if (2 * h <= 32) slo = (ws[0] >> (64 - 4 * h)) & hm;
            else slo = ((h == 32) ? ws[W > 1 ? 1 : 0] : (((ws[0] << (2 * h)) | (ws[W > 1 ? 1 : 0] >> (64 - 2 * h))) >> (64 - 2 * h)));
            const uint64_t rhi = (rev2(slo) >> (64 - 2 * h)) ^ hm; // revcomp of the second half comes first
            const uint64_t rlo = (rev2(shi) >> (64 - 2 * h)) ^ hm;
            uint64_t *so = seeds + r * 4;
            so[0] = shi; so[1] = slo; so[2] = rhi; so[3] = rlo;
        }
    }
    if (t < nr) patl_out[r] = ok ? patl : 0u;
    if (!want_q) return;

    // ---- qualities: two oriented rows per read (straight, reversed), 16-byte aligned ----
    s_base[t] = lb; s_patl[t] = ok ? patl : 0u;
    __syncthreads(); // everyone is done with the bases in LDS
    if (qual) stage_bytes(lds, qual + o_begin, o_end - o_begin, skew);
    __syncthreads();
    const uint32_t cpr = 2 * QS / 16; // 16-byte chunks per read
    for (uint32_t c = t; c < nr * cpr; c += PACK_RB) {
        const uint32_t rr = c / cpr, cc = c - rr * cpr;
        const uint32_t pl = s_patl[rr];
        if (!pl) continue;
        const uint32_t rev = (cc * 16 >= QS) ? 1u : 0u;
        const uint32_t i0 = cc * 16 - rev * QS;
        const uint8_t *qs = lds + s_base[rr];
        uint32_t v[4] = {0, 0, 0, 0};
#pragma unroll
        for (uint32_t u = 0; u < 16; ++u) {
            const uint32_t i = i0 + u;
            uint32_t q = 0;
            if (i < pl) q = qual ? qs[rev ? (pl - 1 - i) : i] : 30u; // PatternBase::getQuality = 30, Pattern.hpp:42-45
            v[u >> 2] |= q << (8 * (u & 3));
        }
        *reinterpret_cast<uint4 *>(qrows + (r0 + rr) * (2ull * QS) + cc * 16) = make_uint4(v[0], v[1], v[2], v[3]);
    }
}

__global__ void max_patl_kernel(const uint64_t *__restrict__ off, uint64_t n, uint32_t *out)
{
    uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t v = 0;
    if (r < n) v = (uint32_t)(off[r + 1] - off[r]);
    for (int d = 32; d; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    if ((threadIdx.x & 63) == 0) atomicMax(out, v);
}

// ---------------------------------------------------------------------------
// match kernel
// ---------------------------------------------------------------------------
template <int W, bool SCORES, bool ALL>
__global__ __launch_bounds__(256) void match_kernel(MatchArgs a)
{
    __shared__ double sLL[SCORES ? 1024 : 1];
    if (SCORES) {
        for (int i = threadIdx.x; i < 1024; i += 256) sLL[i] = a.LL[i];
        __syncthreads();
    }
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned cR = 0, cL = 0, cP = 0, cC = 0, cS = 0, cH = 0, cV = 0;
    uint32_t patl = (r < a.b.n_reads) ? a.b.patl[r] : 0u;

    if (patl) {
        cR = 1;
        const uint32_t l = a.l, bb = a.b_bits, half = l >> 1;
        const uint64_t mb = (bb >= 64) ? ~0ull : ((1ull << bb) - 1);
        const uint32_t restlen = patl - l;
        const uint32_t nw = (patl + 31) >> 5;
        const uint64_t lastmask = ~0ull << (64 - 2 * (patl - 32 * (nw - 1)));
        const float eps = (float)(a.filter_mult * (double)patl);
        const uint64_t *__restrict__ T = a.t.text;
        uint64_t info = 0;
        float iscore = 0.f;
        if (!ALL) {
            info = a.info[r];
            if (SCORES) iscore = a.score[r];
        }

        for (int inv = 0; inv < 2; ++inv) {
            uint64_t O[W];
            {
                const uint64_t *wp = a.b.words + r * (2 * W) + inv * W;
#pragma unroll
                for (int j = 0; j < W; ++j) O[j] = wp[j];
            }
            const uint64_t shi = a.b.seeds[r * 4 + inv * 2], slo = a.b.seeds[r * 4 + inv * 2 + 1];
            const uint32_t so = inv ? restlen : 0u; // RestMatch::getMatchOffset, RestMatch.hpp:84-89
            const uint64_t m0 = shi >> bb, m1 = shi & mb, m2 = slo >> bb, m3 = slo & mb;
            // one-entry memo of the last verified position of this strand: the same
            // window is reached through up to six lists; its verdict is a function of
            // (strand,pos) only, so re-deriving it from the memo is exact.
            uint32_t cpos = 0xffffffffu, ck = 0, cfrag = 0;
            float cscore = 1.0f;
            bool cok = false;

            for (int la = 0; la < 6; ++la) {
                if (!ALL && !SCORES && la == 1) {
                    // uni0s / uni0r early-out, matchUniqueImplementation.cpp:434-436,470-472
                    unsigned st = (unsigned)(info >> ST_SHIFT), er = (unsigned)(info >> ER_SHIFT) & 15;
                    if (st == (unsigned)(inv ? ST_REVERSE : ST_STRAIGHT) && er == 0) break;
                }
                // s_a of list la (SignatureConstruction.hpp:62-67) and which segments it covers
                uint64_t ma, mc;
                switch (la) {
                case 0: ma = m0; mc = m1; break;
                case 1: ma = m0; mc = m2; break;
                case 2: ma = m0; mc = m3; break;
                case 3: ma = m1; mc = m2; break;
                case 4: ma = m1; mc = m3; break;
                default: ma = m2; mc = m3; break;
                }
                const uint64_t sa = (bb >= 32) ? ((ma << bb) | mc) : (((ma << bb) | mc));
                const uint32_t prefix = (uint32_t)(sa >> a.ix.pshift);
                const uint32_t fp = (uint32_t)(sa >> a.ix.fshift);
                const uint32_t *__restrict__ bk = a.ix.bkt[la];
                uint32_t lo = bk[prefix], hi = bk[prefix + 1];
                const uint2 *__restrict__ E = a.ix.ent[la];
                cL++;
                if (hi - lo > 16) { // large bucket: lower_bound on the fingerprint first
                    uint32_t x = lo, y = hi;
                    while (x < y) {
                        uint32_t mid = x + ((y - x) >> 1);
                        cP++;
                        if (E[mid].x < fp) x = mid + 1; else y = mid;
                    }
                    lo = x;
                }
                for (uint32_t j = lo; j < hi; ++j) {
                    const uint2 e = E[j];
                    cP++;
                    if (e.x > fp) break;
                    if (e.x < fp) continue;
                    const uint32_t rpos = e.y;
                    // seed window of the genome at rpos, as the two halves (m0|m1), (m2|m3)
                    const uint64_t xhi = text_bits(T, rpos, half) ^ shi;
                    const uint64_t xlo = text_bits(T, (uint64_t)rpos + half, half) ^ slo;
                    const uint64_t dhi = ((xhi >> 1) | xhi) & M55, dlo = ((xlo >> 1) | xlo) & M55;
                    const unsigned k0 = __popcll(dhi >> bb), k1 = __popcll(dhi & mb);
                    const unsigned k2 = __popcll(dlo >> bb), k3 = __popcll(dlo & mb);
                    unsigned ka, kc;
                    switch (la) {
                    case 0: ka = k0; kc = k1; break;
                    case 1: ka = k0; kc = k2; break;
                    case 2: ka = k0; kc = k3; break;
                    case 3: ka = k1; kc = k2; break;
                    case 4: ka = k1; kc = k3; break;
                    default: ka = k2; kc = k3; break;
                    }
                    if (ka | kc) continue; // not a member of the reference's equal range (sig wider than prefix+32)
                    cC++;
                    const unsigned seedk = k0 + k1 + k2 + k3; // = diffcountpair(s_b, partner), match.hpp:386
                    if (seedk > a.seedkmax) continue;
                    cS++;
                    if (rpos < so) continue; // match.hpp:393
                    const uint32_t pos = rpos - so;
                    if (pos != cpos) {
                        cpos = pos;
                        cok = false;
                        cV++;
                        uint32_t frag;
                        if (!frag_valid(a.t, pos, patl, frag)) continue;
                        if (a.t.has_wild && !wild_free(a.t.wild, pos, patl)) continue;
                        // Hamming distance of the whole oriented read against text[pos, pos+patl):
                        // = seedk + RestMatch::computeDistance (RestMatch.hpp:39-81)
                        const uint64_t wi = pos >> 5;
                        const unsigned sh = 2u * (pos & 31);
                        uint64_t tw[W];
                        unsigned total = 0;
                        {
                            uint64_t cur = T[wi];
#pragma unroll
                            for (int j = 0; j < W; ++j) {
                                if ((uint32_t)j < nw) {
                                    uint64_t nxt = T[wi + j + 1];
                                    uint64_t al = sh ? ((cur << sh) | (nxt >> (64 - sh))) : cur;
                                    tw[j] = al;
                                    uint64_t x = al ^ O[j];
                                    uint64_t d = ((x >> 1) | x) & M55;
                                    if ((uint32_t)j == nw - 1) d &= lastmask;
                                    total += __popcll(d);
                                    cur = nxt;
                                } else {
                                    tw[j] = 0;
                                }
                            }
                        }
                        if (total > a.totalkmax) continue;
                        float sc = 1.0f; // ComputeScore<...,false>, ComputeScore.hpp:31-45
                        if (SCORES) {
                            // ComputeScore<...,true>::computeScore, ComputeScore.hpp:50-190: sequential
                            // FP64 sum in base order starting at 1.0, cast to float once.
                            double raw = 1.0;
                            const uint8_t *__restrict__ qp = a.b.qrows + r * (2ull * a.b.QS) + (uint64_t)inv * a.b.QS;
#pragma unroll
                            for (int j = 0; j < W; ++j) {
                                if ((uint32_t)j < nw) {
                                    const uint64_t twj = tw[j], owj = O[j];
                                    const uint32_t nb = min(32u, patl - 32u * j);
                                    for (uint32_t h = 0; h < nb; h += 16) {
                                        const uint4 qv = *reinterpret_cast<const uint4 *>(qp + 32 * j + h);
                                        const uint32_t qa[4] = {qv.x, qv.y, qv.z, qv.w};
                                        const uint32_t lim = min(16u, nb - h);
#pragma unroll
                                        for (uint32_t u = 0; u < 16; ++u) {
                                            if (u < lim) {
                                                const uint32_t bpos = h + u;
                                                const uint32_t ref = (uint32_t)(twj >> (62 - 2 * bpos)) & 3;
                                                const uint32_t rb = (uint32_t)(owj >> (62 - 2 * bpos)) & 3;
                                                const uint32_t q = (qa[u >> 2] >> (8 * (u & 3))) & 0xff;
                                                raw += sLL[((ref << 8) | (rb << 6) | q) & 1023];
                                            }
                                        }
                                    }
                                }
                            }
                            sc = (float)raw;
                        }
                        cok = true; ck = total; cscore = sc; cfrag = frag;
                    }
                    if (!cok) continue;
                    cH++; // one updater::update call, match.hpp:411
                    if (ALL) {
                        // unifyMatches (matchAllImplementation.cpp:150-161) only removes exact
                        // duplicates: the same (strand,pos) reached through a later list.  A hit is
                        // kept iff la is the first list whose two segments are mismatch free.
                        const bool z0 = !k0, z1 = !k1, z2 = !k2, z3 = !k3;
                        const int first = (z0 && z1) ? 0 : (z0 && z2) ? 1 : (z0 && z3) ? 2 : (z1 && z2) ? 3 : (z1 && z3) ? 4 : 5;
                        if (first == la) {
                            unsigned long long slot = wave_append_slot(a.raw_count);
                            if (slot < a.raw_cap)
                                a.raw[slot] = make_uint4((uint32_t)r, cpos, __float_as_uint(cscore),
                                                         ck | ((uint32_t)inv << 8) | (cfrag << 16));
                        }
                    } else {
                        fold_update<SCORES>(inv != 0, a.t.fileid, cpos, ck, cscore, eps, cfrag, info, iscore);
                    }
                }
            }
        }
        if (!ALL) {
            a.info[r] = info;
            if (SCORES) a.score[r] = iscore;
        }
    }

    // work counters: wave reduction, one atomic per wave and counter
    unsigned c[7] = {cR, cL, cP, cC, cS, cH, cV};
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        unsigned v = c[k];
        for (int d = 32; d; d >>= 1) v += __shfl_xor((int)v, d);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(a.counters + k, (unsigned long long)v);
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
template <int W>
static void launch_pack_w(real_hip_ctx *ctx, const uint8_t *d_bases, const uint8_t *d_qual, const uint64_t *d_off,
                          uint32_t upatl, uint64_t n, uint32_t QS)
{
    dim3 grid((unsigned)((n + PACK_RB - 1) / PACK_RB)), block(PACK_RB);
    const size_t lds_bytes = (size_t)PACK_RB * 32 * W + 32; // the block's contiguous byte range + alignment skew
    // above the 64 KiB default only for W = 8; gfx950 has 160 KiB of LDS per CU
    (void)hipFuncSetAttribute((const void *)pack_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(pack_kernel<W>, grid, block, lds_bytes, ctx->stream, d_bases, d_qual, d_off, upatl, n, ctx->prm.seedl, QS,
                       (int)(ctx->prm.scores != 0), (uint64_t *)ctx->words.p, (uint64_t *)ctx->seeds.p,
                       (uint8_t *)ctx->qrows.p, (uint32_t *)ctx->patl.p);
}

int rh_launch_pack(real_hip_ctx *ctx, const uint8_t *d_bases, const uint8_t *d_qual, const uint64_t *d_off,
                   uint32_t upatl, uint64_t n, uint32_t W, uint32_t QS)
{
    if (!n) return REAL_HIP_OK;
    RhTimer tm(ctx, REAL_HIP_K_PACK);
    switch (W) {
    case 1: launch_pack_w<1>(ctx, d_bases, d_qual, d_off, upatl, n, QS); break;
    case 2: launch_pack_w<2>(ctx, d_bases, d_qual, d_off, upatl, n, QS); break;
    case 3: launch_pack_w<3>(ctx, d_bases, d_qual, d_off, upatl, n, QS); break;
    case 4: launch_pack_w<4>(ctx, d_bases, d_qual, d_off, upatl, n, QS); break;
    case 5: launch_pack_w<5>(ctx, d_bases, d_qual, d_off, upatl, n, QS); break;
    case 6: launch_pack_w<6>(ctx, d_bases, d_qual, d_off, upatl, n, QS); break;
    case 7: launch_pack_w<7>(ctx, d_bases, d_qual, d_off, upatl, n, QS); break;
    case 8: launch_pack_w<8>(ctx, d_bases, d_qual, d_off, upatl, n, QS); break;
    default: return rh_fail(ctx, REAL_HIP_E_UNSUPPORTED, "read longer than REAL_HIP_MAX_PATL", hipSuccess);
    }
    RH_HIP(ctx, hipGetLastError());
    return REAL_HIP_OK;
}

template <int W>
static void launch_match_w(real_hip_ctx *ctx, const MatchArgs &a, bool all)
{
    dim3 grid((unsigned)((a.b.n_reads + 255) / 256)), block(256);
    const bool sc = ctx->prm.scores != 0;
    if (all) {
        if (sc) hipLaunchKernelGGL((match_kernel<W, true, true>), grid, block, 0, ctx->stream, a);
        else    hipLaunchKernelGGL((match_kernel<W, false, true>), grid, block, 0, ctx->stream, a);
    } else {
        if (sc) hipLaunchKernelGGL((match_kernel<W, true, false>), grid, block, 0, ctx->stream, a);
        else    hipLaunchKernelGGL((match_kernel<W, false, false>), grid, block, 0, ctx->stream, a);
    }
}

int rh_launch_match(real_hip_ctx *ctx, const MatchArgs &a, bool all)
{
    if (!a.b.n_reads) return REAL_HIP_OK;
    RhTimer tm(ctx, all ? REAL_HIP_K_MATCH_ALL : REAL_HIP_K_MATCH_UNIQUE);
    switch (a.b.W) {
    case 1: launch_match_w<1>(ctx, a, all); break;
    case 2: launch_match_w<2>(ctx, a, all); break;
    case 3: launch_match_w<3>(ctx, a, all); break;
    case 4: launch_match_w<4>(ctx, a, all); break;
    case 5: launch_match_w<5>(ctx, a, all); break;
    case 6: launch_match_w<6>(ctx, a, all); break;
    case 7: launch_match_w<7>(ctx, a, all); break;
    case 8: launch_match_w<8>(ctx, a, all); break;
    default: return rh_fail(ctx, REAL_HIP_E_UNSUPPORTED, "read longer than REAL_HIP_MAX_PATL", hipSuccess);
    }
    RH_HIP(ctx, hipGetLastError());
    return REAL_HIP_OK;
}

int rh_max_patl(real_hip_ctx *ctx, const uint64_t *d_off, uint64_t n, uint32_t *out)
{
    int rc = rh_reserve(ctx, ctx->maxpatl, 4);
    if (rc) return rc;
    RH_HIP(ctx, hipMemsetAsync(ctx->maxpatl.p, 0, 4, ctx->stream));
    if (n) hipLaunchKernelGGL(max_patl_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_off, n,
                              (uint32_t *)ctx->maxpatl.p);
    RH_HIP(ctx, hipMemcpyAsync(out, ctx->maxpatl.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return REAL_HIP_OK;
}
