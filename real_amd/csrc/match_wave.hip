// match_wave.hip -- the wave-cooperative matcher: ONE WAVE owns one read.
//
// The per-read matcher (match_kernel.hip) gives every read one lane, as the reference gives it one thread
// (match.hpp:383-413 is a serial loop over the equal range).  A read that lands on a low-complexity signature
// -- an equal range of 10^5..10^6 entries in a real genome -- would keep that lane, and with it its wave, busy
// for 0.1..1 s.  Such reads (an equal range or bucket scan longer than BIG_T entries; more verified locations or
// update() events than a lane can park), and reads longer than a lane's registers hold, are handed over to this
// kernel instead.  The read sits in the wave's LDS (words of 32 bases, both strands; its qualities), twelve lanes look
// up its twelve equal ranges at once, and then the 64 lanes of the wave take 64 entries of an equal range at a time:
//     entry -> partner filter (match.hpp:386 on the partner bits the entry carries) -> seed window on the text
//           -> position / fragment / N checks (match.hpp:390-398) -> Hamming verify (RestMatch.hpp:39-81)
//           -> score (ComputeScore.hpp:50-190), every lane for its own candidate;
// the survivors are compacted with __ballot: in entry order they are folded into the read's record by the whole
// wave (the fold is order dependent, SURVEY 8a10: lane order = entry order = the reference's candidate order), or
// appended to the hit list behind one atomic (ballot + prefix popcount).  Lists and strands are walked in the
// canonical order (strand, list 0..5, ascending position inside an equal range), so records, score bits, hit
// lists and the work counters L, C, S, H are those of the reference.
// All four table kinds are read here without their shortcuts (digests, fingerprints): bucket bounds, then the
// entries themselves.
#include "match_common.h"

#define WV_W RH_MAXW // words of a read that the lane-per-read kernels take (REAL_HIP_MAX_PATL bases): the short form of this kernel holds as many

struct WaveRange {
    const uint2 *E;      // entries {key, pos} in an entry array, or
    const uint32_t *row; // the 32 dwords of a bucket row (entries of 6 bytes behind the directory)
    uint32_t lo, cnt;    // first entry, entries to look at
    uint32_t key;        // what an entry's key is compared with
    uint32_t mode;       // 0: array, member iff (e.x >> pbits) == key, then the partner filter on e.x & pmask (pbits != 0)
                         // 1: row entries (narrow): partner filter on the 16 key bits
                         // 2: row entries (wide): 16-bit fingerprint of the key
                         // 3: overflow array of a row (narrow): partner filter on e.x & pmask
                         // 4: overflow array of a row (wide): e.x == key
    uint32_t partner;    // leading pbits of the read's partner signature s_b (modes 0, 1, 3)
    uint32_t counted;    // members of the reference's equal range known without looking at the entries (rows, narrow)
};

__device__ __forceinline__ uint64_t pick4(uint64_t m0, uint64_t m1, uint64_t m2, uint64_t m3, uint32_t x)
{
    return x == 0 ? m0 : x == 1 ? m1 : x == 2 ? m2 : m3;
}

// the equal range of list la for the oriented read with seed halves (shi, slo) (called by twelve lanes at once, each for its own list and strand)
__device__ __forceinline__ WaveRange wave_lookup(const MatchArgs &a, uint64_t shi, uint64_t slo, int la)
{
    WaveRange R;
    R.E = a.ix.ent[la]; R.row = nullptr; R.lo = 0; R.cnt = 0; R.key = 0; R.mode = 0; R.partner = 0; R.counted = 0;
    const uint32_t bb = a.b_bits, l = a.l;
    const uint64_t mb = (bb >= 64) ? ~0ull : ((1ull << bb) - 1);
    const uint64_t m0 = shi >> bb, m1 = shi & mb, m2 = slo >> bb, m3 = slo & mb;
    // s0..s5 = segments (0,1),(0,2),(0,3),(1,2),(1,3),(2,3), SignatureConstruction.hpp:62-67; partner = list 5-la
    const uint32_t xa = (0x940u >> (2 * la)) & 3u, xc = (0xfb9u >> (2 * la)) & 3u;
    const int lb = 5 - la;
    const uint32_t xb = (0x940u >> (2 * lb)) & 3u, xd = (0xfb9u >> (2 * lb)) & 3u;
    const uint64_t sa = (pick4(m0, m1, m2, m3, xa) << bb) | pick4(m0, m1, m2, m3, xc);
    const uint64_t sb = (pick4(m0, m1, m2, m3, xb) << bb) | pick4(m0, m1, m2, m3, xd);
    const uint32_t pbits = a.ix.pbits;
    R.partner = pbits ? (uint32_t)(sb >> (l - pbits)) : 0u;
    const uint32_t prefix = (uint32_t)(sa >> a.ix.pshift);
    if (a.ix.fine != 3) {
        // entry arrays in list order behind bucket starts (u32, or the .x of the 16-byte directory entries)
        uint32_t lo, hi;
        if (a.ix.fine == 0) { lo = a.ix.bkt[la][prefix]; hi = a.ix.bkt[la][prefix + 1]; }
        else {
            const uint4 *B = reinterpret_cast<const uint4 *>(a.ix.bkt[la]);
            lo = B[prefix].x; hi = B[prefix + 1].x;
        }
        const uint32_t f = (uint32_t)((sa >> a.ix.fshift) & ((a.ix.fbits >= 32) ? 0xffffffffull : ((1ull << a.ix.fbits) - 1)));
        // bounds of the key inside the bucket (entries of a bucket are sorted by key)
        uint32_t x = lo, y = hi;
        while (x < y) { const uint32_t mid = x + ((y - x) >> 1); if ((R.E[mid].x >> pbits) < f) x = mid + 1; else y = mid; }
        const uint32_t first = x;
        y = hi;
        while (x < y) { const uint32_t mid = x + ((y - x) >> 1); if ((R.E[mid].x >> pbits) <= f) x = mid + 1; else y = mid; }
        R.lo = first; R.cnt = x - first; R.key = f; R.mode = 0;
        return R;
    }
    // bucket rows: addressed by the mixed signature (real_hip_internal.h: rh_mix32 / rh_mix64)
    const bool wide = pbits == 0;
    uint32_t g, bucket;
    if (!wide) {
        const uint32_t gbits = a.ix.fbits;
        const uint32_t msa = rh_mix32((uint32_t)sa, l);
        bucket = msa >> gbits; g = msa & ((1u << gbits) - 1);
        R.key = R.partner;
    } else {
        const uint64_t msa = rh_mix64(sa, l);
        bucket = (uint32_t)(msa >> a.ix.pshift);
        R.key = (uint32_t)(msa >> a.ix.fshift);
        g = R.key >> 28;
    }
    const uint32_t *row = a.ix.bkt[la] + (uint64_t)bucket * 32;
    const uint32_t h0 = row[0], h1 = row[1];
    if ((h0 & h1) != 0xffffffffu) { // simple bucket: sixteen 4-bit counts, entries of 6 bytes
        uint32_t base = 0, cnt = 0;
        for (uint32_t q = 0; q < 16; ++q) {
            const uint32_t c = ((q < 8 ? h0 : h1) >> (4 * (q & 7))) & 15u;
            if (q < g) base += c;
            if (q == g) cnt = c;
        }
        R.row = row; R.lo = base; R.cnt = cnt; R.mode = wide ? 2u : 1u;
    } else { // complex bucket: entries in the overflow array, sixteen 8-bit counts (255 = "255 or more")
        const uint32_t o0 = row[2], tot = row[3];
        uint32_t base = 0, cnt = 0;
        bool sat = false;
        for (uint32_t q = 0; q <= g; ++q) {
            const uint32_t c = (row[4 + (q >> 2)] >> (8 * (q & 3))) & 255u;
            sat = sat || c == 255u;
            if (q < g) base += c; else cnt = c;
        }
        uint32_t first = o0 + base;
        if (sat) { // bounds of the key group by binary search
            const uint32_t gs = wide ? 28u : pbits;
            uint32_t x = o0, y = o0 + tot;
            while (x < y) { const uint32_t mid = x + ((y - x) >> 1); if ((R.E[mid].x >> gs) < g) x = mid + 1; else y = mid; }
            first = x; y = o0 + tot;
            while (x < y) { const uint32_t mid = x + ((y - x) >> 1); if ((R.E[mid].x >> gs) <= g) x = mid + 1; else y = mid; }
            cnt = x - first;
        }
        R.lo = first; R.cnt = cnt; R.mode = wide ? 4u : 3u;
    }
    if (!wide) R.counted = R.cnt; // (wide: counted when the text confirms the membership)
    return R;
}

// ---- the read as LDS words (any length up to REAL_HIP_MAX_PATL_LONG) ---------------------------------------------------
// The read sits in the wave's LDS as words of 32 bases, straight and reverse-complemented; every lane walks the words of
// its own candidate in run-time loops (an LDS word is the same address in all lanes: a broadcast).
#define WV_NWL (REAL_HIP_MAX_PATL_LONG / 32u)
#define WV_QL 2048u

// word j of the read (32 bases, MSB first) from the batch; *bad is set if a base is > 3 (byte input)
__device__ __forceinline__ uint64_t long_word(const MatchArgs &a, uint64_t o0, uint32_t patl, uint32_t j, bool *bad)
{
    const uint32_t nb = min(32u, patl - 32u * j);
    uint64_t w = 0;
    if (a.b.packed) {
        const uint64_t g0 = o0 + 32ull * j;          // first base of the word inside the batch
        const uint8_t *p = a.b.bases + (g0 >> 2);
        const uint32_t sh = 2u * (uint32_t)(g0 & 3);
        const uint32_t nbytes = (uint32_t)(((g0 & 3) + nb + 3) >> 2);
        for (uint32_t k = 0; k < 8 && k < nbytes; ++k) w |= (uint64_t)p[k] << (56 - 8 * k);
        if (sh) { w <<= sh; if (nbytes > 8) w |= (uint64_t)p[8] >> (8 - sh); }
    } else {
        const uint8_t *p = a.b.bases + o0 + 32ull * j;
        for (uint32_t i = 0; i < nb; ++i) { const uint32_t c = p[i]; if (c > 3) *bad = true; w |= (uint64_t)(c & 3) << (62 - 2 * i); }
    }
    if (nb < 32) w &= ~0ull << (64 - 2 * nb);
    return w;
}

// Hamming distance of the oriented read (LDS words cur[0..nw)) against text[pos, pos+patl); stops counting once it
// exceeds limit.  = seedk + RestMatch::computeDistance (RestMatch.hpp:39-81)
__device__ __forceinline__ uint32_t long_distance(const uint64_t *__restrict__ T, const uint64_t *cur, uint32_t pos, uint32_t nw, uint64_t lastmask,
                                                  uint32_t limit)
{
    const uint64_t wi = pos >> 5;
    const unsigned sh = 2u * (pos & 31);
    uint32_t total = 0;
    uint64_t t0 = T[wi];
    for (uint32_t j = 0; j < nw && total <= limit; ++j) {
        const uint64_t t1 = T[wi + j + 1];
        const uint64_t al = sh ? ((t0 << sh) | (t1 >> (64 - sh))) : t0;
        const uint64_t x = al ^ cur[j];
        uint64_t d = ((x >> 1) | x) & M55;
        if (j + 1 == nw) d &= lastmask;
        total += __popcll(d);
        t0 = t1;
    }
    return total;
}

// ComputeScore<...,true>::computeScore (ComputeScore.hpp:50-190) for a long read: the same sequential FP64 sum in base order
__device__ __forceinline__ float long_score(const double *sLL, const uint64_t *__restrict__ T, const uint64_t *cur, uint32_t pos, uint32_t patl,
                                            const uint8_t *qual, uint32_t inv)
{
    const uint64_t wi = pos >> 5;
    const unsigned sh = 2u * (pos & 31);
    const uint32_t nw = (patl + 31) >> 5;
    double raw = 1.0;
    uint64_t t0 = T[wi];
    for (uint32_t j = 0; j < nw; ++j) {
        const uint64_t t1 = T[wi + j + 1];
        const uint64_t al = sh ? ((t0 << sh) | (t1 >> (64 - sh))) : t0;
        const uint64_t rw = cur[j];
        const uint32_t nb = min(32u, patl - 32u * j);
        for (uint32_t u = 0; u < nb; ++u) {
            const uint32_t i = 32u * j + u;
            const uint32_t ref = (uint32_t)(al >> (62 - 2 * u)) & 3u, rb = (uint32_t)(rw >> (62 - 2 * u)) & 3u;
            const uint32_t q = qual ? (uint32_t)qual[inv ? patl - 1 - i : i] : 30u; // no qualities => 30 (Pattern.hpp:42-45)
            raw += sLL[((ref << 8) | (rb << 6) | q) & 1023];
        }
        t0 = t1;
    }
    return (float)raw;
}

// the score of the location at pos if the wave has computed it for this strand already (n = wave-uniform number of entries)
__device__ __forceinline__ bool memo_score(const uint32_t *mpos, const float *msc, uint32_t n, uint32_t pos, float &sc)
{
    bool found = false;
    for (uint32_t m = 0; m < n; ++m)
        if (mpos[m] == pos) { sc = msc[m]; found = true; }
    return found;
}

// LONG: the batch may hold reads longer than the registers of a lane hold (REAL_HIP_MAX_PATL): 32 KiB of LDS per workgroup
// for their words, three workgroups per CU.  Without them the kernel needs 10 KiB, and eight workgroups share a CU.
#define WV_MEMO 8u // scores of the locations a wave has scored for the strand it is on: a window is reached through up to six lists
template <bool SCORES, bool ALL, bool LONG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void match_wave_kernel(MatchArgs a)
{
    constexpr uint32_t QL = LONG ? WV_QL : 32u * WV_W; // qualities of a read that has at most this many bases are staged in LDS
    __shared__ double sLL[SCORES ? RH_LL_SLOTS : 1];
    __shared__ uint64_t sLong[4][2][LONG ? WV_NWL : WV_W]; // per wave: the words of its read, straight and reverse-complemented
    __shared__ __attribute__((aligned(16))) uint8_t sQual[4][16 + QL + 16]; // ... the qualities of the read (16 bytes in front and behind: LdsRow reads whole dwords)
    __shared__ uint32_t sMemoPos[4][2][WV_MEMO]; // (per wave and strand)
    __shared__ float sMemoSc[4][2][WV_MEMO];
    if (SCORES) {
        for (int i = threadIdx.x; i < 1024; i += 256) sLL[i] = a.LL[i];
        if (threadIdx.x == 0) sLL[RH_LL_ZERO] = 0.0;
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t n_items = (uint64_t)*a.ovf2_count;
    const uint64_t n_waves = (uint64_t)gridDim.x * 4;
    const uint32_t l = a.l, bb = a.b_bits, pbits = a.ix.pbits, pmask = pbits ? ((1u << pbits) - 1) : 0u, p16 = pbits < 16 ? pbits : 16;
    const uint64_t mb = (bb >= 64) ? ~0ull : ((1ull << bb) - 1);
    const uint64_t *__restrict__ T = a.t.text;
    unsigned cR = 0, cL = 0, cP = 0, cC = 0, cS = 0, cH = 0, cV = 0;

    for (uint64_t it = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); it < n_items; it += n_waves) { // (wave-uniform trip count)
        const uint64_t r = a.ovf2_list[it];
        const uint64_t o0 = a.b.off ? a.b.off[r] : r * (uint64_t)a.b.upatl;
        const uint64_t span = a.b.off ? a.b.off[r + 1] - o0 : (uint64_t)a.b.upatl; // (64 bits: a span of 2^32 and more must not alias to a short read)
        const uint32_t patl = span > (uint64_t)REAL_HIP_MAX_PATL_LONG ? 0u : (uint32_t)span; // (0: not eligible below; the lane matcher has raised the error flag)
        // The read sits in the wave's LDS as words of 32 bases, straight and reverse-complemented, whatever its length; every
        // lane walks the words of its own candidate in run-time loops (an LDS word is the same address in all lanes: a
        // broadcast).  Few registers per lane that way: many waves per CU, and this kernel lives on waves in flight.
        uint64_t *const sO = sLong[threadIdx.x >> 6][0], *const sR = sLong[threadIdx.x >> 6][1];
        const uint32_t nw = (patl + 31) >> 5;
        // The matcher hands over reads it has packed (eligible) and reads it could not even stage (too long for its
        // registers, or next to such a read): eligibility (matchUniqueImplementation.cpp:376-394) is settled here.
        bool elig = patl >= l && patl <= (LONG ? REAL_HIP_MAX_PATL_LONG : 32u * WV_W);
        if (!LONG && patl > 32u * WV_W && lane == 0) atomicOr(a.err_flags, 1u); // (the host picks LONG from the longest read of the batch: an error)
        if (elig && a.b.packed && a.b.nflags && ((a.b.nflags[r >> 3] >> (r & 7)) & 1)) elig = false;
        uint64_t O[2] = {0ull, 0ull}; // (the seed halves come from the first two words)
        if (elig) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // (the previous read's words are no longer in use)
            __builtin_amdgcn_wave_barrier();
            bool bad = false;
            for (uint32_t j = lane; j < nw; j += 64) sO[j] = long_word(a, o0, patl, j, &bad);
            elig = !__any(bad);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // revcomp: out[j] = complement of the 2-bit reversal of the words nw-1-j and nw-2-j, shifted by the pad
            const uint32_t pad = 64 * nw - 2 * patl;
            for (uint32_t j = lane; j < nw; j += 64) {
                const uint64_t x = rev2(sO[nw - 1 - j]), y = (j + 2 <= nw) ? rev2(sO[nw - 2 - j]) : 0ull;
                const uint64_t v = pad ? ((x << pad) | (y >> (64 - pad))) : x;
                sR[j] = ~v & ((j + 1 < nw) ? ~0ull : (~0ull << pad));
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            O[0] = sO[0];
            O[1] = nw >= 2 ? sO[1] : 0ull;
        }
        if (!elig) continue; // skipped like the reference skips it; the matcher has written its hit count 0
        // the qualities of the read: the scoring loop reads one per base -- from LDS, not one global load each
        uint8_t *const qst = sQual[threadIdx.x >> 6];
        const bool q_lds = SCORES && a.b.qual && patl <= QL;
        if (q_lds) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // (the previous read's are no longer in use)
            __builtin_amdgcn_wave_barrier();
            for (uint32_t i = lane; i < patl; i += 64) qst[16 + i] = a.b.qual[o0 + i];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const uint64_t lastmask = ~0ull << (64 - 2 * (patl - 32 * (nw - 1)));
        const float eps = (float)(a.filter_mult * (double)patl); // RealOptions.hpp:74-77
        uint64_t info = 0;
        float iscore = 0.f;
        if (!ALL) {
            info = a.info[r];
            if (SCORES) iscore = a.score[r];
        }
        uint32_t nhit = 0;
        uint64_t shi, slo, rhi, rlo;
        seed_halves<2>(O, l, shi, slo, rhi, rlo);
        if (lane == 0) cR++;
        // the twelve equal ranges of the read, one per lane and all at once: their bucket bounds / rows / binary searches are
        // chains of dependent loads that would otherwise follow each other (lane 6 inv + la holds the range of list la)
        WaveRange RL;
        {
            const uint32_t q = lane < 12 ? lane : 0u;
            RL = wave_lookup(a, q >= 6 ? rhi : shi, q >= 6 ? rlo : slo, (int)(q >= 6 ? q - 6 : q));
        }
        // one candidate of list la of strand inv, every lane for its own: entry -> partner filter -> seed window on the text ->
        // position / fragment / N checks -> Hamming distance (no score yet).  R, inv, la may differ from lane to lane.
        auto candidate = [&](const WaveRange &R, uint32_t i, bool cand, uint32_t inv, uint32_t la, uint32_t &pos, uint32_t &total, uint32_t &frag,
                             uint32_t &first) -> bool {
            const uint32_t xa = (0x940u >> (2 * la)) & 3u, xc = (0xfb9u >> (2 * la)) & 3u;
            const uint64_t hi_ = inv ? rhi : shi, lo_ = inv ? rlo : slo;
            const uint64_t *const cur = inv ? sR : sO;
            const uint32_t so = inv ? (patl - l) : 0u; // RestMatch::getMatchOffset, RestMatch.hpp:84-89
            uint32_t rpos = 0;
            if (cand) {
                if (R.mode == 1 || R.mode == 2) { // 6 bytes at halfword 4 + 3 * (lo + i) of the row
                    const uint32_t h = 4 + 3 * (R.lo + i);
                    const uint32_t d0 = R.row[h >> 1], d1 = R.row[(h >> 1) + 1];
                    const uint32_t key = (h & 1) ? (d0 >> 16) : (d0 & 0xffffu);
                    rpos = (h & 1) ? d1 : ((d0 >> 16) | (d1 << 16));
                    const uint32_t x = key ^ (R.key >> (pbits - p16));
                    cand = R.mode == 2 ? (key == rh_fp16(R.key)) : (__popc(((x >> 1) | x) & 0x5555u) <= a.seedkmax);
                } else {
                    const uint2 e = R.E[R.lo + i];
                    rpos = e.y;
                    if (R.mode == 4) cand = e.x == R.key;
                    else if (R.mode == 0 && !pbits) cand = true; // (the bounds were found on the whole key)
                    else {
                        // seed popcount filter (match.hpp:386) on the partner symbols the entry carries: more than
                        // seedkmax known mismatches => rejected without touching the text (exact: the full count can
                        // only be larger)
                        if (R.mode == 0) cC++; // a member of the reference's equal range
                        const uint32_t x = (e.x & pmask) ^ R.partner;
                        cand = __popc(((x >> 1) | x) & 0x55555555u) <= a.seedkmax;
                    }
                }
            }
            if (!cand) return false;
            const uint64_t wi0 = rpos >> 5;
            const U64x2 tt = load2(T + wi0);
            const uint64_t t2 = (l > 32) ? T[wi0 + 2] : 0ull;
            const unsigned sh0 = 2u * (rpos & 31);
            const uint64_t xhi = extract_bits(tt.a, tt.b, t2, sh0, l) ^ hi_;
            const uint64_t xlo = extract_bits(tt.a, tt.b, t2, sh0 + l, l) ^ lo_;
            const uint64_t dhi = ((xhi >> 1) | xhi) & M55, dlo = ((xlo >> 1) | xlo) & M55;
            const uint32_t k0 = __popcll(dhi >> bb), k1 = __popcll(dhi & mb), k2 = __popcll(dlo >> bb), k3 = __popcll(dlo & mb);
            const bool z0 = !k0, z1 = !k1, z2 = !k2, z3 = !k3;
            // a member of list la's equal range iff both segments the list is keyed on are mismatch free
            const bool za = xa == 0 ? z0 : xa == 1 ? z1 : z2, zc = xc == 1 ? z1 : xc == 2 ? z2 : z3;
            bool ok = za && zc;
            if (ok && !pbits) cC++;
            const uint32_t seedk = k0 + k1 + k2 + k3; // = diffcountpair(s_b, list_b[p->ptr].sign), match.hpp:386
            ok = ok && seedk <= a.seedkmax;
            if (ok) cS++;
            ok = ok && rpos >= so; // match.hpp:393
            if (ok) {
                pos = rpos - so;
                cV++;
                ok = frag_valid(a.t, pos, patl, frag) && !(a.t.has_wild && !wild_free(a.t.wild, pos, patl));
            }
            if (!ok) return false;
            // Hamming distance of the whole oriented read against text[pos, pos+patl)
            // = seedk + RestMatch::computeDistance (RestMatch.hpp:39-81)
            total = long_distance(T, cur, pos, nw, lastmask, a.totalkmax);
            if (total > a.totalkmax) return false;
            cH++; // one updater::update call per list, match.hpp:411
            first = (z0 && z1) ? 0u : (z0 && z2) ? 1u : (z0 && z3) ? 2u : (z1 && z2) ? 3u : (z1 && z3) ? 4u : 5u;
            return true;
        };
        const uint8_t *const qrow = !a.b.qual ? nullptr : (q_lds ? (const uint8_t *)qst + 16 : a.b.qual + o0);
        // ---- scores on / matchAll: the twelve equal ranges are ONE sequence of candidates in the canonical order (strand, list,
        // entry), taken 64 at a time, lane = candidate.  A read on a handful of copies of a locus -- what most of the reads
        // that come here are -- is one round: one chain of dependent loads (entry -> text -> verdict -> score) instead of
        // twelve; a read with a long equal range takes as many rounds as its candidates need, not one per list and 64 entries.
        // The survivors are folded in lane order, round by round = strand, list, entry = the reference's order of update()
        // calls.  (Without scores the 12-call driver may skip lists 1..5 of a strand after list 0,
        // matchUniqueImplementation.cpp:434-436: that mode walks the lists one by one below.)
        uint32_t c_mine = lane < 12 ? RL.cnt : 0u, c_before = 0, c_total = 0;
        {
            uint32_t run = 0;
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const uint32_t c = (uint32_t)__shfl((int)c_mine, q);
                if ((uint32_t)q == lane) c_before = run;
                run += c;
            }
            c_total = run;
        }
        if (SCORES || ALL) {
            if (lane < 12) { cL++; cP += RL.cnt; cC += RL.counted; }
            // where the twelve ranges start in the sequence: wave-uniform, compared in registers round after round
            uint32_t starts[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) starts[q] = (uint32_t)__builtin_amdgcn_readlane((int)c_before, q);
            WaveRange R = RL;
            uint32_t src_have = lane < 12 ? lane : 0xffffffffu; // the range whose description this lane holds in R
            uint32_t nmemo0 = 0, nmemo1 = 0; // scored locations of the read, per strand (wave-uniform)
#pragma unroll 1
            for (uint32_t f0 = 0; f0 < c_total; f0 += 64) {
                const uint32_t f = f0 + lane;
                // the (strand, list) whose range candidate f falls into: the last q with starts[q] <= f
                uint32_t src = 0;
#pragma unroll
                for (int q = 1; q < 12; ++q) src = starts[q] <= f ? (uint32_t)q : src;
                // its description comes from the lane that looked it up -- not again while a lane stays inside one range
                // (a long equal range is many rounds of the same list)
                if (__any(src != src_have)) {
                    const uint64_t e = (uint64_t)(uintptr_t)RL.E, w = (uint64_t)(uintptr_t)RL.row;
                    R.E = (const uint2 *)(uintptr_t)(((uint64_t)(uint32_t)__shfl((int)(e >> 32), (int)src) << 32) | (uint32_t)__shfl((int)e, (int)src));
                    R.row = (const uint32_t *)(uintptr_t)(((uint64_t)(uint32_t)__shfl((int)(w >> 32), (int)src) << 32) | (uint32_t)__shfl((int)w, (int)src));
                    R.lo = (uint32_t)__shfl((int)RL.lo, (int)src); R.cnt = (uint32_t)__shfl((int)RL.cnt, (int)src); R.key = (uint32_t)__shfl((int)RL.key, (int)src);
                    R.mode = (uint32_t)__shfl((int)RL.mode, (int)src); R.partner = (uint32_t)__shfl((int)RL.partner, (int)src); R.counted = (uint32_t)__shfl((int)RL.counted, (int)src);
                    src_have = src;
                }
                const uint32_t inv = src >= 6 ? 1u : 0u, la = src - 6 * inv;
                const uint32_t i = f - (uint32_t)__shfl((int)c_before, (int)src);
                uint32_t pos = 0, total = 0, frag = 0, first = 0;
                const bool hit = candidate(R, i, f < c_total, inv, la, pos, total, frag, first);
                float sc = 1.0f; // ComputeScore<...,false>, ComputeScore.hpp:31-45
                // A location reached through several lists in ONE round is scored by each of its lanes (they run the loop
                // together anyway); the scores of a round are kept for the rounds to come (the same window, the same score:
                // a read with a long equal range finds its locus again in every later list, a round each)
                bool scored = false;
                if (SCORES && hit && !memo_score(sMemoPos[threadIdx.x >> 6][inv], sMemoSc[threadIdx.x >> 6][inv], inv ? nmemo1 : nmemo0, pos, sc)) {
                    sc = long_score(sLL, T, inv ? sR : sO, pos, patl, qrow, inv);
                    scored = true;
                }
                if (SCORES) {
                    const unsigned long long sm0 = __ballot(scored && !inv), sm1 = __ballot(scored && inv);
                    if (sm0 | sm1) {
                        const unsigned long long below = (1ull << lane) - 1ull;
                        const uint32_t slot = inv ? nmemo1 + (uint32_t)__popcll(sm1 & below) : nmemo0 + (uint32_t)__popcll(sm0 & below);
                        if (scored && slot < WV_MEMO) { sMemoPos[threadIdx.x >> 6][inv][slot] = pos; sMemoSc[threadIdx.x >> 6][inv][slot] = sc; }
                        nmemo0 = min(WV_MEMO, nmemo0 + (uint32_t)__popcll(sm0));
                        nmemo1 = min(WV_MEMO, nmemo1 + (uint32_t)__popcll(sm1));
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                if (ALL) {
                    // unifyMatches (matchAllImplementation.cpp:150-161) removes exact duplicates: a (strand, pos) is kept
                    // from the first list whose two segments are mismatch free
                    const bool keep = hit && first == la;
                    if (keep) {
                        const unsigned long long slot = wave_append_slot(a.raw_count); // ballot + prefix popcount, one atomic
                        if (slot < a.raw_cap) a.raw[slot] = make_uint4((uint32_t)r, pos, __float_as_uint(sc), total | (inv << 8) | (frag << 16));
                    }
                    nhit += (uint32_t)__popcll(__ballot(keep));
                } else {
                    unsigned long long hm = __ballot(hit);
                    while (hm) { // UpdateUniqueInfo::update in candidate order; the record is wave-uniform
                        const int j = __ffsll((long long)hm) - 1;
                        hm &= hm - 1;
                        fold_update<SCORES>(__shfl((int)inv, j) != 0, a.t.fileid, (uint32_t)__shfl((int)pos, j), (unsigned)__shfl((int)total, j), __shfl(sc, j), eps,
                                            (unsigned)__shfl((int)frag, j), info, iscore);
                    }
                }
            }
        } else
        for (int inv = 0; inv < 2; ++inv) {
            uint32_t nmemo = 0; // scored locations of this strand (wave-uniform)
#pragma unroll 1
            for (int la = 0; la < 6; ++la) {
                if (!ALL && !SCORES && la == 1) {
                    // uni0s / uni0r early-out (matchUniqueImplementation.cpp:434-436, 470-472)
                    const unsigned st = (unsigned)(info >> ST_SHIFT), er = (unsigned)(info >> ER_SHIFT) & 15;
                    if (st == (unsigned)(inv ? ST_REVERSE : ST_STRAIGHT) && er == 0) break;
                }
                WaveRange R; // (broadcast from the lane that looked it up)
                {
                    const int src = 6 * inv + la;
                    const uint64_t e = (uint64_t)(uintptr_t)RL.E, w = (uint64_t)(uintptr_t)RL.row;
                    R.E = (const uint2 *)(uintptr_t)(((uint64_t)(uint32_t)__shfl((int)(e >> 32), src) << 32) | (uint32_t)__shfl((int)e, src));
                    R.row = (const uint32_t *)(uintptr_t)(((uint64_t)(uint32_t)__shfl((int)(w >> 32), src) << 32) | (uint32_t)__shfl((int)w, src));
                    R.lo = (uint32_t)__shfl((int)RL.lo, src); R.cnt = (uint32_t)__shfl((int)RL.cnt, src); R.key = (uint32_t)__shfl((int)RL.key, src);
                    R.mode = (uint32_t)__shfl((int)RL.mode, src); R.partner = (uint32_t)__shfl((int)RL.partner, src); R.counted = (uint32_t)__shfl((int)RL.counted, src);
                }
                if (lane == 0) { cL++; cP += R.cnt; cC += R.counted; }
#pragma unroll 1
                for (uint32_t c0 = 0; c0 < R.cnt; c0 += 64) {
                    const uint32_t i = c0 + lane;
                    uint32_t pos = 0, total = 0, frag = 0, first = 0;
                    const bool hit = candidate(R, i, i < R.cnt, (uint32_t)inv, (uint32_t)la, pos, total, frag, first);
                    bool scored = false;
                    float sc = 1.0f; // ComputeScore<...,false>, ComputeScore.hpp:31-45
                    if (SCORES && hit && !memo_score(sMemoPos[threadIdx.x >> 6][inv], sMemoSc[threadIdx.x >> 6][inv], nmemo, pos, sc)) {
                        sc = long_score(sLL, T, inv ? sR : sO, pos, patl, qrow, (uint32_t)inv);
                        scored = true;
                    }
                    if (SCORES) { // the scores computed in this round are kept for the lists to come (the same window, the same score)
                        const unsigned long long sm = __ballot(scored);
                        if (sm) {
                            const uint32_t slot = nmemo + (uint32_t)__popcll(sm & ((1ull << lane) - 1ull));
                            if (scored && slot < WV_MEMO) { sMemoPos[threadIdx.x >> 6][inv][slot] = pos; sMemoSc[threadIdx.x >> 6][inv][slot] = sc; }
                            nmemo = min(WV_MEMO, nmemo + (uint32_t)__popcll(sm));
                            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                    // ---- the survivors, compacted by ballot, in entry order
                    if (ALL) {
                        // unifyMatches (matchAllImplementation.cpp:150-161) removes exact duplicates: a (strand, pos) is kept
                        // from the first list whose two segments are mismatch free
                        const bool keep = hit && first == (uint32_t)la;
                        if (keep) {
                            const unsigned long long slot = wave_append_slot(a.raw_count); // ballot + prefix popcount, one atomic
                            if (slot < a.raw_cap) a.raw[slot] = make_uint4((uint32_t)r, pos, __float_as_uint(sc), total | ((uint32_t)inv << 8) | (frag << 16));
                        }
                        nhit += (uint32_t)__popcll(__ballot(keep));
                    } else {
                        unsigned long long hm = __ballot(hit);
                        while (hm) { // UpdateUniqueInfo::update in candidate order; the record is wave-uniform
                            const int j = __ffsll((long long)hm) - 1;
                            hm &= hm - 1;
                            fold_update<SCORES>(inv != 0, a.t.fileid, (uint32_t)__shfl((int)pos, j), (unsigned)__shfl((int)total, j), __shfl(sc, j), eps,
                                                (unsigned)__shfl((int)frag, j), info, iscore);
                        }
                    }
                }
            }
        }
        if (lane == 0) {
            if (!ALL) {
                a.info[r] = info;
                if (SCORES) a.score[r] = iscore;
            } else {
                a.hit_cnt[r] = nhit;
            }
        }
    }

    // work counters: wave reduction, one atomic per wave and counter; [7] = reads matched here
    unsigned c[8] = {cR, cL, cP, cC, cS, cH, cV, cR};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned v = wave_sum(c[k]);
        if (lane == 0 && v)
            atomicAdd(a.counters + (size_t)((blockIdx.x * 4u + (threadIdx.x >> 6)) & (RH_CSTRIPES - 1)) * 16 + k, (unsigned long long)v);
    }
}

// the reads the matcher handed over: far fewer than the batch; a fixed grid of waves strides over the list, whose
// length the kernel reads from device memory (no host round trip)
void rh_launch_match_wave(real_hip_ctx *ctx, const MatchArgs &a, bool all)
{
    const uint64_t waves = a.b.n_reads; // (at most one wave per read of the batch)
    const uint64_t blocks = (waves + 3) / 4;
    dim3 grid((unsigned)(blocks < 2048 ? blocks : 2048)), block(256); // eight workgroups per CU
    const bool sc = ctx->prm.scores != 0;
    const bool lng = a.b.maxpatl > 32u * RH_MAXW; // a read of the batch may be longer than the registers of a lane hold
#define RH_WAVE_LAUNCH(S, A)                                                                                  \
    do {                                                                                                      \
        if (lng) hipLaunchKernelGGL((match_wave_kernel<S, A, true>), grid, block, 0, ctx->stream, a);         \
        else     hipLaunchKernelGGL((match_wave_kernel<S, A, false>), grid, block, 0, ctx->stream, a);        \
    } while (0)
    if (all) { if (sc) RH_WAVE_LAUNCH(true, true); else RH_WAVE_LAUNCH(false, true); }
    else     { if (sc) RH_WAVE_LAUNCH(true, false); else RH_WAVE_LAUNCH(false, false); }
#undef RH_WAVE_LAUNCH
}
