// kernel_common.h -- device helpers shared by the gfx950 kernels of the read-matching path.
#pragma once
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

// sum of v over the 64 lanes of the wave, the same value in every lane.  Six adds on the data-parallel-primitive paths of
// the vector ALU (a scan inside the rows of 16, then the row totals are handed on) -- nothing goes through the LDS
// crossbar, where a butterfly of __shfl_xor queues behind the LDS traffic of the other waves of the CU.
__device__ __forceinline__ unsigned wave_sum(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true); // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true); // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true); // row_shr:8: lane 15 of a row = its total
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true); // row_bcast:15 into rows 1 and 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true); // row_bcast:31 into rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
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

// two consecutive u64 with one 16-byte request.  The address is only 8-byte aligned: global loads
// need dword alignment on gfx950, and one request instead of two keeps the vector-memory pipeline short.
struct __attribute__((packed, aligned(8))) U64x2 { uint64_t a, b; };
__device__ __forceinline__ U64x2 load2(const uint64_t *__restrict__ p) { return *reinterpret_cast<const U64x2 *>(p); }

// bits [o, o+nbits) of the 192-bit string t0|t1|t2 (o <= 127, 1 <= nbits <= 64), right aligned
__device__ __forceinline__ uint64_t extract_bits(uint64_t t0, uint64_t t1, uint64_t t2, unsigned o, unsigned nbits)
{
    const uint64_t x = (o < 64) ? t0 : t1, y = (o < 64) ? t1 : t2;
    o &= 63;
    const uint64_t v = o ? ((x << o) | (y >> (64 - o))) : x;
    return v >> (64 - nbits);
}

// reverse the order of the 2-bit symbols of a word
__device__ __forceinline__ uint64_t rev2(uint64_t x)
{
    x = __brevll(x);
    return ((x >> 1) & M55) | ((x & M55) << 1);
}

