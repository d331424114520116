// match_common.h -- device helpers shared by the per-read matcher (match_kernel.hip: one lane per read) and the
// wave-cooperative matcher of repeat-rich reads (match_wave.hip: one wave per read).
#pragma once
#include "kernel_common.h"

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

// the same from 2-bit packed bases: base g of the batch at bits 7-2(g%4)-1.. of byte g/4 (MSB first = the byte order of a
// word of the read, read big-endian).  row = the bytes from the one that holds the read's first base on; skew = that
// base's index inside its byte (0..3): the words are shifted left by 2*skew bits, the next byte fills in from the right.
template <int W, class Row>
__device__ __forceinline__ void pack_read_packed(const Row &row, uint32_t patl, uint32_t skew, uint64_t *O)
{
    const uint32_t nbytes = (skew + patl + 3) >> 2; // bytes that hold a base of this read
    const uint32_t nw = (patl + 31) >> 5, sh = 2 * skew;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        uint64_t w = 0;
        if ((uint32_t)j < nw) {
            const uint32_t hi = __builtin_bswap32(row.dword(8 * j, nbytes));
            const uint32_t lo = __builtin_bswap32(row.dword(8 * j + 4, nbytes));
            w = ((uint64_t)hi << 32) | lo;
            if (sh) w = (w << sh) | ((uint64_t)(row.dword(8 * j + 8, nbytes) & 0xffu) >> (8 - sh));
            const uint32_t rem = patl - 32u * j; // bases of this word that belong to the read
            if (rem < 32) w &= ~0ull << (64 - 2 * rem);
        }
        O[j] = w;
    }
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
// sLL holds the 1024 table entries and, behind them, one 0.0 (RH_LL_ZERO): positions behind the end of the read
// add that instead of being branched around -- x + 0.0 is x, the chain of sums is the reference's -- so that the
// sixteen positions of a step are straight-line code: eight table reads in flight, then their eight adds in order.
#define RH_LL_ZERO 1024u
#define RH_LL_SLOTS 1025u
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
        for (int h = 0; h < 2; ++h) {
            double x[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const uint32_t u = 8u * h + v;
                const uint32_t ref = (tr >> (30 - 2 * u)) & 3;
                const uint32_t rb = (rr >> (30 - 2 * u)) & 3;
                const uint32_t q = (qa[u >> 2] >> (8 * (u & 3))) & 0xff;
                const uint32_t idx = ((ref << 8) | (rb << 6) | q) & 1023;
                x[v] = sLL[u < lim ? idx : RH_LL_ZERO];
            }
#pragma unroll
            for (int v = 0; v < 8; ++v) raw += x[v];
        }
#pragma unroll
        for (int i = 0; i + 1 < NQ; ++i) { th[i] = th[i + 1]; oh[i] = oh[i + 1]; }
    }
    return (float)raw;
}
