/*
 * real_oracle.c -- CPU restatement of REAL's read-matching hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see real_oracle.h).  Written from scratch from a
 * reading of the reference; every function cites the reference lines (relative
 * to /root/reference/src) it follows.  Deliberately simple: the data layout is
 * the reference's (six sorted lists with cross pointers, 22-bit prefix tables,
 * rank dictionaries), so that the ordered sequence of candidates, filters and
 * updater::update calls is the reference's.
 */
#include "real_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#if defined(_OPENMP)
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* popcount (PopCountTable.hpp:57-131)                                       */
/* ------------------------------------------------------------------------ */
static inline unsigned popcount8(uint64_t v) { return (unsigned)__builtin_popcountll(v); }

/* PopCountTable.hpp:103-111 bitcountpair: number of non-zero 2-bit symbols */
static inline unsigned bitcountpair64(uint64_t n)
{
    return popcount8(((n >> 1) | n) & 0x5555555555555555ull);
}
/* PopCountTable.hpp:113-131 */
unsigned ora_diffcountpair32(uint32_t a, uint32_t b)
{
    uint32_t n = a ^ b;
    return popcount8(((n >> 1) | n) & 0x55555555ul);
}
unsigned ora_diffcountpair64(uint64_t a, uint64_t b) { return bitcountpair64(a ^ b); }

/* ------------------------------------------------------------------------ */
/* bit access (ERank222B.hpp:55-85 getBits64)                                */
/* ------------------------------------------------------------------------ */
static inline uint64_t get_bits64(const uint64_t *A, uint64_t offset, unsigned numbits)
{
    if (!numbits) return 0;
    uint64_t word = offset >> 6;
    unsigned bitskip = (unsigned)(offset & 63);
    unsigned restbits = 64 - bitskip;
    uint64_t b = A[word];
    b &= (~(uint64_t)0) >> (64 - restbits);
    if (restbits == numbits) return b;
    if (numbits < restbits) return b >> (restbits - numbits);
    numbits -= restbits;
    if (numbits) b = (b << numbits) | (A[word + 1] >> (64 - numbits));
    return b;
}

/* AutoTextArray.hpp:122-125 getTextWord(i,l) */
uint64_t ora_get_text_word(const ora_genome *g, uint64_t i, unsigned l)
{
    return get_bits64(g->text, i << 1, l << 1);
}

/* ERank222B.hpp:512-543 (construction), :562-566 (rank1) */
static void rank_build(const uint64_t *bits, uint64_t nwords, uint64_t **S, uint16_t **M)
{
    uint64_t nsuper = (nwords * 64 + 65535) >> 16;
    *S = (uint64_t *)calloc(nsuper ? nsuper : 1, sizeof(uint64_t));
    *M = (uint16_t *)calloc(nwords ? nwords : 1, sizeof(uint16_t));
    uint64_t c = 0;
    int64_t s = -1;
    for (uint64_t mi = 0; mi < nwords; ++mi) {
        if (((mi * 64) & 65535) == 0) (*S)[++s] = c;
        (*M)[mi] = (uint16_t)(c - (*S)[s]);
        c += popcount8(bits[mi]);
    }
}
uint64_t ora_rank1(const uint64_t *bits, const uint64_t *S, const uint16_t *M, uint64_t i)
{
    uint64_t mi = i >> 6;
    /* popcount8(val, i) = popcount8(val >> (63-i)), PopCountTable.hpp:98-101 */
    return S[i >> 16] + M[mi] + popcount8(bits[mi] >> (63 - (i - (mi << 6))));
}

/* ------------------------------------------------------------------------ */
/* genome (AutoTextArray.hpp:28-61, RangeVector.hpp:24-58)                    */
/* ------------------------------------------------------------------------ */
ora_genome *ora_genome_create(const uint8_t *sym, uint64_t n,
                              const uint64_t *frag_start, uint32_t n_frag)
{
    ora_genome *g = (ora_genome *)calloc(1, sizeof(ora_genome));
    g->n = n;
    /* getTextArray: numwords = ceil(2n/64); one spare word so that a window
       ending at the last symbol never reads past the allocation */
    g->n_words = (2 * n + 63) / 64;
    g->text = (uint64_t *)calloc(g->n_words + 2, sizeof(uint64_t));
    g->n_wwords = (n + 63) / 64;
    g->wild = (uint64_t *)calloc(g->n_wwords + 2, sizeof(uint64_t));
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t s = sym[i];
        g->text[i >> 5] |= (s & 3) << (62 - 2 * (i & 31)); /* writer.write(sym&3,2) MSB first */
        if (s > 3) { g->wild[i >> 6] |= (uint64_t)1 << (63 - (i & 63)); g->n_wild++; }
    }
    rank_build(g->wild, g->n_wwords, &g->wild_S, &g->wild_M);
    /* RangeVector::fillRange: one bit per symbol + terminal bit; set at fragment starts */
    g->n_frag = n_frag;
    g->frag_start = (uint64_t *)malloc((n_frag + 1) * sizeof(uint64_t));
    memcpy(g->frag_start, frag_start, (n_frag + 1) * sizeof(uint64_t));
    g->n_fwords = (n + 1 + 63) / 64;
    g->fbits = (uint64_t *)calloc(g->n_fwords + 1, sizeof(uint64_t));
    for (uint32_t j = 0; j <= n_frag; ++j) {
        uint64_t p = frag_start[j];
        g->fbits[p >> 6] |= (uint64_t)1 << (63 - (p & 63));
    }
    rank_build(g->fbits, g->n_fwords, &g->frag_S, &g->frag_M);
    return g;
}
void ora_genome_free(ora_genome *g)
{
    if (!g) return;
    free(g->text); free(g->wild); free(g->wild_S); free(g->wild_M);
    free(g->frag_start); free(g->fbits); free(g->frag_S); free(g->frag_M);
    free(g);
}
/* RangeVector.hpp:59-62 */
unsigned ora_position_to_range(const ora_genome *g, uint64_t pos)
{
    return (unsigned)(ora_rank1(g->fbits, g->frag_S, g->frag_M, pos) - 1);
}
/* RangeVector.hpp:63-80 */
int ora_is_position_valid(const ora_genome *g, uint64_t pos, unsigned patl)
{
    unsigned range = ora_position_to_range(g, pos);
    return (pos + patl) <= g->frag_start[range + 1];
}
/* AutoTextArray.hpp:167-172 */
int ora_is_dontcare_free(const ora_genome *g, uint64_t i, unsigned l)
{
    uint64_t bef = i ? ora_rank1(g->wild, g->wild_S, g->wild_M, i - 1) : 0;
    uint64_t aft = ora_rank1(g->wild, g->wild_S, g->wild_M, i + l - 1);
    return (aft - bef) == 0;
}

/* ------------------------------------------------------------------------ */
/* signatures (SignatureConstruction.hpp:47-67, 218-280, 347-410)            */
/* ------------------------------------------------------------------------ */
int ora_signature_mapped(unsigned l, const uint8_t *w, uint32_t vm[4])
{
    unsigned syms = l / 4; /* syms_m0..2 = l/nu, syms_m3 = l - 3(l/nu); l%4==0 */
    for (int j = 0; j < 4; ++j) {
        uint32_t m = 0;
        for (unsigned i = 0; i < syms; ++i) {
            uint8_t c = *(w++);
            if (c > 3) return 0;
            m = (m << 2) | c;
        }
        vm[j] = m;
    }
    return 1;
}
int ora_reverse_mapped_signature(unsigned l, const uint8_t *w, uint32_t vm[4])
{
    unsigned syms = l / 4;
    /* vm[3] = revcomp(read[0..syms)), vm[2] = revcomp(read[syms..2syms)), ... */
    for (int j = 3; j >= 0; --j) {
        const uint8_t *e = w + (size_t)(4 - j) * syms; /* one past the segment */
        uint32_t m = 0;
        for (unsigned i = 0; i < syms; ++i) {
            uint8_t c = *(--e);
            if (c > 3) return 0;
            m = (m << 2) | (uint32_t)(3 - c);
        }
        vm[j] = m;
    }
    return 1;
}
/* SignatureConstruction.hpp:62-67 */
void ora_signatures(unsigned l, const uint32_t m[4], uint64_t s[6])
{
    unsigned bits = 2 * (l / 4);
    uint64_t mask = (2 * bits >= 64) ? ~(uint64_t)0 : (((uint64_t)1 << (2 * bits)) - 1);
    s[0] = (((uint64_t)m[0] << bits) | m[1]) & mask;
    s[1] = (((uint64_t)m[0] << bits) | m[2]) & mask;
    s[2] = (((uint64_t)m[0] << bits) | m[3]) & mask;
    s[3] = (((uint64_t)m[1] << bits) | m[2]) & mask;
    s[4] = (((uint64_t)m[1] << bits) | m[3]) & mask;
    s[5] = (((uint64_t)m[2] << bits) | m[3]) & mask;
}

/* ------------------------------------------------------------------------ */
/* index build (MapTextFile.hpp:68-230, ListSet.hpp:41-63, u_sort.hpp:29-132,*/
/* getLookupTable.hpp:26-51)                                                 */
/* ------------------------------------------------------------------------ */
/* stable LSD radix sort of (key, payload idx); any stable sort gives the     */
/* reference's order (ParallelRadixSort.hpp:160-203 is stable).              */
static void stable_argsort(const uint64_t *key, uint32_t *perm, uint64_t n, unsigned bits)
{
    uint32_t *tmp = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    for (uint64_t i = 0; i < n; ++i) perm[i] = (uint32_t)i;
    uint64_t *hist = (uint64_t *)malloc(65537 * sizeof(uint64_t));
    for (unsigned sh = 0; sh < bits; sh += 16) {
        memset(hist, 0, 65537 * sizeof(uint64_t));
        for (uint64_t i = 0; i < n; ++i) hist[((key[perm[i]] >> sh) & 0xffff) + 1]++;
        for (unsigned d = 0; d < 65536; ++d) hist[d + 1] += hist[d];
        for (uint64_t i = 0; i < n; ++i) tmp[hist[(key[perm[i]] >> sh) & 0xffff]++] = perm[i];
        memcpy(perm, tmp, n * sizeof(uint32_t));
    }
    free(hist); free(tmp);
}

static void apply_perm_u64(uint64_t *a, const uint32_t *perm, uint64_t n)
{
    uint64_t *t = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < n; ++i) t[i] = a[perm[i]];
    memcpy(a, t, n * sizeof(uint64_t)); free(t);
}
static void apply_perm_u32(uint32_t *a, const uint32_t *perm, uint64_t n)
{
    uint32_t *t = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    for (uint64_t i = 0; i < n; ++i) t[i] = a[perm[i]];
    memcpy(a, t, n * sizeof(uint32_t)); free(t);
}

/* getLookupTable.hpp:26-51 */
static uint64_t *lookup_table(const uint64_t *sign, uint64_t masks, unsigned shift)
{
    uint64_t histsize = (uint64_t)1 << ORA_SAMPLE_BITS;
    uint64_t *h = (uint64_t *)calloc(2 * histsize, sizeof(uint64_t));
    uint32_t lasthash = 0;
    uint64_t lastindex = 0;
    for (uint64_t i = 0; i < masks; ++i) {
        uint32_t hash = (uint32_t)(sign[i] >> shift);
        if (hash != lasthash) {
            h[2 * (uint64_t)lasthash + 0] = lastindex; h[2 * (uint64_t)lasthash + 1] = i;
            lastindex = i; lasthash = hash;
        }
    }
    h[2 * (uint64_t)lasthash + 0] = lastindex; h[2 * (uint64_t)lasthash + 1] = masks;
    return h;
}

ora_index *ora_index_build(const ora_genome *g, unsigned l, uint64_t first_window, uint64_t max_entries)
{
    if (l < 4 || l > 64 || (l % 4)) return NULL;
    ora_index *ix = (ora_index *)calloc(1, sizeof(ora_index));
    ix->seedl = l; ix->sig_bits = l;
    ix->shift = (l >= ORA_SAMPLE_BITS) ? (l - ORA_SAMPLE_BITS) : 0; /* SignatureConstruction.hpp:76-86 */
    ix->first_window = first_window;

    /* MapTextFile::readNextSignature/readFullSignature (:118-179): the windows
       emitted are exactly the starts i with symbols [i,i+l) all in 0..3, in
       text order; fragment boundaries are NOT window boundaries. */
    uint64_t cap = 0, cnt = 0, ord = 0;
    uint32_t *wpos = NULL;
    uint64_t run = 0; /* current run of non-N symbols ending at i */
    int more = 0;
    for (uint64_t i = 0; i < g->n; ++i) {
        int isn = (int)((g->wild[i >> 6] >> (63 - (i & 63))) & 1);
        run = isn ? 0 : run + 1;
        if (run >= l) {
            if (ord >= first_window) {
                if (cnt == max_entries) { more = 1; break; }
                if (cnt == cap) { cap = cap ? 2 * cap : 1024; wpos = (uint32_t *)realloc(wpos, cap * sizeof(uint32_t)); }
                wpos[cnt++] = (uint32_t)(i + 1 - l);
            }
            ord++;
        }
    }
    ix->n = cnt; ix->have_next = more;
    uint64_t n = cnt;
    unsigned syms = l / 4;
    uint64_t m[4];
    for (int k = 0; k < 6; ++k) {
        ix->sign[k] = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
        ix->ptr[k]  = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
        ix->pos[k]  = (k < 3) ? (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t)) : NULL;
    }
    /* MapTextFile::readLists :211-216 */
    for (uint64_t i = 0; i < n; ++i) {
        for (int j = 0; j < 4; ++j) m[j] = ora_get_text_word(g, (uint64_t)wpos[i] + (uint64_t)j * syms, syms);
        uint32_t m32[4] = { (uint32_t)m[0], (uint32_t)m[1], (uint32_t)m[2], (uint32_t)m[3] };
        uint64_t s[6];
        ora_signatures(l, m32, s);
        for (int k = 0; k < 6; ++k) { ix->sign[k][i] = s[k]; ix->ptr[k][i] = (uint32_t)i; }
        for (int k = 0; k < 3; ++k) ix->pos[k][i] = wpos[i];
    }
    free(wpos);
    /* ListSet::sort :41-44 -> MaskSort::sort(full[i], base[2-i]) (u_sort.hpp:104-131) */
    uint32_t *perm = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    for (int a = 0; a < 3; ++a) {
        int b = 5 - a; /* full[a] <-> base[2-a] == list 5-a */
        stable_argsort(ix->sign[a], perm, n, ix->sig_bits);
        apply_perm_u64(ix->sign[a], perm, n); apply_perm_u32(ix->ptr[a], perm, n); apply_perm_u32(ix->pos[a], perm, n);
        for (uint64_t j = 0; j < n; ++j) ix->ptr[b][ix->ptr[a][j]] = (uint32_t)j;  /* updatePrev(a,b) */
        stable_argsort(ix->sign[b], perm, n, ix->sig_bits);
        apply_perm_u64(ix->sign[b], perm, n); apply_perm_u32(ix->ptr[b], perm, n);
        for (uint64_t j = 0; j < n; ++j) ix->ptr[a][ix->ptr[b][j]] = (uint32_t)j;  /* updatePrev(b,a) */
    }
    free(perm);
    for (int k = 0; k < 6; ++k) ix->lookup[k] = lookup_table(ix->sign[k], n, ix->shift);
    return ix;
}
/* compact form: see real_oracle.h */
ora_index *ora_index_from_entries(const ora_genome *g, unsigned l, uint64_t n, const uint32_t *const sign[6],
                                  const uint32_t *const pos[6])
{
    (void)g;
    if (l < 4 || l > 32 || (l % 4)) return NULL;
    ora_index *ix = (ora_index *)calloc(1, sizeof(ora_index));
    ix->seedl = l; ix->sig_bits = l; ix->n = n; ix->compact = 1;
    ix->shift = (l >= ORA_SAMPLE_BITS) ? (l - ORA_SAMPLE_BITS) : 0;
    uint64_t histsize = (uint64_t)1 << ORA_SAMPLE_BITS;
    for (int k = 0; k < 6; ++k) {
        ix->csign[k] = sign[k]; ix->cpos[k] = pos[k];
        uint64_t *h = (uint64_t *)calloc(2 * histsize, sizeof(uint64_t));
        const uint32_t *E = sign[k];
        /* [low,high) of every non-empty prefix, [0,0) otherwise (getLookupTable.hpp:26-51) */
#if defined(_OPENMP)
#pragma omp parallel for schedule(static)
#endif
        for (int64_t p = 0; p < (int64_t)histsize; ++p) {
            uint64_t key = (uint64_t)p << ix->shift, lo = 0, hi = n;
            while (lo < hi) { uint64_t mid = lo + ((hi - lo) >> 1); if ((uint64_t)E[mid] < key) lo = mid + 1; else hi = mid; }
            uint64_t first = lo, key2 = ((uint64_t)p + 1) << ix->shift;
            hi = n;
            while (lo < hi) { uint64_t mid = lo + ((hi - lo) >> 1); if ((uint64_t)E[mid] < key2) lo = mid + 1; else hi = mid; }
            if (lo > first) { h[2 * p] = first; h[2 * p + 1] = lo; }
        }
        ix->lookup[k] = h;
    }
    return ix;
}

void ora_index_free(ora_index *ix)
{
    if (!ix) return;
    for (int k = 0; k < 6; ++k) { free(ix->sign[k]); free(ix->ptr[k]); free(ix->pos[k]); free(ix->lookup[k]); }
    free(ix);
}
/* Mask.hpp:36-40 / :55-59 getPos */
uint32_t ora_index_getpos(const ora_index *ix, int list, uint64_t j)
{
    if (list < 3) return ix->pos[list][j];
    return ix->pos[5 - list][ix->ptr[list][j]];
}

/* ------------------------------------------------------------------------ */
/* scoring (Scoring.cpp:28-36, 61-133, 155-171, 204-208)                     */
/* ------------------------------------------------------------------------ */
static const double Q_PRB[65] = {
    1.0000000, 0.7943282, 0.6309573, 0.5011872, 0.3981072, 0.3162278, 0.2511886, 0.1995262, 0.1584893, 0.1258925,
    0.1000000, 0.0794328, 0.0630957, 0.0501187, 0.0398107, 0.0316228, 0.0251189, 0.0199526, 0.0158489, 0.0125893,
    0.0100000, 0.0079433, 0.0063096, 0.0050119, 0.0039811, 0.0031623, 0.0025119, 0.0019953, 0.0015849, 0.0012589,
    0.0010000, 0.0007943, 0.0006310, 0.0005012, 0.0003981, 0.0003162, 0.0002512, 0.0001995, 0.0001585, 0.0001259,
    0.0001000, 0.0000794, 0.0000631, 0.0000501, 0.0000398, 0.0000316, 0.0000251, 0.0000200, 0.0000158, 0.0000126,
    0.0000100, 0.0000079, 0.0000063, 0.0000050, 0.0000040, 0.0000032, 0.0000025, 0.0000020, 0.0000016, 0.0000013,
    0.0000010, 0.0000008, 0.0000006, 0.0000005, 0.0000004
};
void ora_scoring_defaults(double *similarity, double *gc, double *trans, double *err, double *gcmut_bias)
{
    *similarity = 0.995; *err = 0.00; *trans = 0.71; *gc = 0.41; *gcmut_bias = 2;
}
void ora_scoring_table(double similarity, double gcContent, double transitRate, double errorRate,
                       double gcMutBias, double LL[1024], double odds_out[16])
{
    volatile double odds[4][4]; /* Scoring.cpp is built with -ffloat-store */
    double bg[4];
    double transit = transitRate * (1 - similarity);
    double transver = (1 - transitRate) * (1 - similarity);
    bg[0] = (1 - gcContent) / 2; bg[3] = (1 - gcContent) / 2;
    bg[1] = gcContent / 2;       bg[2] = gcContent / 2;
    gcMutBias = gcMutBias * (1 - gcContent) / gcContent;
    odds[0][2] = transit / (gcMutBias + 1) / (1 - gcContent);
    odds[3][1] = transit / (gcMutBias + 1) / (1 - gcContent);
    odds[2][0] = transit / (gcMutBias + 1) / gcContent * gcMutBias;
    odds[1][3] = transit / (gcMutBias + 1) / gcContent * gcMutBias;
    odds[0][1] = transver / 2 / (gcMutBias + 1) / (1 - gcContent);
    odds[3][2] = transver / 2 / (gcMutBias + 1) / (1 - gcContent);
    odds[0][3] = transver / 2 / (gcMutBias + 1) / (1 - gcContent);
    odds[3][0] = transver / 2 / (gcMutBias + 1) / (1 - gcContent);
    odds[1][0] = transver / 2 / (gcMutBias + 1) / gcContent * gcMutBias;
    odds[2][3] = transver / 2 / (gcMutBias + 1) / gcContent * gcMutBias;
    odds[1][2] = transver / 2 / (gcMutBias + 1) / gcContent * gcMutBias;
    odds[2][1] = transver / 2 / (gcMutBias + 1) / gcContent * gcMutBias;
    odds[0][0] = 1 - odds[0][1] - odds[0][2] - odds[0][3];
    odds[3][3] = 1 - odds[3][0] - odds[3][1] - odds[3][2];
    odds[2][2] = 1 - odds[2][0] - odds[2][1] - odds[2][3];
    odds[1][1] = 1 - odds[1][0] - odds[1][2] - odds[1][3];
    for (int x = 0; x < 4; ++x)
        for (int y = 0; y < 4; ++y) {
            odds[x][y] *= 1 - errorRate;
            odds[x][y] /= bg[y];
        }
    for (unsigned c0 = 0; c0 < 4; ++c0)
        for (unsigned c1 = 0; c1 < 4; ++c1)
            for (unsigned q = 0; q < 64; ++q)
                /* Scoring::getScore(char,char,int) :155-171 */
                LL[(c0 << 8) | (c1 << 6) | q] = log(odds[c0][c1]) / log(2.0) * (1 - Q_PRB[q]);
    if (odds_out)
        for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) odds_out[4 * x + y] = odds[x][y];
}

/* ComputeScore.hpp:50-190: raw = 1.0; raw += LL[ref_i, read_i, q_i] for
   i = 0..patl-1 in order; straight: read=mapped[i], q=quality[i]; inverted:
   read=transposed[i]=3-mapped[patl-1-i], q=quality[patl-1-i]; result (float)raw.
   The reference walks the text word by word; symbol i is text[pos+i]. */
float ora_compute_score(const ora_genome *g, const double *LL, int inverted,
                        const uint8_t *mapped, const uint8_t *qual, uint32_t pos, unsigned patl)
{
    double raw = 1.0f;
    for (unsigned i = 0; i < patl; ++i) {
        uint64_t p = (uint64_t)pos + i;
        unsigned ref = (unsigned)((g->text[p >> 5] >> (62 - 2 * (p & 31))) & 3);
        unsigned pat, q;
        if (inverted) { pat = 3u - mapped[patl - 1 - i]; q = (unsigned)(int)(signed char)qual[patl - 1 - i]; }
        else          { pat = mapped[i];                 q = (unsigned)(int)(signed char)qual[i]; }
        raw += LL[(ref << 8) | (pat << 6) | q];
    }
    return (float)raw;
}

/* ------------------------------------------------------------------------ */
/* record (UniqueMatchInfo.hpp:29-56)                                        */
/* ------------------------------------------------------------------------ */
#define POSBITS 35
#define FILESHIFT 35
#define ERRSHIFT 41
#define FRAGSHIFT 45
#define STATESHIFT 61
uint64_t ora_record_pack(unsigned state, unsigned frag, unsigned errors, unsigned fileid, uint64_t pos)
{
    return ((uint64_t)state << STATESHIFT) | ((uint64_t)(frag & 0xffff) << FRAGSHIFT) |
           ((uint64_t)(errors & 15) << ERRSHIFT) | ((uint64_t)(fileid & 63) << FILESHIFT) |
           (pos & (((uint64_t)1 << POSBITS) - 1));
}
void ora_record_unpack(uint64_t d, unsigned *state, unsigned *frag, unsigned *errors, unsigned *fileid, uint64_t *pos)
{
    unsigned st = (unsigned)(d >> STATESHIFT);
    if (st > 4) st = 4; /* getState(): default -> NonUnique */
    if (state) *state = st;
    if (frag) *frag = (unsigned)((d >> FRAGSHIFT) & 0xffff);
    if (errors) *errors = (unsigned)((d >> ERRSHIFT) & 15);
    if (fileid) *fileid = (unsigned)((d >> FILESHIFT) & 63);
    if (pos) *pos = d & (((uint64_t)1 << POSBITS) - 1);
}

/* UpdateUniqueInfo<false>::update matchUniqueImplementation.cpp:97-160,
   UpdateUniqueInfo<true>::update  :179-248 */
void ora_update_unique(int scores, int inverted, unsigned fileid, uint32_t pos, unsigned totalk,
                       float score, float epsilon, unsigned fragid, uint64_t *info, float *info_score)
{
    unsigned st, frag, err, file; uint64_t ipos;
    ora_record_unpack(*info, &st, &frag, &err, &file, &ipos);
    unsigned newstate = inverted ? ORA_REVERSE : ORA_STRAIGHT;
    int take = 0, nonunique = 0;
    if (!scores) {
        switch (st) {
        case ORA_NOMATCH: case ORA_GAPPED: take = 1; break;
        case ORA_STRAIGHT: case ORA_REVERSE:
            if (totalk < err) take = 1;
            else if (totalk == err && ((pos != ipos) || (fileid != file) || (fragid != frag))) nonunique = 1;
            break;
        case ORA_NONUNIQUE:
            if (totalk < err) take = 1;
            break;
        }
    } else {
        float old = *info_score;
        switch (st) {
        case ORA_NOMATCH: case ORA_GAPPED: take = 1; break;
        case ORA_STRAIGHT: case ORA_REVERSE:
            if (score > old + epsilon) take = 1;
            else if ((score > old - epsilon) && ((pos != ipos) || (fileid != file) || (fragid != frag))) nonunique = 1;
            break;
        case ORA_NONUNIQUE:
            if (score > old + epsilon) take = 1;
            break;
        }
    }
    if (take) {
        *info = ora_record_pack(newstate, fragid, totalk, fileid, pos);
        if (scores) *info_score = score;
    } else if (nonunique) {
        *info = (*info & ~((uint64_t)7 << STATESHIFT)) | ((uint64_t)ORA_NONUNIQUE << STATESHIFT);
    }
}

/* ------------------------------------------------------------------------ */
/* per-read matcher                                                          */
/* ------------------------------------------------------------------------ */
typedef void (*hit_fn)(void *u, int inverted, uint32_t pos, unsigned totalk, unsigned seedk,
                       float score, unsigned fragid, int list);

typedef struct read_ctx {
    const ora_genome *g; const ora_index *ix; const ora_params *p;
    const uint8_t *mapped; const uint8_t *qual; unsigned patl;
    /* RestWordBuffer.hpp:33-78 */
    uint64_t bstraight[520], breverse[520]; /* reads of up to 16 k bases (the limit of this test harness, not of the reference) */
    unsigned fullrestwords, fracrestsyms;
    unsigned matchoffset[2]; int textrestoffset[2];
    ora_counters *c;
} read_ctx;

/* RestMatch.hpp:214-265 / :267-318, geometry :84-111 */
static void rest_setup(read_ctx *r)
{
    unsigned l = r->p->seedl, patl = r->patl, restlen = patl - l;
    r->fullrestwords = restlen / 32;
    r->fracrestsyms = restlen - r->fullrestwords * 32;
    r->matchoffset[0] = 0;        r->matchoffset[1] = restlen;
    r->textrestoffset[0] = (int)l; r->textrestoffset[1] = -(int)restlen;
    const uint8_t *s = r->mapped + l;
    const uint8_t *e = r->mapped + patl;
    unsigned nw = r->fullrestwords + (r->fracrestsyms ? 1 : 0);
    for (unsigned i = 0; i < nw; ++i) {
        unsigned cnt = (i < r->fullrestwords) ? 32 : r->fracrestsyms;
        uint64_t w = 0, v = 0;
        for (unsigned j = 0; j < cnt; ++j) { w = (w << 2) | (uint64_t)(*(s++) & 3); v = (v << 2) | (uint64_t)(3 - (*(--e) & 3)); }
        r->bstraight[i] = w; r->breverse[i] = v;
    }
}

/* RestMatch.hpp:39-81 computeDistance */
static unsigned rest_distance(const read_ctx *r, const uint64_t *words, uint32_t o)
{
    unsigned dist = 0;
    for (unsigned i = 0; i < r->fullrestwords; ++i, o += 32)
        dist += ora_diffcountpair64(words[i], ora_get_text_word(r->g, o, 32));
    if (r->fracrestsyms)
        dist += ora_diffcountpair64(words[r->fullrestwords], ora_get_text_word(r->g, o, r->fracrestsyms));
    return dist;
}

/* ::match, match.hpp:335-416 */
static void match_list(read_ctx *r, int a, uint64_t s_a, uint64_t s_b, int inverted, hit_fn fn, void *u)
{
    const ora_index *ix = r->ix; const ora_genome *g = r->g; const ora_params *p = r->p;
    int b = 5 - a;
    unsigned matchoffset = r->matchoffset[inverted];
    int textrestoffset = r->textrestoffset[inverted];
    const uint64_t *words = inverted ? r->breverse : r->bstraight;
    r->c->lookups++;
    unsigned prefix = (unsigned)(s_a >> ix->shift);
    uint64_t low = ix->lookup[a][2 * (uint64_t)prefix], high = ix->lookup[a][2 * (uint64_t)prefix + 1];
    /* std::equal_range on .sign */
    const uint64_t *S = ix->sign[a];
    const uint32_t *E = ix->compact ? ix->csign[a] : NULL;
    const uint32_t *EP = ix->compact ? ix->cpos[a] : NULL;
#define SIGN_AT(i) (E ? (uint64_t)E[(i)] : S[(i)])
    uint64_t lo = low, hi = high;
    while (lo < hi) { uint64_t mid = lo + ((hi - lo) >> 1); r->c->probes++; if (SIGN_AT(mid) < s_a) lo = mid + 1; else hi = mid; }
    uint64_t eq_lo = lo; hi = high;
    while (lo < hi) { uint64_t mid = lo + ((hi - lo) >> 1); r->c->probes++; if (SIGN_AT(mid) <= s_a) lo = mid + 1; else hi = mid; }
    uint64_t eq_hi = lo;
#undef SIGN_AT
    static const int SEG_A[6] = {0, 0, 0, 1, 1, 2}, SEG_C[6] = {1, 2, 3, 2, 3, 3};
    for (uint64_t q = eq_lo; q < eq_hi; ++q) {
        r->c->candidates++;
        uint64_t partner;
        if (E) { /* list_b[p->ptr].sign = the other two segments of the window at pos */
            unsigned syms = ix->seedl / 4;
            uint64_t wp = EP[q];
            partner = (ora_get_text_word(g, wp + (uint64_t)SEG_A[b] * syms, syms) << (2 * syms)) |
                      ora_get_text_word(g, wp + (uint64_t)SEG_C[b] * syms, syms);
        } else
            partner = ix->sign[b][ix->ptr[a][q]];
        unsigned seedk = (ix->sig_bits <= 32) ? ora_diffcountpair32((uint32_t)s_b, (uint32_t)partner)
                                              : ora_diffcountpair64(s_b, partner);
        if (seedk <= p->seedkmax) {
            r->c->seedpass++;
            uint32_t rpos = E ? EP[q] : ora_index_getpos(ix, a, q);
            if (rpos >= matchoffset) {
                uint32_t pos = rpos - matchoffset;
                if (ora_is_position_valid(g, pos, r->patl) && ora_is_dontcare_free(g, pos, r->patl)) {
                    uint32_t restpos = rpos + (uint32_t)textrestoffset;
                    unsigned restk = rest_distance(r, words, restpos);
                    unsigned totalk = seedk + restk;
                    if (totalk <= p->totalkmax) {
                        unsigned fragid = ora_position_to_range(g, pos);
                        float score = p->scores ? ora_compute_score(g, p->LL, inverted, r->mapped, r->qual, pos, r->patl) : 1.0f;
                        r->c->hits++;
                        fn(u, inverted, pos, totalk, seedk, score, fragid, a);
                    }
                }
            }
        }
    }
}

/* common front: eligibility (matchUniqueImplementation.cpp:376-394), RWB setup */
static int read_begin(read_ctx *r)
{
    if (r->patl < r->p->seedl) return 0; /* "Skipping pattern ... shorter than seed length." */
    for (unsigned i = 0; i < r->patl; ++i) if (r->mapped[i] > 3) return 0;
    if (r->patl - r->p->seedl > 32 * 519) return 0; /* oracle limit, far above any test */
    rest_setup(r);
    return 1;
}

/* ---- matchUnique -------------------------------------------------------- */
typedef struct uni_state {
    const ora_params *p; float epsilon; uint64_t info; float score;
    uint64_t read; ora_event *ev; uint64_t cap; uint64_t *nev;
} uni_state;

static void uni_hit(void *u, int inverted, uint32_t pos, unsigned totalk, unsigned seedk,
                    float score, unsigned fragid, int list)
{
    uni_state *s = (uni_state *)u;
    if (s->ev) {
        uint64_t k;
#if defined(_OPENMP)
#pragma omp atomic capture
#endif
        k = (*s->nev)++;
        if (k < s->cap) {
            ora_event e; memset(&e, 0, sizeof e);
            e.read = s->read; e.pos = pos; e.frag = fragid; e.score = score;
            e.inverted = (uint8_t)inverted; e.list = (uint8_t)list; e.totalk = (uint8_t)totalk; e.seedk = (uint8_t)seedk;
            s->ev[k] = e;
        }
    }
    ora_update_unique((int)s->p->scores, inverted, s->p->fileid, pos, totalk, score, s->epsilon, fragid, &s->info, &s->score);
}

/* UniqueMatcher::match matchUniqueImplementation.cpp:369-500 */
static void match_unique_read(read_ctx *r, uni_state *st)
{
    if (!read_begin(r)) return;
    r->c->reads++;
    unsigned l = r->p->seedl;
    uint32_t m[4]; uint64_t s[6];
    if (!ora_signature_mapped(l, r->mapped, m)) return;
    ora_signatures(l, m, s);
    st->epsilon = (float)(r->p->filter_mult * r->patl); /* RealOptions.hpp:74-77, float const epsilon :405 */
    int scores = (int)r->p->scores;
    for (int inv = 0; inv < 2; ++inv) {
        if (inv) { ora_reverse_mapped_signature(l, r->mapped, m); ora_signatures(l, m, s); }
        match_list(r, 0, s[0], s[5], inv, uni_hit, st);
        unsigned state, err;
        ora_record_unpack(st->info, &state, NULL, &err, NULL, NULL);
        int uni0 = (state == (unsigned)(inv ? ORA_REVERSE : ORA_STRAIGHT)) && (err == 0); /* :434, :470 */
        if (!uni0 || scores)
            for (int a = 1; a < 6; ++a) match_list(r, a, s[a], s[5 - a], inv, uni_hit, st);
    }
}

int ora_match_unique(const ora_genome *g, const ora_index *ix, const ora_params *p,
                     const uint8_t *bases, const uint8_t *qual, const uint64_t *offsets, uint64_t n_reads,
                     uint64_t *info, float *score, ora_counters *ctr,
                     ora_event *events, uint64_t event_cap, uint64_t *n_events)
{
    if (ix->seedl != p->seedl) return -2;
    ora_counters tot; memset(&tot, 0, sizeof tot);
    uint64_t nev = 0;
    int threads = (int)p->threads;
    if (events) threads = 1; /* the event stream is the ordered one */
#if defined(_OPENMP)
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
#endif
    {
        ora_counters c; memset(&c, 0, sizeof c);
#if defined(_OPENMP)
#pragma omp for schedule(dynamic, 1024)
#endif
        for (int64_t i = 0; i < (int64_t)n_reads; ++i) {
            read_ctx r; r.g = g; r.ix = ix; r.p = p; r.c = &c;
            r.mapped = bases + offsets[i]; r.qual = qual ? qual + offsets[i] : NULL;
            r.patl = (unsigned)(offsets[i + 1] - offsets[i]);
            uni_state st; st.p = p; st.info = info[i]; st.score = score ? score[i] : 0.0f;
            st.read = (uint64_t)i; st.ev = events; st.cap = event_cap; st.nev = &nev; st.epsilon = 0;
            match_unique_read(&r, &st);
            info[i] = st.info; if (score && p->scores) score[i] = st.score;
        }
#if defined(_OPENMP)
#pragma omp critical
#endif
        {
            tot.reads += c.reads; tot.lookups += c.lookups; tot.probes += c.probes;
            tot.candidates += c.candidates; tot.seedpass += c.seedpass; tot.hits += c.hits;
        }
    }
    if (ctr) *ctr = tot;
    if (n_events) *n_events = nev;
    return (events && nev > event_cap) ? -1 : 0;
}

/* ---- raw event stream (for pinning against oracle/_ref) ------------------ */
typedef struct ev_state { ora_event *ev; uint64_t cap, n; uint64_t read; int call_base; } ev_state;
static void ev_hit(void *u, int inverted, uint32_t pos, unsigned totalk, unsigned seedk,
                   float score, unsigned fragid, int list)
{
    ev_state *s = (ev_state *)u;
    if (s->n < s->cap) {
        ora_event e; memset(&e, 0, sizeof e);
        e.read = s->read; e.pos = pos; e.frag = fragid; e.score = score; e.inverted = (uint8_t)inverted;
        e.list = (uint8_t)(inverted * 6 + list); e.totalk = (uint8_t)totalk; e.seedk = (uint8_t)seedk;
        s->ev[s->n] = e;
    }
    s->n++;
}
int ora_match_events(const ora_genome *g, const ora_index *ix, const ora_params *p,
                     const uint8_t *bases, const uint8_t *qual, const uint64_t *offsets, uint64_t n_reads,
                     ora_event *events, uint64_t event_cap, uint64_t *n_events, ora_counters *ctr)
{
    if (ix->seedl != p->seedl) return -2;
    ora_counters c; memset(&c, 0, sizeof c);
    ev_state st; st.ev = events; st.cap = event_cap; st.n = 0; st.call_base = 0;
    for (uint64_t i = 0; i < n_reads; ++i) {
        read_ctx r; r.g = g; r.ix = ix; r.p = p; r.c = &c;
        r.mapped = bases + offsets[i]; r.qual = qual ? qual + offsets[i] : NULL;
        r.patl = (unsigned)(offsets[i + 1] - offsets[i]);
        st.read = i;
        if (!read_begin(&r)) continue;
        c.reads++;
        unsigned l = p->seedl; uint32_t m[4]; uint64_t s[6];
        if (!ora_signature_mapped(l, r.mapped, m)) continue;
        for (int inv = 0; inv < 2; ++inv) {
            if (inv) ora_reverse_mapped_signature(l, r.mapped, m);
            ora_signatures(l, m, s);
            for (int a = 0; a < 6; ++a) match_list(&r, a, s[a], s[5 - a], inv, ev_hit, &st);
        }
    }
    if (ctr) *ctr = c;
    if (n_events) *n_events = st.n;
    return st.n > event_cap ? -1 : 0;
}

/* ---- matchAll ----------------------------------------------------------- */
typedef struct all_state { ora_hit *v; uint64_t n, cap; uint64_t read; unsigned fileid; } all_state;

static void all_hit(void *u, int inverted, uint32_t pos, unsigned totalk, unsigned seedk,
                    float score, unsigned fragid, int list)
{
    (void)seedk; (void)list;
    all_state *s = (all_state *)u;
    if (s->n == s->cap) { s->cap = s->cap ? 2 * s->cap : 16; s->v = (ora_hit *)realloc(s->v, s->cap * sizeof(ora_hit)); }
    ora_hit h; memset(&h, 0, sizeof h);
    h.read = s->read; h.pos = pos; h.frag = fragid; h.score = score; h.inverted = (uint8_t)inverted;
    h.k = (uint8_t)totalk; h.fileid = (uint16_t)s->fileid;
    s->v[s->n++] = h;
}
/* operator< matchAllImplementation.cpp:122-136 */
static int hit_cmp(const void *pa, const void *pb)
{
    const ora_hit *A = (const ora_hit *)pa, *B = (const ora_hit *)pb;
    if (A->k != B->k) return A->k < B->k ? -1 : 1;
    if (A->pos != B->pos) return A->pos < B->pos ? -1 : 1;
    if (A->fileid != B->fileid) return A->fileid < B->fileid ? -1 : 1;
    if (A->frag != B->frag) return A->frag < B->frag ? -1 : 1;
    if ((double)A->score != (double)B->score) return (double)A->score < (double)B->score ? -1 : 1;
    if (A->inverted != B->inverted) return A->inverted < B->inverted ? -1 : 1;
    return 0;
}

int ora_match_all(const ora_genome *g, const ora_index *ix, const ora_params *p,
                  const uint8_t *bases, const uint8_t *qual, const uint64_t *offsets, uint64_t n_reads,
                  ora_hit *out, uint64_t cap, uint64_t *n_out, uint64_t *hit_offsets, ora_counters *ctr)
{
    if (ix->seedl != p->seedl) return -2;
    ora_counters c; memset(&c, 0, sizeof c);
    uint64_t total = 0;
    all_state st; memset(&st, 0, sizeof st); st.fileid = p->fileid;
    for (uint64_t i = 0; i < n_reads; ++i) {
        if (hit_offsets) hit_offsets[i] = total;
        read_ctx r; r.g = g; r.ix = ix; r.p = p; r.c = &c;
        r.mapped = bases + offsets[i]; r.qual = qual ? qual + offsets[i] : NULL;
        r.patl = (unsigned)(offsets[i + 1] - offsets[i]);
        st.n = 0; st.read = i;
        /* AllMatcher::match matchAllImplementation.cpp:261-355: all 12 lookups, no early-out */
        if (read_begin(&r)) {
            c.reads++;
            unsigned l = p->seedl; uint32_t m[4]; uint64_t s[6];
            if (ora_signature_mapped(l, r.mapped, m)) {
                for (int inv = 0; inv < 2; ++inv) {
                    if (inv) ora_reverse_mapped_signature(l, r.mapped, m);
                    ora_signatures(l, m, s);
                    for (int a = 0; a < 6; ++a) match_list(&r, a, s[a], s[5 - a], inv, all_hit, &st);
                }
            }
        }
        if (st.n) {
            /* unifyMatches :150-161 */
            qsort(st.v, st.n, sizeof(ora_hit), hit_cmp);
            uint64_t w = 0;
            for (uint64_t j = 0; j < st.n; ++j)
                if (j == 0 || hit_cmp(&st.v[j], &st.v[j - 1]) != 0) {
                    if (total + w < cap && out) out[total + w] = st.v[j];
                    w++;
                }
            total += w;
        }
    }
    if (hit_offsets) hit_offsets[n_reads] = total;
    free(st.v);
    if (ctr) *ctr = c;
    if (n_out) *n_out = total;
    return (total > cap) ? -1 : 0;
}
