// index_build.hip -- device-resident genome text and index.
//
//   * text packing: AutoTextArray's layout (AutoTextArray.hpp:28-61) from mapped symbols;
//   * index layout: each of the reference's six sorted lists (ListSet.hpp:23-31,
//     Mask.hpp:22-64) becomes an array of {fingerprint, position} in the same order,
//     plus a table of bucket starts keyed by the top `pb` signature bits.  `ptr`
//     and the partner list are not needed on the device: list_b[p->ptr].sign is
//     the other two seed segments of the same text window and is re-read from the
//     2-bit text at `pos`;
//   * device index build: window enumeration (MapTextFile.hpp:118-230),
//     six stable LSD radix sorts (rocPRIM; the index build is outside the hot path)
//     and the bucket tables;
//   * the matchAll post-pass (per-read ordering of unifyMatches,
//     matchAllImplementation.cpp:122-161).
#include "real_hip_internal.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

// ---------------------------------------------------------------------------
// text
// ---------------------------------------------------------------------------
__global__ void pack_text_kernel(const uint8_t *__restrict__ sym, uint64_t n, uint64_t *__restrict__ text,
                                 uint64_t *__restrict__ wild, unsigned long long *n_wild)
{
    uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // one wildcard word = 64 symbols
    uint64_t base = w * 64;
    if (base >= n) return;
    uint64_t t0 = 0, t1 = 0, wd = 0;
    for (int b = 0; b < 64; ++b) {
        uint64_t i = base + b;
        if (i < n) {
            uint64_t c = sym[i];
            if (c > 3) wd |= 1ull << (63 - b); // writeBit(utext[i] > 3)
            c &= 3;                              // writer.write(utext[i] & 0x3, 2)
            if (b < 32) t0 |= c << (62 - 2 * b); else t1 |= c << (62 - 2 * (b - 32));
        }
    }
    text[2 * w] = t0;
    text[2 * w + 1] = t1; // buffers are padded, 2w+1 is always inside
    wild[w] = wd;
    if (wd) atomicAdd(n_wild, (unsigned long long)__popcll(wd));
}

__global__ void count_wild_kernel(const uint64_t *__restrict__ wild, uint64_t nw, unsigned long long *n_wild)
{
    uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nw && wild[w]) atomicAdd(n_wild, (unsigned long long)__popcll(wild[w]));
}

int rh_pack_text(real_hip_ctx *ctx, const uint8_t *d_sym, uint64_t n)
{
    uint64_t nw = (n + 63) / 64;
    unsigned long long *d_cnt = (unsigned long long *)ctx->counters.p + (size_t)RH_CSTRIPES * 16; // scratch slot behind the stripes
    RH_HIP(ctx, hipMemsetAsync(d_cnt, 0, 8, ctx->stream));
    if (nw)
        hipLaunchKernelGGL(pack_text_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, ctx->stream, d_sym, n,
                           (uint64_t *)ctx->text.p, (uint64_t *)ctx->wild.p, d_cnt);
    RH_HIP(ctx, hipGetLastError());
    unsigned long long h = 0;
    RH_HIP(ctx, hipMemcpyAsync(&h, d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
    RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_wild = h;
    return REAL_HIP_OK;
}

int rh_count_wild(real_hip_ctx *ctx, uint64_t n)
{
    uint64_t nw = (n + 63) / 64;
    unsigned long long *d_cnt = (unsigned long long *)ctx->counters.p + (size_t)RH_CSTRIPES * 16;
    RH_HIP(ctx, hipMemsetAsync(d_cnt, 0, 8, ctx->stream));
    if (nw)
        hipLaunchKernelGGL(count_wild_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const uint64_t *)ctx->wild.p, nw, d_cnt);
    unsigned long long h = 0;
    RH_HIP(ctx, hipMemcpyAsync(&h, d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
    RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_wild = h;
    return REAL_HIP_OK;
}

// ---------------------------------------------------------------------------
// index layout from a sorted list
// ---------------------------------------------------------------------------
void rh_choose_tables(real_hip_ctx *ctx, uint64_t n_entries)
{
    const uint32_t l = ctx->prm.seedl, want = ctx->prm.table_kind;
    uint32_t pb = ctx->prm.prefix_bits;
    const bool auto_pb = (pb == 0);
    uint32_t lg = 0;
    while ((1ull << (lg + 1)) <= (n_entries ? n_entries : 1)) lg++;
    const bool big = lg >= 27 && want != 1;
    bool rows = want == 3;
    if (want == 0 && auto_pb && big && !ctx->no_rows) {
        // bucket rows are the faster layout (one HBM line per lookup) when they fit: 128 B x 2^pb x 6 lists plus the
        // build's transients (the full entry array of one list, the window positions, the bucket starts and overflow scans)
        uint32_t rpb = 1;
        while (rpb < 30 && (double)n_entries / (double)(1ull << rpb) > 11.5) rpb++;
        if (l <= 32 && rpb + 4 < l) rpb = l - 4;
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        const double need = 6.0 * 128.0 * (double)(1ull << rpb) + 12.0 * (double)n_entries + 12.0 * (double)(1ull << rpb);
        // (only when the rows are reasonably full: a 200 Mbp genome would take the same 2^(l-4) rows as a 3 Gbp one)
        rows = (l <= 32 ? rpb < l : rpb + 32 <= l) && (double)n_entries / (double)(1ull << rpb) >= 4.0 && need * 1.08 <= (double)total_b;
    }
    if (auto_pb) {
        if (l <= 32 && big) {
            // large index, 32-bit signatures: prefix = all signature bits but three ("fine" tables: the bucket
            // table also holds size and partner digest of the (at most eight) key groups of a bucket, so a
            // lookup lands on the reference's equal range without scanning, and on nothing at all when the
            // range is one chance entry); 16 B x 2^(l-3) per list
            pb = l - 3;
        } else if (l > 32 && big) {
            // large index, 64-bit signatures: mean bucket of 4..8 entries, described by fingerprints
            pb = lg - 2;
        } else {
            // mean bucket of 2..4 entries (the first two entries of every bucket are prefetched together)
            pb = lg > 1 ? lg - 1 : 1;
            if (pb < 8) pb = 8;
        }
    }
    if (rows && auto_pb) {
        // bucket rows: about 11 entries per 128-byte row of 20; 32-bit signatures: at most 16 signature values per row
        pb = 1;
        while (pb < 30 && (double)n_entries / (double)(1ull << pb) > 11.5) pb++;
        if (l <= 32) {
            if (pb + 4 < l) pb = l - 4;
            if (pb + 1 > l) pb = l > 1 ? l - 1 : 1;
        } else if (pb + 32 > l) pb = l - 32;
    }
    if (pb > l) pb = l; // a signature has seedl bits (two segments of seedl/4 bases)
    if (pb > 30) pb = 30;
    if (pb < 1) pb = 1;
    ctx->pb = pb;
    uint32_t pshift, fshift, fbits, pbits;
    rh_index_geometry(l, pb, &pshift, &fshift, &fbits, &pbits);
    if (want == 1) ctx->fine = 0;
    else if (rows && ((l <= 32 && l >= pb && l - pb >= 1 && l - pb <= 4) || (l > 32 && pb + 32 <= l))) ctx->fine = 3;
    else if (rh_is_fine(l, pb)) ctx->fine = 1;
    else if (pbits == 0 && (want == 2 || (auto_pb && big))) ctx->fine = 2;
    else ctx->fine = 0;
}

__device__ __forceinline__ uint64_t dev_text_bits(const uint64_t *__restrict__ T, uint64_t i, unsigned nb)
{
    uint64_t w = i >> 5;
    unsigned sh = 2u * (unsigned)(i & 31);
    uint64_t v = T[w] << sh;
    if (sh + 2 * nb > 64) v |= T[w + 1] >> (64 - sh);
    return v >> (64 - 2 * nb);
}

// signature of list `list` of the window at text position p (MapTextFile::readLists, MapTextFile.hpp:211-216)
__device__ __forceinline__ uint64_t window_signature(const uint64_t *__restrict__ T, uint64_t p, uint32_t l, int list)
{
    const uint32_t q = l >> 2, bb = 2 * q;
    const int sa_seg = (list < 3) ? 0 : (list < 5) ? 1 : 2;
    const int sc_seg = (list == 0) ? 1 : (list == 1 || list == 3) ? 2 : 3;
    return (dev_text_bits(T, p + (uint64_t)sa_seg * q, q) << bb) | dev_text_bits(T, p + (uint64_t)sc_seg * q, q);
}

// entry j of the device list: {signature bits below the bucket prefix | top bits of the partner
// signature, window start}.  The partner signature list_b[p->ptr].sign (match.hpp:386) is the
// signature of list 5-k of the same window; keeping its top bits lets the matcher discard nearly
// every chance candidate (seed popcount filter on the known symbols) without touching the text.
template <typename K>
__global__ void entries_kernel(const K *__restrict__ sign, const uint32_t *__restrict__ pos, uint64_t n, uint32_t l,
                               int list, const uint64_t *__restrict__ T, uint32_t pshift, uint32_t fshift, uint32_t fbits,
                               uint32_t pbits, uint32_t nbuckets, uint2 *__restrict__ ent, uint32_t *__restrict__ bkt)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t s = (uint64_t)sign[j];
    const uint32_t wp = pos[j];
    uint64_t x = (s >> fshift) & ((fbits >= 32) ? 0xffffffffull : ((1ull << fbits) - 1));
    if (pbits) {
        const uint64_t partner = window_signature(T, wp, l, 5 - list);
        x = (x << pbits) | (partner >> (l - pbits));
    }
    ent[j] = make_uint2((uint32_t)x, wp);
    uint32_t p = (uint32_t)(s >> pshift);
    int64_t pprev = j ? (int64_t)(uint32_t)((uint64_t)sign[j - 1] >> pshift) : -1;
    // bucket q starts at the first entry whose prefix is >= q (getLookupTable.hpp:26-51 keeps
    // [low,high) per prefix; consecutive starts carry the same information, empty = [x,x))
    for (int64_t q = pprev + 1; q <= (int64_t)p; ++q) bkt[q] = (uint32_t)j;
    if (j == n - 1)
        for (uint64_t q = (uint64_t)p + 1; q <= nbuckets; ++q) bkt[q] = (uint32_t)n;
}

// fine bucket table: uint4 {start, 96 bits = 8 x {size:4, digest:8}} per bucket, field g = key group g
// (= signature value g of the bucket).  size 15 means "15 or more": the matcher then finds that group's
// bounds by binary search inside the bucket.  digest = the leading (at most 8) partner-signature bits of
// the group's first entry: for a group of one entry the matcher evaluates the seed popcount filter on
// those symbols first and, if it already fails, never touches the entry (a whole 128-byte line saved
// for 3/4 of the chance candidates).
__global__ void fine_table_kernel(const uint32_t *__restrict__ bkt, const uint2 *__restrict__ ent, uint64_t nbuckets,
                                  uint32_t pbits, uint32_t fbits, uint4 *__restrict__ out)
{
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p > nbuckets) return;
    const uint32_t start = bkt[p];
    uint32_t fld[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (p < nbuckets) {
        const uint32_t end = bkt[p + 1];
        const uint32_t dbits = pbits < 8 ? pbits : 8, gmask = (1u << fbits) - 1, pmask = pbits ? ((1u << pbits) - 1) : 0u;
        for (uint32_t j = start; j < end; ++j) {
            const uint32_t x = ent[j].x;
            const uint32_t k = (x >> pbits) & gmask;
            uint32_t f = 0;
#pragma unroll
            for (int g = 0; g < 8; ++g) if (k == (uint32_t)g) f = fld[g];
            if ((f & 15u) == 0) f = (((x & pmask) >> (pbits - dbits)) << 4) | 1u;
            else if ((f & 15u) < RH_FINE_SAT) f++;
#pragma unroll
            for (int g = 0; g < 8; ++g) if (k == (uint32_t)g) fld[g] = f;
        }
    }
    uint64_t lo = 0;
#pragma unroll
    for (int g = 0; g < 5; ++g) lo |= (uint64_t)fld[g] << (12 * g);
    lo |= (uint64_t)(fld[5] & 15u) << 60;
    const uint32_t hi = (fld[5] >> 4) | (fld[6] << 8) | (fld[7] << 20);
    out[p] = make_uint4(start, (uint32_t)lo, (uint32_t)(lo >> 32), hi);
}

// fingerprint bucket table (entries hold a 32-bit key, the signature is wider): uint4 {start, count:8 |
// fingerprint:11 x 8}.  count saturates at 255; only the first RH_FP_SLOTS entries are described, a bucket
// with more is searched by key.
__global__ void fp_table_kernel(const uint32_t *__restrict__ bkt, const uint2 *__restrict__ ent, uint64_t nbuckets,
                                uint4 *__restrict__ out)
{
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p > nbuckets) return;
    const uint32_t start = bkt[p];
    uint32_t cnt = 0;
    uint64_t lo = 0, hi = 0; // 96-bit string: count in bits 0..7, fingerprint j in bits 8+11j ..
    if (p < nbuckets) {
        const uint32_t end = bkt[p + 1];
        cnt = end - start;
        for (uint32_t j = 0; j < cnt && j < RH_FP_SLOTS; ++j) {
            const uint64_t f = rh_fp11(ent[start + j].x);
            const uint32_t b = 8 + 11 * j;
            if (b < 64) { lo |= f << b; if (b + 11 > 64) hi |= f >> (64 - b); }
            else hi |= f << (b - 64);
        }
        lo |= cnt < 255 ? cnt : 255;
    }
    out[p] = make_uint4(start, (uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi);
}

// ---------------------------------------------------------------------------
// bucket rows (table kind 3): one 128-byte row per bucket holds the directory AND the entries, so that a
// lookup is ONE line of HBM, fetched by eight lanes with one coalesced request (match_lists_rows).
//   simple bucket (at most RH_ROW_CAP entries, every key group at most 15):
//     u64  sixteen 4-bit counts, nibble g = entries of key group g (= signature value g of the bucket)
//     then the entries in list order, 6 bytes each: u16 leading partner-signature bits, u32 position
//   complex bucket: u64 all ones, u32 first entry in the overflow array, u32 entries, sixteen u8 group counts
//     (255 = "255 or more": bounds by binary search); its entries live in the overflow array as {key, pos}
// ---------------------------------------------------------------------------
// (pbits == 0: wide signatures, the entries hold a 32-bit key; key group = its leading four bits)
__device__ __forceinline__ void bucket_groups(const uint2 *__restrict__ ent, uint32_t start, uint32_t end, uint32_t pbits, uint32_t gmask,
                                              uint32_t cnt[16])
{
#pragma unroll
    for (int g = 0; g < 16; ++g) cnt[g] = 0;
    for (uint32_t j = start; j < end; ++j) {
        const uint32_t k = pbits ? ((ent[j].x >> pbits) & gmask) : (ent[j].x >> 28);
#pragma unroll
        for (int g = 0; g < 16; ++g) if (k == (uint32_t)g) cnt[g]++;
    }
}

__global__ void rows_overflow_kernel(const uint32_t *__restrict__ bkt, const uint2 *__restrict__ ent, uint64_t nbuckets, uint32_t pbits,
                                     uint32_t fbits, uint32_t *__restrict__ ovf_cnt)
{
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p > nbuckets) return;
    uint32_t out = 0;
    if (p < nbuckets) {
        const uint32_t start = bkt[p], end = bkt[p + 1], c = end - start;
        bool complex_ = c > RH_ROW_CAP;
        if (!complex_ && c > 15) {
            uint32_t cnt[16];
            bucket_groups(ent, start, end, pbits, fbits >= 32 ? 15u : (1u << fbits) - 1, cnt);
#pragma unroll
            for (int g = 0; g < 16; ++g) complex_ = complex_ || cnt[g] > 15;
        }
        out = complex_ ? c : 0u;
    }
    ovf_cnt[p] = out;
}

__global__ void rows_fill_kernel(const uint32_t *__restrict__ bkt, const uint2 *__restrict__ ent, uint64_t nbuckets, uint32_t pbits,
                                 uint32_t fbits, const uint32_t *__restrict__ ovf_start, uint4 *__restrict__ rows, uint2 *__restrict__ ovf)
{
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nbuckets) return;
    const uint32_t start = bkt[p], end = bkt[p + 1], c = end - start;
    const uint32_t os = ovf_start[p], oc = ovf_start[p + 1] - os;
    uint32_t w[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) w[i] = 0;
    uint32_t cnt[16];
    bucket_groups(ent, start, end, pbits, fbits >= 32 ? 15u : (1u << fbits) - 1, cnt);
    if (oc) { // complex
        w[0] = w[1] = 0xffffffffu;
        w[2] = os; w[3] = c;
#pragma unroll
        for (int g = 0; g < 16; ++g) w[4 + (g >> 2)] |= (cnt[g] < 255 ? cnt[g] : 255u) << (8 * (g & 3));
        for (uint32_t j = 0; j < c; ++j) ovf[os + j] = ent[start + j];
    } else {
#pragma unroll
        for (int g = 0; g < 16; ++g) w[g >> 3] |= cnt[g] << (4 * (g & 7));
        const uint32_t p16 = pbits < 16 ? pbits : 16;
        const uint32_t pmask = pbits ? ((1u << pbits) - 1) : 0u;
        for (uint32_t j = 0; j < c; ++j) {
            const uint2 e = ent[start + j];
            const uint32_t key = pbits ? ((e.x & pmask) >> (pbits - p16)) : rh_fp16(e.x);
            // 6 bytes at byte 8 + 6j: halfwords 4+3j (key), 5+3j (pos low), 6+3j (pos high)
            const uint32_t hw[3] = {key & 0xffffu, e.y & 0xffffu, e.y >> 16};
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const uint32_t h = 4 + 3 * j + t;
                const uint32_t v = hw[t] << (16 * (h & 1));
#pragma unroll
                for (int i = 2; i < 32; ++i) if ((h >> 1) == (uint32_t)i) w[i] |= v;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) rows[p * 8 + i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// ordered traversal of the rows: per-bucket sizes, then (after a scan) the entries in list order
__global__ void rows_sizes_kernel(const uint4 *__restrict__ rows, uint64_t nbuckets, uint32_t *__restrict__ size)
{
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p > nbuckets) return;
    uint32_t c = 0;
    if (p < nbuckets) {
        const uint4 h = rows[p * 8];
        if (h.x == 0xffffffffu && h.y == 0xffffffffu) c = h.w;
        else {
            const uint64_t hd = (uint64_t)h.x | ((uint64_t)h.y << 32);
            for (int g = 0; g < 16; ++g) c += (uint32_t)(hd >> (4 * g)) & 15u;
        }
    }
    size[p] = c;
}
__global__ void rows_unpack_kernel(const uint4 *__restrict__ rows, const uint2 *__restrict__ ovf, uint64_t nbuckets,
                                   const uint32_t *__restrict__ off, uint2 *__restrict__ out)
{
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nbuckets) return;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(rows + p * 8);
    const uint32_t o = off[p], c = off[p + 1] - o;
    if (w[0] == 0xffffffffu && w[1] == 0xffffffffu) {
        for (uint32_t j = 0; j < c; ++j) out[o + j] = ovf[w[2] + j];
    } else {
        const uint16_t *hw = reinterpret_cast<const uint16_t *>(w);
        for (uint32_t j = 0; j < c; ++j)
            out[o + j] = make_uint2(hw[4 + 3 * j], (uint32_t)hw[5 + 3 * j] | ((uint32_t)hw[6 + 3 * j] << 16));
    }
}

// entries of list `list` in list order as {key, pos} (key = the row's 16 partner bits, or the overflow entry's key) and the
// bucket starts; either output may be null.  Used by index_download / index_export.
int rh_rows_unpack(real_hip_ctx *ctx, int list, uint2 *d_entries, uint32_t *d_starts)
{
    const uint64_t nb = 1ull << ctx->pb;
    int rc;
    ScopedBuf sizes(ctx), offs(ctx);
    if ((rc = rh_reserve(ctx, sizes, (nb + 1) * 4))) return rc;
    if ((rc = rh_reserve(ctx, offs, (nb + 1) * 4))) return rc;
    hipLaunchKernelGGL(rows_sizes_kernel, dim3((unsigned)((nb + 1 + 255) / 256)), dim3(256), 0, ctx->stream, (const uint4 *)ctx->bkt[list].p,
                       nb, (uint32_t *)sizes.p);
    size_t tmp = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, tmp, (uint32_t *)sizes.p, (uint32_t *)offs.p, 0u, (size_t)(nb + 1), rocprim::plus<uint32_t>(), ctx->stream);
    if (e == hipSuccess && !(rc = rh_reserve(ctx, ctx->sort_tmp, tmp ? tmp : 8)))
        e = rocprim::exclusive_scan(ctx->sort_tmp.p, tmp, (uint32_t *)sizes.p, (uint32_t *)offs.p, 0u, (size_t)(nb + 1), rocprim::plus<uint32_t>(), ctx->stream);
    if (e == hipSuccess && !rc && d_entries)
        hipLaunchKernelGGL(rows_unpack_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, ctx->stream, (const uint4 *)ctx->bkt[list].p,
                           (const uint2 *)ctx->ent[list].p, nb, (const uint32_t *)offs.p, d_entries);
    if (e == hipSuccess && !rc && d_starts) e = hipMemcpyAsync(d_starts, offs.p, (nb + 1) * 4, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    else (void)hipStreamSynchronize(ctx->stream);
    if (rc) return rc;
    if (e != hipSuccess) return rh_fail(ctx, REAL_HIP_E_DEVICE, "rows unpack", e);
    return REAL_HIP_OK;
}

// ---------------------------------------------------------------------------
// scratch of one index build: ONE allocation for all six lists.  (hipMalloc / hipFree of tens of gigabytes cost
// far more than the kernels of the build -- the driver clears and maps every page -- so the transients are laid
// out once, by offset, and regions whose lifetimes do not overlap share their bytes.)
//   pair B         keys_b (n x sig_bytes) + vals_b (n x 4): destination of the upload (host-built lists); one of the
//   pair A         keys_a + vals_x                          two buffers of the device sort (rocPRIM double buffers:
//                  the sort ping-pongs between the pairs and needs no temporary of its own beyond histograms)
//   entries        bucket rows only: the full entry array {key, pos} the rows are cut from lies over whichever pair
//                  does NOT hold the sorted list
//   tables         bucket starts; rows: overflow counts and their scan
// Persistent outputs (rows / overflow entries, or entries / bucket tables) are allocations of their own.
// ---------------------------------------------------------------------------
struct BuildScratch {
    ScopedBuf arena;
    uint8_t *keys_a = nullptr, *keys_b = nullptr;
    uint32_t *vals_x = nullptr, *vals_b = nullptr;
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    uint2 *ent = nullptr;
    uint32_t *bkt = nullptr, *ocnt = nullptr, *ostart = nullptr;
    void *scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    explicit BuildScratch(real_hip_ctx *c) : arena(c) {}
};

template <typename K>
static hipError_t sort_pairs(void *tmp, size_t &tmp_bytes, rocprim::double_buffer<K> &keys, rocprim::double_buffer<uint32_t> &vals, uint64_t n,
                             uint32_t bits, hipStream_t st)
{
    // stable LSD radix sort over the signature bits: equal signatures keep ascending position, as the reference's
    // ParallelRadixSort (ParallelRadixSort.hpp:160-203) does
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, vals, (size_t)n, 0u, bits, st);
}

static int plan_scratch(real_hip_ctx *ctx, BuildScratch &S, uint64_t n, unsigned sig_bytes, bool need_sort)
{
    const uint32_t l = ctx->prm.seedl;
    const size_t nn = n ? n : 1, nb1 = ((size_t)1 << ctx->pb) + 1;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t sort_tmp = 0, scan_tmp = 0;
    if (need_sort && n) {
        rocprim::double_buffer<uint32_t> v(nullptr, nullptr);
        hipError_t e;
        if (sig_bytes == 4) { rocprim::double_buffer<uint32_t> k(nullptr, nullptr); e = sort_pairs<uint32_t>(nullptr, sort_tmp, k, v, n, l, ctx->stream); }
        else { rocprim::double_buffer<uint64_t> k(nullptr, nullptr); e = sort_pairs<uint64_t>(nullptr, sort_tmp, k, v, n, l, ctx->stream); }
        if (e != hipSuccess) return rh_fail(ctx, REAL_HIP_E_DEVICE, "radix sort (size query)", e);
    }
    const bool rows = ctx->fine == 3;
    if (rows) {
        hipError_t e = rocprim::exclusive_scan(nullptr, scan_tmp, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, nb1, rocprim::plus<uint32_t>(), ctx->stream);
        if (e != hipSuccess) return rh_fail(ctx, REAL_HIP_E_DEVICE, "scan (size query)", e);
    }
    // (a pair = keys then values, back to back: the entry array of the rows lies over a whole pair, n x (sig_bytes + 4) >= n x 8)
    const size_t pair = al(nn * sig_bytes + nn * 4);
    const size_t o_keys_b = 0, o_vals_b = nn * sig_bytes, o_x = pair;
    size_t x = need_sort ? pair + al(sort_tmp ? sort_tmp : 8) : 0;
    if (rows && !need_sort) x = al(nn * sizeof(uint2)); // (host-built lists: no pair A, the entries get room of their own)
    const size_t o_bkt = o_x + x;
    const size_t o_ocnt = o_bkt + (ctx->fine ? al(nb1 * 4) : 0);       // (kind 0 keeps the bucket starts: an allocation of their own)
    const size_t o_ostart = o_ocnt + (rows ? al(nb1 * 4) : 0);
    const size_t o_scan = o_ostart + (rows ? al(nb1 * 4) : 0);
    const size_t total = o_scan + al(scan_tmp ? scan_tmp : 8);
    int rc = rh_reserve(ctx, S.arena, total);
    if (rc) return rc;
    uint8_t *base = (uint8_t *)S.arena.p;
    S.keys_b = base + o_keys_b; S.vals_b = (uint32_t *)(base + o_vals_b);
    S.keys_a = base + o_x; S.vals_x = (uint32_t *)(base + o_x + nn * sig_bytes); S.sort_tmp = base + o_x + pair; S.sort_tmp_bytes = sort_tmp;
    S.ent = rows ? (uint2 *)(base + o_x) : nullptr; // (the device build moves it to the pair that is free after the sort)
    S.bkt = ctx->fine ? (uint32_t *)(base + o_bkt) : nullptr;
    S.ocnt = rows ? (uint32_t *)(base + o_ocnt) : nullptr;
    S.ostart = rows ? (uint32_t *)(base + o_ostart) : nullptr;
    S.scan_tmp = base + o_scan; S.scan_tmp_bytes = scan_tmp;
    return REAL_HIP_OK;
}

// the device tables of list `list` from its sorted {sign, pos} arrays (in S.keys_b / S.vals_b or anywhere else on the device)
static int index_from_sorted(real_hip_ctx *ctx, BuildScratch &S, int list, const void *d_sign, const uint32_t *d_pos, uint64_t n,
                             unsigned sig_bytes)
{
    const uint32_t l = ctx->prm.seedl, pb = ctx->pb;
    uint32_t pshift, fshift, fbits, pbits;
    rh_index_geometry(l, pb, &pshift, &fshift, &fbits, &pbits);
    const uint32_t nb = 1u << pb;
    const bool rows = ctx->fine == 3;
    int rc;
    // The tables of a previous block stay allocated when they have about the size this block needs (the next block of
    // a genome, the next file of a directory: same layout, and hipMalloc of 200 GB costs seconds); otherwise they go
    // first, their bytes are needed.
    auto fit = [&](DevBuf &b, size_t need) -> int {
        if (b.cap >= need && b.cap <= 2 * need + ((size_t)1 << 20)) return REAL_HIP_OK;
        rh_release(ctx, b);
        return rh_reserve(ctx, b, need);
    };
    if (!n) {
        const size_t esz = rows ? 128 : (ctx->fine ? 16 : 4);
        if ((rc = fit(ctx->bkt[list], ((size_t)nb + 1) * esz))) return rc;
        if ((rc = fit(ctx->ent[list], sizeof(uint2)))) return rc;
        RH_HIP(ctx, hipMemsetAsync(ctx->bkt[list].p, 0, ((size_t)nb + 1) * esz, ctx->stream));
        return REAL_HIP_OK;
    }
    uint2 *d_ent = S.ent;
    if (!rows) {
        if ((rc = fit(ctx->ent[list], n * sizeof(uint2)))) return rc;
        d_ent = (uint2 *)ctx->ent[list].p;
    }
    uint32_t *d_bkt = S.bkt;
    if (!ctx->fine) {
        if ((rc = fit(ctx->bkt[list], ((size_t)nb + 1) * 4))) return rc;
        d_bkt = (uint32_t *)ctx->bkt[list].p;
    }
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    const uint64_t *T = (const uint64_t *)ctx->text.p;
    rh_time_begin(ctx, ctx->stream, REAL_HIP_K_INDEX);
    if (sig_bytes == 4)
        hipLaunchKernelGGL(entries_kernel<uint32_t>, grid, block, 0, ctx->stream, (const uint32_t *)d_sign, d_pos, n, l, list, T,
                           pshift, fshift, fbits, pbits, nb, d_ent, d_bkt);
    else
        hipLaunchKernelGGL(entries_kernel<uint64_t>, grid, block, 0, ctx->stream, (const uint64_t *)d_sign, d_pos, n, l, list, T,
                           pshift, fshift, fbits, pbits, nb, d_ent, d_bkt);
    RH_HIP(ctx, hipGetLastError());
    if (rows) {
        // rows: overflow sizes, scan, fill
        const dim3 g1((unsigned)(((uint64_t)nb + 1 + 255) / 256)), b1(256);
        hipLaunchKernelGGL(rows_overflow_kernel, g1, b1, 0, ctx->stream, (const uint32_t *)d_bkt, (const uint2 *)d_ent, (uint64_t)nb, pbits, fbits, S.ocnt);
        size_t tmp = S.scan_tmp_bytes;
        RH_HIP(ctx, rocprim::exclusive_scan(S.scan_tmp, tmp, S.ocnt, S.ostart, 0u, (size_t)nb + 1, rocprim::plus<uint32_t>(), ctx->stream));
        rh_time_end(ctx, ctx->stream);
        uint32_t n_ovf = 0;
        RH_HIP(ctx, hipMemcpyAsync(&n_ovf, S.ostart + nb, 4, hipMemcpyDeviceToHost, ctx->stream));
        RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if ((rc = fit(ctx->bkt[list], (size_t)nb * 128))) return rc;
        if ((rc = fit(ctx->ent[list], ((size_t)n_ovf + 1) * sizeof(uint2)))) return rc;
        rh_time_begin(ctx, ctx->stream, REAL_HIP_K_INDEX);
        hipLaunchKernelGGL(rows_fill_kernel, dim3((unsigned)(((uint64_t)nb + 255) / 256)), b1, 0, ctx->stream, (const uint32_t *)d_bkt,
                           (const uint2 *)d_ent, (uint64_t)nb, pbits, fbits, (const uint32_t *)S.ostart, (uint4 *)ctx->bkt[list].p, (uint2 *)ctx->ent[list].p);
        rh_time_end(ctx, ctx->stream);
        RH_HIP(ctx, hipGetLastError());
        return REAL_HIP_OK;
    }
    if (ctx->fine) {
        if ((rc = fit(ctx->bkt[list], ((size_t)nb + 1) * sizeof(uint4)))) return rc;
        if (ctx->fine == 1)
            hipLaunchKernelGGL(fine_table_kernel, dim3((unsigned)(((uint64_t)nb + 1 + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const uint32_t *)d_bkt, (const uint2 *)d_ent, (uint64_t)nb, pbits, fbits, (uint4 *)ctx->bkt[list].p);
        else
            hipLaunchKernelGGL(fp_table_kernel, dim3((unsigned)(((uint64_t)nb + 1 + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const uint32_t *)d_bkt, (const uint2 *)d_ent, (uint64_t)nb, (uint4 *)ctx->bkt[list].p);
        RH_HIP(ctx, hipGetLastError());
    }
    rh_time_end(ctx, ctx->stream);
    return REAL_HIP_OK;
}

template <typename K>
static int sort_list(real_hip_ctx *ctx, BuildScratch &S, int list, const uint32_t *d_wpos, uint64_t first_window, uint64_t n, bool uploaded = false);

// host-built form (real_hip_set_index_block): the six sorted lists are uploaded one after the other through the scratch
int rh_index_from_host_lists(real_hip_ctx *ctx, uint64_t n, const void *const sign[6], const uint32_t *const pos[6], unsigned sig_bytes)
{
    const double t0 = rh_now_ms();
    BuildScratch S(ctx);
    const bool rows = ctx->fine == 3; // bucket rows: the lists are sorted once more, by the mixed signature (sort_list)
    int rc = plan_scratch(ctx, S, n, sig_bytes, rows);
    if (rc) return rc;
    for (int k = 0; k < 6; ++k) {
        if (n) {
            RH_HIP(ctx, hipMemcpyAsync(S.keys_b, sign[k], n * sig_bytes, hipMemcpyHostToDevice, ctx->stream));
            RH_HIP(ctx, hipMemcpyAsync(S.vals_b, pos[k], n * 4, hipMemcpyHostToDevice, ctx->stream));
        }
        if (rows) rc = sig_bytes == 4 ? sort_list<uint32_t>(ctx, S, k, nullptr, 0, n, true) : sort_list<uint64_t>(ctx, S, k, nullptr, 0, n, true);
        else rc = index_from_sorted(ctx, S, k, S.keys_b, S.vals_b, n, sig_bytes);
        if (rc) return rc;
    }
    RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rh_time_resolve(ctx);
    ctx->build_wall_ms += rh_now_ms() - t0;
    return REAL_HIP_OK;
}

// ---------------------------------------------------------------------------
// device index build
// ---------------------------------------------------------------------------
// window [i,i+l) is emitted iff it holds no N (MapTextFile::readNextSignature /
// readFullSignature, MapTextFile.hpp:118-179); fragment boundaries do not cut windows.
__global__ void window_flags_kernel(const uint64_t *__restrict__ wild, uint64_t nwin, uint32_t l, uint8_t *__restrict__ flags)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwin) return;
    uint64_t a = i, e = i + l - 1, wa = a >> 6, we = e >> 6;
    bool ok = true;
    for (uint64_t w = wa; w <= we; ++w) {
        uint64_t m = ~0ull;
        if (w == wa) m &= ~0ull >> (a & 63);
        if (w == we) m &= ~0ull << (63 - (e & 63));
        if (wild[w] & m) ok = false;
    }
    flags[i] = ok ? 1 : 0;
}

__global__ void iota_kernel(uint32_t *out, uint64_t first, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)(first + i);
}

// list k signature of the window at wpos[j]; mix: the sort key of the bucket rows (rh_mix32 / rh_mix64)
template <typename K>
__device__ __forceinline__ K mixed_key(K v, uint32_t l) { return sizeof(K) == 4 ? (K)rh_mix32((uint32_t)v, l) : (K)rh_mix64((uint64_t)v, l); }
template <typename K>
__global__ void keys_kernel(const uint64_t *__restrict__ T, const uint32_t *__restrict__ wpos, uint64_t n, uint32_t l,
                            int list, bool mix, K *__restrict__ keys)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const K v = (K)window_signature(T, wpos[j], l, list);
    keys[j] = mix ? mixed_key<K>(v, l) : v;
}
template <typename K>
__global__ void mix_keys_kernel(K *__restrict__ keys, uint64_t n, uint32_t l)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) keys[j] = mixed_key<K>(keys[j], l);
}

// d_wpos: the window starts of the block in ascending order, or null = every window from first_window on (no N in the text)
// uploaded: the list is in (S.keys_b, S.vals_b) already, sorted by signature (host-built lists for bucket rows): only
// mixed and sorted again
template <typename K>
static int sort_list(real_hip_ctx *ctx, BuildScratch &S, int list, const uint32_t *d_wpos, uint64_t first_window, uint64_t n, bool uploaded)
{
    const uint32_t l = ctx->prm.seedl;
    const void *d_sign = S.keys_b;
    const uint32_t *d_pos = S.vals_b;
    const bool mix = ctx->fine == 3;
    if (n) {
        rh_time_begin(ctx, ctx->stream, REAL_HIP_K_INDEX);
        const dim3 grid((unsigned)((n + 255) / 256)), block(256);
        rocprim::double_buffer<K> keys((K *)S.keys_a, (K *)S.keys_b);
        rocprim::double_buffer<uint32_t> vals(S.vals_x, S.vals_b);
        if (uploaded) {
            hipLaunchKernelGGL(mix_keys_kernel<K>, grid, block, 0, ctx->stream, (K *)S.keys_b, n, l);
            keys = rocprim::double_buffer<K>((K *)S.keys_b, (K *)S.keys_a);
            vals = rocprim::double_buffer<uint32_t>(S.vals_b, S.vals_x);
        } else {
            // the values of the sort: a fresh copy of the window starts for every list (the sort clobbers both buffers)
            if (d_wpos) RH_HIP(ctx, hipMemcpyAsync(S.vals_x, d_wpos, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
            else hipLaunchKernelGGL(iota_kernel, grid, block, 0, ctx->stream, S.vals_x, first_window, n);
            hipLaunchKernelGGL(keys_kernel<K>, grid, block, 0, ctx->stream, (const uint64_t *)ctx->text.p, (const uint32_t *)S.vals_x, n, l, list, mix,
                               (K *)S.keys_a);
        }
        size_t tmp = S.sort_tmp_bytes;
        RH_HIP(ctx, sort_pairs<K>(S.sort_tmp, tmp, keys, vals, n, l, ctx->stream));
        rh_time_end(ctx, ctx->stream);
        d_sign = keys.current(); d_pos = vals.current();
        // the entry array of the rows goes over the pair the sorted list is NOT in
        if (S.ent) S.ent = (uint2 *)((const void *)keys.current() == (const void *)S.keys_a ? S.keys_b : S.keys_a);
    }
    return index_from_sorted(ctx, S, list, d_sign, d_pos, n, sizeof(K));
}

int rh_index_build_device(real_hip_ctx *ctx, uint64_t first_window, uint64_t max_entries, uint64_t *n_entries,
                          int *have_next)
{
    const double t0 = rh_now_ms();
    const uint32_t l = ctx->prm.seedl;
    const uint64_t n = ctx->n_bases;
    const uint64_t nwin_all = (n >= l) ? (n - l + 1) : 0;
    uint64_t total_valid = nwin_all;
    const uint32_t *d_wpos = nullptr;
    uint64_t cnt = 0;
    int rc;
    if (ctx->n_wild == 0) { // every window from first_window on: the starts are generated, not stored
        cnt = (first_window < nwin_all) ? (nwin_all - first_window) : 0;
        if (cnt > max_entries) cnt = max_entries;
        d_wpos = nullptr;
    } else {
        // flags -> compacted ascending window starts (all blocks), then slice
        ScopedBuf flags(ctx), sel(ctx), dcount(ctx);
        if ((rc = rh_reserve(ctx, flags, nwin_all ? nwin_all : 1))) return rc;
        if ((rc = rh_reserve(ctx, ctx->vals_a, (nwin_all ? nwin_all : 1) * 4))) return rc;
        if ((rc = rh_reserve(ctx, dcount, 8))) return rc;
        RH_HIP(ctx, hipMemsetAsync(dcount.p, 0, 8, ctx->stream));
        if (nwin_all) {
            size_t tmp = 0;
            RH_HIP(ctx, rocprim::select(nullptr, tmp, rocprim::counting_iterator<uint32_t>(0), (uint8_t *)flags.p,
                                        (uint32_t *)ctx->vals_a.p, (size_t *)dcount.p, (size_t)nwin_all, ctx->stream));
            if ((rc = rh_reserve(ctx, sel, tmp ? tmp : 8))) return rc;
            rh_time_begin(ctx, ctx->stream, REAL_HIP_K_INDEX);
            hipLaunchKernelGGL(window_flags_kernel, dim3((unsigned)((nwin_all + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const uint64_t *)ctx->wild.p, nwin_all, l, (uint8_t *)flags.p);
            RH_HIP(ctx, rocprim::select(sel.p, tmp, rocprim::counting_iterator<uint32_t>(0), (uint8_t *)flags.p,
                                        (uint32_t *)ctx->vals_a.p, (size_t *)dcount.p, (size_t)nwin_all, ctx->stream));
            rh_time_end(ctx, ctx->stream);
        }
        size_t hcount = 0;
        RH_HIP(ctx, hipMemcpyAsync(&hcount, dcount.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        total_valid = hcount;
        cnt = (first_window < total_valid) ? (total_valid - first_window) : 0;
        if (cnt > max_entries) cnt = max_entries;
        d_wpos = (const uint32_t *)ctx->vals_a.p + first_window;
    }
    ctx->n_entries = cnt;
    rh_choose_tables(ctx, cnt);
    for (int attempt = 0;; ++attempt) {
        BuildScratch S(ctx);
        rc = plan_scratch(ctx, S, cnt, l <= 32 ? 4 : 8, true);
        for (int k = 0; k < 6 && !rc; ++k)
            rc = (l <= 32) ? sort_list<uint32_t>(ctx, S, k, d_wpos, first_window, cnt) : sort_list<uint64_t>(ctx, S, k, d_wpos, first_window, cnt);
        if (!rc) RH_HIP(ctx, hipStreamSynchronize(ctx->stream)); // (before the scratch goes)
        if (rc == REAL_HIP_E_NOMEM && ctx->fine == 3 && ctx->prm.table_kind == 0 && !ctx->no_rows && attempt == 0) {
            // the rows did not fit after all (memory held by others): once more with directory tables
            (void)hipStreamSynchronize(ctx->stream);
            for (int j = 0; j < 6; ++j) { rh_release(ctx, ctx->ent[j]); rh_release(ctx, ctx->bkt[j]); }
            ctx->no_rows = true;
            rh_choose_tables(ctx, cnt);
            continue;
        }
        if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
        break;
    }
    rh_release(ctx, ctx->vals_a); // 4 bytes per window: give it back
    rh_time_resolve(ctx);
    ctx->have_index = true;
    if (n_entries) *n_entries = cnt;
    if (have_next) *have_next = (first_window + cnt < total_valid) ? 1 : 0;
    ctx->build_wall_ms += rh_now_ms() - t0;
    return REAL_HIP_OK;
}

// ---------------------------------------------------------------------------
// matchAll post-pass: order the raw hit records per read as unifyMatches does
// (operator<, matchAllImplementation.cpp:122-136: k, pos, file, frag, score, inverted;
// file is constant inside a call and frag is a monotone function of pos).
// The matcher appends hits in wave order and counts them per read; a read has a handful of
// hits, so: offsets = exclusive scan of the counts, scatter every raw record into its read's
// segment, rank the records of a segment against each other (one thread per read; one
// workgroup per read with more than ALL_SMALL hits).  No global sort.  A read with one hit -- most
// of those that have any -- needs neither: its record goes straight from the raw list to its place.
// ---------------------------------------------------------------------------
#define ALL_SMALL 32u

struct U32ToU64 {
    __host__ __device__ uint64_t operator()(uint32_t v) const { return (uint64_t)v; }
};

__device__ __forceinline__ bool hit_less(const uint4 &x, const uint4 &y)
{
    const uint32_t kx = x.w & 0xffu, ky = y.w & 0xffu;
    if (kx != ky) return kx < ky;
    if (x.y != y.y) return x.y < y.y;
    const float sx = __uint_as_float(x.z), sy = __uint_as_float(y.z);
    if (sx != sy) return sx < sy;
    return ((x.w >> 8) & 1u) < ((y.w >> 8) & 1u);
}
__device__ __forceinline__ real_hip_hit hit_record(const uint4 &h)
{
    real_hip_hit o;
    o.read = h.x; o.pos = h.y; o.score = __uint_as_float(h.z);
    o.frag = (uint16_t)(h.w >> 16); o.k = (uint8_t)(h.w & 0xff); o.inverted = (uint8_t)((h.w >> 8) & 1);
    return o;
}

// most reads that have a hit have exactly one: its record goes straight to its place.  The hits of the others are
// collected in their segment (cursor[] counts them; it is all zero between two calls) and the read is listed once.
__global__ void all_scatter_kernel(const uint4 *__restrict__ raw, uint64_t n, const uint64_t *__restrict__ off, const uint32_t *__restrict__ cnt,
                                   uint32_t *__restrict__ cursor, uint4 *__restrict__ seg, real_hip_hit *__restrict__ out,
                                   uint32_t *__restrict__ multi_list, unsigned long long *multi_count)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 h = raw[i];
    const uint64_t o = off[h.x];
    if (cnt[h.x] == 1u) { out[o] = hit_record(h); return; }
    const uint32_t slot = atomicAdd(&cursor[h.x], 1u);
    seg[o + slot] = h;
    if (slot == 0) multi_list[atomicAdd(multi_count, 1ull)] = h.x;
}

__global__ void all_rank_kernel(const uint4 *__restrict__ seg, const uint64_t *__restrict__ off, const uint32_t *__restrict__ multi_list,
                                const unsigned long long *__restrict__ multi_count, uint32_t *__restrict__ cursor,
                                real_hip_hit *__restrict__ out, uint32_t *__restrict__ big_list, unsigned long long *big_count)
{
    const uint64_t nm = *multi_count;
    for (uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; m < nm; m += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = multi_list[m];
        cursor[r] = 0; // (for the next call)
        const uint64_t o = off[r];
        const uint32_t c = (uint32_t)(off[r + 1] - o);
        if (c > ALL_SMALL) { big_list[atomicAdd(big_count, 1ull)] = (uint32_t)r; continue; }
        for (uint32_t i = 0; i < c; ++i) {
            const uint4 h = seg[o + i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < c; ++j) {
                const uint4 g = seg[o + j];
                rank += (hit_less(g, h) || (j < i && !hit_less(h, g))) ? 1u : 0u;
            }
            out[o + rank] = hit_record(h);
        }
    }
}

// reads with many hits (repeats): one workgroup per read.  Up to ALL_HUGE hits the same quadratic ranking; beyond
// (a read on a low-complexity locus has 10^5 hits) a stable LSD radix sort of the segment by (k, pos) -- ten passes of
// four bits between the segment and the same range of the raw array, which is free once the records are scattered --
// and a look at the neighbours for the one tie (k, pos) leaves open: the same position on both strands.
#define ALL_HUGE 1024u
__device__ __forceinline__ uint64_t hit_key40(const uint4 &h) { return ((uint64_t)(h.w & 0xffu) << 32) | (uint64_t)h.y; }

__global__ __launch_bounds__(256) void all_rank_big_kernel(uint4 *__restrict__ seg, uint4 *__restrict__ scratch, const uint64_t *__restrict__ off,
                                                           const uint32_t *__restrict__ big_list, const unsigned long long *__restrict__ big_count,
                                                           real_hip_hit *__restrict__ out)
{
    __shared__ uint32_t hist[16 * 256];
    __shared__ uint32_t tot[16];
    const uint32_t t = threadIdx.x;
    const uint64_t nb = *big_count;
    for (uint64_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const uint64_t r = big_list[b], o = off[r];
        const uint32_t c = (uint32_t)(off[r + 1] - o);
        if (c <= ALL_HUGE) {
            for (uint32_t i = t; i < c; i += blockDim.x) {
                const uint4 h = seg[o + i];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < c; ++j) {
                    const uint4 g = seg[o + j];
                    rank += (hit_less(g, h) || (j < i && !hit_less(h, g))) ? 1u : 0u;
                }
                out[o + rank] = hit_record(h);
            }
            continue;
        }
        uint4 *A = seg + o, *B = scratch + o;
        const uint32_t chunk = (c + 255u) / 256u, lo = min(c, t * chunk), hi = min(c, lo + chunk);
        for (uint32_t pass = 0; pass < 10; ++pass) { // (every thread of the workgroup makes the same trips)
            const uint32_t sh = 4 * pass;
            uint32_t cnt[16];
#pragma unroll
            for (int d = 0; d < 16; ++d) cnt[d] = 0;
            for (uint32_t i = lo; i < hi; ++i) {
                const uint32_t dg = (uint32_t)(hit_key40(A[i]) >> sh) & 15u;
#pragma unroll
                for (int d = 0; d < 16; ++d) cnt[d] += (dg == (uint32_t)d) ? 1u : 0u;
            }
#pragma unroll
            for (int d = 0; d < 16; ++d) hist[d * 256 + t] = cnt[d];
            __syncthreads();
            if (t < 16) { // exclusive scan of digit t's counts over the threads (= over the chunks, in order: stable)
                uint32_t run = 0;
                for (uint32_t j = 0; j < 256; ++j) { const uint32_t v = hist[t * 256 + j]; hist[t * 256 + j] = run; run += v; }
                tot[t] = run;
            }
            __syncthreads();
            if (t == 0) {
                uint32_t run = 0;
                for (int d = 0; d < 16; ++d) { const uint32_t v = tot[d]; tot[d] = run; run += v; }
            }
            __syncthreads();
#pragma unroll
            for (int d = 0; d < 16; ++d) cnt[d] = tot[d] + hist[d * 256 + t];
            for (uint32_t i = lo; i < hi; ++i) {
                const uint4 h = A[i];
                const uint32_t dg = (uint32_t)(hit_key40(h) >> sh) & 15u;
                uint32_t dst = 0;
#pragma unroll
                for (int d = 0; d < 16; ++d) if (dg == (uint32_t)d) dst = cnt[d]++;
                B[dst] = h;
            }
            __threadfence_block();
            __syncthreads();
            uint4 *x = A; A = B; B = x;
        }
        // (ten passes: the sorted records are in the segment again)
        for (uint32_t i = t; i < c; i += blockDim.x) {
            const uint4 h = A[i];
            uint32_t rank = i;
            if (i + 1 < c) { const uint4 g = A[i + 1]; if (hit_key40(g) == hit_key40(h) && hit_less(g, h)) rank = i + 1; }
            if (i > 0) { const uint4 g = A[i - 1]; if (hit_key40(g) == hit_key40(h) && hit_less(h, g)) rank = i - 1; }
            out[o + rank] = hit_record(h);
        }
        __syncthreads();
    }
}

int rh_all_finish(real_hip_ctx *ctx, uint64_t n_raw, uint64_t n_reads, real_hip_hit *d_out, uint64_t *d_hit_offsets)
{
    RhTimer tm(ctx, REAL_HIP_K_ALL_SORT);
    int rc;
    if (!n_reads) {
        if (d_hit_offsets) RH_HIP(ctx, hipMemsetAsync(d_hit_offsets, 0, 8, ctx->stream));
        return REAL_HIP_OK;
    }
    uint64_t *off = d_hit_offsets;
    if (!off) {
        if ((rc = rh_reserve(ctx, ctx->hit_off, (n_reads + 1) * 8))) return rc;
        off = (uint64_t *)ctx->hit_off.p;
    }
    uint32_t *cnt = (uint32_t *)ctx->hit_cnt.p; // written by the matcher for every read of the batch
    RH_HIP(ctx, hipMemsetAsync(cnt + n_reads, 0, 4, ctx->stream));
    {
        rocprim::transform_iterator<const uint32_t *, U32ToU64, uint64_t> in((const uint32_t *)cnt, U32ToU64());
        size_t tmp = 0;
        RH_HIP(ctx, rocprim::exclusive_scan(nullptr, tmp, in, off, (uint64_t)0, (size_t)(n_reads + 1), rocprim::plus<uint64_t>(), ctx->stream));
        if ((rc = rh_reserve(ctx, ctx->sort_tmp, tmp ? tmp : 8))) return rc;
        RH_HIP(ctx, rocprim::exclusive_scan(ctx->sort_tmp.p, tmp, in, off, (uint64_t)0, (size_t)(n_reads + 1), rocprim::plus<uint64_t>(), ctx->stream));
    }
    if (n_raw) {
        if ((rc = rh_reserve(ctx, ctx->keys_a, n_raw * sizeof(uint4)))) return rc;       // the per-read segments
        if (n_reads * 4 > ctx->all_cursor.cap) {                                          // scatter cursors: zero between two calls
            if ((rc = rh_reserve(ctx, ctx->all_cursor, n_reads * 4))) return rc;
            RH_HIP(ctx, hipMemsetAsync(ctx->all_cursor.p, 0, ctx->all_cursor.cap, ctx->stream));
        }
        if ((rc = rh_reserve(ctx, ctx->big_list, n_reads * 8 + 16))) return rc;          // reads with more than one hit; with more than ALL_SMALL
        uint32_t *multi_list = (uint32_t *)ctx->big_list.p, *big_list = multi_list + n_reads;
        unsigned long long *counts = (unsigned long long *)((uint8_t *)ctx->big_list.p + n_reads * 8);
        RH_HIP(ctx, hipMemsetAsync(counts, 0, 16, ctx->stream));
        const uint4 *raw = (const uint4 *)ctx->raw.p;
        uint4 *seg = (uint4 *)ctx->keys_a.p;
        hipLaunchKernelGGL(all_scatter_kernel, dim3((unsigned)((n_raw + 255) / 256)), dim3(256), 0, ctx->stream, raw, n_raw,
                           (const uint64_t *)off, (const uint32_t *)cnt, (uint32_t *)ctx->all_cursor.p, seg, d_out, multi_list, counts);
        hipLaunchKernelGGL(all_rank_kernel, dim3(2048), dim3(256), 0, ctx->stream, (const uint4 *)seg, (const uint64_t *)off,
                           (const uint32_t *)multi_list, (const unsigned long long *)counts, (uint32_t *)ctx->all_cursor.p, d_out, big_list, counts + 1);
        hipLaunchKernelGGL(all_rank_big_kernel, dim3(1024), dim3(256), 0, ctx->stream, seg, (uint4 *)ctx->raw.p, (const uint64_t *)off,
                           (const uint32_t *)big_list, (const unsigned long long *)(counts + 1), d_out);
    }
    RH_HIP(ctx, hipGetLastError());
    return REAL_HIP_OK;
}

// ---------------------------------------------------------------------------
// export of a device list in the reference's form {signature, position} (tests, CPU baseline)
// ---------------------------------------------------------------------------
template <typename K>
__global__ void export_kernel(const uint2 *__restrict__ ent, const uint64_t *__restrict__ T, uint64_t n, uint32_t l, int list,
                              K *__restrict__ sign, uint32_t *__restrict__ pos)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t p = ent[j].y;
    sign[j] = (K)window_signature(T, p, l, list);
    pos[j] = p;
}

int rh_index_export(real_hip_ctx *ctx, int list, void *h_sign, uint32_t *h_pos)
{
    const uint64_t n = ctx->n_entries;
    if (!n) return REAL_HIP_OK;
    const uint32_t l = ctx->prm.seedl;
    const unsigned sb = l <= 32 ? 4 : 8;
    int rc;
    if ((rc = rh_reserve(ctx, ctx->keys_a, n * sb))) return rc;
    if ((rc = rh_reserve(ctx, ctx->vals_a, n * 4))) return rc;
    ScopedBuf unpacked(ctx); // bucket rows: the entries in list order first
    const uint2 *d_ent = (const uint2 *)ctx->ent[list].p;
    const bool rows = ctx->fine == 3;
    size_t sort_tmp = 0;
    if (rows) {
        // ... and that order is the one of the mixed signatures (real_hip_internal.h: rh_mix32): signatures and positions are
        // sorted back into the reference's list order (stable: equal signatures are together and in ascending position
        // already).  The unpacked entries are dead once the signatures are computed: their room is the sort's other buffer.
        rocprim::double_buffer<uint32_t> v(nullptr, nullptr);
        hipError_t e;
        if (sb == 4) { rocprim::double_buffer<uint32_t> k(nullptr, nullptr); e = sort_pairs<uint32_t>(nullptr, sort_tmp, k, v, n, l, ctx->stream); }
        else { rocprim::double_buffer<uint64_t> k(nullptr, nullptr); e = sort_pairs<uint64_t>(nullptr, sort_tmp, k, v, n, l, ctx->stream); }
        if (e != hipSuccess) return rh_fail(ctx, REAL_HIP_E_DEVICE, "radix sort (size query)", e);
        const size_t room = n * (sb + 4) > n * sizeof(uint2) ? n * (sb + 4) : n * sizeof(uint2);
        if ((rc = rh_reserve(ctx, unpacked, room))) return rc;
        if ((rc = rh_reserve(ctx, ctx->sort_tmp, sort_tmp ? sort_tmp : 8))) return rc;
        if ((rc = rh_rows_unpack(ctx, list, (uint2 *)unpacked.p, nullptr))) return rc; // (uses ctx->sort_tmp for its scan: before the sort)
        d_ent = (const uint2 *)unpacked.p;
    }
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (sb == 4)
        hipLaunchKernelGGL(export_kernel<uint32_t>, grid, block, 0, ctx->stream, d_ent,
                           (const uint64_t *)ctx->text.p, n, l, list, (uint32_t *)ctx->keys_a.p, (uint32_t *)ctx->vals_a.p);
    else
        hipLaunchKernelGGL(export_kernel<uint64_t>, grid, block, 0, ctx->stream, d_ent,
                           (const uint64_t *)ctx->text.p, n, l, list, (uint64_t *)ctx->keys_a.p, (uint32_t *)ctx->vals_a.p);
    RH_HIP(ctx, hipGetLastError());
    const void *d_sign = ctx->keys_a.p;
    const uint32_t *d_pos = (const uint32_t *)ctx->vals_a.p;
    if (rows) {
        if ((rc = rh_reserve(ctx, ctx->sort_tmp, sort_tmp ? sort_tmp : 8))) return rc; // (rh_rows_unpack may have left a smaller one)
        uint8_t *alt = (uint8_t *)unpacked.p;
        rocprim::double_buffer<uint32_t> vals((uint32_t *)ctx->vals_a.p, (uint32_t *)(alt + n * sb));
        size_t tmp = sort_tmp;
        if (sb == 4) {
            rocprim::double_buffer<uint32_t> keys((uint32_t *)ctx->keys_a.p, (uint32_t *)alt);
            RH_HIP(ctx, sort_pairs<uint32_t>(ctx->sort_tmp.p, tmp, keys, vals, n, l, ctx->stream));
            d_sign = keys.current();
        } else {
            rocprim::double_buffer<uint64_t> keys((uint64_t *)ctx->keys_a.p, (uint64_t *)alt);
            RH_HIP(ctx, sort_pairs<uint64_t>(ctx->sort_tmp.p, tmp, keys, vals, n, l, ctx->stream));
            d_sign = keys.current();
        }
        d_pos = vals.current();
    }
    if (h_sign) RH_HIP(ctx, hipMemcpyAsync(h_sign, d_sign, n * sb, hipMemcpyDeviceToHost, ctx->stream));
    if (h_pos) RH_HIP(ctx, hipMemcpyAsync(h_pos, d_pos, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return REAL_HIP_OK;
}
