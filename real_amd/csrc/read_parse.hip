// read_parse.hip -- read ingestion on the device: FASTA / FASTQ text -> the arrays a
// real_hip_batch takes.  Restates, for text in canonical form (every record field on one line),
//   FastQReader::getNextPatternUnlocked   FastQReader.hpp:130-180   '@'id \n seq \n '+'... \n quality \n
//   FastAReader::getNextPatternUnlocked   FastAReader.hpp:107-138   '>'id \n seq \n
//   Pattern::computeMapped / mapChar      Pattern.hpp:105-128, acgtnMap.hpp:39-50   A,C,G,T -> 0..3, anything else -> 4
//   quality = character - offset          FastQReader.hpp:165-173
// The reference's readers are character-level state machines that also accept wrapped sequences and stray
// white space; text that is not canonical is refused here (REAL_HIP_E_UNSUPPORTED) and stays with the host
// reader (real_amd/host/ReadReader.cpp), which implements the general form.
#include "real_hip_internal.h"

#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

struct IsNewline {
    const char *text;
    __host__ __device__ bool operator()(uint32_t i) const { return text[i] == '\n'; }
};
struct LenToU64 {
    __host__ __device__ uint64_t operator()(uint32_t v) const { return (uint64_t)v; }
};

// ---- newline positions: count per 16 KiB block, scan, fill (the text is read twice with 16-byte loads) ----
#define NL_BLOCK_BYTES (256u * 16u * 4u)

// bit 7 of every byte of x that equals '\n'
__device__ __forceinline__ uint32_t nl_mask(uint32_t x)
{
    const uint32_t t = x ^ 0x0a0a0a0au;
    return ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t | 0x7f7f7f7fu);
}
// the 16 bytes at text + o (o a multiple of 16, text 16-byte aligned); bytes at or behind n_bytes read as 0
__device__ __forceinline__ uint4 text16(const char *__restrict__ text, uint64_t o, uint64_t n_bytes)
{
    if (o + 16 <= n_bytes) return *reinterpret_cast<const uint4 *>(text + o);
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint64_t b = o; b < n_bytes; ++b) w[(b - o) >> 2] |= (uint32_t)(uint8_t)text[b] << (8 * ((b - o) & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// bit 7 of every byte of x that equals '\r'
__device__ __forceinline__ uint32_t cr_mask(uint32_t x)
{
    const uint32_t t = x ^ 0x0d0d0d0du;
    return ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t | 0x7f7f7f7fu);
}

__global__ __launch_bounds__(256) void nl_count_kernel(const char *__restrict__ text, uint64_t n_bytes, uint32_t *__restrict__ block_count,
                                                       unsigned int *__restrict__ any_cr)
{
    __shared__ uint32_t part[4];
    uint32_t c = 0, cr = 0;
    const uint64_t base = (uint64_t)blockIdx.x * NL_BLOCK_BYTES;
    for (int it = 0; it < 4; ++it) {
        const uint64_t o = base + (uint64_t)it * 4096 + threadIdx.x * 16;
        if (o < n_bytes) {
            const uint4 v = text16(text, o, n_bytes);
            c += __popc(nl_mask(v.x)) + __popc(nl_mask(v.y)) + __popc(nl_mask(v.z)) + __popc(nl_mask(v.w));
            cr |= cr_mask(v.x) | cr_mask(v.y) | cr_mask(v.z) | cr_mask(v.w);
        }
    }
    if (__any(cr != 0) && (threadIdx.x & 63) == 0) atomicOr(any_cr, 1u);
    for (int d = 32; d; d >>= 1) c += __shfl_xor((int)c, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(256) void nl_fill_kernel(const char *__restrict__ text, uint64_t n_bytes, const uint32_t *__restrict__ block_off,
                                                      uint32_t *__restrict__ nl)
{
    __shared__ uint32_t wsum[4];
    const uint64_t base = (uint64_t)blockIdx.x * NL_BLOCK_BYTES;
    uint32_t run = block_off[blockIdx.x];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int it = 0; it < 4; ++it) {
        const uint64_t o = base + (uint64_t)it * 4096 + threadIdx.x * 16;
        uint32_t mk[4] = {0, 0, 0, 0};
        if (o < n_bytes) {
            const uint4 v = text16(text, o, n_bytes);
            mk[0] = nl_mask(v.x); mk[1] = nl_mask(v.y); mk[2] = nl_mask(v.z); mk[3] = nl_mask(v.w);
        }
        const uint32_t c = __popc(mk[0]) + __popc(mk[1]) + __popc(mk[2]) + __popc(mk[3]);
        // exclusive prefix of c over the 256 threads (= over the 4 KiB in byte order)
        uint32_t incl = c;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up((int)incl, d); if (lane >= (uint32_t)d) incl += t; }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (int w = 0; w < 4; ++w) { if ((uint32_t)w < wv) before += wsum[w]; total += wsum[w]; }
        uint32_t at = run + before + incl - c;
        for (int q = 0; q < 4; ++q) {
            uint32_t mq = mk[q];
            while (mq) {
                const int bit = __ffs((int)mq) - 1; // bit 7 of byte bit/8
                nl[at++] = (uint32_t)(o + 4 * q + (bit >> 3));
                mq &= mq - 1;
            }
        }
        run += total;
        __syncthreads();
    }
}

// start and length of line j of the chunk (without its '\n' and a '\r' in front of it); line m (after the last
// newline) exists when the chunk does not end in a newline
__device__ __forceinline__ void line_span(const char *__restrict__ text, const uint32_t *__restrict__ nl, uint64_t m, uint64_t n_bytes,
                                          bool has_cr, uint64_t j, uint32_t &start, uint32_t &len)
{
    const uint32_t s = j ? nl[j - 1] + 1 : 0u;
    uint32_t e = j < m ? nl[j] : (uint32_t)n_bytes;
    if (has_cr && e > s && text[e - 1] == '\r') e--; // (no '\r' anywhere in the chunk: nothing to look at)
    start = s; len = e - s;
}

// one thread per record: field spans, form check, length
__global__ void record_spans_kernel(const char *__restrict__ text, const uint32_t *__restrict__ nl, uint64_t m, uint64_t n_bytes,
                                    uint64_t n_rec, int fastq, const unsigned int *__restrict__ any_cr, uint32_t *__restrict__ seq_start, uint32_t *__restrict__ seq_len,
                                    uint32_t *__restrict__ qual_start, uint32_t *__restrict__ id_start, uint32_t *__restrict__ id_len,
                                    unsigned int *__restrict__ bad, unsigned int *__restrict__ max_len)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t mylen = 0;
    const bool has_cr = *any_cr != 0;
    if (r < n_rec) {
        const uint64_t l0 = r * (fastq ? 4 : 2);
        uint32_t s, l;
        line_span(text, nl, m, n_bytes, has_cr, l0, s, l);
        if (l == 0 || text[s] != (fastq ? '@' : '>')) atomicOr(bad, 1u);
        id_start[r] = s + 1; id_len[r] = l ? l - 1 : 0;
        line_span(text, nl, m, n_bytes, has_cr, l0 + 1, s, l);
        seq_start[r] = s; seq_len[r] = l; mylen = l;
        if (fastq) {
            uint32_t ps, pl, qs, ql;
            line_span(text, nl, m, n_bytes, has_cr, l0 + 2, ps, pl);
            line_span(text, nl, m, n_bytes, has_cr, l0 + 3, qs, ql);
            if (pl == 0 || text[ps] != '+' || ql != l) atomicOr(bad, 2u);
            qual_start[r] = qs;
        }
    }
    for (int d = 32; d; d >>= 1) mylen = max(mylen, (uint32_t)__shfl_xor((int)mylen, d));
    // (one atomic per wave onto one address serialises the kernel: only a wave that can raise the maximum issues it)
    if ((threadIdx.x & 63) == 0 && mylen > *(volatile unsigned int *)max_len) atomicMax(max_len, mylen);
}

// byte-wide variant (text at an address that is not a multiple of four): one wave per 64 records, the wave walks its
// records one after the other and its lanes take consecutive bytes
__global__ __launch_bounds__(256) void record_gather_bytes_kernel(const char *__restrict__ text, uint64_t n_rec,
                                                            const uint32_t *__restrict__ seq_start, const uint32_t *__restrict__ qual_start,
                                                            const uint64_t *__restrict__ off, int fastq, int qoff, uint8_t *__restrict__ bases,
                                                            uint8_t *__restrict__ qual, unsigned int *__restrict__ bad)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t r0 = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (r0 >= n_rec) return;
    const uint64_t r = r0 + lane;
    const bool in = r < n_rec;
    const uint64_t my_o = in ? off[r] : 0;
    const uint32_t my_len = in ? (uint32_t)(off[r + 1] - my_o) : 0u;
    const uint32_t my_s = in ? seq_start[r] : 0u, my_q = (in && fastq) ? qual_start[r] : 0u;
    const uint32_t cnt = (uint32_t)min((uint64_t)64, n_rec - r0);
    bool space = false;
    // work items = (record of the wave, 64-byte chunk of it); four items' loads are in flight before their stores
    uint32_t maxlen = my_len;
    for (int d = 32; d; d >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, d));
    const uint32_t chunks = (maxlen + 63) >> 6, items = cnt * chunks;
    for (uint32_t t0 = 0; t0 < items; t0 += 4) {
        char c[4], d[4];
        uint64_t dst[4];
        bool on[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t t = t0 + u, k = min(t / chunks, cnt - 1), i = (t % chunks) * 64 + lane;
            const uint64_t o = __shfl(my_o, (int)k);
            const uint32_t len = __shfl(my_len, (int)k), sk = __shfl(my_s, (int)k), qk = __shfl(my_q, (int)k);
            on[u] = t < items && i < len;
            dst[u] = o + i;
            c[u] = on[u] ? text[sk + i] : 'A';
            d[u] = (on[u] && fastq) ? text[qk + i] : '!';
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!on[u]) continue;
            uint8_t v;
            switch (c[u]) { case 'A': v = 0; break; case 'C': v = 1; break; case 'G': v = 2; break; case 'T': v = 3; break; default: v = 4; }
            space = space || c[u] == ' ' || (c[u] >= '\t' && c[u] <= '\r') || d[u] == ' ' || (d[u] >= '\t' && d[u] <= '\r'); // isspace: the reference would skip it
            bases[dst[u]] = v;
            if (fastq) qual[dst[u]] = (uint8_t)(d[u] - qoff);
        }
    }
    if (space) atomicOr(bad, 4u);
}

// one wave per 64 records: their symbols and qualities go to one contiguous range of the batch arrays.  A record is
// handled by LPR lanes (32 for reads up to 128 bases, else 64), four bytes per lane: two aligned dword loads and a
// byte funnel bring the lane's four characters, they are mapped and stored as one dword where the destination allows
// it.  Four work items (record pairs x 4*LPR-byte chunks) are in flight before their stores.
__device__ __forceinline__ uint32_t map4(uint32_t x, bool &space)
{
    uint32_t out = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const uint32_t c = (x >> (8 * b)) & 0xffu;
        const uint32_t v = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
        space = space || c == ' ' || (c >= '\t' && c <= '\r'); // isspace: the reference would skip it
        out |= v << (8 * b);
    }
    return out;
}
__device__ __forceinline__ uint32_t sub4(uint32_t x, uint32_t qoff, bool &space)
{
    uint32_t out = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const uint32_t c = (x >> (8 * b)) & 0xffu;
        space = space || c == ' ' || (c >= '\t' && c <= '\r');
        out |= ((c - qoff) & 0xffu) << (8 * b);
    }
    return out;
}
// the four bytes at text + at (text 4-byte aligned); bytes at or behind n_bytes read as 0
__device__ __forceinline__ uint32_t load4(const char *__restrict__ text, uint64_t at, uint64_t n_bytes)
{
    const uint64_t a = at & ~(uint64_t)3;
    if (a + 8 <= n_bytes) {
        const uint32_t w0 = *reinterpret_cast<const uint32_t *>(text + a), w1 = *reinterpret_cast<const uint32_t *>(text + a + 4);
        return __builtin_amdgcn_alignbyte(w1, w0, (uint32_t)at & 3u);
    }
    uint32_t v = 0; // the last bytes of the text
    for (uint32_t b = 0; b < 4; ++b)
        if (at + b < n_bytes) v |= (uint32_t)(uint8_t)text[at + b] << (8 * b);
    return v;
}
__device__ __forceinline__ void store4(uint8_t *__restrict__ dst, uint64_t at, uint32_t v, uint32_t nvalid)
{
    if (nvalid >= 4 && (at & 3) == 0) { *reinterpret_cast<uint32_t *>(dst + at) = v; return; }
    for (uint32_t b = 0; b < nvalid && b < 4; ++b) dst[at + b] = (uint8_t)(v >> (8 * b));
}

__global__ __launch_bounds__(256) void record_gather_kernel(const char *__restrict__ text, uint64_t n_bytes, uint64_t n_rec,
                                                            const uint32_t *__restrict__ seq_start, const uint32_t *__restrict__ qual_start,
                                                            const uint64_t *__restrict__ off, int fastq, int qoff, uint8_t *__restrict__ bases,
                                                            uint8_t *__restrict__ qual, unsigned int *__restrict__ bad)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t r0 = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (r0 >= n_rec) return;
    const uint64_t r = r0 + lane;
    const bool in = r < n_rec;
    const uint64_t my_o = in ? off[r] : 0;
    const uint32_t my_len = in ? (uint32_t)(off[r + 1] - my_o) : 0u;
    const uint32_t my_s = in ? seq_start[r] : 0u, my_q = (in && fastq) ? qual_start[r] : 0u;
    bool space = false;
    uint32_t maxlen = my_len;
    for (int d = 32; d; d >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, d));
    const uint32_t lpr = maxlen <= 128 ? 32u : 64u;                 // lanes per record
    const uint32_t rpi = 64 / lpr;                                  // records per work item
    const uint32_t chunks = (maxlen + 4 * lpr - 1) / (4 * lpr);     // 4*lpr-byte chunks per record
    const uint32_t items = (64 / rpi) * chunks;
    const uint32_t sub = lane & (lpr - 1), which = lane / lpr;
    for (uint32_t t0 = 0; t0 < items; t0 += 4) {
        uint32_t c[4], d[4], nv[4];
        uint64_t dst[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t t = t0 + u, k = min((t / chunks) * rpi + which, 63u), i = ((t % chunks) * lpr + sub) * 4;
            const uint64_t o = __shfl(my_o, (int)k);
            const uint32_t len = __shfl(my_len, (int)k), sk = __shfl(my_s, (int)k), qk = __shfl(my_q, (int)k);
            nv[u] = (t < items && i < len) ? min(4u, len - i) : 0u;
            dst[u] = o + i;
            c[u] = nv[u] ? load4(text, (uint64_t)sk + i, n_bytes) : 0x41414141u;
            d[u] = (nv[u] && fastq) ? load4(text, (uint64_t)qk + i, n_bytes) : 0x21212121u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!nv[u]) continue;
            const uint32_t keep = nv[u] >= 4 ? 0xffffffffu : ((1u << (8 * nv[u])) - 1u);
            // (bytes behind the record -- its newline, the next field -- are not part of it: neutral characters)
            store4(bases, dst[u], map4((c[u] & keep) | (0x41414141u & ~keep), space), nv[u]);
            if (fastq) store4(qual, dst[u], sub4((d[u] & keep) | (0x21212121u & ~keep), (uint32_t)qoff, space), nv[u]);
        }
    }
    if (space) atomicOr(bad, 4u);
}

int rh_parse_reads(real_hip_ctx *ctx, const char *d_text, uint64_t n_bytes, int fastq, int qoff, real_hip_parsed *out)
{
    int rc;
    memset(out, 0, sizeof *out);
    out->struct_size = sizeof *out;
    if (!n_bytes) return REAL_HIP_OK;
    if (n_bytes >= 0xffffffffull) return rh_fail(ctx, REAL_HIP_E_UNSUPPORTED, "text chunk of 4 GiB or more", hipSuccess);
    // 1. newline positions
    if ((rc = rh_reserve(ctx, ctx->p_nl, (n_bytes + 1) * 4))) return rc; // (room for a text made of newlines only)
    if ((rc = rh_reserve(ctx, ctx->p_scal, 64))) return rc;
    RH_HIP(ctx, hipMemsetAsync(ctx->p_scal.p, 0, 64, ctx->stream));
    size_t *d_count = (size_t *)ctx->p_scal.p;
    unsigned int *d_bad = (unsigned int *)((uint8_t *)ctx->p_scal.p + 16), *d_max = (unsigned int *)((uint8_t *)ctx->p_scal.p + 24);
    unsigned int *d_cr = (unsigned int *)((uint8_t *)ctx->p_scal.p + 32);
    size_t m = 0;
    char last = 0;
    if (((uintptr_t)d_text & 15) == 0) {
        const uint64_t nblk = (n_bytes + NL_BLOCK_BYTES - 1) / NL_BLOCK_BYTES;
        if ((rc = rh_reserve(ctx, ctx->p_len1, (nblk + 1) * 4 * 2))) return rc;
        uint32_t *bc = (uint32_t *)ctx->p_len1.p, *bo = bc + nblk + 1;
        hipLaunchKernelGGL(nl_count_kernel, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, d_text, n_bytes, bc, d_cr);
        RH_HIP(ctx, hipMemsetAsync(bc + nblk, 0, 4, ctx->stream));
        size_t tmp = 0;
        RH_HIP(ctx, rocprim::exclusive_scan(nullptr, tmp, bc, bo, 0u, (size_t)(nblk + 1), rocprim::plus<uint32_t>(), ctx->stream));
        if ((rc = rh_reserve(ctx, ctx->sort_tmp, tmp ? tmp : 8))) return rc;
        RH_HIP(ctx, rocprim::exclusive_scan(ctx->sort_tmp.p, tmp, bc, bo, 0u, (size_t)(nblk + 1), rocprim::plus<uint32_t>(), ctx->stream));
        hipLaunchKernelGGL(nl_fill_kernel, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, d_text, n_bytes, (const uint32_t *)bo,
                           (uint32_t *)ctx->p_nl.p);
        uint32_t h_m = 0;
        RH_HIP(ctx, hipMemcpyAsync(&h_m, bo + nblk, 4, hipMemcpyDeviceToHost, ctx->stream));
        RH_HIP(ctx, hipMemcpyAsync(&last, d_text + n_bytes - 1, 1, hipMemcpyDeviceToHost, ctx->stream));
        RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        m = h_m;
    } else { // (text at an odd address: generic compaction; '\r' is then looked for at every line end)
        const unsigned int one = 1;
        RH_HIP(ctx, hipMemcpyAsync(d_cr, &one, 4, hipMemcpyHostToDevice, ctx->stream));
        IsNewline pred{d_text};
        size_t tmp = 0;
        RH_HIP(ctx, rocprim::select(nullptr, tmp, rocprim::counting_iterator<uint32_t>(0), (uint32_t *)ctx->p_nl.p, d_count,
                                    (size_t)n_bytes, pred, ctx->stream));
        if ((rc = rh_reserve(ctx, ctx->sort_tmp, tmp ? tmp : 8))) return rc;
        RH_HIP(ctx, rocprim::select(ctx->sort_tmp.p, tmp, rocprim::counting_iterator<uint32_t>(0), (uint32_t *)ctx->p_nl.p, d_count,
                                    (size_t)n_bytes, pred, ctx->stream));
        RH_HIP(ctx, hipMemcpyAsync(&m, d_count, sizeof(size_t), hipMemcpyDeviceToHost, ctx->stream));
        RH_HIP(ctx, hipMemcpyAsync(&last, d_text + n_bytes - 1, 1, hipMemcpyDeviceToHost, ctx->stream));
        RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    const uint64_t lines = m + (last == '\n' ? 0 : 1);
    const uint64_t lpr = fastq ? 4 : 2;
    if (lines % lpr) return rh_fail(ctx, REAL_HIP_E_UNSUPPORTED, "text is not whole records of one line per field", hipSuccess);
    const uint64_t n = lines / lpr;
    // 2. spans + form check
    if ((rc = rh_reserve(ctx, ctx->p_spans, (n ? n : 1) * 5 * 4))) return rc;
    uint32_t *seq_start = (uint32_t *)ctx->p_spans.p, *seq_len = seq_start + n, *qual_start = seq_len + n, *id_start = qual_start + n,
             *id_len = id_start + n;
    if ((rc = rh_reserve(ctx, ctx->p_off, (n + 1) * 8))) return rc;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(record_spans_kernel, grid, block, 0, ctx->stream, d_text, (const uint32_t *)ctx->p_nl.p, (uint64_t)m, n_bytes, n,
                       fastq, (const unsigned int *)d_cr, seq_start, seq_len, qual_start, id_start, id_len, d_bad, d_max);
    // 3. offsets = exclusive scan of the lengths (n + 1 values: the last is the total)
    {
        if ((rc = rh_reserve(ctx, ctx->p_len1, (n + 1) * 4))) return rc;
        RH_HIP(ctx, hipMemcpyAsync(ctx->p_len1.p, seq_len, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        RH_HIP(ctx, hipMemsetAsync((uint32_t *)ctx->p_len1.p + n, 0, 4, ctx->stream));
        rocprim::transform_iterator<const uint32_t *, LenToU64, uint64_t> in((const uint32_t *)ctx->p_len1.p, LenToU64());
        size_t tmp = 0;
        RH_HIP(ctx, rocprim::exclusive_scan(nullptr, tmp, in, (uint64_t *)ctx->p_off.p, (uint64_t)0, (size_t)(n + 1),
                                            rocprim::plus<uint64_t>(), ctx->stream));
        if ((rc = rh_reserve(ctx, ctx->sort_tmp, tmp ? tmp : 8))) return rc;
        RH_HIP(ctx, rocprim::exclusive_scan(ctx->sort_tmp.p, tmp, in, (uint64_t *)ctx->p_off.p, (uint64_t)0, (size_t)(n + 1),
                                            rocprim::plus<uint64_t>(), ctx->stream));
    }
    uint64_t total = 0;
    unsigned int h[4] = {0, 0, 0, 0};
    RH_HIP(ctx, hipMemcpyAsync(&total, (uint64_t *)ctx->p_off.p + n, 8, hipMemcpyDeviceToHost, ctx->stream));
    RH_HIP(ctx, hipMemcpyAsync(h, d_bad, 16, hipMemcpyDeviceToHost, ctx->stream));
    RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h[0]) return rh_fail(ctx, REAL_HIP_E_UNSUPPORTED, "text is not in one-line-per-field form (marker / '+' line / quality length)", hipSuccess);
    // 4. symbols and qualities
    if ((rc = rh_reserve(ctx, ctx->p_bases, total ? total : 1))) return rc;
    if (fastq && (rc = rh_reserve(ctx, ctx->p_qual, total ? total : 1))) return rc;
    if (((uintptr_t)d_text & 3) == 0)
        hipLaunchKernelGGL(record_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_text, n_bytes, n,
                           (const uint32_t *)seq_start, (const uint32_t *)qual_start, (const uint64_t *)ctx->p_off.p, fastq, qoff,
                           (uint8_t *)ctx->p_bases.p, (uint8_t *)ctx->p_qual.p, d_bad);
    else
        hipLaunchKernelGGL(record_gather_bytes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_text, n,
                           (const uint32_t *)seq_start, (const uint32_t *)qual_start, (const uint64_t *)ctx->p_off.p, fastq, qoff,
                           (uint8_t *)ctx->p_bases.p, (uint8_t *)ctx->p_qual.p, d_bad);
    RH_HIP(ctx, hipGetLastError());
    RH_HIP(ctx, hipMemcpyAsync(h, d_bad, 16, hipMemcpyDeviceToHost, ctx->stream));
    RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h[0]) return rh_fail(ctx, REAL_HIP_E_UNSUPPORTED, "white space inside a sequence or quality line", hipSuccess);
    out->n_reads = n; out->n_symbols = total; out->max_patl = h[2];
    out->bases = (const uint8_t *)ctx->p_bases.p;
    out->qual = fastq ? (const uint8_t *)ctx->p_qual.p : nullptr;
    out->offsets = (const uint64_t *)ctx->p_off.p;
    out->id_start = id_start; out->id_len = id_len;
    return REAL_HIP_OK;
}
