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

// start and length of line j of the chunk (without its '\n' and a '\r' in front of it); line m (after the last
// newline) exists when the chunk does not end in a newline
__device__ __forceinline__ void line_span(const char *__restrict__ text, const uint32_t *__restrict__ nl, uint64_t m, uint64_t n_bytes,
                                          uint64_t j, uint32_t &start, uint32_t &len)
{
    const uint32_t s = j ? nl[j - 1] + 1 : 0u;
    uint32_t e = j < m ? nl[j] : (uint32_t)n_bytes;
    if (e > s && text[e - 1] == '\r') e--;
    start = s; len = e - s;
}

// one thread per record: field spans, form check, length
__global__ void record_spans_kernel(const char *__restrict__ text, const uint32_t *__restrict__ nl, uint64_t m, uint64_t n_bytes,
                                    uint64_t n_rec, int fastq, uint32_t *__restrict__ seq_start, uint32_t *__restrict__ seq_len,
                                    uint32_t *__restrict__ qual_start, uint32_t *__restrict__ id_start, uint32_t *__restrict__ id_len,
                                    unsigned int *__restrict__ bad, unsigned int *__restrict__ max_len)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t mylen = 0;
    if (r < n_rec) {
        const uint64_t l0 = r * (fastq ? 4 : 2);
        uint32_t s, l;
        line_span(text, nl, m, n_bytes, l0, s, l);
        if (l == 0 || text[s] != (fastq ? '@' : '>')) atomicOr(bad, 1u);
        id_start[r] = s + 1; id_len[r] = l ? l - 1 : 0;
        line_span(text, nl, m, n_bytes, l0 + 1, s, l);
        seq_start[r] = s; seq_len[r] = l; mylen = l;
        if (fastq) {
            uint32_t ps, pl, qs, ql;
            line_span(text, nl, m, n_bytes, l0 + 2, ps, pl);
            line_span(text, nl, m, n_bytes, l0 + 3, qs, ql);
            if (pl == 0 || text[ps] != '+' || ql != l) atomicOr(bad, 2u);
            qual_start[r] = qs;
        }
    }
    for (int d = 32; d; d >>= 1) mylen = max(mylen, (uint32_t)__shfl_xor((int)mylen, d));
    if ((threadIdx.x & 63) == 0 && mylen) atomicMax(max_len, mylen);
}

// one thread per record: symbols and qualities to their place in the batch arrays
__global__ void record_gather_kernel(const char *__restrict__ text, uint64_t n_rec, const uint32_t *__restrict__ seq_start,
                                     const uint32_t *__restrict__ qual_start, const uint64_t *__restrict__ off, int fastq, int qoff,
                                     uint8_t *__restrict__ bases, uint8_t *__restrict__ qual, unsigned int *__restrict__ bad)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rec) return;
    const uint64_t o = off[r];
    const uint32_t len = (uint32_t)(off[r + 1] - o);
    const char *s = text + seq_start[r];
    bool space = false;
    for (uint32_t i = 0; i < len; ++i) {
        const char c = s[i];
        uint8_t v;
        switch (c) { case 'A': v = 0; break; case 'C': v = 1; break; case 'G': v = 2; break; case 'T': v = 3; break; default: v = 4; }
        space = space || c == ' ' || (c >= '\t' && c <= '\r'); // isspace: the reference would skip it
        bases[o + i] = v;
    }
    if (fastq) {
        const char *q = text + qual_start[r];
        for (uint32_t i = 0; i < len; ++i) {
            const char c = q[i];
            space = space || c == ' ' || (c >= '\t' && c <= '\r');
            qual[o + i] = (uint8_t)(c - qoff);
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
    {
        IsNewline pred{d_text};
        size_t tmp = 0;
        RH_HIP(ctx, rocprim::select(nullptr, tmp, rocprim::counting_iterator<uint32_t>(0), (uint32_t *)ctx->p_nl.p, d_count,
                                    (size_t)n_bytes, pred, ctx->stream));
        if ((rc = rh_reserve(ctx, ctx->sort_tmp, tmp ? tmp : 8))) return rc;
        RH_HIP(ctx, rocprim::select(ctx->sort_tmp.p, tmp, rocprim::counting_iterator<uint32_t>(0), (uint32_t *)ctx->p_nl.p, d_count,
                                    (size_t)n_bytes, pred, ctx->stream));
    }
    size_t m = 0;
    char last = 0;
    RH_HIP(ctx, hipMemcpyAsync(&m, d_count, sizeof(size_t), hipMemcpyDeviceToHost, ctx->stream));
    RH_HIP(ctx, hipMemcpyAsync(&last, d_text + n_bytes - 1, 1, hipMemcpyDeviceToHost, ctx->stream));
    RH_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
                       fastq, seq_start, seq_len, qual_start, id_start, id_len, d_bad, d_max);
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
    hipLaunchKernelGGL(record_gather_kernel, grid, block, 0, ctx->stream, d_text, n, (const uint32_t *)seq_start, (const uint32_t *)qual_start,
                       (const uint64_t *)ctx->p_off.p, fastq, qoff, (uint8_t *)ctx->p_bases.p, (uint8_t *)ctx->p_qual.p, d_bad);
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
