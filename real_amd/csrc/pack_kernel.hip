// pack_kernel.hip -- 2-bit packing of read batches (a1, a3, a4, a5).
#include "kernel_common.h"

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
// launchers
// ---------------------------------------------------------------------------
template <int W>
static void launch_pack_w(real_hip_ctx *ctx, hipStream_t st, const uint8_t *d_bases, const uint8_t *d_qual, const uint64_t *d_off,
                          uint32_t upatl, uint64_t n, uint32_t QS, const PackOut &out)
{
    dim3 grid((unsigned)((n + PACK_RB - 1) / PACK_RB)), block(PACK_RB);
    const size_t lds_bytes = (size_t)PACK_RB * 32 * W + 32; // the block's contiguous byte range + alignment skew
    // above the 64 KiB default only for W = 8; gfx950 has 160 KiB of LDS per CU
    (void)hipFuncSetAttribute((const void *)pack_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(pack_kernel<W>, grid, block, lds_bytes, st, d_bases, d_qual, d_off, upatl, n, ctx->prm.seedl, QS,
                       (int)(ctx->prm.scores != 0), out.words, out.seeds, out.qrows, out.patl);
}

int rh_launch_pack(real_hip_ctx *ctx, hipStream_t st, const uint8_t *d_bases, const uint8_t *d_qual, const uint64_t *d_off,
                   uint32_t upatl, uint64_t n, uint32_t W, uint32_t QS, const PackOut &out)
{
    if (!n) return REAL_HIP_OK;
    rh_time_begin(ctx, st, REAL_HIP_K_PACK);
    switch (W) {
    case 1: launch_pack_w<1>(ctx, st, d_bases, d_qual, d_off, upatl, n, QS, out); break;
    case 2: launch_pack_w<2>(ctx, st, d_bases, d_qual, d_off, upatl, n, QS, out); break;
    case 3: launch_pack_w<3>(ctx, st, d_bases, d_qual, d_off, upatl, n, QS, out); break;
    case 4: launch_pack_w<4>(ctx, st, d_bases, d_qual, d_off, upatl, n, QS, out); break;
    case 5: launch_pack_w<5>(ctx, st, d_bases, d_qual, d_off, upatl, n, QS, out); break;
    case 6: launch_pack_w<6>(ctx, st, d_bases, d_qual, d_off, upatl, n, QS, out); break;
    case 7: launch_pack_w<7>(ctx, st, d_bases, d_qual, d_off, upatl, n, QS, out); break;
    case 8: launch_pack_w<8>(ctx, st, d_bases, d_qual, d_off, upatl, n, QS, out); break;
    default: return rh_fail(ctx, REAL_HIP_E_UNSUPPORTED, "read longer than REAL_HIP_MAX_PATL", hipSuccess);
    }
    rh_time_end(ctx, st);
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
