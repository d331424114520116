// batch_kernel.hip -- small helpers over a read batch that lives on the device.
#include "kernel_common.h"

// longest read of a ragged device batch (decides the register form of the matcher)
__global__ void max_patl_kernel(const uint64_t *__restrict__ off, uint64_t n, uint32_t *out)
{
    uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t v = 0;
    if (r < n) v = (uint32_t)(off[r + 1] - off[r]);
    for (int d = 32; d; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    if ((threadIdx.x & 63) == 0) atomicMax(out, v);
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

// 2-bit packed bases -> one symbol per byte (what the matcher stages through LDS): base g of the batch sits at bits
// 7-2(g%4)-1.. of byte g/4 (MSB first, like the reference's packed text and its rewritten read files,
// TemporaryFile.hpp:335-373).  One thread expands four packed bytes into sixteen symbols (one 16-byte store).
__global__ void unpack_bases_kernel(const uint8_t *__restrict__ pk, uint64_t n_sym, uint8_t *__restrict__ out)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t g0 = t * 16;
    if (g0 >= n_sym) return;
    const uint64_t nbytes = (n_sym + 3) / 4;
    uint32_t w = 0;
    if (4 * t + 4 <= nbytes && (((uintptr_t)pk) & 3) == 0) w = reinterpret_cast<const uint32_t *>(pk)[t];
    else
        for (int k = 0; k < 4; ++k) if (4 * t + k < nbytes) w |= (uint32_t)pk[4 * t + k] << (8 * k);
    uint32_t d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t b = (w >> (8 * k)) & 0xffu;
        d[k] = (b >> 6) | (((b >> 4) & 3u) << 8) | (((b >> 2) & 3u) << 16) | ((b & 3u) << 24);
    }
    *reinterpret_cast<uint4 *>(out + g0) = make_uint4(d[0], d[1], d[2], d[3]); // (the buffer is padded by 16 bytes)
}
// reads the caller flagged as holding a symbol > 3 (they cannot be packed): their first symbol becomes 4, which makes
// the matcher skip them as the reference does (matchUniqueImplementation.cpp:376-394)
__global__ void apply_nflags_kernel(const uint8_t *__restrict__ nflags, const uint64_t *__restrict__ off, uint32_t upatl, uint64_t n, uint8_t *__restrict__ out)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n || !((nflags[r >> 3] >> (r & 7)) & 1)) return;
    const uint64_t o0 = off ? off[r] : r * (uint64_t)upatl, o1 = off ? off[r + 1] : o0 + upatl;
    if (o1 > o0) out[o0] = 4;
}

int rh_unpack_bases(real_hip_ctx *ctx, const uint8_t *d_packed, uint64_t n_sym, const uint8_t *d_nflags, const uint64_t *d_off,
                    uint32_t upatl, uint64_t n_reads, uint8_t *d_out)
{
    const uint64_t threads = (n_sym + 15) / 16;
    if (threads) hipLaunchKernelGGL(unpack_bases_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, d_packed, n_sym, d_out);
    if (d_nflags && n_reads)
        hipLaunchKernelGGL(apply_nflags_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, ctx->stream, d_nflags, d_off, upatl, n_reads, d_out);
    RH_HIP(ctx, hipGetLastError());
    return REAL_HIP_OK;
}
