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

