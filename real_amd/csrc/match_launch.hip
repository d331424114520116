// match_launch.hip -- launches of the per-read matcher: the lane-per-read kernel over the whole batch (one
// translation unit per read width, match_kernel.hip) and, behind it on the same stream, the wave-per-read kernel
// over the reads it handed over (match_wave.hip).
#include "real_hip_internal.h"

#define RH_DECL_W(N)                                                                    \
    void rh_launch_match_w##N(real_hip_ctx *ctx, const MatchArgs &a, bool all);         \
    void rh_launch_match2_w##N(real_hip_ctx *ctx, const MatchArgs &a, bool all);        \
    uint32_t rh_stage_bytes_w##N(int tk);
RH_DECL_W(1) RH_DECL_W(2) RH_DECL_W(3) RH_DECL_W(4) RH_DECL_W(5) RH_DECL_W(6) RH_DECL_W(7) RH_DECL_W(8) RH_DECL_W(9) RH_DECL_W(10)
void rh_launch_match_wave(real_hip_ctx *ctx, const MatchArgs &a, bool all);

int rh_launch_match(real_hip_ctx *ctx, const MatchArgs &args, bool all, int state_slot)
{
    if (!args.b.n_reads) return REAL_HIP_OK;
    MatchArgs a = args;
    typedef void (*launch_fn)(real_hip_ctx *, const MatchArgs &, bool);
    typedef uint32_t (*stage_fn)(int);
    static const launch_fn launch[RH_MAXW] = {rh_launch_match_w1, rh_launch_match_w2, rh_launch_match_w3, rh_launch_match_w4, rh_launch_match_w5,
                                              rh_launch_match_w6, rh_launch_match_w7, rh_launch_match_w8, rh_launch_match_w9, rh_launch_match_w10};
    static const launch_fn launch2[RH_MAXW] = {rh_launch_match2_w1, rh_launch_match2_w2, rh_launch_match2_w3, rh_launch_match2_w4, rh_launch_match2_w5,
                                               rh_launch_match2_w6, rh_launch_match2_w7, rh_launch_match2_w8, rh_launch_match2_w9, rh_launch_match2_w10};
    static const stage_fn stage[RH_MAXW] = {rh_stage_bytes_w1, rh_stage_bytes_w2, rh_stage_bytes_w3, rh_stage_bytes_w4, rh_stage_bytes_w5,
                                            rh_stage_bytes_w6, rh_stage_bytes_w7, rh_stage_bytes_w8, rh_stage_bytes_w9, rh_stage_bytes_w10};
    if (a.b.W < 1 || a.b.W > RH_MAXW) return rh_fail(ctx, REAL_HIP_E_UNSUPPORTED, "read longer than REAL_HIP_MAX_PATL", hipSuccess);
    int rc;
    { // reads a wave stages at a time: their bytes (+ alignment skew, pad, one dword of over-read) fit its LDS region
        const uint32_t maxlen = (a.b.off || a.b.upatl > 32u * a.b.W) ? 32u * a.b.W : a.b.upatl;
        const uint32_t region = stage[a.b.W - 1](a.ix.fine == 3 ? 3 : (a.ix.fine ? 1 : 0));
        uint32_t gl = 64;
        while (gl > 1 && (uint64_t)gl * maxlen + 16 + 32 > region) gl >>= 1;
        a.b.gl = gl;
    }
    // hand-over lists: first pass -> second pass (ovf_list), -> wave-per-read kernel (ovf2_list).  The words of ovf_count:
    // [0] length of ovf_list, [1] error flags, [2] tile counter of the first pass, [3] of the second, [4] length of ovf2_list
    if ((rc = rh_reserve(ctx, ctx->ovf_list, a.b.n_reads * 4))) return rc;
    if ((rc = rh_reserve(ctx, ctx->ovf2_list, a.b.n_reads * 4))) return rc;
    if ((rc = rh_reserve(ctx, ctx->ovf_count, 64))) return rc;
    unsigned long long *oc = (unsigned long long *)ctx->ovf_count.p;
    a.ovf_list = (uint32_t *)ctx->ovf_list.p;
    a.ovf_count = oc;
    a.err_flags = (uint32_t *)(oc + 1);
    a.tile_ctr = (uint32_t *)(oc + 2);
    a.tile_ctr2 = (uint32_t *)(oc + 3);
    a.ovf2_list = (uint32_t *)ctx->ovf2_list.p;
    a.ovf2_count = oc + 4;
    RH_HIP(ctx, hipMemsetAsync(ctx->ovf_count.p, 0, 64, ctx->stream));
    rh_time_begin(ctx, ctx->stream, all ? REAL_HIP_K_MATCH_ALL : REAL_HIP_K_MATCH_UNIQUE);
    launch[a.b.W - 1](ctx, a, all);
    rh_time_end(ctx, ctx->stream);
    RH_HIP(ctx, hipGetLastError());
    rh_time_begin(ctx, ctx->stream, REAL_HIP_K_MATCH_REPEAT);
    launch2[a.b.W - 1](ctx, a, all);
    rh_launch_match_wave(ctx, a, all);
    rh_time_end(ctx, ctx->stream);
    RH_HIP(ctx, hipGetLastError());
    RH_HIP(ctx, hipMemcpyAsync(ctx->h_state + 2 * state_slot, ctx->ovf_count.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    return REAL_HIP_OK;
}

// after the stream has been synchronised: what the kernels of the last launch reported
int rh_match_finish(real_hip_ctx *ctx, int state_slot)
{
    const unsigned long long flags = ctx->h_state[2 * state_slot + 1];
    ctx->h_state[2 * state_slot + 1] = 0;
    if (flags & 1u)
        return rh_fail(ctx, REAL_HIP_E_INVALID, "a read of the batch is longer than REAL_HIP_MAX_PATL_LONG, or the device offsets are not monotone", hipSuccess);
    return REAL_HIP_OK;
}
