// randrow2.hip -- what short-lived workgroups cost a lookup kernel: the access pattern of match_lists_rows (a group of 8
// lanes reads one random 128-byte row with one 16-byte load per lane, 8 independent loads per wave and round = 64 rows in
// flight per wave, `rounds` rounds per wave) over `ntab` tables of gb/ntab GiB, launched either as many short workgroups
// (rounds = 12, as the matcher's: one per 256 reads) or as a few long-lived ones; optionally every wave first reads its
// own 1600 bytes of a sequential array and derives its addresses from them (the matcher's front: bases -> signatures).
//   usage: randrow2 <gb> <ntab> <blocks> <rounds> <lds_kib> <front 0|1>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>

__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct Tabs { const uint4 *t[8]; };

__global__ __launch_bounds__(256) void k(Tabs T, int ntab, uint64_t nlines, int rounds, const uint4 *seq, int front, uint64_t *out)
{
    extern __shared__ uint4 lds[];
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint64_t s = mix((wave * 64 + (lane & ~7u)) * 0x9E3779B97F4A7C15ull + 1);
    if (front) { // 1600 bytes per wave, sequential; the addresses depend on them
        uint4 v = seq[wave * 100 + lane];
        uint4 w = lane < 36 ? seq[wave * 100 + 64 + lane] : make_uint4(0, 0, 0, 0);
        uint32_t a = v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.w;
        a += __shfl_xor((int)a, 1); a += __shfl_xor((int)a, 2); a += __shfl_xor((int)a, 4);
        s = mix(s ^ a);
    }
    uint64_t acc = 0;
    for (int r = 0; r < rounds; ++r) {
        uint4 v[8];
        const uint4 *tab = T.t[r % ntab];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint64_t line = __umul64hi(mix(s + q * 0x632BE59BD9B4E019ull + r), nlines);
            v[q] = tab[line * 8 + (lane & 7)];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += v[q].x + v[q].w;
    }
    if (acc == 0x1234567) { out[0] = acc; lds[threadIdx.x] = make_uint4(1, 2, 3, 4); }
}

int main(int argc, char **argv)
{
    double gb = argc > 1 ? atof(argv[1]) : 64;
    int ntab = argc > 2 ? atoi(argv[2]) : 6;
    int blocks = argc > 3 ? atoi(argv[3]) : 195313;
    int rounds = argc > 4 ? atoi(argv[4]) : 12;
    int lds_kib = argc > 5 ? atoi(argv[5]) : 51;
    int front = argc > 6 ? atoi(argv[6]) : 0;
    uint64_t nlines = (uint64_t)(gb / ntab * (1ull << 30)) / 128;
    Tabs T;
    for (int i = 0; i < ntab; ++i) {
        void *p;
        if (hipMalloc(&p, nlines * 128) != hipSuccess) { printf("alloc failed\n"); return 1; }
        hipMemset(p, 1, nlines * 128);
        T.t[i] = (const uint4 *)p;
    }
    uint4 *seq; uint64_t *out;
    hipMalloc(&seq, (size_t)blocks * 4 * 1600 + 4096);
    hipMemset(seq, 3, (size_t)blocks * 4 * 1600 + 4096);
    hipMalloc(&out, 16);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kib * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds_kib * 1024, 0, T, ntab, nlines, rounds, seq, front, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        double lines = (double)blocks * 32 * rounds * 8;
        if (it == 2) printf("gb=%.0f ntab=%d blocks=%d rounds=%d lds=%dKiB front=%d : %.3f ms, %.2f G lines/s = %.2f TB/s  (%s)\n", gb, ntab, blocks, rounds, lds_kib, front, ms,
                            lines / ms / 1e6, lines * 128 / ms / 1e9, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
