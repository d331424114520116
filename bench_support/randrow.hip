// randrow.hip -- micro-benchmark for lookups by lane groups: every group of 8 lanes reads one random 128-byte line,
// 16 bytes per lane, with one load instruction (8 lines per wave instruction), MLP independent instructions per round.
//   usage: randrow <gb> <rounds> <mlp: 1|2|4|8> [waves/simd]
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

template <int MLP>
__global__ void randrow_kernel(const uint4 *__restrict__ buf, uint64_t nlines, int rounds, uint64_t *out)
{
    const uint32_t lane = threadIdx.x & 63;
    uint64_t s = mix(((uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~7u)) * 0x9E3779B97F4A7C15ull + 1); // one stream per group of 8
    uint64_t acc = 0;
    for (int r = 0; r < rounds; ++r) {
        uint4 v[MLP];
#pragma unroll
        for (int k = 0; k < MLP; ++k) {
            const uint64_t line = __umul64hi(mix(s + k * 0x632BE59BD9B4E019ull), nlines);
            v[k] = buf[line * 8 + (lane & 7)];
        }
#pragma unroll
        for (int k = 0; k < MLP; ++k) acc += v[k].x + v[k].w;
        // the group's next addresses depend on what it read
        uint32_t a = (uint32_t)acc;
        a += __shfl_xor((int)a, 1); a += __shfl_xor((int)a, 2); a += __shfl_xor((int)a, 4);
        s = mix(s ^ a);
    }
    if (acc == 0x1234567) out[0] = acc;
}

int main(int argc, char **argv)
{
    double gb = argc > 1 ? atof(argv[1]) : 64;
    int rounds = argc > 2 ? atoi(argv[2]) : 32;
    int mlp = argc > 3 ? atoi(argv[3]) : 4;
    int wps = argc > 4 ? atoi(argv[4]) : 3;
    uint64_t nlines = (uint64_t)(gb * (1ull << 30)) / 128;
    uint4 *buf; uint64_t *out;
    if (hipMalloc(&buf, nlines * 128) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 16);
    hipMemset(buf, 1, nlines * 128);
    int blocks = 256 * wps * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        switch (mlp) {
        case 1: hipLaunchKernelGGL(randrow_kernel<1>, dim3(blocks), dim3(256), 0, 0, buf, nlines, rounds, out); break;
        case 2: hipLaunchKernelGGL(randrow_kernel<2>, dim3(blocks), dim3(256), 0, 0, buf, nlines, rounds, out); break;
        case 4: hipLaunchKernelGGL(randrow_kernel<4>, dim3(blocks), dim3(256), 0, 0, buf, nlines, rounds, out); break;
        default: hipLaunchKernelGGL(randrow_kernel<8>, dim3(blocks), dim3(256), 0, 0, buf, nlines, rounds, out); break;
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        double lines = (double)blocks * 32 * rounds * mlp; // 32 groups of 8 lanes per block
        if (it == 2) printf("gb=%.0f rounds=%d mlp=%d waves/simd=%d : %.3f ms, %.2f G lines/s = %.2f TB/s\n", gb, rounds, mlp, wps, ms, lines / ms / 1e6, lines * 128 / ms / 1e9);
    }
    return 0;
}
