// randline.hip -- micro-benchmark for the bucket index: a lane reads NQ x 16 bytes of ONE random 128-byte line
// with NQ loads issued back to back (no dependency between them).  Do the extra loads of the same line cost
// fabric requests, or are they merged with the first one (hit under miss)?
//   usage: randline <gb> <rounds> <nq: 1|2|4|8> [waves/simd]
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

template <int NQ>
__global__ void randline_kernel(const uint4 *__restrict__ buf, uint64_t nlines, int rounds, uint64_t *out)
{
    uint64_t s = mix(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1);
    uint64_t acc = 0;
    for (int r = 0; r < rounds; ++r) {
        uint4 v[2][NQ];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint64_t line = __umul64hi(mix(s + k * 0x632BE59BD9B4E019ull), nlines);
#pragma unroll
            for (int q = 0; q < NQ; ++q) v[k][q] = buf[line * 8 + q];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc += v[k][q].x + v[k][q].w;
        s = mix(s ^ acc);
    }
    if (acc == 0x1234567) out[0] = acc;
}

int main(int argc, char **argv)
{
    double gb = argc > 1 ? atof(argv[1]) : 64;
    int rounds = argc > 2 ? atoi(argv[2]) : 32;
    int nq = argc > 3 ? atoi(argv[3]) : 1;
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
        switch (nq) {
        case 1: hipLaunchKernelGGL(randline_kernel<1>, dim3(blocks), dim3(256), 0, 0, buf, nlines, rounds, out); break;
        case 2: hipLaunchKernelGGL(randline_kernel<2>, dim3(blocks), dim3(256), 0, 0, buf, nlines, rounds, out); break;
        case 4: hipLaunchKernelGGL(randline_kernel<4>, dim3(blocks), dim3(256), 0, 0, buf, nlines, rounds, out); break;
        default: hipLaunchKernelGGL(randline_kernel<8>, dim3(blocks), dim3(256), 0, 0, buf, nlines, rounds, out); break;
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        double lookups = (double)blocks * 256 * rounds * 2;
        if (it == 2) printf("gb=%.0f rounds=%d nq=%d waves/simd=%d : %.3f ms, %.2f G lines/s\n", gb, rounds, nq, wps, ms, lookups / ms / 1e6);
    }
    return 0;
}
