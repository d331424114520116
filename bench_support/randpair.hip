// randpair.hip -- micro-benchmark for the line-bucket index: a lane reads 8 bytes of a random 128-byte line
// (the bucket header), then -- dependent on that data -- 8 more bytes of the SAME line.  Is the second read an
// L2 hit (cheap) or has the line already been evicted by the random traffic of the other waves?
//   usage: randpair <gb> <rounds> <mode: 0 header only | 1 second read in the other 64-B half of the line | 2 in another random line | 3 in the same 64-B half> [waves/simd]
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

template <int MODE>
__global__ void randpair_kernel(const uint64_t *__restrict__ buf, uint64_t nlines, int rounds, uint64_t *out)
{
    uint64_t s = mix(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1);
    uint64_t acc = 0;
    for (int r = 0; r < rounds; ++r) {
        uint64_t v[6], line[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            line[k] = __umul64hi(mix(s + k * 0x632BE59BD9B4E019ull), nlines);
            v[k] = buf[line[k] * 16];
        }
        if (MODE) {
            uint64_t w[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const uint64_t l2 = (MODE == 2) ? __umul64hi(mix(s + v[k] + k), nlines) : line[k];
                // MODE 1: the other 64-byte half of the line; MODE 3: the same half as the header
                w[k] = buf[l2 * 16 + (MODE == 3 ? 1 + ((v[k] + k) % 7) : 8 + ((v[k] + k) & 7))];
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) acc += w[k];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) acc += v[k];
        s = mix(s ^ acc);
    }
    if (acc == 0x1234567) out[0] = acc;
}

int main(int argc, char **argv)
{
    double gb = argc > 1 ? atof(argv[1]) : 64;
    int rounds = argc > 2 ? atoi(argv[2]) : 32;
    int mode = argc > 3 ? atoi(argv[3]) : 1;
    int wps = argc > 4 ? atoi(argv[4]) : 3;
    uint64_t nlines = (uint64_t)(gb * (1ull << 30)) / 128;
    uint64_t *buf, *out;
    if (hipMalloc(&buf, nlines * 128) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 16);
    hipMemset(buf, 1, nlines * 128);
    int blocks = 256 * wps * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(randpair_kernel<0>, dim3(blocks), dim3(256), 0, 0, buf, nlines, rounds, out);
        else if (mode == 1) hipLaunchKernelGGL(randpair_kernel<1>, dim3(blocks), dim3(256), 0, 0, buf, nlines, rounds, out);
        else if (mode == 2) hipLaunchKernelGGL(randpair_kernel<2>, dim3(blocks), dim3(256), 0, 0, buf, nlines, rounds, out);
        else hipLaunchKernelGGL(randpair_kernel<3>, dim3(blocks), dim3(256), 0, 0, buf, nlines, rounds, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        double lookups = (double)blocks * 256 * rounds * 6;
        if (it == 2) printf("gb=%.0f rounds=%d mode=%d waves/simd=%d : %.3f ms, %.2f G lookups/s\n", gb, rounds, mode, wps, ms, lookups / ms / 1e6);
    }
    return 0;
}
