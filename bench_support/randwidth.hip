// randwidth.hip -- random per-lane loads of 4 / 8 / 16 bytes (one dependent chain per lane):
// does the request rate depend on the width?   usage: randwidth <gb> <rounds> <bytes 4|8|16> <waves_per_simd>
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
template <typename T>
__global__ void k(const T *__restrict__ buf, uint64_t n, int rounds, uint64_t *out)
{
    uint64_t s = mix(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1), acc = 0;
    for (int r = 0; r < rounds; ++r) {
        T v = buf[__umul64hi(mix(s), n)];
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&v);
        for (unsigned i = 0; i < sizeof(T) / 4; ++i) acc += w[i];
        s = mix(s ^ acc);
    }
    if (acc == 0x1234567) out[0] = acc;
}
int main(int argc, char **argv)
{
    double gb = argc > 1 ? atof(argv[1]) : 64; int rounds = argc > 2 ? atoi(argv[2]) : 256; int bytes = argc > 3 ? atoi(argv[3]) : 8; int wps = argc > 4 ? atoi(argv[4]) : 3;
    uint64_t nb = (uint64_t)(gb * (1ull << 30)); void *buf; uint64_t *out;
    if (hipMalloc(&buf, nb) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, nb); (void)hipMalloc(&out, 16);
    int blocks = 256 * wps; hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(e0);
        if (bytes == 4) hipLaunchKernelGGL(k<uint32_t>, dim3(blocks), dim3(256), 0, 0, (const uint32_t *)buf, nb / 4, rounds, out);
        else if (bytes == 8) hipLaunchKernelGGL(k<uint2>, dim3(blocks), dim3(256), 0, 0, (const uint2 *)buf, nb / 8, rounds, out);
        else hipLaunchKernelGGL(k<uint4>, dim3(blocks), dim3(256), 0, 0, (const uint4 *)buf, nb / 16, rounds, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        double acc = (double)blocks * 256 * rounds;
        if (it == 2) printf("%d-byte random loads, %.0f GiB, waves/simd=%d : %.3f ms, %.2f G requests/s, round trip %.2f us\n", bytes, gb, wps, ms, acc / ms / 1e6, ms * 1e3 / rounds);
    }
    return 0;
}
